#!/bin/bash
# the subnormal term of the key bound (VQH_KEY_ABS): the test case with it, and the negative control without it
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 30 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "zero_rows_zero_codes or zeros_and_padding or duplicates" -p no:cacheprovider > gpurun_out/vqabs.log 2>&1; tail -3 gpurun_out/vqabs.log
CRANK_AMD_LIB=$PWD/crank_amd/libcrank_hip_noabs.so timeout 30 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "zero_rows_zero_codes" -p no:cacheprovider > gpurun_out/vqabs_control.log 2>&1; grep -E "passed|failed|assert|AssertionError" gpurun_out/vqabs_control.log | tail -4
