#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "vq or codebook or quantizer or ema" > $O/c5_vq_tests.log 2>&1; tail -5 $O/c5_vq_tests.log
timeout 200 python tools/vq_phase_cycles.py > $O/c5_vq_phases.txt 2>&1; cat $O/c5_vq_phases.txt | grep -v amdgpu.ids
