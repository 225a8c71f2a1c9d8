#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "logmel" 2>&1 | tail -2
rm -f $O/c7_logmel.txt
for lib in libcrank_hip.so; do
  echo "== $lib" >> $O/c7_logmel.txt
  CRANK_AMD_LIB=$PWD/crank_amd/$lib timeout 120 python tools/prof_logmel.py 2>&1 | grep -v amdgpu.ids >> $O/c7_logmel.txt
done
cat $O/c7_logmel.txt
