#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 300 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "logmel" 2>&1 | tail -15
timeout 600 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "raw" 2>&1 | tail -3
rm -f $O/c11_logmel.txt
for v in 0 1; do
  echo "== CRK_LOGMEL_WAVE=$v" >> $O/c11_logmel.txt
  CRK_LOGMEL_WAVE=$v timeout 120 python tools/prof_logmel.py 2>&1 | grep -v amdgpu.ids >> $O/c11_logmel.txt
  rm -rf /tmp/lm; CRK_LOGMEL_WAVE=$v timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/lm -- python tools/prof_logmel.py 20 > /dev/null 2>&1
  python tools/kstats.py /tmp/lm logmel >> $O/c11_logmel.txt
done
cat $O/c11_logmel.txt
