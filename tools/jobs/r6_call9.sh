#!/bin/bash
# round 6: the whole GPU suite on the pruned tree, twice (a sporadic abort was seen once in call 3), then the evidence session
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
for rep in 1 2; do
  timeout 1700 python -m pytest tests -m gpu -x -q > $O/r6_c9_pytest_gpu_$rep.log 2>&1; echo "pytest rc=$?" >> $O/r6_c9_pytest_gpu_$rep.log; tail -3 $O/r6_c9_pytest_gpu_$rep.log
done
