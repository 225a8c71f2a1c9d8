#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
for i in 1 2 3; do timeout 600 python -m pytest tests/test_gpu_dp.py -m gpu -q -p no:cacheprovider -x -k "rccl" > $OUT/r3_s26_dp_$i.log 2>&1; tail -2 $OUT/r3_s26_dp_$i.log; grep -E "HIP error|hip error|Error|terminate|what\(\)" $OUT/r3_s26_dp_$i.log | head -5; done
