#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -q -p no:cacheprovider -k "bit_for_bit or graph_replayed or two_batch_shapes" > $OUT/r3_s28_tests.log 2>&1; tail -12 $OUT/r3_s28_tests.log
