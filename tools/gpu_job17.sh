#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_ops.py -m gpu -q -p no:cacheprovider -k "recon or stft or feature_losses" > $OUT/r3_s17_ops.log 2>&1; tail -4 $OUT/r3_s17_ops.log
bash tools/gpu_job16.sh | grep -E "recon_fwd|case"
