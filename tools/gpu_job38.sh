#!/bin/bash
cd $GRAFT_REPO_ROOT
for tp in 1 2; do echo "CRK_VQ_TP=$tp"; CRK_VQ_TP=$tp timeout 300 python tools/time_vq.py 2>&1 | tail -5; done
timeout 600 python -m pytest tests/test_gpu_ops.py -x -q -m gpu -k "vq or quantizer or ema" 2>&1 | tail -3
for tp in 1 2; do CRK_VQ_TP=$tp timeout 300 python bench.py --steps 200 --warmup 20 2>&1 | tail -1 | cut -c1-200; done
