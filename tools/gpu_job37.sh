#!/bin/bash
cd $GRAFT_REPO_ROOT
./tools/probe/mfma_f32 | tee gpurun_out/r3_s37_mfma_f32.txt
