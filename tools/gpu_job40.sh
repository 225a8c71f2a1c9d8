#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 120 ./tools/probe/cndmask | tee gpurun_out/r3_s40_cndmask.txt
