#!/bin/bash
# groups of chunks for the generator stacks' plain convs (first conv / head weight gradients, one launch for the four stacks)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
bash tools/ab_libs.sh r6c_c12 "pstack_wgrad|wnorm" $PWD/crank_amd/libcrank_hip.so $PWD/crank_amd/libcrank_hip_g0_43.so $PWD/crank_amd/libcrank_hip_g0_64.so $PWD/crank_amd/libcrank_hip_g0_22.so
