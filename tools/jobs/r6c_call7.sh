#!/bin/bash
# weight-norm backward of the generator: band rows / loads in flight
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
bash tools/ab_libs.sh r6c_c7 "wnorm" $PWD/crank_amd/libcrank_hip.so $PWD/crank_amd/libcrank_hip_wnif32.so $PWD/crank_amd/libcrank_hip_wnrb4.so $PWD/crank_amd/libcrank_hip_wnrb4if32.so
