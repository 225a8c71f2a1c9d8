#!/bin/bash
# who has the SIMD's priority when: frame half 1 throughout (0), nobody (1), handed to frame half 0 inside the block (2, 3, 4)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
bash tools/ab_libs.sh r6c_c11 "stack2_fwd" $PWD/crank_amd/libcrank_hip.so $PWD/crank_amd/libcrank_hip_prio1.so $PWD/crank_amd/libcrank_hip_prio2.so $PWD/crank_amd/libcrank_hip_prio3.so $PWD/crank_amd/libcrank_hip_prio4.so
