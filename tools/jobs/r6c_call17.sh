#!/bin/bash
# per-chunk tables of the EMA statistics: how many chunks
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
bash tools/ab_libs.sh r6c_c17 "vq_ema" $PWD/crank_amd/libcrank_hip.so $PWD/crank_amd/libcrank_hip_emac32.so $PWD/crank_amd/libcrank_hip_emac48.so $PWD/crank_amd/libcrank_hip_emac96.so
