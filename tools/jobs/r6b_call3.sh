#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
CRANK_AMD_LIB=$PWD/crank_amd/libcrank_hip_prof_f.so timeout 200 python tools/store_cost_phases.py fwd 2>&1 | grep -v -i warn | tee $O/r6b_c3_store_phases.txt
