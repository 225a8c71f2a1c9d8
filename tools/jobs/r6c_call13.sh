#!/bin/bash
# where the classifier's update runs: in line (0), forked at the top of the step (1, default), forked behind the generator's update (2)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
REPS=3 bash tools/ab_envs.sh r6c_c13 "CRANK_AMD_OVERLAP_C=0" "CRANK_AMD_OVERLAP_C=1" "CRANK_AMD_OVERLAP_C=2"
