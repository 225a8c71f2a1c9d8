#!/bin/bash
# the step's own stream as a high-priority stream
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
REPS=3 bash tools/ab_envs.sh r6c_c14 "CRANK_AMD_MAIN_PRIORITY=0" "CRANK_AMD_MAIN_PRIORITY=1"
