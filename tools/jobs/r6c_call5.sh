#!/bin/bash
# runtime knobs and the plain convs' chunks-per-group on the replayed step
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp
REPS=2 bash tools/ab_envs.sh r6c_c5 "-" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=0" "DEBUG_HIP_FORCE_GRAPH_QUEUES=1" "DEBUG_HIP_FORCE_GRAPH_QUEUES=2" "ROC_USE_FGS_KERNARG=0" "CRK_WG_CPG=16" "CRK_WG_CPG=12" "CRK_WG_CPG=32"
