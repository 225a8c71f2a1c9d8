#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
for d in 0 2 4 6; do
( cd /tmp && rm -rf /tmp/sa2 && CRK_S2B_DBG=$d timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sa2 -- python $GRAFT_REPO_ROOT/tools/prof_stacks_alone.py 8 > /tmp/sa.log 2>&1; echo "dbg=$d"; python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/sa2 stack2_bwd )
done
