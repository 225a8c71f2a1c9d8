#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/sa2 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sa2 -- python $GRAFT_REPO_ROOT/tools/prof_stacks_alone.py 8 > /tmp/sa.log 2>&1; python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/sa2 | head -12 )
