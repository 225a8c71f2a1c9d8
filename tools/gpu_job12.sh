#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 2400 python -m pytest tests -m gpu -q -p no:cacheprovider -k "lsgan or dropout_mask or two_ranks or rccl or channel_split or generator_stack or pin or grouped or wnorm or goldens" > $OUT/r3_s12_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s12_pytest.log
tail -12 $OUT/r3_s12_pytest.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/sa2 && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sa2 -- python $GRAFT_REPO_ROOT/tools/prof_stacks_alone.py 8 > /tmp/sa.log 2>&1; python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/sa2 | head -12 )
