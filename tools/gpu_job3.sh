#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_properties.py tests/test_gpu_nets.py -m gpu -q -p no:cacheprovider -x -k "channel_split or pin or dropout_mask or generator_stack or full_size_gan" > $OUT/r3_s3_pytest.log 2>&1; echo "pytest rc $?" >> $OUT/r3_s3_pytest.log
tail -25 $OUT/r3_s3_pytest.log
export TMPDIR=/tmp
for v in 2 1; do
( cd /tmp && rm -rf /tmp/sa$v && CRK_SKB_V=$v timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/sa$v -- python $GRAFT_REPO_ROOT/tools/prof_stacks_alone.py 8 > /tmp/sa.log 2>&1; python $GRAFT_REPO_ROOT/tools/kstats.py /tmp/sa$v stack > $OUT/r3_s3_kstats_v$v.txt 2>&1 )
echo "== CRK_SKB_V=$v"; head -12 $OUT/r3_s3_kstats_v$v.txt
done
