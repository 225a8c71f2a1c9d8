#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
: > $OUT/r3_s16_recon.txt
for c in 0 1 2; do
( cd /tmp && rm -rf /tmp/rk && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rk -- python $GRAFT_REPO_ROOT/tools/time_recon.py $c >> $OUT/r3_s16_recon.txt 2>/tmp/rk.err; tail -2 /tmp/rk.err; f=$(find /tmp/rk -name "*kernel_stats.csv" | head -1); echo "case $c" >> $OUT/r3_s16_recon.txt; grep -E "recon" "$f" | cut -c1-110 >> $OUT/r3_s16_recon.txt )
done
cat $OUT/r3_s16_recon.txt
