#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
rm -rf $O/c9_p; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c9_p -- python bench.py --gpus 1 --force-dist --steps 40 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $O/c9_prof.log 2>&1
kt=$(find $O/c9_p -name '*kernel_trace.csv' | head -1)
python tools/step_sequence.py "$kt" $O/c9_force_dist_step_sequence.txt
python tools/kstats.py $O/c9_p > $O/c9_force_dist_kstats.txt; head -12 $O/c9_force_dist_kstats.txt; rm -rf $O/c9_p
awk '{ if ($4+0 > 8.0) print }' $O/c9_force_dist_step_sequence.txt | head -40
