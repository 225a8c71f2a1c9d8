#!/bin/bash
# SQ counters of the gated-stack kernels (forward, data gradient, weight gradient) over the four generator stacks in
# isolation (tools/prof_stacks_alone.py), three passes of <= 8 SQ counters.  Run on the GPU box:
#   bash tools/pmc_stacks.sh OUT.txt
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=${1:-$ROOT/gpurun_out/pmc_stacks.txt}
: > $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmcs$i -- python $ROOT/tools/prof_stacks_alone.py 3 > /tmp/pmcs$i.log 2>&1 || tail -3 /tmp/pmcs$i.log >> $OUT
  f=$(find /tmp/pmcs$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for r in csv.DictReader(open(sys.argv[1])):
    name = r["Kernel_Name"].split("(")[0]
    if "stack2_" in name or "stack_wgrad" in name:
        a = acc[name][r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for name in sorted(acc):
    print(name)
    for k, (n, v) in sorted(acc[name].items()):
        print(f"    {k:28s} launches {n:4d}  avg/launch {v / n:18.1f}")
PY
done
cat $OUT | head -150
