#!/bin/bash
# PMC passes over the G forward (fused stack kernel); run on the GPU box via gpurun.
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_MFMA" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d /tmp/pmc$i -- python /root/repo/tools/prof_fwd.py ${1:-} > /tmp/pmc$i.log 2>&1 || tail -3 /tmp/pmc$i.log
  f=$(find /tmp/pmc$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python - "$f" "${2:-stack_fwd}" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[2] in r["Kernel_Name"]:
        a = acc[r["Counter_Name"]]; a[0] += 1; a[1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print(f"{k:32s} launches {n:4d}  avg/launch {v / n:16.1f}")
PY
done
