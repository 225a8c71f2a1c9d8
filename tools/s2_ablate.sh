#!/bin/bash
# Ablation timing of stack2_fwd_kernel: builds libcrank_hip_abl<N>.so with -DS2_ABL=<N> (container, hipcc) when called
# with "build", otherwise (GPU box) times the no-grad G forward per variant with a kernel trace.
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/crank_amd/csrc
if [ "$1" = "build" ]; then
  shift
  for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DS2_ABL=$n -c $CS/stack2_kernels.hip -o $CS/stack2_kernels.abl$n.o || exit 1
    objs=""; for s in conv_kernels stack_kernels stack2b_kernels stack2x_kernels pstack_kernels pstack2_kernels pstack2x_kernels net vq_kernels loss_kernels mlfb_kernels dataset_kernels mcd_kernels; do objs="$objs $CS/$s.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $CS/stack2_kernels.abl$n.o -o $REPO/crank_amd/libcrank_hip_abl$n.so || exit 1
  done
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  lib=$REPO/crank_amd/libcrank_hip_abl$n.so; [ "$n" = "0" ] && lib=$REPO/crank_amd/libcrank_hip.so
  rm -rf /tmp/abl_$n
  CRANK_AMD_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/abl_$n -- python $REPO/tools/prof_fwd.py ${MODE:-nograd} 6 > /tmp/abl.log 2>&1 || tail -3 /tmp/abl.log
  f=$(find /tmp/abl_$n -name "*kernel_trace.csv" | head -1)
  python - "$f" "abl=$n" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "stack2" in n:
        acc[n.split("(")[0][:48]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[2], "  ".join(f"{k[22:]}: {sum(sorted(v)[:-1])/(len(v)-1):6.1f}" for k, v in sorted(acc.items())))
PY
done
