#!/bin/bash
# Ablation timing of stack2_bwd_kernel (the data-gradient chain): `tools/s2b_ablate.sh build N...` (container, hipcc) builds
# crank_amd/libcrank_hip_s2babl<N>.so with -DS2B_ABL=<N>; `tools/s2b_ablate.sh N...` (GPU box) times the four generator stacks'
# backward per variant from a kernel trace.  Bits: 1 no gate arithmetic, 2 no MFMAs, 4 no LDS fragment reads, 8 no weight
# loads, 16 no gate-plane loads, 32 no plane stores.  Timing only: the results of an ablated build are wrong.
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/crank_amd/csrc
if [ "$1" = "build" ]; then
  shift
  for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DS2B_ABL=$n -c $CS/stack2b_kernels.hip -o $CS/stack2b_kernels.abl$n.o || exit 1
    objs=""; for s in conv_kernels stack_kernels stack2_kernels pstack_kernels pstack2_kernels net vq_kernels loss_kernels mlfb_kernels dataset_kernels mcd_kernels; do objs="$objs $CS/$s.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $CS/stack2b_kernels.abl$n.o -o $REPO/crank_amd/libcrank_hip_s2babl$n.so || exit 1
  done
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  lib=$REPO/crank_amd/libcrank_hip_s2babl$n.so; [ "$n" = "0" ] && lib=$REPO/crank_amd/libcrank_hip.so
  rm -rf /tmp/s2babl_$n
  CRANK_AMD_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/s2babl_$n -- python $REPO/tools/prof_stacks_alone.py 4 > /tmp/s2babl.log 2>&1 || tail -3 /tmp/s2babl.log
  f=$(find /tmp/s2babl_$n -name "*kernel_trace.csv" | head -1)
  python - "$f" "abl=$n" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"]
    if "stack2_bwd" in n:
        acc[n.split("(")[0][5:]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print("%-8s" % sys.argv[2], "  ".join(f"{k[17:]}: {sum(sorted(v)[:-1])/(len(v)-1):6.1f}" for k, v in sorted(acc.items())))
PY
done
