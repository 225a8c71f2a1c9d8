#!/bin/bash
# Ablation timing of pstack_kernel: "build N..." (container) makes libcrank_hip_psabl<N>.so with -DPS_ABL=<N>; otherwise
# (GPU box) times the speaker classifier's forward + backward per variant with a kernel trace.
REPO=$(cd "$(dirname "$0")/.." && pwd)
CS=$REPO/crank_amd/csrc
if [ "$1" = "build" ]; then
  shift
  for n in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DPS_ABL=$n -c $CS/pstack_kernels.hip -o $CS/pstack_kernels.abl$n.o || exit 1
    objs=""; for s in conv_kernels stack_kernels stack2_kernels net vq_kernels loss_kernels mlfb_kernels dataset_kernels mcd_kernels; do objs="$objs $CS/$s.o"; done
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs $CS/pstack_kernels.abl$n.o -o $REPO/crank_amd/libcrank_hip_psabl$n.so || exit 1
  done
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
for n in "$@"; do
  lib=$REPO/crank_amd/libcrank_hip_psabl$n.so; [ "$n" = "0" ] && lib=$REPO/crank_amd/libcrank_hip.so
  rm -rf /tmp/psabl_$n
  CRANK_AMD_LIB=$lib rocprofv3 --kernel-trace --output-format csv -d /tmp/psabl_$n -- python $REPO/tools/prof_c.py 6 > /tmp/psabl.log 2>&1 || tail -3 /tmp/psabl.log
  echo "abl=$n"; python $REPO/tools/kstats.py /tmp/psabl_$n --by-grid | grep pstack_kernel | cut -c1-130
done
