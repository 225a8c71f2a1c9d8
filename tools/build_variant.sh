#!/bin/bash
# Builds an instrumented / experimental copy of the library next to the product one:
#   tools/build_variant.sh <tag> <file.hip> "<extra flags>" [<file2.hip> "<flags2>" ...]
# compiles each named source with its flags into csrc/<file>.<tag>.o and links it with the product objects of every other
# source into crank_amd/libcrank_hip_<tag>.so (select with CRANK_AMD_LIB=...).  `make -C crank_amd/csrc` must have run.
set -e
REPO=$(cd "$(dirname "$0")/.." && pwd); C=$REPO/crank_amd/csrc
TAG=$1; shift
declare -A REPL
while [ $# -ge 2 ]; do
  f=$1; fl=$2; shift 2
  extra=""; [ "$f" = "vq_kernels.hip" ] && extra="-mllvm -amdgpu-mfma-vgpr-form"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result $extra $fl -c $C/$f -o $C/${f%.hip}.$TAG.o
  REPL[${f%.hip}]=1
done
OBJS=""
for s in $(grep '^SRCS' $C/Makefile | sed 's/SRCS := //'); do
  b=${s%.hip}
  if [ -n "${REPL[$b]}" ]; then OBJS="$OBJS $C/$b.$TAG.o"; else OBJS="$OBJS $C/$b.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $OBJS -o $REPO/crank_amd/libcrank_hip_$TAG.so
echo built crank_amd/libcrank_hip_$TAG.so
