#!/bin/bash
# (record of an experiment: the variant libraries it loops over were built from source edits that were NOT kept - see the
# profiles/round5_* file of the same experiment for what each variant was)
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out/vqvar.txt; : > $O
for v in a b c d; do
  echo "== variant $v" >> $O
  VQ_PROF_LIB=$PWD/crank_amd/libcrank_hip_vqp_$v.so timeout 200 python tools/vq_phase_cycles.py 2>&1 | grep -v amdgpu.ids | grep "prepared image" >> $O
done
cat $O
