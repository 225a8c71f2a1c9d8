#!/bin/bash
# Round 5, first GPU session: where the round starts (bench line), what the 1e-3 mode (bf16x3f) spends its time on, the
# on-the-fly log-mel kernel at the use_raw benchmark shape (old library vs the sparse mel projection), lsgan kernel statistics.
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=gpurun_out
timeout 120 python -m pytest tests/test_gpu_ops.py -m gpu -x -q -k "logmel" > $O/c1_logmel_tests.log 2>&1; tail -2 $O/c1_logmel_tests.log
for lib in libcrank_hip_r4.so libcrank_hip.so; do
  echo "== $lib" >> $O/c1_logmel.txt
  CRANK_AMD_LIB=$PWD/crank_amd/$lib timeout 120 python tools/prof_logmel.py >> $O/c1_logmel.txt 2>&1
  rm -rf $O/c1_lm; CRANK_AMD_LIB=$PWD/crank_amd/$lib timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1_lm -- python tools/prof_logmel.py 20 > /dev/null 2>&1
  python tools/kstats.py $O/c1_lm logmel >> $O/c1_logmel.txt
done
rm -rf $O/c1_lm; cat $O/c1_logmel.txt
timeout 600 python bench.py --no-cpu-baseline 2> $O/c1_bench.err | tail -1 > $O/c1_bench_line.json; cat $O/c1_bench_line.json | head -c 1500; echo
for mode in bf16x3f; do
  rm -rf $O/c1_p; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1_p -- python bench.py --precision $mode --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/c1_${mode}.log 2>&1
  python tools/kstats.py $O/c1_p > $O/c1_${mode}_kstats.txt; head -30 $O/c1_${mode}_kstats.txt; rm -rf $O/c1_p
done
rm -rf $O/c1_p; timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/c1_p -- python bench.py --trainer lsgan --steps 30 --warmup 5 --no-cpu-baseline --no-extras --no-roofline > $O/c1_lsgan.log 2>&1
python tools/kstats.py $O/c1_p > $O/c1_lsgan_kstats.txt; head -30 $O/c1_lsgan_kstats.txt; rm -rf $O/c1_p
