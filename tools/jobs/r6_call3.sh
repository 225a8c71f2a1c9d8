#!/bin/bash
# round 6, call 3: re-run of the tests call 2 failed, the rest of the suite behind them, forward-kernel ablations
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_step.py -m gpu -x -q -s -k "conversion_matches or trains_like_fp32" > $O/r6_c3_new.log 2>&1; tail -3 $O/r6_c3_new.log; grep -E "^\.?\[|smoothed" $O/r6_c3_new.log | head -30
timeout 1500 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_dp.py -m gpu -x -q -k "n_ranks or eight_ranks" > $O/r6_c3_dp.log 2>&1; tail -5 $O/r6_c3_dp.log
for mode in nograd fwd; do
  echo "mode $mode (nograd: nothing saved; fwd: planes written, as in training)"
  MODE=$mode bash tools/s2_ablate.sh 0 16 32 48 64 112 0
done > $O/r6_s2_ablation.txt 2>&1
cat $O/r6_s2_ablation.txt
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_properties.py tests/test_gpu_dp.py tests/test_gpu_bench_dp.py tests/test_gpu_mcd.py tests/test_gpu_fallback.py tests/test_gpu_dataset.py -m gpu -q > $O/r6_c3_rest.log 2>&1; tail -6 $O/r6_c3_rest.log
