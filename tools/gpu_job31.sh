#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out; cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_step.py tests/test_gpu_dp.py tests/test_gpu_properties.py -m gpu -q -p no:cacheprovider -x -k "goldens_bf16x3 or replays or bit_for_bit or two_ranks or rccl or clip or full_size_gan" > $OUT/r3_s31_tests.log 2>&1; tail -5 $OUT/r3_s31_tests.log
export TMPDIR=/tmp
( cd /tmp && rm -rf /tmp/bk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bk -- python $GRAFT_REPO_ROOT/bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > $OUT/r3_s31_bench_under_rocprof.json 2>/tmp/bk.err; f=$(find /tmp/bk -name "*kernel_stats.csv" | head -1); cp "$f" $OUT/r3_s31_kernel_stats.csv; grep -E "weighted|at::native" $OUT/r3_s31_kernel_stats.csv | cut -c1-150; cat $OUT/r3_s31_bench_under_rocprof.json | cut -c1-200 )
