#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 30 python -m pytest tests/test_gpu_properties.py -m gpu -x -q -k "classifier_update_on_a_second_stream" -p no:cacheprovider > gpurun_out/bi_prop.log 2>&1; tail -2 gpurun_out/bi_prop.log
rm -rf /tmp/bip; timeout 55 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bip -- python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-extras --no-roofline > /tmp/bi_b.log 2>&1
kt=$(find /tmp/bip -name '*kernel_trace.csv' | head -1); ks=$(find /tmp/bip -name '*kernel_stats.csv' | head -1)
[ -n "$kt" ] && python tools/step_sequence.py "$kt" gpurun_out/bi_step_sequence.txt && head -1 gpurun_out/bi_step_sequence.txt
[ -n "$ks" ] && cp "$ks" gpurun_out/bi_kernel_stats.csv && grep -E "vq_image|vq_ema_blend" "$ks" | cut -d, -f1-4
grep '^{' /tmp/bi_b.log | tail -1 | python -c 'import json,sys;print("ms/step under rocprof", round(json.loads(sys.stdin.read())["ms_per_step"],4))'
