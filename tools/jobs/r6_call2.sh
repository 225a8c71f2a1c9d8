#!/bin/bash
# round 6, call 2: the new tests, then the whole GPU suite, then the bench line with parity gates / lsgan record
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 900 python -m pytest tests/test_gpu_nets.py tests/test_gpu_step.py tests/test_gpu_properties.py -m gpu -x -q -s -k "never_allocate or conversion_matches or trains_like_fp32 or forward_backward_vs_oracle or mcd_between" > $O/r6_c2_new.log 2>&1; tail -5 $O/r6_c2_new.log; grep -E "^\[|smoothed|MCD|decoded rel" $O/r6_c2_new.log | head -30
timeout 1200 python -m pytest tests/test_gpu_dp.py tests/test_gpu_bench_dp.py -m gpu -x -q -k "n_ranks or eight_ranks" > $O/r6_c2_dp.log 2>&1; tail -5 $O/r6_c2_dp.log
timeout 1500 python -m pytest tests -m gpu -x -q > $O/r6_c2_pytest_gpu.log 2>&1; echo "pytest rc=$?" >> $O/r6_c2_pytest_gpu.log; tail -4 $O/r6_c2_pytest_gpu.log
timeout 900 python bench.py 2> $O/r6_c2_bench.err | tail -1 > $O/r6_c2_bench_line.json; python - <<'PY'
import json
d = json.load(open("gpurun_out/r6_c2_bench_line.json"))
print("ms", d["ms_per_step"], "roof", d["roofline"]["frac"], "stacks", d["stacks_alone"].get("frac_of_mfma_peak"))
print("gates", json.dumps(d.get("parity_gates"))[:1500])
l = d["other_configs"]["lsgan"]; print("lsgan", l["ms_per_step"], l.get("cpu_baseline"), l.get("roofline", {}).get("kernel"), l.get("roofline", {}).get("frac"))
PY
tail -5 $O/r6_c2_bench.err
