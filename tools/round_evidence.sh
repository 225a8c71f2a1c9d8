#!/bin/bash
# The round's evidence in one GPU session: the driver's test command in its order, smoke(), the default bench line, the
# rocprofv3 kernel statistics of the same bench command and the launch sequence of one replayed step.
# usage (on the GPU box): bash tools/round_evidence.sh TAG [TESTS]  -> gpurun_out/TAG_*  (copy what is to be judged into
# profiles/); TESTS: what pytest runs (default: tests, the driver's command)
tag=${1:-evidence}
tests=${2:-tests}
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 1500 python -m pytest $tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1
echo "pytest rc=$?" >> gpurun_out/${tag}_pytest_gpu.log
tail -3 gpurun_out/${tag}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${tag}_smoke.log 2>&1; echo "smoke rc=$?" | tee -a gpurun_out/${tag}_smoke.log
timeout 900 python bench.py 2> gpurun_out/${tag}_bench.err | tail -1 > gpurun_out/${tag}_bench_line.json
cat gpurun_out/${tag}_bench_line.json
rm -rf gpurun_out/${tag}_prof
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/${tag}_prof_bench.log 2>&1
ks=$(find gpurun_out/${tag}_prof -name '*kernel_stats.csv' | head -1)
kt=$(find gpurun_out/${tag}_prof -name '*kernel_trace.csv' | head -1)
[ -n "$ks" ] && cp "$ks" gpurun_out/${tag}_kernel_stats.csv
[ -n "$kt" ] && python tools/step_sequence.py "$kt" gpurun_out/${tag}_step_sequence.txt
rm -rf gpurun_out/${tag}_prof
# the same with the classifier's update in line (one stream): every kernel's own duration, what bench.py's roofline prices
timeout 600 env CRANK_AMD_OVERLAP_C=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${tag}_prof1 -- python bench.py --steps 100 --warmup 20 --no-cpu-baseline --no-extras > gpurun_out/${tag}_prof1_bench.log 2>&1
ks=$(find gpurun_out/${tag}_prof1 -name '*kernel_stats.csv' | head -1)
[ -n "$ks" ] && cp "$ks" gpurun_out/${tag}_kernel_stats_one_stream.csv
rm -rf gpurun_out/${tag}_prof gpurun_out/${tag}_prof1
# HBM traffic per kernel of the same bench command (PMC passes of their own, no tracing besides the kernel trace): the
# source of roofline.traffic - regenerated in the session of the bench line
bash tools/pmc_traffic.sh > gpurun_out/${tag}_pmc.log 2>&1; [ -f gpurun_out/pmc_traffic.csv ] && cp gpurun_out/pmc_traffic.csv gpurun_out/${tag}_pmc_traffic.csv
