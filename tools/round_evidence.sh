#!/bin/bash
# The round's evidence in one GPU session: the driver's test command in its order, smoke(), the default bench line, the
# rocprofv3 kernel statistics of the same bench command and the launch sequence of one replayed step.
# usage (on the GPU box): bash tools/round_evidence.sh TAG   -> gpurun_out/TAG_*  (copy what is to be judged into profiles/)
tag=${1:-evidence}
mkdir -p gpurun_out
cd "${GRAFT_REPO_ROOT:-.}"
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/${tag}_pytest_gpu.log 2>&1
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
