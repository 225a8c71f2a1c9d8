#!/bin/bash
# Round 6, third session, call 1: phase cycles of the plain chains as they are today; the bench line of HEAD
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out; O=$PWD/gpurun_out
timeout 200 python tools/ps2_phase_cycles.py 2>&1 | grep -v -i warn | tee $O/r6c_ps2_phase.txt
timeout 600 python bench.py > $O/r6c_bench_head.json 2> $O/r6c_bench_head.err; tail -c 600 $O/r6c_bench_head.json
