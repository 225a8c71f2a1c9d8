#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
# negative control: with GraphedStep's check of the codebook count disabled the new test must FAIL
timeout 40 python - > gpurun_out/bi_control.log 2>&1 <<'PY'
import sys
sys.path.insert(0, ".")
from crank_amd.net.trainer import basetrainer
basetrainer.GraphedStep._codebook_state = lambda self: ()
import tests.test_gpu_step as t
try:
    t.test_replayed_steps_see_a_codebook_written_between_two_replays()
    print("CONTROL: test passed WITHOUT the refresh (it does not detect the hazard)")
except AssertionError as e:
    print("CONTROL: test fails without the refresh, as it must:", str(e)[:300])
PY
tail -2 gpurun_out/bi_control.log
timeout 60 python -m pytest tests/test_gpu_step.py -m gpu -x -q -k "every_trainer_replays or capture_survives" -p no:cacheprovider > gpurun_out/bi_step3.log 2>&1; tail -3 gpurun_out/bi_step3.log
