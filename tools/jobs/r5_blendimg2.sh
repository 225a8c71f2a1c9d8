#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}"; export TMPDIR=/tmp; mkdir -p gpurun_out
timeout 112 python -m pytest "tests/test_gpu_step.py::test_replayed_steps_see_a_codebook_written_between_two_replays" "tests/test_gpu_step.py::test_replayed_vqvae_steps_equal_eager_steps_bit_for_bit" "tests/test_gpu_step.py::test_graphs_of_two_batch_shapes_alternate" "tests/test_gpu_step.py::test_graph_replayed_steps_equal_eager_steps" -m gpu -x -q --durations=6 -p no:cacheprovider > gpurun_out/bi_step2.log 2>&1; tail -14 gpurun_out/bi_step2.log
