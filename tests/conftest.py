import os
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

# Collection order of the GPU suite: the hot-path parity tests (kernels -> nets -> whole steps ->
# size-independent properties -> the fallback kernels) run first, then the rows either side of the path (8f), and the
# multi-process tests (several ranks on one GPU over gloo, RCCL in a world of one) last - a failure in process plumbing
# must never stand in front of the parity evidence of an `-x` run.  Unknown files sort between the two groups.
_ORDER = ["test_gpu_ops", "test_gpu_nets", "test_gpu_step", "test_gpu_properties", "test_gpu_fallback",
          "test_gpu_dataset", "test_gpu_mcd"]
_LAST = ["test_gpu_dp", "test_gpu_bench_dp"]


def _rank(item):
    name = os.path.splitext(os.path.basename(str(item.fspath)))[0]
    if name in _ORDER:
        return _ORDER.index(name)
    if name in _LAST:
        return 1000 + _LAST.index(name)
    return 500


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_collection_modifyitems(config, items):
    import torch

    items.sort(key=_rank)  # (stable: the order inside a file is kept)
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for item in items:
        if "gpu" in item.keywords:
            item.add_marker(skip)
