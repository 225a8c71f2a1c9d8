"""Every switch of the package, read from the environment ONCE, when the package is imported.

Nothing in `crank_amd/` reads `os.environ` after this module has run: the training step, the trainers' sub-updates and
the autograd functions look at attributes of `cfg`.  Tests and `bench.py` that need another value inside one process use
`override(name=value)` (a context manager); a different value for a whole process is an environment variable set
before the import.  The kernels' own A/B switches (`CRK_*`, read once per process by the library) are listed in DESIGN.md.

    name                    environment variable                 default   meaning
    precision               CRANK_AMD_PRECISION                  bf16      arithmetic of the conv stacks: bf16 | bf16x3f | bf16x3
    lib_path                CRANK_AMD_LIB                        (in-tree) another build of libcrank_hip.so (instrumented builds)
    default_yaml            CRANK_DEFAULT_YAML                   (in-tree) recipe defaults (`utils.load_yaml`)
    overlap_c               CRANK_AMD_OVERLAP_C                  1         speaker classifier's update on a second stream: 0 in line,
                                                                           1 forked at the start of the step, 2 forked after G's update
    separate_ce             CRANK_AMD_SEPARATE_CE                0         cross entropy and the net's backward as separate launches
    separate_commit         CRANK_AMD_SEPARATE_COMMIT            0         commitment loss outside the quantizer's launch
    recon_dense             CRANK_AMD_RECON_DENSE                0         L1 / MSE / STFT losses of the decoded features unfused
    stft_two_pass           CRANK_AMD_STFT_TWO_PASS              0         STFT loss forward and gradient as two passes
    force_dist              CRANK_AMD_FORCE_DIST                 0         data-parallel code path in a process group of one rank
    dist_backend            CRANK_AMD_DIST_BACKEND               (auto)    gloo: several ranks sharing one GPU (tests)
    dp_graph_collectives    CRANK_AMD_DP_GRAPH_COLLECTIVES       1         RCCL collectives captured with the step (0: chain of graphs)
    capture_mode            CRANK_AMD_CAPTURE_MODE               thread_local   stream-capture mode of GraphedStep
    test_refuse_capture_rank CRANK_AMD_TEST_REFUSE_CAPTURE_RANK  (none)    test hook: this rank's capture fails
"""
import contextlib
import os
import types

_get = os.environ.get  # (the one place the package touches the environment; `torchrun` below reads its launch contract)


def _flag(name, default):
    v = _get(name)
    return default if v is None else v not in ("0", "")


def _read():
    return types.SimpleNamespace(
        precision=_get("CRANK_AMD_PRECISION", "bf16"),
        lib_path=_get("CRANK_AMD_LIB") or None,
        default_yaml=_get("CRANK_DEFAULT_YAML") or None,
        overlap_c=int(_get("CRANK_AMD_OVERLAP_C", "1") or 0),
        separate_ce=_flag("CRANK_AMD_SEPARATE_CE", False),
        separate_commit=_flag("CRANK_AMD_SEPARATE_COMMIT", False),
        recon_dense=_flag("CRANK_AMD_RECON_DENSE", False),
        stft_two_pass=_flag("CRANK_AMD_STFT_TWO_PASS", False),
        force_dist=_flag("CRANK_AMD_FORCE_DIST", False),
        dist_backend=_get("CRANK_AMD_DIST_BACKEND") or None,
        dp_graph_collectives=_flag("CRANK_AMD_DP_GRAPH_COLLECTIVES", True),
        capture_mode=_get("CRANK_AMD_CAPTURE_MODE", "thread_local"),
        test_refuse_capture_rank=_get("CRANK_AMD_TEST_REFUSE_CAPTURE_RANK"),
    )


cfg = _read()


def reload():
    """Re-read the environment (a launcher that sets variables after the import, e.g. `bench.py --force-dist`)."""
    cfg.__dict__.update(_read().__dict__)


def torchrun():
    """The launch contract of torch.distributed.run: (RANK or None, LOCAL_RANK, WORLD_SIZE)."""
    r = _get("RANK")
    return (None if r is None else int(r)), int(_get("LOCAL_RANK", "0")), int(_get("WORLD_SIZE", "1"))


@contextlib.contextmanager
def override(**kw):
    """Temporarily set switches inside one process (tests, bench.py's one-stream event pass)."""
    unknown = [k for k in kw if not hasattr(cfg, k)]
    if unknown:
        raise AttributeError(f"unknown switch(es): {unknown}")
    old = {k: getattr(cfg, k) for k in kw}
    cfg.__dict__.update(kw)
    try:
        yield cfg
    finally:
        cfg.__dict__.update(old)
