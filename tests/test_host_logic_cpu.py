"""CPU: the product trainers (host logic: loss algebra, update order, conditioning)
driven with the oracle modules must reproduce the golden step vectors that were
produced by the REFERENCE's own trainer / VQVAE2 / SpeakerAdversarialNetwork classes
(tests/golden/make_golden.py)."""
import numpy as np
import pytest
import torch

from tests.helpers import STEP_CASES, compare_losses, run_golden_case, state_summary


def _oracle_factories():
    from oracle import modules as om

    def sched(conf, optimizer):
        return {m: torch.optim.lr_scheduler.StepLR(o, conf["optim"][m]["decay_step_size"], conf["optim"][m]["decay_size"])
                for m, o in optimizer.items()}

    return (lambda conf, n, scaler=None: om.get_model(conf, n, scaler), om.get_optimizer, om.get_criterion, sched)


@pytest.mark.parametrize("tag", list(STEP_CASES))
def test_trainer_matches_reference_step(tag):
    torch.set_num_threads(4)
    bm, bo, bc, bs = _oracle_factories()
    losses, models, trainer, fx, post = run_golden_case(tag, bm, bo, bc, bs)
    bad = compare_losses(losses, fx, rtol=2e-4)
    assert not bad, bad
    np.testing.assert_allclose(post["decoded"].numpy(), fx["post_decoded"], rtol=1e-3, atol=1e-4)
    assert (post["qidx"][0].numpy() == fx["post_qidx0"]).mean() > 0.999
    assert (post["qidx"][1].numpy() == fx["post_qidx1"]).mean() > 0.999
    summ = state_summary(models)
    for k, v in summ.items():
        np.testing.assert_allclose(v, fx[k], rtol=2e-3, atol=2e-4, err_msg=k)


def test_device_loader_shards_global_batches_across_ranks():
    """Rank r gets utterances [r*B, (r+1)*B) of every global batch, all ranks see the same permutation and
    run the same number of steps; the short last batch is spread, or dropped when it cannot
    feed every rank."""
    from crank_amd.net.trainer.dataset import DeviceLoader

    class Fake:
        def __init__(self, n):
            self.n = n

        def __len__(self):
            return self.n

        def assemble(self, idx):
            return list(idx)

    for n, B, W in [(23, 4, 2), (24, 4, 2), (17, 4, 4), (7, 2, 1)]:
        per_rank = [list(DeviceLoader(Fake(n), B, shuffle=True, rank=r, world_size=W, seed=5)) for r in range(W)]
        steps = {len(b) for b in per_rank}
        assert len(steps) == 1, (n, B, W)
        seen = []
        for s in range(steps.pop()):
            glob = [i for r in range(W) for i in per_rank[r][s]]
            assert len(set(glob)) == len(glob) and all(len(per_rank[r][s]) >= 1 for r in range(W))
            if len(glob) == B * W:
                assert all(len(per_rank[r][s]) == B for r in range(W))
            seen += glob
        assert len(set(seen)) == len(seen) and len(seen) >= n - (W - 1) and set(seen) <= set(range(n))
        # a new epoch reshuffles, identically on every rank
        l0, l1 = (DeviceLoader(Fake(n), B, shuffle=True, rank=r, world_size=W, seed=5) for r in (0, W - 1))
        e1a, e2a = list(l0), list(l0)
        e1b, e2b = list(l1), list(l1)
        assert len(e1a) == len(e1b) == len(e2a) == len(e2b)
        if n > B * W:
            assert e1a != e2a


def test_lossbook_totals_equal_the_reference_accumulation():
    """loss.add(key, w, term) == the reference's loss[key] += w * term: same value, same gradients; a zero
    weight leaves the term out of the graph; constants and an existing tensor total are carried."""
    import torch

    from crank_amd.net.trainer.basetrainer import LossBook

    torch.manual_seed(0)
    xs = [torch.randn((), requires_grad=True) for _ in range(5)]
    ws = [2.0, 0.0, 1.0, 0.25, 0.1]
    ref = 0.0
    for w, x in zip(ws, xs):
        ref = ref + w * (x * x)
    ref.backward()
    g_ref = [x.grad.clone() for x in xs]
    for x in xs:
        x.grad = None
    book = LossBook()
    terms = [x * x for x in xs]
    for w, t in zip(ws, terms):
        book.add("G", w, t)
    total = book["G"]
    assert torch.allclose(total, ref.detach(), rtol=1e-6)
    assert book["G"] is total  # settled once, cached
    total.backward()
    for x, g, w in zip(xs, g_ref, ws):
        if w == 0.0:
            assert x.grad is None  # not part of the graph
        else:
            assert torch.allclose(x.grad, g, rtol=1e-6)
    book.add("G", 1.0, terms[0].detach())  # adding after a read extends the settled total
    assert torch.allclose(book["G"], ref.detach() + terms[0].detach(), rtol=1e-6)
    book.add("objective", 1.0, total)
    assert book["objective"] is total  # a single unit-weight term is passed through, no kernel
    book.add("C", 0.5, 3.0)
    assert book["C"] == 1.5 and dict(book.items())["D"] == 0.0


def test_scp_list_files(tmp_path):
    """wav.scp / utt2spk / spk2utt / feats.scp parsing with the dict layout the reference's train.py and dataset
    read (crank/utils/utils.py:33-64): speaker order = spk2utt order, blank lines tolerated, malformed lines rejected."""
    import pytest

    from crank_amd.utils import open_featsscp, open_scpdir

    d = tmp_path / "train"
    d.mkdir()
    (d / "wav.scp").write_text("SF1_10001 downloads/wav/SF1/10001.wav\nTM2_30002 downloads/wav/TM2/30002.wav\n\n")
    (d / "utt2spk").write_text("SF1_10001 SF1\nTM2_30002 TM2\n")
    (d / "spk2utt").write_text("TM2 TM2_30002\nSF1 SF1_10001\nSM9\n")
    scp = open_scpdir(d)
    assert scp["spkrs"] == ["TM2", "SF1", "SM9"] and scp["spk2utt"] == {"TM2": ["TM2_30002"], "SF1": ["SF1_10001"], "SM9": []}
    assert scp["wav"]["TM2_30002"].endswith("30002.wav") and scp["utt2spk"]["SF1_10001"] == "SF1" and scp["feats"] == {}
    (tmp_path / "feats.scp").write_text("SF1_10001 /data/h5/SF1/10001.h5\n")
    assert open_featsscp(tmp_path / "feats.scp") == {"SF1_10001": "/data/h5/SF1/10001.h5"}
    (tmp_path / "bad.scp").write_text("only_one_column\n")
    with pytest.raises(ValueError):
        open_featsscp(tmp_path / "bad.scp")


def test_cat_channels_is_a_view_of_side_by_side_slices_and_falls_back_otherwise():
    """ops.cat_channels: pieces that are the adjacent column slices of one buffer come back as that buffer (no copy),
    with the gradient handed out in slices; anything else is torch.cat.  Pure host / autograd logic."""
    from crank_amd import ops

    torch.manual_seed(0)
    B, T = 3, 7
    buf = torch.randn(B, T, 12)
    a = buf[..., 0:4].clone().requires_grad_(True)
    b = buf[..., 4:12].clone().requires_grad_(True)

    class Put(torch.autograd.Function):  # a producer that writes its result into a column slice of a wider buffer
        @staticmethod
        def forward(ctx, x, dst, col):
            # (the HIP kernels write through raw pointers: no torch in-place op, no version bump - numpy does the same here)
            dst.numpy()[..., col: col + x.shape[2]] = x.detach().numpy()
            return dst[..., col: col + x.shape[2]]

        @staticmethod
        def backward(ctx, g):
            return g, None, None

    dst = torch.empty(B, T, 12)
    pa, pb = Put.apply(a, dst, 0), Put.apply(b, dst, 4)
    y = ops.cat_channels([pa, pb])
    assert y.data_ptr() == dst.data_ptr() and y.shape == (B, T, 12) and y.is_contiguous()
    w = torch.randn(B, T, 12)
    (y * w).sum().backward()
    assert torch.equal(y, torch.cat([a, b], -1).detach())
    assert torch.equal(a.grad, w[..., 0:4]) and torch.equal(b.grad, w[..., 4:12])
    # not adjacent (wrong order / separate tensors / time-sliced views): a real concatenation, same values
    for pieces in ([pb, pa], [a, b], [pa[:, 1:], pb[:, 1:]]):
        z = ops.cat_channels(pieces)
        assert torch.equal(z, torch.cat(pieces, -1)) and z.data_ptr() != dst.data_ptr()


def test_lossvalues_resolve_lazily_and_behave_like_a_dict_of_floats():
    from crank_amd.net.trainer.basetrainer import LossValues

    v = LossValues(["G", "C"], torch.tensor([1.5, 2.5]), None, ["D"])
    assert set(v.keys()) == {"G", "C", "D", "objective", "SPKRADV"} or {"G", "C", "D"} <= set(v.keys())
    assert v["G"] == 1.5 and v.get("C") == 2.5 and v["D"] == 0.0
    assert all(isinstance(x, float) for x in v.values())


def test_committed_bench_line_follows_the_bench_contract():
    """The newest profiles/round*_bench_line.json (what `python bench.py` printed on the MI355X box) carries every field
    of the driver's contract, the metric of BASELINE.json's first clause, a roofline object whose fraction is
    achieved / peak, and a CPU baseline that says what was timed."""
    import glob
    import json
    import os

    from tests.helpers import REPO

    lines = sorted(glob.glob(os.path.join(REPO, "profiles", "round*_bench_line.json")))
    assert lines, "no committed bench line under profiles/"
    d = json.load(open(lines[-1]))
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(REPO, "BASELINE.json")))
    assert d["metric"].split()[0:2] == base["metric"].split()[0:2] and d["unit"] == "frames/s"  # "train frames/sec"
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # value = whole-job frames per second of the timed steps
    frames = d["config"]["global_batch"] * d["config"]["batch_len"]
    assert abs(d["value"] - frames / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s")
    assert abs(r["frac"] - r["achieved"] / r["peak"]) <= 1e-9 and 0 < r["frac"] < 1
    assert r["traffic"] is None or r["traffic"] > 0
    c = d["cpu_baseline"]
    assert c["kind"] in ("reference", "port") and c["cores"] >= 1 and c["value"] > 0 and c["unit"] == d["unit"] and c["sample"]


def test_vq_search_key_bounds_cover_what_seven_replaced_mantissa_bits_can_do():
    """The split-f16 VQ search (csrc/vq_kernels.hip) tracks its candidates as KEYS - the approximate distance with its 7 low
    mantissa bits replaced by a tag - and widens its decision threshold by VQH_KEY_EPS * |key| + VQH_KEY_ABS per key.  That
    is a claim about fp32: for every finite value v and every tag, |key - v| <= EPS * |key| + ABS - including subnormal v
    (an all-zero frame facing all-zero codes: keys are bare tags), zeros of both signs and the largest finite values.  The
    constants are read from the kernel source."""
    import os
    import re

    from tests.helpers import REPO

    text = open(os.path.join(REPO, "crank_amd", "csrc", "vq_kernels.hip")).read()
    eps = np.float32(re.search(r"#define\s+VQH_KEY_EPS\s+([0-9.e+-]+)f", text).group(1))
    ab = np.float32(re.search(r"#define\s+VQH_KEY_ABS\s+([0-9.e+-]+)f", text).group(1))
    assert re.search(r"&\s*keymask\)\s*\|\s*tag", text) and "keymask = ~0x7fu" in text  # (the operation restated below)
    assert ab >= np.finfo(np.float32).tiny  # a NORMAL number: a subnormal constant could be flushed where the keys are not
    rng = np.random.default_rng(3)
    bits = rng.integers(0, 2 ** 32, size=2_000_000, dtype=np.uint64).astype(np.uint32)
    special = np.array([0x00000000, 0x80000000, 0x00000001, 0x0000007f, 0x00000080, 0x007fffff, 0x00800000, 0x00800001,
                        0x7f7fffff, 0xff7fffff, 0x3f800000, 0x3f80007f, 0x807fffff], dtype=np.uint32)
    bits = np.concatenate([bits, special, (special[:, None] ^ np.arange(128, dtype=np.uint32)[None, :]).reshape(-1)])
    v = bits.view(np.float32)
    keep = np.isfinite(v)
    bits, v = bits[keep], v[keep].astype(np.float64)
    for tag in (0, 1, 0x3a, 0x7f):
        key = ((bits & np.uint32(0xffffff80)) | np.uint32(tag)).view(np.float32).astype(np.float64)
        assert np.isfinite(key).all()  # (finite values keep their exponent: no key of a finite value is inf / NaN)
        slack = float(eps) * np.abs(key) + float(ab) - np.abs(key - v)
        assert slack.min() >= 0.0, (tag, float(v[slack.argmin()]), float(key[slack.argmin()]))
    # ... and the relative term alone does NOT cover the subnormal range - why the absolute one exists
    sub = np.array([0x00000005], dtype=np.uint32)
    key = ((sub & np.uint32(0xffffff80)) | np.uint32(2)).view(np.float32).astype(np.float64)
    assert abs(key[0] - float(sub.view(np.float32)[0])) > float(eps) * abs(key[0])


def test_flag_constants_of_the_binding_equal_the_c_abi_header():
    """crank_amd/ops.py names the crk_net_forward / crk_net_backward flags; the values are those of include/crank_hip.h
    (and of the library's own copy in csrc/common.h)."""
    import os
    import re

    from crank_amd import ops
    from tests.helpers import REPO

    for path in ("include/crank_hip.h", "crank_amd/csrc/common.h"):
        text = open(os.path.join(REPO, path)).read()
        found = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define\s+(CRK_FLAG_\w+)\s+(\d+)", text)}
        assert len(found) == 7, (path, found)
        for name, value in found.items():
            assert getattr(ops, name) == value, (path, name)


def test_lossvalues_pick_their_numbers_out_of_an_arena_by_position():
    """With an index the host vector is the step's whole scalar arena and key i sits at index[i] (keys may share a slot)."""
    from crank_amd.net.trainer.basetrainer import LossValues

    arena = torch.arange(16, dtype=torch.float32) * 0.5
    v = LossValues(["G", "objective", "C_real"], arena, None, ["D"], index=[4, 4, 9])
    assert v["G"] == 2.0 and v["objective"] == 2.0 and v["C_real"] == 4.5 and v["D"] == 0.0


def test_label_runs_recognise_the_first_label_view_and_nothing_else():
    """ops._label_runs: the stride-0 view ``h[:, 0:1].expand(-1, T)`` of a contiguous (B,T) label tensor goes to the lookup
    kernels as h itself with run = T (they read idx[n - n % run]); every other layout as a contiguous copy with run = 1."""
    from crank_amd import ops

    h = torch.full((3, 7), -100, dtype=torch.long)
    h[:, :4] = torch.tensor([[2], [0], [5]])
    view = h[:, 0:1].expand(-1, 7)
    t, run = ops._label_runs(view)
    assert run == 7 and t.data_ptr() == h.data_ptr()
    flat = h.reshape(-1)
    n = torch.arange(21)
    assert torch.equal(flat[n - n % run].view(3, 7), view)  # what the kernels read is what the view holds
    for other in (h, view.contiguous(), h[:, 1:2].expand(-1, 7)[:, :6], h.t()[0:1].expand(3, -1)):
        t, run = ops._label_runs(other)
        assert run == 1 and t.is_contiguous() and torch.equal(t, other)
    one = h[:1, 0:1].expand(-1, 7)  # a batch of one utterance
    assert ops._label_runs(one)[1] == 7


def test_scalar_arena_hands_out_aligned_slices_and_stays_off_on_the_cpu():
    from crank_amd import ops

    assert ops.begin_scalar_arena(None) is None and ops.begin_scalar_arena("cpu") is None and ops.scalar_arena() is None
    t = ops._scalars(2, torch.device("cpu"))
    assert t.shape == (2,) and t.dtype == torch.float32  # no arena: a tensor of its own
    a = ops._ScalarArena(torch.device("cpu"), n=16)  # (the class itself is device agnostic)
    ops._ARENA = a
    try:
        x, y, z = ops._scalars(2, a.buf.device), ops._scalars(5, a.buf.device), ops._scalars(1, a.buf.device)
        assert (x.storage_offset(), y.storage_offset(), z.storage_offset()) == (0, 4, 12)
        assert all(v.untyped_storage().data_ptr() == a.buf.untyped_storage().data_ptr() for v in (x, y, z))
        big = ops._scalars(8, a.buf.device)  # does not fit any more: a tensor of its own
        assert big.untyped_storage().data_ptr() != a.buf.untyped_storage().data_ptr()
    finally:
        ops._ARENA = None


def test_collector_is_held_off_for_a_capture_and_handed_back():
    import gc

    from crank_amd.net.trainer.basetrainer import hold_collector_for_capture

    class Node:
        pass

    a, b = Node(), Node()
    a.other, b.other = b, a
    import weakref

    alive = weakref.ref(a)
    del a, b
    assert gc.isenabled()
    was_on = hold_collector_for_capture()
    try:
        assert was_on and not gc.isenabled() and alive() is None  # collected BEFORE the capture, none during it
    finally:
        gc.enable()
