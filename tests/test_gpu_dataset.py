"""Batch assembly and decode-side post-processing on the device.

* against tests/golden/dataset.npz = outputs of the reference's own BaseDataset / default collate /
  convert_f0 / sklearn scalers / BaseTrainer._store_features / _get_cvf0: bit-exact (float64 f0
  through exp(): 4 ulp);
* against the numpy oracle at the benchmark shape (B=64, T=500, 80-dim, 14 speakers): bit-exact;
* the host-side RNG draws follow the reference's order.
"""
import random
import time
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import dataset as ods
from tests.test_dataset_cpu import BATCH_KEYS, CASES, load_case

pytestmark = pytest.mark.gpu


def _scaler_objects(scaler, ftype, spkrs):
    """The shape of the reference's scaler.pkl: sklearn-like objects with mean_ / scale_ / var_."""
    sc = {ftype: SimpleNamespace(mean_=scaler["feat_mean"], scale_=scaler["feat_scale"]),
          "lcf0": SimpleNamespace(mean_=scaler["lcf0_mean"], scale_=scaler["lcf0_scale"])}
    for i, s in enumerate(spkrs):
        sc[s] = {"lcf0": SimpleNamespace(mean_=scaler["spk_lcf0_mean"][i : i + 1], var_=scaler["spk_lcf0_var"][i : i + 1])}
    return sc


def _dataset(corpus, scaler_np, ftype, spkrs, blen, use_mcep_0th=True, cap=None):
    from crank_amd.net.trainer.dataset import BaseDataset

    files = {f"/nonexistent/h5/{spkrs[u['spk']]}/utt{i:04d}.h5": u for i, u in enumerate(corpus)}

    def reader(h5f, ext="mlfb"):
        u = files[h5f]
        if ext == "cap":
            return np.zeros((u["feat"].shape[0], 2), np.float32) if cap is None else cap[h5f]
        return u["feat"] if ext == ftype else u[ext]

    conf = {"batch_len": blen, "input_feat_type": ftype, "output_feat_type": ftype, "use_raw": False,
            "ignore_scaler": [], "use_mcep_0th": use_mcep_0th, "spec_augment": False}
    scp = {"train": {"feats": {k: k for k in files}, "spkrs": spkrs}}
    return BaseDataset(conf, scp, _scaler_objects(scaler_np, ftype, spkrs), phase="train", reader=reader), conf


def _fix_var(scaler, g):
    # the scaler objects carry var_ (like sklearn's); sqrt(var_) is taken by the product where the reference takes it
    scaler = dict(scaler)
    scaler["spk_lcf0_var"] = g("spk_lcf0_var")
    return scaler


@pytest.mark.parametrize("case", CASES)
def test_collate_matches_reference_dataset_bitwise(case):
    fx, g, corpus, scaler = load_case(case)
    blen = g("batch/in_feats").shape[1]
    SPKRS = [str(v) for v in g("spkrs")]
    ftype = "mcep" if case == "mcep" else "mlfb"
    dset, _ = _dataset(corpus, _fix_var(scaler, g), ftype, SPKRS, blen, use_mcep_0th=(case != "mcep"))
    d = g("draws_utt_cv_p")
    draws = [(SPKRS[int(c)], int(p)) for _, c, p in d]
    batch = dset.assemble(d[:, 0].tolist(), draws=draws)
    torch.cuda.synchronize()
    for k in BATCH_KEYS + (["mcep_0th"] if case == "mcep" else []):
        ref = g(f"batch/{k}")
        got = batch[k].cpu().numpy()
        assert got.shape == ref.shape and got.dtype == ref.dtype, (k, got.shape, got.dtype, ref.shape, ref.dtype)
        assert np.array_equal(got, ref), (k, np.abs(got.astype(np.float64) - ref).max())
    # same draws from the same seed, in the reference's order (target speaker, then crop start, per sample)
    random.seed(99)
    again = dset.assemble(list(range(len(dset))))
    assert again["cv_spkr_name"] == [c for c, _ in draws]
    assert torch.equal(again["in_feats"], batch["in_feats"]) and torch.equal(again["cv_lcf0"], batch["cv_lcf0"])
    one = dset[3]
    assert one["in_feats"].shape == (blen, batch["in_feats"].shape[-1]) and one["flbl"] == batch["flbl"][3]


@pytest.mark.parametrize("case", CASES)
def test_decode_postprocessing_matches_reference_trainer(case):
    from crank_amd.net.trainer.basetrainer import BaseTrainer

    fx, g, corpus, scaler = load_case(case)
    scaler = _fix_var(scaler, g)
    B, blen = g("batch/in_feats").shape[:2]
    SPKRS = [str(v) for v in g("spkrs")]
    ftype = "mcep" if case == "mcep" else "mlfb"
    tr = object.__new__(BaseTrainer)
    tr.conf = {"output_feat_type": ftype, "use_mcep_0th": case != "mcep", "ignore_scaler": []}
    tr.scaler = _scaler_objects(scaler, ftype, SPKRS)
    tr.spkrs = {s: i for i, s in enumerate(SPKRS)}
    tr.device = "cuda"
    cu = lambda k: torch.from_numpy(g(k)).cuda()  # noqa: E731
    batch = {k: cu(f"batch/{k}") for k in ["in_feats", "lcf0", "uv"] + (["mcep_0th"] if case == "mcep" else [])}
    batch["flen"] = torch.clamp(cu("batch/flen"), max=blen)
    batch["org_spkr_name"] = [SPKRS[int(h)] for h in g("batch/org_h")[:, 0]]
    batch["flbl"] = [f"u{n}" for n in range(B)]
    tgt = SPKRS[int(g("target_spk"))]
    feats = tr._store_features(batch, {"decoded": cu("decoded")}, tgt)
    torch.cuda.synchronize()
    for n, f in enumerate(feats):
        for k in ["feats", "lcf0", "uv", "normed_lcf0", "normed_feat"] + (["rmcep"] if case == "mcep" else []):
            ref = g(f"store/{n}/{k}")
            got = f[k].cpu().numpy()
            assert got.dtype == ref.dtype and np.array_equal(got, ref), (n, k, got.dtype, ref.dtype)
        np.testing.assert_allclose(f["f0"].cpu().numpy(), g(f"store/{n}/f0"), rtol=1e-15)
    assert np.array_equal(tr._get_cvf0(batch, tgt).cpu().numpy(), g("cvf0"))


def test_collate_full_size_against_oracle_and_rate():
    """B=64, T=500, 80-dim, 14 speakers over a 300-utterance ragged corpus: bit-exact against the numpy
    oracle; prints the assembled bytes per second (algorithmic: every output byte written once, every
    kept input byte read once)."""
    rs = np.random.RandomState(7)
    S, D, T, B = 14, 80, 500, 64
    spkrs = [f"spk{i:02d}" for i in range(S)]
    lens = rs.randint(250, 701, size=300)
    lens[:4] = [500, 501, 1, 1400]
    corpus = [{"feat": (rs.standard_normal((n, D)) * 2 + 1).astype(np.float32), "lcf0": (5 + 0.3 * rs.standard_normal((n, 1))).astype(np.float32),
               "uv": (rs.uniform(size=(n, 1)) < 0.7).astype(np.float32), "spk": int(i % S)} for i, n in enumerate(lens)]
    allf = np.concatenate([u["feat"] for u in corpus]).astype(np.float64)
    alll = np.concatenate([u["lcf0"] for u in corpus]).astype(np.float64)
    scaler = {"feat_mean": allf.mean(0), "feat_scale": allf.std(0), "lcf0_mean": alll.mean(0), "lcf0_scale": alll.std(0),
              "spk_lcf0_mean": np.array([np.concatenate([u["lcf0"] for u in corpus if u["spk"] == s]).astype(np.float64).mean() for s in range(S)]),
              "spk_lcf0_var": np.array([np.concatenate([u["lcf0"] for u in corpus if u["spk"] == s]).astype(np.float64).var() for s in range(S)])}
    scaler["spk_lcf0_std"] = np.sqrt(scaler["spk_lcf0_var"])
    dset, _ = _dataset(corpus, scaler, "mlfb", spkrs, T)
    idx = [0, 1, 2, 3] + rs.randint(0, 300, size=B - 4).tolist()
    cvs = [(corpus[i]["spk"] + 1 + int(rs.randint(0, S - 1))) % S for i in idx]
    ps = [int(rs.randint(0, lens[i] - T)) if lens[i] > T else 0 for i in idx]
    batch = dset.assemble(idx, draws=[(spkrs[c], p) for c, p in zip(cvs, ps)])
    torch.cuda.synchronize()
    ref = ods.make_batch(corpus, scaler, S, T, idx, cvs, ps)
    for k in BATCH_KEYS:
        got = batch[k].cpu().numpy()
        assert got.dtype == ref[k].dtype and np.array_equal(got, ref[k]), k
    assert int(batch["encoder_mask"].sum()) == int(np.minimum(lens[idx], T).sum())
    # rate of the one launch (+ the four mask clones), steady state
    draws = [(spkrs[c], p) for c, p in zip(cvs, ps)]
    for _ in range(5):
        dset.assemble(idx, draws=draws)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 50
    for _ in range(n):
        dset.assemble(idx, draws=draws)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    out_bytes = sum(v.numel() * v.element_size() for v in batch.values() if isinstance(v, torch.Tensor))
    in_bytes = int(np.minimum(lens[idx], T).sum()) * (2 * D + 3) * 4
    print(f"collate B={B} T={T}: {dt * 1e6:.1f} us per batch incl. host draws/launch, {(out_bytes + in_bytes) / dt / 1e9:.1f} GB/s "
          f"algorithmic ({out_bytes / 1e6:.1f} MB out, {in_bytes / 1e6:.1f} MB in)")


def test_scaler_round_trip_and_loader():
    from crank_amd.net.trainer.dataset import DeviceLoader, scaler_apply

    g = torch.Generator().manual_seed(0)
    x = (torch.randn(1000, 80, generator=g) * 3 + 1).cuda()
    mean = torch.randn(80, generator=g, dtype=torch.float64).cuda()
    scale = (torch.rand(80, generator=g, dtype=torch.float64) + 0.5).cuda()
    y = scaler_apply(x, mean, scale)
    ref = ((x.double() - mean).float().double() / scale).float()
    assert torch.equal(y, ref)
    back = scaler_apply(y, mean, scale, inverse=True)
    assert float((back - x).abs().max()) < 1e-5
    fx, gg, corpus, scaler = load_case("mlfb")
    dset, _ = _dataset(corpus, _fix_var(scaler, gg), "mlfb", [str(v) for v in gg("spkrs")], 40)
    random.seed(1)
    torch.manual_seed(1)
    loader = DeviceLoader(dset, 5, shuffle=True)
    sizes = [b["in_feats"].shape[0] for b in loader]
    assert len(loader) == 3 and sizes == [5, 5, 2]


def test_use_raw_batches_match_padding_raw():
    """use_raw: the waveform row of every sample is the reference's padding_raw (golden cases: reflect padding incl.
    pads longer than the signal, the unpadded branch, zero extension, the cropped branch)."""
    from crank_amd.net.trainer.dataset import BaseDataset
    from tests.helpers import golden

    fx = golden("dataset.npz")
    n = int(fx["raw/n"])
    args = [[int(v) for v in fx[f"raw/args{k}"]] for k in range(n)]
    blen, fftl, hop = args[0][1:4]
    spk = ["A", "B"]
    files = {f"/nonexistent/h5/{spk[k % 2]}/u{k:02d}.h5": k for k in range(n)}

    def reader(h5f, ext="mlfb"):
        k = files[h5f]
        flen = args[k][0]
        if ext == "raw":
            return fx[f"raw/x{k}"]
        return np.full((flen, 4 if ext == "mlfb" else 1), float(k), np.float32)

    conf = {"batch_len": blen, "input_feat_type": "mlfb", "output_feat_type": "mlfb", "use_raw": True, "ignore_scaler": [],
            "spec_augment": False, "feature": {"fftl": fftl, "hop_size": hop}}
    dset = BaseDataset(conf, {"train": {"feats": {f: f for f in files}, "spkrs": spk}}, None, reader=reader)
    batch = dset.assemble(list(range(n)), draws=[(spk[(k + 1) % 2], args[k][4]) for k in range(n)])
    torch.cuda.synchronize()
    raw = batch["raw"].cpu().numpy()
    assert raw.shape == (n, fftl + hop * blen - 1) and "cv_lcf0" not in batch  # no scaler: no F0 statistics
    for k in range(n):
        assert np.array_equal(raw[k], fx[f"raw/out{k}"].astype(np.float32)), (k, args[k])
        assert np.array_equal(raw[k], ods.padding_raw(fx[f"raw/x{k}"], args[k][0], blen, fftl, hop, args[k][4]))
