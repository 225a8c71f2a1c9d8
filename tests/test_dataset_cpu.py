"""oracle/dataset.py against tests/golden/dataset.npz (outputs of the reference's own BaseDataset,
convert_f0, sklearn scalers, BaseTrainer._store_features / _get_cvf0; see make_golden.py)."""
import numpy as np
import pytest

from tests.helpers import golden
from oracle import dataset as ods


def load_case(case):
    fx = golden("dataset.npz")
    g = lambda k: fx[f"{case}/{k}"]  # noqa: E731
    lens = g("lens")
    starts = np.concatenate([[0], np.cumsum(lens)])
    corpus = [{"feat": g("feat")[a:b], "lcf0": g("lcf0")[a:b, None], "uv": g("uv")[a:b, None], "spk": int(s)}
              for a, b, s in zip(starts[:-1], starts[1:], g("utt_spk"))]
    scaler = {"feat_mean": g("feat_mean"), "feat_scale": g("feat_scale"), "lcf0_mean": g("lcf0_mean"),
              "lcf0_scale": g("lcf0_scale"), "spk_lcf0_mean": g("spk_lcf0_mean"),
              "spk_lcf0_std": np.sqrt(g("spk_lcf0_var"))}
    return fx, g, corpus, scaler


CASES = ["mlfb", "mcep", "pkl"]  # "pkl": normalised with the reference's own test/data/scaler.pkl (80-dim mlfb, 12 VCC2018 speakers)
BATCH_KEYS = ["in_feats", "out_feats", "lcf0", "uv", "cv_lcf0", "org_h", "cv_h", "org_h_onehot", "cv_h_onehot",
              "encoder_mask", "decoder_mask", "cycle_encoder_mask", "cycle_decoder_mask", "flen"]


@pytest.mark.parametrize("case", CASES)
def test_batch_assembly_matches_reference_dataset(case):
    fx, g, corpus, scaler = load_case(case)
    d = g("draws_utt_cv_p")
    blen = g("batch/in_feats").shape[1]
    got = ods.make_batch(corpus, scaler, len(g("spkrs")), blen, d[:, 0], d[:, 1], d[:, 2], drop_0th=(case == "mcep"))
    for k in BATCH_KEYS + (["mcep_0th"] if case == "mcep" else []):
        ref = g(f"batch/{k}")
        assert got[k].shape == ref.shape, k
        assert got[k].dtype == ref.dtype, (k, got[k].dtype, ref.dtype)
        assert np.array_equal(got[k], ref), (k, np.abs(got[k].astype(np.float64) - ref).max())
    # the fixture covers: shorter, equal, one longer, much longer than batch_len, a 1-frame utterance
    lens = g("lens")
    assert (lens < blen).any() and (lens == blen).any() and (lens == blen + 1).any() and (lens > 2 * blen).any() and (lens == 1).any()
    assert (d[:, 2] > 0).any()


@pytest.mark.parametrize("case", CASES)
def test_decode_postprocessing_matches_reference_trainer(case):
    fx, g, corpus, scaler = load_case(case)
    B, blen = g("batch/in_feats").shape[:2]
    tgt = int(g("target_spk"))
    flen = np.minimum(g("batch/flen"), blen)
    cvf0 = []
    for n in range(B):
        org = int(g("batch/org_h")[n, 0])
        kw = dict(mcep_0th=g("batch/mcep_0th")[n], in_feats=g("batch/in_feats")[n]) if case == "mcep" else {}
        got = ods.store_features(g("decoded")[n], g("batch/lcf0")[n], g("batch/uv")[n], int(flen[n]), org, tgt, scaler, **kw)
        for k in ["feats", "lcf0", "uv", "normed_lcf0", "normed_feat"] + (["rmcep"] if case == "mcep" else []):
            ref = g(f"store/{n}/{k}")
            assert got[k].dtype == ref.dtype and np.array_equal(got[k], ref), (n, k)
        np.testing.assert_allclose(got["f0"], g(f"store/{n}/f0"), rtol=1e-15)
        full = ods.store_features(g("decoded")[n], g("batch/lcf0")[n], g("batch/uv")[n], blen, org, tgt, scaler, **kw)
        cvf0.append(full["normed_lcf0"].astype(np.float32))
    assert np.array_equal(np.stack(cvf0), g("cvf0"))


def test_padding_raw_matches_reference():
    fx = golden("dataset.npz")
    for k in range(int(fx["raw/n"])):
        flen, blen, fftl, hop, p = [int(v) for v in fx[f"raw/args{k}"]]
        got = ods.padding_raw(fx[f"raw/x{k}"], flen, blen, fftl, hop, p)
        ref = fx[f"raw/out{k}"]
        assert got.shape == ref.shape == (fftl + hop * blen - 1,)
        assert np.array_equal(got, ref.astype(np.float32)), k
