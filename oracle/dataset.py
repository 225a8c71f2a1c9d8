"""CPU oracle (numpy) for batch assembly and decode-side post-processing.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates, for utterances held in
memory instead of HDF5 files,

* ``BaseDataset.__getitem__`` + torch's default collate
  (crank/net/trainer/dataset.py:58-139 for the sample dict, :158-198 and :239-258 for
  the pad / crop rule, :229-236 for the one-hot code, :288-293 for ``convert_f0``);
* sklearn's ``StandardScaler.transform / inverse_transform`` as the reference calls them
  on float32 features (dataset.py:146-150, basetrainer.py:341-345): the in-place
  ``X -= mean_; X /= scale_`` on a float32 copy evaluates each operation in float64 and
  rounds to float32 after it;
* ``BaseTrainer._store_features`` / ``_get_cvf0`` (crank/net/trainer/basetrainer.py:311-320,
  :340-386).

Pinned by tests/golden/dataset.npz, which is produced by the reference's own
``BaseDataset`` / ``convert_f0`` / sklearn scalers (tests/golden/make_golden.py).

Deliberate difference, documented in DESIGN.md: when an utterance is exactly ``batch_len``
frames long the reference's ``padding`` returns its input unconverted (dataset.py:243-249),
so ``cv_lcf0`` stays float64 for that sample; here every continuous output is float32.
"""
import numpy as np


def scaler_transform(x, mean, scale):
    """StandardScaler.transform on a float32 array: fl32(fl64(fl32(fl64(x) - mean)) / scale)."""
    x = np.asarray(x, dtype=np.float32)
    y = (x.astype(np.float64) - mean).astype(np.float32)
    return (y.astype(np.float64) / scale).astype(np.float32)


def scaler_inverse(x, mean, scale):
    """StandardScaler.inverse_transform on a float32 array: scale first, then shift."""
    x = np.asarray(x, dtype=np.float32)
    y = (x.astype(np.float64) * scale).astype(np.float32)
    return (y.astype(np.float64) + mean).astype(np.float32)


def convert_f0(lcf0, mean_org, std_org, mean_cv, std_cv):
    """dataset.py:288-293 with std = sqrt(var_) taken by the caller; float64 result."""
    return (np.asarray(lcf0).astype(np.float64) - mean_org) / std_org * std_cv + mean_cv


def pad_or_crop(x, flen, blen, value, p):
    """dataset.py:158-198 + :239-258: tail-pad with ``value`` up to blen, or keep blen frames from p."""
    if flen > blen:
        return x[p : p + blen].copy()
    out = np.full((blen,) + x.shape[1:], value, dtype=x.dtype)
    out[:flen] = x
    return out


def padding_raw(x, flen, blen, fftl, hop, p):
    """dataset.py:261-285 with dlen = blen - flen: target length fftl + hop * blen - 1."""
    x = np.asarray(x).reshape(-1)
    target = fftl + hop * blen - 1
    if blen - flen > 0 or p == 0:
        if len(x) < target - fftl:
            x = np.pad(x, fftl // 2, mode="reflect")
    else:
        x = np.concatenate([np.zeros(fftl // 2), x[p * hop :]])
    if len(x) < target:
        x = np.concatenate([x, np.zeros(target - len(x))])
    return x[:target].astype(np.float32)


def get_item(utt, scaler, n_spkrs, blen, cv_spk, p, drop_0th=False):
    """One sample.  utt: dict feat (flen, D) f32 raw, lcf0 (flen, 1) f32 raw, uv (flen, 1) f32, spk int.
    scaler: dict feat_mean, feat_scale (D) | None, lcf0_mean, lcf0_scale (1) | None,
    spk_lcf0_mean, spk_lcf0_std (n_spkrs) float64.  cv_spk / p: the two random draws of
    dataset.py:84-86 and :161, made by the caller."""
    flen = utt["feat"].shape[0]
    org = int(utt["spk"])
    cv = convert_f0(utt["lcf0"], scaler["spk_lcf0_mean"][org], scaler["spk_lcf0_std"][org],
                    scaler["spk_lcf0_mean"][cv_spk], scaler["spk_lcf0_std"][cv_spk]).astype(np.float32)
    feat = utt["feat"] if scaler.get("feat_mean") is None else scaler_transform(utt["feat"], scaler["feat_mean"], scaler["feat_scale"])
    lcf0 = utt["lcf0"] if scaler.get("lcf0_mean") is None else scaler_transform(utt["lcf0"], scaler["lcf0_mean"], scaler["lcf0_scale"])
    s = {"flen": flen}
    if drop_0th:  # dataset.py:108-110
        s["mcep_0th"] = pad_or_crop(feat[:, :1], flen, blen, 0.0, p)
        feat = feat[:, 1:]
    s["in_feats"] = pad_or_crop(np.ascontiguousarray(feat), flen, blen, 0.0, p)
    s["out_feats"] = s["in_feats"].copy()
    s["lcf0"] = pad_or_crop(lcf0.astype(np.float32), flen, blen, 0.0, p)
    s["uv"] = pad_or_crop(utt["uv"].astype(np.float32), flen, blen, 0.0, p)
    s["cv_lcf0"] = pad_or_crop(cv, flen, blen, 0.0, p)
    mask = pad_or_crop(np.ones((flen, 1), dtype=bool), flen, blen, False, p)
    for k in ("encoder_mask", "decoder_mask", "cycle_encoder_mask", "cycle_decoder_mask"):
        s[k] = mask.copy()
    eye = np.eye(n_spkrs, dtype=np.float32)
    for name, c in (("org", org), ("cv", cv_spk)):
        s[f"{name}_h"] = pad_or_crop(np.full(flen, c, dtype=np.int64), flen, blen, -100, p)
        s[f"{name}_h_onehot"] = pad_or_crop(np.tile(eye[c], (flen, 1)), flen, blen, 0.0, p)
    return s


def collate(samples):
    """torch default_collate on ndarray / int fields: stack along a new leading axis."""
    out = {}
    for k in samples[0]:
        v = [s[k] for s in samples]
        out[k] = np.asarray(v, dtype=np.int64) if k == "flen" else np.stack(v)
    return out


def make_batch(corpus, scaler, n_spkrs, blen, utt_ids, cv_spks, crops, drop_0th=False):
    return collate([get_item(corpus[u], scaler, n_spkrs, blen, int(c), int(p), drop_0th)
                    for u, c, p in zip(utt_ids, cv_spks, crops)])


def store_features(decoded, lcf0, uv, flen, org_spk, cv_spk, scaler, mcep_0th=None, in_feats=None):
    """basetrainer.py:340-386 for one utterance of a batch (arrays already cut to [:flen] by the caller
    or cut here).  Returns the dict the reference hands to its HDF5 / vocoder writers."""
    feat = np.asarray(decoded[:flen], dtype=np.float32)
    out = {}
    if mcep_0th is not None:  # basetrainer.py:360-366
        z = np.asarray(mcep_0th[:flen], dtype=np.float32)
        feat = np.ascontiguousarray(np.hstack([z, feat]))
        rm = np.ascontiguousarray(np.hstack([z, np.asarray(in_feats[:flen], dtype=np.float32)]))
        out["rmcep"] = rm if scaler.get("feat_mean") is None else scaler_inverse(rm, scaler["feat_mean"], scaler["feat_scale"])
    out["feats"] = feat if scaler.get("feat_mean") is None else scaler_inverse(feat, scaler["feat_mean"], scaler["feat_scale"])
    l = np.asarray(lcf0[:flen], dtype=np.float32)
    org_cf0 = l if scaler.get("lcf0_mean") is None else scaler_inverse(l, scaler["lcf0_mean"], scaler["lcf0_scale"])
    cv_cf0 = convert_f0(org_cf0, scaler["spk_lcf0_mean"][org_spk], scaler["spk_lcf0_std"][org_spk],
                        scaler["spk_lcf0_mean"][cv_spk], scaler["spk_lcf0_std"][cv_spk])
    out["lcf0"] = cv_cf0
    out["uv"] = np.asarray(uv[:flen], dtype=np.float32)
    out["f0"] = np.exp(cv_cf0) * out["uv"]
    # StandardScaler.transform of a float64 array stays in float64 (basetrainer.py:384)
    out["normed_lcf0"] = (cv_cf0 - scaler["lcf0_mean"]) / scaler["lcf0_scale"] if scaler.get("lcf0_mean") is not None else cv_cf0
    out["normed_feat"] = feat
    return out
