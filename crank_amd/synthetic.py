"""Synthetic inputs for benchmarks and parity tests.

``make_batch`` builds the collated batch dict the trainers consume, with the
layout produced by the reference dataset (crank/net/trainer/dataset.py:76-139 for
the keys, :158-198 for the zero / False / -100 tail padding) but filled with
seeded pseudo-random data (z-scored features ~ N(0,1),
uv ~ Bernoulli(0.7), flen ~ U{250..700} clipped to batch_len, cv speaker != org).

``deterministic_state`` fills a state-dict with reproducible values that depend only
on the (sorted) key names and shapes, so that the golden-vector generator (which
runs the reference's classes) and the product load identical weights without
shipping megabytes of parameters as fixtures.

numpy's legacy ``RandomState`` stream is frozen by numpy policy, so these are stable
across numpy versions.
"""
import numpy as np
import torch


def make_batch(B, T, n_spkrs, in_dim=80, out_dim=None, seed=1234, min_len=None, max_len=None,
               use_raw=False, fftl=1024, hop_size=128, device="cpu", full_length=False):
    out_dim = in_dim if out_dim is None else out_dim
    rs = np.random.RandomState(seed)
    lo = int(0.5 * T) if min_len is None else min_len
    hi = int(1.4 * T) if max_len is None else max_len
    flen = np.minimum(rs.randint(lo, hi + 1, size=B), T).astype(np.int64)
    if full_length:
        flen[:] = T
    feats = rs.standard_normal((B, T, in_dim)).astype(np.float32)
    lcf0 = rs.standard_normal((B, T, 1)).astype(np.float32)
    cv_lcf0 = rs.standard_normal((B, T, 1)).astype(np.float32)
    uv = (rs.uniform(size=(B, T, 1)) < 0.7).astype(np.float32)
    org = rs.randint(0, n_spkrs, size=B)
    cv = (org + rs.randint(1, n_spkrs, size=B)) % n_spkrs if n_spkrs > 1 else org.copy()
    mask = np.arange(T)[None, :] < flen[:, None]  # (B,T)
    m3 = mask[:, :, None]
    feats = feats * m3
    lcf0, cv_lcf0, uv = lcf0 * m3, cv_lcf0 * m3, uv * m3
    org_h = np.where(mask, org[:, None], -100).astype(np.int64)
    cv_h = np.where(mask, cv[:, None], -100).astype(np.int64)
    eye = np.eye(n_spkrs, dtype=np.float32)
    org_oh = eye[org][:, None, :] * m3
    cv_oh = eye[cv][:, None, :] * m3
    if out_dim == in_dim:
        out_feats = feats.copy()
    else:
        out_feats = (rs.standard_normal((B, T, out_dim)).astype(np.float32)) * m3
    batch = {
        "in_feats": feats,
        "out_feats": out_feats,
        "lcf0": lcf0.astype(np.float32),
        "cv_lcf0": cv_lcf0.astype(np.float32),
        "uv": uv.astype(np.float32),
        "org_h": org_h,
        "cv_h": cv_h,
        "org_h_onehot": org_oh.astype(np.float32),
        "cv_h_onehot": cv_oh.astype(np.float32),
        "flen": flen,
    }
    for k in ["encoder_mask", "decoder_mask", "cycle_encoder_mask", "cycle_decoder_mask"]:
        batch[k] = m3.copy()
    if use_raw:
        n = fftl + hop_size * T - 1  # dataset.py:262
        raw = (0.1 * rs.standard_normal((B, n))).astype(np.float32)
        batch["raw"] = raw
    out = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in batch.items()}
    out["flbl"] = [f"spk{int(o)}/utt{i:05d}" for i, o in enumerate(org)]
    out["org_spkr_name"] = [f"spk{int(o)}" for o in org]
    out["cv_spkr_name"] = [f"spk{int(c)}" for c in cv]
    return out


def deterministic_state(shapes, seed=4321):
    """shapes: mapping name -> tuple shape.  Returns name -> float32 ndarray."""
    rs = np.random.RandomState(seed)
    out = {}
    pending_g = []
    for name in sorted(shapes):
        shp = tuple(shapes[name])
        n = int(np.prod(shp)) if len(shp) else 1
        z = rs.standard_normal(n).reshape(shp)
        if name.endswith("weight_v") or (name.endswith(".weight") and len(shp) == 3):
            fan_in = int(np.prod(shp[1:]))
            out[name] = (z * np.sqrt(2.0 / fan_in)).astype(np.float32)
        elif name.endswith("weight_g"):
            out[name] = (0.8 + 0.4 * rs.uniform(size=shp)).astype(np.float32)
            pending_g.append(name)
        elif name.endswith("bias"):
            out[name] = (0.05 * z).astype(np.float32)
        elif name.endswith("ema_size"):
            out[name] = np.zeros(shp, dtype=np.float32)  # reference init, vqvae2.py:303
        elif name.endswith("ema_w"):
            out[name] = z.astype(np.float32)  # reference init, vqvae2.py:302-304
        elif "quantizers" in name and name.endswith("embedding.weight"):
            out[name] = (0.6 * z).astype(np.float32)
        elif name in ("mean",):
            out[name] = (0.1 * z).astype(np.float32)
        elif name in ("std",):
            out[name] = (1.0 + 0.1 * np.abs(z)).astype(np.float32)
        else:
            out[name] = z.astype(np.float32)
    # weight_g := factor * ||v|| so that effective weights have kaiming-like scale
    for gname in pending_g:
        vname = gname[: -len("weight_g")] + "weight_v"
        if vname in out:
            v = out[vname]
            nrm = np.sqrt((v.astype(np.float64) ** 2).reshape(v.shape[0], -1).sum(1)).reshape(out[gname].shape)
            out[gname] = (out[gname] * nrm).astype(np.float32)
    return out


def load_deterministic(module, seed=4321):
    sd = module.state_dict()
    vals = deterministic_state({k: tuple(v.shape) for k, v in sd.items() if v.dtype.is_floating_point}, seed)
    new = {k: (torch.from_numpy(vals[k]).to(v.device) if k in vals else v) for k, v in sd.items()}
    module.load_state_dict(new)
    return module
