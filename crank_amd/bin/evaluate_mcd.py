"""MCD between converted and ground-truth mel-cepstra (crank/bin/evaluate_mcd.py:45-79) for a batch of
utterance pairs in one launch: voiced-frame selection on the host (the features come from files), FastDTW
alignment and the distortion on the device (crk_mcd_fastdtw, one wavefront per pair).

The reference's file handling (HDF5 / WORLD analysis of converted waveforms, feats.scp lookup, per-pair
summary) is not reproduced; ``mcd_fastdtw`` takes the arrays ``calculate()`` works on.
"""
import numpy as np
import torch

from crank_amd import _lib
from crank_amd._lib import check, ptr, stream_ptr


def mcd_fastdtw(cv_mceps, cv_f0s, gt_mceps, gt_f0s, radius=1, return_paths=False, device="cuda"):
    """Lists of per-utterance arrays: mcep (n, D), f0 (n,) or (n, 1).  Returns the list of MCDs in dB
    (and the warping paths as (len, 2) int arrays)."""
    P = len(cv_mceps)
    if not (P == len(cv_f0s) == len(gt_mceps) == len(gt_f0s)) or P == 0:
        raise ValueError("need the same, non-zero number of converted and ground-truth utterances")
    cv, gt = [], []
    for m, f, out in [(a, b, cv) for a, b in zip(cv_mceps, cv_f0s)] + [(a, b, gt) for a, b in zip(gt_mceps, gt_f0s)]:
        m = np.asarray(m, dtype=np.float64)
        out.append(np.ascontiguousarray(m[np.where(np.asarray(f).reshape(-1) > 0)[0]]))  # evaluate_mcd.py:64-67
    D = cv[0].shape[1]
    for a in cv + gt:
        if a.ndim != 2 or a.shape[1] != D:
            raise ValueError("all mel-cepstra must be (frames, D) with the same D")
        if a.shape[0] == 0:
            raise ValueError("an utterance has no voiced frame")
    nx, ny = [a.shape[0] for a in cv], [a.shape[0] for a in gt]
    dev = torch.device(device)
    up = lambda arrs: torch.as_tensor(np.concatenate(arrs), device=dev)  # noqa: E731
    off = lambda ns: torch.as_tensor(np.concatenate([[0], np.cumsum(ns)]).astype(np.int64), device=dev)  # noqa: E731
    x, y, xo, yo = up(cv), up(gt), off(nx), off(ny)
    L = _lib.lib()
    mx, my = max(nx), max(ny)
    nbytes = L.crk_mcd_scratch_bytes(P, mx, my, D, radius)
    if nbytes < 0:
        raise ValueError("unsupported sizes")
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    mcd = torch.empty(P, dtype=torch.float64, device=dev)
    plen = torch.empty(P, dtype=torch.int32, device=dev)
    status = torch.empty(P, dtype=torch.int32, device=dev)
    stride = 2 * (mx + my)
    paths = torch.empty(P, stride, dtype=torch.int32, device=dev) if return_paths else None
    check(L.crk_mcd_fastdtw(ptr(x), ptr(xo), ptr(y), ptr(yo), P, D, radius, mx, my, ptr(mcd), ptr(plen), ptr(paths), stride,
                            ptr(scratch), ptr(status), stream_ptr()), "mcd_fastdtw")
    if int(status.max()) != 0:
        raise RuntimeError("crk_mcd_fastdtw: a pair exceeded the scratch sizing")
    vals = mcd.cpu().tolist()
    if not return_paths:
        return vals
    lens = plen.cpu().tolist()
    pc = paths.cpu().numpy()
    return vals, [pc[i, : 2 * lens[i]].reshape(-1, 2) for i in range(P)]
