"""crk_mcd_fastdtw (MCD with FastDTW alignment on the device) against oracle/mcd.py:
identical warping paths, MCD to 1e-12 relative (the mean is a wave-order sum), ragged batch with the
degenerate lengths (1, 2, 3 frames; odd / even; very different lengths)."""
import time

import numpy as np
import pytest

from oracle import mcd as om

pytestmark = pytest.mark.gpu


def _pairs(rs, shapes, D):
    cv, gt, f0c, f0g = [], [], [], []
    for nx, ny in shapes:
        t = np.cumsum(rs.standard_normal((max(nx, ny) + 40, D)) * 0.3, 0)
        # voiced masks drop ~20 % of the frames; lengths above are the VOICED counts
        def with_unvoiced(seq, n):
            total = n + rs.randint(0, 6)
            f0 = np.zeros(total)
            f0[rs.choice(total, n, replace=False)] = 100.0 + rs.uniform(size=n)
            full = rs.standard_normal((total, D))
            full[f0 > 0] = seq
            return full, f0
        a = t[np.sort(rs.choice(len(t), nx, replace=True))] + 0.05 * rs.standard_normal((nx, D))
        b = t[np.sort(rs.choice(len(t), ny, replace=True))] + 0.05 * rs.standard_normal((ny, D))
        A, fa = with_unvoiced(a, nx)
        B, fb = with_unvoiced(b, ny)
        cv.append(A); f0c.append(fa); gt.append(B); f0g.append(fb)
    return cv, f0c, gt, f0g


def test_mcd_fastdtw_matches_oracle_paths_and_values():
    from crank_amd.bin.evaluate_mcd import mcd_fastdtw

    rs = np.random.RandomState(0)
    shapes = [(1, 1), (1, 9), (2, 2), (3, 2), (3, 3), (4, 7), (5, 5), (33, 64), (97, 100), (128, 31), (201, 255), (400, 377)]
    cv, f0c, gt, f0g = _pairs(rs, shapes, 35)
    vals, paths = mcd_fastdtw(cv, f0c, gt, f0g, return_paths=True)
    for p, (nx, ny) in enumerate(shapes):
        ref, rpath = om.mcd(cv[p], f0c[p], gt[p], f0g[p])
        assert [tuple(v) for v in paths[p].tolist()] == rpath, (nx, ny)
        assert np.isclose(vals[p], ref, rtol=1e-12, atol=0), (nx, ny, vals[p], ref)
    # radius 2 as well (wider windows, other base-level sizes)
    vals2, paths2 = mcd_fastdtw(cv[6:10], f0c[6:10], gt[6:10], f0g[6:10], radius=2, return_paths=True)
    for k, p in enumerate(range(6, 10)):
        ref, rpath = om.mcd(cv[p], f0c[p], gt[p], f0g[p], radius=2)
        assert [tuple(v) for v in paths2[k].tolist()] == rpath
        assert np.isclose(vals2[k], ref, rtol=1e-12)


def test_mcd_fastdtw_evaluation_sized_batch():
    """560 pairs of ~600 voiced frames x 35 coefficients (VCC2018: 35 utterances x 4 x 4 speaker pairs):
    one launch; identical sequences give 0 dB along the diagonal; prints the time."""
    from crank_amd.bin.evaluate_mcd import mcd_fastdtw
    import torch

    rs = np.random.RandomState(1)
    shapes = [(int(n), int(m)) for n, m in zip(rs.randint(400, 800, 560), rs.randint(400, 800, 560))]
    cv, f0c, gt, f0g = _pairs(rs, shapes, 35)
    cv[0], f0c[0] = gt[0].copy(), f0g[0].copy()
    vals = mcd_fastdtw(cv, f0c, gt, f0g)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    vals = mcd_fastdtw(cv, f0c, gt, f0g)
    dt = time.perf_counter() - t0
    assert vals[0] == 0.0 and all(np.isfinite(v) and v > 0 for v in vals[1:])
    ref, _ = om.mcd(cv[7], f0c[7], gt[7], f0g[7])
    assert np.isclose(vals[7], ref, rtol=1e-12)
    print(f"MCD + FastDTW of 560 pairs (~600 x 600 frames, D=35): {dt * 1e3:.1f} ms incl. host packing and upload")
