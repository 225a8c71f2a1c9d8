"""oracle/mcd.py (the FastDTW restatement, parity unpinned: the package is absent) checked for what can be
checked without it: against an exhaustive DTW where the window covers everything, path validity, and
against an interval formulation of the same algorithm (per-row [lo, hi] windows, rolling cost rows, one
back pointer per cell) that the device kernel crk_mcd_fastdtw implements line for line."""
import numpy as np
import pytest

from oracle import mcd as om


def exact_dtw_cost(x, y):
    n, m = len(x), len(y)
    D = np.full((n + 1, m + 1), np.inf)
    D[0, 0] = 0.0
    for i in range(1, n + 1):
        for j in range(1, m + 1):
            D[i, j] = om.euclidean(x[i - 1], y[j - 1]) + min(D[i - 1, j], D[i, j - 1], D[i - 1, j - 1])
    return D[n, m]


def interval_fastdtw(x, y, radius=1):
    """The device kernel's formulation (crank_amd/csrc/mcd_kernels.hip) in Python."""
    x = [np.asarray(v, dtype=np.float64) for v in x]
    y = [np.asarray(v, dtype=np.float64) for v in y]
    xs, ys = [x], [y]
    while len(xs[-1]) >= radius + 2 and len(ys[-1]) >= radius + 2:
        xs.append(om.reduce_by_half(xs[-1]))
        ys.append(om.reduce_by_half(ys[-1]))
    L = len(xs) - 1
    lo, hi = [0] * len(xs[L]), [len(ys[L]) - 1] * len(xs[L])
    path = None
    for l in range(L, -1, -1):
        xl, yl = xs[l], ys[l]
        nx, ny = len(xl), len(yl)
        back, prev, plo, phi = [], None, 0, -1
        for i in range(nx):
            cur, brow = [], []
            for c, j in enumerate(range(lo[i], hi[i] + 1)):
                dt = om.euclidean(xl[i], yl[j])
                if i == 0:
                    up, diag = np.inf, (0.0 if j == 0 else np.inf)
                else:
                    up = prev[j - plo] if plo <= j <= phi else np.inf
                    diag = prev[j - 1 - plo] if plo <= j - 1 <= phi else np.inf
                left = cur[c - 1] if c > 0 else np.inf
                best, bp = up + dt, 0
                if left + dt < best:
                    best, bp = left + dt, 1
                if diag + dt < best:
                    best, bp = diag + dt, 2
                cur.append(best)
                brow.append(bp)
            back.append(brow)
            prev, plo, phi = cur, lo[i], hi[i]
        i, j, path = nx - 1, ny - 1, []
        while i >= 0 and j >= 0:
            path.append((i, j))
            bp = back[i][j - lo[i]]
            if bp == 0:
                i -= 1
            elif bp == 1:
                j -= 1
            else:
                i, j = i - 1, j - 1
        path.reverse()
        if l > 0:
            jmin, jmax = [10 ** 9] * nx, [-1] * nx
            for ci, cj in path:
                jmin[ci], jmax[ci] = min(jmin[ci], cj), max(jmax[ci], cj)
            nxf, nyf = len(xs[l - 1]), len(ys[l - 1])
            lo, hi, start_j = [], [], 0
            for fi in range(nxf):
                ci = fi >> 1
                rows = [cr for cr in range(ci - radius, ci + radius + 1) if 0 <= cr < nx and jmax[cr] >= 0]
                e0, e1 = min(jmin[cr] for cr in rows) - radius, max(jmax[cr] for cr in rows) + radius
                f0, f1 = max(2 * e0, start_j), min(2 * e1 + 1, nyf - 1)
                lo.append(f0)
                hi.append(f1)
                start_j = f0
    return path


@pytest.mark.parametrize("seed", range(6))
def test_fastdtw_restatement_properties(seed):
    rs = np.random.RandomState(seed)
    for nx, ny in [(1, 1), (1, 7), (2, 9), (3, 3), (5, 4), (17, 40), (64, 33), (101, 96), (37, 150)]:
        D = 4
        base = np.cumsum(rs.standard_normal((max(nx, ny) + 8, D)), 0)
        x = base[rs.choice(len(base), nx, replace=True)][np.argsort(rs.uniform(size=nx))] if seed % 2 else base[:nx] + 0.1 * rs.standard_normal((nx, D))
        y = base[:ny] + 0.1 * rs.standard_normal((ny, D))
        cost, path = om.fastdtw(x, y, radius=1)
        # a valid warping path: ends fixed, steps in {(1,0),(0,1),(1,1)}
        assert path[0] == (0, 0) and path[-1] == (nx - 1, ny - 1)
        for (a, b), (c, d) in zip(path[:-1], path[1:]):
            assert (c - a, d - b) in ((1, 0), (0, 1), (1, 1))
        # its cost is the sum along the path and can only exceed the exhaustive optimum
        assert np.isclose(cost, sum(om.euclidean(x[i], y[j]) for i, j in path), rtol=1e-12)
        opt = exact_dtw_cost(x, y)
        assert cost >= opt * (1 - 1e-12)
        # with a radius as large as the sequences the window covers every cell: exact
        cost_full, _ = om.fastdtw(x, y, radius=max(nx, ny))
        assert np.isclose(cost_full, opt, rtol=1e-12)
        # the interval formulation the kernel uses gives the same path, cell for cell
        assert interval_fastdtw(x, y, radius=1) == path, (nx, ny)
        if min(nx, ny) >= 4:
            assert interval_fastdtw(x, y, radius=2) == om.fastdtw(x, y, radius=2)[1]


def test_mcd_formula_and_voiced_selection():
    rs = np.random.RandomState(3)
    cv, gt = rs.standard_normal((30, 5)), rs.standard_normal((26, 5))
    f0c, f0g = (rs.uniform(size=30) < 0.8) * 120.0, (rs.uniform(size=26) < 0.8) * 110.0
    val, path = om.mcd(cv, f0c, gt, f0g)
    a, b = cv[f0c > 0], gt[f0g > 0]
    ref = np.mean([10.0 / np.log(10.0) * np.sqrt(2 * np.sum((a[i] - b[j]) ** 2)) for i, j in path])
    assert np.isclose(val, ref, rtol=1e-13)
    # identical sequences: zero distortion along the diagonal
    val0, path0 = om.mcd(cv, np.ones(30), cv, np.ones(30))
    assert val0 == 0.0 and path0 == [(i, i) for i in range(30)]
