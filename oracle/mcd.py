"""CPU oracle for the MCD evaluation.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Restates

* ``calculate`` of crank/bin/evaluate_mcd.py:45-79: voiced-frame selection (f0 > 0) of the converted and
  the ground-truth mel-cepstra, alignment with ``fastdtw(cv, gt, dist=scipy.spatial.distance.euclidean)``,
  ``MCD = mean_path 10 / ln 10 * sqrt(2 * sum_d (cv - gt)^2)``;
* the third-party ``fastdtw`` package the reference imports (crank/bin/evaluate_mcd.py:18,70).  It is an
  un-pinned transitive dependency (tools/requirements.txt lists ``sprocket-vc``, which depends on it), it is
  absent from /root/reference and from this image, and the reference holds no test or vector for it:
  **parity unpinned**.  The functions below follow its published pure-Python algorithm (Salvador & Chan's
  FastDTW as implemented by the ``fastdtw`` package: default radius 1; recursion on sequences halved by
  averaging neighbours until one is shorter than radius + 2, exact DTW there; the coarse path, widened by
  the radius and projected to the finer level, is the search window of the next DTW; among equal costs the
  predecessor order is (i-1, j), (i, j-1), (i-1, j-1)).

Distances are float64, accumulated over the dimensions in index order.
"""
import math

import numpy as np


def euclidean(u, v):
    s = 0.0
    for a, b in zip(u, v):
        s += (a - b) * (a - b)
    return math.sqrt(s)


def reduce_by_half(x):
    n = len(x) - len(x) % 2
    return [(x[i] + x[i + 1]) / 2 for i in range(0, n, 2)]


def expand_window(path, len_x, len_y, radius):
    cells = set(path)
    for i, j in path:
        for a in range(-radius, radius + 1):
            for b in range(-radius, radius + 1):
                cells.add((i + a, j + b))
    fine = set()
    for i, j in cells:
        fine.update(((2 * i, 2 * j), (2 * i, 2 * j + 1), (2 * i + 1, 2 * j), (2 * i + 1, 2 * j + 1)))
    window, start_j = [], 0
    for i in range(len_x):
        new_start = None
        for j in range(start_j, len_y):
            if (i, j) in fine:
                window.append((i, j))
                if new_start is None:
                    new_start = j
            elif new_start is not None:
                break
        start_j = new_start
    return window


def windowed_dtw(x, y, window):
    len_x, len_y = len(x), len(y)
    if window is None:
        window = [(i, j) for i in range(len_x) for j in range(len_y)]
    inf = float("inf")
    D = {(0, 0): (0.0, 0, 0)}
    get = lambda i, j: D.get((i, j), (inf,))[0]  # noqa: E731
    for i0, j0 in window:
        i, j = i0 + 1, j0 + 1
        dt = euclidean(x[i0], y[j0])
        best = (get(i - 1, j) + dt, i - 1, j)
        for cand in ((get(i, j - 1) + dt, i, j - 1), (get(i - 1, j - 1) + dt, i - 1, j - 1)):
            if cand[0] < best[0]:
                best = cand
        D[i, j] = best
    path = []
    i, j = len_x, len_y
    while not (i == 0 and j == 0):
        path.append((i - 1, j - 1))
        _, i, j = D[i, j]
    path.reverse()
    return D[len_x, len_y][0], path


def fastdtw(x, y, radius=1):
    x = [np.asarray(v, dtype=np.float64) for v in x]
    y = [np.asarray(v, dtype=np.float64) for v in y]
    return _fastdtw(x, y, radius)


def _fastdtw(x, y, radius):
    if len(x) < radius + 2 or len(y) < radius + 2:
        return windowed_dtw(x, y, None)
    _, path = _fastdtw(reduce_by_half(x), reduce_by_half(y), radius)
    return windowed_dtw(x, y, expand_window(path, len(x), len(y), radius))


def mcd(cv_mcep, cv_f0, gt_mcep, gt_f0, radius=1):
    """evaluate_mcd.py:61-77 for one utterance pair; returns (mcd, path)."""
    cv = np.asarray(cv_mcep, dtype=np.float64)[np.where(np.asarray(cv_f0).reshape(-1) > 0)[0]]
    gt = np.asarray(gt_mcep, dtype=np.float64)[np.where(np.asarray(gt_f0).reshape(-1) > 0)[0]]
    _, path = fastdtw(cv, gt, radius)
    twf = np.array(path).T
    diff2sum = np.sum((cv[twf[0]] - gt[twf[1]]) ** 2, 1)
    return float(np.mean(10.0 / np.log(10.0) * np.sqrt(2 * diff2sum), 0)), path
