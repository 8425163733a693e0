"""Shared comparison helpers for the parity tests."""
import numpy as np


def knn_rows_equivalent(idx_a, idx_b, dist_fn, rtol=0.0, atol=0.0):
    """Tie-aware comparison of two (rows, k) neighbour lists.

    Rows that differ are accepted iff the float64 distances of both selections agree
    position by position within (rtol, atol) -- i.e. they differ only by the order /
    choice among exactly (or, with a tolerance, nearly) tied candidates.
    Returns (n_identical_rows, n_tie_rows, n_bad_rows)."""
    ident = tie = bad = 0
    for r in range(idx_a.shape[0]):
        a, b = idx_a[r], idx_b[r]
        if np.array_equal(a, b):
            ident += 1
            continue
        da, db = np.sort(dist_fn(r, a)), np.sort(dist_fn(r, b))
        if np.allclose(da, db, rtol=rtol, atol=atol):
            tie += 1
        else:
            bad += 1
    return ident, tie, bad


def sqdist64(x):
    """x (C,N) -> callable(row, idx) giving float64 squared distances from point `row`."""
    x64 = x.astype(np.float64)

    def f(r, idx):
        d = x64[:, idx] - x64[:, r:r + 1]
        return (d * d).sum(0)

    return f


def pn_metric64(x):
    """knn_points_normals metric (M4:62-75) in float64."""
    x64 = x.astype(np.float64)

    def f(r, idx):
        p, n = x64[0:3], x64[3:6]
        d = p[:, idx] - p[:, r:r + 1]
        pp = (d * d).sum(0)
        nn = 2.0 - 2.0 * (n[:, idx] * n[:, r:r + 1]).sum(0)
        return pp * (1.0 + nn)

    return f
