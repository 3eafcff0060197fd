"""Sigma-point tables handed to the kernels (host side, built once per call).

Interface of matlab/symmetric-cubature-rules/utp_ws.m ([W,SX] = utp_ws(p,n)) and
matlab/mvhermgauss.m; the fully symmetric sets are enumerated combinatorially (support choice x
sign pattern) instead of the reference's recursive sym_set.m, so the point ORDER differs from the
reference -- the weighted sums the kernels form do not depend on it.  The 9th-order centre weight
keeps the reference's sign typo (ut9_ws.m:78-79, SURVEY C-1) unless quirks=False, because every
reference result with p=9 and n>=3 includes it.
"""
import itertools
import math

import numpy as np


def _fs_set(n, gens):
    """All points with the generator values placed on len(gens) distinct coordinates (every
    assignment of distinct generator values to positions) and all sign patterns: (n, npts)."""
    k = len(gens)
    if k == 0:
        return np.zeros((n, 1))
    if k > n:
        return np.zeros((n, 0))
    pts = []
    distinct_perms = sorted(set(itertools.permutations(gens)))
    for support in itertools.combinations(range(n), k):
        for perm in distinct_perms:
            for signs in itertools.product((1.0, -1.0), repeat=k):
                x = np.zeros(n)
                for pos, g, s in zip(support, perm, signs):
                    x[pos] = s * g
                pts.append(x)
    return np.stack(pts, axis=1)


def _binom(n, k):
    # the reference's ndownk: prod(n-k+1:n)/k!  (may take n < k, SURVEY C-2)
    out = 1.0
    for v in range(n - k + 1, n + 1):
        out *= v
    return out / math.factorial(k)


def _uv(c4, c2, c0):
    """positive roots of c4 x^4 + c2 x^2 + c0, larger first (u), smaller second (v)."""
    disc = math.sqrt(c2 * c2 - 4.0 * c4 * c0)
    r = sorted([(-c2 + disc) / (2.0 * c4), (-c2 - disc) / (2.0 * c4)], reverse=True)
    return math.sqrt(r[0]), math.sqrt(r[1])


def utp_ws(p, n, quirks=True):
    """[W, SX] = utp_ws(p, n): weights (npts,), unit sigma points (n, npts)."""
    if p == 3:
        W = np.concatenate([[0.0], np.full(2 * n, 1.0 / (2.0 * n))])
        SX = math.sqrt(n) * np.hstack([np.zeros((n, 1)), np.eye(n), -np.eye(n)])
        return W, SX
    if p == 5:
        u = math.sqrt(3.0)
        A0 = 1.0 - n / 9.0 * (3.0 - 0.5 * (n - 1))
        A1 = (3.0 - (n - 1)) / 18.0
        A11 = 1.0 / 36.0
        sets = [(_fs_set(n, ()), A0), (_fs_set(n, (u,)), A1), (_fs_set(n, (u, u)), A11)]
    elif p == 7:
        u, v = _uv(-2.0, 12.0, -6.0)
        u2, u4, u6 = u ** 2, u ** 4, u ** 6
        v2, v4, v6 = v ** 2, v ** 4, v ** 6
        A111 = 1.0 / 8.0 / u6
        A11, A22 = 0.25 * np.linalg.solve([[u4, v4], [u6, v6]], np.array([1.0, 3.0]) - 8 * (n - 2) * np.array([u4, u6]) * A111)
        A1, A2 = -2 * (n - 1) * np.array([A11, A22]) + 0.5 * np.linalg.solve(
            [[u2, v2], [u4, v4]], np.array([1.0, 3.0]) - 4 * (n - 1) * (n - 2) * np.array([u2, u4]) * A111)
        A0 = 1.0 - 2 * n * (A1 + A2) - 2 * n * (n - 1) * (A11 + A22) - 8 * n * (n - 1) * (n - 2) / 6 * A111
        sets = [(_fs_set(n, ()), A0), (_fs_set(n, (u,)), A1), (_fs_set(n, (v,)), A2), (_fs_set(n, (u, u)), A11),
                (_fs_set(n, (v, v)), A22), (_fs_set(n, (u, u, u)), A111)]
    elif p == 9:
        u, v = _uv(-6.0, 60.0, -90.0)
        u2, u4, u6, u8 = u ** 2, u ** 4, u ** 6, u ** 8
        v2, v4, v6, v8 = v ** 2, v ** 4, v ** 6, v ** 8
        A1111 = 1.0 / 16.0 / u8
        M68 = [[u6, v6], [u8, v8]]
        A111, A222 = 1 / 8 * np.linalg.solve(M68, np.array([1.0, 3.0]) - 16 * (n - 3) * A1111 * np.array([u6, u8]))
        A12 = (15.0 - 9.0) / (4 * u2 * v2 * (u2 - v2) ** 2)
        A11, A22 = -2 * (n - 2) * np.array([A111, A222]) + 1 / 4 * np.linalg.solve(
            M68, np.array([3.0, 15.0]) - 4 * np.array([u4 * v2 + u2 * v4, u6 * v2 + u2 * v6]) * A12
            - 16 * _binom(n - 2, 2) * np.array([u6, u8]) * A1111)
        A1, A2 = (-2 * (n - 1) * np.array([A11 + A12, A22 + A12]) - 4 * _binom(n - 1, 2) * np.array([A111, A222])
                  + 0.5 * np.linalg.solve([[u2, v2], [u4, v4]],
                                          np.array([1.0, 3.0]) - 16 * _binom(n - 1, 3) * np.array([u2, u4]) * A1111))
        s3 = +1.0 if quirks else -1.0       # reference: "- -8*ndownk(n,3)*(A111+A222)"
        A0 = (1.0 - 2 * n * (A1 + A2) - 4 * _binom(n, 2) * (A11 + 2 * A12 + A22)
              + s3 * 8 * _binom(n, 3) * (A111 + A222) - 16 * _binom(n, 4) * A1111)
        sets = [(_fs_set(n, ()), A0), (_fs_set(n, (u,)), A1), (_fs_set(n, (v,)), A2), (_fs_set(n, (u, u)), A11),
                (_fs_set(n, (u, v)), A12), (_fs_set(n, (v, v)), A22), (_fs_set(n, (u, u, u)), A111),
                (_fs_set(n, (v, v, v)), A222), (_fs_set(n, (u, u, u, u)), A1111)]
    else:
        raise ValueError('Not implemented')
    SX = np.hstack([s for s, _ in sets])
    W = np.concatenate([np.full(s.shape[1], a) for s, a in sets])
    return W, SX


def gauher(N):
    """[x,w] = gauher(N) (matlab/gauher.m): Gauss-Hermite rule for the standard normal weight.
    N == 20 returns the reference's tabulated (rounded) values."""
    if N == 20:
        x = np.array([-7.619048541679757, -6.510590157013656, -5.578738805893203, -4.734581334046057,
                      -3.943967350657318, -3.18901481655339, -2.458663611172367, -1.745247320814127,
                      -1.042945348802751, -0.346964157081356, 0.346964157081356, 1.042945348802751,
                      1.745247320814127, 2.458663611172367, 3.18901481655339, 3.943967350657316,
                      4.734581334046057, 5.578738805893202, 6.510590157013653, 7.619048541679757])
        w = np.array([0.000000000000126, 0.000000000248206, 0.000000061274903, 0.00000440212109,
                      0.000128826279962, 0.00183010313108, 0.013997837447101, 0.061506372063977,
                      0.161739333984, 0.260793063449555])
        return x, np.concatenate([w, w[::-1]])
    x, w = np.polynomial.hermite_e.hermegauss(N)
    return x, w / math.sqrt(2.0 * math.pi)


def mvhermgauss_unit(dim, N):
    """Unit tensor Gauss-Hermite grid: (wn (N^dim,), xn_unscaled (dim, N^dim))."""
    t, w = gauher(N)
    idx = np.indices((N,) * dim).reshape(dim, -1)
    return np.prod(w[idx], axis=0), t[idx]


def sigma_points(p, dim, quirks=True):
    """Rule selection of likModulator*Power.m:33-41."""
    return utp_ws(p, dim, quirks) if p in (3, 5, 7, 9) else mvhermgauss_unit(dim, p)
