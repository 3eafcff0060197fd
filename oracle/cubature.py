"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's sigma-point / quadrature table builders.
PARITY UNPINNED: the reference ships no golden vectors for these and MATLAB/Octave
are not available, so this file is pinned only by its own mathematical
self-checks (tests/test_oracle_selfpins.py: test_cubature_polynomial_exactness, test_ut9_sign_typo_only_touches_centre_weight).

Follows (file:line under /root/reference/matlab):
  symmetric-cubature-rules/utp_ws.m:3-14
  symmetric-cubature-rules/sym_set.m:1-41
  symmetric-cubature-rules/ut3_ws.m:7-27
  symmetric-cubature-rules/ut5_ws.m:7-23
  symmetric-cubature-rules/ut7_ws.m:7-49
  symmetric-cubature-rules/ut9_ws.m:21-100   (incl. the sign typo at :78-79, SURVEY C-1)
  gauher.m:33-54, mvhermgauss.m:11-23
"""
import math
import numpy as np


def nupk(n, k):
    # ut9_ws.m:104-106  prod((n-k+1):n); empty range -> 1; may include 0/negatives (C-2)
    out = 1.0
    for v in range(n - k + 1, n + 1):
        out *= v
    return out


def ndownk(n, k):
    # ut9_ws.m:108-110
    return nupk(n, k) / math.factorial(k)


def sym_set(n, gen):
    """sym_set.m:1-41 -- fully symmetric point set; returns (n, npts) array.

    `nonzero` is always 0 in the reference (nargin is never 3), so an empty
    generator returns the single origin column.
    """
    gen = list(gen)
    if len(gen) == 0:
        return np.zeros((n, 1))
    cols = []
    for i in range(n):                      # MATLAB i = 1..n  -> i0 = i
        u = np.zeros(n)
        u[i] = gen[0]
        if len(gen) > 1:
            if abs(gen[0] - gen[1]) < np.finfo(float).eps:
                V = sym_set(n - (i + 1), gen[1:])
                for j in range(V.shape[1]):
                    u[i + 1:] = V[:, j]
                    cols.append(u.copy())
                    cols.append(-u.copy())
            else:
                V = sym_set(n - 1, gen[1:])
                idx = [q for q in range(n) if q != i]
                for j in range(V.shape[1]):
                    u[idx] = V[:, j]
                    cols.append(u.copy())
                    cols.append(-u.copy())
        else:
            cols.append(u.copy())
            cols.append(-u.copy())
    if len(cols) == 0:
        return np.zeros((n, 0))
    return np.stack(cols, axis=1)


def ut3_ws(n):
    # ut3_ws.m:7-27 ; kappa forced to 0 (:9-10)
    kappa = 0.0
    W = np.zeros(2 * n + 1)
    W[0] = kappa / (n + kappa)
    W[1:] = 1.0 / (2.0 * (n + kappa))
    SX = np.hstack([np.zeros((n, 1)), np.eye(n), -np.eye(n)])
    SX = math.sqrt(n + kappa) * SX
    return W, SX


def ut5_ws(n):
    # ut5_ws.m:7-23
    I0, I2, I4, I22 = 1.0, 1.0, 3.0, 1.0
    u = math.sqrt(I4 / I2)
    A0 = I0 - n * (I2 / I4) ** 2 * (I4 - 0.5 * (n - 1) * I22)
    A1 = 0.5 * (I2 / I4) ** 2 * (I4 - (n - 1) * I22)
    A11 = 0.25 * (I2 / I4) ** 2 * I22
    U0 = sym_set(n, [])
    U1 = sym_set(n, [u])
    U2 = sym_set(n, [u, u])
    SX = np.hstack([U0, U1, U2])
    W = np.concatenate([A0 * np.ones(U0.shape[1]), A1 * np.ones(U1.shape[1]),
                        A11 * np.ones(U2.shape[1])])
    return W, SX


def _pos_roots(coeffs):
    """MATLAB `tmp = roots(c); tmp = tmp(tmp>0); u=tmp(1); v=tmp(2)`.

    The biquadratic's roots are +-sqrt(r1), +-sqrt(r2); MATLAB's `roots` (eig of
    the companion matrix) returns the larger-magnitude pair first, as NumPy's
    does; we fix u = larger, v = smaller (SURVEY C-1) -- the one ordering
    assumption that cannot be confirmed without MATLAB.
    """
    r = np.roots(coeffs)
    r = np.real(r[np.abs(np.imag(r)) < 1e-12])
    r = np.sort(r[r > 0])[::-1]
    return float(r[0]), float(r[1])


def ut7_ws(n):
    # ut7_ws.m:7-49
    I222, I22, I24, I2, I6, I4, I0 = 1.0, 1.0, 3.0, 1.0, 15.0, 3.0, 1.0
    u, v = _pos_roots([I2 ** 2 - I0 * I4, 0, -(I2 * I4 - I0 * I6), 0, (I4 ** 2 - I2 * I6)])
    u2 = u * u; u4 = u2 * u2; u6 = u4 * u2
    v2 = v * v; v4 = v2 * v2; v6 = v4 * v2
    A111 = I222 / 8 / u6
    tmp = 0.25 * np.linalg.solve(np.array([[u4, v4], [u6, v6]]),
                                 np.array([I22, I24]) - 8 * (n - 2) * np.array([u4, u6]) * A111)
    A11, A22 = tmp
    tmp = -2 * (n - 1) * np.array([A11, A22]) + 0.5 * np.linalg.solve(
        np.array([[u2, v2], [u4, v4]]),
        np.array([I2, I4]) - 8 * (n - 1) * (n - 2) / 2 * np.array([u2, u4]) * A111)
    A1, A2 = tmp
    A0 = I0 - 2 * n * (A1 + A2) - 4 * n * (n - 1) / 2 * (A11 + A22) - 8 * n * (n - 1) * (n - 2) / 6 * A111
    U0 = sym_set(n, []); U1 = sym_set(n, [u]); V1 = sym_set(n, [v])
    U2 = sym_set(n, [u, u]); V2 = sym_set(n, [v, v]); U3 = sym_set(n, [u, u, u])
    SX = np.hstack([U0, U1, V1, U2, V2, U3])
    W = np.concatenate([A0 * np.ones(U0.shape[1]), A1 * np.ones(U1.shape[1]), A2 * np.ones(V1.shape[1]),
                        A11 * np.ones(U2.shape[1]), A22 * np.ones(V2.shape[1]), A111 * np.ones(U3.shape[1])])
    return W, SX


def ut9_ws(n, quirks=True):
    # ut9_ws.m:21-100
    I2222 = 1.0; I224 = 3.0; I222 = 1.0; I44 = 9.0; I26 = 15.0; I24 = 3.0; I22 = 1.0
    I8 = 105.0; I6 = 15.0; I4 = 3.0; I2 = 1.0; I0 = 1.0
    u, v = _pos_roots([I4 ** 2 - I2 * I6, 0, -(I4 * I6 - I2 * I8), 0, (I6 ** 2 - I4 * I8)])
    u2 = u * u; u4 = u2 * u2; u6 = u4 * u2; u8 = u4 * u4
    v2 = v * v; v4 = v2 * v2; v6 = v4 * v2; v8 = v4 * v4
    A1111 = I2222 / 16 / u8
    M68 = np.array([[u6, v6], [u8, v8]])
    tmp = 1 / 8 * np.linalg.solve(M68, np.array([I222, I224]) - 16 * (n - 3) * A1111 * np.array([u6, u8]))
    A111, A222 = tmp
    A12 = (I26 - I44) / (4 * u2 * v2 * (u2 - v2) ** 2)
    tmp = -2 * (n - 2) * np.array([A111, A222]) + 1 / 4 * np.linalg.solve(
        M68, np.array([I24, I26]) - 4 * np.array([u4 * v2 + u2 * v4, u6 * v2 + u2 * v6]) * A12
        - 16 * ndownk(n - 2, 2) * np.array([u6, u8]) * A1111)
    A11, A22 = tmp
    tmp = (-2 * (n - 1) * np.array([A11 + A12, A22 + A12]) - 4 * ndownk(n - 1, 2) * np.array([A111, A222])
           + 0.5 * np.linalg.solve(np.array([[u2, v2], [u4, v4]]),
                                   np.array([I2, I4]) - 16 * ndownk(n - 1, 3) * np.array([u2, u4]) * A1111))
    A1, A2 = tmp
    # ut9_ws.m:78-79: "... - -8*ndownk(n,3)*(A111+A222)"  (double minus = PLUS; sign typo C-1)
    sgn = +1.0 if quirks else -1.0
    A0 = (I0 - 2 * n * (A1 + A2) - 4 * ndownk(n, 2) * (A11 + 2 * A12 + A22)
          + sgn * 8 * ndownk(n, 3) * (A111 + A222) - 16 * ndownk(n, 4) * A1111)
    U0 = sym_set(n, []); U1 = sym_set(n, [u]); V1 = sym_set(n, [v])
    U2 = sym_set(n, [u, u]); UV = sym_set(n, [u, v]); V2 = sym_set(n, [v, v])
    U3 = sym_set(n, [u, u, u]); V3 = sym_set(n, [v, v, v]); U4 = sym_set(n, [u, u, u, u])
    SX = np.hstack([U0, U1, V1, U2, UV, V2, U3, V3, U4])
    W = np.concatenate([A0 * np.ones(U0.shape[1]), A1 * np.ones(U1.shape[1]), A2 * np.ones(V1.shape[1]),
                        A11 * np.ones(U2.shape[1]), A12 * np.ones(UV.shape[1]), A22 * np.ones(V2.shape[1]),
                        A111 * np.ones(U3.shape[1]), A222 * np.ones(V3.shape[1]),
                        A1111 * np.ones(U4.shape[1])])
    return W, SX


def utp_ws(p, n, quirks=True):
    # utp_ws.m:3-14 -> (W (npts,), SX (n, npts))
    if p == 3:
        return ut3_ws(n)
    if p == 5:
        return ut5_ws(n)
    if p == 7:
        return ut7_ws(n)
    if p == 9:
        return ut9_ws(n, quirks)
    raise ValueError('Not implemented')


_GH20_X = np.array([-7.619048541679757, -6.510590157013656, -5.578738805893203, -4.734581334046057,
                    -3.943967350657318, -3.18901481655339, -2.458663611172367, -1.745247320814127,
                    -1.042945348802751, -0.346964157081356, 0.346964157081356, 1.042945348802751,
                    1.745247320814127, 2.458663611172367, 3.18901481655339, 3.943967350657316,
                    4.734581334046057, 5.578738805893202, 6.510590157013653, 7.619048541679757])
_GH20_W = np.array([0.000000000000126, 0.000000000248206, 0.000000061274903, 0.00000440212109,
                    0.000128826279962, 0.00183010313108, 0.013997837447101, 0.061506372063977,
                    0.161739333984, 0.260793063449555, 0.260793063449555, 0.161739333984,
                    0.061506372063977, 0.013997837447101, 0.00183010313108, 0.000128826279962,
                    0.00000440212109, 0.000000061274903, 0.000000000248206, 0.000000000000126])


def gauher(N):
    # gauher.m:33-54 (probabilists' weight exp(-x^2/2)/sqrt(2pi))
    if N == 20:
        return _GH20_X.copy(), _GH20_W.copy()
    b = np.sqrt(np.arange(1, N) / 2.0)
    T = np.diag(b, 1) + np.diag(b, -1)
    D, V = np.linalg.eigh(T)
    w = V[0, :] ** 2
    x = math.sqrt(2.0) * D
    return x, w


def mvhermgauss_unit(dim, N):
    """mvhermgauss.m:11-23 with mu=0, s2=1: returns (wn (N^dim,), xn_unscaled (dim, N^dim)).

    ndgrid ordering: first dimension varies fastest.
    """
    t, w = gauher(N)
    grids = np.meshgrid(*([np.arange(N)] * dim), indexing='ij')
    # MATLAB x(:) is column-major flatten of ndgrid output
    idx = [g.flatten(order='F') for g in grids]
    x_loc = np.stack([t[i] for i in idx], axis=0)       # (dim, N^dim)
    w_loc = np.stack([w[i] for i in idx], axis=0)
    return np.prod(w_loc, axis=0), x_loc


def sigma_points(p, dim, quirks=True):
    """What likModulator*Power.m:33-41 selects: symmetric rule for p in {3,5,7,9}, else GH grid."""
    if p in (3, 5, 7, 9):
        return utp_ws(p, dim, quirks)
    return mvhermgauss_unit(dim, p)
