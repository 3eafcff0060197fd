"""Host-side model construction (runs once per call, before the C-ABI boundary).

Mirrors the reference interfaces
  ss_modulators(w,k1,k2)            matlab/ss_modulators.m:1
  ss_modulators_nmf(ws,wm,k1,k2)    matlab/ss_modulators_nmf.m:1
  lti_disc(F,L,Qc,dt)               matlab/unifying_prob_tf/lti_disc.m:1
  sigmoid / inv_sigmoid             matlab/sigmoid.m, inv_sigmoid.m
  cf_<kernel>_to_ss                 matlab/unifying_prob_tf/cf_{exp,matern32,matern52,matern72}_to_ss.m
but is written block-wise: the SDE is block-diagonal (one block per sub-band / modulator,
ss_modulators_nmf.m:128-132), so the discretisation works on the <=4x4 (8x8 Van-Loan) blocks
instead of the S x S matrices.  MATLAB built-ins map to SciPy (expm, matrix_balance = LAPACK
?gebal as MATLAB's `balance`, cholesky).
"""
import math

import numpy as np
import scipy.linalg as sla

_SQ = {'matern32': math.sqrt(3.0), 'matern52': math.sqrt(5.0), 'matern72': math.sqrt(7.0)}
KERNEL_ORDER = {'exp': 1, 'matern32': 2, 'matern52': 3, 'matern72': 4}


def kernel_block(kernel, sigma2, ell):
    """(F, L, Qc, Pinf) of one stationary kernel in companion form; H = e_1'."""
    if kernel == 'exp':
        return (np.array([[-1.0 / ell]]), np.array([[1.0]]), 2.0 * sigma2 / ell, np.array([[sigma2]]))
    lam = _SQ[kernel] / ell
    if kernel == 'matern32':
        F = np.array([[0.0, 1.0], [-lam ** 2, -2.0 * lam]])
        Qc = 12.0 * math.sqrt(3.0) / ell ** 3 * sigma2
        Pinf = np.diag([sigma2, 3.0 * sigma2 / ell ** 2])
    elif kernel == 'matern52':
        F = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-lam ** 3, -3.0 * lam ** 2, -3.0 * lam]])
        Qc = sigma2 * 400.0 * math.sqrt(5.0) / 3.0 / ell ** 5
        kap = 5.0 / 3.0 * sigma2 / ell ** 2
        Pinf = np.array([[sigma2, 0.0, -kap], [0.0, kap, 0.0], [-kap, 0.0, 25.0 * sigma2 / ell ** 4]])
    elif kernel == 'matern72':
        F = np.array([[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0],
                      [-lam ** 4, -4.0 * lam ** 3, -6.0 * lam ** 2, -4.0 * lam]])
        Qc = sigma2 * 10976.0 * math.sqrt(7.0) / 5.0 / ell ** 7
        kap = 7.0 / 5.0 * sigma2 / ell ** 2
        kap2 = 9.8 * sigma2 / ell ** 4
        Pinf = np.array([[sigma2, 0.0, -kap, 0.0], [0.0, kap, 0.0, -kap2], [-kap, 0.0, kap2, 0.0],
                         [0.0, -kap2, 0.0, 343.0 * sigma2 / ell ** 6]])
    else:
        raise ValueError('unsupported kernel %r (supported: exp, matern32, matern52, matern72)' % kernel)
    L = np.zeros((F.shape[0], 1)); L[-1, 0] = 1.0
    return F, L, Qc, Pinf


def kernel_block_derivs(kernel, sigma2, ell):
    """Derivatives of kernel_block's (F, Pinf) w.r.t. (sigma2, ell) -- the dF / dPinf outputs of cf_<kernel>_to_ss
    (cf_exp_to_ss.m:116-146, cf_matern32_to_ss.m:121-157, cf_matern52_to_ss.m:127-166), obtained through lam = c / ell:
    d/d ell = (-lam / ell) d/d lam.  Returns ((dF_sigma2, dPinf_sigma2), (dF_ell, dPinf_ell))."""
    F, _, _, Pinf = kernel_block(kernel, sigma2, ell)
    n = F.shape[0]
    if kernel == 'exp':
        return (np.zeros((1, 1)), np.ones((1, 1))), (np.array([[1.0 / ell ** 2]]), np.zeros((1, 1)))
    lam = _SQ[kernel] / ell
    binom = {'matern32': [1.0, 2.0], 'matern52': [1.0, 3.0, 3.0]}.get(kernel)
    if binom is None:
        raise ValueError('kernel derivatives exist for exp, matern32, matern52 (the kernels of the drivers)')
    # last row of the companion matrix: -C(n,i) lam^(n-i), i = 0..n-1
    dF_lam = np.zeros((n, n))
    for i in range(n):
        dF_lam[n - 1, i] = -binom[i] * (n - i) * lam ** (n - i - 1)
    dF_ell = dF_lam * (-lam / ell)
    # Pinf entries are sigma2 * c_ik * lam^(i+k) (i + k even, 0-based): linear in sigma2, power i+k of lam
    pw = np.add.outer(np.arange(n), np.arange(n)).astype(float)
    dP_ell = Pinf * pw * (-1.0 / ell)
    return (np.zeros((n, n)), Pinf / sigma2), (dF_ell, dP_ell)


class BlockSS:
    """Block-diagonal continuous-time model: lists of per-block (F, LQL', Pinf) and the H pattern."""

    def __init__(self, F, LQL, Pinf, D, N):
        self.F, self.LQL, self.Pinf, self.D, self.N = F, LQL, Pinf, D, N
        self.sizes = [f.shape[0] for f in F]
        self.offsets = np.concatenate([[0], np.cumsum(self.sizes)]).astype(np.int32)
        self.S = int(self.offsets[-1]); self.M = len(F)
        self.h_val = np.ones(self.M)

    def dense(self):
        """(F, L*Qc*L', H, Pinf) as dense matrices."""
        H = np.zeros((self.M, self.S))
        H[np.arange(self.M), self.offsets[:-1]] = self.h_val
        return sla.block_diag(*self.F), sla.block_diag(*self.LQL), H, sla.block_diag(*self.Pinf)


def _subband_block(kernel, sigma2, ell, omega):
    F1, L1, Qc, P1 = kernel_block(kernel, sigma2, ell)
    I2 = np.eye(2)
    F = np.kron(F1, I2) + np.kron(np.eye(F1.shape[0]), np.array([[0.0, -omega], [omega, 0.0]]))
    LQL = np.kron(L1 @ L1.T * Qc, I2)
    return F, LQL, np.kron(P1, I2)


def ss_blocks_nmf(w_subband, w_modulator, kernel1, kernel2):
    w1 = np.asarray(w_subband, float).ravel(); w2 = np.asarray(w_modulator, float).ravel()
    D = w1.size // 3; N = w2.size // 2
    Fs, Qs, Ps = [], [], []
    for d in range(D):
        F, LQL, P = _subband_block(kernel1, w1[d], w1[D + d], w1[2 * D + d])
        Fs.append(F); Qs.append(LQL); Ps.append(P)
    for n in range(N):
        F, L, Qc, P = kernel_block(kernel2, w2[n], w2[N + n])
        Fs.append(F); Qs.append(L @ L.T * Qc); Ps.append(P)
    return BlockSS(Fs, Qs, Ps, D, N)


def ss_modulators_nmf(w_subband, w_modulator, kernel1, kernel2):
    """[F,L,Qc,H,Pinf] = ss_modulators_nmf(w_subband,w_modulator,kernel1,kernel2) (dense, reference layout).
    L is returned as I and Qc as L*Qc*L' (the only combination the drivers use)."""
    blk = ss_blocks_nmf(w_subband, w_modulator, kernel1, kernel2)
    F, LQL, H, Pinf = blk.dense()
    return F, np.eye(blk.S), LQL, H, Pinf


def ss_modulators(w, kernel1, kernel2):
    w = np.asarray(w, float).ravel(); D = w.size // 5
    return ss_modulators_nmf(w[:3 * D], w[3 * D:], kernel1, kernel2)


def lti_disc_block(F, LQL, dt=1.0):
    """lti_disc.m:73-82 on one diagonal block (matrix-fraction / Van Loan form)."""
    n = F.shape[0]
    A = sla.expm(F * dt)
    Phi = np.zeros((2 * n, 2 * n))
    Phi[:n, :n] = F; Phi[:n, n:] = LQL; Phi[n:, n:] = -F.T
    E = sla.expm(Phi * dt)
    Q = E[:n, n:] @ np.linalg.inv(E[n:, n:])
    return A, Q


def lti_disc(F, L, Qc, dt=1.0):
    """Dense-interface lti_disc for callers that hold dense matrices."""
    return lti_disc_block(np.asarray(F, float), np.asarray(L, float) @ np.atleast_2d(Qc) @ np.asarray(L, float).T, dt)


def balance_blocks(blk):
    """[T,F]=balance(F); H=H*T; LL=T\\chol(Pinf,'lower'); Pinf=LL*LL' (gf_ep_modulator.m:75-81), block-wise.
    The process-noise term L*Qc*L' transforms as T\\(LQL')/T'."""
    for n in range(blk.M):
        Fb, T = sla.matrix_balance(blk.F[n], permute=True, scale=True, separate=False)
        t = np.diag(T)
        if not np.allclose(T, np.diag(t)):
            raise ValueError('balance() permuted block %d; the single-nonzero-per-row structure of H is lost' % n)
        blk.F[n] = Fb
        if not hasattr(blk, 'tbal'):
            blk.tbal = [np.ones(k) for k in blk.sizes]           # the diagonal of T per block (what the derivative stacks need)
        blk.tbal[n] = blk.tbal[n] * t
        blk.LQL[n] = blk.LQL[n] / np.outer(t, t)
        LL = np.linalg.cholesky(blk.Pinf[n]) / t[:, None]
        blk.Pinf[n] = LL @ LL.T
        blk.h_val[n] = blk.h_val[n] * t[0]
    return blk


def discretise(blk, symmetrize_Q=False, stationary_Q=False):
    """Per-block (A, Q); returns dense column-major S x S arrays for the C ABI.
    stationary_Q: A = expm(F), Q = Pinf - A*Pinf*A' (gf_giekf_modulator_nmf_constraints.m:377-378) instead of lti_disc."""
    S = blk.S
    A = np.zeros((S, S), order='F'); Q = np.zeros((S, S), order='F'); P = np.zeros((S, S), order='F')
    for n in range(blk.M):
        a, q = lti_disc_block(blk.F[n], blk.LQL[n])
        if stationary_Q:
            a = sla.expm(blk.F[n]); q = blk.Pinf[n] - a @ blk.Pinf[n] @ a.T
        if symmetrize_Q:                       # ihgp_ep_modulator_nmf.m:97
            q = (q + q.T) / 2
        o, e = blk.offsets[n], blk.offsets[n + 1]
        A[o:e, o:e] = a; Q[o:e, o:e] = q; P[o:e, o:e] = blk.Pinf[n]
    return A, Q, P


def sigmoid(x, sig_range=(0.0, 20.0), c=0.0, a=1.0):
    lo, up = sig_range[0], sig_range[-1]
    return (up - lo) / (1.0 + np.exp(-a * (np.asarray(x, float) - c))) + lo


def inv_sigmoid(y, sig_range=(0.0, 20.0), c=0.0, a=1.0):
    lo, up = sig_range[0], sig_range[-1]
    y = np.asarray(y, float)
    if np.any(y <= lo) or np.any(y >= up):
        raise ValueError('Error with inverse sigmoid transformation: parameter outside of user specified range')
    return c - np.log((up - y) / (y - lo)) / a
