"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's model construction (host side, once per call).
PARITY UNPINNED (no reference fixtures, no MATLAB); self-pinned by
tests/test_oracle_ss.py (Q == Pinf - A Pinf A', kernel autocovariances).

Follows (file:line under /root/reference/matlab):
  unifying_prob_tf/cf_exp_to_ss.m:93-109, cf_matern32_to_ss.m:93-115,
  cf_matern52_to_ss.m:93-121, cf_matern72_to_ss.m:93-128
  ss_modulators.m:1-134, ss_modulators_nmf.m:1-137   (F,L,Qc,H,Pinf only; the
      derivative stacks dF,dQc,dPinf feed only the EKF nlml gradient, SURVEY f-4)
  unifying_prob_tf/lti_disc.m:60-82
  balance step  gf_ep_modulator.m:75-81 (identical in the other drivers)
  sigmoid.m:17-19, inv_sigmoid.m:17-23
  parameter unpacking gf_ep_modulator_nmf.m:72-75, gf_ep_modulator_nmf_constraints.m:75-110
MATLAB built-ins replaced by SciPy (closed source, unpinned): expm -> scipy.linalg.expm,
balance -> scipy.linalg.matrix_balance (same LAPACK ?gebal), chol -> numpy.linalg.cholesky.
"""
import math
import numpy as np
import scipy.linalg as sla


def cf_exp_to_ss(magnSigma2=1.0, lengthScale=1.0):
    F = np.array([[-1.0 / lengthScale]])
    L = np.array([[1.0]])
    Qc = np.array([[2.0 * magnSigma2 / lengthScale]])
    H = np.array([[1.0]])
    Pinf = np.array([[magnSigma2]])
    return F, L, Qc, H, Pinf


def cf_matern32_to_ss(magnSigma2=1.0, lengthScale=1.0):
    lam = math.sqrt(3.0) / lengthScale
    F = np.array([[0.0, 1.0], [-lam ** 2, -2 * lam]])
    L = np.array([[0.0], [1.0]])
    Qc = np.array([[12.0 * math.sqrt(3.0) / lengthScale ** 3 * magnSigma2]])
    H = np.array([[1.0, 0.0]])
    Pinf = np.array([[magnSigma2, 0.0], [0.0, 3.0 * magnSigma2 / lengthScale ** 2]])
    return F, L, Qc, H, Pinf


def cf_matern52_to_ss(magnSigma2=1.0, lengthScale=1.0):
    lam = math.sqrt(5.0) / lengthScale
    F = np.array([[0.0, 1.0, 0.0], [0.0, 0.0, 1.0], [-lam ** 3, -3 * lam ** 2, -3 * lam]])
    L = np.array([[0.0], [0.0], [1.0]])
    Qc = np.array([[magnSigma2 * 400.0 * math.sqrt(5.0) / 3.0 / lengthScale ** 5]])
    H = np.array([[1.0, 0.0, 0.0]])
    kappa = 5.0 / 3.0 * magnSigma2 / lengthScale ** 2
    Pinf = np.array([[magnSigma2, 0.0, -kappa], [0.0, kappa, 0.0],
                     [-kappa, 0.0, 25.0 * magnSigma2 / lengthScale ** 4]])
    return F, L, Qc, H, Pinf


def cf_matern72_to_ss(magnSigma2=1.0, lengthScale=1.0):
    lam = math.sqrt(7.0) / lengthScale
    F = np.array([[0.0, 1.0, 0.0, 0.0], [0.0, 0.0, 1.0, 0.0], [0.0, 0.0, 0.0, 1.0],
                  [-lam ** 4, -4 * lam ** 3, -6 * lam ** 2, -4 * lam]])
    L = np.array([[0.0], [0.0], [0.0], [1.0]])
    Qc = np.array([[magnSigma2 * 10976.0 * math.sqrt(7.0) / 5.0 / lengthScale ** 7]])
    H = np.array([[1.0, 0.0, 0.0, 0.0]])
    kappa = 7.0 / 5.0 * magnSigma2 / lengthScale ** 2
    kappa2 = 9.8 * magnSigma2 / lengthScale ** 4
    Pinf = np.array([[magnSigma2, 0.0, -kappa, 0.0], [0.0, kappa, 0.0, -kappa2],
                     [-kappa, 0.0, kappa2, 0.0], [0.0, -kappa2, 0.0, 343.0 * magnSigma2 / lengthScale ** 6]])
    return F, L, Qc, H, Pinf


_CF = {'exp': cf_exp_to_ss, 'matern32': cf_matern32_to_ss,
       'matern52': cf_matern52_to_ss, 'matern72': cf_matern72_to_ss}


def _blkdiag(mats):
    mats = [np.atleast_2d(m) for m in mats]
    return sla.block_diag(*mats) if mats else np.zeros((0, 0))


def _subband_part(sig1, len1, omega, kernel1):
    """ss_modulators_nmf.m:24-78 (identical to ss_modulators.m:22-77)."""
    cf1 = _CF[kernel1]
    D = len(sig1)
    tau1 = cf1(1.0, 1.0)[0].shape[0]
    F1s, L1s, Qc1s, H1s, P1s = [], [], [], [], []
    for d in range(D):
        F1d, L1d, Qc1d, H1d, Pinf1d = cf1(sig1[d], len1[d])
        F1s.append(F1d); L1s.append(L1d); Qc1s.append(Qc1d); H1s.append(H1d); P1s.append(Pinf1d)
    F1 = _blkdiag(F1s); L1 = np.vstack(L1s); Qc1 = _blkdiag(Qc1s); H1 = _blkdiag(H1s); Pinf1 = _blkdiag(P1s)
    I2 = np.eye(2)
    F_cos_kron, L_sm, Qc_sm = [], [], []
    for d in range(D):
        F_cos_d = np.array([[0.0, -omega[d]], [omega[d], 0.0]])
        F_cos_kron.append(np.kron(np.eye(tau1), F_cos_d))
        L_sm.append(np.kron(L1[tau1 * d:tau1 * (d + 1)], I2))
        Qc_sm.append(np.kron(Qc1[d:d + 1, d:d + 1], I2))
    F_sm = np.kron(F1, I2) + _blkdiag(F_cos_kron)
    H_sm = np.kron(H1, np.array([[1.0, 0.0]]))
    Pinf_sm = np.kron(Pinf1, I2)
    return F_sm, _blkdiag(L_sm), _blkdiag(Qc_sm), H_sm, Pinf_sm


def _modulator_part(sig2, len2, kernel2):
    """ss_modulators_nmf.m:96-119."""
    cf2 = _CF[kernel2]
    Fs, Ls, Qs, Hs, Ps = [], [], [], [], []
    for d in range(len(sig2)):
        F2d, L2d, Qc2d, H2d, Pinf2d = cf2(sig2[d], len2[d])
        Fs.append(F2d); Ls.append(L2d); Qs.append(Qc2d); Hs.append(H2d); Ps.append(Pinf2d)
    return _blkdiag(Fs), _blkdiag(Ls), _blkdiag(Qs), _blkdiag(Hs), _blkdiag(Ps)


def ss_modulators_nmf(w_subband, w_modulator, kernel1, kernel2):
    """ss_modulators_nmf.m:1-137 -> F,L,Qc,H,Pinf."""
    w_subband = np.asarray(w_subband, float).ravel(); w_modulator = np.asarray(w_modulator, float).ravel()
    D = len(w_subband) // 3; N = len(w_modulator) // 2
    sig1, len1, omega = w_subband[:D], w_subband[D:2 * D], w_subband[2 * D:3 * D]
    sig2, len2 = w_modulator[:N], w_modulator[N:2 * N]
    a = _subband_part(sig1, len1, omega, kernel1)
    b = _modulator_part(sig2, len2, kernel2)
    return tuple(_blkdiag([x, y]) for x, y in zip(a, b))


def ss_modulators(w, kernel1, kernel2):
    """ss_modulators.m:1-134 -> F,L,Qc,H,Pinf (one modulator per sub-band)."""
    w = np.asarray(w, float).ravel()
    D = len(w) // 5
    return ss_modulators_nmf(w[:3 * D], w[3 * D:5 * D], kernel1, kernel2)


def lti_disc(F, L, Qc, dt=1.0):
    """lti_disc.m:60-82."""
    n = F.shape[0]
    A = sla.expm(F * dt)
    Phi = np.block([[F, L @ Qc @ L.T], [np.zeros((n, n)), -F.T]])
    AB = sla.expm(Phi * dt) @ np.vstack([np.zeros((n, n)), np.eye(n)])
    # MATLAB  AB1/AB2  ==  AB1 * inv(AB2)
    Q = np.linalg.solve(AB[n:, :].T, AB[:n, :].T).T
    return A, Q


def balance_ss(F, L, H, Pinf):
    """gf_ep_modulator.m:75-81: [T,F]=balance(F); L=T\\L; H=H*T; LL=T\\chol(Pinf,'lower'); Pinf=LL*LL'."""
    Fb, T = sla.matrix_balance(F, permute=True, scale=True, separate=False)
    Lb = np.linalg.solve(T, L)
    Hb = H @ T
    LL = np.linalg.solve(T, np.linalg.cholesky(Pinf))
    return Fb, Lb, Hb, LL @ LL.T, T


def sigmoid(x, sig_range=(0.0, 20.0), c=0.0, a=1.0):
    lo, up = sig_range[0], sig_range[-1]
    return (up - lo) / (1.0 + np.exp(-a * (np.asarray(x, float) - c))) + lo


def inv_sigmoid(y, sig_range=(0.0, 20.0), c=0.0, a=1.0):
    lo, up = sig_range[0], sig_range[-1]
    y = np.asarray(y, float)
    return c - np.log((up - y) / (y - lo)) / a


def unpack_log(w, num_lik_params, D, N):
    """gf_ep_modulator_nmf.m:72-75."""
    w = np.asarray(w, float).ravel()
    n0 = num_lik_params
    lik_param = w[:n0]
    param1 = np.exp(w[n0:n0 + 3 * D])
    param2 = np.exp(w[n0 + 3 * D:n0 + 3 * D + 2 * N])
    Wnmf = np.exp(w[n0 + 3 * D + 2 * N:]).reshape((D, N), order='F')
    return lik_param, param1, param2, Wnmf


def unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N):
    """gf_ep_modulator_nmf_constraints.m:75-110."""
    w = np.asarray(w, float).ravel(); w_fixed = np.asarray(w_fixed, float).ravel()
    constraints = np.asarray(constraints, float)
    wi = 0; wfi = 0
    if tune_hypers[0]:
        lik_param = w[:num_lik_params]; wi += num_lik_params
    else:
        lik_param = w_fixed[:num_lik_params]; wfi += num_lik_params
    param1 = []; param2 = []
    for i in range(2, 7):                     # MATLAB i = 2..6
        cnt = D if i <= 4 else N
        if tune_hypers[i - 1]:
            val = sigmoid(w[wi:wi + cnt], constraints[i - 2]); wi += cnt
        else:
            val = sigmoid(w_fixed[wfi:wfi + cnt], constraints[i - 2]); wfi += cnt
        (param1 if i <= 4 else param2).append(val)
    if tune_hypers[6]:
        Wnmf = sigmoid(w[wi:], constraints[5]).reshape((D, N), order='F')
    else:
        Wnmf = sigmoid(w_fixed[wfi:], constraints[5]).reshape((D, N), order='F')
    return lik_param, np.concatenate(param1), np.concatenate(param2), Wnmf


def block_starts(H):
    """ihgp_ep_modulator_nmf.m:104  ilist = [find(sum(H,1)) size(H,2)+1]  (0-based here)."""
    cols = np.nonzero(np.sum(H, axis=0))[0]
    return np.concatenate([cols, [H.shape[1]]]).astype(int)


# ---------------------------------------------------------------------------------------------
# Derivative stacks (dF, dQc, dPinf) -- used only by the EKF nlml gradient (gf_giekf_modulator_nmf_constraints.m:332-480)
def cf_derivs(kernel, magnSigma2, lengthScale):
    """[dF, dPinf] of cf_<kernel>_to_ss w.r.t. (magnSigma2, lengthScale): cf_exp_to_ss.m:116-146, cf_matern32_to_ss.m:121-157,
    cf_matern52_to_ss.m:127-166.  (dQc is not read by the gradient recursion, which uses Q = Pinf - A Pinf A'.)"""
    s2, ell = float(magnSigma2), float(lengthScale)
    if kernel == 'exp':
        dF = [np.array([[0.0]]), np.array([[1.0 / ell ** 2]])]
        dP = [np.array([[1.0]]), np.array([[0.0]])]
    elif kernel == 'matern32':
        dF = [np.zeros((2, 2)), np.array([[0.0, 0.0], [6.0 / ell ** 3, 2.0 * math.sqrt(3.0) / ell ** 2]])]
        dP = [np.array([[1.0, 0.0], [0.0, 3.0 / ell ** 2]]), np.array([[0.0, 0.0], [0.0, -6.0 * s2 / ell ** 3]])]
    elif kernel == 'matern52':
        dF = [np.zeros((3, 3)), np.array([[0.0, 0.0, 0.0], [0.0, 0.0, 0.0],
                                          [15.0 * math.sqrt(5.0) / ell ** 4, 30.0 / ell ** 3, 3.0 * math.sqrt(5.0) / ell ** 2]])]
        Pinf = cf_matern52_to_ss(s2, ell)[4]
        kappa = 5.0 / 3.0 * s2 / ell ** 2; kappa2 = -2.0 * kappa / ell
        dP = [Pinf / s2, np.array([[0.0, 0.0, -kappa2], [0.0, kappa2, 0.0], [-kappa2, 0.0, -100.0 * s2 / ell ** 5]])]
    else:
        raise ValueError('derivatives restated for exp / matern32 / matern52 (the kernels the drivers use)')
    return dF, dP


def ss_modulators_nmf_derivs(w_subband, w_modulator, kernel1, kernel2):
    """ss_modulators_nmf.m:24-132, derivative outputs: dF, dPinf of shape (3D+2N, S, S), parameter order
    [sig1 (D), len1 (D), omega (D), sig2 (N), len2 (N)] -- the order of `cat(3, dF_sm_1, dF_sm_2, dF_cos_kron)` (:88) and
    `cat(3, dF_sm, dF2_blk)` (:130).  UNBALANCED, as the reference uses them (the balancing lines
    gf_giekf_modulator_nmf_constraints.m:117-119 are commented out)."""
    w_subband = np.asarray(w_subband, float).ravel(); w_modulator = np.asarray(w_modulator, float).ravel()
    D = len(w_subband) // 3; N = len(w_modulator) // 2
    sig1, len1 = w_subband[:D], w_subband[D:2 * D]
    sig2, len2 = w_modulator[:N], w_modulator[N:2 * N]
    tau1 = _CF[kernel1](1.0, 1.0)[0].shape[0]; tau3 = _CF[kernel2](1.0, 1.0)[0].shape[0]
    zt = 2 * tau1; S = zt * D + tau3 * N
    dF = np.zeros((3 * D + 2 * N, S, S)); dP = np.zeros_like(dF)
    I2 = np.eye(2)
    for d in range(D):
        dFd, dPd = cf_derivs(kernel1, sig1[d], len1[d])
        sl = slice(zt * d, zt * (d + 1))
        for q in range(2):                                   # :83-86  kron(dF1, eye(tau2))
            dF[q * D + d, sl, sl] = np.kron(dFd[q], I2)
            dP[q * D + d, sl, sl] = np.kron(dPd[q], I2)
        dF[2 * D + d, sl, sl] = np.kron(np.eye(tau1), np.array([[0.0, -1.0], [1.0, 0.0]]))     # :56, 70-72  d/d omega
    for n in range(N):
        dFn, dPn = cf_derivs(kernel2, sig2[n], len2[n])
        sl = slice(zt * D + tau3 * n, zt * D + tau3 * (n + 1))
        for q in range(2):                                   # :108-124
            dF[3 * D + q * N + n, sl, sl] = dFn[q]
            dP[3 * D + q * N + n, sl, sl] = dPn[q]
    return dF, dP
