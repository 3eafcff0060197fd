"""TEST INFRASTRUCTURE ONLY -- CPU restatement (NumPy/SciPy) of the stationary filterbank path that precedes the hot
path in every real-audio script of the reference:

  matlab/unifying_prob_tf/get_disc_model.m:1-72          discrete-time spectral-mixture model (sub-band = kernel x cosine)
  matlab/unifying_prob_tf/kernel_ss_kalmanFastFB.m:1-168  infinite-horizon Kalman filter + steady-state RTS smoother

PARITY UNPINNED (no MATLAB here, no fixtures in the reference; `dare` = scipy.linalg.solve_discrete_are).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package.
"""
import numpy as np
import scipy.linalg as sla

from . import ss as oss


def get_disc_model(lamx, varx, omega, D, kernel, se_approx_order=6):
    """get_disc_model.m:1-72 -> A, Q, H, Pinf, K, tau1."""
    lamx = np.asarray(lamx, float).ravel(); varx = np.asarray(varx, float).ravel(); omega = np.asarray(omega, float).ravel()
    if kernel == 'exp':                                   # :9-20 hyper-parameter mapping
        ls = 1.0 / lamx
    elif kernel == 'matern32':
        ls = np.sqrt(3.0) / lamx
    else:
        ls = np.sqrt(5.0) / lamx
    cf = getattr(oss, 'cf_%s_to_ss' % kernel)
    F1, L1, Qc1, H1, P1 = [], [], [], [], []
    for d in range(D):                                    # :26-35
        F1d, L1d, Qc1d, H1d, Pinf1d = cf(varx[d], ls[d])
        F1.append(F1d); L1.append(L1d); Qc1.append(float(Qc1d[0, 0])); H1.append(H1d); P1.append(Pinf1d)
    tau1 = L1[-1].shape[0]; tau2 = 2
    F1m = sla.block_diag(*F1); H1m = np.hstack(H1); Pinf1 = sla.block_diag(*P1); L1v = np.vstack(L1)
    F2k, Lb, Qcb = [], [], []
    for d in range(D):                                    # :53-61
        F2d = np.array([[0.0, -omega[d]], [omega[d], 0.0]])
        F2k.append(np.kron(np.eye(tau1), F2d))
        Lb.append(np.kron(L1v[tau1 * d:tau1 * (d + 1)], np.eye(tau2)))
        Qcb.append(np.kron(np.array([[Qc1[d]]]), np.eye(tau2)))
    F = np.kron(F1m, np.eye(tau2)) + sla.block_diag(*F2k)  # :62
    L = sla.block_diag(*Lb); Qc = sla.block_diag(*Qcb)
    H = np.kron(H1m, np.array([[1.0, 0.0]]))                # :63
    Pinf = np.kron(Pinf1, np.eye(tau2))                     # :64
    A, Q = oss.lti_disc(F, L, Qc, 1.0)                      # :69
    return A, Q, H, Pinf, D * tau1, tau1


def steady_state(A, Q, C, vary):
    """The matrices kernel_ss_kalmanFastFB.m builds before its loops (:46-77, :127-132)."""
    H = np.asarray(C, float).reshape(1, -1); R = float(vary)
    PP = sla.solve_discrete_are(A.T, H.T, Q, np.array([[R]]))        # dare(A',H',Q,R)
    S = float((H @ PP @ H.T)[0, 0]) + R
    K = (PP @ H.T / S).ravel()
    AKHA = A - np.outer(K, (H @ A).ravel())
    PF2 = PP - np.outer(K, (H @ PP).ravel())
    HA = (H @ A).ravel()
    G = np.linalg.solve(PP.T, (PF2 @ A.T).T).T                       # PF2*A'/PP
    QQ = PF2 - G @ PP @ G.T; QQ = (QQ + QQ.T) / 2
    P = sla.solve_discrete_lyapunov(G, QQ)                           # dare(G',0,QQ)
    return dict(PP=PP, S=S, K=K, AKHA=AKHA, PF2=PF2, HA=HA, G=G, P=P)


def kernel_ss_kalmanFastFB(A, Q, C, P0, K, vary, y, verbose=0, KF=0):
    """kernel_ss_kalmanFastFB.m:1-168 -> lik, MS (S x T), P_last (filter covariance PF2), P_smooth (or None when KF=1).
    (The reference returns Xfin = reshape(MS,[1 S T]) and Pfin with PF2 / P repeated T times.)"""
    y = np.asarray(y, float).ravel(); T = y.size
    st = steady_state(A, Q, C, vary)
    m = np.zeros(A.shape[0]); MS = np.zeros((A.shape[0], T))
    lik = 0.5 * np.log(2 * np.pi) * T + 0.5 * np.log(st['S']) * T      # :80
    for k in range(T):                                                 # :83-110
        if not np.isnan(y[k]):
            v = y[k] - st['HA'] @ m
            m = st['AKHA'] @ m + st['K'] * y[k]
            lik += 0.5 * v ** 2 / st['S']
        else:
            m = A @ m
        MS[:, k] = m
    P_s = None
    if KF != 1:                                                        # :122-151
        for k in range(T - 2, -1, -1):
            m = MS[:, k] + st['G'] @ (m - A @ MS[:, k])
            MS[:, k] = m
        P_s = st['P']
    return -lik, MS, st['PF2'], P_s
