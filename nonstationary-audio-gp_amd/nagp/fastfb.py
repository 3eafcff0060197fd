"""Host-side mirror of the stationary filterbank path:

    [A,Q,H,Pinf,K,tau1] = get_disc_model(lamx,varx,omega,D,kernel,se_approx_order)     unifying_prob_tf/get_disc_model.m
    [lik,Xfin,Pfin]     = kernel_ss_kalmanFastFB(A,Q,C,P0,K,vary,y,verbose,KF)         unifying_prob_tf/kernel_ss_kalmanFastFB.m

The set-up (`dare`, stationary gain, smoother gain) stays on the host exactly where the .m has it -- SciPy's
solve_discrete_are / solve_discrete_lyapunov stand for the Control-System-Toolbox `dare` -- and the two O(T) loops run on
the GPU (nagp_fastfb_run).  There is no CPU fallback.
"""
import ctypes as C

import numpy as np
import scipy.linalg as sla

from . import _lib as L
from . import ss as ssm


def get_disc_model(lamx, varx, omega, D, kernel, se_approx_order=6):
    """get_disc_model.m:1-72.  Sub-band d = `kernel`(varx_d, lengthscale(lamx_d)) x cosine(omega_d); the observation row H
    sums the real parts of all sub-bands.  Returns A, Q, H (1 x S), Pinf, K (= D*tau1), tau1."""
    lamx = np.asarray(lamx, float).ravel(); varx = np.asarray(varx, float).ravel(); omega = np.asarray(omega, float).ravel()
    if kernel not in ('exp', 'matern32', 'matern52'):
        raise ValueError("kernel must be 'exp', 'matern32' or 'matern52'")
    ls = {'exp': 1.0, 'matern32': np.sqrt(3.0), 'matern52': np.sqrt(5.0)}[kernel] / lamx     # :9-20
    Ab, Qb, Pb, Hb = [], [], [], []
    for d in range(D):
        F1, L1, Qc1, P1 = ssm.kernel_block(kernel, varx[d], ls[d])
        tau1 = F1.shape[0]
        F = np.kron(F1, np.eye(2)) + np.kron(np.eye(tau1), np.array([[0.0, -omega[d]], [omega[d], 0.0]]))     # :53-62
        Lm = np.kron(L1, np.eye(2)); LQL = Qc1 * (Lm @ Lm.T)
        A_d, Q_d = ssm.lti_disc_block(F, LQL, 1.0)                                                             # :69, block-wise
        H1 = np.zeros((1, tau1)); H1[0, 0] = 1.0
        Ab.append(A_d); Qb.append(Q_d); Pb.append(np.kron(P1, np.eye(2))); Hb.append(np.kron(H1, np.array([[1.0, 0.0]])))
    return sla.block_diag(*Ab), sla.block_diag(*Qb), np.hstack(Hb), sla.block_diag(*Pb), D * tau1, tau1


def kernel_ss_kalmanFastFB(A, Q, C_, P0, K, vary, y, verbose=0, KF=0, steady=False, device=0):
    """[lik,Xfin,Pfin] = kernel_ss_kalmanFastFB(A,Q,C,P0,K,vary,y,verbose,KF) (kernel_ss_kalmanFastFB.m:1).
    Xfin is 1 x S x T, Pfin S x S x T (every slice the steady-state covariance, the last one the filter's -- as the .m
    stores them); steady=True returns Pfin = (P_smoother or None, P_filter) instead of the T-fold copy."""
    A = L.f64(A); S = A.shape[0]
    H = np.asarray(C_, float).reshape(1, -1); R = float(np.ravel(vary)[0])
    y = L.f64(np.asarray(y, float).ravel(), 'C'); T = y.size
    try:
        PP = sla.solve_discrete_are(A.T, H.T, np.asarray(Q, float), np.array([[R]]))      # :46
    except Exception as e:                                                                 # :51-53
        raise L.NagpError('Unstable DARE solution!') from e
    Sinn = float((H @ PP @ H.T)[0, 0]) + R
    Kg = (PP @ H.T / Sinn).ravel()                                                         # :56
    HA = (H @ A).ravel()
    AKHA = L.f64(A - np.outer(Kg, HA))                                                     # :59
    PF2 = PP - np.outer(Kg, (H @ PP).ravel())                                              # :73
    G = None; Psm = None
    if KF != 1:
        G = np.linalg.solve(PP.T, (PF2 @ A.T).T).T                                         # :127  PF2*A'/PP
        QQ = PF2 - G @ PP @ G.T; QQ = (QQ + QQ.T) / 2                                      # :130-131
        Psm = sla.solve_discrete_lyapunov(G, QQ)                                           # :132  dare(G',0,QQ)
        G = L.f64(G)
    MS = np.zeros((S, T), order='F'); sv2 = C.c_double(0.0)
    L.check(L.lib().nagp_fastfb_run(S, L.dptr(A), L.dptr(AKHA), L.dptr(L.f64(HA, 'C')), L.dptr(L.f64(Kg, 'C')), L.dptr(G),
                                    L.dptr(y), T, L.dptr(MS), C.byref(sv2), int(device)))
    n_obs = T                                                                              # :80 counts every step, observed or not
    lik = -(0.5 * np.log(2 * np.pi) * n_obs + 0.5 * np.log(Sinn) * n_obs + 0.5 * sv2.value / Sinn)
    Xfin = MS.reshape(1, S, T, order='F')
    if steady:
        return lik, Xfin, (Psm, PF2)
    if S * S * T * 8 > (1 << 30):
        raise MemoryError('Pfin would take %.1f GiB (the reference stores the same S x S matrix T times); pass steady=True'
                          % (S * S * T * 8 / 2 ** 30))
    Pfin = np.empty((S, S, T), order='F')
    Pfin[:] = (PF2 if Psm is None else Psm)[:, :, None]
    Pfin[:, :, T - 1] = PF2
    return lik, Xfin, Pfin
