"""ORACLE (test infrastructure only -- never imported by the product path).

Line-by-line CPU restatement of the reference's globally-iterated EKF filter + RTS smoother.
PARITY UNPINNED (no reference fixtures, no MATLAB); self-pinned by tests/test_oracle_selfpins.py
(Jacobian vs finite differences; linear measurement model == Kalman filter).

Follows (file:line under /root/reference/matlab):
  ekf_update1.m:103-109, iekf_update1.m:110-117
  gf_giekf_modulator_nmf.m:55-106 (set-up, balance ON), :108-291 (predict mode), :445-459 (h, Jacobian)
  gf_giekf_modulator_nmf_constraints.m:144-327 (P reset every global iteration, :163-168),
      :492-502 (corrected Jacobian  dy = partials'*H, used for BOTH variants here, SURVEY C-13)
  gf_giekf_modulator_nmf_constraints.m:332-480 with GradObj='off' (row a11 as train_GTFNMF.m:199 uses it): run_nlml.
      GradObj='on' is not restated (the gradient loop indexes numel(w) outputs with size(dF,3) slices, :339-341, 432-441);
      the nlml branch of the non-constraints file cannot run as committed (mm/PP undefined with GradObj='off', :320-370;
      funhd/funhd2 select columns with sum(H,1)==1 after balancing, :457, 471).
"""
import math
import scipy.linalg as sla
import numpy as np
from . import ss as ssm
from .gf_ep import merge_inputs, rts_step, assemble


def linkf(x):
    return np.log(1.0 + np.exp(x))


def dlinkf(x):
    return np.exp(x) / (np.exp(x) + 1.0)


def funh(x, H, D, N, W):
    """gf_giekf_modulator_nmf.m:445-449."""
    return float((H[:D] @ x) @ W @ linkf(H[D:D + N] @ x))


def funhd(x, H, D, N, W):
    """gf_giekf_modulator_nmf_constraints.m:497-502 (dy = partials'*H)."""
    g = H[D:D + N] @ x
    partials = np.concatenate([W @ linkf(g), ((H[:D] @ x) @ W) * dlinkf(g)])
    return partials @ H


def ekf_update1(M, P, y, Hfun, R, hfun):
    """ekf_update1.m:103-109 (scalar measurement)."""
    H_ = Hfun(M); MU = hfun(M)
    S = R + H_ @ P @ H_
    K = P @ H_ / S
    M = M + K * (y - MU)
    P = P - np.outer(K, K) * S
    return M, P, K, MU, S


def iekf_update1(M, P, y, Hfun, R, hfun, iters=5):
    """iekf_update1.m:110-117 (not the textbook IEKF -- reproduced as written)."""
    for _ in range(iters):
        H_ = Hfun(M); MU = hfun(M)
        S = R + H_ @ P @ H_
        K = P @ H_ / S
        M = M + K * (y - MU)
    P = P - np.outer(K, K) * S
    return M, P, K, MU, S


def run_predict(model, yall, D, N, g_iter, l_iter, constraints_variant=False):
    """gf_giekf_modulator_nmf.m:108-291."""
    A, Q, H, Pinf, Wnmf, lik_param = (model[k] for k in ('A', 'Q', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    S = A.shape[0]; T = yall.size
    sigma2 = float(np.exp(np.ravel(lik_param)[0]))
    MS = np.zeros((S, T)); PS = np.zeros((T, S, S))
    hfun = lambda x: funh(x, H, D, N, Wnmf)
    Hfun = lambda x: funhd(x, H, D, N, Wnmf)
    counters = {}
    mdP = np.zeros(g_iter)
    m = None; P = None
    for itt in range(1, g_iter + 1):
        if itt == 1:
            m = np.zeros(S); P = Pinf.copy()
        if constraints_variant:
            P = Pinf.copy()                                   # _constraints.m:168
        maxDiffP = 0.0; PSP = PS.copy()
        for k in range(T):
            if k > 0:
                m = A @ m; P = A @ P @ A.T + Q
            if not np.isnan(yall[k]):
                m, P, *_ = iekf_update1(m, P, yall[k], Hfun, sigma2, hfun, l_iter)
            MS[:, k] = m; PS[k] = P
        MF = MS.copy(); PF = PS.copy()
        for k in range(T - 2, -1, -1):
            m, P = rts_step(A, Q, PS[k], MS[:, k], m, P, counters)
            MS[:, k] = m; PS[k] = P
            maxDiffP = max(maxDiffP, np.max(np.abs(H @ PSP[k] @ H.T - H @ P @ H.T)))
        mdP[itt - 1] = maxDiffP
    Eft = H @ MS
    Varft = np.stack([np.diag(H @ PS[k] @ H.T) for k in range(T)], axis=1)
    return dict(Eft=Eft, Varft=Varft, MS=MS, PS=PS, MF=MF, PF=PF, maxDiffP=mdP, counters=counters)


def gf_giekf_modulator_nmf(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N, g_iter, l_iter,
                           GradObj='off', nargout=6):
    """gf_giekf_modulator_nmf.m:1-2 (predict mode)."""
    yall, return_ind = merge_inputs(x, y, xt)
    lik_param, param1, param2, Wnmf = ssm.unpack_log(w, num_lik_params, D, N)
    model = assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance=True)
    if xt is None or np.size(xt) == 0:
        raise NotImplementedError('EKF nlml/gradient mode not restated (SURVEY f-4)')
    res = run_predict(model, yall, D, N, g_iter, l_iter, constraints_variant=False)
    return _outputs(res, return_ind, nargout)


def gf_giekf_modulator_nmf_constraints(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N, g_iter, l_iter,
                                       constraints, w_fixed, tune_hypers, GradObj='off', nargout=6):
    """gf_giekf_modulator_nmf_constraints.m:1-2 (predict mode)."""
    yall, return_ind = merge_inputs(x, y, xt)
    lik_param, param1, param2, Wnmf = ssm.unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
    model = assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance=True)
    if xt is None or np.size(xt) == 0:
        raise NotImplementedError('EKF nlml/gradient mode not restated (SURVEY f-4)')
    res = run_predict(model, yall, D, N, g_iter, l_iter, constraints_variant=True)
    return _outputs(res, return_ind, nargout)


def _outputs(res, return_ind, nargout):
    Eft = res['Eft'][:, return_ind]; Varft = res['Varft'][:, return_ind]
    if nargout <= 1:
        return Eft
    if nargout <= 3:
        return Eft, Varft
    lb = Eft - 1.96 * np.sqrt(Varft); ub = Eft + 1.96 * np.sqrt(Varft)
    return Eft, Varft, None, lb, ub, res


def run_nlml(model, yall, D, N):
    """gf_giekf_modulator_nmf_constraints.m:332-480, GradObj='off' (nparam = 0): edata only.  `model` is the balanced
    assembly (F, Pinf, H, Wnmf, lik_param); A = expm(F), Q = Pinf - A*Pinf*A' (:377-378); prediction at every step
    including the first (:405-406); no isnan guard."""
    F, H, Pinf, Wnmf, lik_param = (model[k] for k in ('F', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    R = math.exp(float(np.ravel(lik_param)[0]))
    A = sla.expm(F)
    Q = Pinf - A @ Pinf @ A.T
    m = np.zeros(F.shape[0]); P = Pinf.copy()
    edata = 0.0
    for k in range(yall.size):
        m = A @ m
        P = A @ P @ A.T + Q
        mu = funh(m, H, D, N, Wnmf)
        JH = funhd(m, H, D, N, Wnmf)
        S = JH @ P @ JH + R
        if not S > 0:                                   # chol failed -> jitter 1e-4*rand (rand -> 0.5, C-7)
            S = S + 0.5e-4
            if not S > 0:
                return float('nan')                     # :423-426
        LS = math.sqrt(S)
        HtiS = JH / LS / LS
        K = P @ HtiS
        v = yall[k] - mu
        vtiS = v / LS / LS
        edata = edata + 0.5 * math.log(2 * math.pi) + math.log(LS) + 0.5 * vtiS * v
        m = m + K * v
        P = P - np.outer(K, K) * S
    return edata


def gf_giekf_modulator_nmf_constraints_nlml(w, x, y, kernel1, kernel2, num_lik_params, D, N, constraints, w_fixed, tune_hypers):
    """[e, eg] of gf_giekf_modulator_nmf_constraints(w,x,y,ss,mom,[],...,GradObj='off')."""
    yall, _ = merge_inputs(x, y, None)
    lik_param, param1, param2, Wnmf = ssm.unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
    model = assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance=True)
    return run_nlml(model, yall, D, N), np.zeros(np.size(w))


# ---------------------------------------------------------------------------------------------
# EKF energy WITH its gradient recursion: gf_giekf_modulator_nmf_constraints.m:332-480, GradObj='on'
def d2linkf(x):
    return dlinkf(x) * (1.0 - dlinkf(x))


def funhd2(x, H, D, N, W):
    """gf_giekf_modulator_nmf_constraints.m:505-514 (Hessian of h)."""
    z = H[:D] @ x; g = H[D:D + N] @ x
    Wd = W * dlinkf(g)[None, :]
    partials = np.block([[np.zeros((D, D)), Wd], [Wd.T, np.diag((z @ W) * d2linkf(g))]])
    return H.T @ partials @ H


def grad_setup(model, param1, param2, kernel1, kernel2, consistent=False):
    """The derivative inputs of the recursion: AA_j = expm([F 0; dF_j F]) (:355-366), dPinf_j, dR_j for the nparam = 1+3D+2N slices
    the reference builds (:121-125: a zero slice for the noise parameter in front).
    consistent=False -- as the reference: the UNBALANCED dF, dPinf of ss_modulators_nmf beside the balanced F, Pinf (:117-119 commented out).
    consistent=True  -- the derivative stacks carried through the balancing transformation (T\\dF*T, T\\dPinf/T'): the recursion then
                        differentiates the energy it runs beside, which central differences can confirm (test pin, not reference behaviour)."""
    F, Pinf = model['F'], model['Pinf']
    S = F.shape[0]
    dF0, dP0 = ssm.ss_modulators_nmf_derivs(param1, param2, kernel1, kernel2)
    if consistent and model.get('Tbal') is not None:
        T = model['Tbal']; Ti = np.linalg.inv(T)
        dF0 = np.stack([Ti @ a @ T for a in dF0]); dP0 = np.stack([Ti @ a @ Ti.T for a in dP0])
    dF = np.concatenate([np.zeros((1, S, S)), dF0]); dP = np.concatenate([np.zeros((1, S, S)), dP0])
    nparam = dF.shape[0]
    AA = np.stack([sla.expm(np.block([[F, np.zeros((S, S))], [dF[j], F]])) for j in range(nparam)])
    dR = np.zeros(nparam); dR[0] = 1.0
    return dict(AA=AA, dPinf=dP, dR=dR)


def run_nlml_grad(model, gs, yall, D, N, n_w, consistent=False):
    """gf_giekf_modulator_nmf_constraints.m:332-480 with GradObj='on' -> (edata, gdata[n_w]).
    consistent=False: the statements as written -- gdata has length(w) = n_w entries while the loop runs over nparam = size(dF,3)
    (:336-342): MATLAB stops with an index error when n_w < nparam (IndexError here); for j > nparam - D*N the Jacobian derivative
    is taken w.r.t. entry j-nparam+D*N of W (:440-444) while dm, dP still carry kernel parameter j; derivatives are w.r.t. the
    natural parameters (sigma2 itself for j = 1), the chain rule to w is commented out (:480).
    consistent=True: the same recursion as the true gradient of the energy w.r.t. [sigma2, kernel parameters (3D+2N), W(:) (D*N)]:
    kernel parameters take the Hessian term only, W entries their own slices (dF = dPinf = 0) with the direct terms d mu/dW, dJH/dW."""
    F, H, Pinf, Wnmf, lik_param = (model[k] for k in ('F', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    R = math.exp(float(np.ravel(lik_param)[0]))
    AA, dPinf, dR = gs['AA'], gs['dPinf'], gs['dR']
    d = F.shape[0]; nparam = AA.shape[0]
    if consistent:
        AW = np.block([[sla.expm(F), np.zeros((d, d))], [np.zeros((d, d)), sla.expm(F)]])
        AA = np.concatenate([AA, np.repeat(AW[None], D * N, axis=0)])
        dPinf = np.concatenate([dPinf, np.zeros((D * N, d, d))]); dR = np.concatenate([dR, np.zeros(D * N)])
        nparam += D * N
    if n_w < nparam:
        raise IndexError('gdata(j) read past length(w) = %d (nparam = %d): Index exceeds the number of array elements' % (n_w, nparam))
    gdata = np.zeros(n_w); edata = 0.0
    m = np.zeros(d); P = Pinf.copy()
    dm = np.zeros((d, nparam)); dP = dPinf.copy()
    A = sla.expm(F); Q = Pinf - A @ Pinf @ A.T
    nk = nparam - D * N                                           # j <= nparam - D*N (1-based): Hessian branch (:438)
    for k in range(yall.size):
        for j in range(nparam):
            dm[:, j] = AA[j][d:, :] @ np.concatenate([m, dm[:, j]])
            dA = AA[j][d:, :d]
            dAPinfAt = dA @ Pinf @ A.T
            dQ = dPinf[j] - dAPinfAt - A @ dPinf[j] @ A.T - dAPinfAt.T
            dAPAt = dA @ P @ A.T
            dP[j] = dAPAt + A @ dP[j] @ A.T + dAPAt.T + dQ
        m = A @ m; P = A @ P @ A.T + Q
        mu = funh(m, H, D, N, Wnmf); JH = funhd(m, H, D, N, Wnmf); dJH = funhd2(m, H, D, N, Wnmf)
        S = JH @ P @ JH + R
        if not S > 0:
            S = S + 0.5e-4
            if not S > 0:
                return float('nan'), np.full(n_w, np.nan)
        HtiS = JH / S; K = P @ HtiS; v = yall[k] - mu; vtiS = v / S
        for j in range(nparam):
            dmu_dir = 0.0
            if consistent:
                dmdJH = dm[:, j] @ dJH
                if j >= nk:
                    W_ = np.zeros(D * N); W_[j - nk] = 1.0; W_ = W_.reshape((D, N), order='F')
                    dmdJH = dmdJH + funhd(m, H, D, N, W_)
                    dmu_dir = funh(m, H, D, N, W_)
            elif j < nk:
                dmdJH = dm[:, j] @ dJH
            else:
                W_ = np.zeros(D * N); W_[j - nk] = 1.0; W_ = W_.reshape((D, N), order='F')
                dmdJH = funhd(m, H, D, N, W_)
            dmu = JH @ dm[:, j] + dmu_dir
            dS = dmdJH @ P @ JH + JH @ dP[j] @ JH + JH @ P @ dmdJH + dR[j]
            gdata[j] += 0.5 * dS / S - 0.5 * dmu * vtiS - 0.5 * vtiS * dS * vtiS - 0.5 * vtiS * dmu
            dK = dP[j] @ HtiS + P @ dmdJH / S - P @ HtiS * dS / S
            dm[:, j] = dm[:, j] + dK * v - K * dmu
            dKSKt = np.outer(dK, K) * S
            dP[j] = dP[j] - dKSKt - np.outer(K, K) * dS - dKSKt.T
        edata += 0.5 * math.log(2 * math.pi) + math.log(math.sqrt(S)) + 0.5 * vtiS * v
        m = m + K * v; P = P - np.outer(K, K) * S
    return edata, gdata
