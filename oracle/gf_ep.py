"""ORACLE (test infrastructure only -- never imported by the product path).

Line-by-line CPU restatement (NumPy, FP64, dense "as written") of the reference's
full-covariance Power-EP Kalman filter + RTS smoother.
PARITY UNPINNED: the reference has no tests/golden vectors for this path and no
MATLAB/Octave exists here; pinned by tests/test_oracle_selfpins.py (Gaussian-site EP
== exact GP regression, etc).

Follows (file:line under /root/reference/matlab):
  gf_ep_modulator_nmf.m:58-66 (input merge), :92-352 (predict mode), :357-533 (nlml mode)
  gf_ep_modulator_nmf_constraints.m:75-121 (unpack + balance ON), rest identical
  gf_ep_modulator.m:69-81, :87-352 (predicts at k=1, SURVEY C-8), :355-538
Quirks reproduced: C-5, C-6 (association (K*H)*P), C-7 (jitter retry; deterministic
0.5 instead of rand, counted), C-8, C-9, C-15 (scalar damping broadcast).
"""
import math
import numpy as np
from . import ss as ssm


def merge_inputs(x, y, xt):
    """gf_ep_modulator_nmf.m:58-66."""
    x = np.asarray(x, float).ravel(); y = np.asarray(y, float).ravel()
    xt = np.asarray(xt, float).ravel() if xt is not None else np.zeros(0)
    xall = np.concatenate([x, xt])
    yall = np.concatenate([y, np.full(xt.size, np.nan)])
    # unique(xall,'first'): sorted unique values, index of FIRST occurrence, inverse map
    _, sort_ind, return_ind = np.unique(xall, return_index=True, return_inverse=True)
    yall = yall[sort_ind]
    return_ind = return_ind[xall.size - xt.size:] if xt.size else return_ind[:0]
    return yall, return_ind


def _damp(ep_damping, ep_itts):
    d = np.atleast_1d(np.asarray(ep_damping, float)).ravel()
    if d.size == 1:                                   # C-15: broadcast scalars
        d = np.full(max(ep_itts, 1), d[0])
    return d


def _chol_lower(PSkp, counters):
    """gf_ep_modulator_nmf.m:216-223 -- uses the lower triangle only, like MATLAB chol(.,'lower')."""
    Ls = np.tril(PSkp); Ls = Ls + np.tril(Ls, -1).T
    try:
        return np.linalg.cholesky(Ls)
    except np.linalg.LinAlgError:
        counters['chol_retries'] = counters.get('chol_retries', 0) + 1
        jitter = math.sqrt(1e-4) * np.diag(0.5 * np.ones(PSkp.shape[0]))   # C-7: rand -> 0.5
        Ls = np.tril(PSkp + jitter); Ls = Ls + np.tril(Ls, -1).T
        return np.linalg.cholesky(Ls)


def rts_step(A, Q, PSk, MSk, m, P, counters):
    """gf_ep_modulator_nmf.m:210-230."""
    PSkp = A @ PSk @ A.T + Q
    L = _chol_lower(PSkp, counters)
    # G = PSk*A'/L'/L
    B = PSk @ A.T
    X = np.linalg.solve(L, B.T).T          # B / L'
    G = np.linalg.solve(L.T, X.T).T        # X / L
    m = MSk + G @ (m - A @ MSk)
    P = PSk + G @ (P - PSkp) @ G.T
    return m, P


def site_update_filter(ttau_k, tnu_k, fmu, HPH, dlZ, d2lZ, ep_damp):
    """gf_ep_modulator_nmf.m:147-148."""
    with np.errstate(all='ignore'):
        ttau_k = (1 - ep_damp) * ttau_k + ep_damp * (-d2lZ / (1 + d2lZ * HPH))
        tnu_k = (1 - ep_damp) * tnu_k + ep_damp * ((dlZ - fmu * d2lZ) / (1 + d2lZ * HPH))
    return ttau_k, tnu_k


def matlab_max0(v):
    """MATLAB max(v,0): NaN -> 0 (SURVEY C-3)."""
    v = np.array(v, float)
    v[~(v > 0)] = 0.0
    return v


def ep_site_update_smoother(ttau_k, tnu_k, mm, vm, mom, lik_param, Wnmf, ep_fraction, ep_damp, yall, k):
    """gf_ep_modulator_nmf.m:242-259 (cavity, mom, damped power-EP update of the selected sites)."""
    with np.errstate(all='ignore'):
        v_cav = 1.0 / (1.0 / vm - ep_fraction * ttau_k)
        m_cav = v_cav * (mm / vm - ep_fraction * tnu_k)
        upd = v_cav > 0
        lZk, dlZ, d2lZ = mom(lik_param, m_cav, v_cav, Wnmf, ep_fraction, yall, k)
        ttau_k = ttau_k.copy(); tnu_k = tnu_k.copy()
        ttau_k[upd] = (1 - ep_damp * ep_fraction) * ttau_k[upd] + ep_damp * (-d2lZ[upd] / (1 + d2lZ[upd] * v_cav[upd]))
        tnu_k[upd] = (1 - ep_damp * ep_fraction) * tnu_k[upd] + ep_damp * (
            (dlZ[upd] - m_cav[upd] * d2lZ[upd]) / (1 + d2lZ[upd] * v_cav[upd]))
    return lZk, ttau_k, tnu_k, upd


def kalman_update_split(m, P, H, W, HPH, fmu, ttau_k, tnu_k):
    """gf_ep_modulator_nmf.m:159-176 -- per-site split on ttau==0, diagonal innovation (C-5),
    non-symmetric covariance forms with (K*H)*P association (C-6)."""
    with np.errstate(all='ignore'):
        ii = (ttau_k == 0)
        if np.any(ii):
            z = ttau_k[ii] * HPH[ii] + 1
            K = W[:, ii] * (ttau_k[ii] / z)[None, :]
            v = ttau_k[ii] * fmu[ii] - tnu_k[ii]
            m = m - W[:, ii] @ (v / z)
            P = P - K @ W[:, ii].T
        if np.any(~ii):
            jj = ~ii
            K = W[:, jj] / (HPH[jj] + 1.0 / ttau_k[jj])[None, :]
            v = tnu_k[jj] / ttau_k[jj] - fmu[jj]
            m = m + K @ v
            P = P - (K @ H[jj, :]) @ P
    return m, P


def kalman_update_legacy(m, P, H, W, fs2, fmu, ttau_k, tnu_k):
    """gf_ep_modulator_nmf.m:428-439 (nlml mode, C-9)."""
    with np.errstate(all='ignore'):
        if np.min(ttau_k) == 0:
            z = ttau_k * fs2 + 1
            K = W * (ttau_k / z)[None, :]
            v = ttau_k * fmu - tnu_k
            m = m - W @ (v / z)
            P = P - K @ W.T
        else:
            K = W / (fs2 + 1.0 / ttau_k)[None, :]
            v = tnu_k / ttau_k - fmu
            m = m + K @ v
            P = P - (K @ H) @ P
    return m, P


def run_predict(model, yall, mom, ep_fraction, ep_damping, ep_itts, predict_at_k1=False, verbose=False, sites0=None):
    """gf_ep_modulator_nmf.m:92-352 on an assembled model dict
    {A,Q,H,Pinf,Wnmf,lik_param}. Returns dict with Eft,Varft (all steps), MS,PS,ttau,tnu,R,lZ,nlZ,
    maxDiffM,maxDiffP (per sweep), MF,PF (filtered, last sweep).
    sites0 = (ttau0, tnu0): NOT in the reference (which starts from zeros, :96-97) -- the initial sites of a warm start,
    used only to check the library's nagp_plan_upload_sites extension."""
    A, Q, H, Pinf, Wnmf, lik_param = (model[k] for k in ('A', 'Q', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    S = A.shape[0]; M = H.shape[0]; T = yall.size
    MS = np.zeros((S, T)); PS = np.zeros((T, S, S))
    ttau = np.zeros((M, T)); tnu = np.zeros((M, T)); lZ = np.zeros(T); R = np.zeros((M, T))
    if sites0 is not None:
        ttau = np.array(sites0[0], dtype=float).copy(); tnu = np.array(sites0[1], dtype=float).copy()
    nlZ = np.zeros(ep_itts); mdM = np.zeros(ep_itts); mdP = np.zeros(ep_itts)
    damp = _damp(ep_damping, ep_itts)
    ep_damp = damp[0]
    counters = {}
    MF = PF = None
    for itt in range(1, ep_itts + 1):
        m = np.zeros(S); P = Pinf.copy()
        maxDiffP = 0.0; maxDiffM = 0.0
        PSP = PS.copy(); MSP = MS.copy()
        for k in range(T):
            if k > 0 or predict_at_k1:
                m = A @ m
                P = A @ P @ A.T + Q
            if not np.isnan(yall[k]):
                fmu = H @ m; W = P @ H.T; HPH = np.diag(H @ P @ H.T).copy()
                if itt == 1 or k == T - 1:
                    lZ[k], dlZ, d2lZ = mom(lik_param, fmu, HPH, Wnmf, 1.0, yall, k)
                    ttau[:, k], tnu[:, k] = site_update_filter(ttau[:, k], tnu[:, k], fmu, HPH, dlZ, d2lZ, ep_damp)
                    ttau[:, k] = matlab_max0(ttau[:, k])
                    with np.errstate(all='ignore'):
                        R[:, k] = 1.0 / ttau[:, k]
                m, P = kalman_update_split(m, P, H, W, HPH, fmu, ttau[:, k], tnu[:, k])
            MS[:, k] = m; PS[k] = P
        if itt == 1:
            nlZ[0] = -np.sum(lZ)
        MF = MS.copy(); PF = PS.copy()
        if itt < ep_itts:
            ep_damp = damp[itt]
        for k in range(T - 2, -1, -1):
            m, P = rts_step(A, Q, PS[k], MS[:, k], m, P, counters)
            MS[:, k] = m; PS[k] = P
            if itt < ep_itts and not np.isnan(yall[k]):
                mm = H @ m; vm = np.diag(H @ P @ H.T).copy()
                lZ[k], ttau[:, k], tnu[:, k], _ = ep_site_update_smoother(
                    ttau[:, k], tnu[:, k], mm, vm, mom, lik_param, Wnmf, ep_fraction, ep_damp, yall, k)
                ttau[:, k] = matlab_max0(ttau[:, k])
                with np.errstate(all='ignore'):
                    R[:, k] = 1.0 / ttau[:, k]
            maxDiffM = max(maxDiffM, np.max(np.abs(H @ MSP[:, k] - H @ m)))
            maxDiffP = max(maxDiffP, np.max(np.abs(H @ PSP[k] @ H.T - H @ P @ H.T)))
        if itt < ep_itts:
            nlZ[itt] = -np.sum(lZ)
        mdM[itt - 1] = maxDiffM; mdP[itt - 1] = maxDiffP
        if verbose:
            print('%02i - max diff in m: %.6g - max diff in P: %.6g - nll: %.6g' % (itt, maxDiffM, maxDiffP, nlZ[itt - 1]))
    Eft = H @ MS
    Varft = np.stack([np.diag(H @ PS[k] @ H.T) for k in range(T)], axis=1)
    return dict(Eft=Eft, Varft=Varft, MS=MS, PS=PS, ttau=ttau, tnu=tnu, R=R, lZ=lZ, nlZ=nlZ,
                maxDiffM=mdM, maxDiffP=mdP, MF=MF, PF=PF, counters=counters)


def run_nlml(model, yall, mom, ep_fraction, ep_damping, ep_itts):
    """gf_ep_modulator_nmf.m:357-533 -> (edata, lZ, ttau, tnu)."""
    A, Q, H, Pinf, Wnmf, lik_param = (model[k] for k in ('A', 'Q', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    S = A.shape[0]; M = H.shape[0]; T = yall.size
    ttau = np.zeros((M, T)); tnu = np.zeros((M, T)); lZ = np.zeros(T)
    MS = np.zeros((S, T)); PS = np.zeros((T, S, S)) if ep_itts > 1 else None
    damp = _damp(ep_damping, ep_itts)
    ep_damp = damp[0]
    counters = {}
    for itt in range(1, ep_itts + 1):
        m = np.zeros(S); P = Pinf.copy()
        if itt == 1 or itt < ep_itts:
            for k in range(T):
                if k > 0:
                    m = A @ m; P = A @ P @ A.T + Q
                if not np.isnan(yall[k]):
                    fmu = H @ m; W = P @ H.T; fs2 = np.diag(H @ P @ H.T).copy()
                    if itt == 1 or k == T - 1:
                        lZ[k], dlZ, d2lZ = mom(lik_param, fmu, fs2, Wnmf, 1.0, yall, k)
                        ttau[:, k], tnu[:, k] = site_update_filter(ttau[:, k], tnu[:, k], fmu, fs2, dlZ, d2lZ, ep_damp)
                    ttau[:, k] = matlab_max0(ttau[:, k])
                    m, P = kalman_update_legacy(m, P, H, W, fs2, fmu, ttau[:, k], tnu[:, k])
                if itt < ep_itts:
                    MS[:, k] = m; PS[k] = P
        if itt < ep_itts:
            ep_damp = damp[itt]
            for k in range(T - 2, -1, -1):
                m, P = rts_step(A, Q, PS[k], MS[:, k], m, P, counters)
                MS[:, k] = m; PS[k] = P
                if not np.isnan(yall[k]):
                    mm = H @ m; vm = np.diag(H @ P @ H.T).copy()
                    lZ[k], ttau[:, k], tnu[:, k], _ = ep_site_update_smoother(
                        ttau[:, k], tnu[:, k], mm, vm, mom, lik_param, Wnmf, ep_fraction, ep_damp, yall, k)
    return -np.sum(lZ), lZ, ttau, tnu


def build_model_nmf(w, kernel1, kernel2, num_lik_params, D, N, balance):
    """gf_ep_modulator_nmf.m:72-86,108."""
    lik_param, param1, param2, Wnmf = ssm.unpack_log(w, num_lik_params, D, N)
    return assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance)


def assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance, symmetrize_Q=False):
    F, L, Qc, H, Pinf = ssm.ss_modulators_nmf(param1, param2, kernel1, kernel2)
    T = None
    if balance:
        F, L, H, Pinf, T = ssm.balance_ss(F, L, H, Pinf)
    A, Q = ssm.lti_disc(F, L, Qc, 1.0)
    if symmetrize_Q:                                   # ihgp_ep_modulator_nmf.m:97
        Q = (Q + Q.T) / 2
    return dict(A=A, Q=Q, H=H, Pinf=Pinf, Wnmf=Wnmf, lik_param=lik_param, F=F, L=L, Qc=Qc, Tbal=T)


def _outputs(res, return_ind, nargout):
    Eft = res['Eft'][:, return_ind]; Varft = res['Varft'][:, return_ind]
    if nargout <= 2:
        return (Eft, Varft)[:max(nargout, 1)] if nargout > 1 else Eft
    lb = Eft - 1.96 * np.sqrt(Varft); ub = Eft + 1.96 * np.sqrt(Varft)
    return Eft, Varft, None, lb, ub, res


def gf_ep_modulator_nmf(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N,
                        ep_fraction=0.5, ep_damping=None, ep_itts=30, nargout=6):
    """gf_ep_modulator_nmf.m:1 -- same positional signature; `ss` is accepted for signature
    fidelity (the oracle always uses ss_modulators_nmf, as every driver does)."""
    yall, return_ind = merge_inputs(x, y, xt)
    model = build_model_nmf(w, kernel1, kernel2, num_lik_params, D, N, balance=False)   # :80 `if false`
    if xt is not None and np.size(xt) > 0:
        res = run_predict(model, yall, mom, ep_fraction, ep_damping, ep_itts)
        return _outputs(res, return_ind, nargout)
    edata, *_ = run_nlml(model, yall, mom, ep_fraction, ep_damping, ep_itts)
    return edata, np.zeros(np.size(w))


def gf_ep_modulator_nmf_constraints(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N,
                                    ep_fraction, ep_damping, ep_itts, constraints, w_fixed, tune_hypers, nargout=6):
    """gf_ep_modulator_nmf_constraints.m:1-2."""
    yall, return_ind = merge_inputs(x, y, xt)
    lik_param, param1, param2, Wnmf = ssm.unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
    model = assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance=True)   # :115 `if true`
    if xt is not None and np.size(xt) > 0:
        res = run_predict(model, yall, mom, ep_fraction, ep_damping, ep_itts)
        return _outputs(res, return_ind, nargout)
    edata, *_ = run_nlml(model, yall, mom, ep_fraction, ep_damping, ep_itts)
    return edata, np.zeros(np.size(w))


def gf_ep_modulator(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params,
                    ep_fraction=0.5, ep_damping=None, ep_itts=30, nargout=6):
    """gf_ep_modulator.m:1 -- one modulator per sub-band, balance ON (:75), predicts at k=1
    in predict mode (:131-133) but not in nlml mode (:399-402)."""
    yall, return_ind = merge_inputs(x, y, xt)
    w = np.asarray(w, float).ravel()
    lik_param = w[:num_lik_params]
    param = np.exp(w[num_lik_params:])
    D = param.size // 5
    model = assemble(lik_param, param[:3 * D], param[3 * D:], None, kernel1, kernel2, balance=True)
    mom7 = lambda hyp, mu, s2, Wn, a, yy, k: mom(hyp, mu, s2, None, a, yy, k)
    if xt is not None and np.size(xt) > 0:
        res = run_predict(model, yall, mom7, ep_fraction, ep_damping, ep_itts, predict_at_k1=True)
        return _outputs(res, return_ind, nargout)
    edata, *_ = run_nlml(model, yall, mom7, ep_fraction, ep_damping, ep_itts)
    return edata, np.zeros(np.size(w))
