"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the source-separation variants: J GT-NMF models stacked into one state space
(block-diagonal Wnmf), the older Power-EP rule.  PARITY UNPINNED (no reference fixtures, no MATLAB);
self-pinned by tests/test_oracle_selfpins.py (J = 1 with ep_fraction = 1 equals the main functions).

Follows (file:line under /root/reference/matlab/experiments):
  gf_ep_mods_nmf_mixture.m:76-84 (input merge), :89-128 (stacking), :137-349 (EP sweeps, outputs)
  ihgp_ep_mods_nmf_mixture.m:86-125 (stacking), :129-234 (DARE tables, as ihgp_ep_modulator_nmf.m),
      :236-472 (EP sweeps), :501-536 (outputs)
Differences to gf_ep_modulator_nmf.m / ihgp_ep_modulator_nmf.m that are reproduced here:
  * w is a cell {log sn2, {param1_j}, {param2_j}, {W_j}} with param1/param2/W in natural units; no balancing
  * `mom` is called with SIX arguments (hyp,mu,s2,Wnmf,yall,k) (gf :183,277; ihgp :291,442): the power-EP
    fraction lives inside the closure, so the filter's ADF step also runs at power ep_fraction.
    (source_sep_piano.m:94 as committed builds a 7-argument closure, which these call sites cannot take;
    the 6-argument form is the commented-out line :93.)  `mom` below is the oracle's 7-argument Mom and
    ep_fraction is bound here.
  * site refresh  (1-d)*site + d/ep_fraction*(...)  in the filter and in the smoother (gf :186-187,280-281)
  * gf: R = 1/ttau BEFORE the clamp, clamp at every filter step (:190-195), single-branch update
    (:197-208), no clamp after the smoother's refresh (:280-284)
  * ihgp: R starts at 0 (:248), maxDiffM only over refreshed steps (:458), no abs(Varft) (:510)
  * scalar ep_damping
"""
import numpy as np
from . import ss as ssm
from .gf_ep import merge_inputs, matlab_max0, rts_step, kalman_update_legacy
from .ihgp import build_tables, nearest_index


def stack_models(w, kernel1, kernel2, J):
    """gf_ep_mods_nmf_mixture.m:89-128: all sub-band blocks of all sources first, then all modulator blocks."""
    lik_param = w[0]
    z = [[] for _ in range(5)]; g = [[] for _ in range(5)]; Ws = []
    D = 0; N = 0
    for j in range(J):
        param1 = np.asarray(w[1][j], float).ravel(); param2 = np.asarray(w[2][j], float).ravel()
        D_ = param1.size // 3; N_ = param2.size // 2
        D += D_; N += N_
        Ws.append(np.atleast_2d(np.asarray(w[3][j], float)))
        tau1 = ssm._CF[kernel1[j]](1.0, 1.0)[0].shape[0]
        z_tau = 2 * tau1
        parts = ssm.ss_modulators_nmf(param1, param2, kernel1[j], kernel2[j])   # F, L, Qc, H, Pinf
        F_j, L_j, Qc_j, H_j, Pinf_j = parts
        nz = D_ * z_tau; cz = D_ * 2
        for lst, blk in zip(z, (F_j[:nz, :nz], L_j[:nz, :cz], Qc_j[:cz, :cz], H_j[:D_, :nz], Pinf_j[:nz, :nz])):
            lst.append(blk)
        for lst, blk in zip(g, (F_j[nz:, nz:], L_j[nz:, cz:], Qc_j[cz:, cz:], H_j[D_:, nz:], Pinf_j[nz:, nz:])):
            lst.append(blk)
    F, L, Qc, H, Pinf = (ssm._blkdiag(zz + gg) for zz, gg in zip(z, g))
    Wnmf = ssm._blkdiag(Ws)
    return dict(F=F, L=L, Qc=Qc, H=H, Pinf=Pinf, Wnmf=Wnmf, lik_param=lik_param, D=D, N=N)


def _refresh(site_t, site_n, mean, var, dlZ, d2lZ, d, a, sel=None):
    """(1-d)*site + d/a*(...)  (gf_ep_mods_nmf_mixture.m:186-187, 280-281)."""
    with np.errstate(all='ignore'):
        t_new = (1 - d) * site_t + d / a * (-d2lZ / (1 + d2lZ * var))
        n_new = (1 - d) * site_n + d / a * ((dlZ - mean * d2lZ) / (1 + d2lZ * var))
    if sel is None:
        return t_new, n_new
    t = site_t.copy(); n = site_n.copy()
    t[sel] = t_new[sel]; n[sel] = n_new[sel]
    return t, n


def run_gf(model, yall, mom, ep_fraction, ep_damping, ep_itts):
    """gf_ep_mods_nmf_mixture.m:137-349."""
    F, L, Qc, H, Pinf, Wnmf, lik_param = (model[k] for k in ('F', 'L', 'Qc', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    mom6 = lambda hyp, mu, s2, W, yv, k: mom(hyp, mu, s2, W, ep_fraction, yv, k)
    d = float(np.ravel(ep_damping)[0]); a = float(ep_fraction)
    S = F.shape[0]; M = H.shape[0]; T = yall.size
    MS = np.zeros((S, T)); PS = np.zeros((T, S, S))
    ttau = np.zeros((M, T)); tnu = np.zeros((M, T)); lZ = np.zeros(T); R = np.zeros((M, T))
    A, Q = ssm.lti_disc(F, L, Qc, 1.0)
    counters = {}
    mdM = np.zeros(ep_itts); mdP = np.zeros(ep_itts); nll = np.zeros(ep_itts)
    for itt in range(1, ep_itts + 1):
        m = np.zeros(S); P = Pinf.copy()
        maxDiffP = 0.0; maxDiffM = 0.0
        PSP = PS.copy(); MSP = MS.copy()
        for k in range(T):
            if k > 0:
                m = A @ m; P = A @ P @ A.T + Q
            if not np.isnan(yall[k]):
                fmu = H @ m; W = P @ H.T; HPH = np.diag(H @ P @ H.T).copy()
                if itt == 1 or k == T - 1:
                    lZ[k], dlZ, d2lZ = mom6(lik_param, fmu, HPH, Wnmf, yall, k)
                    ttau[:, k], tnu[:, k] = _refresh(ttau[:, k], tnu[:, k], fmu, HPH, dlZ, d2lZ, d, a)
                    with np.errstate(all='ignore'):
                        R[:, k] = 1.0 / ttau[:, k]
                ttau[:, k] = matlab_max0(ttau[:, k])
                m, P = kalman_update_legacy(m, P, H, W, HPH, fmu, ttau[:, k], tnu[:, k])   # the same two branches (:197-208)
            MS[:, k] = m; PS[k] = P
        MF = MS.copy(); PF = PS.copy()
        for k in range(T - 2, -1, -1):
            m, P = rts_step(A, Q, PS[k], MS[:, k], m, P, counters)
            MS[:, k] = m; PS[k] = P
            if itt < ep_itts and not np.isnan(yall[k]):
                mm = H @ m; vm = np.diag(H @ P @ H.T).copy()
                with np.errstate(all='ignore'):
                    v_cav = 1.0 / (1.0 / vm - a * ttau[:, k])
                    m_cav = v_cav * (mm / vm - a * tnu[:, k])
                upd = v_cav > 0
                _, dlZ_, d2lZ_ = mom6(lik_param, m_cav, v_cav, Wnmf, yall, k)
                ttau[:, k], tnu[:, k] = _refresh(ttau[:, k], tnu[:, k], m_cav, v_cav, dlZ_, d2lZ_, d, a, upd)
                with np.errstate(all='ignore'):
                    R[:, k] = 1.0 / ttau[:, k]
                maxDiffM = max(maxDiffM, np.max(np.abs(H @ MSP[:, k] - H @ m)))
                maxDiffP = max(maxDiffP, np.max(np.abs(H @ PSP[k] @ H.T - H @ P @ H.T)))
        mdM[itt - 1] = maxDiffM; mdP[itt - 1] = maxDiffP; nll[itt - 1] = -np.sum(lZ)
    Eft = H @ MS
    Varft = np.stack([np.diag(H @ PS[k] @ H.T) for k in range(T)], axis=1)
    return dict(Eft=Eft, Varft=Varft, MS=MS, PS=PS, ttau=ttau, tnu=tnu, R=R, lZ=lZ, MF=MF, PF=PF,
                maxDiffM=mdM, maxDiffP=mdP, nll=nll, counters=counters, A=A, Q=Q)


def run_ihgp(model, yall, mom, ep_fraction, ep_damping, ep_itts, tables=None):
    """ihgp_ep_mods_nmf_mixture.m:129-472."""
    F, L, Qc, H, Pinf, Wnmf, lik_param = (model[k] for k in ('F', 'L', 'Qc', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    mom6 = lambda hyp, mu, s2, W, yv, k: mom(hyp, mu, s2, W, ep_fraction, yv, k)
    d = float(np.ravel(ep_damping)[0]); a = float(ep_fraction)
    A, Q = ssm.lti_disc(F, L, Qc, 1.0)
    Q = (Q + Q.T) / 2                                                  # :136
    S = A.shape[0]; M = H.shape[0]; T = yall.size
    ilist, r, PPlist, PGlist = tables if tables is not None else build_tables(dict(A=A, Q=Q, H=H))
    blocks = [slice(ilist[n], ilist[n + 1]) for n in range(M)]
    bs = [ilist[n + 1] - ilist[n] for n in range(M)]
    m = np.zeros(S); P = Pinf.copy()                                   # set once, before the sweeps (:238-239)
    MS = np.zeros((S, T)); ttau = np.zeros((M, T)); tnu = np.zeros((M, T))
    R = np.zeros((M, T)); ys = np.full((M, T), np.nan)
    mdM = np.zeros(ep_itts); mdP = np.zeros(ep_itts); nll = np.zeros(ep_itts)
    for itt in range(1, ep_itts + 1):
        lZ = 0.0; maxDiffM = 0.0
        PSP = P.copy(); MSP = MS.copy()
        for k in range(T):
            if k > 0:
                PP = np.zeros((S, S))
                for n in range(M):
                    ind = nearest_index(r, R[n, k - 1])
                    PP[blocks[n], blocks[n]] = PPlist[n][ind].reshape((bs[n], bs[n]), order='F')
            else:
                PP = Pinf
            fmu = H @ A @ m; W = PP @ H.T; HPH = np.diag(H @ W).copy()
            if itt == 1 or k == T - 1:
                lZ_k, dlZ, d2lZ = mom6(lik_param, fmu, HPH, Wnmf, yall, k)
                lZ = lZ + lZ_k
                ttau[:, k], tnu[:, k] = _refresh(ttau[:, k], tnu[:, k], fmu, HPH, dlZ, d2lZ, d, a)
                with np.errstate(all='ignore'):
                    R[:, k] = 1.0 / ttau[:, k]
            ttau[:, k] = matlab_max0(ttau[:, k])
            with np.errstate(all='ignore'):
                ys[:, k] = tnu[:, k] / ttau[:, k]
            for n in range(M):
                ii = blocks[n]
                if ttau[n, k] == 0:
                    R[n, k] = np.inf
                    m[ii] = A[ii, ii] @ m[ii]
                    P[ii, ii] = PP[ii, ii]
                else:
                    K = W[ii, n] / (HPH[n] + R[n, k])
                    AKHA = A[ii, ii] - np.outer(K, H[n, ii]) @ A[ii, ii]
                    m[ii] = AKHA @ m[ii] + K * ys[n, k]
                    P[ii, ii] = PP[ii, ii] - np.outer(K, K) * R[n, k]
            MS[:, k] = m
        lZ_filter = lZ
        MF = MS.copy()
        P = np.zeros((S, S)); G = np.zeros((S, S))
        for k in range(T - 2, -1, -1):
            for n in range(M):
                ind = nearest_index(r, R[n, k])
                if np.isinf(R[n, k]):
                    ind = r.size - 1
                b = bs[n]; PG = PGlist[n][ind]
                P[blocks[n], blocks[n]] = PG[:b * b].reshape((b, b), order='F')
                G[blocks[n], blocks[n]] = PG[b * b:].reshape((b, b), order='F')
            m = MS[:, k] + G @ (m - A @ MS[:, k])
            MS[:, k] = m
            if itt < ep_itts and not np.isnan(yall[k]):
                mm = H @ m; vm = np.diag(H @ P @ H.T).copy()
                with np.errstate(all='ignore'):
                    v_cav = 1.0 / (1.0 / vm - a * ttau[:, k])
                    m_cav = v_cav * (mm / vm - a * tnu[:, k])
                upd = v_cav > 0
                lZ_k, dlZ_, d2lZ_ = mom6(lik_param, m_cav, v_cav, Wnmf, yall, k)
                if itt > 1:
                    lZ = lZ + lZ_k
                ttau[:, k], tnu[:, k] = _refresh(ttau[:, k], tnu[:, k], m_cav, v_cav, dlZ_, d2lZ_, d, a, upd)
                with np.errstate(all='ignore'):
                    R[upd, k] = 1.0 / ttau[upd, k]
                maxDiffM = max(maxDiffM, np.max(np.abs(H @ MSP[:, k] - H @ m)))
        maxDiffP = np.max(np.abs(H @ PSP @ H.T - H @ P @ H.T))
        mdM[itt - 1] = maxDiffM; mdP[itt - 1] = maxDiffP; nll[itt - 1] = -lZ
    Eft = H @ MS
    Varft = np.repeat(np.diag(H @ P @ H.T)[:, None], T, axis=1)       # no abs() here (:510)
    return dict(Eft=Eft, Varft=Varft, MS=MS, ttau=ttau, tnu=tnu, R=R, lZ=lZ_filter, MF=MF, maxDiffM=mdM, maxDiffP=mdP,
                nll=nll, r=r, PPlist=PPlist, PGlist=PGlist, ilist=ilist, A=A, Q=Q)


def _outputs(res, return_ind, nargout, ihgp):
    Eft = res['Eft'][:, return_ind]
    Varft = res['Varft'][:, :len(return_ind)] if ihgp else res['Varft'][:, return_ind]
    if nargout <= 1:
        return Eft
    if nargout <= 3:
        return Eft, Varft
    with np.errstate(all='ignore'):
        lb = Eft - 1.96 * np.sqrt(Varft); ub = Eft + 1.96 * np.sqrt(Varft)
    return (Eft, Varft, None, lb, ub, res) if nargout > 5 else (Eft, Varft, None, lb, ub)


def gf_ep_mods_nmf_mixture(w, x, y, ss, mom, xt, kernel1, kernel2, J, ep_fraction=0.5, ep_damping=0.1, ep_itts=30, nargout=6):
    """gf_ep_mods_nmf_mixture.m:1."""
    yall, return_ind = merge_inputs(x, y, xt)
    if xt is None or np.size(xt) == 0:
        raise RuntimeError('this mixture script is not for training')   # :376
    res = run_gf(stack_models(w, kernel1, kernel2, J), yall, mom, ep_fraction, ep_damping, ep_itts)
    return _outputs(res, return_ind, nargout, False)


def ihgp_ep_mods_nmf_mixture(w, x, y, ss, mom, xt, kernel1, kernel2, J, ep_fraction=0.5, ep_damping=0.1, ep_itts=30, nargout=6):
    """ihgp_ep_mods_nmf_mixture.m:1."""
    yall, return_ind = merge_inputs(x, y, xt)
    if xt is None or np.size(xt) == 0:
        raise RuntimeError('this mixture script is not for training')   # :547
    res = run_ihgp(stack_models(w, kernel1, kernel2, J), yall, mom, ep_fraction, ep_damping, ep_itts)
    return _outputs(res, return_ind, nargout, True)
