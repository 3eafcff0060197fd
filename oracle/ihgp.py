"""ORACLE (test infrastructure only -- never imported by the product path).

Line-by-line CPU restatement of the reference's infinite-horizon (steady-state) Power-EP
filter/smoother.  PARITY UNPINNED (no reference fixtures, no MATLAB); self-pinned by
tests/test_oracle_selfpins.py (DARE residuals, IHGP filter == full KF after burn-in).

Follows (file:line under /root/reference/matlab):
  ihgp_ep_modulator_nmf.m:59-97 (set-up), :99-141 (forward DARE tables), :148-191 (smoother
      tables), :197-454 (EP sweeps), :484-524 (outputs)
  ihgp_ep_modulator_nmf_constraints.m (differences: unpacking :76-111, R initialised to 0 :243,
      no abs() on Varft :517-518)
  apxGrid.m:448-498, 555-565, 695-707  ('interp' on a non-equispaced grid = linear weights, C-12)
MATLAB `dare` (Control System Toolbox, closed source) -> scipy.linalg.solve_discrete_are /
solve_discrete_lyapunov.
Quirks reproduced: C-3, C-4, C-10, C-12, C-18, C-21, C-22.
"""
import numpy as np
import scipy.linalg as sla
from . import ss as ssm
from .gf_ep import merge_inputs, _damp, matlab_max0, site_update_filter, ep_site_update_smoother, assemble


def neqinterp_matrix(s, t):
    """apxGrid.m:555-565 (neqinterp) assembled as the dense (nt x ns) matrix interpgrid returns."""
    s = np.asarray(s, float).ravel(); t = np.asarray(t, float).ravel()
    ns = s.size
    order = np.argsort(s, kind='stable'); ss_ = s[order]
    edges = np.concatenate([[-np.inf], ss_[1:-1], [np.inf]])
    # histc: bin i  <=>  edges(i) <= t < edges(i+1); the last edge catches t == inf only
    ii = np.searchsorted(edges, t, side='right') - 1
    ii = np.clip(ii, 0, ns - 2)
    d0 = t - ss_[ii]; d1 = ss_[ii + 1] - t
    d0 = np.where(d0 < 0, 0.0, d0); d1 = np.where(d1 < 0, 0.0, d1)
    U = np.zeros((t.size, ns))
    rows = np.arange(t.size)
    U[rows, order[ii]] += d1 / (d1 + d0)
    U[rows, order[ii + 1]] += d0 / (d1 + d0)
    return U


def forward_tables(A, Q, H, ilist, r_grid_n=200, ro_n=32):
    """ihgp_ep_modulator_nmf.m:107-134."""
    M = H.shape[0]
    r = np.logspace(-2, 4, r_grid_n)
    PPlist = []; PPlisto = []; ro_list = []
    for n in range(M):
        ii = slice(ilist[n], ilist[n + 1])
        Aii = A[ii, ii]; Qii = Q[ii, ii]; Hn = H[n:n + 1, ii]
        ro = np.logspace(-2, 4, ro_n)
        rows = []
        keep = []
        for j in range(ro.size):
            try:
                PP = sla.solve_discrete_are(Aii.T, Hn.T, Qii, np.array([[ro[j]]]))
                rows.append(PP.flatten(order='F')); keep.append(j)
            except Exception:                                   # :118-126 drop failed grid points
                pass
        ro = ro[keep]
        PPo = np.array(rows)
        U = neqinterp_matrix(ro, r)
        PPlist.append(U @ PPo); PPlisto.append(PPo); ro_list.append(ro)
    return r, PPlist, PPlisto, ro_list


def smoother_tables(A, Q, H, ilist, r, PPlisto, ro_list):
    """ihgp_ep_modulator_nmf.m:153-191 -> PGlist{n} rows = [PS2(:)' G(:)']."""
    M = H.shape[0]
    PGlist = []
    for n in range(M):
        ii = slice(ilist[n], ilist[n + 1])
        Aii = A[ii, ii]; Qii = Q[ii, ii]; Hn = H[n:n + 1, ii]
        b = Aii.shape[0]
        ro = ro_list[n].copy()
        rows = []; keep = []
        for j in range(ro.size):
            PP = PPlisto[n][j].reshape((b, b), order='F')
            Sx = (Hn @ PP @ Hn.T)[0, 0] + ro[j]
            K = PP @ Hn.T / Sx
            P = PP - K * ro[j] @ K.T
            L = np.linalg.cholesky(_lower_sym(Aii @ P @ Aii.T + Qii))   # :165 (non-PD branch is broken, C-14)
            B = P @ Aii.T
            G = np.linalg.solve(L.T, np.linalg.solve(L, B.T)).T           # P*A'/L'/L
            QQ = P - G @ PP @ G.T; QQ = (QQ + QQ.T) / 2
            DD, V = np.linalg.eigh(QQ); ind = DD > 0
            QQ = (V[:, ind] * DD[ind][None, :]) @ V[:, ind].T
            try:
                PS2 = sla.solve_discrete_lyapunov(G, QQ)               # dare(G',0*G,QQ): X = G X G' + QQ
                keep.append(j)
            except Exception:
                continue
            rows.append(np.concatenate([PS2.flatten(order='F'), G.flatten(order='F')]))
        ro = ro[keep]
        U = neqinterp_matrix(ro, r)
        PGlist.append(U @ np.array(rows))
    return PGlist


def _lower_sym(X):
    Ls = np.tril(X)
    return Ls + np.tril(Ls, -1).T


def nearest_index(r, Rv):
    """[~,ind] = min(abs(r-Rv)): first minimiser; all-NaN/Inf distances -> index 0 (C-4)."""
    with np.errstate(all='ignore'):
        d = np.abs(r - Rv)
    if np.all(np.isnan(d)):
        return 0
    return int(np.nanargmin(d))


def build_tables(model):
    """The DARE set-up of ihgp_ep_modulator_nmf.m:99-191 (host side, once per call)."""
    A, Q, H = model['A'], model['Q'], model['H']
    ilist = ssm.block_starts(H)
    r, PPlist, PPlisto, ro_list = forward_tables(A, Q, H, ilist)
    PGlist = smoother_tables(A, Q, H, ilist, r, PPlisto, ro_list)
    return ilist, r, PPlist, PGlist


def run_predict(model, yall, mom, ep_fraction, ep_damping, ep_itts, constraints_variant=False, verbose=False, tables=None):
    """ihgp_ep_modulator_nmf.m:99-524 on an assembled (balanced, Q-symmetrised) model."""
    A, Q, H, Pinf, Wnmf, lik_param = (model[k] for k in ('A', 'Q', 'H', 'Pinf', 'Wnmf', 'lik_param'))
    S = A.shape[0]; M = H.shape[0]; T = yall.size
    ilist, r, PPlist, PGlist = tables if tables is not None else build_tables(model)
    blocks = [slice(ilist[n], ilist[n + 1]) for n in range(M)]
    bs = [ilist[n + 1] - ilist[n] for n in range(M)]

    m = np.zeros(S); P = Pinf.copy()
    MS = np.zeros((S, T)); ttau = np.zeros((M, T)); tnu = np.zeros((M, T))
    if constraints_variant:
        R = np.zeros((M, T))                                          # _constraints.m:243
    else:
        R = np.exp(float(np.ravel(lik_param)[0])) * np.ones((M, T))   # :209
    nlZ = np.zeros(ep_itts); ys = np.full((M, T), np.nan)
    mdM = np.zeros(ep_itts); mdP = np.zeros(ep_itts)
    damp = _damp(ep_damping, ep_itts); ep_damp = damp[0]
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
                lZ_k, dlZ, d2lZ = mom(lik_param, fmu, HPH, Wnmf, 1.0, yall, k)
                lZ = lZ + lZ_k
                ttau[:, k], tnu[:, k] = site_update_filter(ttau[:, k], tnu[:, k], fmu, HPH, dlZ, d2lZ, ep_damp)
                with np.errstate(all='ignore'):
                    R[:, k] = 1.0 / ttau[:, k]                        # before the clamp (:269)
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
        if itt == 1:
            nlZ[0] = -lZ
        MF = MS.copy()
        P = np.zeros((S, S)); G = np.zeros((S, S))
        if itt < ep_itts:
            ep_damp = damp[itt]
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
                lZ_k, ttau[:, k], tnu[:, k], upd = ep_site_update_smoother(
                    ttau[:, k], tnu[:, k], mm, vm, mom, lik_param, Wnmf, ep_fraction, ep_damp, yall, k)
                if itt > 1:
                    lZ = lZ + lZ_k
                with np.errstate(all='ignore'):
                    R[upd, k] = 1.0 / ttau[upd, k]                    # no clamp here (:427-434)
            maxDiffM = max(maxDiffM, np.max(np.abs(H @ MSP[:, k] - H @ m)))
        maxDiffP = np.max(np.abs(H @ PSP @ H.T - H @ P @ H.T))
        if itt < ep_itts:
            nlZ[itt] = -lZ
        mdM[itt - 1] = maxDiffM; mdP[itt - 1] = maxDiffP
        if verbose:
            print('%02i - max diff in m: %.6g - max diff in P: %.6g - nll: %.6g' % (itt, maxDiffM, maxDiffP, nlZ[itt - 1]))
    Eft = H @ MS
    Varft = np.repeat(np.diag(H @ P @ H.T)[:, None], T, axis=1)
    if not constraints_variant:
        Varft = np.abs(Varft)                                         # :493-496 (always taken, C-10)
    return dict(Eft=Eft, Varft=Varft, MS=MS, ttau=ttau, tnu=tnu, R=R, nlZ=nlZ, maxDiffM=mdM, maxDiffP=mdP,
                MF=MF, r=r, PPlist=PPlist, PGlist=PGlist, ilist=ilist, Plast=P)


def ihgp_ep_modulator_nmf(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N,
                          ep_fraction=0.5, ep_damping=None, ep_itts=30, nargout=6):
    """ihgp_ep_modulator_nmf.m:1 (predict mode only; nlml mode is broken in the reference, C-11)."""
    yall, return_ind = merge_inputs(x, y, xt)
    lik_param, param1, param2, Wnmf = ssm.unpack_log(w, num_lik_params, D, N)
    model = assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance=True, symmetrize_Q=True)
    if xt is None or np.size(xt) == 0:
        raise NotImplementedError('IHGP nlml mode is broken in the reference (SURVEY C-11)')
    res = run_predict(model, yall, mom, ep_fraction, ep_damping, ep_itts, constraints_variant=False)
    return _ihgp_outputs(res, return_ind, nargout)


def ihgp_ep_modulator_nmf_constraints(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N,
                                      ep_fraction, ep_damping, ep_itts, constraints, w_fixed, tune_hypers, nargout=6):
    """ihgp_ep_modulator_nmf_constraints.m:1-2."""
    yall, return_ind = merge_inputs(x, y, xt)
    lik_param, param1, param2, Wnmf = ssm.unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
    model = assemble(lik_param, param1, param2, Wnmf, kernel1, kernel2, balance=True, symmetrize_Q=True)
    if xt is None or np.size(xt) == 0:
        raise NotImplementedError('IHGP nlml mode is broken in the reference (SURVEY C-11)')
    res = run_predict(model, yall, mom, ep_fraction, ep_damping, ep_itts, constraints_variant=True)
    return _ihgp_outputs(res, return_ind, nargout)


def _ihgp_outputs(res, return_ind, nargout):
    Eft = res['Eft'][:, return_ind]; Varft = res['Varft'][:, :len(return_ind)]
    if nargout <= 1:
        return Eft
    if nargout <= 3:
        return Eft, Varft
    lb = Eft - 1.96 * np.sqrt(Varft); ub = Eft + 1.96 * np.sqrt(Varft)
    return Eft, Varft, None, lb, ub, res
