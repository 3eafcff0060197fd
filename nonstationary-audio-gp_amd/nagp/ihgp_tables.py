"""Host-side steady-state look-up tables of the IHGP path (once per call, before the C ABI).

Replaces matlab/ihgp_ep_modulator_nmf.m:99-141 (forward DARE per channel on ro=logspace(-2,4,32),
linear interpolation to r=logspace(-2,4,200) -- apxGrid('interp') on a non-equispaced grid is
linear, SURVEY C-12) and :148-191 (steady-state smoother gain G and covariance PS2 per grid point).
MATLAB's Control-System-Toolbox `dare` maps to scipy.linalg.solve_discrete_are /
solve_discrete_lyapunov.
"""
import numpy as np
import scipy.linalg as sla


def _interp_rows(ro, tab, r):
    """piece-wise linear in r between bracketing ro knots, clamped at the ends (apxGrid.m:555-565)."""
    out = np.empty((r.size, tab.shape[1]))
    for c in range(tab.shape[1]):
        out[:, c] = np.interp(r, ro, tab[:, c])
    return out


def build_tables(A, Q, offsets, h_val, n_grid=200, n_knots=32):
    """Returns (r, PPlist, pp_offsets, PGlist, pg_offsets) in the flat layout of nagp_ihgp_tables."""
    M = len(h_val)
    r = np.logspace(-2, 4, n_grid)
    pp_parts, pg_parts, pp_off, pg_off = [], [], [], []
    npp = npg = 0
    for n in range(M):
        o, e = int(offsets[n]), int(offsets[n + 1]); b = e - o
        Ab = np.ascontiguousarray(A[o:e, o:e]); Qb = np.ascontiguousarray(Q[o:e, o:e])
        Hn = np.zeros((1, b)); Hn[0, 0] = h_val[n]
        ro = np.logspace(-2, 4, n_knots)
        good, PPs, PGs = [], [], []
        for j, rj in enumerate(ro):
            try:
                PP = sla.solve_discrete_are(Ab.T, Hn.T, Qb, np.array([[rj]]))
            except Exception:
                continue                                   # :118-126: failed grid points are dropped
            S = float((Hn @ PP @ Hn.T)[0, 0]) + rj
            K = PP @ Hn.T / S
            P = PP - rj * (K @ K.T)
            PSkp = Ab @ P @ Ab.T + Qb
            Lc = np.linalg.cholesky(np.tril(PSkp) + np.tril(PSkp, -1).T)
            G = sla.cho_solve((Lc, True), (P @ Ab.T).T).T   # P*A'/L'/L
            QQ = P - G @ PP @ G.T; QQ = (QQ + QQ.T) / 2
            lam, V = np.linalg.eigh(QQ); pos = lam > 0
            QQ = (V[:, pos] * lam[pos]) @ V[:, pos].T
            try:
                PS2 = sla.solve_discrete_lyapunov(G, QQ)    # dare(G',0*G,QQ)
            except Exception:
                continue
            good.append(j); PPs.append(PP.flatten(order='F'))
            PGs.append(np.concatenate([PS2.flatten(order='F'), G.flatten(order='F')]))
        ro = ro[good]
        pp = _interp_rows(ro, np.array(PPs), r)
        pg = _interp_rows(ro, np.array(PGs), r)
        pp_off.append(npp); pg_off.append(npg)
        pp_parts.append(pp.ravel()); pg_parts.append(pg.ravel())
        npp += pp.size; npg += pg.size
    return (r, np.concatenate(pp_parts), np.array(pp_off, dtype=np.int64),
            np.concatenate(pg_parts), np.array(pg_off, dtype=np.int64))
