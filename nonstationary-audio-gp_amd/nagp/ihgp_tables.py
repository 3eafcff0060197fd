"""Host-side steady-state look-up tables of the IHGP path (once per call, before the C ABI).

Replaces matlab/ihgp_ep_modulator_nmf.m:99-141 (forward DARE per channel on ro=logspace(-2,4,32),
linear interpolation to r=logspace(-2,4,200) -- apxGrid('interp') on a non-equispaced grid is
linear, SURVEY C-12) and :148-191 (steady-state smoother gain G and covariance PS2 per grid point).
MATLAB's Control-System-Toolbox `dare` is stood in for by a structure-preserving doubling iteration that solves all
(channel, grid point) Riccati equations of one block size in one batched NumPy sweep (1 216 equations at cfg3: 0.03 s
instead of 1.0 s with one scipy.linalg.solve_discrete_are call each); any equation the doubling does not converge on
falls back to SciPy's QZ solver, and a point that fails there is dropped as the reference drops it (:118-126).
"""
import numpy as np
import scipy.linalg as sla


def _interp_rows(ro, tab, r):
    """piece-wise linear in r between bracketing ro knots, clamped at the ends (apxGrid.m:555-565)."""
    out = np.empty((r.size, tab.shape[1]))
    for c in range(tab.shape[1]):
        out[:, c] = np.interp(r, ro, tab[:, c])
    return out


def _dare_batch(At, h, Q, rr, iters=60, tol=1e-15):
    """X = dare(A', H', Q, r) for a batch: At (B,b,b) = A', h (B,b) = H (row vector), Q (B,b,b), rr (B,).
    Doubling (SDA):  W = I + G H ; A+ = A W^-1 A ; G+ = G + A W^-1 G A' ; H+ = H + A' H W^-1 A  ->  H_k -> X."""
    B, b, _ = At.shape
    A = At.copy()                                   # the `a` argument of dare
    G = h[:, :, None] * h[:, None, :] / rr[:, None, None]
    H = Q.copy()
    eye = np.broadcast_to(np.eye(b), (B, b, b))
    ok = np.ones(B, bool)
    for _ in range(iters):
        W = eye + G @ H
        WiA = np.linalg.solve(W, A)
        WiG = np.linalg.solve(W, G)
        An = A @ WiA
        Gn = G + A @ WiG @ np.swapaxes(A, 1, 2)
        Hn = H + np.swapaxes(A, 1, 2) @ H @ WiA
        d = np.max(np.abs(Hn - H), axis=(1, 2)) / np.maximum(np.max(np.abs(Hn), axis=(1, 2)), 1e-300)
        A, G, H = An, Gn, Hn
        if np.all(d < tol):
            break
    H = (H + np.swapaxes(H, 1, 2)) / 2
    ok &= np.isfinite(H).all(axis=(1, 2)) & (d < 1e-10)
    return H, ok


def build_tables(A, Q, offsets, h_val, n_grid=200, n_knots=32):
    """Returns (r, PPlist, pp_offsets, PGlist, pg_offsets) in the flat layout of nagp_ihgp_tables."""
    M = len(h_val)
    r = np.logspace(-2, 4, n_grid); ro = np.logspace(-2, 4, n_knots)
    sizes = [int(offsets[n + 1]) - int(offsets[n]) for n in range(M)]
    PPs = [None] * M; PGs = [None] * M; good = [None] * M
    for b in sorted(set(sizes)):
        idx = [n for n in range(M) if sizes[n] == b]
        nb = len(idx); B = nb * n_knots
        Ab = np.stack([A[int(offsets[n]):int(offsets[n]) + b, int(offsets[n]):int(offsets[n]) + b] for n in idx])
        Qb = np.stack([Q[int(offsets[n]):int(offsets[n]) + b, int(offsets[n]):int(offsets[n]) + b] for n in idx])
        hv = np.array([h_val[n] for n in idx])
        Ar = np.repeat(Ab, n_knots, axis=0); Qr = np.repeat(Qb, n_knots, axis=0)
        hr = np.zeros((B, b)); hr[:, 0] = np.repeat(hv, n_knots)
        rr = np.tile(ro, nb)
        PP, ok = _dare_batch(np.swapaxes(Ar, 1, 2), hr, Qr, rr)
        for q in np.where(~ok)[0]:                                      # rare: fall back to the QZ solver
            try:
                PP[q] = sla.solve_discrete_are(Ar[q].T, hr[q][:, None], Qr[q], np.array([[rr[q]]])); ok[q] = True
            except Exception:
                pass
        S = hr[:, 0] ** 2 * PP[:, 0, 0] + rr                              # H PP H' + r
        K = PP[:, :, 0] * hr[:, 0][:, None] / S[:, None]                  # PP H' / S
        P = PP - rr[:, None, None] * (K[:, :, None] * K[:, None, :])      # :163  (C-23: K r K', not K S K')
        PSkp = Ar @ P @ np.swapaxes(Ar, 1, 2) + Qr
        PSkp = np.tril(PSkp) + np.swapaxes(np.tril(PSkp, -1), 1, 2)       # chol(.,'lower') reads the lower triangle
        PAt = P @ np.swapaxes(Ar, 1, 2)
        with np.errstate(all='ignore'):
            try:
                G = np.swapaxes(np.linalg.solve(PSkp, np.swapaxes(PAt, 1, 2)), 1, 2)    # P*A'/L'/L
            except np.linalg.LinAlgError:
                G = np.full_like(P, np.nan)
        QQ = P - G @ PP @ np.swapaxes(G, 1, 2); QQ = (QQ + np.swapaxes(QQ, 1, 2)) / 2
        fin = np.isfinite(QQ).all(axis=(1, 2))
        QQ[~fin] = 0.0
        lam, V = np.linalg.eigh(QQ)                                        # cholcov-style projection on the PSD cone (:169-175)
        lam = np.where(lam > 0, lam, 0.0)
        QQ = (V * lam[:, None, :]) @ np.swapaxes(V, 1, 2)
        # PS2 = dare(G',0,QQ): the Stein equation X = G X G' + QQ as a b^2 x b^2 linear system
        Gs = np.where(np.isfinite(G), G, 0.0)
        KK = np.einsum('qij,qkl->qikjl', Gs, Gs).reshape(B, b * b, b * b)
        rhs = QQ.reshape(B, b * b, 1)
        try:
            PS2 = np.linalg.solve(np.broadcast_to(np.eye(b * b), (B, b * b, b * b)) - KK, rhs).reshape(B, b, b)
        except np.linalg.LinAlgError:
            PS2 = np.full_like(P, np.nan)
        ok &= fin & np.isfinite(G).all(axis=(1, 2)) & np.isfinite(PS2).all(axis=(1, 2))
        ok &= np.linalg.eigvalsh(PSkp).min(axis=1) > 0                     # chol of PSkp must exist
        for ii, n in enumerate(idx):
            sl = slice(ii * n_knots, (ii + 1) * n_knots)
            g = np.where(ok[sl])[0]
            good[n] = g
            # MATLAB column-major flattening PP(:)'
            PPs[n] = np.stack([PP[sl][j].flatten(order='F') for j in g])
            PGs[n] = np.stack([np.concatenate([PS2[sl][j].flatten(order='F'), G[sl][j].flatten(order='F')]) for j in g])
    pp_parts, pg_parts, pp_off, pg_off = [], [], [], []
    npp = npg = 0
    for n in range(M):
        pp = _interp_rows(ro[good[n]], PPs[n], r)
        pg = _interp_rows(ro[good[n]], PGs[n], r)
        pp_off.append(npp); pg_off.append(npg)
        pp_parts.append(pp.ravel()); pg_parts.append(pg.ravel())
        npp += pp.size; npg += pg.size
    return (r, np.concatenate(pp_parts), np.array(pp_off, dtype=np.int64),
            np.concatenate(pg_parts), np.array(pg_off, dtype=np.int64))
