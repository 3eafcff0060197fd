"""Host-side mirror of the reference's operator interface for the hot path.

Same names, positional argument lists and return conventions as the MATLAB functions
(matlab/gf_ep_modulator.m:1, gf_ep_modulator_nmf.m:1, gf_ep_modulator_nmf_constraints.m:1-2,
ihgp_ep_modulator_nmf.m:1, ihgp_ep_modulator_nmf_constraints.m:1-2, gf_giekf_modulator_nmf.m:1-2,
gf_giekf_modulator_nmf_constraints.m:1-2); `nargout` stands in for MATLAB's nargout.  Everything up
to the discrete-time model runs here on the host (as it would in the .m wrapper); the per-time-step
loops run in libnagp.so on the GPU.  There is no CPU fallback.

The two function handles of the reference become:
  ss  -- any callable ss(x, p1, p2, k1, k2) -> (F, L, Qc, H, Pinf) with block-diagonal F (one block per
         row of H); `nagp.ss_modulators_nmf` / `nagp.ss_modulators` wrapped by SSHandle are the shipped ones
  mom -- a `Mom` descriptor (the wrapper inspects it the way the MEX wrapper inspects
         functions(mom).workspace{1}: likfunc, link, p_cubature or precomputed wn/xn_unscaled)
"""
import ctypes as C

import numpy as np
import scipy.linalg as sla

from . import _lib as L
from . import cubature, ihgp_tables
from . import ss as ssm


class Mom:
    """Descriptor of the reference's `mom` closure (demo_toy_modulators_nmf.m:81, demo_toy_modulators.m:81,
    experiments/train_GTFNMF.m:149)."""
    _KINDS = {'likModulatorPower': L.LIK_POWER, 'likModulatorNMFPower': L.LIK_POWER_NMF,
              'likModulatorPreCalcwn': L.LIK_POWER_NMF_SQRT}

    def __init__(self, likfunc='likModulatorNMFPower', link='softplus', link_shift=0.0, p_cubature=9,
                 wn=None, xn_unscaled=None, quirks=True):
        if likfunc not in self._KINDS:
            raise ValueError('unknown likelihood %r (shipped: %s)' % (likfunc, ', '.join(self._KINDS)))
        if link not in ('softplus', 'exp'):
            raise ValueError("link must be 'softplus' (log(1+exp(g-shift))) or 'exp'")
        self.likfunc, self.kind = likfunc, self._KINDS[likfunc]
        self.link, self.link_shift, self.p, self.quirks = link, float(link_shift), p_cubature, quirks
        self.wn, self.xn_unscaled = wn, xn_unscaled
        if self.kind == L.LIK_POWER_NMF_SQRT and (wn is None or xn_unscaled is None):
            raise ValueError('likModulatorPreCalcwn needs precomputed wn, xn_unscaled')

    def tables(self, dim):
        if self.wn is not None:
            wn = np.asarray(self.wn, float).ravel(); xn = np.asarray(self.xn_unscaled, float)
            if xn.shape != (dim, wn.size):
                raise ValueError('xn_unscaled must be %d x %d' % (dim, wn.size))
            return wn, xn
        return cubature.sigma_points(self.p, dim, self.quirks)

    def __call__(self, hyp, mu, s2, *rest, device=0):
        """The callback itself, with the reference's arities: mom(hyp, mu, s2, ep_frac, yall, k) for likModulatorPower
        (demo_toy_modulators.m:81) and mom(hyp, mu, s2, nmfW, ep_frac, yall, k) for the NMF likelihoods
        (demo_toy_modulators_nmf.m:81); k is 0-based here.  Returns lZ, dlZ (M,), d2lZ (M,).  mu, s2 may also be
        M x n with yall[k] an n-vector: n independent evaluations in one launch (nagp_mom_eval)."""
        if self.kind == L.LIK_POWER:
            Wnmf = None; ep_frac, yall, k = rest
        else:
            Wnmf, ep_frac, yall, k = rest
        mu = np.asarray(mu, float); s2 = np.asarray(s2, float)
        single = mu.ndim == 1
        mu2 = L.f64(mu.reshape(mu.shape[0], -1)); s22 = L.f64(s2.reshape(s2.shape[0], -1))
        M, n = mu2.shape
        y = L.f64(np.atleast_1d(np.asarray(yall, float)[k]).ravel(), 'C')
        if y.size != n or s22.shape != mu2.shape:
            raise ValueError('mu, s2 must be M x n and yall[k] must hold n observations')
        if self.kind == L.LIK_POWER:
            D, N, dim, Wc = M // 2, 0, M // 2, None
        else:
            Wc = L.f64(Wnmf); D, N = Wc.shape; dim = N
        o, keep = make_opts(L.KIND_GF_EP, L.MODE_PREDICT, self, dim, ep_frac, None, 1, device=device)
        lZ = np.zeros(n); dl = np.zeros((M, n), order='F'); d2l = np.zeros((M, n), order='F')
        L.check(L.lib().nagp_mom_eval(C.byref(o), D, N, L.dptr(Wc), float(np.ravel(hyp)[0]), n, L.dptr(y), L.dptr(mu2), L.dptr(s22),
                                      L.dptr(lZ), L.dptr(dl), L.dptr(d2l)))
        return (float(lZ[0]), dl[:, 0].copy(), d2l[:, 0].copy()) if single else (lZ, dl, d2l)



class MeasModel:
    """The EKF measurement handles of gf_giekf_modulator_nmf.m:108-113 as one object:
    handle = @(x,p) funh(x,H,linkf,D,N,Wnmf), dhandle = @(x,p) funhd(x,H,...) -> pass `model.h` / `model.dh` where the
    reference passes `handle` / `dhandle` (h(x) = (H_z x)' W softplus(H_g x), gf_giekf_modulator_nmf_constraints.m:492-502)."""

    def __init__(self, H, Wnmf, D, N):
        H = np.asarray(H, float)
        if H.shape[0] != D + N or np.any((H != 0).sum(axis=1) != 1):
            raise ValueError('H must be (D+N) x S with exactly one non-zero per row')
        self.H, self.W, self.D, self.N = H, L.f64(Wnmf), int(D), int(N)
        self.col = np.argmax(H != 0, axis=1).astype(np.int32); self.val = H[np.arange(D + N), self.col].copy()

    def h(self, x, param=None):       # only a token: the update below evaluates the model on the device
        raise L.NagpError('MeasModel.h is evaluated on the GPU inside ekf_update1 / iekf_update1')

    dh = h


def iekf_update1(M, P, y, H, R, h, V=None, param=None, iters=5, device=0):
    """[M,P,K,MU,S] = iekf_update1(M,P,y,H,R,h,V,param,iters) (iekf_update1.m:48, :110-117), scalar y, with
    H = model.dh and h = model.h of a MeasModel.  Returns new arrays (inputs are not modified)."""
    mdl = getattr(H, '__self__', None)
    if not isinstance(mdl, MeasModel) or getattr(h, '__self__', None) is not mdl:
        raise ValueError('H and h must be the dh / h handles of one MeasModel (the model the reference drivers pass)')
    if V is not None:
        raise NotImplementedError('a noise-Jacobian V is never passed on the hot path (gf_giekf_modulator_nmf.m:161)')
    m = L.f64(np.array(M, float).ravel().copy(), 'C'); S = m.size
    Pd = L.f64(np.array(P, float).copy())
    if Pd.shape != (S, S) or mdl.H.shape[1] != S:
        raise ValueError('P must be S x S and H must have S columns')
    K = np.zeros(S); MU = C.c_double(0.0); Sx = C.c_double(0.0)
    L.check(L.lib().nagp_iekf_update1(S, mdl.D, mdl.N, mdl.col.ctypes.data_as(C.POINTER(C.c_int32)), L.dptr(L.f64(mdl.val, 'C')),
                                      L.dptr(mdl.W), float(np.ravel(R)[0]), float(np.ravel(y)[0]), int(iters), L.dptr(m), L.dptr(Pd),
                                      L.dptr(K), C.byref(MU), C.byref(Sx), int(device)))
    return m, Pd, K, MU.value, Sx.value


def ekf_update1(M, P, y, H, R, h, V=None, param=None, device=0):
    """[M,P,K,MU,S] = ekf_update1(M,P,y,H,R,h,V,param) (ekf_update1.m:48, :106-109): one linearisation."""
    return iekf_update1(M, P, y, H, R, h, V, param, 1, device)

class SSHandle:
    """ss = @(x,p1,p2,k1,k2) ss_modulators_nmf(p1,p2,k1,k2)  /  @(x,p,k1,k2) ss_modulators(p,k1,k2)."""

    def __init__(self, name='ss_modulators_nmf'):
        self.name = name

    def __call__(self, x, *a):
        if self.name == 'ss_modulators_nmf':
            return ssm.ss_modulators_nmf(*a)
        return ssm.ss_modulators(*a)


def _merge_inputs(x, y, xt):
    """gf_ep_modulator_nmf.m:58-66."""
    x = np.asarray(x, float).ravel(); y = np.asarray(y, float).ravel()
    xt = np.zeros(0) if xt is None else np.asarray(xt, float).ravel()
    xall = np.concatenate([x, xt]); yall = np.concatenate([y, np.full(xt.size, np.nan)])
    _, first, inv = np.unique(xall, return_index=True, return_inverse=True)
    return np.ascontiguousarray(yall[first]), inv[xall.size - xt.size:]


def _blocks_from_dense(F, L_, Qc, H, Pinf, D, N):
    """Recover the per-block form from what an `ss` handle returned (block starts = non-zero columns
    of H, ihgp_ep_modulator_nmf.m:104)."""
    H = np.asarray(H, float); F = np.asarray(F, float); Pinf = np.asarray(Pinf, float)
    LQL = np.asarray(L_, float) @ np.atleast_2d(Qc) @ np.asarray(L_, float).T
    M, S = H.shape
    starts = np.nonzero(np.abs(H).sum(axis=0))[0]
    if starts.size != M or not all(np.count_nonzero(H[n]) == 1 and H[n, starts[n]] != 0 for n in range(M)):
        raise ValueError('H must have exactly one non-zero per row, at the start of its block')
    off = np.concatenate([starts, [S]])
    mask = np.zeros((S, S), bool)
    Fs, Qs, Ps = [], [], []
    for n in range(M):
        o, e = off[n], off[n + 1]
        mask[o:e, o:e] = True
        Fs.append(F[o:e, o:e].copy()); Qs.append(LQL[o:e, o:e].copy()); Ps.append(Pinf[o:e, o:e].copy())
    if np.any(F[~mask] != 0) or np.any(LQL[~mask] != 0) or np.any(Pinf[~mask] != 0):
        raise ValueError('the state-space model is not block diagonal with the blocks of H')
    blk = ssm.BlockSS(Fs, Qs, Ps, D, N)
    blk.h_val = H[np.arange(M), starts].copy()
    return blk


def _damping(ep_damping, ep_itts):
    d = np.atleast_1d(np.asarray(ep_damping, float)).ravel()
    if d.size == 1:                        # SURVEY C-15: drivers pass scalars with ep_itts > 1
        d = np.full(max(int(ep_itts), 1), d[0])
    if d.size < ep_itts:
        raise ValueError('ep_damping has fewer than ep_itts entries')
    return np.ascontiguousarray(d[:ep_itts])


class _Problem:
    """Everything the C ABI needs for one call, with the numpy arrays kept alive."""

    def __init__(self, blk, Wnmf, lik_param, symmetrize_Q=False, stationary_Q=False, overrides=None):
        self.blk = blk
        A, Q, P = ssm.discretise(blk, symmetrize_Q, stationary_Q)
        if overrides:      # the C ABI takes A, Q, Pinf as plain arrays: a caller (or a test) may hand over its own discrete model
            A = overrides.get('A', A); Q = overrides.get('Q', Q); P = overrides.get('Pinf', P)
        self.A, self.Q, self.Pinf = L.f64(A), L.f64(Q), L.f64(P)
        self.h_val = L.f64(blk.h_val)
        self.offsets = np.ascontiguousarray(blk.offsets, dtype=np.int32)
        self.W = None if Wnmf is None else L.f64(Wnmf)
        self.model = L.Model(S=blk.S, M=blk.M, D=blk.D, N=blk.N,
                             block_offsets=self.offsets.ctypes.data_as(L.c_ip), A=L.dptr(self.A), Q=L.dptr(self.Q),
                             Pinf=L.dptr(self.Pinf), h_val=L.dptr(self.h_val), Wnmf=L.dptr(self.W),
                             lik_param=float(np.ravel(lik_param)[0]))


class _Outputs:
    def __init__(self, M, S, T, I, want_PS=False, want_MS=True, want_MF=False):
        self.MF = np.zeros((S, T), order='F') if want_MF else None
        self.Eft = np.zeros((M, T), order='F'); self.Varft = np.zeros((M, T), order='F')
        self.MS = np.zeros((S, T), order='F') if want_MS else None
        self.PS = np.zeros((S, S, T), order='F') if want_PS else None
        self.ttau = np.zeros((M, T), order='F'); self.tnu = np.zeros((M, T), order='F'); self.R = np.zeros((M, T), order='F')
        self.lZ = np.zeros(T); self.nlZ = np.zeros(I); self.maxDiffM = np.zeros(I); self.maxDiffP = np.zeros(I)
        self.counters = np.zeros(4, dtype=np.int64)
        self.c = L.Out(Eft=L.dptr(self.Eft), Varft=L.dptr(self.Varft), MS=L.dptr(self.MS), PS=L.dptr(self.PS),
                       ttau=L.dptr(self.ttau), tnu=L.dptr(self.tnu), R=L.dptr(self.R), lZ=L.dptr(self.lZ),
                       nlZ=L.dptr(self.nlZ), maxDiffM=L.dptr(self.maxDiffM), maxDiffP=L.dptr(self.maxDiffP),
                       counters=self.counters.ctypes.data_as(L.c_lp), MF=L.dptr(self.MF))

    def as_dict(self):
        d = dict(tnu=self.tnu, ttau=self.ttau, lZ=self.lZ, R=self.R, MS=self.MS, PS=self.PS, nlZ=self.nlZ,
                 maxDiffM=self.maxDiffM, maxDiffP=self.maxDiffP,
                 counters=dict(chol_retries=int(self.counters[0]), clamped=int(self.counters[1]),
                               nan_obs=int(self.counters[2]), not_pd=int(self.counters[3])),
                 Eft=self.Eft, Varft=self.Varft)
        return d


def make_opts(kind, mode, mom, dim, ep_fraction, damping, ep_itts, l_iter=0, predict_at_k1=0, flags=0, device=0, chunk=0):
    keep = {}
    o = L.Opts(kind=kind, mode=mode, ep_fraction=float(ep_fraction), ep_itts=int(ep_itts), l_iter=int(l_iter),
               predict_at_k1=int(predict_at_k1), flags=int(flags), device=int(device), chunk=int(chunk))
    if mom is not None and kind != L.KIND_GIEKF:
        wn, xn = mom.tables(dim)
        keep['wn'] = L.f64(wn, 'C'); keep['xn'] = L.f64(xn)
        o.lik_kind = mom.kind; o.link_kind = L.LINK_SOFTPLUS if mom.link == 'softplus' else L.LINK_EXP
        o.link_shift = mom.link_shift; o.n_pts = keep['wn'].size; o.cub_dim = dim
        o.wn = L.dptr(keep['wn']); o.xn_unscaled = L.dptr(keep['xn'])
    if damping is not None:
        keep['damp'] = L.f64(damping, 'C'); o.ep_damping = L.dptr(keep['damp'])
    return o, keep


def _returns(out, return_ind, nargout, predict_extra=None):
    Eft = out.Eft[:, return_ind]; Varft = out.Varft[:, return_ind]
    if nargout <= 1:
        return Eft
    if nargout == 2:
        return Eft, Varft
    lb = Eft - 1.96 * np.sqrt(Varft); ub = Eft + 1.96 * np.sqrt(Varft)
    res = (Eft, Varft, None, lb, ub, out.as_dict())
    return res[:max(nargout, 3)] if nargout < 6 else res


def _unpack_log(w, num_lik_params, D, N):
    w = np.asarray(w, float).ravel(); n0 = num_lik_params
    return (w[:n0], np.exp(w[n0:n0 + 3 * D]), np.exp(w[n0 + 3 * D:n0 + 3 * D + 2 * N]),
            np.exp(w[n0 + 3 * D + 2 * N:]).reshape((D, N), order='F'))


def _unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N):
    """gf_ep_modulator_nmf_constraints.m:75-110."""
    w = np.asarray(w, float).ravel(); wf = np.asarray(w_fixed, float).ravel(); cons = np.asarray(constraints, float)
    pos = {True: 0, False: 0}
    src = {True: w, False: wf}

    def take(tuned, cnt):
        tuned = bool(tuned); a = pos[tuned]; pos[tuned] = a + cnt
        return src[tuned][a:a + cnt]

    lik_param = take(tune_hypers[0], num_lik_params)
    groups = [ssm.sigmoid(take(tune_hypers[i], D if i <= 3 else N), cons[i - 1]) for i in range(1, 6)]
    t7 = bool(tune_hypers[6])
    Wnmf = ssm.sigmoid(src[t7][pos[t7]:], cons[5]).reshape((D, N), order='F')
    return lik_param, np.concatenate(groups[:3]), np.concatenate(groups[3:]), Wnmf


def _run_gf(blk, Wnmf, lik_param, yall, return_ind, mom, ep_fraction, ep_damping, ep_itts, predict, nargout,
            predict_at_k1=0, device=0, flags=0):
    prob = _Problem(blk, Wnmf, lik_param)
    dim = blk.D if mom.kind == L.LIK_POWER else blk.N
    damp = _damping(ep_damping, ep_itts)
    want_PS = predict and nargout >= 6
    opts, keep = make_opts(L.KIND_GF_EP, L.MODE_PREDICT if predict else L.MODE_NLML, mom, dim, ep_fraction, damp,
                           ep_itts, predict_at_k1=predict_at_k1, flags=flags | (L.FLAG_WANT_PS if want_PS else 0), device=device)
    out = _Outputs(blk.M, blk.S, yall.size, ep_itts, want_PS=want_PS)
    L.check(L.lib().nagp_ep_run(C.byref(prob.model), L.dptr(yall), yall.size, C.byref(opts), C.byref(out.c)))
    return out


def gf_ep_modulator_nmf(w, x, y, ss, mom, xt=None, kernel1='matern32', kernel2='matern52', num_lik_params=1, D=None, N=None,
                        ep_fraction=0.5, ep_damping=None, ep_itts=30, nargout=2, device=0):
    """[Eft,Varft,Covft,lb,ub,out] (xt given) or [e,eg] (xt empty) -- matlab/gf_ep_modulator_nmf.m:1."""
    yall, return_ind = _merge_inputs(x, y, xt)
    lik_param, p1, p2, Wnmf = _unpack_log(w, num_lik_params, D, N)
    blk = _blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N)           # balance OFF (:80 `if false`)
    predict = xt is not None and np.size(xt) > 0
    out = _run_gf(blk, Wnmf, lik_param, yall, return_ind, mom, ep_fraction, ep_damping, ep_itts, predict, nargout, device=device)
    if predict:
        return _returns(out, return_ind, nargout)
    return float(out.nlZ[0]), np.zeros(np.size(w))                             # eg is all zeros (:363, :531)


def gf_ep_modulator_nmf_constraints(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N, ep_fraction, ep_damping,
                                    ep_itts, constraints, w_fixed, tune_hypers, nargout=2, device=0):
    """matlab/gf_ep_modulator_nmf_constraints.m:1-2 (sigmoid-constrained parameters, balance ON :115)."""
    yall, return_ind = _merge_inputs(x, y, xt)
    lik_param, p1, p2, Wnmf = _unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
    blk = ssm.balance_blocks(_blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N))
    predict = xt is not None and np.size(xt) > 0
    out = _run_gf(blk, Wnmf, lik_param, yall, return_ind, mom, ep_fraction, ep_damping, ep_itts, predict, nargout, device=device)
    if predict:
        return _returns(out, return_ind, nargout)
    return float(out.nlZ[0]), np.zeros(np.size(w))


def gf_ep_modulator(w, x, y, ss, mom, xt=None, kernel1='matern32', kernel2='matern52', num_lik_params=1,
                    ep_fraction=0.5, ep_damping=None, ep_itts=30, nargout=2, device=0):
    """matlab/gf_ep_modulator.m:1 (one modulator per sub-band; balance ON :75; predicts at k=1 in
    predict mode :131-133)."""
    yall, return_ind = _merge_inputs(x, y, xt)
    w = np.asarray(w, float).ravel()
    lik_param = w[:num_lik_params]; param = np.exp(w[num_lik_params:])
    D = param.size // 5
    blk = ssm.balance_blocks(_blocks_from_dense(*ss(x, param, kernel1, kernel2), D, D))
    predict = xt is not None and np.size(xt) > 0
    out = _run_gf(blk, None, lik_param, yall, return_ind, mom, ep_fraction, ep_damping, ep_itts, predict, nargout,
                  predict_at_k1=1, device=device)
    if predict:
        return _returns(out, return_ind, nargout)
    return float(out.nlZ[0]), np.zeros(np.size(w))


def _run_ihgp(blk, Wnmf, lik_param, yall, mom, ep_fraction, ep_damping, ep_itts, constraints_variant, device=0, flags=0):
    prob = _Problem(blk, Wnmf, lik_param, symmetrize_Q=True)                    # :97  Q=(Q+Q')/2
    r, PP, ppo, PG, pgo = ihgp_tables.build_tables(prob.A, prob.Q, blk.offsets, blk.h_val)
    r = L.f64(r, 'C'); PP = L.f64(PP, 'C'); PG = L.f64(PG, 'C')
    tabs = L.IhgpTables(n_grid=r.size, r_grid=L.dptr(r), PPlist=L.dptr(PP), pp_offsets=ppo.ctypes.data_as(L.c_lp),
                        PGlist=L.dptr(PG), pg_offsets=pgo.ctypes.data_as(L.c_lp))
    damp = _damping(ep_damping, ep_itts)
    opts, keep = make_opts(L.KIND_IHGP, L.MODE_PREDICT, mom, blk.N, ep_fraction, damp, ep_itts,
                           flags=flags | (L.FLAG_IHGP_CONSTRAINTS if constraints_variant else 0), device=device)
    out = _Outputs(blk.M, blk.S, yall.size, ep_itts)
    L.check(L.lib().nagp_ihgp_run(C.byref(prob.model), C.byref(tabs), L.dptr(yall), yall.size, C.byref(opts), C.byref(out.c)))
    return out


def ihgp_ep_modulator_nmf(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N, ep_fraction=0.5, ep_damping=None,
                          ep_itts=30, nargout=2, device=0):
    """matlab/ihgp_ep_modulator_nmf.m:1 (predict mode; the reference's nlml mode is broken, SURVEY C-11)."""
    if xt is None or np.size(xt) == 0:
        raise NotImplementedError('ihgp nlml mode is broken in the reference (P undefined, ihgp_ep_modulator_nmf.m:555)')
    yall, return_ind = _merge_inputs(x, y, xt)
    lik_param, p1, p2, Wnmf = _unpack_log(w, num_lik_params, D, N)
    blk = ssm.balance_blocks(_blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N))   # :81 `if true`
    out = _run_ihgp(blk, Wnmf, lik_param, yall, mom, ep_fraction, ep_damping, ep_itts, False, device)
    return _returns(out, return_ind, nargout)


def ihgp_ep_modulator_nmf_constraints(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N, ep_fraction, ep_damping,
                                      ep_itts, constraints, w_fixed, tune_hypers, nargout=2, device=0):
    """matlab/ihgp_ep_modulator_nmf_constraints.m:1-2."""
    if xt is None or np.size(xt) == 0:
        raise NotImplementedError('ihgp nlml mode is broken in the reference (SURVEY C-11)')
    yall, return_ind = _merge_inputs(x, y, xt)
    lik_param, p1, p2, Wnmf = _unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
    blk = ssm.balance_blocks(_blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N))
    out = _run_ihgp(blk, Wnmf, lik_param, yall, mom, ep_fraction, ep_damping, ep_itts, True, device)
    return _returns(out, return_ind, nargout)


def _stack_sources(ss, x, w, kernel1, kernel2, J):
    """experiments/gf_ep_mods_nmf_mixture.m:89-128: J models side by side, all sub-band blocks first, then all
    modulator blocks; Wnmf block diagonal.  w = {log sn2, {param1_j}, {param2_j}, {W_j}} in natural units."""
    import scipy.linalg as sla
    zs = [[], [], [], []]; gs = [[], [], [], []]; Ws = []
    D = N = 0
    for j in range(J):
        p1 = np.asarray(w[1][j], float).ravel(); p2 = np.asarray(w[2][j], float).ravel()
        D_ = p1.size // 3; N_ = p2.size // 2
        D += D_; N += N_
        Ws.append(np.atleast_2d(np.asarray(w[3][j], float)))
        F, L_, Qc, H, Pinf = (np.asarray(a, float) for a in ss(x, p1, p2, kernel1[j], kernel2[j]))
        LQL = L_ @ np.atleast_2d(Qc) @ L_.T
        nz = D_ * 2 * ssm.KERNEL_ORDER[kernel1[j]]
        for lst, a in zip(zs, (F[:nz, :nz], LQL[:nz, :nz], H[:D_, :nz], Pinf[:nz, :nz])):
            lst.append(a)
        for lst, a in zip(gs, (F[nz:, nz:], LQL[nz:, nz:], H[D_:, nz:], Pinf[nz:, nz:])):
            lst.append(a)
    F, LQL, H, Pinf = (sla.block_diag(*(z + g)) for z, g in zip(zs, gs))
    blk = _blocks_from_dense(F, np.eye(F.shape[0]), LQL, H, Pinf, D, N)
    return blk, sla.block_diag(*Ws), np.atleast_1d(np.asarray(w[0], float))


def _mixture_mom(mom):
    if mom.kind == L.LIK_POWER:
        raise ValueError('the mixture variants take the NMF likelihoods only')
    return mom


def gf_ep_mods_nmf_mixture(w, x, y, ss, mom, xt, kernel1, kernel2, J, ep_fraction=0.5, ep_damping=0.1, ep_itts=30,
                           nargout=2, device=0):
    """matlab/experiments/gf_ep_mods_nmf_mixture.m:1 -- source separation: J stacked GT-NMF models, the older
    Power-EP rule (mom at power ep_fraction in the filter too, d/ep_fraction scaling, clamp in the filter pass,
    NAGP_FLAG_MIXTURE_RULE).  `mom` is the usual Mom object; its ep_frac argument is bound to ep_fraction as the
    6-argument closure of these files does.  Scalar ep_damping."""
    if xt is None or np.size(xt) == 0:
        raise RuntimeError('this mixture script is not for training')            # :376
    yall, return_ind = _merge_inputs(x, y, xt)
    blk, Wnmf, lik_param = _stack_sources(ss, x, w, kernel1, kernel2, J)
    out = _run_gf(blk, Wnmf, lik_param, yall, return_ind, _mixture_mom(mom), ep_fraction, float(np.ravel(ep_damping)[0]), ep_itts,
                  True, nargout, device=device, flags=L.FLAG_MIXTURE_RULE)
    return _returns(out, return_ind, nargout)


def ihgp_ep_mods_nmf_mixture(w, x, y, ss, mom, xt, kernel1, kernel2, J, ep_fraction=0.5, ep_damping=0.1, ep_itts=30,
                             nargout=2, device=0):
    """matlab/experiments/ihgp_ep_mods_nmf_mixture.m:1 (the inference of source_sep_piano.m:137-141): as above on
    the infinite-horizon filter/smoother; no balancing, R starts at 0, no abs(Varft)."""
    if xt is None or np.size(xt) == 0:
        raise RuntimeError('this mixture script is not for training')            # :547
    yall, return_ind = _merge_inputs(x, y, xt)
    blk, Wnmf, lik_param = _stack_sources(ss, x, w, kernel1, kernel2, J)
    out = _run_ihgp(blk, Wnmf, lik_param, yall, _mixture_mom(mom), ep_fraction, float(np.ravel(ep_damping)[0]), ep_itts, False,
                    device, flags=L.FLAG_MIXTURE_RULE)
    return _returns(out, return_ind, nargout)


def _run_giekf(blk, Wnmf, lik_param, yall, g_iter, l_iter, reset_P, nargout, device=0, nlml=False):
    prob = _Problem(blk, Wnmf, lik_param, stationary_Q=nlml)
    want_PS = nargout >= 6 and not nlml
    flags = (L.FLAG_EKF_RESET_P if reset_P else 0) | (L.FLAG_WANT_PS if want_PS else 0)
    opts, keep = make_opts(L.KIND_GIEKF, L.MODE_NLML if nlml else L.MODE_PREDICT, None, blk.N, 0.0, None, g_iter, l_iter=l_iter,
                           flags=flags, device=device)
    out = _Outputs(blk.M, blk.S, yall.size, g_iter, want_PS=want_PS)
    L.check(L.lib().nagp_giekf_run(C.byref(prob.model), L.dptr(yall), yall.size, C.byref(opts), C.byref(out.c)))
    return out


def gf_giekf_modulator_nmf(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N, g_iter, l_iter, GradObj='off',
                           nargout=2, device=0):
    """matlab/gf_giekf_modulator_nmf.m:1-2 (predict mode; `mom` is accepted and unused, as in the reference :13)."""
    if xt is None or np.size(xt) == 0:
        # gf_giekf_modulator_nmf.m:296-439 cannot run as committed: with GradObj='off' the loop that defines mm/PP is
        # skipped (:320-370), and funhd/funhd2 index with sum(H,1)==1 (:457,471), which no longer selects the scaled columns
        # of the balanced H (:78-81).  The training scripts call the _constraints variant (train_GTFNMF.m:199).
        raise NotImplementedError('the nlml branch of gf_giekf_modulator_nmf.m does not run in the reference; '
                                  'use gf_giekf_modulator_nmf_constraints (train_GTFNMF.m:199)')
    yall, return_ind = _merge_inputs(x, y, xt)
    lik_param, p1, p2, Wnmf = _unpack_log(w, num_lik_params, D, N)
    blk = ssm.balance_blocks(_blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N))   # :78 `if true`
    out = _run_giekf(blk, Wnmf, lik_param, yall, g_iter, l_iter, False, nargout, device)
    return _returns(out, return_ind, nargout)


def _giekf_grad_slices(blk, p1, p2, kernel1, kernel2, consistent):
    """Per-slice inputs of nagp_giekf_nlml_grad for the balanced block model `blk`: the slices the reference builds
    (gf_giekf_modulator_nmf_constraints.m:121-125: noise variance, then [sig1, len1, omega, sig2, len2] of ss_modulators_nmf.m:88, 130)
    -- dA_j from expm([F 0; dF_j F]) (:355-366), dQ_j (:392-394), dPinf_j -- block by block (every slice touches one block).
    consistent=False: dF_j, dPinf_j UNBALANCED beside the balanced F, Pinf, as the reference has them (:117-119 commented out);
    True: carried through the balancing (T\\dF*T, T\\dPinf/T')."""
    D, N, S = blk.D, blk.N, blk.S
    n_par = 1 + 3 * D + 2 * N
    dA = np.zeros((n_par, S, S)); dQ = np.zeros((n_par, S, S)); dPi = np.zeros((n_par, S, S))
    tb = getattr(blk, 'tbal', [np.ones(k) for k in blk.sizes])
    p1 = np.asarray(p1, float).ravel(); p2 = np.asarray(p2, float).ravel()

    def put(j, n, dF, dP):
        o, e = blk.offsets[n], blk.offsets[n + 1]; b = e - o
        if consistent:
            t = tb[n]; dF = dF * np.outer(1.0 / t, t); dP = dP / np.outer(t, t)
        Fn, Pn = blk.F[n], blk.Pinf[n]
        A = sla.expm(Fn)
        E = sla.expm(np.block([[Fn, np.zeros((b, b))], [dF, Fn]]))
        dAn = E[b:, :b]
        X = dAn @ Pn @ A.T
        dA[j, o:e, o:e] = dAn; dPi[j, o:e, o:e] = dP
        dQ[j, o:e, o:e] = dP - X - A @ dP @ A.T - X.T

    I2 = np.eye(2)
    for d in range(D):
        (dFs, dPs), (dFl, dPl) = ssm.kernel_block_derivs(kernel1, p1[d], p1[D + d])
        t1 = dFs.shape[0]
        put(1 + d, d, np.kron(dFs, I2), np.kron(dPs, I2))
        put(1 + D + d, d, np.kron(dFl, I2), np.kron(dPl, I2))
        put(1 + 2 * D + d, d, np.kron(np.eye(t1), np.array([[0.0, -1.0], [1.0, 0.0]])), np.zeros((2 * t1, 2 * t1)))
    for n in range(N):
        (dFs, dPs), (dFl, dPl) = ssm.kernel_block_derivs(kernel2, p2[n], p2[N + n])
        put(1 + 3 * D + n, D + n, dFs, dPs)
        put(1 + 3 * D + N + n, D + n, dFl, dPl)
    return dA, dQ, dPi


def giekf_nlml_grad(blk, Wnmf, lik_param, p1, p2, kernel1, kernel2, yall, consistent=False, device=0):
    """(edata, gdata) of the EKF energy and its gradient recursion on the GPU (nagp_giekf_nlml_grad).
    consistent=False: the reference's statements as written -- 1+3D+2N slices, the last D*N of them with the Jacobian derivative
    taken w.r.t. an entry of W while dm, dP carry the kernel parameter of the same index (:438-444), unbalanced dF / dPinf.
    consistent=True: the gradient of the energy w.r.t. [sigma2, sig1, len1, omega, sig2, len2, W(:)] (1+3D+2N+D*N entries)."""
    D, N, S = blk.D, blk.N, blk.S
    dA, dQ, dPi = _giekf_grad_slices(blk, p1, p2, kernel1, kernel2, consistent)
    n_k = 1 + 3 * D + 2 * N
    if consistent:
        z = np.zeros((D * N, S, S))
        dA = np.concatenate([dA, z]); dQ = np.concatenate([dQ, z]); dPi = np.concatenate([dPi, z])
        hess = np.ones(n_k + D * N, np.int32); widx = np.concatenate([-np.ones(n_k, np.int32), np.arange(D * N, dtype=np.int32)])
        wdir = (widx >= 0).astype(np.int32)
    else:
        nk = n_k - D * N
        idx = np.arange(n_k)
        hess = (idx < nk).astype(np.int32); widx = np.where(idx < nk, -1, idx - nk).astype(np.int32); wdir = np.zeros(n_k, np.int32)
    n_par = dA.shape[0]
    dR = np.zeros(n_par); dR[0] = 1.0
    prob = _Problem(blk, Wnmf, lik_param, stationary_Q=True)
    cm = lambda a: L.f64(np.ascontiguousarray(np.transpose(a, (0, 2, 1))), 'C')       # every slice column-major
    dAc, dQc, dPc = cm(dA), cm(dQ), cm(dPi)
    y = L.f64(np.asarray(yall, float).ravel(), 'C')
    models = (L.Model * 1)(prob.model)
    arr = lambda a: (L.c_dp * 1)(L.dptr(a))
    e = np.zeros(1); g = np.zeros(n_par)
    L.check(L.lib().nagp_giekf_nlml_grad(1, models, arr(y), y.size, n_par, arr(dAc), arr(dQc), arr(dPc), L.dptr(L.f64(dR, 'C')),
                                         hess.ctypes.data_as(L.c_ip), widx.ctypes.data_as(L.c_ip), wdir.ctypes.data_as(L.c_ip),
                                         L.dptr(e), L.dptr(g), int(device)))
    return float(e[0]), g


def gf_giekf_modulator_nmf_constraints(w, x, y, ss, mom, xt, kernel1, kernel2, num_lik_params, D, N, g_iter, l_iter,
                                       constraints, w_fixed, tune_hypers, GradObj='off', nargout=2, device=0):
    """matlab/gf_giekf_modulator_nmf_constraints.m:1-2.  xt empty: [e, eg] of :332-480 with GradObj='off' (what
    train_GTFNMF.m:199 / train_model.m:239 pass): one plain EKF pass, e = sum_k log(2pi)/2 + log sqrt(S_k) + v_k^2/(2 S_k),
    eg = zeros(1,numel(w)).  GradObj='on': the dm/dP recursion of :355-402, 437-466 as written (nagp_giekf_nlml_grad; it indexes
    gdata(1..size(dF,3)) in a vector of numel(w) entries, so it runs only when at least 1+3D+2N parameters are tuned -- an
    IndexError otherwise, as MATLAB's).  GradObj='consistent' (not in the reference): the gradient of the energy with respect to
    the natural parameters, which central differences of the energy confirm."""
    yall, return_ind = _merge_inputs(x, y, xt)
    lik_param, p1, p2, Wnmf = _unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
    blk = ssm.balance_blocks(_blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N))
    if xt is None or np.size(xt) == 0:
        if GradObj == 'on':
            # :332-480 as written: gdata = zeros(1,length(w)) is indexed with j = 1..size(dF,3) = 1+3D+2N (:336-342, :453-457) --
            # MATLAB stops with an index error when fewer parameters than that are tuned; with all groups tuned it runs
            n_par = 1 + 3 * D + 2 * N
            if np.size(w) < n_par:
                raise IndexError('Index exceeds the number of array elements (%d): gdata(j) is read for j up to size(dF,3) = %d '
                                 '(gf_giekf_modulator_nmf_constraints.m:453-457)' % (np.size(w), n_par))
            e, g = giekf_nlml_grad(blk, Wnmf, lik_param, p1, p2, kernel1, kernel2, yall, consistent=False, device=device)
            eg = np.zeros(np.size(w)); eg[:n_par] = g
            return e, eg
        if GradObj == 'consistent':
            # not in the reference: the same recursion as the true gradient of the energy w.r.t. the NATURAL parameters
            # [sigma2, sig1, len1, omega, sig2, len2, W(:)] (central differences of the energy confirm it, tests/test_gpu_parity.py)
            return giekf_nlml_grad(blk, Wnmf, lik_param, p1, p2, kernel1, kernel2, yall, consistent=True, device=device)
        if GradObj != 'off':
            raise ValueError("GradObj must be 'off', 'on' or 'consistent'")
        out = _run_giekf(blk, Wnmf, lik_param, yall, 1, 1, True, 2, device, nlml=True)
        return float(out.nlZ[0]), np.zeros(np.size(w))
    out = _run_giekf(blk, Wnmf, lik_param, yall, g_iter, l_iter, True, nargout, device)
    return _returns(out, return_ind, nargout)
