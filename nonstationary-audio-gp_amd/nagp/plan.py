"""Device-resident, batched form of the C ABI (nagp_plan_*): B independent problems of identical shape
(audio segments / hyper-parameter replicas) with inputs and all intermediates kept in HBM."""
import ctypes as C

import numpy as np

from . import _lib as L
from .api import _Outputs, _Problem, make_opts, _damping
from . import ihgp_tables


class Plan:
    def __init__(self, kind, problems, T, mom=None, ep_fraction=0.5, ep_damping=None, ep_itts=3, mode=L.MODE_PREDICT,
                 l_iter=1, predict_at_k1=0, flags=0, device=0, chunk=0):
        """problems: list of (BlockSS, Wnmf, lik_param[, overrides]) -- already balanced if the variant balances; `overrides` is an
        optional dict of dense S x S arrays 'A', 'Q', 'Pinf' replacing the discretised blocks (block diagonal with the same blocks)."""
        self.kind, self.T, self.I = kind, int(T), int(ep_itts)
        sym = kind == L.KIND_IHGP
        statq = kind == L.KIND_GIEKF and mode == L.MODE_NLML        # Q = Pinf - A Pinf A' (gf_giekf_modulator_nmf_constraints.m:378)
        self.probs = [_Problem(pr[0], pr[1], pr[2], symmetrize_Q=sym, stationary_Q=statq, overrides=(pr[3] if len(pr) > 3 else None)) for pr in problems]
        self.B = len(self.probs)
        blk0 = problems[0][0]
        self.M, self.S = blk0.M, blk0.S
        dim = None
        if mom is not None:
            dim = blk0.D if mom.kind == L.LIK_POWER else blk0.N
        damp = _damping(ep_damping, ep_itts) if ep_damping is not None else None
        self.opts, self._keep = make_opts(kind, mode, mom, dim, ep_fraction, damp, ep_itts, l_iter=l_iter,
                                          predict_at_k1=predict_at_k1, flags=flags, device=device, chunk=chunk)
        models = (L.Model * self.B)(*[p.model for p in self.probs])
        tabs_arr = None
        self._tabkeep = []
        if kind == L.KIND_IHGP:
            tl = []
            for p in self.probs:
                r, PP, ppo, PG, pgo = ihgp_tables.build_tables(p.A, p.Q, p.blk.offsets, p.blk.h_val)
                r = L.f64(r, 'C'); PP = L.f64(PP, 'C'); PG = L.f64(PG, 'C')
                self._tabkeep.append((r, PP, ppo, PG, pgo))
                tl.append(L.IhgpTables(n_grid=r.size, r_grid=L.dptr(r), PPlist=L.dptr(PP), pp_offsets=ppo.ctypes.data_as(L.c_lp),
                                       PGlist=L.dptr(PG), pg_offsets=pgo.ctypes.data_as(L.c_lp)))
            tabs_arr = (L.IhgpTables * self.B)(*tl)
        self._h = C.c_void_p()
        L.check(L.lib().nagp_plan_create(C.byref(self._h), self.B, models, tabs_arr, self.T, C.byref(self.opts)))

    def upload(self, ys):
        ys = [L.f64(y, 'C') for y in ys]
        assert len(ys) == self.B and all(y.size == self.T for y in ys)
        arr = (L.c_dp * self.B)(*[L.dptr(y) for y in ys])
        L.check(L.lib().nagp_plan_upload_y(self._h, arr))

    def upload_sites(self, ttau0=None, tnu0=None):
        """Warm start: initial site parameters (lists of M x T arrays, e.g. the ttau / tnu a previous call returned) in place
        of the reference's zeros (gf_ep_modulator_nmf.m:96-97); None, None returns to cold starts."""
        if ttau0 is None and tnu0 is None:
            L.check(L.lib().nagp_plan_upload_sites(self._h, None, None)); return
        tt = [L.f64(a) for a in ttau0]; tn = [L.f64(a) for a in tnu0]
        assert len(tt) == self.B and len(tn) == self.B and all(a.shape == (self.M, self.T) for a in tt + tn)
        a1 = (L.c_dp * self.B)(*[L.dptr(a) for a in tt]); a2 = (L.c_dp * self.B)(*[L.dptr(a) for a in tn])
        L.check(L.lib().nagp_plan_upload_sites(self._h, a1, a2))

    def execute(self):
        L.check(L.lib().nagp_plan_execute(self._h))

    def timings(self):
        t = L.Timings()
        L.check(L.lib().nagp_plan_timings(self._h, C.byref(t)))
        return dict(ms={k: t.ms[i] for i, k in enumerate(L.KERNEL_NAMES)},
                    launches={k: int(t.launches[i]) for i, k in enumerate(L.KERNEL_NAMES)}, total_ms=t.total_ms)

    def download(self, want_PS=False, want_MS=True, want_MF=False):
        outs = [_Outputs(self.M, self.S, self.T, self.I, want_PS=want_PS, want_MS=want_MS, want_MF=want_MF) for _ in range(self.B)]
        arr = (L.Out * self.B)(*[o.c for o in outs])
        L.check(L.lib().nagp_plan_download(self._h, arr))
        return outs

    def download_nlz(self):
        """(B, ep_itts) negative log marginal likelihoods only (nothing else crosses PCIe)."""
        nlz = np.zeros((self.B, self.I))
        arr = (L.Out * self.B)(*[L.Out(nlZ=nlz[q].ctypes.data_as(L.c_dp)) for q in range(self.B)])
        L.check(L.lib().nagp_plan_download(self._h, arr))
        return nlz

    def device_bytes(self):
        return int(L.lib().nagp_plan_device_bytes(self._h))

    def close(self):
        if self._h:
            L.lib().nagp_plan_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def batch_partition(n_problems, n_gpus):
    """problem -> device map of nagp_batch_run (problem i on device i mod n_gpus).  Pure host code."""
    dev = np.zeros(max(n_problems, 1), dtype=np.int32)
    L.check(L.lib().nagp_batch_partition(int(n_problems), int(n_gpus), dev.ctypes.data_as(L.c_ip)))
    return dev[:n_problems]


def batch_run(kind, problems, ys, T, mom=None, ep_fraction=0.5, ep_damping=None, ep_itts=3, mode=L.MODE_PREDICT, l_iter=1,
              predict_at_k1=0, flags=0, n_gpus=1, want_PS=False, want_MS=True):
    """nagp_batch_run: the problems spread over the GPUs of this node inside ONE process (host thread + plan per device),
    nlZ summed over all problems by an RCCL all-reduce.  Returns (outputs per problem, nlZ_total[ep_itts])."""
    sym = kind == L.KIND_IHGP
    statq = kind == L.KIND_GIEKF and mode == L.MODE_NLML
    probs = [_Problem(b, W, lp, symmetrize_Q=sym, stationary_Q=statq) for (b, W, lp) in problems]
    B = len(probs); blk0 = problems[0][0]
    dim = None
    if mom is not None:
        dim = blk0.D if mom.kind == L.LIK_POWER else blk0.N
    damp = _damping(ep_damping, ep_itts) if ep_damping is not None else None
    opts, keep = make_opts(kind, mode, mom, dim, ep_fraction, damp, ep_itts, l_iter=l_iter, predict_at_k1=predict_at_k1, flags=flags)
    models = (L.Model * B)(*[p.model for p in probs])
    tabs_arr, tabkeep = None, []
    if kind == L.KIND_IHGP:
        tl = []
        for p in probs:
            r, PP, ppo, PG, pgo = ihgp_tables.build_tables(p.A, p.Q, p.blk.offsets, p.blk.h_val)
            r = L.f64(r, 'C'); PP = L.f64(PP, 'C'); PG = L.f64(PG, 'C')
            tabkeep.append((r, PP, ppo, PG, pgo))
            tl.append(L.IhgpTables(n_grid=r.size, r_grid=L.dptr(r), PPlist=L.dptr(PP), pp_offsets=ppo.ctypes.data_as(L.c_lp),
                                   PGlist=L.dptr(PG), pg_offsets=pgo.ctypes.data_as(L.c_lp)))
        tabs_arr = (L.IhgpTables * B)(*tl)
    yk = [L.f64(y, 'C') for y in ys]
    assert len(yk) == B and all(y.size == T for y in yk)
    yarr = (L.c_dp * B)(*[L.dptr(y) for y in yk])
    outs = [_Outputs(blk0.M, blk0.S, int(T), int(ep_itts), want_PS=want_PS, want_MS=want_MS) for _ in range(B)]
    oarr = (L.Out * B)(*[o.c for o in outs])
    tot = np.zeros(int(ep_itts))
    L.check(L.lib().nagp_batch_run(B, models, tabs_arr, yarr, int(T), C.byref(opts), oarr, int(n_gpus), tot.ctypes.data_as(L.c_dp)))
    return outs, tot
