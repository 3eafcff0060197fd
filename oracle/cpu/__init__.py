"""ORACLE / CPU BASELINE (test and measurement infrastructure only -- never imported by the product path).

ctypes loader of oracle/cpu/nagp_cpu.cpp: the compiled, dense-as-written CPU restatement of the reference's hot loops
(plain loops, g++ -O3 -march=native -fopenmp, own code; see the header of the .cpp for the reference lines it follows).
Two uses: (i) a second restatement held against the NumPy oracle on the committed golden vectors
(tests/test_cpu_restatement.py), (ii) the timed `cpu_baseline` of bench.py (kind "port").

The shared object is compiled ON THE HOST THAT RUNS IT (-march=native differs between the build container and the GPU
box) into oracle/_build/<flags+source hash>/, a git-ignored directory.
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, 'nagp_cpu.cpp')
BUILD_ROOT = os.path.join(os.path.dirname(_HERE), '_build')
FLAGS = ['-O3', '-march=native', '-fopenmp', '-shared', '-fPIC', '-std=c++17']

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)
_lib = None


def _cpu_tag():
    """what -march=native resolves to here (model name + flags line of /proc/cpuinfo)"""
    tag = ''
    try:
        with open('/proc/cpuinfo') as fh:
            for ln in fh:
                if ln.startswith(('model name', 'flags')):
                    tag += ln
                if ln.startswith('flags'):
                    break
    except OSError:
        pass
    return tag


def build(force=False):
    with open(SRC, 'rb') as fh:
        h = hashlib.sha256(fh.read() + ' '.join(FLAGS).encode() + _cpu_tag().encode()).hexdigest()[:16]
    d = os.path.join(BUILD_ROOT, h)
    so = os.path.join(d, 'libnagp_cpu.so')
    if os.path.exists(so) and not force:
        return so
    os.makedirs(d, exist_ok=True)
    tmp = so + '.tmp.%d' % os.getpid()
    r = subprocess.run(['g++'] + FLAGS + ['-o', tmp, SRC], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError('g++ failed on oracle/cpu/nagp_cpu.cpp:\n' + r.stderr)
    os.replace(tmp, so)
    return so


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.nagp_cpu_mom.restype = C.c_double
        for f in ('nagp_cpu_gf_predict', 'nagp_cpu_ihgp_predict', 'nagp_cpu_giekf_predict', 'nagp_cpu_segments', 'nagp_cpu_threads'):
            getattr(L, f).restype = C.c_int
        _lib = L
    return _lib


def _d(a):
    return np.ascontiguousarray(np.asarray(a, dtype=np.float64))


def _p(a):
    return a.ctypes.data_as(c_dp)


def _cub(mom, dim):
    """(lik_kind, link_kind, link_shift, wn, xn[dim][npts]) of an oracle.lik.Mom"""
    from .. import cubature as ocub, lik as olik
    if mom.kind == olik.LIK_POWER_NMF_SQRT:
        wn, xn = mom.wn, mom.xn_unscaled
    else:
        wn, xn = ocub.sigma_points(mom.p, dim, mom.quirks)
    # the link is a closure in the oracle: probe it (softplus(shift) or exp)
    v0 = float(mom.link(np.array([0.0]))[0])
    if abs(v0 - 1.0) < 1e-15:
        link_kind, shift = 1, 0.0
    else:
        link_kind, shift = 0, float(-np.log(np.expm1(v0)))      # log(1+exp(-shift)) = v0
    return mom.kind, link_kind, shift, _d(np.ravel(wn)), _d(xn)


def mom(mom_obj, hyp, y, mu, s2, W, ep_fraction):
    from .. import lik as olik
    mu = _d(mu); s2 = _d(s2)
    if mom_obj.kind == olik.LIK_POWER:
        D = mu.size // 2; N = D; Wc = _d(np.eye(D))
    else:
        Wc = _d(W); D, N = Wc.shape
    kind, lk, sh, wn, xn = _cub(mom_obj, N)
    dl = np.zeros(mu.size); d2 = np.zeros(mu.size)
    lZ = lib().nagp_cpu_mom(C.c_int(kind), C.c_int(lk), C.c_double(sh), C.c_int(wn.size), C.c_int(xn.shape[0]), _p(wn), _p(xn), C.c_double(float(np.ravel(hyp)[0])),
                            C.c_double(float(y)), _p(mu), _p(s2), _p(Wc), C.c_int(D), C.c_int(N), C.c_double(ep_fraction), _p(dl), _p(d2))
    return lZ, dl, d2


def _model_args(model, D, N):
    A, Q, H, Pinf = (_d(model[k]) for k in ('A', 'Q', 'H', 'Pinf'))
    W = _d(model['Wnmf']) if model.get('Wnmf') is not None else _d(np.eye(D))
    S, M = A.shape[0], H.shape[0]
    keep = (A, Q, H, Pinf, W)
    return [C.c_int(S), C.c_int(M), C.c_int(D), C.c_int(N), _p(A), _p(Q), _p(H), _p(Pinf), _p(W), C.c_double(float(np.ravel(model['lik_param'])[0]))], keep, S, M


def _ilist(model):
    from .. import ss as oss
    return np.ascontiguousarray(oss.block_starts(model['H']), dtype=np.int32)


def gf_predict(model, y, mom_obj, ep_fraction, ep_damping, ep_itts, D, N, predict_at_k1=False, structured=False):
    """oracle.gf_ep.run_predict on the compiled restatement (same assembled model dict).  structured=True: block-diagonal A,
    selection H, symmetric gain update (the stronger CPU baseline); False: dense as written."""
    margs, keep, S, M = _model_args(model, D, N)
    il = _ilist(model)
    kind, lk, sh, wn, xn = _cub(mom_obj, D if mom_obj.kind == 0 else N)
    y = _d(y); T = y.size; I = int(ep_itts)
    damp = _d(np.broadcast_to(np.atleast_1d(np.asarray(ep_damping, float)).ravel() if np.size(ep_damping) > 1 else np.full(I, float(np.ravel(ep_damping)[0])), (I,)))
    out = {k: np.zeros((M, T)) for k in ('Eft', 'Varft', 'ttau', 'tnu')}
    out.update(nlZ=np.zeros(I), lZ=np.zeros(T), maxDiffM=np.zeros(I), maxDiffP=np.zeros(I))
    cnt = np.zeros(2, dtype=np.int64)
    st = lib().nagp_cpu_gf_predict(*margs, C.c_int(kind), C.c_int(lk), C.c_double(sh), C.c_int(wn.size), C.c_int(xn.shape[0]), _p(wn), _p(xn), _p(y), C.c_int64(T),
                                   C.c_double(ep_fraction), _p(damp), C.c_int(I), C.c_int(1 if predict_at_k1 else 0), il.ctypes.data_as(c_ip), C.c_int(1 if structured else 0), _p(out['Eft']), _p(out['Varft']), _p(out['nlZ']),
                                   _p(out['ttau']), _p(out['tnu']), _p(out['lZ']), _p(out['maxDiffM']), _p(out['maxDiffP']), cnt.ctypes.data_as(c_lp))
    out['status'] = st; out['counters'] = dict(chol_retries=int(cnt[0]), not_pd=int(cnt[1]))
    return out


def _flat_tables(tables, M):
    ilist, r, PPlist, PGlist = tables
    ilist = np.ascontiguousarray(ilist, dtype=np.int32)
    pp = np.concatenate([_d(PPlist[n]).ravel() for n in range(M)]); pg = np.concatenate([_d(PGlist[n]).ravel() for n in range(M)])
    ppo = np.zeros(M, dtype=np.int64); pgo = np.zeros(M, dtype=np.int64)
    a = b = 0
    for n in range(M):
        ppo[n] = a; pgo[n] = b
        a += np.size(PPlist[n]); b += np.size(PGlist[n])
    return ilist, _d(r), pp, ppo, pg, pgo


def ihgp_predict(model, y, mom_obj, ep_fraction, ep_damping, ep_itts, D, N, tables, constraints_variant=False, structured=False):
    """oracle.ihgp.run_predict on the compiled restatement; `tables` = oracle.ihgp.build_tables(model)."""
    margs, keep, S, M = _model_args(model, D, N)
    kind, lk, sh, wn, xn = _cub(mom_obj, N)
    ilist, r, pp, ppo, pg, pgo = _flat_tables(tables, M)
    y = _d(y); T = y.size; I = int(ep_itts)
    damp = _d(np.full(I, float(np.ravel(ep_damping)[0])) if np.size(ep_damping) == 1 else np.asarray(ep_damping, float)[:I])
    out = {k: np.zeros((M, T)) for k in ('Eft', 'Varft', 'ttau', 'tnu', 'R')}
    out.update(nlZ=np.zeros(I), maxDiffM=np.zeros(I), maxDiffP=np.zeros(I))
    st = lib().nagp_cpu_ihgp_predict(*margs, C.c_int(kind), C.c_int(lk), C.c_double(sh), C.c_int(wn.size), C.c_int(xn.shape[0]), _p(wn), _p(xn),
                                     ilist.ctypes.data_as(c_ip), C.c_int(r.size), _p(r), _p(pp), ppo.ctypes.data_as(c_lp), _p(pg), pgo.ctypes.data_as(c_lp),
                                     _p(y), C.c_int64(T), C.c_double(ep_fraction), _p(damp), C.c_int(I), C.c_int(1 if constraints_variant else 0), C.c_int(1 if structured else 0),
                                     _p(out['Eft']), _p(out['Varft']), _p(out['nlZ']), _p(out['ttau']), _p(out['tnu']), _p(out['R']), _p(out['maxDiffM']), _p(out['maxDiffP']))
    out['status'] = st
    return out


def giekf_predict(model, y, D, N, g_iter, l_iter, constraints_variant=False, structured=False):
    """oracle.giekf.run_predict on the compiled restatement."""
    margs, keep, S, M = _model_args(model, D, N)
    il = _ilist(model)
    y = _d(y); T = y.size
    out = dict(Eft=np.zeros((M, T)), Varft=np.zeros((M, T)), maxDiffP=np.zeros(g_iter))
    cnt = np.zeros(2, dtype=np.int64)
    st = lib().nagp_cpu_giekf_predict(*margs, _p(y), C.c_int64(T), C.c_int(g_iter), C.c_int(l_iter), C.c_int(1 if constraints_variant else 0), il.ctypes.data_as(c_ip), C.c_int(1 if structured else 0),
                                      _p(out['Eft']), _p(out['Varft']), _p(out['maxDiffP']), cnt.ctypes.data_as(c_lp))
    out['status'] = st; out['counters'] = dict(chol_retries=int(cnt[0]), not_pd=int(cnt[1]))
    return out


def segments(kind, model, ys, mom_obj, ep_fraction, ep_damping, ep_itts, D, N, tables=None, l_iter=1, threads=None, structured=False):
    """`len(ys)` independent segments of one family over the host cores (OpenMP): returns (nlZ[nseg][I], status).
    kind: 'gf' | 'ihgp' | 'giekf'."""
    k = {'gf': 0, 'ihgp': 1, 'giekf': 2}[kind]
    margs, keep, S, M = _model_args(model, D, N)
    if mom_obj is not None:
        ck, lk, sh, wn, xn = _cub(mom_obj, N)
    else:
        ck, lk, sh, wn, xn = 1, 0, 0.0, _d(np.ones(1)), _d(np.zeros((N, 1)))
    if tables is not None:
        ilist, r, pp, ppo, pg, pgo = _flat_tables(tables, M)
    else:
        ilist = _ilist(model); r = _d(np.ones(1)); pp = pg = _d(np.zeros(1)); ppo = pgo = np.zeros(M, dtype=np.int64)
    Y = _d(np.stack([np.asarray(y, float) for y in ys])); nseg, T = Y.shape; I = int(ep_itts)
    damp = _d(np.full(I, 0.5) if ep_damping is None else (np.full(I, float(np.ravel(ep_damping)[0])) if np.size(ep_damping) == 1 else np.asarray(ep_damping, float)[:I]))
    nlz = np.zeros((nseg, I))
    st = lib().nagp_cpu_segments(C.c_int(k), C.c_int(nseg), *margs, C.c_int(ck), C.c_int(lk), C.c_double(sh), C.c_int(wn.size), C.c_int(xn.shape[0]), _p(wn), _p(xn),
                                 ilist.ctypes.data_as(c_ip), C.c_int(r.size), _p(r), _p(pp), ppo.ctypes.data_as(c_lp), _p(pg), pgo.ctypes.data_as(c_lp),
                                 _p(Y), C.c_int64(T), C.c_double(ep_fraction), _p(damp), C.c_int(I), C.c_int(l_iter), C.c_int(1 if structured else 0), C.c_int(int(threads or 0)), _p(nlz))
    return nlz, st


def threads():
    return int(lib().nagp_cpu_threads())
