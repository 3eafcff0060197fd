"""developer diagnostics for one widened fuzz case: python tools/diag_mix.py seed index"""
import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import nagp
from nagp import harness, SSHandle
import gpu_fuzz as gf
from oracle import mixture as omx

seed, idx = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(seed)
# replay the draws of one_widened up to case idx (same rng consumption)
orig = gf.one_widened
cap = {}
def spy(rng):
    J = int(rng.integers(2, 4))
    shapes = [(int(rng.integers(1, 6)), int(rng.integers(1, 4))) for _ in range(J)]
    while sum(n for _, n in shapes) > 8:
        shapes[int(rng.integers(0, J))] = (2, 1)
    k1 = [str(rng.choice(['exp', 'matern32'])) for _ in range(J)]; k2 = [str(rng.choice(['matern32', 'matern52'])) for _ in range(J)]
    T = int(rng.integers(20, 70)); N = sum(n for _, n in shapes)
    p = int(rng.choice([5, 7] if N <= 4 else [7]))
    kind = str(rng.choice(['nmf', 'sqrt'])); shift = float(rng.choice([0.0, 1.0]))
    alpha = float(rng.choice([0.5, 0.75])); damp = float(rng.uniform(0.02, 0.4)); itts = int(rng.integers(1, 4))
    mp = harness.mixture_problem(shapes, T, int(rng.integers(1, 10 ** 6)), k1, k2)
    y = mp['y'].copy(); y[rng.random(T) < 0.08] = np.nan
    D = int(rng.integers(2, 12)); N2 = int(rng.integers(1, 5)); T2 = int(rng.integers(30, 200)); rng.integers(1, 10 ** 6)
    return dict(J=J, shapes=shapes, k1=k1, k2=k2, T=T, N=N, p=p, kind=kind, shift=shift, alpha=alpha, damp=damp, itts=itts, mp=mp, y=y)
for i in range(idx + 1):
    c = spy(rng)
print({k: v for k, v in c.items() if k not in ('mp', 'y')})
mom, omom = gf.moms(dict(kind=c['kind'], link='softplus', shift=c['shift'], p=c['p'], N=c['N']))
t = np.arange(1, c['T'] + 1.0); mp = c['mp']; J = c['J']
for itts in range(1, c['itts'] + 1):
    args = (t, mp['y'], SSHandle(), mom, t, c['k1'], c['k2'], J, c['alpha'], c['damp'], itts)
    a = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], *args, nargout=6)
    os.environ['NAGP_NO_SRC'] = '1'
    b = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], *args, nargout=6)
    del os.environ['NAGP_NO_SRC']
    o = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], None, omom, t, c['k1'], c['k2'], J, c['alpha'], c['damp'], itts)
    o2 = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'] * (1 + 1e-13), None, omom, t, c['k1'], c['k2'], J, c['alpha'], c['damp'], itts)
    r = gf.rel
    print('itts %d: src-vs-oracle Eft %.1e ttau %.1e | nosrc-vs-oracle Eft %.1e ttau %.1e | src-vs-nosrc Eft %.1e | oracle self Eft %.1e ttau %.1e | max|ttau| %.2e min R %.2e'
          % (itts, r(a[0], o[0]), r(a[5]['ttau'], o[5]['ttau']), r(b[0], o[0]), r(b[5]['ttau'], o[5]['ttau']), r(a[0], b[0]),
             r(o2[0], o[0]), r(o2[5]['ttau'], o[5]['ttau']), np.nanmax(np.abs(o[5]['ttau'])), np.nanmin(np.abs(o[5]['R']))))
    e = np.abs(a[5]['ttau'] - o[5]['ttau']) / (np.abs(o[5]['ttau']) + 1e-9 * np.nanmax(np.abs(o[5]['ttau'])))
    bad = np.where(np.nanmax(e, axis=0) > 1e-6)[0]
    if bad.size:
        k = bad[0]; print('   first bad step', k, 'of', bad.size, 'gpu', a[5]['ttau'][:, k], 'ora', o[5]['ttau'][:, k], 'R', o[5]['R'][:, k])

print('---- full-covariance variant')
for itts in range(1, c['itts'] + 1):
    args = (t, c['y'], SSHandle(), mom, t, c['k1'], c['k2'], J, c['alpha'], c['damp'], itts)
    a = nagp.gf_ep_mods_nmf_mixture(mp['w'], *args, nargout=6)
    o = omx.gf_ep_mods_nmf_mixture(mp['w'], t, c['y'], None, omom, t, c['k1'], c['k2'], J, c['alpha'], c['damp'], itts)
    yp = c['y'] * (1 + 1e-13)
    o2 = omx.gf_ep_mods_nmf_mixture(mp['w'], t, yp, None, omom, t, c['k1'], c['k2'], J, c['alpha'], c['damp'], itts)
    r = gf.rel
    print('itts %d: gpu-vs-oracle Eft %.1e Varft %.1e ttau %.1e | oracle self Eft %.1e Varft %.1e ttau %.1e | NaN Eft gpu/ora %d/%d  Varft<0 ora %d max|ttau| %.2e'
          % (itts, r(a[0], o[0]), r(a[1], o[1]), r(a[5]['ttau'], o[5]['ttau']), r(o2[0], o[0]), r(o2[1], o[1]), r(o2[5]['ttau'], o[5]['ttau']),
             int(np.isnan(a[0]).sum()), int(np.isnan(o[0]).sum()), int((o[1] < 0).sum()), np.nanmax(np.abs(o[5]['ttau']))))

    d = np.abs(a[5]['ttau'] - o[5]['ttau']); d[np.isnan(d)] = 0
    i, k = np.unravel_index(np.argmax(d), d.shape)
    print('   worst ttau entry: site %d step %d gpu %.17g ora %.17g | tnu gpu %.17g ora %.17g | y[k] %r | Varft gpu %.6g ora %.6g | neighbours ora ttau[:,k] %s' % (
        i, k, a[5]['ttau'][i, k], o[5]['ttau'][i, k], a[5]['tnu'][i, k], o[5]['tnu'][i, k], c['y'][k], a[1][i, k], o[1][i, k], o[5]['ttau'][:, k]))

print('---- mom at the worst step, same inputs on both sides')
rec = {}
class Spy:
    def __init__(self, m): self.m = m
    def __call__(self, hyp, mu, s2, W, a_, yv, k):
        out = self.m(hyp, mu, s2, W, a_, yv, k); rec[k] = (np.array(mu, float).copy(), np.array(s2, float).copy(), out, a_); return out
o = omx.gf_ep_mods_nmf_mixture(mp['w'], t, c['y'], None, Spy(omom), t, c['k1'], c['k2'], J, c['alpha'], c['damp'], 1)
from oracle import mixture as _m
st = _m.stack_models(mp['w'], c['k1'], c['k2'], J)
mu, s2, out, a_ = rec[k]
print('k', k, 'alpha', a_, 'mu', mu, 's2', s2)
g = mom(mp['w'][0], mu, s2, st['Wnmf'], a_, c['y'], int(k))
print('oracle lZ %.17g  gpu lZ %.17g' % (out[0], g[0]))
print('oracle dlZ ', np.ravel(out[1])); print('gpu    dlZ ', np.ravel(g[1]))
print('oracle d2lZ', np.ravel(out[2])); print('gpu    d2lZ', np.ravel(g[2]))
print('1 + d2*s2 oracle', 1 + np.ravel(out[2]) * s2, ' gpu', 1 + np.ravel(g[2]) * s2)
