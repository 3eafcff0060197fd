"""Re-run ONE draw of tools/gpu_fuzz.py (main family) through ihgp_ep_modulator_nmf with the kernel-selection switches,
against the oracle and the oracle's own sensitivity (developer tool):   python tools/diag_fuzz_ihgp.py [seed] [index]"""
import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import nagp
from nagp import SSHandle
from oracle import ihgp as oih
from gpu_fuzz_draws import draw, moms
import gpu_fuzz
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 99
index = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(seed)
for i in range(index + 1):
    c = draw(rng)
D, N, T, k1, k2, itts, alpha, damp, pr = (c[k] for k in ('D', 'N', 'T', 'k1', 'k2', 'itts', 'alpha', 'damp', 'pr'))
t = np.arange(1, T + 1.0); mom, omom = moms(c)
print(D, N, T, k1, k2, 'itts', itts, 'alpha', alpha, c['kind'], c['link'], c['shift'], 'p', c['p'])
rel = gpu_fuzz.rel
for it in range(1, itts + 1):
    d = damp[:it]
    o = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], None, omom, t, k1, k2, 1, D, N, alpha, d, it)
    o2 = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'] * (1 + 1e-13), None, omom, t, k1, k2, 1, D, N, alpha, d, it)
    print('itts', it, 'oracle self-sensitivity Eft %.2e ttau %.2e' % (rel(o2[0], o[0]), rel(o2[5]['ttau'], o[5]['ttau'])),
          'NaN in oracle ttau/tnu/Eft', int(np.isnan(o[5]['ttau']).sum()), int(np.isnan(o[5]['tnu']).sum()), int(np.isnan(o[0]).sum()),
          'max |ttau| %.2e' % np.nanmax(np.abs(o[5]['ttau'])))
    for env in ({}, {'NAGP_IH_PACK': '0'}, {'NAGP_IH_ROLES': '0'}, {'NAGP_NO_SPARSE': '1'}):
        for k_ in ('NAGP_IH_PACK', 'NAGP_IH_ROLES', 'NAGP_NO_SPARSE'):
            os.environ.pop(k_, None)
        os.environ.update(env)
        r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, k1, k2, 1, D, N, alpha, d, it, nargout=6)
        print('   ', env or 'default', 'Eft %.2e Varft %.2e ttau %.2e nlZ %.2e' % (rel(r[0], o[0]), rel(r[1], o[1]), rel(r[5]['ttau'], o[5]['ttau']), rel(r[5]['nlZ'], o[5]['nlZ'])),
              'NaN gpu ttau/tnu/Eft', int(np.isnan(r[5]['ttau']).sum()), int(np.isnan(r[5]['tnu']).sum()), int(np.isnan(r[0]).sum()))

if len(sys.argv) > 3:      # detail: first step where the default device path and the oracle part (one sweep)
    for k_ in ('NAGP_IH_PACK', 'NAGP_IH_ROLES', 'NAGP_NO_SPARSE'):
        os.environ.pop(k_, None)
    np.set_printoptions(linewidth=200, precision=5)
    d = damp[:1]
    o = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], None, omom, t, k1, k2, 1, D, N, alpha, d, 1)
    r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, k1, k2, 1, D, N, alpha, d, 1, nargout=6)
    a, b = r[5]['tnu'], o[5]['tnu']
    bad = np.where(np.any(np.isnan(a) != np.isnan(b), axis=0) | (np.nanmax(np.abs(a - b), axis=0) > 1e-6 * np.nanmax(np.abs(b))))[0]
    print('first bad step', bad[:5], 'of', T)
    for k in (bad[0] - 1, bad[0]):
        for nm in ('ttau', 'tnu', 'R', 'lZ'):
            v = r[5][nm]; w = o[5][nm]
            print('k=%d %s gpu' % (k, nm), v[..., k] if v.ndim > 1 else v[k]); print('k=%d %s ora' % (k, nm), w[..., k] if w.ndim > 1 else w[k])
