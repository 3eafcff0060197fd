"""Developer probe: per-step site differences of the two device cubature forms against the oracle (one IHGP ADF sweep)."""
import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np, nagp
from nagp import harness, Mom, SSHandle
from oracle import ihgp as oih, lik as olik
D, N, p, T = [int(v) for v in sys.argv[1:5]]
pr = harness.nmf_problem(D, N, T, 900 + D, 'constraints'); t = np.arange(1, T + 1.0)
y = pr['y'].copy(); y[20:23] = np.nan
mom = Mom('likModulatorNMFPower', p_cubature=p); d = np.array([0.5])
res = {}
for mode in ('generic', 'sparse'):
    if mode == 'generic': os.environ['NAGP_NO_SPARSE'] = '1'
    else: os.environ.pop('NAGP_NO_SPARSE', None)
    res[mode] = nagp.ihgp_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 1, nargout=6)[5]
ref = oih.ihgp_ep_modulator_nmf(pr['w'], t, y, None, olik.Mom(olik.LIK_POWER_NMF, p=p), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 1)[5]
for k in range(T):
    a, b, r = res['generic']['ttau'][:, k], res['sparse']['ttau'][:, k], ref['ttau'][:, k]
    da = np.nanmax(np.abs(a - r)); db = np.nanmax(np.abs(b - r)); i = int(np.nanargmax(np.abs(b - r)))
    if k < 6 or db > 1e-9 * np.nanmax(np.abs(r)):
        print('k=%3d |ttau|max %.3e  generic-oracle %.2e  sparse-oracle %.2e (site %d: ttau %.6e R %.3e)' % (
            k, np.nanmax(np.abs(r)), da, db, i, r[i], ref['R'][i, k]))
