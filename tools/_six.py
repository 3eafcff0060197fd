import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, SSHandle
from oracle import gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik
def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.nanmax(np.abs(a - b)) / max(np.nanmax(np.abs(b)), 1e-300))
fam = sys.argv[1] if len(sys.argv) > 1 else 'ihgp'
for (D, N, T, k1, itts) in [(2, 1, 60, 'matern52', 1), (8, 3, 300, 'matern52', 3), (3, 2, 200, 'matern72', 2), (8, 3, 1, 'matern52', 2)]:
    pr = harness.nmf_problem(D, N, T, 11, kernel1=k1); t = np.arange(1, T + 1.0)
    damp = [0.5] * itts
    if fam == 'ihgp':
        r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), Mom('likModulatorNMFPower', p_cubature=7), t, k1, 'matern52', 1, D, N, 0.5, damp, itts, nargout=6)
        o = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], None, olik.Mom(olik.LIK_POWER_NMF, p=7), t, k1, 'matern52', 1, D, N, 0.5, damp, itts)
    elif fam == 'gf':
        r = nagp.gf_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), Mom('likModulatorNMFPower', p_cubature=7), t, k1, 'matern52', 1, D, N, 0.5, damp, itts, nargout=6)
        o = ogf.gf_ep_modulator_nmf(pr['w'], t, pr['y'], None, olik.Mom(olik.LIK_POWER_NMF, p=7), t, k1, 'matern52', 1, D, N, 0.5, damp, itts)
    print(fam, D, N, T, k1, itts, 'Eft %.2e Varft %.2e ttau %.2e tnu %.2e nlZ %.2e' % (rel(r[0], o[0]), rel(r[1], o[1]), rel(r[5]['ttau'], o[5]['ttau']), rel(r[5]['tnu'], o[5]['tnu']), rel(r[5]['nlZ'], o[5]['nlZ'])), flush=True)
