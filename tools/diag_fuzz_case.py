"""Re-run ONE draw of tools/gpu_fuzz.py (main family) for the EKF functions, sweep by sweep, with the oracle's own sensitivity
beside the device-vs-oracle difference (developer tool):   python tools/diag_fuzz_case.py [seed] [index]"""
import os, sys
ROOT = '/root/repo' if os.path.exists('/root/repo/tools') else os.environ.get('GRAFT_REPO_ROOT', '.')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import nagp
from nagp import SSHandle
from oracle import giekf as oek
from gpu_fuzz_draws import draw, moms
import gpu_fuzz
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 4242
index = int(sys.argv[2]) if len(sys.argv) > 2 else 36
rng = np.random.default_rng(seed)
c = None
for i in range(index + 1):
    c = draw(rng)
D, N, T, k1, k2, itts, pr, y = (c[k] for k in ('D', 'N', 'T', 'k1', 'k2', 'itts', 'pr', 'y'))
print(D, N, T, k1, k2, itts, c['li'], 'nan in y', int(np.isnan(y).sum()))
t = np.arange(1, T + 1.0)
def rel(a, b): return gpu_fuzz.rel(a, b)
for gi in range(1, itts + 1):
    r = nagp.gf_giekf_modulator_nmf(pr['w'], t, y, SSHandle(), None, t, k1, k2, 1, D, N, gi, c['li'], nargout=6)
    o = oek.gf_giekf_modulator_nmf(pr['w'], t, y, None, None, t, k1, k2, 1, D, N, gi, c['li'])
    o2 = oek.gf_giekf_modulator_nmf(pr['w'], t, y * (1 + 1e-13), None, None, t, k1, k2, 1, D, N, gi, c['li'])
    print('g_iter', gi, 'gpu-vs-oracle Eft %.2e Varft %.2e | oracle self-sensitivity Eft %.2e Varft %.2e' % (rel(r[0], o[0]), rel(r[1], o[1]), rel(o2[0], o[0]), rel(o2[1], o[1])),
          'counters', r[5].get('counters') if isinstance(r[5], dict) else None)
