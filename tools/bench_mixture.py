"""Row f-1 timing: the source-separation inference of experiments/source_sep_piano.m:78-141 -- three sources x
(16 sub-bands, 3 NMF components), exp sub-band kernels, likModulatorPreCalcwn with ut9 in 9 dimensions (3 973 sigma
points), ihgp_ep_mods_nmf_mixture with ep_fraction 0.75, ep_damping 0.025, ep_itts 10, T = 96 000.
python tools/bench_mixture.py [T] [ep_itts] [p_cubature] [check_T] [ihgp|gf]   (gf: the full-covariance gf_ep_mods_nmf_mixture at the same size, 57 sites)
Before the timing, a short prefix (check_T steps, 2 sweeps) is compared with the CPU oracle."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, SSHandle, cubature

T = int(sys.argv[1]) if len(sys.argv) > 1 else 96000
I = int(sys.argv[2]) if len(sys.argv) > 2 else 10
p = int(sys.argv[3]) if len(sys.argv) > 3 else 9
Tc = int(sys.argv[4]) if len(sys.argv) > 4 else 12
variant = sys.argv[5] if len(sys.argv) > 5 else 'ihgp'
fn = nagp.ihgp_ep_mods_nmf_mixture if variant == 'ihgp' else nagp.gf_ep_mods_nmf_mixture
shapes = [(16, 3)] * 3; k1 = ['exp'] * 3; k2 = ['matern52'] * 3
mp = harness.mixture_problem(shapes, T, 3, k1, k2)
wn, xn = cubature.utp_ws(p, 9)
mom = Mom('likModulatorPreCalcwn', link_shift=1.0, wn=wn, xn_unscaled=xn)
t = np.arange(1, T + 1.0)
res = {'workload': variant + '_ep_mods_nmf_mixture, 3 x (D=16, N=3): M=57, S=123, %d sigma points, T=%d, ep_itts=%d' % (wn.size, T, I)}
if Tc > 0:
    from oracle import mixture as omx, lik as olik
    omom = olik.Mom(olik.LIK_POWER_NMF_SQRT, link=olik.softplus_link(1.0), wn=wn, xn_unscaled=xn)
    tc = t[:Tc]; yc = mp['y'][:Tc]
    t0 = time.perf_counter(); b = (omx.ihgp_ep_mods_nmf_mixture if variant == 'ihgp' else omx.gf_ep_mods_nmf_mixture)(mp['w'], tc, yc, None, omom, tc, k1, k2, 3, 0.75, 0.025, 2); to = time.perf_counter() - t0
    a = fn(mp['w'], tc, yc, SSHandle(), mom, tc, k1, k2, 3, 0.75, 0.025, 2, nargout=6)
    res['check'] = {'T': Tc, 'max_rel_Eft': float(np.max(np.abs(a[0] - b[0])) / np.max(np.abs(b[0]))),
                    'max_rel_ttau': float(np.max(np.abs(a[5]['ttau'] - b[5]['ttau'])) / np.max(np.abs(b[5]['ttau']))),
                    'oracle_s_per_step_sweep': to / (Tc * 2)}
fn(mp['w'], t[:64], mp['y'][:64], SSHandle(), mom, t[:64], k1, k2, 3, 0.75, 0.025, 2)   # warm-up
t0 = time.perf_counter()
Eft, Varft = fn(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 3, 0.75, 0.025, I)
dt = time.perf_counter() - t0
res.update({'seconds': dt, 'samples_per_s_per_sweep': T * I / dt, 'finite': bool(np.all(np.isfinite(Eft)))})
if Tc > 0:
    res['vs_oracle'] = res['check']['oracle_s_per_step_sweep'] * T * I / dt
print(json.dumps(res))
