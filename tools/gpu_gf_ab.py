"""Developer probe: gf_ep ADF launches in the sparse-point form against the generic mom (NAGP_NO_SPARSE=1) on the same inputs."""
import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np, nagp
from nagp import harness, Mom, _lib as L, ss as ssm
T = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
v = [int(t) for t in sys.argv[2:]]
shapes = [tuple(v[i:i + 3]) for i in range(0, len(v) - 2, 3)] or [(16, 3, 9), (32, 6, 7)]
for (D, N, p) in shapes:
    recipe = 'demo_nmf' if D == 16 else 'constraints'
    pr = harness.nmf_problem(D, N, T, 1000, recipe)
    y = pr['y'].copy(); y[T // 3: T // 3 + 5] = np.nan
    blk = ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    if recipe != 'demo_nmf': blk = ssm.balance_blocks(blk)
    res = {}
    for mode in ('generic', 'sparse'):
        if mode == 'generic': os.environ['NAGP_NO_SPARSE'] = '1'
        else: os.environ.pop('NAGP_NO_SPARSE', None)
        plan = nagp.Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=Mom('likModulatorNMFPower', p_cubature=p), ep_fraction=0.5, ep_damping=0.5 * np.ones(3), ep_itts=3)
        plan.upload([y]); plan.execute(); plan.execute(); tm = plan.timings()
        res[mode] = (plan.download(want_MS=False)[0], tm['ms']['filter'] / (T + 2) * 1e3, tm['ms']['filter_lin'] / (2 * (T - 1)) * 1e3); plan.close()
    a, b = res['generic'][0], res['sparse'][0]
    def rd(u, w):
        m = np.isfinite(u) & np.isfinite(w)
        return np.max(np.abs(u[m] - w[m])) / max(np.max(np.abs(u[m])), 1e-300)
    print('D=%d N=%d p=%d: ADF us/step generic %.2f sparse %.2f (fixed-site %.2f) | rel diff Eft %.1e Varft %.1e ttau %.1e tnu %.1e nlZ %.1e' % (
        D, N, p, res['generic'][1], res['sparse'][1], res['sparse'][2], rd(a.Eft, b.Eft), rd(a.Varft, b.Varft), rd(a.ttau, b.ttau), rd(a.tnu, b.tnu), np.max(np.abs(a.nlZ - b.nlZ) / np.abs(a.nlZ))))
    sys.stdout.flush()
