"""Developer probe: per-step divergence between the generic and the sparse-point IHGP ADF sweep (one sweep)."""
import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np, nagp
from nagp import harness, Mom, _lib as L, ss as ssm
T = int(sys.argv[1]); v = [int(t) for t in sys.argv[2:]]
for (D, N, p) in [tuple(v[i:i + 3]) for i in range(0, len(v) - 2, 3)]:
    pr = harness.nmf_problem(D, N, T, 1000, 'constraints')
    blk = ssm.balance_blocks(ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    res = {}
    for mode in ('generic', 'sparse'):
        if mode == 'generic': os.environ['NAGP_NO_SPARSE'] = '1'
        else: os.environ.pop('NAGP_NO_SPARSE', None)
        plan = nagp.Plan(L.KIND_IHGP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=Mom('likModulatorNMFPower', p_cubature=p), ep_fraction=0.5, ep_damping=0.5 * np.ones(1), ep_itts=1)
        plan.upload([pr['y']]); plan.execute(); res[mode] = plan.download(want_MS=False)[0]; plan.close()
    a, b = res['generic'], res['sparse']
    print('D=%d N=%d p=%d' % (D, N, p))
    for k in list(range(0, 12)) + list(range(12, T, max(1, T // 12))):
        dt = np.abs(a.ttau[:, k] - b.ttau[:, k]); dn = np.abs(a.tnu[:, k] - b.tnu[:, k])
        print('  k=%4d lZ %.17g / %.17g  max|dttau| %.2e (|ttau| %.2e)  max|dtnu| %.2e (|tnu| %.2e) argmax site %d  clamped %d/%d' % (
            k, a.lZ[k], b.lZ[k], np.nanmax(dt), np.nanmax(np.abs(a.ttau[:, k])), np.nanmax(dn), np.nanmax(np.abs(a.tnu[:, k])), int(np.nanargmax(dt)),
            int(np.sum(a.ttau[:, k] == 0)), int(np.sum(b.ttau[:, k] == 0))))
