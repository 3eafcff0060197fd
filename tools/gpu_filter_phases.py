"""Developer probe: which phase of the gf filter step costs what -- NAGP_FILTER_DBG skips one phase per run (results are garbage)."""
import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np, nagp
from nagp import harness, Mom, _lib as L, ss as ssm
T = 10000
for (D, N, p) in [(16, 3, 9), (32, 6, 7), (24, 3, 0)]:
    recipe = 'demo_nmf' if D == 16 else 'constraints'
    pr = harness.nmf_problem(D, N, T, 1000, recipe)
    blk = ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    if recipe != 'demo_nmf': blk = ssm.balance_blocks(blk)
    for probe, what in ((0, 'everything'), (1, 'no rank-M update'), (2, 'no PF stores'), (4, 'no congruence'), (8, 'no panel writes'), (16, 'no mean update'), (31, 'none of them')):
        os.environ['NAGP_FILTER_DBG'] = str(probe)
        if p == 0:     # EKF family (cfg4 shape): (1 = the covariance update P -= K S K')
            plan = nagp.Plan(L.KIND_GIEKF, [(ssm.balance_blocks(blk), pr['W'], np.log(pr['w_lik']))], T, ep_itts=3, l_iter=1)
        else:
            plan = nagp.Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=Mom('likModulatorNMFPower', p_cubature=p), ep_fraction=0.5, ep_damping=0.5 * np.ones(3), ep_itts=3)
        plan.upload([pr['y']])
        try:
            plan.execute(); plan.execute()
        except Exception as e:
            pass
        tm = plan.timings()
        if p == 0: print('D=%d EKF %-18s: %.2f us per step' % (D, what, tm['ms']['filter'] / (3 * T) * 1e3))
        else: print('D=%d %-18s: ADF %.2f us, fixed-site %.2f us' % (D, what, tm['ms']['filter'] / (T + 2) * 1e3, tm['ms']['filter_lin'] / (2 * (T - 1)) * 1e3))
        sys.stdout.flush()
        plan.close()
