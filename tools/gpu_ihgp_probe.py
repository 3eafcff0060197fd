import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np, nagp
from nagp import harness, Mom, _lib as L, ss as ssm
T = 20000
for (D, N, p) in [(32, 6, 3), (32, 6, 5), (32, 6, 7), (32, 6, 9), (16, 3, 7), (4, 2, 7)]:
    pr = harness.nmf_problem(D, N, T, 1000, 'constraints')
    blk = ssm.balance_blocks(ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    mom = Mom('likModulatorNMFPower', p_cubature=p)
    plan = nagp.Plan(L.KIND_IHGP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=0.5, ep_damping=0.5 * np.ones(3), ep_itts=3)
    plan.upload([pr['y']]); plan.execute(); plan.execute(); tm = plan.timings()
    print('D=%d N=%d p=%d npts=%d: ADF filter %.2f us/step, lin filter %.2f us/step, scan %.2f us/step, ep %.3f ms' % (
        D, N, p, mom.tables(N)[0].size, tm['ms']['filter'] / T * 1e3, tm['ms']['filter_lin'] / 2 / T * 1e3, tm['ms']['scan'] / 3 / T * 1e3, tm['ms']['epsite'] / 2))
    plan.close()
