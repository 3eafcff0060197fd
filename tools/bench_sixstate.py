"""Time one execute of a plan with Matern-5/2 sub-bands (six-state blocks: split tile rows / block stride 8) beside the same shape with
Matern-3/2 sub-bands (the tuned kernels):   python tools/bench_sixstate.py [D N T]      (developer tool, GPU box)"""
import os, sys, time
os.environ.setdefault('NAGP_DEVELOPER', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
from nagp import harness, Mom, Plan, _lib as L, ss as pss
D, N, T = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (16, 3, 20000)
mom = Mom('likModulatorNMFPower', p_cubature=7)
for kind, name in ((L.KIND_GF_EP, 'gf_ep_modulator_nmf'), (L.KIND_IHGP, 'ihgp_ep_modulator_nmf'), (L.KIND_GIEKF, 'gf_giekf_modulator_nmf')):
    for k1 in ('matern32', 'matern52'):
        pr = harness.nmf_problem(D, N, T, 5, kernel1=k1)
        blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, 'matern52')
        if kind == L.KIND_IHGP: blk = pss.balance_blocks(blk)
        kw = dict(l_iter=1, ep_itts=3) if kind == L.KIND_GIEKF else dict(mom=mom, ep_fraction=0.5, ep_damping=[0.1] * 3, ep_itts=3)
        plan = Plan(kind, [(blk, pr['W'], np.log(pr['w_lik']))], T, **kw)
        plan.upload([pr['y']]); plan.execute(); plan.execute()
        t = plan.timings()
        print('%-24s D=%d N=%d T=%d kernel1=%-8s S=%3d  %8.1f ms per execute (3 sweeps) = %7.0f k samples/s per sweep   %s' % (
            name, D, N, T, k1, blk.S, t['total_ms'], 3 * T / t['total_ms'], {k: round(v) for k, v in t['ms'].items() if v >= 1}), flush=True)
        plan.close()
