"""GPU vs oracle for one gf_ep_modulator_nmf shape, filter and smoother outputs apart: python tools/diag_shape.py D N [T] [p] [itts] [chunk]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, Plan, _lib as L, ss as pss
from oracle import gf_ep as ogf, lik as olik
def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    return float(np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-300))
D, N = int(sys.argv[1]), int(sys.argv[2])
T = int(sys.argv[3]) if len(sys.argv) > 3 else 24; p = int(sys.argv[4]) if len(sys.argv) > 4 else 3
itts = int(sys.argv[5]) if len(sys.argv) > 5 else 1; chunk = int(sys.argv[6]) if len(sys.argv) > 6 else 0
pr = harness.nmf_problem(D, N, T, 4242, 'constraints')
blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'exp', 'matern32'))
mom = Mom('likModulatorNMFPower', p_cubature=p); om = olik.Mom(olik.LIK_POWER_NMF, p=p)
damp = 0.5 * np.ones(itts)
plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=0.5, ep_damping=damp, ep_itts=itts, chunk=chunk)
plan.upload([pr['y']]); plan.execute(); r = plan.download(want_MF=True)[0]; plan.close()
o = ogf.run_predict(ogf.assemble(np.log(pr['w_lik']) * np.ones(1), pr['param1'], pr['param2'], pr['W'], 'exp', 'matern32', True), pr['y'], om, 0.5, damp, itts)
print('D=%d N=%d M=%d S=%d T=%d p=%d itts=%d chunk=%d' % (D, N, blk.M, blk.S, T, p, itts, chunk),
      ' '.join('%s %.1e' % (f, rel(getattr(r, f), o[f])) for f in ('MF', 'lZ', 'ttau', 'MS', 'Eft', 'Varft') if f in o))
if 'MF' in o:
    e = np.abs(r.MF - o['MF']).max(axis=0); print('   first step with MF error > 1e-8:', (np.nonzero(e > 1e-8 * np.abs(o['MF']).max())[0][:1]))
