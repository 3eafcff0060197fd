"""Developer probe: the IHGP ADF sweep in the sparse-point form (ihgp_adf_kernel) against the generic kernel
(NAGP_NO_SPARSE=1) on the same inputs.  python tools/gpu_ihgp_ab.py [T] [D N p ...]"""
import os, sys
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np, nagp
from nagp import harness, Mom, _lib as L, ss as ssm

T = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
shapes = [(32, 6, 7), (16, 3, 9), (8, 2, 5), (12, 4, 9), (30, 5, 7), (5, 7, 5)]
if len(sys.argv) > 4:
    v = [int(t) for t in sys.argv[2:]]
    shapes = [tuple(v[i:i + 3]) for i in range(0, len(v) - 2, 3)]
for (D, N, p) in shapes:
    pr = harness.nmf_problem(D, N, T, 1000, 'constraints')
    y = pr['y'].copy()
    y[T // 3: T // 3 + 7] = np.nan            # missing data
    blk = ssm.balance_blocks(ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    res = {}
    for mode in ('generic', 'sparse'):
        if mode == 'generic':
            os.environ['NAGP_NO_SPARSE'] = '1'
        else:
            os.environ.pop('NAGP_NO_SPARSE', None)
        mom = Mom('likModulatorNMFPower', p_cubature=p)
        plan = nagp.Plan(L.KIND_IHGP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=0.5, ep_damping=0.5 * np.ones(3), ep_itts=3)
        plan.upload([y]); plan.execute(); plan.execute(); tm = plan.timings()
        out = plan.download(want_MS=False)[0]
        res[mode] = (out, tm['ms']['filter'] / T * 1e3)
        plan.close()
    a, b = res['generic'][0], res['sparse'][0]

    def rd(u, v):
        m = np.isfinite(u) & np.isfinite(v)
        same = np.array_equal(np.isnan(u), np.isnan(v)) and np.array_equal(np.isinf(u), np.isinf(v))
        return (np.max(np.abs(u[m] - v[m])) / max(np.max(np.abs(u[m])), 1e-300) if m.any() else 0.0), same
    print('D=%d N=%d p=%d npts=%d: ADF us/step generic %.2f sparse %.2f | rel diff Eft %.1e ttau %.1e tnu %.1e lZ %.1e nlZ %.1e | nan/inf pattern same: %s'
          % (D, N, p, mom.tables(N)[0].size, res['generic'][1], res['sparse'][1], rd(a.Eft, b.Eft)[0], rd(a.ttau, b.ttau)[0], rd(a.tnu, b.tnu)[0],
             rd(a.lZ, b.lZ)[0], np.max(np.abs(a.nlZ - b.nlZ) / np.abs(a.nlZ)), all(rd(getattr(a, f), getattr(b, f))[1] for f in ('Eft', 'ttau', 'tnu', 'R', 'lZ'))))
    sys.stdout.flush()
