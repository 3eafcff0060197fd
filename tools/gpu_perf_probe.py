"""Developer probe: time the kernels at BASELINE shapes (no oracle). python tools/gpu_perf_probe.py [cfg] [T] [B]"""
import os, sys, time
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, _lib as L
from nagp import ss as ssm

cfg = sys.argv[1] if len(sys.argv) > 1 else 'cfg2'
T = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
B = int(sys.argv[3]) if len(sys.argv) > 3 else 1
LIK = sys.argv[4] if len(sys.argv) > 4 else 'nmf'      # 'nmf' = likModulatorNMFPower, 'sqrt' = likModulatorPreCalcwn
shapes = dict(cfg2=(16, 3, 9, 'demo_nmf', L.KIND_GF_EP), cfg3=(32, 6, 7, 'constraints', L.KIND_IHGP),
              cfg5=(32, 6, 7, 'constraints', L.KIND_GF_EP), cfg4=(24, 3, 9, 'demo_nmf', L.KIND_GIEKF))
D, N, p, recipe, kind = shapes[cfg]
t0 = time.time()
probs, ys = [], []
for q in range(B):
    pr = harness.nmf_problem(D, N, T, 1000 + q, recipe)
    blk = ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    if kind != L.KIND_GF_EP or cfg == 'cfg5':
        blk = ssm.balance_blocks(blk)
    probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(pr['y'])
print('setup %.1fs  S=%d M=%d' % (time.time() - t0, probs[0][0].S, probs[0][0].M)); sys.stdout.flush()
if LIK == 'sqrt':
    from nagp import cubature
    wn, xn = cubature.utp_ws(p, N)
    mom = Mom('likModulatorPreCalcwn', wn=wn, xn_unscaled=xn)
else:
    mom = Mom('likModulatorNMFPower', p_cubature=p)
t0 = time.time()
plan = nagp.Plan(kind, probs, T, mom=mom if kind != L.KIND_GIEKF else None, ep_fraction=0.5, ep_damping=0.5 * np.ones(3), ep_itts=3, l_iter=1)
print('plan create %.1fs, device MB %.1f' % (time.time() - t0, plan.device_bytes() / 1e6)); sys.stdout.flush()
plan.upload(ys)
for rep in range(2):
    t0 = time.time(); plan.execute(); dt = time.time() - t0
    tm = plan.timings()
    print('execute %.3fs  -> %.0f samples/s per sweep' % (dt, B * T * 3 / dt))
    print('  ', {k: (round(v, 2), tm['launches'][k]) for k, v in tm['ms'].items() if tm['launches'][k]}, 'total_ms', round(tm['total_ms'], 2))
    sys.stdout.flush()
outs = plan.download(want_MS=False)
print('nlZ', outs[0].nlZ, 'counters', outs[0].counters)
