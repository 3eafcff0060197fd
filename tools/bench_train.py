"""Row f-3 timing: one fminunc iteration's objective evaluations (numel(w)+1 replicas) batched vs serial.
python tools/bench_train.py [D] [N] [T]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, SSHandle

D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
N = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = int(sys.argv[3]) if len(sys.argv) > 3 else 20000
pr = harness.nmf_problem(D, N, T, 7, 'constraints'); t = np.arange(1, T + 1.0)
cons = harness.CONSTRAINTS_DEMO(D); w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
mom = Mom('likModulatorNMFPower', p_cubature=7); d = np.array([0.1])
args = (t, pr['y'], SSHandle(), mom, 'matern32', 'matern52', 1, D, N, 0.5, d, 1)
kw = dict(constraints=cons, w_fixed=wf, tune_hypers=harness.TUNE_DEMO)
nagp.nlml_batch([w], *args, **kw)                                   # warm-up
t0 = time.perf_counter(); f0, g = nagp.fd_value_and_gradient(w, *args, **kw); tb = time.perf_counter() - t0
t0 = time.perf_counter()
fs = nagp.gf_ep_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), mom, None, 'matern32', 'matern52', 1, D, N, 0.5, d, 1, cons, wf, harness.TUNE_DEMO)[0]
ts = time.perf_counter() - t0
print(json.dumps({'workload': 'gf_ep_modulator_nmf_constraints nlml, D=%d N=%d T=%d, ep_itts=1, %d replicas (forward differences)' % (D, N, T, w.size + 1),
                  'batched_s': tb, 'one_serial_call_s': ts, 'serial_equivalent_s': ts * (w.size + 1), 'speedup': ts * (w.size + 1) / tb,
                  'f0': f0, 'f0_serial': fs, 'grad_norm': float(np.linalg.norm(g))}))
