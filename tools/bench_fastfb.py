"""Stationary filterbank timing: python tools/bench_fastfb.py [D] [T] [kernel]
GPU = nagp.kernel_ss_kalmanFastFB (steady-state set-up on the host + nagp_fastfb_run incl. PCIe), CPU = oracle restatement."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from oracle import fastfb as offb

D = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T = int(sys.argv[2]) if len(sys.argv) > 2 else 84010
kernel = sys.argv[3] if len(sys.argv) > 3 else 'exp'
rng = np.random.default_rng(0)
lam = 1.0 / rng.uniform(20, 400, D); var = rng.uniform(0.1, 1.0, D); om = np.linspace(np.pi / 3, np.pi / 50, D)
A, Q, H, Pinf, K, tau1 = nagp.get_disc_model(lam, var, om, D, kernel, 6)
y = rng.normal(size=T); y[1000:1200] = np.nan
nagp.kernel_ss_kalmanFastFB(A, Q, H, Pinf, K, 0.01, y[:100], 0, 0, steady=True)      # warm-up (library load, context)
t0 = time.perf_counter(); lik, X, P = nagp.kernel_ss_kalmanFastFB(A, Q, H, Pinf, K, 0.01, y, 0, 0, steady=True); t_gpu = time.perf_counter() - t0
nc = min(T, 20000)
t0 = time.perf_counter(); lo, MSo, _, _ = offb.kernel_ss_kalmanFastFB(A, Q, H, Pinf, K, 0.01, y[:nc], 0, 0); t_cpu = (time.perf_counter() - t0) * T / nc
S = A.shape[0]
print(json.dumps({'workload': 'kernel_ss_kalmanFastFB, %s kernel, D=%d (S=%d), T=%d, filter + smoother' % (kernel, D, S, T),
                  'gpu_s_incl_setup_and_pcie': t_gpu, 'samples_per_s': T / t_gpu,
                  'cpu_oracle_s_extrapolated': t_cpu, 'speedup': t_cpu / t_gpu,
                  'algorithmic_bytes': 8.0 * T * (1 + 3 * S), 'lik': lik}))
