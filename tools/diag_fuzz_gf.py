"""Re-run ONE draw of tools/gpu_fuzz.py (main family) through gf_ep_modulator_nmf, sweep by sweep, with the MFMA gain kernel and with the
4x4-tile VALU gain kernel (NAGP_NO_GAIN_MFMA=1), beside the oracle and the oracle's own sensitivity (developer tool):
    python tools/diag_fuzz_gf.py [seed] [index]"""
import os, sys, subprocess, json
os.environ.setdefault('NAGP_DEVELOPER', '1')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 20270105
index = int(sys.argv[2]) if len(sys.argv) > 2 else 58
if len(sys.argv) > 3 and sys.argv[3] == 'child':
    import nagp
    from nagp import SSHandle
    from oracle import gf_ep as ogf
    from gpu_fuzz_draws import draw, moms
    import gpu_fuzz
    rng = np.random.default_rng(seed)
    for i in range(index + 1):
        c = draw(rng)
    D, N, T, p, k1, k2, itts, alpha, damp, pr, y = (c[k] for k in ('D', 'N', 'T', 'p', 'k1', 'k2', 'itts', 'alpha', 'damp', 'pr', 'y'))
    mom, omom = moms(c)
    t = np.arange(1, T + 1.0)
    print(os.environ.get('NAGP_NO_GAIN_MFMA', '0'), 'D N T', D, N, T, k1, k2, 'itts', itts, 'alpha', alpha, 'damp', damp, 'nan in y', int(np.isnan(y).sum()))
    for it in range(1, itts + 1):
        r = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, k1, k2, 1, D, N, alpha, damp[:it], it, nargout=6)
        o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, omom, t, k1, k2, 1, D, N, alpha, damp[:it], it)
        o2 = ogf.gf_ep_modulator_nmf(pr['w'], t, y * (1 + 1e-13), None, omom, t, k1, k2, 1, D, N, alpha, damp[:it], it)
        rel = gpu_fuzz.rel
        print(' sweeps %d: gpu-oracle Eft %.2e Varft %.2e ttau %.2e nlZ %.2e | oracle moves Eft %.2e Varft %.2e ttau %.2e | max|ttau| gpu %.2e oracle %.2e | counters %s | oracle retries %s'
              % (it, rel(r[0], o[0]), rel(r[1], o[1]), rel(r[5]['ttau'], o[5]['ttau']), rel(r[5]['nlZ'], o[5]['nlZ']), rel(o2[0], o[0]), rel(o2[1], o[1]), rel(o2[5]['ttau'], o[5]['ttau']),
                 np.nanmax(np.abs(r[5]['ttau'])), np.nanmax(np.abs(o[5]['ttau'])), r[5].get('counters'), o[5].get('counters', o[5].get('n_retry'))))
    sys.exit(0)
for mode in ('0', '1'):
    env = dict(os.environ)
    if mode == '1':
        env['NAGP_NO_GAIN_MFMA'] = '1'
    print(subprocess.run([sys.executable, __file__, str(seed), str(index), 'child'], env=env, capture_output=True, text=True).stdout)
