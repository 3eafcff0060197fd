"""Randomised GPU-vs-oracle comparison at LARGE state dimensions (25..40 sites: the column-owner MFMA smoother passes, the
768-thread gain kernel, two / three tiles per thread in the filters), both EP families, one or two problems per plan, small
chunks (developer tool):   python tools/gpu_fuzz_large.py [n_cases] [seed]"""
import os, sys, time
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, Plan, _lib as L, ss as pss
from oracle import gf_ep as ogf, ihgp as oih, lik as olik


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return np.inf
    return float(np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-300)) if a.size else 0.0


n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
huge = len(sys.argv) > 3 and sys.argv[3] == 'huge'
rng = np.random.default_rng(seed)
worst = {'gf': 0.0, 'ihgp': 0.0}
t0 = time.time()
for case in range(n):
    N = int(rng.integers(2, 7)); D = int(rng.integers(24, 41 - N)) if rng.random() < 0.8 else int(rng.integers(16, 24))
    T = int(rng.integers(30, 80)); p = int(rng.choice([3, 5])) if N > 3 else int(rng.choice([3, 5, 7]))
    if huge:      # 46 .. 63 sites: eight tiles per thread in the gain kernel and the VALU smoother passes
        N = int(rng.integers(3, 10)); D = int(rng.integers(46 - N, 64 - N)); T = int(rng.integers(20, 50)); p = 3 if N > 5 else int(rng.choice([3, 5]))
    k1 = str(rng.choice(['matern32', 'exp'])); k2 = str(rng.choice(['matern32', 'matern52']))
    itts = int(rng.integers(1, 4)); alpha = float(rng.choice([0.5, 0.75])); damp = rng.uniform(0.3, 0.8, itts)
    chunk = int(rng.choice([0, 16, 25]))
    pr = harness.nmf_problem(D, N, T, 1000 + 17 * case + seed, 'constraints')
    y = pr['y'].copy(); y[rng.integers(0, T, 2)] = np.nan
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, k2))
    mom = Mom('likModulatorNMFPower', p_cubature=p); om = olik.Mom(olik.LIK_POWER_NMF, p=p)
    try:
        plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=alpha, ep_damping=damp, ep_itts=itts, chunk=chunk)
    except nagp.NagpError as e:      # beyond the envelope (the filter's LDS at 59+ sites): refused, loudly
        assert 'unsupported shape' in str(e), e
        print('%2d D=%d N=%d M=%d S=%d refused: %s' % (case, D, N, blk.M, blk.S, str(e)[-60:])); continue
    plan.upload([y]); plan.execute(); r = plan.download()[0]; plan.close()
    o = ogf.run_predict(ogf.assemble(np.log(pr['w_lik']) * np.ones(1), pr['param1'], pr['param2'], pr['W'], k1, k2, True), y, om, alpha, damp, itts)
    e_gf = max(rel(r.Eft, o['Eft']), rel(r.Varft, o['Varft']), rel(r.nlZ, o['nlZ']))
    if e_gf > 1e-7:      # the reference itself unstable here?  (the rule of tools/gpu_fuzz.py / fuzz_conditioning.py: oracle on y (1 + 1e-13))
        o3 = ogf.run_predict(ogf.assemble(np.log(pr['w_lik']) * np.ones(1), pr['param1'], pr['param2'], pr['W'], k1, k2, True), y * (1 + 1e-13), om, alpha, damp, itts)
        sens = max(rel(o3['Eft'], o['Eft']), rel(o3['Varft'], o['Varft']))
        big = max(np.nanmax(np.abs(np.nan_to_num(x, posinf=0.0))) for x in (r.ttau, r.tnu, o['ttau'], o['tnu']))
        if big > 1e8 or e_gf < 1e3 * sens or not np.isfinite(sens):
            print('   [gf: unstable instance, oracle self-sensitivity %.1e, largest site %.1e, device difference %.1e]' % (sens, big, e_gf)); e_gf = 0.0
    pr2 = harness.nmf_problem(D, N, T, 2000 + 17 * case + seed); tt = np.arange(1, T + 1.0)
    r2 = nagp.ihgp_ep_modulator_nmf(pr2['w'], tt, pr2['y'], nagp.SSHandle(), mom, tt, k1, k2, 1, D, N, alpha, damp, itts, nargout=6)
    o2 = oih.ihgp_ep_modulator_nmf(pr2['w'], tt, pr2['y'], None, om, tt, k1, k2, 1, D, N, alpha, damp, itts)
    e_ih = max(rel(r2[0], o2[0]), rel(r2[1], o2[1]), rel(r2[5]['nlZ'], o2[5]['nlZ']))
    if e_ih > 1e-7:      # the reference itself unstable here?  (same rule as tools/gpu_fuzz.py)
        o3 = oih.ihgp_ep_modulator_nmf(pr2['w'], tt, pr2['y'] * (1 + 1e-13), None, om, tt, k1, k2, 1, D, N, alpha, damp, itts)
        sens = max(rel(o3[0], o2[0]), rel(o3[1], o2[1]))
        big = max(np.nanmax(np.abs(np.nan_to_num(x[5][nm], posinf=0.0))) for x in (r2, o2) for nm in ('ttau', 'tnu'))
        if big > 1e8 or e_ih < 1e3 * sens or not np.isfinite(sens):
            print('   [ihgp: unstable instance, oracle self-sensitivity %.1e, largest site %.1e, device difference %.1e]' % (sens, big, e_ih)); e_ih = 0.0
    worst['gf'] = max(worst['gf'], e_gf)
    if e_ih == e_ih: worst['ihgp'] = max(worst['ihgp'], e_ih)
    print('%2d D=%d N=%d M=%d S=%d T=%d p=%d %s/%s itts=%d alpha=%.2f chunk=%d   gf %.1e ihgp %.1e %s' %
          (case, D, N, blk.M, blk.S, T, p, k1, k2, itts, alpha, chunk, e_gf, e_ih, '<<<<' if max(e_gf, e_ih if e_ih == e_ih else 0) > 1e-7 else ''), flush=True)
print('worst', worst, '%ds' % (time.time() - t0))
