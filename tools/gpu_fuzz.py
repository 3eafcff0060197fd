"""Randomised GPU-vs-oracle comparison over shapes, likelihoods, links, cubature orders, missing data and all three
function families (developer tool; the fixed-seed subset in tests/ is what the suite runs).
python tools/gpu_fuzz.py [n_cases] [seed]
python tools/gpu_fuzz.py widened [n_cases] [seed]     # mixtures (block-structured cubature included) and the EKF objective"""
import os, sys, time
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np
import nagp
from nagp import harness, Mom, SSHandle, cubature
from oracle import gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik, mixture as omx


NOT_COMPARED = -1.0      # both sides entirely NaN: there is nothing to compare (counted and printed, never taken for agreement)


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return np.inf
    if a.size and np.all(np.isnan(b)):
        return NOT_COMPARED
    return float(np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-300)) if a.size else 0.0


def tally(res, worst, skipped):
    """flag of one draw; the worst figures ignore NOT_COMPARED entries, which are counted per key instead"""
    bad = False
    for k, v in res.items():
        if v == NOT_COMPARED:
            skipped[k] = skipped.get(k, 0) + 1
        else:
            worst[k] = max(worst.get(k, 0.0), v)
            bad = bad or not (v <= 1e-7)          # (NaN fails too)
    return bad


def fmt(res):
    return ' '.join('%s %s' % (k, 'all-NaN:not-compared' if v == NOT_COMPARED else '%.1e' % v) for k, v in res.items())


from gpu_fuzz_draws import draw, moms, draw_widened   # the draws live in a GPU-free module (tools/fuzz_conditioning.py shares them)


def weak_checks(r, o):
    """What still holds for a draw the oracle itself is unstable on (tests assert these for the excused draws): the device
    returns NaN exactly where the oracle does (means, variances, sites), and the sites the rule clamps are non-negative."""
    with np.errstate(all='ignore'):
        same_nan = all(np.array_equal(np.isnan(np.asarray(a, float)), np.isnan(np.asarray(b, float)))
                       for a, b in ((r[0], o[0]), (r[1], o[1]), (r[5]['ttau'], o[5]['ttau']), (r[5]['tnu'], o[5]['tnu'])))
        tt = np.asarray(r[5]['ttau'], float)
        nonneg = bool(np.all(tt[~np.isnan(tt)] >= 0))
    return dict(same_nan_pattern=bool(same_nan), ttau_nonneg=nonneg)


def one(rng, raw=False, k1_override=None):
    """raw=True: plain device-vs-oracle differences, nothing excused (the tests decide from the committed lists of
    tests/golden/fuzz_excused_*.json, which tools/fuzz_conditioning.py derives from the oracle alone)"""
    c = draw(rng, k1_override)
    D, N, T, p, k1, k2, itts, alpha, damp, pr, y = (c[k] for k in ('D', 'N', 'T', 'p', 'k1', 'k2', 'itts', 'alpha', 'damp', 'pr', 'y'))
    # blocks of six / eight states: the infinite-horizon look-up tables of the host and of the oracle come from two DARE solvers and agree to
    # 1e-8 .. 1e-5 only (steady-state covariances conditioned 1e8 .. 1e12) -- on such draws the oracle runs on the HOST's tables (the kernels are
    # what is compared; key 'ihgp'), and the difference to the oracle's own tables is printed beside it
    tol_ih = 1e-7
    t = np.arange(1, T + 1.0)
    mom, omom = moms(c)
    desc = 'D=%d N=%d T=%d p=%d %s %s(%g) %s/%s itts=%d alpha=%.2f' % (D, N, T, p, c['kind'], c['link'], c['shift'], k1, k2, itts, alpha)
    res = {}
    r = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, k1, k2, 1, D, N, alpha, damp, itts, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, omom, t, k1, k2, 1, D, N, alpha, damp, itts)
    res['gf'] = max(rel(r[0], o[0]), rel(r[1], o[1]), rel(r[5]['nlZ'], o[5]['nlZ']))
    c['weak'] = {'gf': weak_checks(r, o)}
    if res['gf'] > 1e-7 and not raw:
        o2 = ogf.gf_ep_modulator_nmf(pr['w'], t, y * (1 + 1e-13), None, omom, t, k1, k2, 1, D, N, alpha, damp, itts)
        sens = max(rel(o2[0], o[0]), rel(o2[1], o[1]))
        big = max(np.nanmax(np.abs(np.nan_to_num(x[5][nm], posinf=0.0))) for x in (r, o) for nm in ('ttau', 'tnu'))
        if big > 1e8:
            sens = max(sens, 1.0)
        if NOT_COMPARED in (rel(o2[0], o[0]), rel(o2[1], o[1])):      # the reference algorithm itself ends in NaN everywhere on this draw: nothing to hold the device against
            desc += ' [gf: the oracle\'s own means or variances are all NaN]'; res['gf'] = NOT_COMPARED
        elif res['gf'] < 1e3 * sens or not np.isfinite(sens):
            desc += ' [gf: unstable instance, oracle self-sensitivity %.1e, device difference %.1e]' % (sens, res['gf']); res['gf'] = 0.0
    yi = pr['y']     # IHGP has no NaN test on y (C-3): feed complete data
    r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, yi, SSHandle(), mom, t, k1, k2, 1, D, N, alpha, damp, itts, nargout=6)
    o = oih.ihgp_ep_modulator_nmf(pr['w'], t, yi, None, omom, t, k1, k2, 1, D, N, alpha, damp, itts)
    if k1_override:
        own = max(rel(r[0], o[0]), rel(r[1], o[1]), rel(r[5]['nlZ'], o[5]['nlZ']))
        from gpu_fuzz_draws import oracle_ihgp_on_host_tables
        oh = oracle_ihgp_on_host_tables(pr['w'], yi, omom, k1, k2, D, N, alpha, damp, itts)
        o = (oh['Eft'], oh['Varft'], None, None, None, oh)
        desc += ' [ihgp against the oracle\'s own tables: %.1e]' % own
    res['ihgp'] = max(rel(r[0], o[0]), rel(r[1], o[1]), rel(r[5]['nlZ'], o[5]['nlZ']))
    c['weak']['ihgp'] = weak_checks(r, o)
    if res['ihgp'] > tol_ih and not raw:
        # is the instance itself unstable?  (site updates -d2/(1+d2*v) with 1+d2*v ~ 0 under full-EP cavities, or an
        # arg-min over the R grid sitting on a midpoint: the reference's own result then moves by percents under a
        # 1e-13 relative change of y, and there is nothing to compare)
        o2 = oih.ihgp_ep_modulator_nmf(pr['w'], t, yi * (1 + 1e-13), None, omom, t, k1, k2, 1, D, N, alpha, damp, itts)
        sens = max(rel(o2[0], o[0]), rel(o2[1], o[1]))
        big = max(np.nanmax(np.abs(np.nan_to_num(x[5][nm], posinf=0.0))) for x in (r, o) for nm in ('ttau', 'tnu'))
        if big > 1e8:      # a site update divided by 1 + d2*v ~ 1e-14: its sign and size are rounding noise in the reference too
            sens = max(sens, 1.0)
        # a refreshed site of underflow / cancellation size (0 < |ttau| < 1e-9 of the scale): its sign decides between R = inf
        # (smoother look-up: LAST grid row, ihgp_ep_modulator_nmf.m:382) and R ~ 1e30 (all |r - R| tie: FIRST row, C-24)
        tts = [x[5]['ttau'] for x in (r, o)]
        if any(np.any((np.abs(v) > 0) & (np.abs(v) < 1e-9 * np.nanmax(np.abs(v)))) for v in tts):
            sens = max(sens, 1.0)
        if NOT_COMPARED in (rel(o2[0], o[0]), rel(o2[1], o[1])):
            desc += ' [ihgp: the oracle\'s own means or variances are all NaN]'; res['ihgp'] = NOT_COMPARED
        elif res['ihgp'] < 1e3 * sens or not np.isfinite(sens):
            desc += ' [ihgp: unstable instance, oracle self-sensitivity %.1e, device difference %.1e]' % (sens, res['ihgp']); res['ihgp'] = 0.0
    if c['li']:
        r = nagp.gf_giekf_modulator_nmf(pr['w'], t, y, SSHandle(), None, t, k1, k2, 1, D, N, itts, c['li'], nargout=2)
        o = oek.gf_giekf_modulator_nmf(pr['w'], t, y, None, None, t, k1, k2, 1, D, N, itts, c['li'])
        res['giekf'] = max(rel(r[0], o[0]), rel(r[1], o[1]))
    return desc, res, c


def diagnose_ihgp(c):
    """sweep by sweep: where do the device and the oracle part, and how sensitive is the oracle itself there"""
    np.set_printoptions(linewidth=220, precision=6)
    D, N, T, k1, k2, alpha, pr = (c[k] for k in ('D', 'N', 'T', 'k1', 'k2', 'alpha', 'pr'))
    t = np.arange(1, T + 1.0); mom, omom = moms(c)
    for itts in range(1, c['itts'] + 1):
        d = c['damp'][:itts]
        r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, k1, k2, 1, D, N, alpha, d, itts, nargout=6)
        o = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], None, omom, t, k1, k2, 1, D, N, alpha, d, itts)
        o2 = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'] * (1 + 1e-13), None, omom, t, k1, k2, 1, D, N, alpha, d, itts)
        print('itts', itts, 'gpu-vs-oracle Eft %.2e ttau %.2e | oracle-vs-oracle(y*(1+1e-13)) Eft %.2e ttau %.2e'
              % (rel(r[0], o[0]), rel(r[5]['ttau'], o[5]['ttau']), rel(o2[0], o[0]), rel(o2[5]['ttau'], o[5]['ttau'])))
        print('   NaN counts gpu/oracle:', {nm: (int(np.isnan(r[5][nm]).sum()), int(np.isnan(o[5][nm]).sum())) for nm in ('ttau', 'tnu', 'R', 'MS')},
              'Eft', int(np.isnan(r[0]).sum()), int(np.isnan(o[0]).sum()))
        a = r[5]['ttau']; b = o[5]['ttau']; e = np.abs(a - b) / (np.abs(b) + 1e-12 * np.nanmax(np.abs(b)))
        e = np.where(np.isnan(a) != np.isnan(b), np.inf, e)
        bad = np.where(np.nanmax(e, axis=0) > 1e-6)[0]
        if bad.size:
            print('   bad steps', bad[:8])
            for k in (bad[0], bad[-1]):
                for nm in ('ttau', 'tnu', 'R'):
                    print('   k=%d gpu' % k, nm, r[5][nm][:, k]); print('   k=%d ora' % k, nm, o[5][nm][:, k])
            if itts == 2:
                r1 = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, k1, k2, 1, D, N, alpha, c['damp'][:1], 1, nargout=6)
                k = bad[0]
                print('   sweep-1 sites at k=%d: ttau' % k, r1[5]['ttau'][:, k], 'R', r1[5]['R'][:, k], 'Varft', r1[1][:, k], 'Eft', r1[0][:, k])


def one_widened(rng, raw=False):
    """a mixture draw (both variants) and an EKF-objective draw"""
    w = draw_widened(rng)
    c, mp, t, y, k1, k2, J, alpha, damp, itts, desc = (w[k] for k in ('c', 'mp', 't', 'y', 'k1', 'k2', 'J', 'alpha', 'damp', 'itts', 'desc'))
    T = t.size
    mom, omom = moms(c)
    res = {}
    def judge(tag, r, o, rerun):
        with np.errstate(all='ignore'):
            v = max(rel(r[0], o[0]), rel(r[1], o[1]), rel(r[5]['ttau'], o[5]['ttau']) * 0.1)
        if v > 1e-7 and not raw:
            # unstable instance?  a site update divided by 1 + d2*v ~ 1e-9 (|ttau| or |tnu| beyond 1e8, or non-finite), or an
            # oracle that itself moves under a 1e-13 relative change of y: nothing to compare (same rule as the main draw)
            o2 = rerun()
            with np.errstate(all='ignore'):
                sens = max(rel(o2[0], o[0]), rel(o2[1], o[1]), rel(o2[5]['ttau'], o[5]['ttau']) * 0.1)
                vals = np.concatenate([np.ravel(x[5][nm]) for x in (r, o) for nm in ('ttau', 'tnu')])
            singular = (not np.all(np.isfinite(vals))) or np.max(np.abs(vals)) > 1e8     # 1 + d2*v = O(1e-15): sign and size are rounding noise
            with np.errstate(all='ignore'):    # or a site of underflow size whose sign switches the look-up rule (see the main draw)
                singular = singular or any(np.any((np.abs(x[5]['ttau']) > 0) & (np.abs(x[5]['ttau']) < 1e-9 * np.nanmax(np.abs(x[5]['ttau'])))) for x in (r, o))
            if singular or v < 1e3 * sens or not np.isfinite(sens):
                return 0.0, ' [%s: unstable instance, oracle self-sensitivity %.1e, device difference %.1e]' % (tag, sens, v)
        return v, ''
    r = nagp.gf_ep_mods_nmf_mixture(mp['w'], t, y, SSHandle(), mom, t, k1, k2, J, alpha, damp, itts, nargout=6)
    o = omx.gf_ep_mods_nmf_mixture(mp['w'], t, y, None, omom, t, k1, k2, J, alpha, damp, itts)
    weak = {'mix_gf': weak_checks(r, o)}
    res['mix_gf'], note = judge('gf mixture', r, o, lambda: omx.gf_ep_mods_nmf_mixture(mp['w'], t, y * (1 + 1e-13), None, omom, t, k1, k2, J, alpha, damp, itts))
    desc += note
    r = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, J, alpha, damp, itts, nargout=6)
    o = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], None, omom, t, k1, k2, J, alpha, damp, itts)
    weak['mix_ihgp'] = weak_checks(r, o)
    res['mix_ihgp'], note = judge('ihgp mixture', r, o, lambda: omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'] * (1 + 1e-13), None, omom, t, k1, k2, J, alpha, damp, itts))
    desc += note
    D, N, T, eseed = w['ekf']
    pr = harness.nmf_problem(D, N, T, eseed, 'constraints')
    cons = harness.CONSTRAINTS_DEMO(D); w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
    e, _ = nagp.gf_giekf_modulator_nmf_constraints(w, t[:0], pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2, cons, wf,
                                                   harness.TUNE_DEMO, 'off') if False else nagp.gf_giekf_modulator_nmf_constraints(
        w, np.arange(1, T + 1.0), pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2, cons, wf, harness.TUNE_DEMO, 'off')
    eo, _ = oek.gf_giekf_modulator_nmf_constraints_nlml(w, np.arange(1, T + 1.0), pr['y'], 'matern32', 'matern52', 1, D, N, cons, wf, harness.TUNE_DEMO)
    res['ekf_e'] = abs(e - eo) / abs(eo)
    desc += ' | ekf D=%d N=%d T=%d' % (D, N, T)
    return desc, res, dict(weak=weak)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'widened':
        n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
        rng = np.random.default_rng(int(sys.argv[3]) if len(sys.argv) > 3 else 0); worst = {}; skipped = {}; t0 = time.time()
        for c in range(n):
            desc, res, _ = one_widened(rng)
            flag = ' <<<<' if tally(res, worst, skipped) else ''
            print('%3d %-100s %s%s' % (c, desc, fmt(res), flag)); sys.stdout.flush()
        print('worst', worst, 'not compared (all-NaN on both sides)', skipped, '%.0fs' % (time.time() - t0))
        sys.exit(0)
    k1o = None
    if '--k1' in sys.argv:      # python tools/gpu_fuzz.py 40 20271006 --k1 matern52 : every draw with six-state sub-band blocks
        i = sys.argv.index('--k1'); k1o = sys.argv[i + 1]; del sys.argv[i:i + 2]
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    rng = np.random.default_rng(seed); worst = {}; skipped = {}; t0 = time.time()
    only = int(sys.argv[3]) if len(sys.argv) > 3 else -1
    for c in range(n):
        desc, res, cfg = one(rng, k1_override=k1o)
        if only >= 0:
            if c == only:
                print(desc, res); diagnose_ihgp(cfg)
            continue
        flag = ' <<<<' if tally(res, worst, skipped) else ''
        print('%3d %-70s %s%s' % (c, desc, fmt(res), flag)); sys.stdout.flush()
    print('worst', worst, 'not compared (all-NaN on both sides)', skipped, '%.0fs' % (time.time() - t0))
