"""Which randomised draws of tools/gpu_fuzz.py are ill-conditioned IN THE REFERENCE ALGORITHM ITSELF -- decided from the oracle
alone (no GPU, no device result involved): the oracle is run on y and on y*(1+1e-13); a draw/family is listed when the two
oracle runs differ by more than 1e-8 of the output scale (the reference's own result moves 1e5 times more than its input), or
a site parameter passes 1e8 / is not finite (a site update divided by 1 + d2*v = O(1e-9 .. 1e-15): sign and size are rounding
noise in the reference too).
    python tools/fuzz_conditioning.py main 30 2024 > tests/golden/fuzz_excused_main.json
    python tools/fuzz_conditioning.py widened 12 7 > tests/golden/fuzz_excused_widened.json
The GPU tests compare every draw that is NOT in these committed lists against the oracle at the full tolerance."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd')); sys.path.insert(0, os.path.join(ROOT, 'tools'))
import numpy as np


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return np.inf
    with np.errstate(all='ignore'):
        return float(np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-300)) if a.size else 0.0


SENS_MAX = 1e-8     # the oracle may move by at most 1e5 times a 1e-13 relative perturbation of y
SITE_MAX = 1e8      # a site parameter beyond this came out of a division by 1 + d2*v = O(1e-9 .. 1e-15)


def verdict(o, o2, ihgp):
    """(ill-conditioned?, self-sensitivity, largest site parameter, reason) from two oracle runs"""
    with np.errstate(all='ignore'):
        sens = max(rel(o2[0], o[0]), rel(o2[1], o[1]), 0.1 * rel(o2[5]['ttau'], o[5]['ttau']))
        vals = np.concatenate([np.ravel(o[5][nm]) for nm in ('ttau', 'tnu')])
        vals = vals[~np.isnan(vals)]                      # NaN tnu marks a missing observation, not an instability
        big = float(np.max(np.abs(vals))) if vals.size else 0.0
    if not np.isfinite(sens) or sens > SENS_MAX:
        return True, sens, big, 'the oracle itself moves by %.1e under a 1e-13 relative change of y' % sens
    if not np.isfinite(big) or big > SITE_MAX:
        return True, sens, big, 'a site parameter of size %.1e: its update was divided by 1 + d2*v ~ 0, sign and size are rounding noise' % big
    return False, sens, big, ''


def main_draws(n, seed):
    import gpu_fuzz_draws as gd
    from oracle import gf_ep as ogf, ihgp as oih
    rng = np.random.default_rng(seed); out = []
    for i in range(n):
        c = gd.draw(rng)
        D, N, T, k1, k2, itts, alpha, damp, pr, y = (c[k] for k in ('D', 'N', 'T', 'k1', 'k2', 'itts', 'alpha', 'damp', 'pr', 'y'))
        t = np.arange(1, T + 1.0); _, omom = gd.moms(c, host_only=True)
        for fam, f, yy in (('gf', ogf.gf_ep_modulator_nmf, y), ('ihgp', oih.ihgp_ep_modulator_nmf, pr['y'])):
            with np.errstate(all='ignore'):
                o = f(pr['w'], t, yy, None, omom, t, k1, k2, 1, D, N, alpha, damp, itts)
                o2 = f(pr['w'], t, yy * (1 + 1e-13), None, omom, t, k1, k2, 1, D, N, alpha, damp, itts)
            bad, sens, big, why = verdict(o, o2, fam == 'ihgp')
            if bad:
                out.append(dict(draw=i, family=fam, self_sensitivity=(None if not np.isfinite(sens) else sens), largest_site=(None if not np.isfinite(big) else big), reason=why,
                                config='D=%d N=%d T=%d p=%d %s %s(%g) %s/%s itts=%d alpha=%.2f' % (D, N, T, c['p'], c['kind'], c['link'], c['shift'], k1, k2, itts, alpha)))
    return out


def widened_draws(n, seed):
    import gpu_fuzz_draws as gd
    from oracle import mixture as omx
    rng = np.random.default_rng(seed); out = []
    for i in range(n):
        w = gd.draw_widened(rng)
        _, omom = gd.moms(w['c'], host_only=True)
        mp, t, y, k1, k2, J, alpha, damp, itts = (w[k] for k in ('mp', 't', 'y', 'k1', 'k2', 'J', 'alpha', 'damp', 'itts'))
        for fam, f, yy in (('mix_gf', omx.gf_ep_mods_nmf_mixture, y), ('mix_ihgp', omx.ihgp_ep_mods_nmf_mixture, mp['y'])):
            with np.errstate(all='ignore'):
                o = f(mp['w'], t, yy, None, omom, t, k1, k2, J, alpha, damp, itts)
                o2 = f(mp['w'], t, yy * (1 + 1e-13), None, omom, t, k1, k2, J, alpha, damp, itts)
            bad, sens, big, why = verdict(o, o2, fam == 'mix_ihgp')
            if bad:
                out.append(dict(draw=i, family=fam, self_sensitivity=(None if not np.isfinite(sens) else sens), largest_site=(None if not np.isfinite(big) else big), reason=why, config=w['desc']))
    return out


if __name__ == '__main__':
    which, n, seed = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    res = main_draws(n, seed) if which == 'main' else widened_draws(n, seed)
    json.dump(dict(generator='tools/fuzz_conditioning.py %s %d %d' % (which, n, seed), n_draws=n, seed=seed, excused=res), sys.stdout, indent=1)
    print()
