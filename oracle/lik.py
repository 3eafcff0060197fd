"""ORACLE (test infrastructure only -- never imported by the product path).

CPU restatement of the reference's `mom` callbacks (tilted-distribution moments by cubature).
PARITY UNPINNED (no reference fixtures); self-pinned by tests/test_oracle_selfpins.py
(test_mom_derivatives_vs_finite_differences, test_ep_with_gaussian_site_is_exact_gp_regression).

Follows (file:line under /root/reference/matlab):
  likModulatorPower.m:25-100            (one modulator per sub-band, jitter 1e-8)
  likModulatorNMFPower.m:28-87          (NMF mixing, jitter 1e-10, pEP_const = 1)
  experiments/likModulatorPreCalcwn.m:28-86  (sqrt-amplitude variant, true power-EP constant,
                                         precomputed sigma points)
"""
import math
import numpy as np
from . import cubature

LIK_POWER = 0
LIK_POWER_NMF = 1
LIK_POWER_NMF_SQRT = 2


def softplus_link(shift=0.0):
    # @(g) log(1+exp(g-shift))  -- literal formula as in demo_toy_modulators.m:16
    return lambda g: np.log(1.0 + np.exp(g - shift))


def exp_link():
    return lambda g: np.exp(g)


def normpdf(x, mu, sigma):
    return np.exp(-0.5 * ((x - mu) / sigma) ** 2) / (math.sqrt(2.0 * math.pi) * sigma)


def _moments(link, sn2, y, mu_z, mu_g, s2_z, s2_g, Wmix, wn, xn_unscaled, ep_fraction, jitter,
             sqrt_amp, pEP_const):
    """Common body of the three callbacks; Wmix=None means identity mixing (non-NMF)."""
    with np.errstate(all='ignore'):
        xn = (mu_g[:, None] + np.sqrt(s2_g)[:, None] * xn_unscaled).T          # (npts, N)
        lk = link(xn)
        link_xn_W = lk if Wmix is None else lk @ Wmix.T                        # (npts, D)
        if sqrt_amp:
            link_xn_W = np.sqrt(link_xn_W)
        sn2_link = sn2 / ep_fraction + (link_xn_W ** 2) @ s2_z                 # (npts,)
        link_mu = link_xn_W @ mu_z
        xg = (xn - mu_g[None, :]) / s2_g[None, :]
        normy = normpdf(y, link_mu, np.sqrt(sn2_link))
        ssum = np.sum(wn * normy)
        # MATLAB max(NaN, jitter) = jitter
        Z = pEP_const * (jitter if (np.isnan(ssum) or ssum < jitter) else ssum)
        Zinv = 1.0 / Z
        lZ = math.log(Z) if Z > 0 else float(np.log(Z + 0j).real)
        c1 = (y - link_mu) / sn2_link
        dZ1 = np.sum((wn * normy * c1)[:, None] * link_xn_W, axis=0)
        dlZ_z = Zinv * pEP_const * dZ1
        dZ2 = np.sum((wn * normy)[:, None] * xg, axis=0)
        dlZ_g = Zinv * pEP_const * dZ2
        d2Z1 = np.sum((wn * normy * (c1 ** 2 - 1.0 / sn2_link))[:, None] * link_xn_W ** 2, axis=0)
        d2lZ_z = -dlZ_z ** 2 + Zinv * pEP_const * d2Z1
        d2Z2 = np.sum((wn * normy)[:, None] * (xg ** 2 - 1.0 / s2_g[None, :]), axis=0)
        d2lZ_g = -dlZ_g ** 2 + Zinv * pEP_const * d2Z2
    return lZ, np.concatenate([dlZ_z, dlZ_g]), np.concatenate([d2lZ_z, d2lZ_g])


def likModulatorPower(link, hyp, y, mu, s2, p, ep_fraction, quirks=True, _cache={}):
    """likModulatorPower.m:25-100."""
    sn2 = math.exp(float(np.ravel(hyp)[0]))
    mu = np.asarray(mu, float).ravel(); s2 = np.asarray(s2, float).ravel()
    D = len(mu) // 2
    key = (p, D, quirks)
    if key not in _cache:
        _cache[key] = cubature.sigma_points(p, D, quirks)
    wn, xn_unscaled = _cache[key]
    return _moments(link, sn2, y, mu[:D], mu[D:], s2[:D], s2[D:], None, wn, xn_unscaled,
                    ep_fraction, 1e-8, False, 1.0)


def likModulatorNMFPower(link, hyp, y, mu, s2, W, p, ep_fraction, quirks=True, _cache={}):
    """likModulatorNMFPower.m:28-87."""
    sn2 = math.exp(float(np.ravel(hyp)[0]))
    mu = np.asarray(mu, float).ravel(); s2 = np.asarray(s2, float).ravel()
    D, N = W.shape
    key = (p, N, quirks)
    if key not in _cache:
        _cache[key] = cubature.sigma_points(p, N, quirks)
    wn, xn_unscaled = _cache[key]
    return _moments(link, sn2, y, mu[:D], mu[D:], s2[:D], s2[D:], W, wn, xn_unscaled,
                    ep_fraction, 1e-10, False, 1.0)


def likModulatorPreCalcwn(link, hyp, y, mu, s2, W, ep_fraction, wn, xn_unscaled):
    """experiments/likModulatorPreCalcwn.m:28-86."""
    sn2 = math.exp(float(np.ravel(hyp)[0]))
    mu = np.asarray(mu, float).ravel(); s2 = np.asarray(s2, float).ravel()
    D, N = W.shape
    pEP_const = (2 * math.pi * sn2) ** (0.5 * (1 - ep_fraction)) * ep_fraction ** (-0.5)
    return _moments(link, sn2, y, mu[:D], mu[D:], s2[:D], s2[D:], W, np.asarray(wn, float).ravel(),
                    np.asarray(xn_unscaled, float), ep_fraction, 1e-10, True, pEP_const)


class Mom:
    """Stand-in for the MATLAB closure `mom` (demo_toy_modulators_nmf.m:81, demo_toy_modulators.m:81,
    train_GTFNMF.m:149): mom(hyp, mu, s2, [Wnmf,] ep_frac, yall, k) -> lZ, dlZ(1xM), d2lZ(1xM)."""

    def __init__(self, kind, link=None, p=9, quirks=True, wn=None, xn_unscaled=None):
        self.kind = kind
        self.link = link if link is not None else softplus_link(0.0)
        self.p = p
        self.quirks = quirks
        self.wn = wn
        self.xn_unscaled = xn_unscaled

    def __call__(self, hyp, mu, s2, Wnmf, ep_frac, yall, k):
        y = yall[k]
        if self.kind == LIK_POWER:
            return likModulatorPower(self.link, hyp, y, mu, s2, self.p, ep_frac, self.quirks)
        if self.kind == LIK_POWER_NMF:
            return likModulatorNMFPower(self.link, hyp, y, mu, s2, Wnmf, self.p, ep_frac, self.quirks)
        if self.kind == LIK_POWER_NMF_SQRT:
            return likModulatorPreCalcwn(self.link, hyp, y, mu, s2, Wnmf, ep_frac, self.wn, self.xn_unscaled)
        raise ValueError(self.kind)
