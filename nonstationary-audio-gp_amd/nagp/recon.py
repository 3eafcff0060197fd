"""Posterior reconstruction of the signal and the modulator amplitudes from the marginals the hot path returns
(matlab/demo_toy_modulators_nmf.m:119-158): nagp_reconstruct of the C ABI."""
import numpy as np

from . import _lib as L
from .cubature import gauher


def reconstruct_signal(Eft, Varft, Wnmf, link='softplus', link_shift=0.0, n_samples=0, seed=0, n_gh=32, device=0):
    """Eft, Varft: (D+N) x T as returned by gf_ep_modulator_nmf & co; Wnmf: D x N.
    n_samples = 0: population means / variances (Gauss-Hermite, closed-form combination);
    n_samples = s >= 2: the reference's sampling estimator (s = 250 in the demos) with a reproducible counter-based generator.
    Returns dict(Esig (T,), Vsig (T,), Eft_mod (N,T), Varft_mod (N,T))."""
    W = L.f64(Wnmf); D, N = W.shape
    E = L.f64(Eft); V = L.f64(Varft)
    if E.shape != V.shape or E.shape[0] != D + N:
        raise ValueError('Eft / Varft must be (D+N) x T')
    T = E.shape[1]
    gx, gw = gauher(int(n_gh))
    gx = L.f64(gx, 'C'); gw = L.f64(gw, 'C')
    Esig = np.zeros(T); Vsig = np.zeros(T); Em = np.zeros((N, T), order='F'); Vm = np.zeros((N, T), order='F')
    L.check(L.lib().nagp_reconstruct(D, N, T, L.dptr(E), L.dptr(V), L.dptr(W), L.LINK_SOFTPLUS if link == 'softplus' else L.LINK_EXP,
                                     float(link_shift), gx.size, L.dptr(gx), L.dptr(gw), int(n_samples), int(seed),
                                     L.dptr(Esig), L.dptr(Vsig), L.dptr(Em), L.dptr(Vm), int(device)))
    return dict(Esig=Esig, Vsig=Vsig, Eft_mod=Em, Varft_mod=Vm)
