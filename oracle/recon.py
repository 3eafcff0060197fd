"""CPU restatement (test infrastructure) of the post-processing block of the demos -- matlab/demo_toy_modulators_nmf.m:119-158:
samples of the independent posterior marginals, Eft_mod / Varft_mod = mean / var of link(modulator samples), Esig / Vsig = mean /
var of sum_d (W link(g))_d z_d.  `sampling` follows the .m statement by statement with the draws of the library's counter-based
generator (MATLAB's randn stream cannot be reproduced); `moments` are the population values those sample
statistics estimate, by the same 1-D Gauss-Hermite rule."""
import numpy as np


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32-10 (Salmon et al. 2011) on arrays of uint32 counters."""
    c0, c1, c2, c3 = (np.asarray(c, np.uint64) & 0xFFFFFFFF for c in (c0, c1, c2, c3))
    k0 = np.uint64(k0 & 0xFFFFFFFF); k1 = np.uint64(k1 & 0xFFFFFFFF)
    for _ in range(10):
        p0 = np.uint64(0xD2511F53) * c0; p1 = np.uint64(0xCD9E8D57) * c2
        n0 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & np.uint64(0xFFFFFFFF); n1 = p1 & np.uint64(0xFFFFFFFF)
        n2 = ((p0 >> np.uint64(32)) ^ c3 ^ k1) & np.uint64(0xFFFFFFFF); n3 = p0 & np.uint64(0xFFFFFFFF)
        c0, c1, c2, c3 = n0, n1, n2, n3
        k0 = (k0 + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFF); k1 = (k1 + np.uint64(0xBB67AE85)) & np.uint64(0xFFFFFFFF)
    return c0, c1, c2, c3


def normals(T, site, n_samp, seed):
    """(T, n_samp) standard normals of one site: counter (t, 0, sample block, site), Box-Muller on 32-bit uniforms."""
    nb = (n_samp + 3) // 4
    t = np.arange(T, dtype=np.uint64)[:, None] + np.zeros((1, nb), np.uint64)
    q = np.zeros((T, 1), np.uint64) + np.arange(nb, dtype=np.uint64)[None, :]
    u = philox4x32_10(t & np.uint64(0xFFFFFFFF), t >> np.uint64(32), q, np.full((T, nb), site, np.uint64), seed & 0xFFFFFFFF, seed >> 32)
    u = [(x.astype(np.float64) + 0.5) * 2.0 ** -32 for x in u]
    z = np.empty((T, nb, 4))
    for h in range(2):
        r = np.sqrt(-2.0 * np.log(u[2 * h])); a = 2.0 * np.pi * u[2 * h + 1]
        z[:, :, 2 * h] = r * np.cos(a); z[:, :, 2 * h + 1] = r * np.sin(a)
    return z.reshape(T, nb * 4)[:, :n_samp]


def sampling(Eft, Varft, W, link, s, seed):
    """demo_toy_modulators_nmf.m:119-158 with s draws per marginal."""
    D, N = W.shape; T = Eft.shape[1]
    sub = np.stack([normals(T, d, s, seed) * np.sqrt(Varft[d])[:, None] + Eft[d][:, None] for d in range(D)])          # :132
    mod = np.stack([normals(T, D + n, s, seed) * np.sqrt(Varft[D + n])[:, None] + Eft[D + n][:, None] for n in range(N)])  # :142
    lm = link(mod)
    Eft_mod = lm.mean(axis=2); Varft_mod = lm.var(axis=2, ddof=1)                                                        # :143-144
    sig = np.einsum('dn,nts,dts->ts', W, lm, sub)                                                                        # :155-157
    return dict(Esig=sig.mean(axis=1), Vsig=sig.var(axis=1, ddof=1), Eft_mod=Eft_mod, Varft_mod=Varft_mod)               # :158-159


def moments(Eft, Varft, W, link, gh_x, gh_w, exp_link=False):
    D, N = W.shape
    mg, vg = Eft[D:], Varft[D:]
    if exp_link:
        e1 = np.exp(mg + 0.5 * vg); e2 = np.exp(2 * mg + 2 * vg)
    else:
        l = link(mg[:, :, None] + np.sqrt(vg)[:, :, None] * gh_x[None, None, :])
        e1 = l @ gh_w; e2 = (l * l) @ gh_w
    var = e2 - e1 * e1
    m, v = Eft[:D], Varft[:D]
    a = W @ e1
    Esig = np.sum(a * m, axis=0)
    Vsig = np.sum(a * a * v, axis=0) + np.sum(var * ((W.T @ m) ** 2 + (W.T ** 2) @ v), axis=0)
    return dict(Esig=Esig, Vsig=Vsig, Eft_mod=e1, Varft_mod=var)
