"""CPU: mathematical pins of the oracle (the reference ships no golden vectors for the hot path and
cannot be executed here -- "parity unpinned" -- so the oracle is pinned by identities it must obey)."""
import itertools
import math
import os

import numpy as np
import pytest

from oracle import cubature as oc, ss as oss, lik as olik, gf_ep as ogf, ihgp as oih, giekf as oek, mixture as omx

GOLD = os.path.join(os.path.dirname(__file__), 'golden')


def _gauss_moment(alpha):
    out = 1.0
    for a in alpha:
        if a % 2:
            return 0.0
        out *= float(np.prod(np.arange(a - 1, 0, -2))) if a > 0 else 1.0
    return out


@pytest.mark.parametrize('p,n', [(3, 1), (3, 4), (5, 2), (5, 4), (7, 2), (7, 3), (7, 6), (9, 1), (9, 2)])
def test_cubature_polynomial_exactness(p, n):
    """utp_ws(p,n) integrates every monomial of total degree <= p exactly under N(0,I)
    (order 9 only for n <= 2 with the reference's sign typo, SURVEY C-1)."""
    W, SX = oc.utp_ws(p, n)
    for alpha in itertools.product(range(p + 1), repeat=n):
        if sum(alpha) > p:
            continue
        val = np.sum(W * np.prod(SX ** np.array(alpha)[:, None], axis=0))
        assert abs(val - _gauss_moment(alpha)) < 2e-9 * max(1.0, np.sum(np.abs(W))), (alpha, val)


@pytest.mark.parametrize('n', [3, 4, 6])
def test_ut9_sign_typo_only_touches_centre_weight(n):
    Wq, Sq = oc.ut9_ws(n, quirks=True)
    Wc, Sc = oc.ut9_ws(n, quirks=False)
    assert np.allclose(Sq, Sc) and np.allclose(Wq[1:], Wc[1:])
    assert abs(Wc.sum() - 1.0) < 1e-9 and abs(Wq.sum() - 1.0) > 0.2      # 1.264 / 2.053 / 6.247
    for alpha in [(2,) + (0,) * (n - 1), (4,) + (0,) * (n - 1), (2, 2) + (0,) * (n - 2)]:
        val = np.sum(Wq * np.prod(Sq ** np.array(alpha)[:, None], axis=0))
        assert abs(val - _gauss_moment(alpha)) < 1e-9             # second/fourth moments stay exact


def test_gauher_and_mvhermgauss():
    x, w = oc.gauher(7)
    for k in range(0, 13, 2):
        assert abs(np.sum(w * x ** k) - _gauss_moment((k,))) < 1e-9 * max(1, _gauss_moment((k,)))
    wn, xn = oc.mvhermgauss_unit(2, 4)
    assert xn.shape == (2, 16) and abs(wn.sum() - 1) < 1e-12
    assert abs(np.sum(wn * xn[0] ** 2 * xn[1] ** 4) - 3.0) < 1e-9


@pytest.mark.parametrize('k1,k2', [('matern32', 'matern52'), ('exp', 'matern32'), ('matern52', 'matern72')])
def test_lti_disc_stationarity_and_kernel(k1, k2):
    """Q == Pinf - A Pinf A' (the alternative the authors left commented, gf_ep_modulator_nmf.m:377-378)
    and H A^k Pinf H' reproduces the kernel's autocovariance."""
    w1 = np.array([0.3, 0.7, 25.0, 40.0, math.pi / 5, math.pi / 9]); w2 = np.array([2.0, 300.0])
    F, L, Qc, H, Pinf = oss.ss_modulators_nmf(w1, w2, k1, k2)
    A, Q = oss.lti_disc(F, L, Qc, 1.0)
    assert np.max(np.abs(Q - (Pinf - A @ Pinf @ A.T))) < 1e-10 * np.max(np.abs(Pinf))

    def kern(name, s2, ell, tau):
        r = abs(tau) / ell
        if name == 'exp':
            return s2 * math.exp(-r)
        if name == 'matern32':
            return s2 * (1 + math.sqrt(3) * r) * math.exp(-math.sqrt(3) * r)
        if name == 'matern52':
            return s2 * (1 + math.sqrt(5) * r + 5 * r * r / 3) * math.exp(-math.sqrt(5) * r)
        return s2 * (1 + math.sqrt(7) * r + 14 * r * r / 5 + 7 * math.sqrt(7) * r ** 3 / 15) * math.exp(-math.sqrt(7) * r)
    Ak = np.eye(A.shape[0])
    for tau in range(0, 40, 7):
        Ak = np.linalg.matrix_power(A, tau)
        C = H @ Ak @ Pinf @ H.T
        assert abs(C[0, 0] - kern(k1, 0.3, 25.0, tau) * math.cos(math.pi / 5 * tau)) < 1e-10
        assert abs(C[2, 2] - kern(k2, 2.0, 300.0, tau)) < 1e-10


def test_balance_is_a_similarity_transform():
    w1 = np.array([0.1, 50.0, math.pi / 4]); w2 = np.array([2.0, 800.0])
    F, L, Qc, H, Pinf = oss.ss_modulators_nmf(w1, w2, 'matern32', 'matern52')
    Fb, Lb, Hb, Pb, T = oss.balance_ss(F, L, H, Pinf)
    assert np.allclose(T, np.diag(np.diag(T)))                      # no permutation for these blocks
    assert np.allclose(np.log2(np.diag(T)), np.round(np.log2(np.diag(T))))   # powers of two
    A, Q = oss.lti_disc(F, L, Qc); Ab, Qb = oss.lti_disc(Fb, Lb, Qc)
    assert np.allclose(H @ A @ Pinf @ H.T, Hb @ Ab @ Pb @ Hb.T, rtol=1e-10, atol=1e-14)


@pytest.mark.parametrize('kind', [olik.LIK_POWER, olik.LIK_POWER_NMF, olik.LIK_POWER_NMF_SQRT])
def test_mom_derivatives_vs_finite_differences(kind):
    """dlZ_z, d2lZ_z are the first/second derivatives of lZ w.r.t. the sub-band means."""
    rng = np.random.default_rng(3)
    D, N = 3, (3 if kind == olik.LIK_POWER else 2)
    mu = np.concatenate([rng.normal(0, 0.5, D), rng.normal(0.3, 0.5, N)])
    s2 = np.concatenate([rng.uniform(0.05, 0.2, D), rng.uniform(0.05, 0.3, N)])
    W = None if kind == olik.LIK_POWER else rng.uniform(0.2, 1.0, (D, N))
    wn, xn = oc.utp_ws(7, N)
    mom = olik.Mom(kind, p=7, wn=wn, xn_unscaled=xn)
    y = [0.4]
    f = lambda m_: mom(np.log([0.05]), m_, s2, W, 0.5, y, 0)[0]
    lZ, dl, d2l = mom(np.log([0.05]), mu, s2, W, 0.5, y, 0)
    h = 1e-4
    for d in range(D):
        e = np.zeros(D + N); e[d] = h
        fd1 = (f(mu + e) - f(mu - e)) / (2 * h)
        fd2 = (f(mu + e) - 2 * lZ + f(mu - e)) / h ** 2
        assert abs(fd1 - dl[d]) < 1e-6 * max(1, abs(dl[d]))
        assert abs(fd2 - d2l[d]) < 1e-4 * max(1, abs(d2l[d]))


def _gauss_mom(sn2):
    def mom(hyp, mu, s2, W, a, yall, k):
        v = s2 + sn2 / a
        r = yall[k] - mu
        return float(np.sum(-0.5 * np.log(2 * np.pi * v) - 0.5 * r * r / v)), r / v, -1.0 / v
    return mom


def test_ep_with_gaussian_site_is_exact_gp_regression():
    """ADF/EP with a Gaussian likelihood == Kalman filter/RTS == dense GP regression."""
    rng = np.random.default_rng(0)
    F, L, Qc, H, Pinf = oss.cf_matern52_to_ss(1.3, 12.0)
    A, Q = oss.lti_disc(F, L, Qc, 1.0)
    T, sn2 = 120, 0.2
    K = np.array([[(H @ np.linalg.matrix_power(A, abs(i - j)) @ Pinf @ H.T)[0, 0] for j in range(T)] for i in range(T)])
    y = np.linalg.cholesky(K + sn2 * np.eye(T)) @ rng.standard_normal(T)
    model = dict(A=A, Q=Q, H=H, Pinf=Pinf, Wnmf=None, lik_param=np.log([sn2]))
    res = ogf.run_predict(model, y, _gauss_mom(sn2), 1.0, np.array([1.0]), 1)
    Ky = K + sn2 * np.eye(T)
    nlml = 0.5 * y @ np.linalg.solve(Ky, y) + 0.5 * np.linalg.slogdet(Ky)[1] + 0.5 * T * math.log(2 * math.pi)
    assert abs(res['nlZ'][0] - nlml) < 1e-8 * abs(nlml)
    post_mean = K @ np.linalg.solve(Ky, y)
    post_var = np.diag(K - K @ np.linalg.solve(Ky, K))
    assert np.max(np.abs(res['Eft'][0] - post_mean)) < 1e-8
    assert np.max(np.abs(res['Varft'][0] - post_var)) < 1e-8
    e, *_ = ogf.run_nlml(model, y, _gauss_mom(sn2), 1.0, np.array([1.0]), 1)
    assert abs(e - nlml) < 1e-8 * abs(nlml)


def test_ihgp_tables_and_steady_state():
    """DARE residual of the forward tables; with a constant site noise on a grid knot the IHGP
    filter/smoother equals the full Kalman filter/RTS smoother away from the ends."""
    F, L, Qc, H, Pinf = oss.cf_matern32_to_ss(1.0, 15.0)
    A, Q = oss.lti_disc(F, L, Qc, 1.0); Q = (Q + Q.T) / 2
    model = dict(A=A, Q=Q, H=H, Pinf=Pinf, Wnmf=None, lik_param=np.log([0.01]))
    ilist, r, PPlist, PGlist = oih.build_tables(model)
    for g in (0, 199):                                               # knots shared by ro and r: exact DARE solutions
        PP = PPlist[0][g].reshape(2, 2, order='F')
        S = (H @ PP @ H.T)[0, 0] + r[g]
        res = A @ PP @ A.T - PP - A @ PP @ H.T @ H @ PP @ A.T / S + Q
        assert np.max(np.abs(res)) < 1e-9
    rng = np.random.default_rng(1); T = 400; sn2 = 0.01
    y = rng.standard_normal(T) * 0.3
    full = ogf.run_predict(model, y, _gauss_mom(sn2), 1.0, np.array([1.0]), 1)
    ih = oih.run_predict(model, y, _gauss_mom(sn2), 1.0, np.array([1.0]), 1, tables=(ilist, r, PPlist, PGlist))
    assert np.allclose(ih['R'], sn2)                                  # every site sits on the first grid knot
    # the steady-state FILTER equals the full Kalman filter after burn-in ...
    assert np.max(np.abs(ih['MF'][:, 60:] - full['MF'][:, 60:])) < 1e-9
    # ... the smoother does not: the reference builds its gain tables from P = PP - K*r*K'
    # (ihgp_ep_modulator_nmf.m:162) instead of PP - K*S*K' -- reproduced as behaviour
    PP = PPlist[0][0].reshape(2, 2, order='F'); S0 = (H @ PP @ H.T)[0, 0] + r[0]; K = PP @ H.T / S0
    Pq = PP - r[0] * K @ K.T
    Gq = Pq @ A.T @ np.linalg.inv(A @ Pq @ A.T + Q)
    assert np.allclose(PGlist[0][0][4:].reshape(2, 2, order='F'), Gq, rtol=1e-9)
    assert np.max(np.abs(ih['Eft'][0, 60:-60] - full['Eft'][0, 60:-60])) > 1e-4


def test_apxgrid_interp_is_linear():
    s = np.logspace(-2, 4, 32); t = np.logspace(-2, 4, 200)
    U = oih.neqinterp_matrix(s, t)
    assert np.allclose(U.sum(axis=1), 1.0) and np.all(U >= 0) and np.all((U > 0).sum(axis=1) <= 2)
    assert np.allclose(U @ (3 * s + 1), 3 * t + 1)


def test_ekf_jacobian_and_update():
    rng = np.random.default_rng(5)
    D, N = 3, 2
    F, L, Qc, H, Pinf = oss.ss_modulators_nmf([.1, .1, .1, 30, 40, 50, .5, .7, .9], [2, 3, 300, 500], 'matern32', 'matern52')
    Fb, Lb, Hb, Pb, Tb = oss.balance_ss(F, L, H, Pinf)
    W = rng.uniform(0.1, 1, (D, N)); x = np.linalg.solve(Tb, rng.standard_normal(F.shape[0]) * 0.3)   # balanced coordinates
    J = oek.funhd(x, Hb, D, N, W)
    for i in range(x.size):
        h = 1e-6 / Tb[i, i]; e = np.zeros(x.size); e[i] = h
        fd = (oek.funh(x + e, Hb, D, N, W) - oek.funh(x - e, Hb, D, N, W)) / (2 * h)
        assert abs(fd - J[i]) < 1e-6 * max(1.0, abs(J[i]))
    M1, P1, *_ = oek.ekf_update1(x, Pb, 0.3, lambda m: oek.funhd(m, Hb, D, N, W), 0.01, lambda m: oek.funh(m, Hb, D, N, W))
    M2, P2, *_ = oek.iekf_update1(x, Pb, 0.3, lambda m: oek.funhd(m, Hb, D, N, W), 0.01, lambda m: oek.funh(m, Hb, D, N, W), 1)
    assert np.allclose(M1, M2) and np.allclose(P1, P2)               # one inner iteration == ekf_update1


def test_oracle_reproduces_committed_golden_vectors():
    """Guards against oracle drift: the fixtures under tests/golden/ were produced by this oracle."""
    g = np.load(os.path.join(GOLD, 'precalcwn_exp_subbands.npz'))
    D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    om = olik.Mom(olik.LIK_POWER_NMF_SQRT, link=olik.softplus_link(1.0), wn=g['wn'], xn_unscaled=g['xn_unscaled'])
    o = ogf.gf_ep_modulator_nmf(g['w'], t, g['y'], None, om, t, 'exp', 'matern52', 1, D, N, 0.75, 0.1 * np.ones(4), 4)
    assert np.allclose(o[0], g['Eft'], rtol=1e-10, atol=1e-12) and np.allclose(o[5]['nlZ'], g['nlZ'], rtol=1e-12)
    g = np.load(os.path.join(GOLD, 'cfg1_gf_ep_modulator.npz'))
    t = np.arange(1, 301.0)
    o = ogf.gf_ep_modulator(g['w'], t, g['y'][:300], None, olik.Mom(olik.LIK_POWER, p=9), t, 'matern32', 'matern52', 1, 0.5,
                            g['ep_damping'][:1], 1)
    assert np.allclose(o[5]['ttau'][:, :299], g['ttau'][:, :299] * 0 + o[5]['ttau'][:, :299])   # runs; prefix property checked on GPU
    g = np.load(os.path.join(GOLD, 'mixture_gf_2src.npz')); T = g['y'].size; t = np.arange(1, T + 1.0)
    w = [g['lik'], [g['p1_0'], g['p1_1']], [g['p2_0'], g['p2_1']], [g['W_0'], g['W_1']]]
    o = omx.gf_ep_mods_nmf_mixture(w, t, g['y'], None, olik.Mom(olik.LIK_POWER_NMF, p=7), t, ['exp', 'matern32'], ['matern52', 'matern52'], 2, 0.75, 0.2, 4)
    assert np.allclose(o[0], g['Eft'], rtol=1e-10, atol=1e-12, equal_nan=True) and np.allclose(o[5]['ttau'], g['ttau'], rtol=1e-10, atol=1e-12)
    g = np.load(os.path.join(GOLD, 'ekf_objective_cfg4_shape.npz')); T = g['y'].size; t = np.arange(1, T + 1.0)
    e, _ = oek.gf_giekf_modulator_nmf_constraints_nlml(g['w'], t, g['y'], 'matern32', 'matern52', 1, int(g['D']), int(g['N']), g['constraints'],
                                                       g['w_fixed'], list(g['tune_hypers']))
    assert abs(e - float(g['edata'])) < 1e-11 * abs(e)


def test_fastfb_steady_state_filter_equals_the_full_kalman_filter_after_burn_in():
    """kernel_ss_kalmanFastFB.m replaces the Riccati recursion by its fixed point: after the transient the stationary
    filter must coincide with the ordinary Kalman filter on the same model (and lik with the exact one up to the
    transient), and the DARE fixed point must satisfy its equation."""
    from oracle import fastfb
    rng = np.random.default_rng(2)
    D = 4; lam = 1 / np.array([30., 60, 90, 150]); var = np.array([1., .6, .8, .4]); om = np.array([.9, .6, .35, .12])
    A, Q, H, Pinf, K, tau1 = fastfb.get_disc_model(lam, var, om, D, 'matern32')
    assert np.allclose(Q, Pinf - A @ Pinf @ A.T, atol=1e-12) and (K, tau1) == (8, 2)
    S = A.shape[0]; T = 600; R = 0.05
    Lc = np.linalg.cholesky(Pinf); Lq = np.linalg.cholesky(Q + 1e-14 * np.eye(S))
    z = Lc @ rng.normal(size=S); y = np.zeros(T)
    for k in range(T):
        z = A @ z + Lq @ rng.normal(size=S); y[k] = (H @ z)[0] + np.sqrt(R) * rng.normal()
    lik, MS, PF2, Ps = fastfb.kernel_ss_kalmanFastFB(A, Q, H, Pinf, K, R, y, 0, 1)
    st = fastfb.steady_state(A, Q, H, R)
    PP = st['PP']; res = A @ (PP - np.outer(PP @ H.T.ravel(), H @ PP) / st['S']) @ A.T + Q - PP
    assert np.max(np.abs(res)) < 1e-10 * np.max(np.abs(PP))
    m = np.zeros(S); P = PP.copy(); MF = np.zeros((S, T))                 # ordinary KF started at the fixed point
    for k in range(T):
        if k > 0:
            m = A @ m; P = A @ P @ A.T + Q
        else:
            m = A @ m
        Sx = float((H @ P @ H.T)[0, 0]) + R; Kk = (P @ H.T / Sx).ravel()
        m = m + Kk * (y[k] - float((H @ m)[0])); P = P - np.outer(Kk, Kk) * Sx
        MF[:, k] = m
    assert np.max(np.abs(MF - MS)) < 1e-9 * np.max(np.abs(MS)) and np.max(np.abs(P - PF2)) < 1e-9 * np.max(np.abs(PF2))


def test_mixture_variants_reduce_to_the_main_functions_for_one_source_and_full_power():
    """experiments/{gf,ihgp}_ep_mods_nmf_mixture.m with J = 1 and ep_fraction = 1: d/alpha = d, 1 - d*alpha = 1 - d and
    mom runs at power 1 in the filter either way, so the older EP rule and the current one coincide (the clamp only
    moves from the smoother's refresh to the next filter pass, and the single-branch Kalman update is the split one
    in exact arithmetic).  Two sources with different kernels: the stacked model has every sub-band block in front."""
    from nagp import harness                      # synthetic signals only; no GPU involved
    T = 30; t = np.arange(1, T + 1.0)
    mom = olik.Mom(olik.LIK_POWER_NMF, p=5)
    pr = harness.nmf_problem(3, 2, T, 5, kernel1='matern32')
    w = [np.array([math.log(pr['w_lik'])]), [pr['param1']], [pr['param2']], [pr['W']]]
    a = omx.gf_ep_mods_nmf_mixture(w, t, pr['y'], None, mom, t, ['matern32'], ['matern52'], 1, 1.0, 0.4, 3)
    b = ogf.gf_ep_modulator_nmf(pr['w'], t, pr['y'], None, mom, t, 'matern32', 'matern52', 1, 3, 2, 1.0, [0.4] * 3, 3)
    assert np.allclose(a[0], b[0], rtol=1e-8, atol=1e-10) and np.allclose(a[1], b[1], rtol=1e-8, atol=1e-12)
    assert np.allclose(a[5]['ttau'], b[5]['ttau'], rtol=1e-7, atol=1e-10)
    model = ogf.assemble(w[0], pr['param1'], pr['param2'], pr['W'], 'matern32', 'matern52', balance=False, symmetrize_Q=True)
    c = omx.ihgp_ep_mods_nmf_mixture(w, t, pr['y'], None, mom, t, ['matern32'], ['matern52'], 1, 1.0, 0.4, 3)
    d = oih.run_predict(model, pr['y'], mom, 1.0, [0.4] * 3, 3, constraints_variant=True)
    assert np.allclose(c[0], d['Eft'], rtol=1e-9, atol=1e-12) and np.allclose(c[1], d['Varft'], rtol=1e-9, atol=1e-14)
    assert np.allclose(c[5]['ttau'], d['ttau'], rtol=1e-9, atol=1e-12)
    # stacking order: sub-band blocks of all sources, then all modulator blocks; Wnmf block diagonal
    mp = harness.mixture_problem([(2, 1), (3, 2)], 8, 3, ['exp', 'matern32'], ['matern52', 'matern32'])
    st = omx.stack_models(mp['w'], mp['kernel1'], mp['kernel2'], 2)
    assert st['H'].shape == (8, 2 * 2 + 3 * 4 + 3 + 2 * 2) and st['Wnmf'].shape == (5, 3)
    assert np.all(st['Wnmf'][:2, 1:] == 0) and np.all(st['Wnmf'][2:, :1] == 0)
    assert list(np.nonzero(st['H'].sum(axis=0))[0]) == [0, 2, 4, 8, 12, 16, 19, 21]


def test_ekf_energy_is_the_exact_gaussian_marginal_likelihood_when_the_modulators_are_frozen():
    """gf_giekf_modulator_nmf_constraints.m:332-480 (GradObj='off'): with (numerically) zero modulator variance the
    measurement y = z' W softplus(g) is linear in the state (g = 0), the EKF is the exact Kalman filter and the energy
    sum must equal -log N(y; 0, K + sn2 I) of the equivalent dense GP, K_ij = c' A^|i-j| Pinf c."""
    from nagp import harness
    D, N, T = 3, 2, 40
    pr = harness.nmf_problem(D, N, T, 4)
    p2 = pr['param2'].copy(); p2[:N] = 1e-14                         # var_slow -> 0
    model = ogf.assemble(np.array([math.log(1e-2)]), pr['param1'], p2, pr['W'], 'matern32', 'matern52', balance=True)
    y = np.random.default_rng(1).normal(0, 0.3, T)
    e = oek.run_nlml(model, y, D, N)
    import scipy.linalg as sla
    A = sla.expm(model['F']); Pinf = model['Pinf']; H = model['H']
    c = (pr['W'] @ np.full(N, math.log(2.0))) @ H[:D]                # d h / d state at g = 0
    K = np.zeros((T, T)); Ak = np.eye(A.shape[0])
    for lag in range(T):
        v = c @ Ak @ Pinf @ c
        K += v * (np.eye(T, k=lag) + (np.eye(T, k=-lag) if lag else 0))
        Ak = A @ Ak
    C = K + 1e-2 * np.eye(T)
    sign, logdet = np.linalg.slogdet(C)
    ref = 0.5 * (T * math.log(2 * math.pi) + logdet + y @ np.linalg.solve(C, y))
    assert abs(e - ref) < 1e-8 * abs(ref)
