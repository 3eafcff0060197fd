"""GPU (-m gpu): parity of the HIP path against the CPU oracle and the committed golden vectors,
called through the product's public interface (reference-named functions -> C ABI -> HIP kernels).

Stated tolerances (north_star: |dlogZ|/|logZ| < 1e-5): filtered/smoothed means and variances within
1e-7 of the largest magnitude of the array, site parameters 1e-6, |dlogZ|/|logZ| < 1e-8.
"""
import os

import numpy as np
import pytest

import nagp
from nagp import harness, Mom, SSHandle, Plan, _lib as L
from nagp import ss as pss
from oracle import gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik, fastfb as offb, mixture as omx, ss as oss

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), 'golden')
TOL_MEAN, TOL_SITE, TOL_LOGZ = 1e-7, 1e-6, 1e-8


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b))
    return float(np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-300)) if a.size else 0.0


def relz(a, b):
    return float(np.max(np.abs(np.asarray(a) - np.asarray(b)) / np.abs(np.asarray(b))))


@pytest.fixture(scope='module', autouse=True)
def _lib(nagp_lib):
    assert nagp_lib.nagp_device_count() >= 1
    return nagp_lib


def gold(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def test_full_length_reference_runs_are_started(full_length_refs):
    """Starts the five CPU legs of the full-length parity tests at the end of this file (threads inside the compiled oracle; the GIL is
    released): they take minutes of one core each and run beside the GPU tests in between."""
    full_length_refs.start()
    assert len(full_length_refs.threads) == 7


def test_golden_cfg1_gf_ep_modulator_full_size():
    g = gold('cfg1_gf_ep_modulator'); T = g['y'].size; t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorPower', p_cubature=9)
    Eft, Varft, _, lb, ub, out = nagp.gf_ep_modulator(g['w'], t, g['y'], SSHandle('ss_modulators'), mom, t, 'matern32', 'matern52', 1,
                                                      0.5, g['ep_damping'], 5, nargout=6)
    assert rel(Eft, g['Eft']) < TOL_MEAN and rel(Varft, g['Varft']) < TOL_MEAN
    assert relz(out['nlZ'], g['nlZ']) < TOL_LOGZ
    assert rel(out['ttau'], g['ttau']) < TOL_SITE and rel(out['tnu'], g['tnu']) < TOL_SITE
    assert rel(out['lZ'], g['lZ']) < 1e-7
    assert rel(out['maxDiffM'], g['maxDiffM']) < 1e-6 and rel(out['maxDiffP'], g['maxDiffP']) < 1e-6
    assert np.allclose(lb, Eft - 1.96 * np.sqrt(Varft)) and np.allclose(ub, Eft + 1.96 * np.sqrt(Varft))
    e, eg = nagp.gf_ep_modulator(g['w'], t, g['y'], SSHandle('ss_modulators'), mom, None, 'matern32', 'matern52', 1, 0.5, g['ep_damping'], 3)
    assert abs(e - float(g['edata_I3'])) < TOL_LOGZ * abs(e) and not np.any(eg)


def test_golden_cfg2_gf_ep_modulator_nmf_with_missing_data():
    g = gold('cfg2_gf_ep_modulator_nmf'); D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorNMFPower', p_cubature=9); d = 0.5 * np.ones(3)
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(g['w'], t, g['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3, nargout=6)
    assert rel(Eft, g['Eft']) < TOL_MEAN and rel(Varft, g['Varft']) < TOL_MEAN and relz(out['nlZ'], g['nlZ']) < TOL_LOGZ
    assert rel(out['ttau'], g['ttau']) < TOL_SITE and rel(out['tnu'], g['tnu']) < TOL_SITE
    assert out['counters']['nan_obs'] == 3 * 60 and out['counters']['chol_retries'] == 0
    for I, key in ((1, 'edata_I1'), (3, 'edata_I3')):
        e, _ = nagp.gf_ep_modulator_nmf(g['w'], t, g['y'], SSHandle(), mom, None, 'matern32', 'matern52', 1, D, N, 0.5, d, I)
        assert abs(e - float(g[key])) < TOL_LOGZ * abs(e)


def test_golden_cfg3_ihgp():
    g = gold('cfg3_ihgp_ep_modulator_nmf'); D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorNMFPower', p_cubature=7); d = 0.5 * np.ones(3)
    Eft, Varft, _, _, _, out = nagp.ihgp_ep_modulator_nmf(g['w'], t, g['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3, nargout=6)
    assert rel(Eft, g['Eft']) < TOL_MEAN and rel(Varft, g['Varft']) < TOL_MEAN and relz(out['nlZ'], g['nlZ']) < TOL_LOGZ
    assert rel(out['ttau'], g['ttau']) < TOL_SITE
    assert np.array_equal(np.isinf(out['R']), np.isinf(g['R']))
    fin = np.isfinite(g['R'])
    assert rel(out['R'][fin], g['R'][fin]) < TOL_SITE
    assert np.all(Varft == Varft[:, :1])                                   # time-constant (SURVEY C-10)


def test_golden_cfg4_giekf_both_variants():
    g = gold('cfg4_gf_giekf_modulator_nmf'); D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    Eft, Varft, _, _, _, out = nagp.gf_giekf_modulator_nmf_constraints(g['w'], t, g['y'], SSHandle(), None, t, 'matern32', 'matern52', 1, D, N,
                                                                       3, 1, g['constraints'], g['w_fixed'], list(g['tune_hypers']), nargout=6)
    assert rel(Eft, g['Eft']) < TOL_MEAN and rel(Varft, g['Varft']) < TOL_MEAN and rel(out['maxDiffP'], g['maxDiffP']) < 1e-6
    Eft, Varft, _, _, _, out = nagp.gf_giekf_modulator_nmf(g['w_log'], t, g['y'], SSHandle(), None, t, 'matern32', 'matern52', 1, D, N, 2, 2, nargout=6)
    assert rel(Eft, g['Eft_plain']) < TOL_MEAN and rel(Varft, g['Varft_plain']) < TOL_MEAN
    assert rel(out['maxDiffP'], g['maxDiffP_plain']) < 1e-6


def test_golden_cfg5_constraints_S146():
    g = gold('cfg5_gf_ep_modulator_nmf_constraints'); D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorNMFPower', p_cubature=7); d = 0.5 * np.ones(3)
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf_constraints(g['w'], t, g['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3,
                                                                    g['constraints'], g['w_fixed'], list(g['tune_hypers']), nargout=6)
    assert rel(Eft, g['Eft']) < TOL_MEAN and rel(Varft, g['Varft']) < TOL_MEAN and relz(out['nlZ'], g['nlZ']) < TOL_LOGZ
    assert rel(out['ttau'], g['ttau']) < TOL_SITE and rel(out['tnu'], g['tnu']) < TOL_SITE


def test_golden_precalcwn_sqrt_likelihood_exp_subbands_shifted_link():
    g = gold('precalcwn_exp_subbands'); D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorPreCalcwn', link='softplus', link_shift=1.0, wn=g['wn'], xn_unscaled=g['xn_unscaled'])
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(g['w'], t, g['y'], SSHandle(), mom, t, 'exp', 'matern52', 1, D, N, 0.75, 0.1 * np.ones(4), 4, nargout=6)
    assert rel(Eft, g['Eft']) < TOL_MEAN and rel(Varft, g['Varft']) < TOL_MEAN and relz(out['nlZ'], g['nlZ']) < TOL_LOGZ


@pytest.mark.parametrize('T,nanpos', [(1, []), (2, []), (3, [0]), (40, [39]), (40, list(range(40))), (65, [0, 1, 2, 63, 64])])
def test_edge_lengths_and_missing_patterns_against_oracle(T, nanpos):
    """T = 1/2, NaN at the first / last step, everything missing, chunk-boundary lengths."""
    D, N = 3, 2
    pr = harness.nmf_problem(D, N, T, 5); y = pr['y'].copy(); y[nanpos] = np.nan
    t = np.arange(1, T + 1.0); d = np.array([0.4, 0.6])
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), Mom('likModulatorNMFPower', p_cubature=5), t, 'matern32', 'matern52',
                                                        1, D, N, 0.5, d, 2, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, olik.Mom(olik.LIK_POWER_NMF, p=5), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(Eft, o[0]) < TOL_MEAN and rel(Varft, o[1]) < TOL_MEAN
    assert np.allclose(out['nlZ'], o[5]['nlZ'], rtol=TOL_LOGZ, atol=1e-12)
    assert rel(out['PS'], np.transpose(o[5]['PS'], (1, 2, 0))) < TOL_MEAN and rel(out['MS'], o[5]['MS']) < TOL_MEAN
    # the other two families on the same input
    r2 = nagp.ihgp_ep_modulator_nmf(pr['w'], t, y, SSHandle(), Mom('likModulatorNMFPower', p_cubature=5), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    o2 = oih.ihgp_ep_modulator_nmf(pr['w'], t, y, None, olik.Mom(olik.LIK_POWER_NMF, p=5), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(r2[0], o2[0]) < TOL_MEAN and rel(r2[1], o2[1]) < TOL_MEAN and np.allclose(r2[5]['nlZ'], o2[5]['nlZ'], rtol=TOL_LOGZ, atol=1e-12)
    r3 = nagp.gf_giekf_modulator_nmf(pr['w'], t, y, SSHandle(), None, t, 'matern32', 'matern52', 1, D, N, 2, 2, nargout=2)
    o3 = oek.gf_giekf_modulator_nmf(pr['w'], t, y, None, None, t, 'matern32', 'matern52', 1, D, N, 2, 2)
    assert rel(r3[0], o3[0]) < TOL_MEAN and rel(r3[1], o3[1]) < TOL_MEAN


@pytest.mark.parametrize('D,N,p,lik', [
    (3, 1, 9, 'likModulatorNMFPower'),      # one modulator: smallest MFMA operand block
    (5, 7, 7, 'likModulatorNMFPower'),      # N = 7: the 16 x 16 MFMA block is full (rows 0..15, cols 0..14)
    (4, 8, 7, 'likModulatorNMFPower'),      # N = 8: does not fit the MFMA block -> 16-lane-group sums
    (3, 6, 4, 'likModulatorNMFPower'),      # Gauss-Hermite 4^6 = 4096 points: several chunks of 1024 sigma points
    (21, 2, 7, 'likModulatorPreCalcwn'),    # sqrt amplitude, 21 sub-bands over 8 lanes x 4 registers
    (5, 8, 7, 'likModulatorPreCalcwn'),     # sqrt amplitude at the largest cubature dimension
])
def test_every_cubature_code_path_against_oracle(D, N, p, lik):
    """mom is instantiated per cubature dimension and has separate code paths per likelihood; each one is compared
    with the oracle through a short two-sweep EP run (ADF filter, smoother, site refresh all call it).  Rules with
    large negative weights (ut3 / ut5 at n >= 5) make the EP iteration itself unstable -- rounding-level differences
    grow to O(1) within a few steps in the oracle as well -- so the orders here keep the rule well behaved."""
    T = 24
    pr = harness.nmf_problem(D, N, T, 77); t = np.arange(1, T + 1.0); d = np.array([0.5, 0.5])
    if lik == 'likModulatorPreCalcwn':
        from nagp import cubature
        wn, xn = cubature.utp_ws(p, N)
        mom = Mom(lik, wn=wn, xn_unscaled=xn); omom = olik.Mom(olik.LIK_POWER_NMF_SQRT, wn=wn, xn_unscaled=xn)
    else:
        mom = Mom(lik, p_cubature=p); omom = olik.Mom(olik.LIK_POWER_NMF, p=p)
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, pr['y'], None, omom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(Eft, o[0]) < TOL_MEAN and rel(Varft, o[1]) < TOL_MEAN
    assert rel(out['ttau'], o[5]['ttau']) < TOL_SITE and rel(out['tnu'], o[5]['tnu']) < TOL_SITE
    assert np.allclose(out['nlZ'], o[5]['nlZ'], rtol=TOL_LOGZ, atol=1e-12)
    r2 = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    o2 = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], None, omom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(r2[0], o2[0]) < TOL_MEAN and rel(r2[1], o2[1]) < TOL_MEAN and np.allclose(r2[5]['nlZ'], o2[5]['nlZ'], rtol=TOL_LOGZ, atol=1e-12)


@pytest.mark.parametrize('D,N,p,kind', [(5, 6, 7, 'nmf'), (5, 7, 7, 'nmf'), (4, 8, 7, 'nmf'), (5, 3, 9, 'nmf'), (7, 2, 5, 'sqrt'),
                                        (5, 7, 7, 'sqrt'), (4, 0, 5, 'power'), (3, 0, 9, 'power'),
                                        (6, 9, 7, 'nmf'), (6, 9, 7, 'sqrt')])      # N = 9: three sources x three components
def test_mom_callback_itself_against_oracle(D, N, p, kind):
    """The `mom` handle called with the reference's own arity on arbitrary (mu, s2, y): likModulatorPower,
    likModulatorNMFPower, likModulatorPreCalcwn (C ABI: nagp_mom_eval), 48 inputs per launch including tiny variances,
    large means and a missing-data-like far-off observation.  Entries that are rounding noise relative to the largest
    entry of their vector (cancellation in -dlZ^2 + ...) are compared on the vector's scale."""
    rng = np.random.default_rng(100 * D + N); n = 48; hyp = np.log(1e-2)
    if kind == 'power':
        M = 2 * D; W = None; mom = Mom('likModulatorPower', p_cubature=p); omom = olik.Mom(olik.LIK_POWER, p=p)
    else:
        M = D + N; W = rng.uniform(0, 0.5, (D, N))
        if kind == 'sqrt':
            from nagp import cubature
            wn, xn = cubature.utp_ws(p, N)
            mom = Mom('likModulatorPreCalcwn', wn=wn, xn_unscaled=xn); omom = olik.Mom(olik.LIK_POWER_NMF_SQRT, wn=wn, xn_unscaled=xn)
        else:
            mom = Mom('likModulatorNMFPower', p_cubature=p); omom = olik.Mom(olik.LIK_POWER_NMF, p=p)
    mu = rng.normal(0, 1, (M, n)); s2 = rng.uniform(0.01, 2.0, (M, n)); y = rng.normal(0, 1, n)
    s2[:, 1] *= 1e-6; mu[:, 2] *= 10; s2[D:, 3] *= 50; y[4] = 40.0
    args = (W,) if W is not None else ()
    lZ, dl, d2l = mom(hyp, mu, s2, *args, 0.5, [y], 0)
    one = mom(hyp, mu[:, 7], s2[:, 7], *args, 0.5, y, 7)               # scalar form, k indexes yall
    assert one[0] == lZ[7] and np.array_equal(one[1], dl[:, 7]) and np.array_equal(one[2], d2l[:, 7])
    for i in range(n):
        a = omom(hyp, mu[:, i], s2[:, i], W, 0.5, y, i)
        assert abs(a[0] - lZ[i]) <= 1e-9 * max(1.0, abs(a[0]))
        assert rel(dl[:, i], np.ravel(a[1])) < 1e-8 and rel(d2l[:, i], np.ravel(a[2])) < 1e-8


@pytest.mark.parametrize('iters', [1, 3])
def test_ekf_update1_and_iekf_update1_standalone(iters):
    """[M,P,K,MU,S] = (i)ekf_update1(M,P,y,H,R,h,[],[],iters) with the drivers' measurement handles, balanced H
    (entries that are powers of two, not 1) -- iekf_update1.m:110-117, ekf_update1.m:106-109."""
    D, N = 4, 2
    pr = harness.nmf_problem(D, N, 8, 9)
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    _, _, H, Pinf = blk.dense()
    S = Pinf.shape[0]; rng = np.random.default_rng(4)
    Lc = np.linalg.cholesky(Pinf); m0 = Lc @ rng.normal(size=S)
    G = rng.normal(size=(S, S)) * 0.1
    P0 = Lc @ (np.eye(S) * 0.6 + G @ G.T) @ Lc.T                    # a dense SPD prior covariance
    model = nagp.MeasModel(H, pr['W'], D, N); y = 0.37; R = 1e-2
    if iters == 1:
        m, P, K, MU, Sx = nagp.ekf_update1(m0, P0, y, model.dh, R, model.h)
        o = oek.ekf_update1(m0.copy(), P0.copy(), y, lambda x: oek.funhd(x, H, D, N, pr['W']), R, lambda x: oek.funh(x, H, D, N, pr['W']))
    else:
        m, P, K, MU, Sx = nagp.iekf_update1(m0, P0, y, model.dh, R, model.h, None, None, iters)
        o = oek.iekf_update1(m0.copy(), P0.copy(), y, lambda x: oek.funhd(x, H, D, N, pr['W']), R, lambda x: oek.funh(x, H, D, N, pr['W']), iters)
    assert rel(m, o[0]) < 1e-12 and rel(P, o[1]) < 1e-12 and rel(K, o[2]) < 1e-12
    assert abs(MU - float(o[3])) < 1e-12 * max(1, abs(float(o[3]))) and abs(Sx - float(o[4])) < 1e-12 * abs(float(o[4]))


@pytest.mark.parametrize('kernel,D,T,KF', [('exp', 16, 3000, 0), ('matern32', 8, 1000, 0), ('matern52', 5, 700, 1), ('exp', 3, 1, 0),
                                           ('exp', 48, 400, 0), ('matern32', 32, 500, 0), ('matern32', 60, 300, 1)])
def test_stationary_filterbank_kalmanFastFB(kernel, D, T, KF):
    """[lik,Xfin,Pfin] = kernel_ss_kalmanFastFB(A,Q,C,P0,K,vary,y,verbose,KF) with the model of get_disc_model
    (SURVEY 8f row f-2): missing samples, filter-only option, T = 1, S = 96 (both constant matrices in the LDS), S = 128 (32 Matern-3/2 channels) and
    S = 240 (matrices read from global memory: a thread per state, S <= 256)."""
    rng = np.random.default_rng(D + T)
    lam = 1.0 / rng.uniform(20, 400, D); var = rng.uniform(0.1, 1.0, D); om = np.linspace(np.pi / 3, np.pi / 50, D)
    A, Q, H, Pinf, K, tau1 = nagp.get_disc_model(lam, var, om, D, kernel, 6)
    Ao, Qo, Ho, Po, Ko, to = offb.get_disc_model(lam, var, om, D, kernel, 6)
    assert rel(A, Ao) < 1e-13 and rel(Q, Qo) < 1e-12 and np.array_equal(H, Ho) and rel(Pinf, Po) < 1e-14 and (K, tau1) == (Ko, to)
    S = A.shape[0]; Lc = np.linalg.cholesky(Pinf); Lq = np.linalg.cholesky(Q + 1e-14 * np.eye(S))
    z = Lc @ rng.normal(size=S); y = np.zeros(T)
    for k in range(T):
        z = A @ z + Lq @ rng.normal(size=S); y[k] = (H @ z)[0] + 0.1 * rng.normal()
    if T > 100:
        y[40:75] = np.nan; y[T - 3] = np.nan
    lik, Xfin, Pfin = nagp.kernel_ss_kalmanFastFB(A, Q, H, Pinf, K, 0.01, y, 0, KF)
    lo, MSo, PF2o, Pso = offb.kernel_ss_kalmanFastFB(Ao, Qo, Ho, Po, Ko, 0.01, y, 0, KF)
    assert Xfin.shape == (1, S, T) and Pfin.shape == (S, S, T)
    assert rel(Xfin[0], MSo) < TOL_MEAN and abs(lik - lo) < TOL_LOGZ * abs(lo)
    assert rel(Pfin[:, :, T - 1], PF2o) < 1e-10 and (T == 1 or rel(Pfin[:, :, 0], PF2o if KF == 1 else Pso) < 1e-10)


def test_stationary_filterbank_refuses_what_does_not_fit():
    A, Q, H, Pinf, K, _ = nagp.get_disc_model(np.full(130, 0.01), np.ones(130), np.linspace(1, 0.1, 130), 130, 'exp')   # S = 260 > 256 (a thread per state)
    with pytest.raises(nagp.NagpError):
        nagp.kernel_ss_kalmanFastFB(A, Q, H, Pinf, K, 0.01, np.zeros(10))


def test_batched_nlml_driver_equals_serial_calls_and_finite_differences():
    """Row f-3: the numel(w)+1 objective evaluations of one fminunc iteration (GradObj off, train_GTFNMF.m:186-201) as one
    batched plan: bitwise equal to the serial calls of gf_ep_modulator_nmf_constraints, and the forward-difference
    gradient agrees with differences of the oracle's objective."""
    D, N, T = 3, 2, 120
    pr = harness.nmf_problem(D, N, T, 31, 'constraints'); t = np.arange(1, T + 1.0)
    cons = harness.CONSTRAINTS_DEMO(D); w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
    mom = Mom('likModulatorNMFPower', p_cubature=5); d = np.array([0.5, 0.5])
    args = (t, pr['y'], SSHandle(), mom, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    f0, g = nagp.fd_value_and_gradient(w, *args, constraints=cons, w_fixed=wf, tune_hypers=harness.TUNE_DEMO)
    serial = nagp.gf_ep_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), mom, None, 'matern32', 'matern52', 1, D, N, 0.5, d, 2,
                                                  cons, wf, harness.TUNE_DEMO)[0]
    assert f0 == serial and g.shape == w.shape and np.all(np.isfinite(g))
    omom = olik.Mom(olik.LIK_POWER_NMF, p=5)
    def obj(wv):
        return ogf.gf_ep_modulator_nmf_constraints(wv, t, pr['y'], None, omom, None, 'matern32', 'matern52', 1, D, N, 0.5, d, 2,
                                                   cons, wf, harness.TUNE_DEMO)[0]
    fo = obj(w)
    assert abs(f0 - fo) < TOL_LOGZ * abs(fo)
    i = int(np.argmax(np.abs(g))); h = 1e-6 * max(abs(w[i]), 1.0)
    e = np.zeros(w.size); e[i] = h
    go = (obj(w + e) - obj(w - e)) / (2 * h)
    assert abs(g[i] - go) < 1e-3 * abs(go)


def _fuzz_module():
    import importlib.util, sys
    tools = os.path.join(os.path.dirname(__file__), '..', 'tools')
    if tools not in sys.path:
        sys.path.insert(0, tools)
    spec = importlib.util.spec_from_file_location('gpu_fuzz', os.path.join(tools, 'gpu_fuzz.py'))
    fz = importlib.util.module_from_spec(spec); spec.loader.exec_module(fz)
    return fz


def _excused(name):
    import json
    with open(os.path.join(GOLD, name)) as fh:
        d = json.load(fh)
    # value: can the oracle's NaN pattern be held against the device?  Yes for the mildly ill-conditioned draws (the oracle moves by
    # < 1e-6 under the 1e-13 perturbation of y and no site passes 1e10); no when the oracle's own NaN pattern moves under the
    # perturbation (self_sensitivity null) or sites of 1e14+ make inf - inf a matter of rounding.  Decided from the committed,
    # oracle-derived list alone.
    def pattern_holds(e):
        return e['self_sensitivity'] is not None and e['self_sensitivity'] < 1e-6 and e['largest_site'] is not None and e['largest_site'] < 1e10
    return d, {(e['draw'], e['family']): pattern_holds(e) for e in d['excused']}


def test_randomised_configurations_against_oracle():
    """30 random draws (seed 2024) of shape, kernels, likelihood, link, cubature order, power, damping, sweeps and missing data
    through all three function families, every one compared at the full tolerance -- except the (draw, family) pairs listed in
    tests/golden/fuzz_excused_main.json.  That list is NOT computed here and does not look at the device: tools/fuzz_conditioning.py
    derives it from the oracle alone (the reference algorithm run twice, on y and y*(1+1e-13); listed when it moves by > 1e-8 or a
    site parameter passes 1e8).  At most one of the 60 gf / ihgp comparisons may be on it."""
    fz = _fuzz_module()
    meta, skip = _excused('fuzz_excused_main.json')
    assert meta['seed'] == 2024 and meta['n_draws'] == 30 and len(skip) <= 1
    rng = np.random.default_rng(2024)
    for i in range(30):
        desc, res, cfg = fz.one(rng, raw=True)
        for fam, v in res.items():
            if (i, fam) in skip:
                # excused from the tolerance, not from everything: clamped sites non-negative, and NaN exactly where the oracle has
                # NaN -- unless the oracle's own NaN pattern moves under the 1e-13 perturbation (self_sensitivity null in the list)
                assert cfg['weak'][fam]['ttau_nonneg'] and (cfg['weak'][fam]['same_nan_pattern'] or not skip[(i, fam)]), (i, fam, desc, cfg['weak'][fam])
                continue
            assert 0.0 <= v < TOL_MEAN, (i, fam, desc, res)      # (a comparison of all-NaN with all-NaN, fz.NOT_COMPARED = -1, is not agreement)


def test_randomised_mixtures_and_ekf_objective_against_oracle():
    """The widened rows under the same rule: 12 draws (seed 7) of J = 2-3 stacked sources through both mixture variants
    (missing data, block-structured cubature whenever Wnmf allows) and of the EKF objective; excusals only from the committed,
    oracle-derived tests/golden/fuzz_excused_widened.json (the older EP rule of the mixtures, d/alpha scaling, divides by
    1 + d2*v ~ 0 far more often -- DESIGN.md section 2)."""
    fz = _fuzz_module()
    meta, skip = _excused('fuzz_excused_widened.json')
    assert meta['seed'] == 7 and meta['n_draws'] == 12 and len(skip) <= 6
    rng = np.random.default_rng(7)
    for i in range(12):
        desc, res, cfg = fz.one_widened(rng, raw=True)
        for fam, v in res.items():
            if (i, fam) in skip:
                # excused from the tolerance, not from everything (see the main draw)
                assert cfg['weak'][fam]['ttau_nonneg'] and (cfg['weak'][fam]['same_nan_pattern'] or not skip[(i, fam)]), (i, fam, desc, cfg['weak'][fam])
                continue
            assert 0.0 <= v < TOL_MEAN, (i, fam, desc, res)      # (a comparison of all-NaN with all-NaN, fz.NOT_COMPARED = -1, is not agreement)


@pytest.mark.parametrize('link', ['exp', 'softplus'])
@pytest.mark.parametrize('kind', ['nmf', 'sqrt'])
def test_mom_exceptional_inputs_same_nan_patterns_and_floors(link, kind):
    """Overflowing link values, huge / negative cavity variances and NaN means: the device returns NaN exactly where the
    reference formulas do, and the same floored log Z (max(Z, 1e-10), SURVEY C-3/C-4)."""
    from nagp import cubature
    rng = np.random.default_rng(5); D, N, p = 4, 3, 7
    W = rng.uniform(0, 0.5, (D, N)); hyp = np.log(1e-2)
    ol = olik.exp_link() if link == 'exp' else olik.softplus_link(0.0)
    if kind == 'sqrt':
        wn, xn = cubature.utp_ws(p, N)
        mom = Mom('likModulatorPreCalcwn', link=link, wn=wn, xn_unscaled=xn); omom = olik.Mom(olik.LIK_POWER_NMF_SQRT, link=ol, wn=wn, xn_unscaled=xn)
    else:
        mom = Mom('likModulatorNMFPower', link=link, p_cubature=p); omom = olik.Mom(olik.LIK_POWER_NMF, link=ol, p=p)
    mu = rng.normal(0, 1, D + N); s2 = rng.uniform(0.1, 1, D + N)
    cases = []
    m1 = mu.copy(); m1[D] = 800.0; cases.append((m1, s2))
    s3 = s2.copy(); s3[D + 1] = 1e6; cases.append((mu, s3))
    s4 = s2.copy(); s4[0] = -0.1; cases.append((mu, s4))
    s5 = s2.copy(); s5[D] = -0.1; cases.append((mu, s5))
    m6 = mu.copy(); m6[1] = np.nan; cases.append((m6, s2))
    for m_, s_ in cases:
        y = np.array([0.3])
        with np.errstate(all='ignore'):
            o = omom(hyp, m_, s_, W, 0.5, y, 0)
        g = mom(hyp, m_, s_, W, 0.5, y, 0)
        o1 = np.real(np.ravel(o[1])); o2 = np.real(np.ravel(o[2]))
        assert np.array_equal(np.isnan(g[1]), np.isnan(o1)) and np.array_equal(np.isnan(g[2]), np.isnan(o2))
        assert abs(g[0] - float(np.real(o[0]))) < 1e-9 * max(1.0, abs(float(np.real(o[0]))))
        fin = np.isfinite(o1)
        if fin.any():
            assert np.max(np.abs(g[1][fin] - o1[fin])) <= 1e-8 * np.max(np.abs(o1[fin]))


def test_test_inputs_subset_and_unsorted_inputs():
    """xt a subset of x, x unsorted: return_ind / unique('first') semantics (gf_ep_modulator_nmf.m:58-66)."""
    D, N, T = 3, 2, 60
    pr = harness.nmf_problem(D, N, T, 8)
    perm = np.random.default_rng(0).permutation(T); x = (np.arange(1, T + 1.0))[perm]; y = pr['y'][perm]
    xt = np.array([5.0, 17.0, 3.0, 60.0, 61.0, 62.5])
    mom = Mom('likModulatorNMFPower', p_cubature=5)
    Eft, Varft = nagp.gf_ep_modulator_nmf(pr['w'], x, y, SSHandle(), mom, xt, 'matern32', 'matern52', 1, D, N, 0.5, [0.5, 0.5], 2, nargout=2)
    o = ogf.gf_ep_modulator_nmf(pr['w'], x, y, None, olik.Mom(olik.LIK_POWER_NMF, p=5), xt, 'matern32', 'matern52', 1, D, N, 0.5, [0.5, 0.5], 2)
    assert Eft.shape == (D + N, xt.size) and rel(Eft, o[0]) < TOL_MEAN and rel(Varft, o[1]) < TOL_MEAN


def test_ihgp_constraints_variant_and_exp_link():
    D, N, T = 4, 2, 150
    pr = harness.nmf_problem(D, N, T, 21, 'constraints'); t = np.arange(1, T + 1.0)
    cons = harness.CONSTRAINTS_DEMO(D); w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
    mom = Mom('likModulatorNMFPower', link='exp', p_cubature=7)
    r = nagp.ihgp_ep_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, 0.3, 3, cons, wf, harness.TUNE_DEMO, nargout=6)
    o = oih.ihgp_ep_modulator_nmf_constraints(w, t, pr['y'], None, olik.Mom(olik.LIK_POWER_NMF, link=olik.exp_link(), p=7), t, 'matern32', 'matern52',
                                              1, D, N, 0.5, 0.3, 3, cons, wf, harness.TUNE_DEMO)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and relz(r[5]['nlZ'], o[5]['nlZ']) < TOL_LOGZ


def _plan_problems(D, N, T, seeds, balance):
    probs, ys = [], []
    for s in seeds:
        pr = harness.nmf_problem(D, N, T, s)
        blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
        probs.append((pss.balance_blocks(blk) if balance else blk, pr['W'], np.log(pr['w_lik']))); ys.append(pr['y'])
    return probs, ys


def test_batched_plan_equals_single_runs_and_is_deterministic_and_chunk_invariant():
    D, N, T = 5, 2, 300
    probs, ys = _plan_problems(D, N, T, [1, 2, 3], False)
    mom = Mom('likModulatorNMFPower', p_cubature=7); kw = dict(mom=mom, ep_fraction=0.5, ep_damping=[0.5, 0.5], ep_itts=2)
    plan = Plan(L.KIND_GF_EP, probs, T, **kw); plan.upload(ys); plan.execute(); a = plan.download()
    plan.execute(); b = plan.download()                                   # re-execution: bitwise identical
    small = Plan(L.KIND_GF_EP, probs, T, chunk=64, **kw); small.upload(ys); small.execute(); c = small.download()
    for q in range(3):
        assert np.array_equal(a[q].Eft, b[q].Eft) and np.array_equal(a[q].nlZ, b[q].nlZ) and np.array_equal(a[q].ttau, b[q].ttau)
        # chunk length changes only the span decomposition of the parallel-in-time smoother: rounding-level differences
        assert rel(a[q].Eft, c[q].Eft) < 1e-11 and rel(a[q].Varft, c[q].Varft) < 1e-11 and rel(a[q].ttau, c[q].ttau) < 1e-10
        one = Plan(L.KIND_GF_EP, probs[q:q + 1], T, **kw); one.upload(ys[q:q + 1]); one.execute(); o = one.download()[0]
        assert np.array_equal(a[q].Eft, o.Eft) and np.array_equal(a[q].Varft, o.Varft) and np.array_equal(a[q].nlZ, o.nlZ)
        one.close()
    assert np.allclose(plan.download_nlz(), np.array([x.nlZ for x in a]))
    t = plan.timings()    # sweep 1 + the ADF step k = T-1 of sweep 2 | the fixed-site steps of sweep 2
    assert t['launches']['filter'] == 2 and t['launches']['filter_lin'] == 1 and t['ms']['scan'] > 0
    plan.close(); small.close()


def test_large_batch_of_replicas_equals_single_problem_plans():
    """112 problems in one plan (more workgroups than half the CUs in every kernel): each must agree with its
    single-problem plan to rounding."""
    D, N, T = 3, 2, 90
    probs2, ys2 = _plan_problems(D, N, T, [11, 12], False)
    B = 112
    probs = [probs2[q % 2] for q in range(B)]; ys = [ys2[q % 2] for q in range(B)]
    mom = Mom('likModulatorNMFPower', p_cubature=5); kw = dict(mom=mom, ep_fraction=0.5, ep_damping=[0.5, 0.5], ep_itts=2)
    big = Plan(L.KIND_GF_EP, probs, T, **kw); big.upload(ys); big.execute(); a = big.download()
    t = big.timings(); assert t['launches']['scan'] == 6      # one chunk per sweep: compose, boundary and apply passes
    for q in (0, 1):
        one = Plan(L.KIND_GF_EP, probs2[q:q + 1], T, **kw); one.upload(ys2[q:q + 1]); one.execute(); o = one.download()[0]
        for qq in (q, q + 2, B - 2 + q):
            assert rel(a[qq].Eft, o.Eft) < 1e-11 and rel(a[qq].Varft, o.Varft) < 1e-11 and rel(a[qq].MS, o.MS) < 1e-11
            assert rel(a[qq].ttau, o.ttau) < 1e-10 and np.allclose(a[qq].nlZ, o.nlZ, rtol=1e-12)
        one.close()
    big.close()


@pytest.mark.parametrize('D,N', [(12, 3), (20, 3), (21, 3), (25, 2)])
def test_smoother_size_classes_against_oracle(D, N):
    """M = 15 / 23 / 24 / 27 sites: dense MFMA smoother tiles Sp = 64 and 96 (the largest that fits the LDS, instantiations
    <4> and <6>) and the first size that falls back to the VALU span kernels (Sp = 112), two tiles per thread in the gain
    kernel from M = 23 on."""
    T = 36
    pr = harness.nmf_problem(D, N, T, 400 + D); t = np.arange(1, T + 1.0); d = np.array([0.5, 0.5])
    mom = Mom('likModulatorNMFPower', p_cubature=5); omom = olik.Mom(olik.LIK_POWER_NMF, p=5)
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, pr['y'], None, omom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(Eft, o[0]) < TOL_MEAN and rel(Varft, o[1]) < TOL_MEAN
    assert rel(out['PS'], np.transpose(o[5]['PS'], (1, 2, 0))) < TOL_MEAN and rel(out['MS'], o[5]['MS']) < TOL_MEAN
    assert np.allclose(out['nlZ'], o[5]['nlZ'], rtol=TOL_LOGZ, atol=1e-12)


def test_ekf_with_two_tiles_per_thread_S146():
    """gf_giekf_modulator_nmf at the 32-channel / 6-component shape (M = 38, 741 lower tiles -> two tiles per thread in the
    EKF filter instantiation, three in the smoother kernels)."""
    D, N, T = 32, 6, 24
    pr = harness.nmf_problem(D, N, T, 77, 'constraints'); t = np.arange(1, T + 1.0)
    r = nagp.gf_giekf_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), None, t, 'matern32', 'matern52', 1, D, N, 2, 2, nargout=2)
    o = oek.gf_giekf_modulator_nmf(pr['w'], t, pr['y'], None, None, t, 'matern32', 'matern52', 1, D, N, 2, 2)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN


def test_ihgp_adf_sites_of_underflow_size():
    """Draw 8 of seed 99 of tools/gpu_fuzz.py (7 channels / 6 components, p = 7): from step 27 on the likelihood underflows, the sites
    are denormal (ttau ~ 6e-310), R = 1/ttau overflows to inf while ys = tnu/ttau stays finite.  The reciprocal-based tail of the
    staged IHGP ADF kernels multiplied a denormal by inf there (NaN); the reference's own divisions give gain 0.  All three kernel
    forms against the oracle."""
    fz = _fuzz_module()
    rng = np.random.default_rng(99)
    for _ in range(9):
        c = fz.draw(rng)
    D, N, T, k1, k2, alpha, damp, pr = (c[k] for k in ('D', 'N', 'T', 'k1', 'k2', 'alpha', 'damp', 'pr'))
    t = np.arange(1, T + 1.0); mom, omom = fz.moms(c)
    o = oih.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], None, omom, t, k1, k2, 1, D, N, alpha, damp[:1], 1)
    assert np.nanmin(np.abs(o[5]['ttau'][o[5]['ttau'] != 0])) < 1e-300          # the case is what it claims to be
    for env in ({}, {'NAGP_IH_PACK': '0'}, {'NAGP_IH_ROLES': '0'}):
        for k_ in ('NAGP_IH_PACK', 'NAGP_IH_ROLES'):
            os.environ.pop(k_, None)
        os.environ.update(env)
        try:
            r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, k1, k2, 1, D, N, alpha, damp[:1], 1, nargout=6)
        finally:
            for k_ in env: os.environ.pop(k_, None)
        assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and relz(r[5]['nlZ'], o[5]['nlZ']) < TOL_LOGZ, env
        assert not np.any(np.isnan(r[5]['tnu'])), env


def test_ekf_two_state_blocks_odd_lds_offset():
    """gf_giekf_modulator_nmf with 2-state blocks only (cos x exp sub-bands, Matern-3/2 modulators; 11 sites, 22 states): the
    arrays in front of the W panel of the filter kernel add up to an odd number of doubles, the panel moves to the next 16-byte
    boundary and the EKF workspace behind it is used to its last element (found by a fresh-seed run of tools/gpu_fuzz.py:
    the LDS size did not count the alignment pad and the last gain entry fell off the allocation)."""
    D, N, T = 7, 4, 62
    pr = harness.nmf_problem(D, N, T, 4242); t = np.arange(1, T + 1.0)
    y = pr['y'].copy(); y[[5, 30, 31]] = np.nan
    for g_iter in (1, 3):
        r = nagp.gf_giekf_modulator_nmf(pr['w'], t, y, SSHandle(), None, t, 'exp', 'matern32', 1, D, N, g_iter, 2, nargout=2)
        o = oek.gf_giekf_modulator_nmf(pr['w'], t, y, None, None, t, 'exp', 'matern32', 1, D, N, g_iter, 2)
        assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN, g_iter


def test_full_length_cfg2_prefix_property_and_finiteness():
    """BASELINE size (T = 84 010, S = 73) through a size-independent property: with one sweep the sites of
    step k depend only on y(1..k), so the first 1500 columns must equal the truncated golden run's filter
    pass; outputs finite; variances positive."""
    g = gold('cfg2_gf_ep_modulator_nmf'); D, N = int(g['D']), int(g['N'])
    T = 84010
    pr = harness.nmf_problem(D, N, T, 100)             # same seed -> same model as the fixture; prior sample differs in length only
    y = np.concatenate([g['y'], pr['y'][g['y'].size:]])
    blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    mom = Mom('likModulatorNMFPower', p_cubature=9)
    plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=0.5, ep_damping=[0.5], ep_itts=1)
    plan.upload([y]); plan.execute(); out = plan.download(want_MS=False)[0]; plan.close()
    Tg = g['y'].size
    ref = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], Tg, mom=mom, ep_fraction=0.5, ep_damping=[0.5], ep_itts=1)
    ref.upload([g['y']]); ref.execute(); o = ref.download(want_MS=False)[0]; ref.close()
    assert np.array_equal(out.ttau[:, :Tg - 1], o.ttau[:, :Tg - 1]) and np.array_equal(out.lZ[:Tg - 1], o.lZ[:Tg - 1])
    o1 = ogf.gf_ep_modulator_nmf(g['w'], np.arange(1, 201.0), g['y'][:200], None, olik.Mom(olik.LIK_POWER_NMF, p=9), np.arange(1, 201.0),
                                 'matern32', 'matern52', 1, D, N, 0.5, [0.5], 1)
    assert rel(out.ttau[:, :199], o1[5]['ttau'][:, :199]) < TOL_SITE
    assert np.all(np.isfinite(out.Eft)) and np.all(out.Varft > 0) and np.isfinite(out.nlZ[0])


def test_golden_sixstate_matern52_subbands_all_three_families():
    """kernel1 = 'matern52' (ss_modulators_nmf.m:13-33, cf_matern52_to_ss.m:93-121): 6-state sub-band blocks, D = 8, N = 3 -- the full-covariance
    plans split each block over two tile rows, the infinite-horizon plans keep it whole (block stride 8)."""
    g = gold('sixstate_matern52_subbands'); D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorNMFPower', p_cubature=7); d = 0.5 * np.ones(3); k1 = k2 = 'matern52'
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(g['w'], t, g['y'], SSHandle(), mom, t, k1, k2, 1, D, N, 0.5, d, 3, nargout=6)
    assert rel(Eft, g['gf_Eft']) < TOL_MEAN and rel(Varft, g['gf_Varft']) < TOL_MEAN and relz(out['nlZ'], g['gf_nlZ']) < TOL_LOGZ
    assert rel(out['ttau'], g['gf_ttau']) < TOL_SITE and rel(out['tnu'], g['gf_tnu']) < TOL_SITE and rel(out['lZ'], g['gf_lZ']) < 1e-7
    assert rel(out['maxDiffM'], g['gf_maxDiffM']) < 1e-6 and rel(out['maxDiffP'], g['gf_maxDiffP']) < 1e-6
    assert out['counters']['nan_obs'] == 3 * 11 and out['counters']['chol_retries'] == 0
    assert out['ttau'].shape == (D + N, T) and out['MS'].shape == (6 * D + 3 * N, T)
    e, _ = nagp.gf_ep_modulator_nmf(g['w'], t, g['y'], SSHandle(), mom, None, k1, k2, 1, D, N, 0.5, d, 3)
    assert abs(e - float(g['gf_edata_I3'])) < TOL_LOGZ * abs(e)
    Eft, Varft, _, _, _, out = nagp.ihgp_ep_modulator_nmf(g['w'], t, g['y'], SSHandle(), mom, t, k1, k2, 1, D, N, 0.5, d, 3, nargout=6)
    # two sets of look-up tables: the oracle's own (SciPy's DARE solver) and the host's (batched doubling) agree to 1e-8 .. 1e-6 on a 6-state block
    # (steady-state covariances conditioned ~1e8), so logZ and the sites are held to 1e-6 / 1e-4 against the independent tables and the
    # kernels are pinned by the oracle run on the HOST's tables (ihh_*)
    assert rel(Eft, g['ih_Eft']) < TOL_MEAN and rel(Varft, g['ih_Varft']) < TOL_MEAN and relz(out['nlZ'], g['ih_nlZ']) < 1e-6
    assert rel(out['ttau'], g['ih_ttau']) < 1e-4 and rel(out['tnu'], g['ih_tnu']) < 1e-4
    assert rel(Eft, g['ihh_Eft']) < 1e-9 and rel(Varft, g['ihh_Varft']) < 1e-9 and relz(out['nlZ'], g['ihh_nlZ']) < TOL_LOGZ and rel(out['ttau'], g['ihh_ttau']) < 1e-4
    Eft, Varft, _, _, _, out = nagp.gf_giekf_modulator_nmf(g['w'], t, g['y_ekf'], SSHandle(), None, t, k1, k2, 1, D, N, 3, 2, nargout=6)
    assert rel(Eft, g['ekf_Eft']) < TOL_MEAN and rel(Varft, g['ekf_Varft']) < TOL_MEAN and rel(out['maxDiffP'], g['ekf_maxDiffP']) < 1e-6


@pytest.mark.parametrize('k1', ['matern52', 'matern72'])
def test_blocks_of_more_than_four_states_against_the_oracle(k1):
    """6- and 8-state sub-band blocks through every entry point of the full-covariance path (EP predict / nlml, plain likelihood with M = 2D,
    constraints variants, EKF smoother and energy), against the oracle on fresh problems."""
    D, N, T = 5, 2, 160
    pr = harness.nmf_problem(D, N, T, 9, 'constraints', kernel1=k1); t = np.arange(1, T + 1.0); y = pr['y'].copy(); y[50:58] = np.nan
    mom = Mom('likModulatorNMFPower', p_cubature=7); om = olik.Mom(olik.LIK_POWER_NMF, p=7)
    r = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, k1, 'matern52', 1, D, N, 0.5, [0.5, 0.4], 2, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, om, t, k1, 'matern52', 1, D, N, 0.5, [0.5, 0.4], 2)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['ttau'], o[5]['ttau']) < TOL_SITE and relz(r[5]['nlZ'], o[5]['nlZ']) < TOL_LOGZ
    assert rel(r[5]['MS'], o[5]['MS']) < TOL_MEAN and rel(r[5]['PS'], np.transpose(o[5]['PS'], (1, 2, 0))) < TOL_MEAN      # (the caller's state order)
    for I in (1, 3):
        e, _ = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, None, k1, 'matern52', 1, D, N, 0.5, [0.5] * I, I)
        eo, _ = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, om, None, k1, 'matern52', 1, D, N, 0.5, [0.5] * I, I)
        assert abs(e - eo) < TOL_LOGZ * abs(eo)
    cons = harness.CONSTRAINTS_DEMO(D); w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
    r = nagp.gf_ep_modulator_nmf_constraints(w, t, y, SSHandle(), mom, t, k1, 'matern52', 1, D, N, 0.5, 0.5 * np.ones(2), 2, cons, wf, harness.TUNE_DEMO, nargout=6)
    o = ogf.gf_ep_modulator_nmf_constraints(w, t, y, None, om, t, k1, 'matern52', 1, D, N, 0.5, 0.5 * np.ones(2), 2, cons, wf, harness.TUNE_DEMO)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['ttau'], o[5]['ttau']) < TOL_SITE
    r = nagp.gf_giekf_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), None, t, k1, 'matern52', 1, D, N, 2, 2, nargout=6)
    o = oek.gf_giekf_modulator_nmf(pr['w'], t, pr['y'], None, None, t, k1, 'matern52', 1, D, N, 2, 2)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['maxDiffP'], o[5]['maxDiffP']) < 1e-6
    r = nagp.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), None, t, k1, 'matern52', 1, D, N, 3, 1, cons, wf, harness.TUNE_DEMO, nargout=6)
    o = oek.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], None, None, t, k1, 'matern52', 1, D, N, 3, 1, cons, wf, harness.TUNE_DEMO)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN
    e, eg = nagp.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), None, None, k1, 'matern52', 1, D, N, 3, 2, cons, wf, harness.TUNE_DEMO, 'off')
    eo, _ = oek.gf_giekf_modulator_nmf_constraints_nlml(w, t, pr['y'], k1, 'matern52', 1, D, N, cons, wf, harness.TUNE_DEMO)
    assert abs(e - eo) < TOL_LOGZ * abs(eo)
    # the plain likelihood: one modulator per sub-band, M = 2D sites, balanced blocks, prediction at the first step
    D1, T1 = 4, 240
    param = np.concatenate([np.full(D1, 0.1), [50.0, 40.0, 50.0, 40.0], [np.pi / 4, np.pi / 6, np.pi / 8, np.pi / 10], [2.0, 3.0, 2.0, 3.0], [500.0, 700.0, 500.0, 700.0]])
    y1 = harness.sample_prior(pss.ss_blocks_nmf(param[:3 * D1], param[3 * D1:], k1, 'matern52'), None, T1, np.random.default_rng(3))
    w1 = np.log(np.concatenate([[1e-5], param])); t1 = np.arange(1, T1 + 1.0)
    r = nagp.gf_ep_modulator(w1, t1, y1, SSHandle('ss_modulators'), Mom('likModulatorPower', p_cubature=9), t1, k1, 'matern52', 1, 0.5, 0.3 * np.ones(3), 3, nargout=6)
    o = ogf.gf_ep_modulator(w1, t1, y1, None, olik.Mom(olik.LIK_POWER, p=9), t1, k1, 'matern52', 1, 0.5, 0.3 * np.ones(3), 3)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['ttau'], o[5]['ttau']) < TOL_SITE and relz(r[5]['nlZ'], o[5]['nlZ']) < TOL_LOGZ


@pytest.mark.parametrize('k1', ['matern52', 'matern72'])
def test_infinite_horizon_sweeps_with_blocks_of_more_than_four_states(k1):
    """ihgp_ep_modulator_nmf{,_constraints} with 6- and 8-state sub-band blocks.  The steady-state covariances of such blocks are badly conditioned
    (1e8 for Matern-5/2, 1e12 for an 8-state Matern-7/2 block): two DARE solvers with residuals of 1e-13 each (the oracle's SciPy one, the host's batched
    doubling) give tables 1e-8 .. 1e-5 apart.  The kernels are pinned by the oracle run on the HOST's tables; the independent tables are held to what
    their own agreement allows."""
    D, N, T = 6, 2, 260
    pr = harness.nmf_problem(D, N, T, 13, kernel1=k1); t = np.arange(1, T + 1.0); y = pr['y'].copy(); y[70:75] = np.nan
    mom = Mom('likModulatorNMFPower', p_cubature=7); om = olik.Mom(olik.LIK_POWER_NMF, p=7); d = [0.5, 0.4, 0.3]
    r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, k1, 'matern52', 1, D, N, 0.5, d, 3, nargout=6)
    o = oih.ihgp_ep_modulator_nmf(pr['w'], t, y, None, om, t, k1, 'matern52', 1, D, N, 0.5, d, 3)
    if k1 == 'matern52':      # against the oracle's own tables (Matern-7/2: SciPy's solver itself warns "ill-conditioned, rcond = 8e-17" on these blocks
        # and its tables are off by up to 10 % of the posterior mean on this instance -- nothing to hold the product to)
        assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['ttau'], o[5]['ttau']) < 1e-4 and relz(r[5]['nlZ'], o[5]['nlZ']) < 1e-6
    from nagp import ihgp_tables
    lik, p1, p2, W = oss.unpack_log(pr['w'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, k1, 'matern52', True, True)
    blk = pss.balance_blocks(pss.ss_blocks_nmf(p1, p2, k1, 'matern52'))
    A, Q, _ = pss.discretise(blk, symmetrize_Q=True)
    r2, PP, ppo, PG, pgo = ihgp_tables.build_tables(A, Q, blk.offsets, blk.h_val)
    PPl = [PP[ppo[n]:ppo[n] + 200 * blk.sizes[n] ** 2].reshape(200, -1) for n in range(D + N)]
    PGl = [PG[pgo[n]:pgo[n] + 400 * blk.sizes[n] ** 2].reshape(200, -1) for n in range(D + N)]
    res = oih.run_predict(model, y, om, 0.5, np.asarray(d), 3, tables=(oih.build_tables(model)[0], r2, PPl, PGl))      # ... and on the host's
    assert rel(r[0], res['Eft']) < TOL_MEAN and rel(r[1], res['Varft']) < TOL_MEAN and rel(r[5]['ttau'], res['ttau']) < 1e-4 and relz(r[5]['nlZ'], res['nlZ']) < TOL_LOGZ
    cons = harness.CONSTRAINTS_DEMO(D); prc = harness.nmf_problem(D, N, T, 13, 'constraints', kernel1=k1); w, wf = harness.constrained_vectors(prc, cons, harness.TUNE_DEMO)
    if k1 == 'matern52':
        r = nagp.ihgp_ep_modulator_nmf_constraints(w, t, prc['y'], SSHandle(), mom, t, k1, 'matern52', 1, D, N, 0.5, 0.3, 2, cons, wf, harness.TUNE_DEMO, nargout=6)
        o = oih.ihgp_ep_modulator_nmf_constraints(w, t, prc['y'], None, om, t, k1, 'matern52', 1, D, N, 0.5, 0.3, 2, cons, wf, harness.TUNE_DEMO)
        assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and relz(r[5]['nlZ'], o[5]['nlZ']) < 1e-6


def test_split_block_plans_batches_chunks_warm_starts_and_many_tile_rows():
    """Plans with split blocks: a batch equals its single runs, a chunked pipelined smoother equals the oracle, sites round-trip through the warm start
    in the caller's M columns, the tile exchange in two and three phases (more than 25 tile rows: D = 16 -> 35, D = 20 -> 43) agrees with the oracle."""
    k1 = 'matern52'; mom = Mom('likModulatorNMFPower', p_cubature=7); om = olik.Mom(olik.LIK_POWER_NMF, p=7)
    for (D, N, T) in [(4, 2, 90), (16, 3, 30), (20, 3, 10), (24, 3, 6)]:      # (19, 35, 43, 51 tile rows: one, two, three exchange phases; eight tiles per thread in the gain kernel)
        probs, ys, orc = [], [], []
        for sd in (1, 2):
            pr = harness.nmf_problem(D, N, T, sd, kernel1=k1)
            probs.append((pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, 'matern52'), pr['W'], np.log(pr['w_lik']))); ys.append(pr['y'])
            orc.append(ogf.gf_ep_modulator_nmf(pr['w'], np.arange(1, T + 1.0), pr['y'], None, om, np.arange(1, T + 1.0), k1, 'matern52', 1, D, N, 0.5, [0.5, 0.5], 2))
        plan = Plan(L.KIND_GF_EP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=[0.5, 0.5], ep_itts=2, flags=0x4)
        plan.upload(ys); plan.execute(); outs = plan.download(want_MS=True, want_PS=True, want_MF=True)
        for q in range(2):
            o = orc[q]
            assert rel(outs[q].Eft, o[0]) < TOL_MEAN and rel(outs[q].Varft, o[1]) < TOL_MEAN and rel(outs[q].ttau, o[5]['ttau']) < TOL_SITE
            assert rel(outs[q].MS, o[5]['MS']) < TOL_MEAN and rel(outs[q].MF, o[5]['MF']) < TOL_MEAN and rel(outs[q].PS, np.transpose(o[5]['PS'], (1, 2, 0))) < TOL_MEAN
        if D == 4:
            one = Plan(L.KIND_GF_EP, probs[1:], T, mom=mom, ep_fraction=0.5, ep_damping=[0.5, 0.5], ep_itts=2); one.upload(ys[1:]); one.execute(); o1 = one.download()[0]; one.close()
            assert np.array_equal(o1.Eft, outs[1].Eft) and np.array_equal(o1.ttau, outs[1].ttau) and np.array_equal(o1.nlZ, outs[1].nlZ)
        plan.close()
    D, N, T = 6, 2, 700
    pr = harness.nmf_problem(D, N, T, 4, kernel1=k1); t = np.arange(1, T + 1.0)
    a = Plan(L.KIND_GF_EP, [(pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, 'matern52'), pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=0.5,
             ep_damping=[0.5] * 3, ep_itts=3, chunk=128)
    a.upload([pr['y']]); a.execute(); oa = a.download()[0]
    o3 = ogf.gf_ep_modulator_nmf(pr['w'], t, pr['y'], None, om, t, k1, 'matern52', 1, D, N, 0.5, [0.5] * 3, 3)
    assert rel(oa.Eft, o3[0]) < TOL_MEAN and rel(oa.Varft, o3[1]) < TOL_MEAN and rel(oa.ttau, o3[5]['ttau']) < TOL_SITE and relz(oa.nlZ, o3[5]['nlZ']) < TOL_LOGZ
    assert oa.ttau.shape == (D + N, T)
    a.upload_sites([oa.ttau], [oa.tnu]); a.execute(); ob = a.download()[0]; a.close()
    assert np.all(np.isfinite(ob.Eft)) and ob.ttau.shape == (D + N, T) and not np.array_equal(ob.ttau, oa.ttau)


@pytest.mark.parametrize('k1', ['matern52', 'matern72'])
def test_split_blocks_edge_lengths_and_tiny_shapes(k1):
    """T = 1, 2, 3 and lengths around the I/O ring, one sub-band / one modulator, NaN at both ends: full-covariance EP and EKF with split blocks."""
    mom = Mom('likModulatorNMFPower', p_cubature=5); om = olik.Mom(olik.LIK_POWER_NMF, p=5)
    for (D, N) in [(1, 1), (2, 3), (7, 1)]:
        for T in (1, 2, 3, 17, 33):
            pr = harness.nmf_problem(D, N, max(T, 8), 3 + D + T, kernel1=k1); t = np.arange(1, T + 1.0); y = pr['y'][:T].copy()
            if T > 2: y[1] = np.nan
            if T > 16: y[-1] = np.nan; y[0] = np.nan
            r = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, k1, 'matern52', 1, D, N, 0.5, [0.5, 0.5], 2, nargout=6)
            o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, om, t, k1, 'matern52', 1, D, N, 0.5, [0.5, 0.5], 2)
            assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['ttau'], o[5]['ttau']) < TOL_SITE and relz(r[5]['nlZ'], o[5]['nlZ']) < TOL_LOGZ, (D, N, T)
            r = nagp.gf_giekf_modulator_nmf(pr['w'], t, y, SSHandle(), None, t, k1, 'matern52', 1, D, N, 2, 2, nargout=2)
            o = oek.gf_giekf_modulator_nmf(pr['w'], t, y, None, None, t, k1, 'matern52', 1, D, N, 2, 2)
            assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN, (D, N, T)


@pytest.mark.parametrize('k1,D,N', [('matern72', 8, 6), ('matern52', 5, 4)])
def test_ekf_inner_iterations_with_split_blocks_and_many_modulators(k1, D, N):
    """iekf_update1 with l_iter = 2, 3 on plans with split blocks: the state lanes that refresh softplus / sigmoid of the modulators for the next inner
    iteration must leave the tail rows out (found by the fuzz draws with --k1 matern72: a tail row counted as a modulator overwrote the sigmoid table)."""
    T = 70
    pr = harness.nmf_problem(D, N, T, 31, kernel1=k1); t = np.arange(1, T + 1.0); y = pr['y'].copy(); y[[5, 6, 40, 69]] = np.nan
    for (gi, li) in [(1, 2), (2, 2), (2, 3)]:
        r = nagp.gf_giekf_modulator_nmf(pr['w'], t, y, SSHandle(), None, t, k1, 'matern32', 1, D, N, gi, li, nargout=2)
        o = oek.gf_giekf_modulator_nmf(pr['w'], t, y, None, None, t, k1, 'matern32', 1, D, N, gi, li)
        assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN, (gi, li)


def test_mixture_variants_with_a_six_state_source():
    """experiments/{gf,ihgp}_ep_mods_nmf_mixture.m with Matern-5/2 sub-bands in one of two stacked sources (the older EP rule: NAGP_FLAG_MIXTURE_RULE)."""
    shapes = [(3, 1), (2, 2)]; k1 = ['matern52', 'matern32']; k2 = ['matern52', 'matern52']; T = 80
    mp = harness.mixture_problem(shapes, T, 21, k1, k2); t = np.arange(1, T + 1.0)
    y = mp['y'].copy(); y[30:36] = np.nan
    mom = Mom('likModulatorNMFPower', p_cubature=7); om = olik.Mom(olik.LIK_POWER_NMF, p=7)
    r = nagp.gf_ep_mods_nmf_mixture(mp['w'], t, y, SSHandle(), mom, t, k1, k2, 2, 0.75, 0.2, 4, nargout=6)
    o = omx.gf_ep_mods_nmf_mixture(mp['w'], t, y, None, om, t, k1, k2, 2, 0.75, 0.2, 4)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['ttau'], o[5]['ttau']) < TOL_SITE
    # infinite horizon: two sweeps (by the fourth this instance divides by 1 + d2 v ~ 0 -- the mixtures' rule, DESIGN section 2 -- and amplifies the 1e-8 by which
    # the two builders of the look-up tables differ on a six-state block to O(1))
    r = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 2, 0.75, 0.2, 2, nargout=6)
    o = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], None, om, t, k1, k2, 2, 0.75, 0.2, 2)
    assert rel(r[0], o[0]) < TOL_MEAN and rel(r[1], o[1]) < TOL_MEAN and rel(r[5]['ttau'], o[5]['ttau']) < TOL_SITE


def test_unsupported_shapes_are_refused_not_emulated():
    # more tile rows (sites + split blocks) than the filter's LDS holds (panel W = P H', exchange buffer, cubature workspace): 33 sites + 30 six-state sub-bands = 63
    D, N, T = 30, 3, 8
    pr = harness.nmf_problem(D, N, T, 1, kernel1='matern52')
    t = np.arange(1, T + 1.0)
    with pytest.raises(nagp.NagpError, match='unsupported'):
        nagp.gf_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), Mom('likModulatorNMFPower', p_cubature=5), t, 'matern52', 'matern52', 1, D, N, 0.5, [0.5], 1)


def _mixture_moms(lik, p, N):
    if lik == 'likModulatorPreCalcwn':
        from nagp import cubature
        wn, xn = cubature.utp_ws(p, N)
        return Mom(lik, link_shift=1.0, wn=wn, xn_unscaled=xn), olik.Mom(olik.LIK_POWER_NMF_SQRT, link=olik.softplus_link(1.0), wn=wn, xn_unscaled=xn)
    return Mom(lik, p_cubature=p), olik.Mom(olik.LIK_POWER_NMF, p=p)


@pytest.mark.parametrize('shapes,k1,k2,lik,p', [
    ([(3, 1), (2, 2)], ['exp', 'matern32'], ['matern52', 'matern52'], 'likModulatorNMFPower', 7),
    ([(4, 2), (4, 2), (3, 1)], ['exp', 'exp', 'exp'], ['matern52', 'matern32', 'matern52'], 'likModulatorPreCalcwn', 7),
])
def test_source_separation_mixtures_against_oracle(shapes, k1, k2, lik, p):
    """experiments/gf_ep_mods_nmf_mixture.m and ihgp_ep_mods_nmf_mixture.m (row f-1): J stacked models with their own
    kernels, block-diagonal Wnmf, the older EP rule (mom at power ep_fraction in the filter too, d/ep_fraction
    scaling, clamp in the filter pass, R before the clamp); ep_fraction = 0.75 as in source_sep_piano.m:86."""
    T = 36; t = np.arange(1, T + 1.0); J = len(shapes); N = sum(n for _, n in shapes)
    mp = harness.mixture_problem(shapes, T, 21, k1, k2)
    y = mp['y'].copy(); y[10:13] = np.nan
    mom, omom = _mixture_moms(lik, p, N)
    a = nagp.gf_ep_mods_nmf_mixture(mp['w'], t, y, SSHandle(), mom, t, k1, k2, J, 0.75, 0.2, 3, nargout=6)
    b = omx.gf_ep_mods_nmf_mixture(mp['w'], t, y, None, omom, t, k1, k2, J, 0.75, 0.2, 3)
    assert rel(a[0], b[0]) < TOL_MEAN and rel(a[1], b[1]) < TOL_MEAN
    assert rel(a[5]['ttau'], b[5]['ttau']) < TOL_SITE and rel(a[5]['tnu'], b[5]['tnu']) < TOL_SITE
    assert rel(a[5]['lZ'], b[5]['lZ']) < 1e-7 and rel(a[5]['MS'], b[5]['MS']) < TOL_MEAN
    fin = np.isfinite(b[5]['R'])
    assert np.array_equal(fin, np.isfinite(a[5]['R'])) and rel(a[5]['R'][fin], b[5]['R'][fin]) < 1e-6
    assert np.all(a[5]['ttau'] >= 0)                                         # sites clamped by the last filter pass (:195)
    c = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, y, SSHandle(), mom, t, k1, k2, J, 0.75, 0.2, 3, nargout=6)
    d = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, y, None, omom, t, k1, k2, J, 0.75, 0.2, 3)
    assert rel(c[0], d[0]) < TOL_MEAN and rel(c[1], d[1]) < TOL_MEAN
    assert rel(c[5]['ttau'], d[5]['ttau']) < TOL_SITE and rel(c[5]['tnu'], d[5]['tnu']) < TOL_SITE
    with pytest.raises(RuntimeError):
        nagp.gf_ep_mods_nmf_mixture(mp['w'], t, y, SSHandle(), mom, None, k1, k2, J, 0.75, 0.2, 3)


def test_three_sources_with_nine_modulators_ihgp_mixture():
    """The shape family of source_sep_piano.m:78-90 (three sources x three NMF components, exp sub-band kernels,
    likModulatorPreCalcwn with the shifted softplus): cubature dimension 9 on the infinite-horizon path AND, since round 3, in the
    ADF launches of the full-covariance filter (gf_ep_mods_nmf_mixture with nine components runs while the 21 sites fit)."""
    shapes = [(4, 3)] * 3; k1 = ['exp'] * 3; k2 = ['matern52'] * 3
    T = 28; t = np.arange(1, T + 1.0)
    mp = harness.mixture_problem(shapes, T, 5, k1, k2)
    mom, omom = _mixture_moms('likModulatorPreCalcwn', 7, 9)
    c = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 3, 0.75, 0.2, 3, nargout=6)
    d = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], None, omom, t, k1, k2, 3, 0.75, 0.2, 3)
    assert c[0].shape == (21, T)
    assert rel(c[0], d[0]) < TOL_MEAN and rel(c[1], d[1]) < TOL_MEAN
    assert rel(c[5]['ttau'], d[5]['ttau']) < TOL_SITE and rel(c[5]['tnu'], d[5]['tnu']) < TOL_SITE
    a = nagp.gf_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 3, 0.75, 0.2, 3, nargout=6)
    b = omx.gf_ep_mods_nmf_mixture(mp['w'], t, mp['y'], None, omom, t, k1, k2, 3, 0.75, 0.2, 3)
    assert rel(a[0], b[0]) < TOL_MEAN and rel(a[1], b[1]) < TOL_MEAN
    assert rel(a[5]['ttau'], b[5]['ttau']) < TOL_SITE and rel(a[5]['tnu'], b[5]['tnu']) < TOL_SITE and rel(a[5]['MS'], b[5]['MS']) < TOL_MEAN


def test_nine_nmf_components_full_covariance_against_oracle():
    """gf_ep_modulator_nmf with nine components (cubature dimension 9, ut3: 19 points): the ADF kernels with covariance tiles."""
    D, N, T = 8, 9, 48
    pr = harness.nmf_problem(D, N, T, 9901); t = np.arange(1, T + 1.0)
    y = pr['y'].copy(); y[11] = np.nan
    d = 0.5 * np.ones(3)
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), Mom('likModulatorNMFPower', p_cubature=3), t, 'matern32', 'matern52', 1, D, N,
                                                        0.5, d, 3, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, olik.Mom(olik.LIK_POWER_NMF, p=3), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3)
    assert rel(Eft, o[0]) < TOL_MEAN and rel(Varft, o[1]) < TOL_MEAN and relz(out['nlZ'], o[5]['nlZ']) < TOL_LOGZ
    assert rel(out['ttau'], o[5]['ttau']) < TOL_SITE and rel(out['tnu'], o[5]['tnu']) < TOL_SITE


def test_block_structured_cubature_equals_per_point_evaluation(monkeypatch):
    """With a block-diagonal Wnmf the library evaluates the amplitudes at the distinct projections of the sigma points
    per source (mom_src) instead of at every point.  NAGP_NO_SRC=1 keeps the per-point evaluation: both must agree to
    rounding, also when the blocks are uneven, and a Wnmf whose blocks are not contiguous must take the per-point
    path by itself (same results as the oracle either way)."""
    shapes = [(5, 2), (2, 1), (4, 3)]; k1 = ['exp', 'matern32', 'exp']; k2 = ['matern52'] * 3
    T = 40; t = np.arange(1, T + 1.0)
    mp = harness.mixture_problem(shapes, T, 8, k1, k2)
    mom, omom = _mixture_moms('likModulatorPreCalcwn', 7, 6)
    a = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 3, 0.75, 0.2, 3, nargout=6)
    monkeypatch.setenv('NAGP_NO_SRC', '1')
    b = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 3, 0.75, 0.2, 3, nargout=6)
    monkeypatch.delenv('NAGP_NO_SRC')
    # (rounding differences of the two summation orders, amplified by three EP sweeps)
    assert rel(a[0], b[0]) < 1e-8 and rel(a[5]['ttau'], b[5]['ttau']) < 1e-7 and rel(a[5]['tnu'], b[5]['tnu']) < 1e-7
    o = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], None, omom, t, k1, k2, 3, 0.75, 0.2, 3)
    assert rel(a[0], o[0]) < TOL_MEAN and rel(a[5]['ttau'], o[5]['ttau']) < TOL_SITE
    # missing observations: the infinite-horizon filter has no isnan guard (ihgp_ep_mods_nmf_mixture.m:284-302), mom sees
    # y = NaN -> Z falls to the floor, NaN moments; both evaluations and the oracle must produce the same NaN pattern
    yn = mp['y'].copy(); yn[7] = np.nan
    with np.errstate(all='ignore'):
        an = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, yn, SSHandle(), mom, t, k1, k2, 3, 0.75, 0.2, 2, nargout=6)
        on = omx.ihgp_ep_mods_nmf_mixture(mp['w'], t, yn, None, omom, t, k1, k2, 3, 0.75, 0.2, 2)
    monkeypatch.setenv('NAGP_NO_SRC', '1')
    bn = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, yn, SSHandle(), mom, t, k1, k2, 3, 0.75, 0.2, 2, nargout=6)
    monkeypatch.delenv('NAGP_NO_SRC')
    for key in ('ttau', 'tnu'):
        assert np.array_equal(np.isnan(an[5][key]), np.isnan(on[5][key])) and np.array_equal(np.isnan(an[5][key]), np.isnan(bn[5][key]))
    assert np.array_equal(np.isnan(an[0]), np.isnan(on[0])) and np.array_equal(np.isnan(an[0]), np.isnan(bn[0]))
    # the main functions see the same structure: a GT-NMF model whose Wnmf happens to be block diagonal ...
    D, N = 6, 4
    pr = harness.nmf_problem(D, N, T, 12)
    W = pr['W'].copy(); W[:3, 2:] = 0.0; W[3:, :2] = 0.0
    mom2 = Mom('likModulatorNMFPower', p_cubature=7); omom2 = olik.Mom(olik.LIK_POWER_NMF, p=7)
    with np.errstate(divide='ignore'):
        w = np.concatenate([pr['w'][:1 + 3 * D + 2 * N], np.log(W.flatten(order='F'))])        # log(0) = -inf -> exp = 0
    d = np.array([0.5, 0.5])
    r1 = nagp.ihgp_ep_modulator_nmf(w, t, pr['y'], SSHandle(), mom2, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    o1 = oih.ihgp_ep_modulator_nmf(w, t, pr['y'], None, omom2, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(r1[0], o1[0]) < TOL_MEAN and rel(r1[5]['ttau'], o1[5]['ttau']) < TOL_SITE
    r1g = nagp.gf_ep_modulator_nmf(w, t, pr['y'], SSHandle(), mom2, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    o1g = ogf.gf_ep_modulator_nmf(w, t, pr['y'], None, omom2, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(r1g[0], o1g[0]) < TOL_MEAN and rel(r1g[5]['ttau'], o1g[5]['ttau']) < TOL_SITE
    # ... and one whose zero pattern is not two contiguous blocks (per-point path)
    W2 = pr['W'].copy(); W2[::2, 2:] = 0.0; W2[1::2, :2] = 0.0
    with np.errstate(divide='ignore'):
        w2 = np.concatenate([pr['w'][:1 + 3 * D + 2 * N], np.log(W2.flatten(order='F'))])
    r2 = nagp.ihgp_ep_modulator_nmf(w2, t, pr['y'], SSHandle(), mom2, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    o2 = oih.ihgp_ep_modulator_nmf(w2, t, pr['y'], None, omom2, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(r2[0], o2[0]) < TOL_MEAN and rel(r2[5]['ttau'], o2[5]['ttau']) < TOL_SITE


@pytest.mark.parametrize('D,N,T', [(5, 2, 300), (24, 3, 64), (32, 6, 40)])
def test_ekf_training_objective_against_oracle(D, N, T):
    """[e, eg] = gf_giekf_modulator_nmf_constraints(w,x,y,ss,mom,[],...,GradObj='off') as train_GTFNMF.m:199 calls it
    (row a11 as far as the reference runs): one plain EKF pass with prediction at the first step, stationary
    Q = Pinf - A Pinf A', energy sum; NaN observations are not skipped in this loop (:385-472)."""
    pr = harness.nmf_problem(D, N, T, 17 + D, 'constraints'); t = np.arange(1, T + 1.0)
    cons = harness.CONSTRAINTS_DEMO(D); w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
    e, eg = nagp.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2,
                                                    cons, wf, harness.TUNE_DEMO, 'off')
    eo, ego = oek.gf_giekf_modulator_nmf_constraints_nlml(w, t, pr['y'], 'matern32', 'matern52', 1, D, N, cons, wf, harness.TUNE_DEMO)
    assert abs(e - eo) < TOL_LOGZ * abs(eo) and eg.shape == ego.shape and not np.any(eg)
    if D == 5:
        y = pr['y'].copy(); y[20] = np.nan
        e2, _ = nagp.gf_giekf_modulator_nmf_constraints(w, t, y, SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2,
                                                        cons, wf, harness.TUNE_DEMO, 'off')
        assert np.isnan(e2) and np.isnan(oek.gf_giekf_modulator_nmf_constraints_nlml(w, t, y, 'matern32', 'matern52', 1, D, N, cons, wf, harness.TUNE_DEMO)[0])
        # GradObj='on' with the demo's tune_hypers: numel(w) = D+2N < size(dF,3) -- the reference statement stops with an index error
        with pytest.raises(IndexError, match='gdata'):
            nagp.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2,
                                                    cons, wf, harness.TUNE_DEMO, 'on')


def test_batched_ekf_objective_equals_serial_calls():
    """nlml_batch(..., inference='EKF'): the replicas of one fminunc iteration of the 'EKF' case (train_GTFNMF.m:198-201)."""
    D, N, T = 6, 2, 200
    pr = harness.nmf_problem(D, N, T, 23, 'constraints'); t = np.arange(1, T + 1.0)
    cons = harness.CONSTRAINTS_DEMO(D); w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
    ws = [w, w + 0.01, w - 0.02]
    f = nagp.nlml_batch(ws, t, pr['y'], SSHandle(), None, 'matern32', 'matern52', 1, D, N, 0.5, None, 1,
                        constraints=cons, w_fixed=wf, tune_hypers=harness.TUNE_DEMO, inference='EKF')
    for wi, fi in zip(ws, f):
        e, _ = nagp.gf_giekf_modulator_nmf_constraints(wi, t, pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2,
                                                       cons, wf, harness.TUNE_DEMO, 'off')
        assert fi == e
    eo = oek.gf_giekf_modulator_nmf_constraints_nlml(ws[1], t, pr['y'], 'matern32', 'matern52', 1, D, N, cons, wf, harness.TUNE_DEMO)[0]
    assert abs(f[1] - eo) < TOL_LOGZ * abs(eo)


def _mixture_w(g, J):
    return [g['lik'], [g['p1_%d' % j] for j in range(J)], [g['p2_%d' % j] for j in range(J)], [g['W_%d' % j] for j in range(J)]]


def test_golden_widened_rows_mixtures_and_ekf_objective():
    """Committed vectors of the widened rows (tools/make_golden.py widened)."""
    g = gold('mixture_ihgp_3x4x3'); T = g['y'].size; t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorPreCalcwn', link_shift=1.0, wn=g['wn'], xn_unscaled=g['xn_unscaled'])
    r = nagp.ihgp_ep_mods_nmf_mixture(_mixture_w(g, 3), t, g['y'], SSHandle(), mom, t, ['exp'] * 3, ['matern52'] * 3, 3, 0.75, 0.025, 4, nargout=6)
    assert rel(r[0], g['Eft']) < TOL_MEAN and rel(r[1], g['Varft']) < TOL_MEAN
    assert rel(r[5]['ttau'], g['ttau']) < TOL_SITE and rel(r[5]['tnu'], g['tnu']) < TOL_SITE
    fin = np.isfinite(g['R']); assert np.array_equal(fin, np.isfinite(r[5]['R'])) and rel(r[5]['R'][fin], g['R'][fin]) < 1e-6
    g = gold('mixture_gf_2src'); T = g['y'].size; t = np.arange(1, T + 1.0)
    r = nagp.gf_ep_mods_nmf_mixture(_mixture_w(g, 2), t, g['y'], SSHandle(), Mom('likModulatorNMFPower', p_cubature=7), t,
                                    ['exp', 'matern32'], ['matern52', 'matern52'], 2, 0.75, 0.2, 4, nargout=6)
    assert rel(r[0], g['Eft']) < TOL_MEAN and rel(r[1], g['Varft']) < TOL_MEAN
    assert rel(r[5]['ttau'], g['ttau']) < TOL_SITE and rel(r[5]['tnu'], g['tnu']) < TOL_SITE and rel(r[5]['lZ'], g['lZ']) < 1e-7
    g = gold('ekf_objective_cfg4_shape'); D, N = int(g['D']), int(g['N']); T = g['y'].size; t = np.arange(1, T + 1.0)
    e, eg = nagp.gf_giekf_modulator_nmf_constraints(g['w'], t, g['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 1,
                                                    g['constraints'], g['w_fixed'], list(g['tune_hypers']), 'off')
    assert abs(e - float(g['edata'])) < TOL_LOGZ * abs(float(g['edata'])) and not np.any(eg)


def test_full_length_ihgp_and_mixture_prefix_properties():
    """BASELINE sizes of the infinite-horizon path through size-independent properties: the sweep-1 (ADF) sites of step k depend only on
    y(1..k), so with one sweep the leading columns of a full-length run must equal a short run bit for bit, and the short run
    is compared with the oracle.  cfg3: 32 channels / 6 components / T = 200 000 (sequential ADF kernel + parallel-in-time
    scans); source separation: 3 x (16 channels, 3 components), T = 96 000, 3 973 sigma points (block-structured cubature)."""
    D, N, T, Ts = 32, 6, 200000, 400
    pr = harness.nmf_problem(D, N, T, 300); t = np.arange(1, T + 1.0); ts = t[:Ts]
    mom = Mom('likModulatorNMFPower', p_cubature=7)
    a = nagp.ihgp_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, [0.5], 1, nargout=6)
    b = nagp.ihgp_ep_modulator_nmf(pr['w'], ts, pr['y'][:Ts], SSHandle(), mom, ts, 'matern32', 'matern52', 1, D, N, 0.5, [0.5], 1, nargout=6)
    assert np.array_equal(a[5]['ttau'][:, :Ts - 1], b[5]['ttau'][:, :Ts - 1]) and np.array_equal(a[5]['tnu'][:, :Ts - 1], b[5]['tnu'][:, :Ts - 1])
    o = oih.ihgp_ep_modulator_nmf(pr['w'], ts[:150], pr['y'][:150], None, olik.Mom(olik.LIK_POWER_NMF, p=7), ts[:150], 'matern32', 'matern52', 1, D, N, 0.5, [0.5], 1)
    assert rel(b[5]['ttau'][:, :149], o[5]['ttau'][:, :149]) < TOL_SITE
    assert np.all(np.isfinite(a[0])) and np.all(np.isfinite(a[5]['ttau'])) and np.isfinite(a[5]['nlZ'][0])
    shapes = [(16, 3)] * 3; k1 = ['exp'] * 3; k2 = ['matern52'] * 3; T = 96000; Ts = 64
    mp = harness.mixture_problem(shapes, T, 3, k1, k2); t = np.arange(1, T + 1.0); ts = t[:Ts]
    mom, omom = _mixture_moms('likModulatorPreCalcwn', 9, 9)
    a = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 3, 0.75, 0.025, 1, nargout=6)
    b = nagp.ihgp_ep_mods_nmf_mixture(mp['w'], ts, mp['y'][:Ts], SSHandle(), mom, ts, k1, k2, 3, 0.75, 0.025, 1, nargout=6)
    assert np.array_equal(a[5]['ttau'][:, :Ts - 1], b[5]['ttau'][:, :Ts - 1]) and np.array_equal(a[5]['tnu'][:, :Ts - 1], b[5]['tnu'][:, :Ts - 1])
    o = omx.ihgp_ep_mods_nmf_mixture(mp['w'], ts[:10], mp['y'][:10], None, omom, ts[:10], k1, k2, 3, 0.75, 0.025, 1)
    assert rel(b[5]['ttau'][:, :9], o[5]['ttau'][:, :9]) < TOL_SITE
    assert np.all(np.isfinite(a[0])) and np.all(np.isfinite(a[5]['ttau']))


# ---------------------------------------------------------------------------------------------
# round 2: multi-GPU entry point of the C ABI, warm start, the two cubature forms, ranks with real plans
def _small_batch(kind_name, B=3, D=4, N=2, T=90, seed=400):
    probs, ys = [], []
    for q in range(B):
        pr = harness.nmf_problem(D, N, T, seed + q, 'constraints' if kind_name == 'ihgp' else 'demo_nmf')
        blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
        if kind_name == 'ihgp':
            blk = pss.balance_blocks(blk)
        y = pr['y'].copy(); y[7 + q] = np.nan
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    return probs, ys


@pytest.mark.parametrize('kind_name', ['gf', 'ihgp'])
def test_batch_run_one_gpu_is_bit_equal_to_the_plan_and_reduces_nlz(kind_name, monkeypatch):
    """nagp_batch_run(n_gpus = 1): the same bits as nagp_plan_* for every problem; nlZ_total = the sum over problems, both
    without a collective and (NAGP_FORCE_RCCL) through ncclAllReduce on a one-device communicator."""
    kind = L.KIND_IHGP if kind_name == 'ihgp' else L.KIND_GF_EP
    probs, ys = _small_batch(kind_name); T = ys[0].size
    mom = Mom('likModulatorNMFPower', p_cubature=5); d = 0.5 * np.ones(3)
    plan = Plan(kind, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
    plan.upload(ys); plan.execute(); ref = plan.download(); plan.close()
    for force in (False, True):
        if force:
            monkeypatch.setenv('NAGP_FORCE_RCCL', '1')
        outs, tot = nagp.batch_run(kind, probs, ys, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, n_gpus=1)
        for a, b in zip(outs, ref):
            for f in ('Eft', 'Varft', 'ttau', 'tnu', 'lZ', 'nlZ', 'MS'):
                assert np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True), f
        assert np.array_equal(tot, np.sum([b.nlZ for b in ref], axis=0))     # same order of summation as the library
    nagp.lib().nagp_shutdown()
    with pytest.raises(nagp.NagpError):
        nagp.batch_run(kind, probs, ys, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, n_gpus=64)


@pytest.mark.parametrize('kind_name', ['gf', 'ihgp'])
def test_batch_run_threads_of_several_devices_on_one_card(kind_name, monkeypatch):
    """The multi-device form of nagp_batch_run with its devices mapped onto this one card (NAGP_TEST_FAKE_DEVICES=4): four host threads,
    four plans alive and executing at the same time, outputs scattered back by problem index -- every problem equals its place in
    the single plan bit for bit, nlZ_total is the sum in device order; an injected failure of one device comes back as that device's
    error, and the next call works."""
    kind = L.KIND_IHGP if kind_name == 'ihgp' else L.KIND_GF_EP
    probs, ys = _small_batch(kind_name); T = ys[0].size
    probs = probs + probs[:3]; ys = ys + [y[::-1].copy() for y in ys[:3]]
    mom = Mom('likModulatorNMFPower', p_cubature=5); d = 0.5 * np.ones(3)
    plan = Plan(kind, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
    plan.upload(ys); plan.execute(); ref = plan.download(); plan.close()
    monkeypatch.setenv('NAGP_TEST_FAKE_DEVICES', '4')
    outs, tot = nagp.batch_run(kind, probs, ys, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, n_gpus=4)
    for a, b in zip(outs, ref):
        for f in ('Eft', 'Varft', 'ttau', 'tnu', 'lZ', 'nlZ', 'MS'):
            assert np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True), f
    dev = nagp.batch_partition(len(probs), 4)
    exp = np.zeros(3)
    for g in range(4):
        part = np.zeros(3)
        for i in range(len(probs)):
            if dev[i] == g:
                part += ref[i].nlZ
        exp += part
    assert np.array_equal(tot, exp)
    monkeypatch.setenv('NAGP_TEST_FAIL_DEVICE', '2')
    with pytest.raises(nagp.NagpError, match='device 2: injected failure'):
        nagp.batch_run(kind, probs, ys, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, n_gpus=4)
    monkeypatch.delenv('NAGP_TEST_FAIL_DEVICE')
    outs, tot2 = nagp.batch_run(kind, probs, ys, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, n_gpus=4)
    assert np.array_equal(tot2, tot)


def test_warm_start_sites():
    """nagp_plan_upload_sites: zeros are the cold start (bit for bit); the sites of a finished run as the start of the next
    agree with the oracle started from the same sites; dropping them returns to the cold start."""
    D, N, T = 4, 2, 80
    pr = harness.nmf_problem(D, N, T, 77); d = 0.5 * np.ones(2)
    blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    mom = Mom('likModulatorNMFPower', p_cubature=5)
    plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=2)
    plan.upload([pr['y']]); plan.execute(); cold = plan.download()[0]
    plan.upload_sites([np.zeros((D + N, T))], [np.zeros((D + N, T))]); plan.execute(); z = plan.download()[0]
    for f in ('Eft', 'Varft', 'ttau', 'tnu', 'nlZ'):
        assert np.array_equal(getattr(z, f), getattr(cold, f)), f
    plan.upload_sites([cold.ttau], [cold.tnu]); plan.execute(); warm = plan.download()[0]
    assert not np.array_equal(warm.ttau, cold.ttau)
    lik, p1, p2, W = oss_unpack(pr, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', False)
    ref = ogf.run_predict(model, pr['y'], olik.Mom(olik.LIK_POWER_NMF, p=5), 0.5, d, 2, sites0=(cold.ttau, cold.tnu))
    assert rel(warm.Eft, ref['Eft']) < TOL_MEAN and rel(warm.ttau, ref['ttau']) < TOL_SITE and relz(warm.nlZ, ref['nlZ']) < TOL_LOGZ
    plan.upload_sites(None, None); plan.execute(); again = plan.download()[0]
    assert np.array_equal(again.Eft, cold.Eft)
    plan.close()


def oss_unpack(pr, D, N):
    from oracle import ss as oss
    return oss.unpack_log(pr['w'], 1, D, N)


@pytest.mark.parametrize('fn', ['gf', 'ihgp'])
@pytest.mark.parametrize('shape', [(5, 2, 5), (8, 3, 7), (6, 4, 9), (9, 2, 3), (7, 7, 5), (33, 2, 9)])
def test_sparse_point_cubature_equals_the_generic_form_and_the_oracle(fn, shape, monkeypatch):
    """likModulatorNMFPower in the staged sparse-point form (nagp_momsp.hpp: ADF launches of gf_ep_* and the IHGP ADF sweep)
    against the generic mom_eval (NAGP_NO_SPARSE=1) and the oracle, one ADF sweep so that nothing amplifies rounding:
    both device forms within 1e-10 of each other and TOL of the oracle; odd sub-band counts, 1..4 non-centre coordinates."""
    D, N, p = shape; T = 60
    pr = harness.nmf_problem(D, N, T, 900 + D, 'constraints'); t = np.arange(1, T + 1.0)
    y = pr['y'].copy(); y[20:23] = np.nan
    mom = Mom('likModulatorNMFPower', p_cubature=p); d = np.array([0.5])
    f = nagp.ihgp_ep_modulator_nmf if fn == 'ihgp' else nagp.gf_ep_modulator_nmf
    of = oih.ihgp_ep_modulator_nmf if fn == 'ihgp' else ogf.gf_ep_modulator_nmf
    res = {}
    modes = ('generic', 'sparse', 'sparse16', 'sparse4') if fn == 'ihgp' else ('generic', 'sparse')
    # sparse: IHGP runs the role-specialised 512-thread kernel, eight points per MFMA step where the rule allows it;
    # sparse16: the same with four points per step (NAGP_IH_PACK=0); sparse4: the four-wave kernel (NAGP_IH_ROLES=0)
    for mode in modes:
        for k_ in ('NAGP_NO_SPARSE', 'NAGP_IH_ROLES', 'NAGP_IH_PACK'):
            monkeypatch.delenv(k_, raising=False)
        if mode == 'generic': monkeypatch.setenv('NAGP_NO_SPARSE', '1')
        if mode == 'sparse4': monkeypatch.setenv('NAGP_IH_ROLES', '0')
        if mode == 'sparse16': monkeypatch.setenv('NAGP_IH_PACK', '0')
        res[mode] = f(pr['w'], t, y, SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 1, nargout=6)
    for k_ in ('NAGP_NO_SPARSE', 'NAGP_IH_ROLES', 'NAGP_IH_PACK'):
        monkeypatch.delenv(k_, raising=False)
    ref = of(pr['w'], t, y, None, olik.Mom(olik.LIK_POWER_NMF, p=p), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 1)
    if fn == 'ihgp':
        for other in ('sparse4', 'sparse16'):
            a, b = res[other], res['sparse']
            assert rel(a[0], b[0]) < 1e-10 and rel(a[5]['ttau'], b[5]['ttau']) < 1e-9 and relz(a[5]['nlZ'], b[5]['nlZ']) < 1e-12, other
    for mode in modes:
        Eft, Varft, out = res[mode][0], res[mode][1], res[mode][5]
        assert rel(Eft, ref[0]) < TOL_MEAN and relz(out['nlZ'], ref[5]['nlZ']) < TOL_LOGZ, mode
        assert rel(out['ttau'], ref[5]['ttau']) < TOL_SITE, mode
    a, b = res['generic'], res['sparse']
    assert rel(a[0], b[0]) < 1e-10 and rel(a[5]['ttau'], b[5]['ttau']) < 1e-9 and relz(a[5]['nlZ'], b[5]['nlZ']) < 1e-12
    assert np.array_equal(np.isnan(a[5]['tnu']), np.isnan(b[5]['tnu']))


@pytest.mark.parametrize('fn', ['ihgp', 'gf'])
@pytest.mark.parametrize('shape', [(32, 6, 7, 'softplus', 1.0), (16, 3, 9, 'softplus', 0.0), (5, 2, 5, 'exp', 0.0), (21, 1, 9, 'softplus', 1.0),
                                   (17, 4, 7, 'softplus', 0.5), (9, 5, 7, 'softplus', 0.0)])
def test_sqrt_amplitude_likelihood_staged_form_equals_the_generic_form_and_the_oracle(fn, shape, monkeypatch):
    """likModulatorPreCalcwn (experiments/likModulatorPreCalcwn.m:28-86, the likelihood of train_model.m:55 / noise_reduction_speech.m:41)
    in the staged form of nagp_momsq.hpp -- square roots on a (sigma point, sub-band) lane grid, amplitudes kept in registers between the
    weights and the sub-band sums, modulator sums from marginal sums -- against the generic mom_eval (NAGP_NO_SPARSE=1) and the oracle:
    two sweeps (ADF sweep, smoother, site refresh, second filter pass), missing observations, 1 .. 6 components, one and two sub-bands
    per lane (D <= 16 / > 16), odd sub-band counts, ut3 / 5 / 7 / 9, both links."""
    D, N, p, link, shift = shape; T = 50
    from nagp import cubature
    pr = harness.nmf_problem(D, N, T, 1300 + D, 'constraints'); t = np.arange(1, T + 1.0)
    y = pr['y'].copy(); y[17:19] = np.nan
    wn, xn = cubature.utp_ws(p, N)
    mom = Mom('likModulatorPreCalcwn', link=link, link_shift=shift, wn=wn, xn_unscaled=xn); d = np.array([0.5, 0.4])
    omom = olik.Mom(olik.LIK_POWER_NMF_SQRT, link=(olik.exp_link() if link == 'exp' else olik.softplus_link(shift)), wn=wn, xn_unscaled=xn)
    f = nagp.ihgp_ep_modulator_nmf if fn == 'ihgp' else nagp.gf_ep_modulator_nmf
    of = oih.ihgp_ep_modulator_nmf if fn == 'ihgp' else ogf.gf_ep_modulator_nmf
    res = {}
    for mode in ('generic', 'staged'):
        monkeypatch.delenv('NAGP_NO_SPARSE', raising=False)
        if mode == 'generic': monkeypatch.setenv('NAGP_NO_SPARSE', '1')
        res[mode] = f(pr['w'], t, y, SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2, nargout=6)
    monkeypatch.delenv('NAGP_NO_SPARSE', raising=False)
    ref = of(pr['w'], t, y, None, omom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    for mode in ('generic', 'staged'):
        Eft, Varft, out = res[mode][0], res[mode][1], res[mode][5]
        assert rel(Eft, ref[0]) < TOL_MEAN and rel(Varft, ref[1]) < TOL_MEAN and relz(out['nlZ'], ref[5]['nlZ']) < TOL_LOGZ, mode
        assert rel(out['ttau'], ref[5]['ttau']) < TOL_SITE and rel(out['tnu'], ref[5]['tnu']) < TOL_SITE, mode
    a, b = res['generic'], res['staged']
    assert rel(a[0], b[0]) < 1e-9 and rel(a[5]['ttau'], b[5]['ttau']) < 1e-8 and relz(a[5]['nlZ'], b[5]['nlZ']) < 1e-11
    assert np.array_equal(np.isnan(a[5]['tnu']), np.isnan(b[5]['tnu']))


@pytest.mark.parametrize('mode', ['predict', 'nlml', 'mixture'])
@pytest.mark.parametrize('shape', [(3, 2, 5, 'matern32'), (16, 3, 5, 'matern32'), (24, 4, 5, 'exp'), (32, 6, 7, 'matern32'), (36, 4, 5, 'matern32'),
                                   (20, 7, 3, 'matern32')])
def test_role_specialised_gf_adf_sweep_equals_the_256_thread_form(shape, mode, monkeypatch):
    """gf_adf8_kernel (nagp_gfadf8.hpp: 512 threads, the cubature in the role layout of the IHGP sweep, the covariance tiles on the worker
    waves) against gf_filter_kernel<TPT, 0, V, 256, 1> (NAGP_NO_GF_ROLES=1), which the oracle tests pin: 5 / 19 / 28 / 38 / 40 / 27 sites =
    one tile per thread, two tiles on the six worker waves (38 sites = 741 tiles), two tiles on all eight waves (40 sites = 820 tiles);
    packed and unpacked MFMA steps (ut3 / ut5 / ut7, seven components), 2-state sub-band blocks, missing observations, two problems per
    plan, three sweeps; predict mode, the nlml mode (legacy update: one more barrier per step) and the mixture rule (mom at power
    ep_fraction, raw R).  Everything the sweeps return, to rounding; clamp counters equal."""
    D, N, p, k1 = shape; T = 44
    probs, ys = [], []
    for q in range(2):
        pr = harness.nmf_problem(D, N, T, 7100 + 10 * D + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, 'matern52'))
        y = pr['y'].copy(); y[11 + q] = np.nan; y[T - 1] = np.nan if q else y[T - 1]
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    mom = Mom('likModulatorNMFPower', p_cubature=p)
    kw = dict(mom=mom, ep_fraction=0.5 if mode != 'mixture' else 0.75, ep_damping=np.array([0.6, 0.5, 0.4]), ep_itts=3,
              mode=L.MODE_NLML if mode == 'nlml' else L.MODE_PREDICT, flags=L.FLAG_MIXTURE_RULE if mode == 'mixture' else 0)
    res = {}
    for form in ('roles', 'flat'):
        monkeypatch.delenv('NAGP_NO_GF_ROLES', raising=False)
        if form == 'flat': monkeypatch.setenv('NAGP_NO_GF_ROLES', '1')
        plan = Plan(L.KIND_GF_EP, probs, T, **kw); plan.upload(ys); plan.execute()
        res[form] = plan.download(want_MF=True); tm = plan.timings(); plan.close()
        assert tm['launches']['filter'] >= 1
    monkeypatch.delenv('NAGP_NO_GF_ROLES', raising=False)
    for q in range(2):
        a, v = res['roles'][q], res['flat'][q]
        fields = (('nlZ', 1e-10),) if mode == 'nlml' else (('Eft', 1e-9), ('Varft', 1e-9), ('MS', 1e-9), ('MF', 1e-9), ('lZ', 1e-9), ('ttau', 1e-7), ('tnu', 1e-7))
        for f, tol in fields:
            x, z = getattr(a, f), getattr(v, f)
            if f in ('ttau', 'tnu'):
                # sites beyond 1e8 are rounding noise of the algorithm itself (1 + d2 HPH ~ 0; tools/fuzz_conditioning.py: SITE_MAX) -- the
                # mixture rule does not clamp them away; they must be wild in BOTH forms, the rest must agree
                wild = (np.abs(res['flat'][q].ttau) > 1e8) | (np.abs(res['roles'][q].ttau) > 1e8)
                assert wild.sum() <= 4 and np.array_equal(np.abs(res['flat'][q].ttau) > 1e6, np.abs(res['roles'][q].ttau) > 1e6), (q, f)
                x, z = np.where(wild, 0.0, x), np.where(wild, 0.0, z)
            assert (relz(x[:1], z[:1]) if f == 'nlZ' else rel(x, z)) < tol, (q, f)      # (nlml mode: one energy per call)
        assert np.array_equal(a.counters, v.counters)
        assert np.all(np.isfinite(a.nlZ))


@pytest.mark.parametrize('D,N', [(5, 2), (32, 6)])
@pytest.mark.parametrize('T,nanpos', [(1, []), (2, [1]), (2, []), (3, [0]), (5, [4]), (17, [16]), (33, list(range(33)))])
def test_gf_sweeps_at_edge_lengths_against_the_oracle(T, nanpos, D, N):
    """gf_ep_modulator_nmf, three sweeps, at T = 1, 2, 3, 5, 17 with a NaN at the first / last step and with everything missing: the
    role-specialised ADF launch needs two steps (T = 1 and the single ADF step k = T-1 of later sweeps take the 256-thread form), the
    fixed-site launches of sweeps >= 2 cover k < T-1 (none at T = 1), the smoother has no step at T = 1 -- 7 and 38 sites."""
    pr = harness.nmf_problem(D, N, T, 900 + T + D, 'constraints'); t = np.arange(1, T + 1.0)
    y = pr['y'].copy(); y[nanpos] = np.nan
    mom = Mom('likModulatorNMFPower', p_cubature=5); om = olik.Mom(olik.LIK_POWER_NMF, p=5); d = np.array([0.6, 0.5, 0.4])
    r = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, om, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3)
    assert rel(r[0], o[0]) < 1e-10 and rel(r[1], o[1]) < 1e-10 and rel(r[5]['ttau'], o[5]['ttau']) < 1e-9 and rel(r[5]['nlZ'], o[5]['nlZ']) < 1e-10


@pytest.mark.parametrize('T,nanpos', [(1, []), (2, [1]), (3, [0]), (17, [16]), (33, list(range(33)))])
def test_sqrt_amplitude_ihgp_sweep_edge_lengths_and_batches(T, nanpos):
    """ihgp_adf8sq_kernel at T = 1, 2, 3 (the launches of sweeps >= 2 start at k = T - 1 from the filtered mean of the step before), a NaN at
    the first / last step, everything missing; and three problems of different data and hyper-parameters in ONE plan (one workgroup each)
    against the oracle.  Three sweeps."""
    from nagp import cubature
    D, N, p = 7, 3, 7
    wn, xn = cubature.utp_ws(p, N)
    mom = Mom('likModulatorPreCalcwn', link='softplus', link_shift=1.0, wn=wn, xn_unscaled=xn)
    omom = olik.Mom(olik.LIK_POWER_NMF_SQRT, link=olik.softplus_link(1.0), wn=wn, xn_unscaled=xn)
    d = 0.2 * np.ones(3); t = np.arange(1, T + 1.0)
    probs, ys, prs = [], [], []
    for q in range(3):
        pr = harness.nmf_problem(D, N, T, 1500 + q, 'constraints', link_shift=1.0, sqrt_amp=True)
        y = pr['y'].copy(); y[nanpos] = np.nan
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y); prs.append(pr)
    plan = Plan(L.KIND_IHGP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
    plan.upload(ys); plan.execute(); outs = plan.download(); plan.close()
    for q in range(3):
        ref = oih.ihgp_ep_modulator_nmf(prs[q]['w'], t, ys[q], None, omom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3)
        o = outs[q]
        assert rel(o.Eft, ref[0]) < TOL_MEAN and rel(o.Varft, ref[1]) < TOL_MEAN, q
        assert np.allclose(o.nlZ, ref[5]['nlZ'], rtol=TOL_LOGZ, atol=1e-12), q
        assert rel(o.ttau, ref[5]['ttau']) < TOL_SITE and np.array_equal(np.isnan(o.tnu), np.isnan(ref[5]['tnu'])), q


_RANK_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import dist as nd, harness, Mom, Plan, _lib as L, ss as pss
rank, lr, world = nd.init('gloo')          # two ranks share the one card of this box; RCCL needs one device per rank
NSEG, I, D, N, T = 5, 3, 4, 2, 70
def problem(q):
    pr = harness.nmf_problem(D, N, T, 800 + q)
    return (pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'), pr['W'], np.log(pr['w_lik'])), pr['y']
def run(idx):
    ps = [problem(q) for q in idx]
    plan = Plan(L.KIND_GF_EP, [p[0] for p in ps], T, mom=Mom('likModulatorNMFPower', p_cubature=5), ep_fraction=0.5, ep_damping=0.5 * np.ones(I), ep_itts=I)
    plan.upload([p[1] for p in ps]); plan.execute(); nlz = plan.download_nlz(); plan.close()
    return nlz
mine = nd.shard(NSEG, rank, world)
tot = nd.allreduce_nlz(run(mine))
serial = run(list(range(NSEG))).sum(axis=0)
assert np.allclose(tot, serial, rtol=1e-14), (tot, serial)
print('rank', rank, 'ok', tot, flush=True)
nd.finalize()
"""


def test_two_ranks_with_real_plans_allreduce_nlz(tmp_path):
    """The N > 1 product path end to end on one card: every rank builds a Plan for its shard of the segments, executes it on
    the GPU and the per-sweep nlZ is all-reduced; equal to one rank running all segments."""
    import subprocess, sys
    from nagp import dist as nd
    script = tmp_path / 'rank_worker.py'
    script.write_text(_RANK_WORKER)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, **nd.file_rendezvous_env(str(tmp_path), 2))      # file store: no port to race for
    procs = [subprocess.Popen([sys.executable, str(script), root], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=300)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_bench_two_ranks_rehearsal_on_one_card_prints_a_consistent_line():
    """`bench.py --gpus 2` end to end as the driver would launch it, except that both ranks share the one card of this box and the
    collective runs over gloo (NAGP_BENCH_REHEARSAL): the launcher spawns two fresh ranks, each builds its plan for its shard, the nlZ sums
    are all-reduced, rank 0 prints ONE line -- n_gpus, world_size, backend and the per-rank device list agree, the strong-scaling extra
    (segments IN TOTAL split over the ranks) is there, every nlZ is finite (bench.py refuses a line whose all-reduced nlZ is not the sum
    of the gathered per-rank sums)."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, NAGP_BENCH_REHEARSAL='1'); env.pop('RANK', None); env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--T', '400', '--steps', '1', '--warmup', '1', '--no-cpu-baseline', '--extras', 'cfg5'],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['n_gpus'] == 2 and d['world_size'] == 2 and d['collective_backend'] == 'gloo' and len(d['devices']) == 2
    assert d['launch'].startswith('self-spawned') and d['scaling'] == 'weak' and d['cfg5_strong']['scaling'] == 'strong'
    assert '8 segment(s) in total = 8/2 per GPU' in d['cfg5_strong']['workload']
    assert np.all(np.isfinite(d['nlZ_allreduced'])) and np.all(np.isfinite(d['cfg5_strong']['nlZ_allreduced'])) and d['value'] > 0


# ---------------------------------------------------------------------------------------------
# the C ABI from plain C, and the MEX gateway of matlab/ against a mock of the MEX API
def _dump_fixture(dirpath, family):
    """Raw little-endian dumps (tests/c/dump.h) of one golden fixture: the arrays a MATLAB wrapper hands to nagp_mex and the
    fixture's expected outputs."""
    from nagp import api as napi, ihgp_tables
    arrs = {}
    if family == 'gf':
        g = gold('cfg2_gf_ep_modulator_nmf'); D, N = int(g['D']), int(g['N'])
        lik, p1, p2, W = napi._unpack_log(g['w'], 1, D, N)
        blk = napi._blocks_from_dense(*SSHandle()(None, p1, p2, 'matern32', 'matern52'), D, N)
        prob = napi._Problem(blk, W, lik); kind, I, l_iter, flags = 0, 3, 0, 0
        mom = Mom('likModulatorNMFPower', p_cubature=9); exp = dict(Eft=g['Eft'], Varft=g['Varft'], nlZ=g['nlZ'], ttau=g['ttau'], tnu=g['tnu'])
    elif family == 'ihgp':
        g = gold('cfg3_ihgp_ep_modulator_nmf'); D, N = int(g['D']), int(g['N'])
        lik, p1, p2, W = napi._unpack_log(g['w'], 1, D, N)
        blk = pss.balance_blocks(napi._blocks_from_dense(*SSHandle()(None, p1, p2, 'matern32', 'matern52'), D, N))
        prob = napi._Problem(blk, W, lik, symmetrize_Q=True); kind, I, l_iter, flags = 1, 3, 0, 0
        r, PP, ppo, PG, pgo = ihgp_tables.build_tables(prob.A, prob.Q, blk.offsets, blk.h_val)
        arrs.update(r=r, PP=PP, PG=PG, pp_off=np.asarray(ppo, np.int64), pg_off=np.asarray(pgo, np.int64))
        mom = Mom('likModulatorNMFPower', p_cubature=7); exp = dict(Eft=g['Eft'], Varft=g['Varft'], nlZ=g['nlZ'], ttau=g['ttau'], tnu=g['tnu'])
    else:
        g = gold('cfg4_gf_giekf_modulator_nmf'); D, N = int(g['D']), int(g['N'])
        lik, p1, p2, W = napi._unpack_log(g['w_log'], 1, D, N)
        blk = pss.balance_blocks(napi._blocks_from_dense(*SSHandle()(None, p1, p2, 'matern32', 'matern52'), D, N))
        prob = napi._Problem(blk, W, lik); kind, I, l_iter, flags = 2, 2, 2, 0
        mom = None; exp = dict(Eft=g['Eft_plain'], Varft=g['Varft_plain'])
    arrs.update(A=prob.A, Q=prob.Q, Pinf=prob.Pinf, block_offsets=prob.offsets.astype(np.int32), h_val=prob.h_val, Wnmf=prob.W, y=g['y'],
                S=[blk.S], M=[blk.M], D=[D], N=[N], lik_param=[float(np.ravel(lik)[0])], kind=[kind], lik_kind=[1], ep_fraction=[0.5], ep_itts=[I],
                l_iter=[l_iter], flags=[flags])
    if mom is not None:
        wn, xn = mom.tables(N)
        arrs.update(wn=wn, xn_unscaled=xn, ep_damping=0.5 * np.ones(I))
    for k, v in exp.items():
        arrs['exp_' + k] = v
    with open(os.path.join(dirpath, 'meta.txt'), 'w') as meta:
        for k, v in arrs.items():
            a = np.asarray(v)
            a = np.asfortranarray(a.astype({'block_offsets': np.int32, 'pp_off': np.int64, 'pg_off': np.int64}.get(k, np.float64)))
            a.ravel(order='F').tofile(os.path.join(dirpath, k + '.bin'))
            meta.write('%s %d\n' % (k, a.size))


def _cc(out, srcs, extra=()):
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, 'nonstationary-audio-gp_amd')
    cmd = ['gcc', '-O1', '-std=c99', '-I', os.path.join(root, 'include'), '-I', os.path.join(root, 'tests', 'c')] + list(extra) + ['-o', out] + srcs + \
          ['-L', pkg, '-lnagp', '-lm', '-Wl,-rpath,' + pkg, '-Wl,-rpath,/opt/rocm/lib', '-Wl,-rpath-link,/opt/rocm/lib']
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return out


@pytest.mark.parametrize('family', ['gf', 'ihgp', 'giekf'])
def test_c_abi_from_plain_c(family, tmp_path):
    """tests/c/abi_golden.c: a C program (gcc, no Python, no torch in the process) links libnagp.so, feeds a golden fixture
    through nagp_ep_run / nagp_ihgp_run / nagp_giekf_run and compares with the expected outputs."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _dump_fixture(str(tmp_path), family)
    exe = _cc(str(tmp_path / 'abi_golden'), [os.path.join(root, 'tests', 'c', 'abi_golden.c')])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert 'invalid call -> -1' in r.stdout


@pytest.mark.parametrize('family', ['gf', 'ihgp', 'giekf'])
def test_mex_gateway_against_golden_fixture(family, tmp_path):
    """matlab/nagp_mex.c compiled against the mock MEX API of tests/c (no MATLAB here) and driven with the structs the .m
    wrappers build: same outputs as the fixture."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    _dump_fixture(str(tmp_path), family)
    c = os.path.join(root, 'tests', 'c')
    exe = _cc(str(tmp_path / 'mex_driver'), [os.path.join(c, 'mex_driver.c'), os.path.join(c, 'mex_mock.c'), os.path.join(root, 'matlab', 'nagp_mex.c')])
    r = subprocess.run([exe, str(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.parametrize('link', ['softplus', 'exp'])
def test_posterior_reconstruction_of_signal_and_amplitudes(link):
    """Row f-4 (demo_toy_modulators_nmf.m:119-158): Esig, Vsig, Eft_mod, Varft_mod from the marginals of a real run.
    Sampling form = the .m statement by statement on the library's reproducible draws (oracle/recon.py) to 1e-10; the moments
    form = the population values, which 40 000 draws estimate to Monte-Carlo accuracy."""
    from oracle import recon as orc
    D, N, T = 6, 3, 200
    pr = harness.nmf_problem(D, N, T, 55); t = np.arange(1, T + 1.0); d = 0.5 * np.ones(2)
    shift = 0.0
    Eft, Varft = nagp.gf_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), Mom('likModulatorNMFPower', link=link, p_cubature=5), t,
                                          'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    if link == 'exp':
        Eft = Eft.copy(); Eft[D:] *= 0.2; Varft = Varft.copy(); Varft[D:] = np.minimum(Varft[D:], 0.5)    # keep exp(g) tame
    lk = (lambda g: np.log(1.0 + np.exp(g - shift))) if link == 'softplus' else np.exp
    W = pr['W']
    got = nagp.reconstruct_signal(Eft, Varft, W, link=link, n_samples=250, seed=2019)
    ref = orc.sampling(Eft, Varft, W, lk, 250, 2019)
    for k in ('Esig', 'Vsig', 'Eft_mod', 'Varft_mod'):
        assert rel(got[k], ref[k]) < 1e-10, k
    from nagp.cubature import gauher
    gx, gw = gauher(32)
    mom_ = nagp.reconstruct_signal(Eft, Varft, W, link=link)
    refm = orc.moments(Eft, Varft, W, lk, gx, gw, exp_link=(link == 'exp'))
    for k in ('Esig', 'Vsig', 'Eft_mod', 'Varft_mod'):
        assert rel(mom_[k], refm[k]) < 1e-12, k
    big = orc.sampling(Eft, Varft, W, lk, 40000, 7)
    assert rel(mom_['Esig'], big['Esig']) < 0.03 and rel(mom_['Eft_mod'], big['Eft_mod']) < 0.03
    assert rel(mom_['Vsig'], big['Vsig']) < 0.1 and rel(mom_['Varft_mod'], big['Varft_mod']) < 0.1
    with pytest.raises(nagp.NagpError):
        nagp.reconstruct_signal(Eft, Varft, W, link=link, n_samples=1)


# ---------------------------------------------------------------------------------------------
# BASELINE sizes of configs[3] and configs[4] through size-independent properties; multi-segment plans at S = 146
def _constrained_problem(g):
    from nagp import api as napi
    D, N = int(g['D']), int(g['N'])
    lik, p1, p2, W = napi._unpack_constraints(g['w'], g['w_fixed'], list(g['tune_hypers']), g['constraints'], 1, D, N)
    blk = pss.balance_blocks(napi._blocks_from_dense(*SSHandle()(None, p1, p2, 'matern32', 'matern52'), D, N))
    return blk, W, lik


def test_full_length_cfg4_ekf_energy_prefix_property():
    """configs[3] at its native length (T = 88 200, 24 channels / 3 components, S = 105), EKF energy mode
    (gf_giekf_modulator_nmf_constraints.m:385-472): the energy of step k depends on y(1..k) only, so the per-step energies of the
    first 599 steps equal those of the 600-step run bit for bit, and the 600-step run is the golden one (compared with the oracle)."""
    g = gold('ekf_objective_cfg4_shape'); Tg = g['y'].size; T = 88200
    blk, W, lik = _constrained_problem(g)
    y = np.tile(g['y'], T // Tg + 1)[:T]
    def run(yy):
        plan = Plan(L.KIND_GIEKF, [(blk, W, lik)], yy.size, ep_itts=1, mode=L.MODE_NLML, l_iter=1)
        plan.upload([yy]); plan.execute(); o = plan.download(want_MS=False)[0]; plan.close()
        return o
    full, short = run(y), run(g['y'])
    assert blk.S == 105 and np.array_equal(full.lZ[:Tg - 1], short.lZ[:Tg - 1])
    assert abs(short.nlZ[0] - float(g['edata'])) < TOL_LOGZ * abs(float(g['edata']))
    assert abs(-np.sum(full.lZ[:Tg]) - float(g['edata'])) < 1e-9 * abs(float(g['edata']))
    assert np.all(np.isfinite(full.lZ)) and np.isfinite(full.nlZ[0])
    assert abs(full.nlZ[0] + np.sum(full.lZ)) < 1e-9 * abs(full.nlZ[0])


def test_full_length_cfg5_prefix_property_and_finiteness():
    """configs[4], one segment at full length (T = 100 000, 32 channels / 6 components, S = 146: 9.5 GB of filtered covariances,
    chunked smoother): with one sweep the sites of step k depend on y(1..k) only -- the leading 499 columns equal the 500-step
    run bit for bit, which in turn is compared with the oracle; everything finite, variances positive."""
    g = gold('cfg5_gf_ep_modulator_nmf_constraints'); Tg = g['y'].size; T = 100000
    blk, W, lik = _constrained_problem(g)
    y = np.tile(g['y'], T // Tg + 1)[:T]
    mom = Mom('likModulatorNMFPower', p_cubature=7)
    def run(yy):
        plan = Plan(L.KIND_GF_EP, [(blk, W, lik)], yy.size, mom=mom, ep_fraction=0.5, ep_damping=[0.5], ep_itts=1)
        plan.upload([yy]); plan.execute(); o = plan.download(want_MS=False)[0]; nb = plan.device_bytes(); plan.close()
        return o, nb
    (full, nbytes), (short, _) = run(y), run(g['y'])
    assert blk.S == 146 and nbytes > 9e9
    assert np.array_equal(full.ttau[:, :Tg - 1], short.ttau[:, :Tg - 1]) and np.array_equal(full.lZ[:Tg - 1], short.lZ[:Tg - 1])
    To = 120; to = np.arange(1, To + 1.0)
    o = ogf.gf_ep_modulator_nmf_constraints(g['w'], to, g['y'][:To], None, olik.Mom(olik.LIK_POWER_NMF, p=7), to, 'matern32', 'matern52', 1,
                                            int(g['D']), int(g['N']), 0.5, [0.5], 1, g['constraints'], g['w_fixed'], list(g['tune_hypers']))
    assert rel(short.ttau[:, :To - 1], o[5]['ttau'][:, :To - 1]) < TOL_SITE and rel(short.lZ[:To - 1], o[5]['lZ'][:To - 1]) < 1e-7
    assert np.all(np.isfinite(full.Eft)) and np.all(full.Varft > 0) and np.isfinite(full.nlZ[0]) and full.counters[0] == 0


@pytest.mark.parametrize('D,N,k1', [(22, 4, 'matern32'), (28, 4, 'matern32'), (30, 5, 'matern32'), (34, 6, 'matern32'), (30, 6, 'exp')])
def test_column_owner_mfma_smoother_against_valu_passes_and_oracle(D, N, k1):
    """Padded state dimensions 112 .. 160 (25 .. 40 sites): the MFMA smoother passes of nagp_mfma_big.hpp (7, 8, 9, 10 waves, one
    tile column each; full tiles at 32 and 40 sites, 2-state sub-band blocks in the last case) against the VALU passes on the
    same plan inputs and against the oracle; two problems per plan, chunks of 24 steps (several spans per chunk, the carry of the
    boundary state between chunks), a missing observation."""
    T = 70
    probs, ys = [], []
    for q in range(2):
        pr = harness.nmf_problem(D, N, T, 8100 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, 'matern52'))
        y = pr['y'].copy(); y[17 + q] = np.nan
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    assert 96 < 16 * ((4 * blk.M + 15) // 16) <= 160
    mom = Mom('likModulatorNMFPower', p_cubature=3); d = np.array([0.6, 0.5, 0.5])
    res = {}
    for mode in ('big', 'valu'):
        if mode == 'valu': os.environ['NAGP_NO_MFMA_BIG'] = '1'
        try:
            plan = Plan(L.KIND_GF_EP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, chunk=24)
            plan.upload(ys); plan.execute(); res[mode] = plan.download(); plan.close()
        finally:
            os.environ.pop('NAGP_NO_MFMA_BIG', None)
    for q in range(2):
        a, v = res['big'][q], res['valu'][q]
        for f, tol in (('Eft', 1e-8), ('Varft', 1e-8), ('MS', 1e-8), ('lZ', 1e-8), ('ttau', TOL_SITE), ('tnu', TOL_SITE)):
            assert rel(getattr(a, f), getattr(v, f)) < tol, (q, f)
        assert relz(a.nlZ, v.nlZ) < 1e-9 and abs(a.maxDiffP[-1] - v.maxDiffP[-1]) <= 1e-7 * max(1.0, abs(v.maxDiffP[-1]))
    pr = harness.nmf_problem(D, N, T, 8101, 'constraints')
    ref = ogf.run_predict(ogf.assemble(np.log(1e-4) * np.ones(1), pr['param1'], pr['param2'], pr['W'], k1, 'matern52', True), ys[1],
                          olik.Mom(olik.LIK_POWER_NMF, p=3), 0.5, d, 3)
    assert rel(res['big'][1].Eft, ref['Eft']) < TOL_MEAN and rel(res['big'][1].Varft, ref['Varft']) < TOL_MEAN
    assert relz(res['big'][1].nlZ, ref['nlZ']) < TOL_LOGZ


@pytest.mark.parametrize('T', [1, 2, 3, 9, 17])
def test_column_owner_passes_edge_lengths(T):
    """T = 1 (no smoothing step), 2, 3, and lengths around the chunk size (chunk = 8: one-step spans, a chunk of one step, the carry
    of the boundary state over three chunks) at 30 sites: the MFMA passes of nagp_mfma_big.hpp against the VALU passes."""
    D, N = 26, 4
    pr = harness.nmf_problem(D, N, max(T, 4), 8300 + T, 'constraints')
    y = pr['y'][:T].copy()
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    mom = Mom('likModulatorNMFPower', p_cubature=3); d = np.array([0.6, 0.5])
    res = {}
    for mode in ('big', 'valu'):
        if mode == 'valu': os.environ['NAGP_NO_MFMA_BIG'] = '1'
        try:
            plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))] * 2, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=2, chunk=8)
            plan.upload([y, y[::-1].copy()]); plan.execute(); res[mode] = plan.download(); plan.close()
        finally:
            os.environ.pop('NAGP_NO_MFMA_BIG', None)
    for q in range(2):
        for f in ('Eft', 'Varft', 'MS', 'nlZ'):
            assert rel(getattr(res['big'][q], f), getattr(res['valu'][q], f)) < 1e-10, (q, f)


def test_column_owner_passes_with_smoothed_covariances_requested_and_plan_reuse():
    """A plan that stores the smoothed covariances runs the last sweep through the VALU passes (tile-major chunk buffer) and the
    earlier ones through the MFMA passes (dense buffer with zero padding rows): executing it twice must give the same result
    (the buffer is zeroed when the layout switches back), equal to the all-VALU plan within rounding."""
    D, N, T = 26, 4, 50                     # 30 sites -> Sp = 128, 8 rows of padding
    pr = harness.nmf_problem(D, N, T, 8200, 'constraints')
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    assert 16 * ((4 * blk.M + 15) // 16) == 128 and 4 * blk.M < 128
    mom = Mom('likModulatorNMFPower', p_cubature=3); d = np.array([0.6, 0.5, 0.5])
    prob = [(blk, pr['W'], np.log(pr['w_lik']))]
    plan = Plan(L.KIND_GF_EP, prob, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, chunk=20, flags=L.FLAG_WANT_PS)
    plan.upload([pr['y']]); plan.execute(); first = plan.download(want_PS=True)[0]
    plan.execute(); second = plan.download(want_PS=True)[0]; plan.close()
    for f in ('Eft', 'Varft', 'MS', 'PS', 'ttau', 'lZ', 'nlZ'):
        assert np.array_equal(getattr(first, f), getattr(second, f), equal_nan=True), f
    os.environ['NAGP_NO_MFMA_BIG'] = '1'
    try:
        ref = Plan(L.KIND_GF_EP, prob, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, chunk=20, flags=L.FLAG_WANT_PS)
        ref.upload([pr['y']]); ref.execute(); v = ref.download(want_PS=True)[0]; ref.close()
    finally:
        os.environ.pop('NAGP_NO_MFMA_BIG', None)
    assert rel(first.PS, v.PS) < 1e-8 and rel(first.MS, v.MS) < 1e-8 and rel(first.Eft, v.Eft) < 1e-8 and relz(first.nlZ, v.nlZ) < 1e-9


def test_large_shape_fuzz_against_oracle():
    """Fixed-seed subset of tools/gpu_fuzz_large.py: random shapes with 22..40 sites (padded dimensions 96..160), kernels, cubature
    orders, sweep counts, chunk sizes and missing samples through both EP families -- the column-owner MFMA smoother passes, the
    768-thread gain kernel and the two/three-tiles-per-thread filters at the full tolerance."""
    import subprocess, sys
    tool = os.path.join(os.path.dirname(__file__), '..', 'tools', 'gpu_fuzz_large.py')
    r = subprocess.run([sys.executable, tool, '6', '5'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert lines[-1].startswith('worst'), r.stdout[-1000:]
    assert not any('<<<<' in ln for ln in lines), r.stdout[-2000:]
    worst = eval(lines[-1].split('worst', 1)[1].rsplit('}', 1)[0] + '}')
    assert worst['gf'] < TOL_MEAN and worst['ihgp'] < TOL_MEAN


def test_eight_segments_at_S146_equal_their_single_problem_plans():
    """configs[4] as the bench runs it (several 32-channel / 6-component segments in one plan, three tiles per thread, the VALU
    smoother passes with eight problems per launch): every segment of the 8-segment plan equals the plan of that segment alone."""
    D, N, T = 32, 6, 160
    probs, ys = [], []
    for q in range(8):
        pr = harness.nmf_problem(D, N, T, 5000 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        y = pr['y'].copy(); y[11 * q + 3] = np.nan
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    mom = Mom('likModulatorNMFPower', p_cubature=7); d = 0.5 * np.ones(3)
    plan = Plan(L.KIND_GF_EP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
    plan.upload(ys); plan.execute(); outs = plan.download(); plan.close()
    for q in (0, 3, 7):
        one = Plan(L.KIND_GF_EP, [probs[q]], T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
        one.upload([ys[q]]); one.execute(); o = one.download()[0]; one.close()
        for f in ('Eft', 'Varft', 'ttau', 'tnu', 'lZ', 'nlZ', 'MS', 'maxDiffP'):
            assert np.array_equal(getattr(outs[q], f), getattr(o, f), equal_nan=True), (q, f)
    ref = ogf.run_predict(ogf.assemble(np.log(1e-4) * np.ones(1), *[harness.nmf_problem(D, N, T, 5003, 'constraints')[k] for k in ('param1', 'param2', 'W')],
                                       'matern32', 'matern52', True), ys[3], olik.Mom(olik.LIK_POWER_NMF, p=7), 0.5, d, 3)
    assert rel(outs[3].Eft, ref['Eft']) < TOL_MEAN and relz(outs[3].nlZ, ref['nlZ']) < TOL_LOGZ


# ---------------------------------------------------------------------------------------------
# the Cholesky jitter-retry branch (SURVEY C-7; gf_ep_modulator_nmf.m:216-223) and its failure
def _indefinite_prior_problem(j, c, T=40):
    """A 3-channel / 2-component model whose Pinf(j,j) is lowered by c: state j is the unobserved last state of a modulator or
    sub-band block, the filter stays stable, and A*PS_k*A'+Q has a small negative eigenvalue -- the smoother's chol fails."""
    D, N = 3, 2
    pr = harness.nmf_problem(D, N, T, 11)
    blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    lik, p1, p2, W = ogf.ssm.unpack_log(pr['w'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', False)
    assert np.allclose(model['Pinf'], np.asarray(pss.discretise(blk)[2]), rtol=1e-12, atol=0)   # same state ordering on both sides
    P = np.array(model['Pinf']); P[j, j] -= c                 # one prior for both: the library takes Pinf as a plain array
    model = dict(model); model['Pinf'] = P.copy()
    return pr, (blk, pr['W'], np.log(pr['w_lik']), dict(Pinf=P)), model


@pytest.mark.parametrize('j,c,every_step', [(17, 1e-8, True), (14, 1e-7, False), (3, 1e-6, False)])
def test_cholesky_jitter_retry_branch_against_oracle(j, c, every_step):
    """PSkp not positive definite -> second attempt with sqrt(1e-4)*0.5 on the diagonal (0.5 stands for the reference's unseeded
    rand): the retry counter is positive and equals the oracle's, the outputs equal the oracle's jitter branch."""
    T = 40
    pr, prob, model = _indefinite_prior_problem(j, c, T)
    mom = Mom('likModulatorNMFPower', p_cubature=5); d = 0.5 * np.ones(2)
    ref = ogf.run_predict(model, pr['y'], olik.Mom(olik.LIK_POWER_NMF, p=5), 0.5, d, 2)
    n_ref = ref['counters'].get('chol_retries', 0)
    assert n_ref > 0 and (n_ref == 2 * (T - 1)) == every_step
    plan = Plan(L.KIND_GF_EP, [prob], T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=2, flags=L.FLAG_WANT_PS)
    plan.upload([pr['y']]); plan.execute(); o = plan.download(want_PS=True)[0]; plan.close()
    assert o.counters[0] == n_ref and o.counters[3] == 0, (o.counters, n_ref)
    assert rel(o.Eft, ref['Eft']) < TOL_MEAN and rel(o.Varft, ref['Varft']) < TOL_MEAN and relz(o.nlZ, ref['nlZ']) < TOL_LOGZ
    assert rel(o.PS, np.transpose(ref['PS'], (1, 2, 0))) < TOL_MEAN and rel(o.ttau, ref['ttau']) < TOL_SITE


def test_cholesky_failure_after_the_retry_is_reported_as_not_pd():
    """Both attempts fail (negative eigenvalue far beyond the jitter): the oracle's chol throws as MATLAB's does inside the catch
    block; the library finishes the sweeps, counts the steps and returns NAGP_ENOTPD from the execute call."""
    T = 40
    pr, prob, model = _indefinite_prior_problem(14, 1e-4, T)
    mom = Mom('likModulatorNMFPower', p_cubature=5); d = 0.5 * np.ones(2)
    with pytest.raises(np.linalg.LinAlgError):
        ogf.run_predict(model, pr['y'], olik.Mom(olik.LIK_POWER_NMF, p=5), 0.5, d, 2)
    plan = Plan(L.KIND_GF_EP, [prob], T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=2)
    plan.upload([pr['y']])
    with pytest.raises(nagp.NagpError, match='not positive definite'):
        plan.execute()
    o = plan.download()[0]; plan.close()                      # the plan stays usable: counters say what happened
    assert o.counters[3] > 0 and o.counters[0] >= o.counters[3]


def test_full_length_cfg4_predict_mode_filter_prefix_and_chunked_smoother():
    """configs[3] in PREDICT mode at its native length (T = 88 200, 24 channels / 3 components, S = 105: 4 GB of filtered
    covariances, chunked smoother).  One global iteration: the EKF filter is causal, so the filtered means MF of the first 799
    steps equal those of the 800-step run bit for bit, and the 800-step run is compared with the oracle (filtered AND smoothed);
    the smoothed output of the long run is finite with positive variances and does not depend on the chunking beyond rounding."""
    g = gold('cfg4_gf_giekf_modulator_nmf'); Tg = g['y'].size; T = 88200
    D, N = int(g['D']), int(g['N'])
    blk, W, lik = _constrained_problem(g)
    y = np.tile(g['y'], T // Tg + 1)[:T]
    def run(yy, chunk=0):
        plan = Plan(L.KIND_GIEKF, [(blk, W, lik)], yy.size, ep_itts=1, l_iter=1, flags=L.FLAG_EKF_RESET_P, chunk=chunk)
        plan.upload([yy]); plan.execute(); o = plan.download(want_MS=False, want_MF=True)[0]; nb = plan.device_bytes(); plan.close()
        return o, nb
    (full, nbytes), (short, _) = run(y), run(g['y'])
    assert blk.S == 105 and nbytes > 4e9
    assert np.array_equal(full.MF[:, :Tg - 1], short.MF[:, :Tg - 1])
    t = np.arange(1, Tg + 1.0)
    ref = oek.gf_giekf_modulator_nmf_constraints(g['w'], t, g['y'], None, None, t, 'matern32', 'matern52', 1, D, N, 1, 1,
                                                 g['constraints'], g['w_fixed'], list(g['tune_hypers']))[5]
    assert rel(short.MF, ref['MF']) < TOL_MEAN and rel(short.Eft, ref['Eft']) < TOL_MEAN and rel(short.Varft, ref['Varft']) < TOL_MEAN
    assert np.all(np.isfinite(full.Eft)) and np.all(full.Varft > 0) and full.counters[0] == 0 and full.counters[3] == 0
    other, _ = run(y, chunk=20000)                             # another chunking / span partition of the same backward recursion
    assert rel(other.Eft, full.Eft) < 1e-9 and rel(other.Varft, full.Varft) < 1e-9
    assert np.array_equal(other.MF, full.MF)


# ---------------------------------------------------------------------------------------------
# chunk-pipelined smoother: same kernels, same operands, another schedule -> bit-equal outputs
def _run_schedules(kind, probs, ys, T, chunk, **kw):
    """the same plan under the pipelined schedule (gain + compose of finished chunks on a second stream beside the filter), the
    pipelined schedule with only two (G, Delta) buffers (gains of the other chunks recomputed after the filter) and the serial one"""
    res = {}
    for name, env in (('pipelined', {}), ('two_buffers', {'NAGP_PIPELINE_SLOTS': '2'}), ('serial', {'NAGP_NO_PIPELINE': '1'})):
        os.environ.update(env)
        try:
            plan = Plan(kind, probs, T, chunk=chunk, **kw)
            plan.upload(ys); plan.execute(); res[name] = plan.download(want_PS=bool(kw.get('flags', 0) & L.FLAG_WANT_PS), want_MF=True); plan.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    return res


@pytest.mark.parametrize('D,N,want_ps', [(32, 6, False), (32, 6, True), (16, 3, False), (16, 3, True), (36, 8, False), (48, 9, False), (48, 9, True)])
def test_pipelined_smoother_is_bit_equal_to_the_serial_schedule(D, N, want_ps):
    """Eight (four) segments, several chunks per sweep, three sweeps: 32 channels / 6 components (S = 146, column-owner MFMA passes;
    with smoothed covariances requested the last sweep runs the VALU passes), 16 / 3 (S = 73, dense MFMA passes), 36 / 8 (44 sites: VALU
    passes only), 48 / 9 (57 sites: eight tiles per thread).  Every output of the pipelined schedules equals the serial schedule's bit for bit."""
    T = 150; B = 8 if D == 32 else 4
    probs, ys = [], []
    for q in range(B):
        pr = harness.nmf_problem(D, N, T, 6100 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        y = pr['y'].copy(); y[7 * q + 5] = np.nan
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    mom = Mom('likModulatorNMFPower', p_cubature=3 if D >= 36 else 5); d = 0.5 * np.ones(3)
    res = _run_schedules(L.KIND_GF_EP, probs, ys, T, 24, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, flags=L.FLAG_WANT_PS if want_ps else 0)
    for q in range(B):
        for other in ('pipelined', 'two_buffers'):
            for f in ('Eft', 'Varft', 'MS', 'MF', 'ttau', 'tnu', 'R', 'lZ', 'nlZ', 'maxDiffM', 'maxDiffP') + (('PS',) if want_ps else ()):
                assert np.array_equal(getattr(res[other][q], f), getattr(res['serial'][q], f), equal_nan=True), (other, q, f)
            assert np.array_equal(res[other][q].counters, res['serial'][q].counters)
    if D == 32 and not want_ps:       # and the serial schedule is the one the oracle pins
        ref = ogf.run_predict(ogf.assemble(np.log(1e-4) * np.ones(1), *[harness.nmf_problem(D, N, T, 6102, 'constraints')[k] for k in ('param1', 'param2', 'W')],
                                           'matern32', 'matern52', True), ys[2], olik.Mom(olik.LIK_POWER_NMF, p=5), 0.5, d, 3)
        assert rel(res['pipelined'][2].Eft, ref['Eft']) < TOL_MEAN and relz(res['pipelined'][2].nlZ, ref['nlZ']) < TOL_LOGZ


def test_pipelined_smoother_ekf_bit_equal_and_long_sequence_against_serial():
    """The EKF family under the three schedules (restart from the smoothed state between global iterations), and a sequence long
    enough for the default chunking to pipeline (T = 9000: five chunks): pipelined == serial bit for bit."""
    D, N, T = 12, 3, 130
    pr = harness.nmf_problem(D, N, T, 6200)
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    res = _run_schedules(L.KIND_GIEKF, [(blk, pr['W'], np.log(pr['w_lik']))] * 2, [pr['y'], pr['y'][::-1].copy()], T, 16, ep_itts=3, l_iter=2)
    for q in range(2):
        for other in ('pipelined', 'two_buffers'):
            for f in ('Eft', 'Varft', 'MS', 'MF', 'maxDiffP'):
                assert np.array_equal(getattr(res[other][q], f), getattr(res['serial'][q], f)), (other, q, f)
    D, N, T = 16, 3, 9000
    pr = harness.nmf_problem(D, N, T, 6300)
    blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    mom = Mom('likModulatorNMFPower', p_cubature=9); d = 0.5 * np.ones(2)
    res = _run_schedules(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], [pr['y']], T, 0, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=2)
    for other in ('pipelined', 'two_buffers'):
        for f in ('Eft', 'Varft', 'MS', 'ttau', 'tnu', 'lZ', 'nlZ'):
            assert np.array_equal(getattr(res[other][0], f), getattr(res['serial'][0], f)), (other, f)
    assert np.all(np.isfinite(res['pipelined'][0].Eft)) and np.all(res['pipelined'][0].Varft > 0)


# ---------------------------------------------------------------------------------------------
# rows a11 / f-4: the EKF energy with its gradient recursion (gf_giekf_modulator_nmf_constraints.m:332-480, GradObj = 'on')
def _grad_problem(D, N, T, seed):
    pr = harness.nmf_problem(D, N, T, seed, 'constraints')
    cons = harness.CONSTRAINTS_DEMO(D); tune = [1] * 7                       # every group tuned: the only case the reference statement runs in
    w, wf = harness.constrained_vectors(pr, cons, tune)
    return pr, cons, tune, w, wf


@pytest.mark.parametrize('D,N,T', [(3, 2, 60), (8, 3, 80), (24, 3, 120)])
def test_ekf_gradient_recursion_as_written_against_oracle(D, N, T):
    """GradObj = 'on' as the reference has it (1+3D+2N slices; the last D*N with the Jacobian derivative w.r.t. an entry of W while
    dm, dP carry the kernel parameter of the same index; unbalanced dF, dPinf beside the balanced F, Pinf): energy and every
    gradient entry against the oracle's literal restatement.  (24, 3) is the configs[3] shape: 27 sites, three tiles per thread."""
    pr, cons, tune, w, wf = _grad_problem(D, N, T, 900 + D)
    t = np.arange(1, T + 1.0)
    e, eg = nagp.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2, cons, wf, tune, 'on')
    lik, p1, p2, W = ogf.ssm.unpack_constraints(w, wf, tune, cons, 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', True)
    gs = oek.grad_setup(model, p1, p2, 'matern32', 'matern52', consistent=False)
    eo, go = oek.run_nlml_grad(model, gs, pr['y'], D, N, w.size, consistent=False)
    n_par = 1 + 3 * D + 2 * N
    assert eg.shape == (w.size,) and not np.any(eg[n_par:])
    assert abs(e - eo) < TOL_LOGZ * abs(eo)
    assert rel(eg[:n_par], go[:n_par]) < 1e-7, np.c_[eg[:n_par], go[:n_par]]
    # the energy is the one the GradObj = 'off' call returns
    e_off, _ = nagp.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2, cons, wf, tune, 'off')
    assert abs(e - e_off) < 1e-11 * abs(e_off)


def test_ekf_gradient_index_error_when_fewer_groups_are_tuned():
    """tune_hypers = [0 0 1 0 1 1 0] (demo_toy_modulators_nmf_constraints.m:40): numel(w) = D + 2N < size(dF,3) -- MATLAB stops at
    gdata(j) with an index error (:453-457); so does the wrapper, before anything runs."""
    D, N, T = 3, 2, 30
    pr = harness.nmf_problem(D, N, T, 5, 'constraints'); cons = harness.CONSTRAINTS_DEMO(D)
    w, wf = harness.constrained_vectors(pr, cons, harness.TUNE_DEMO)
    t = np.arange(1, T + 1.0)
    with pytest.raises(IndexError):
        nagp.gf_giekf_modulator_nmf_constraints(w, t, pr['y'], SSHandle(), None, None, 'matern32', 'matern52', 1, D, N, 3, 2, cons, wf, harness.TUNE_DEMO, 'on')


def test_ekf_gradient_consistent_form_equals_central_differences_of_the_energy():
    """The same recursion with consistent inputs (derivatives carried through the balancing, W entries as slices of their own with
    the direct terms): the gradient of the energy w.r.t. the natural parameters [sigma2, sig1, len1, omega, sig2, len2, W(:)].
    Pinned twice: against the oracle's consistent form, and against central differences of the DEVICE's own energy (GradObj 'off'),
    which the oracle and the golden fixture already pin."""
    from nagp import api as napi
    D, N, T = 4, 2, 70
    pr = harness.nmf_problem(D, N, T, 31)
    p1, p2, W = pr['param1'], pr['param2'], pr['W']; lik = np.array([np.log(pr['w_lik'])])
    theta = np.concatenate([[pr['w_lik']], p1, p2, W.ravel(order='F')])

    def energy(th):
        q1 = th[1:1 + 3 * D]; q2 = th[1 + 3 * D:1 + 3 * D + 2 * N]; Wm = th[1 + 3 * D + 2 * N:].reshape((D, N), order='F')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(q1, q2, 'matern32', 'matern52'))
        plan = Plan(L.KIND_GIEKF, [(blk, Wm, np.array([np.log(th[0])]))], T, ep_itts=1, mode=L.MODE_NLML, l_iter=1)
        plan.upload([pr['y']]); plan.execute(); e = float(plan.download_nlz()[0, 0]); plan.close()
        return e

    blk = pss.balance_blocks(pss.ss_blocks_nmf(p1, p2, 'matern32', 'matern52'))
    e, g = napi.giekf_nlml_grad(blk, W, lik, p1, p2, 'matern32', 'matern52', pr['y'], consistent=True)
    assert abs(e - energy(theta)) < 1e-11 * abs(e) and g.size == theta.size
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', True)
    eo, go = oek.run_nlml_grad(model, oek.grad_setup(model, p1, p2, 'matern32', 'matern52', consistent=True), pr['y'], D, N, theta.size, consistent=True)
    assert abs(e - eo) < TOL_LOGZ * abs(eo) and rel(g, go) < 1e-7
    fd = np.zeros_like(theta)
    for j in range(theta.size):
        h = 1e-5 * max(abs(theta[j]), 1e-3); tp = theta.copy(); tp[j] += h; tm = theta.copy(); tm[j] -= h
        fd[j] = (energy(tp) - energy(tm)) / (2 * h)
    scale = np.maximum(np.abs(fd), 1e-6 * np.max(np.abs(fd)))
    assert np.max(np.abs(g - fd) / scale) < 1e-4, np.c_[g, fd]


@pytest.mark.parametrize('D,N', [(3, 2), (16, 3), (22, 4), (32, 6)])
def test_mfma_gain_kernel_equals_the_valu_gain_kernel(D, N):
    """rts_gain_mfma_kernel (16x16 tiles on the matrix cores, the default when the smoother passes take dense operands) against the
    4x4-tile VALU kernel (NAGP_NO_GAIN_MFMA=1) on the same plans: Sp = 32, 80, 112, 160 (3, 6, 8, 11 waves), two problems, chunks of
    24 steps, a missing observation -- every output to rounding, the jitter counters identical."""
    T = 60
    probs, ys = [], []
    for q in range(2):
        pr = harness.nmf_problem(D, N, T, 8700 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        y = pr['y'].copy(); y[9 + q] = np.nan
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    mom = Mom('likModulatorNMFPower', p_cubature=3); d = np.array([0.6, 0.5])
    res = {}
    for mode in ('mfma', 'solve', 'valu'):      # the default (explicit-inverse form: these blocks of A are well conditioned), its solve form, the VALU kernel
        if mode == 'valu': os.environ['NAGP_NO_GAIN_MFMA'] = '1'
        if mode == 'solve': os.environ['NAGP_GAIN_FORM'] = 'solve'
        try:
            plan = Plan(L.KIND_GF_EP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=2, chunk=24)
            plan.upload(ys); plan.execute(); res[mode] = plan.download(); plan.close()
        finally:
            os.environ.pop('NAGP_NO_GAIN_MFMA', None); os.environ.pop('NAGP_GAIN_FORM', None)
    for q in range(2):
        for mode in ('mfma', 'solve'):
            a, v = res[mode][q], res['valu'][q]
            for f, tol in (('Eft', 1e-9), ('Varft', 1e-9), ('MS', 1e-9), ('lZ', 1e-9), ('ttau', 1e-7), ('tnu', 1e-7)):
                assert rel(getattr(a, f), getattr(v, f)) < tol, (q, f, mode)
            assert relz(a.nlZ, v.nlZ) < 1e-10 and np.array_equal(a.counters, v.counters)
    assert not np.array_equal(res['mfma'][0].MS, res['solve'][0].MS)      # (two different arithmetic routes: equal to rounding, not bit for bit)


def test_gain_kernel_keeps_the_solve_form_when_a_block_of_A_is_badly_conditioned():
    """The explicit-inverse form of the gain (G = A^-1 - A^-1 Q PSkp^-1) multiplies the rounding error of PSkp^-1 by |A_b^-1|; a sub-band
    with a length-scale far below one sample has A_b ~ exp(-50) and the plan must keep the solve form.  Parity with the oracle at the
    usual tolerance, and -- forced onto the inverse form (NAGP_GAIN_FORM=inv) -- visibly worse or not finite, which is what the guard is for."""
    D, N, T = 6, 2, 80
    pr = harness.nmf_problem(D, N, T, 8800)
    p1 = pr['param1'].copy(); p1[D + 1] = 0.02          # length-scale of sub-band 1: 0.02 samples
    blk = pss.ss_blocks_nmf(p1, pr['param2'], 'matern32', 'matern52')
    mom = Mom('likModulatorNMFPower', p_cubature=5); d = 0.5 * np.ones(2)
    w = pr['w'].copy(); w[1 + D + 1] = np.log(0.02)
    t = np.arange(1, T + 1.0)
    o = ogf.gf_ep_modulator_nmf(w, t, pr['y'], None, olik.Mom(olik.LIK_POWER_NMF, p=5), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    out = {}
    for form in ('auto', 'inv'):
        if form == 'inv': os.environ['NAGP_GAIN_FORM'] = 'inv'
        try:
            plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=2, chunk=24)
            plan.upload([pr['y']]); plan.execute(); out[form] = plan.download()[0]; plan.close()
        finally:
            os.environ.pop('NAGP_GAIN_FORM', None)
    assert rel(out['auto'].Eft, o[0]) < TOL_MEAN and rel(out['auto'].Varft, o[1]) < TOL_MEAN and relz(out['auto'].nlZ, o[5]['nlZ']) < TOL_LOGZ
    with np.errstate(all='ignore'):      # (here A_b ~ 1e-36: A_b^-1 overflows the products and the forced inverse form returns NaN)
        bad = np.max(np.abs(out['inv'].Eft - o[0])) / np.max(np.abs(o[0]))
    assert not (bad < TOL_MEAN)


@pytest.mark.parametrize('D,N,k1', [(3, 2, 'matern32'), (16, 3, 'matern32'), (22, 4, 'exp'), (32, 6, 'matern32')])
def test_mfma_fixed_site_filter_equals_the_valu_filter(D, N, k1):
    """gf_filter_lin_mfma_kernel (covariance in the MFMA accumulator layout, rank-M update on the matrix cores; opt-in:
    NAGP_LIN_MFMA=1) against the default 4x4-tile kernel: the sweeps >= 2 of the same plans -- 5, 19, 26, 38 sites (1, 4, 8 waves'
    worth of tiles; 2-state sub-band blocks in the third case), missing observations, a continuation chunk -- to rounding."""
    T = 70
    probs, ys = [], []
    for q in range(2):
        pr = harness.nmf_problem(D, N, T, 8800 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, 'matern52'))
        y = pr['y'].copy(); y[13 + q] = np.nan; y[T - 2] = np.nan
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    mom = Mom('likModulatorNMFPower', p_cubature=3); d = np.array([0.6, 0.5, 0.5])
    res = {}
    for mode in ('mfma', 'valu'):
        if mode == 'mfma': os.environ['NAGP_LIN_MFMA'] = '1'
        try:
            plan = Plan(L.KIND_GF_EP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, chunk=24)
            plan.upload(ys); plan.execute(); res[mode] = plan.download(want_MF=True); tm = plan.timings(); plan.close()
            assert tm['launches']['filter_lin'] >= 2                      # sweeps 2 and 3 ran the fixed-site kernel (one launch per sweep, or per chunk behind the previous sweep's smoother)
        finally:
            os.environ.pop('NAGP_LIN_MFMA', None)
    for q in range(2):
        a, v = res['mfma'][q], res['valu'][q]
        for f, tol in (('Eft', 1e-9), ('Varft', 1e-9), ('MS', 1e-9), ('MF', 1e-9), ('lZ', 1e-9), ('ttau', 1e-7), ('tnu', 1e-7)):
            assert rel(getattr(a, f), getattr(v, f)) < tol, (q, f)
        assert relz(a.nlZ, v.nlZ) < 1e-10 and np.array_equal(a.counters, v.counters)


def test_plan_that_cannot_fit_returns_enomem_and_the_device_stays_usable():
    """A plan whose filtered-covariance history alone exceeds the HBM of the card (64 segments x 400 000 steps at 38 sites = 2.4 TB):
    nagp_plan_create fails with NAGP_ENOMEM (hipMalloc says so before anything is touched), frees what it had allocated, and the
    next plan on the same device works."""
    D, N = 32, 6
    pr = harness.nmf_problem(D, N, 8, 1, 'constraints')
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
    mom = Mom('likModulatorNMFPower', p_cubature=3)
    with pytest.raises(nagp.NagpError, match='out of device memory'):
        Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))] * 64, 400000, mom=mom, ep_fraction=0.5, ep_damping=[0.5, 0.5], ep_itts=2)
    plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], 8, mom=mom, ep_fraction=0.5, ep_damping=[0.5, 0.5], ep_itts=2)
    plan.upload([pr['y']]); plan.execute(); o = plan.download()[0]; plan.close()
    assert np.all(np.isfinite(o.Eft)) and np.all(o.Varft > 0)


def test_randomised_schedules_pipelined_equals_serial():
    """tools/gpu_fuzz_schedules.py, fixed seed: 16 random draws of shape / family (EP, EKF) / segments / chunk length / number of
    (G, Delta) buffers / sweeps / missing data / smoothed covariances -- the pipelined plan equals the serial plan bit for bit in all."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('gpu_fuzz_schedules', os.path.join(os.path.dirname(__file__), '..', 'tools', 'gpu_fuzz_schedules.py'))
    mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)
    assert mod.run_cases(16, 31, verbose=False) == 0


@pytest.mark.parametrize('family', ['gf', 'ihgp'])
def test_sparse_point_site_refresh_equals_the_generic_one(family):
    """ep_site_sp_kernel (likModulatorNMFPower in the staged sparse-point form) against ep_site_kernel (generic mom_eval,
    NAGP_NO_SPARSE_EP=1) on the same plans -- three problems in one launch, three sweeps, so that two refreshes feed the later
    sweeps: sites, marginals and nlZ agree to rounding (1e-9 of the scale; the two forms order their sums differently)."""
    D, N, T = 12, 4, 160
    probs, ys = [], []
    for q in range(3):
        pr = harness.nmf_problem(D, N, T, 9400 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(pr['y'])
    mom = Mom('likModulatorNMFPower', p_cubature=7); d = 0.5 * np.ones(3)
    res = {}
    for mode in ('sparse', 'generic'):
        if mode == 'generic': os.environ['NAGP_NO_SPARSE_EP'] = '1'
        try:
            if family == 'gf':
                plan = Plan(L.KIND_GF_EP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
            else:
                plan = Plan(L.KIND_IHGP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
            plan.upload(ys); plan.execute(); res[mode] = plan.download(); plan.close()
        finally:
            os.environ.pop('NAGP_NO_SPARSE_EP', None)
    for q in range(3):
        a, b = res['sparse'][q], res['generic'][q]
        for f in ('Eft', 'Varft', 'ttau', 'tnu'):
            assert rel(getattr(a, f), getattr(b, f)) < 1e-9, (q, f)
        assert relz(a.nlZ, b.nlZ) < 1e-10


def test_packed_delta_slots_equal_the_dense_ones_bit_for_bit(monkeypatch):
    """Column-owner smoother plans keep Delta as its lower 16x16 tiles (GainPar::dpacked); NAGP_DENSE_DELTA=1 keeps the dense matrix.
    Same values through a different layout: every output of an 8-segment S = 146 plan (several chunks, three sweeps) is bit-equal."""
    D, N, T, B = 32, 6, 120, 4
    probs, ys = [], []
    for q in range(B):
        pr = harness.nmf_problem(D, N, T, 9600 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(pr['y'])
    mom = Mom('likModulatorNMFPower', p_cubature=5); d = 0.5 * np.ones(3)
    res = {}
    for mode in ('packed', 'dense'):
        if mode == 'dense': monkeypatch.setenv('NAGP_DENSE_DELTA', '1')
        plan = Plan(L.KIND_GF_EP, probs, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3, chunk=32)
        plan.upload(ys); plan.execute(); res[mode] = plan.download(); plan.close()
        monkeypatch.delenv('NAGP_DENSE_DELTA', raising=False)
    for q in range(B):
        for f in ('Eft', 'Varft', 'MS', 'ttau', 'tnu', 'R', 'lZ', 'nlZ', 'maxDiffM', 'maxDiffP'):
            assert np.array_equal(getattr(res['packed'][q], f), getattr(res['dense'][q], f), equal_nan=True), (q, f)


@pytest.mark.parametrize('kind', ['gf', 'ekf'])
def test_slots_recycled_from_the_filtered_covariances_equal_the_serial_schedule(kind):
    """When the free memory does not hold a (G, Delta) buffer per chunk, a column-owner plan takes the buffers of the chunks the filter
    finishes last from the part of PF whose gains already exist (nagp_api.hip, "recycled"; padding rows written by the gain kernel,
    the buffer's problems PF's stride apart) instead of computing gains twice.  Forced here with four full buffers for five full chunks
    (gf: S = 146, Sp = 160, four segments; EKF: S = 105, Sp = 112, two segments, restart from the smoothed state of step 0, which reads
    PF_0; three sweeps, T - 1 = 6 x 32): every output equals the serial schedule's and the recompute schedule's bit for bit; the
    plan's memory grows by the delta vectors of the one recycled buffer only."""
    ekf = kind == 'ekf'
    D, N, T, B, chunk = (24, 3, 193, 2, 32) if ekf else (32, 6, 193, 4, 32)
    probs, ys = [], []
    for q in range(B):
        pr = harness.nmf_problem(D, N, T, 9700 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        y = pr['y'].copy()
        if not ekf: y[11 * q + 3] = np.nan
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
    kw = dict(ep_itts=3, l_iter=2) if ekf else dict(mom=Mom('likModulatorNMFPower', p_cubature=5), ep_fraction=0.5, ep_damping=0.5 * np.ones(3), ep_itts=3)
    res, nbytes = {}, {}
    for name, env in (('recycled', {'NAGP_PIPELINE_SLOTS': '4'}), ('recompute', {'NAGP_PIPELINE_SLOTS': '4', 'NAGP_NO_RECYCLE': '1'}),
                      ('serial', {'NAGP_NO_PIPELINE': '1'})):
        os.environ.update(env)
        try:
            plan = Plan(L.KIND_GIEKF if ekf else L.KIND_GF_EP, probs, T, chunk=chunk, **kw)
            plan.upload(ys); plan.execute(); res[name] = plan.download(want_MF=True); nbytes[name] = plan.device_bytes(); plan.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert nbytes['recycled'] - nbytes['recompute'] == B * chunk * probs[0][0].S * 8
    fields = ('Eft', 'Varft', 'MS', 'MF', 'maxDiffP') + (() if ekf else ('ttau', 'tnu', 'R', 'lZ', 'nlZ', 'maxDiffM'))
    for q in range(B):
        for other in ('recycled', 'recompute'):
            for f in fields:
                assert np.array_equal(getattr(res[other][q], f), getattr(res['serial'][q], f), equal_nan=True), (other, q, f)
            assert np.array_equal(res[other][q].counters, res['serial'][q].counters)


@pytest.mark.parametrize('fail_at', [1, 2, 3, 5])
def test_slot_allocations_that_fail_are_done_without(fail_at):
    """The (G, Delta) slots beyond the first are an optimisation sized from one hipMemGetInfo snapshot; an allocation that fails later
    (fragmentation, a second plan or process) must not fail the plan.  Test hook NAGP_TEST_SLOT_ENOMEM=n: the slot allocations of the plan
    fail once n slots exist -- n = 1, 2: not enough for the pipeline, serial schedule on the one slot; 3: two full slots + the small one
    (the third full slot is given back for it); 5: four full + small.  Every outcome equals the serial schedule bit for bit."""
    D, N, T, B, chunk = 16, 3, 193, 2, 32
    probs, ys = [], []
    for q in range(B):
        pr = harness.nmf_problem(D, N, T, 9800 + q)
        blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(pr['y'])
    kw = dict(mom=Mom('likModulatorNMFPower', p_cubature=5), ep_fraction=0.5, ep_damping=0.5 * np.ones(3), ep_itts=3)
    res, nbytes = {}, {}
    for name, env in (('starved', {'NAGP_TEST_SLOT_ENOMEM': str(fail_at)}), ('free', {}), ('serial', {'NAGP_NO_PIPELINE': '1'})):
        os.environ.update(env)
        try:
            plan = Plan(L.KIND_GF_EP, probs, T, chunk=chunk, **kw)
            plan.upload(ys); plan.execute(); res[name] = plan.download(want_MF=True); nbytes[name] = plan.device_bytes(); plan.close()
        finally:
            for k in env:
                os.environ.pop(k, None)
    assert nbytes['starved'] < nbytes['free']
    for q in range(B):
        for f in ('Eft', 'Varft', 'MS', 'MF', 'ttau', 'tnu', 'R', 'lZ', 'nlZ', 'maxDiffM', 'maxDiffP'):
            assert np.array_equal(getattr(res['starved'][q], f), getattr(res['serial'][q], f), equal_nan=True), (q, f)
            assert np.array_equal(getattr(res['free'][q], f), getattr(res['serial'][q], f), equal_nan=True), (q, f)


def test_mixture_at_the_papers_size_on_the_full_covariance_path():
    """experiments/source_sep_piano.m:78-90: three sources of 16 channels and 3 NMF components each -- 48 sub-bands + 9 modulators =
    57 sites, 3 249 covariance tiles: eight tiles per thread in the gain kernel and the (VALU) smoother passes, four lower tiles per
    thread in the filter.  gf_ep_mods_nmf_mixture with the exp sub-band kernels of the driver (S = 123), and gf_ep_modulator_nmf at the
    same 48 / 9 with Matern-3/2 sub-bands (S = 219), against the oracle."""
    shapes = [(16, 3)] * 3; k1 = ['exp'] * 3; k2 = ['matern52'] * 3
    T = 14; t = np.arange(1, T + 1.0)
    mp = harness.mixture_problem(shapes, T, 77, k1, k2)
    mom, omom = _mixture_moms('likModulatorPreCalcwn', 3, 9)
    a = nagp.gf_ep_mods_nmf_mixture(mp['w'], t, mp['y'], SSHandle(), mom, t, k1, k2, 3, 0.75, 0.2, 2, nargout=6)
    b = omx.gf_ep_mods_nmf_mixture(mp['w'], t, mp['y'], None, omom, t, k1, k2, 3, 0.75, 0.2, 2)
    assert a[0].shape == (57, T)
    assert rel(a[0], b[0]) < TOL_MEAN and rel(a[1], b[1]) < TOL_MEAN
    assert rel(a[5]['ttau'], b[5]['ttau']) < TOL_SITE and rel(a[5]['tnu'], b[5]['tnu']) < TOL_SITE and rel(a[5]['MS'], b[5]['MS']) < TOL_MEAN
    # the driver's own rule, ut9 in nine dimensions (3 973 sigma points): the ADF launch evaluates them 256 per pass to fit the LDS beside the W panel
    mom9, omom9 = _mixture_moms('likModulatorPreCalcwn', 9, 9)
    a = nagp.gf_ep_mods_nmf_mixture(mp['w'], t[:8], mp['y'][:8], SSHandle(), mom9, t[:8], k1, k2, 3, 0.75, 0.2, 2, nargout=6)
    b = omx.gf_ep_mods_nmf_mixture(mp['w'], t[:8], mp['y'][:8], None, omom9, t[:8], k1, k2, 3, 0.75, 0.2, 2)
    assert rel(a[0], b[0]) < TOL_MEAN and rel(a[1], b[1]) < TOL_MEAN and rel(a[5]['ttau'], b[5]['ttau']) < TOL_SITE
    D, N, T = 48, 9, 12
    pr = harness.nmf_problem(D, N, T, 9911); t = np.arange(1, T + 1.0)
    d = 0.5 * np.ones(2)
    Eft, Varft, _, _, _, out = nagp.gf_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), Mom('likModulatorNMFPower', p_cubature=3), t, 'matern32', 'matern52', 1, D, N,
                                                        0.5, d, 2, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, pr['y'], None, olik.Mom(olik.LIK_POWER_NMF, p=3), t, 'matern32', 'matern52', 1, D, N, 0.5, d, 2)
    assert rel(Eft, o[0]) < TOL_MEAN and rel(Varft, o[1]) < TOL_MEAN and relz(out['nlZ'], o[5]['nlZ']) < TOL_LOGZ
    assert rel(out['ttau'], o[5]['ttau']) < TOL_SITE and rel(out['tnu'], o[5]['tnu']) < TOL_SITE
    # and the EKF family at the same 48 / 9 (two global iterations, two inner ones)
    r = nagp.gf_giekf_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), None, t, 'exp', 'matern32', 1, D, N, 2, 2, nargout=2)
    oe = oek.gf_giekf_modulator_nmf(pr['w'], t, pr['y'], None, None, t, 'exp', 'matern32', 1, D, N, 2, 2)
    assert rel(r[0], oe[0]) < TOL_MEAN and rel(r[1], oe[1]) < TOL_MEAN


def test_lds_tight_shapes_are_served_or_refused_never_wrong():
    """59 - 63 sites: the filter's W panel alone is 110 - 127 KB of LDS.  The plan falls back to a shorter I/O ring and cubature tables in
    global memory; what still does not fit is refused with NAGP_EUNSUPPORTED -- in the pipelined schedule as well, whose filter launch asks
    for the whole LDS of its CU (that request once replaced a LARGER need: wrong results instead of a refusal)."""
    D, N, T = 56, 3, 40
    pr = harness.nmf_problem(D, N, T, 4242, 'constraints')
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'exp', 'matern32'))
    assert blk.M == 59
    d = 0.5 * np.ones(2)
    o = ogf.run_predict(ogf.assemble(np.log(pr['w_lik']) * np.ones(1), pr['param1'], pr['param2'], pr['W'], 'exp', 'matern32', True), pr['y'],
                        olik.Mom(olik.LIK_POWER_NMF, p=3), 0.5, d, 2)
    for chunk in (16, 0):          # several chunks (pipelined) / one chunk (serial)
        plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=Mom('likModulatorNMFPower', p_cubature=3), ep_fraction=0.5, ep_damping=d, ep_itts=2, chunk=chunk)
        plan.upload([pr['y']]); plan.execute(); r = plan.download()[0]; plan.close()
        assert rel(r.Eft, o['Eft']) < TOL_MEAN and rel(r.Varft, o['Varft']) < TOL_MEAN and relz(r.nlZ, o['nlZ']) < TOL_LOGZ, chunk
    pr = harness.nmf_problem(54, 9, T, 4243, 'constraints')
    blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'exp', 'matern32'))
    assert blk.M == 63
    for chunk in (16, 0):
        with pytest.raises(nagp.NagpError, match='unsupported shape'):
            Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], T, mom=Mom('likModulatorNMFPower', p_cubature=3), ep_fraction=0.5, ep_damping=d, ep_itts=2, chunk=chunk)


# ---------------------------------------------------------------------------------------------
# End-to-end parity at the sizes the contract is stated on (BASELINE.json configs, north_star's "|dlogZ|/|logZ| < 1e-5 on a 200k-sample
# sweep"): the bench.py workloads -- cfg2audio / cfg4audio are bench.py's cfg2 / cfg4 inputs themselves (the decoded audio files, cfg2 with
# the drivers' damping 0.1), cfg2 / cfg4 the same shapes and lengths on prior samples -- ALL sweeps, default chunking, pipelined schedule, parallel-in-time scans -- against the
# sequential algorithm of the compiled oracle (gf_ep_modulator_nmf.m:126-283, ihgp_ep_modulator_nmf.m:233-442,
# gf_giekf_modulator_nmf.m:126-221).  The CPU legs were started by the first test of this file.
@pytest.mark.parametrize('name', ['cfg3', 'cfg3sqrt', 'cfg2', 'cfg2audio', 'cfg5seg', 'cfg4', 'cfg4audio'])
def test_full_length_all_sweeps_against_the_sequential_cpu_algorithm(name, full_length_refs):
    flp = full_length_refs.mod
    full_length_refs.start()
    pr = full_length_refs.problems[name]
    out, _ = flp.gpu_run(name, pr)
    if name in ('cfg2', 'cfg2audio'):      # the pipelined schedule against the serial one at full length, three sweeps: every output bit for bit
        ser, _ = flp.gpu_run(name, pr, env={'NAGP_NO_PIPELINE': '1'})
        assert flp.bit_equal(out, ser) == []
    ref = full_length_refs.result(name)
    assert ref['status'] == 0
    m = flp.compare(name, out, ref)
    bad = {k: v for k, (v, tol) in m.items() if tol is not None and not v <= tol}
    assert not bad, (name, bad)
    assert pr['y'].size == {'cfg3': 200000, 'cfg3sqrt': 200000, 'cfg2': 84010, 'cfg2audio': 84010, 'cfg4': 88200, 'cfg4audio': 88200, 'cfg5seg': 20000}[name]
    if name in ('cfg2audio', 'cfg4audio'):      # the inputs ARE the decoded audio files BASELINE names (speech_74.wav / stim312_wind.wav), as bench.py feeds them
        z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'audio_%s.npz' % {'cfg2audio': 'speech_74', 'cfg4audio': 'stim312_wind'}[name]))
        x = z['samples'].astype(np.float64) / 32768.0
        assert np.array_equal(pr['y'], x / np.std(x))
    if name not in ('cfg4', 'cfg4audio'):      # north_star's sentence with three orders of magnitude to spare
        assert np.max(np.abs(out.nlZ - ref['nlZ']) / np.abs(ref['nlZ'])) < TOL_LOGZ < 1e-5


def test_speech_file_with_the_undamped_recipe_properties_that_hold_without_conditioning(full_length_refs):
    """audio/speech_74.wav with damping 0.5 (the round-4 bench recipe): the reference algorithm itself is chaotic there (its own outputs move
    by 50 % under a 1e-13 relative change of y: tests/golden/audio_conditioning.json, cfg2audio_d05), so no value can be compared.  What
    holds regardless: the pipelined schedule equals the serial one bit for bit, a one-sweep run equals the first sweep of the three-sweep
    run (lZ, nlZ[0]), every output is finite, the clamped sites are non-negative."""
    flp = full_length_refs.mod
    pr = flp.problem('cfg2audio_d05')
    out, _ = flp.gpu_run('cfg2audio_d05', pr)
    ser, _ = flp.gpu_run('cfg2audio_d05', pr, env={'NAGP_NO_PIPELINE': '1'})
    assert flp.bit_equal(out, ser) == []
    assert np.all(np.isfinite(out.Eft)) and np.all(np.isfinite(out.Varft)) and np.all(np.isfinite(out.nlZ)) and np.all(np.isfinite(out.tnu))
    assert np.all(out.ttau >= 0.0)
    blk = pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    plan = Plan(L.KIND_GF_EP, [(blk, pr['W'], np.log(pr['w_lik']))], pr['y'].size, mom=Mom('likModulatorNMFPower', p_cubature=9), ep_fraction=0.5,
                ep_damping=0.5 * np.ones(1), ep_itts=1)
    plan.upload([pr['y']]); plan.execute(); one = plan.download(want_MS=False)[0]; plan.close()
    assert one.nlZ[0] == out.nlZ[0]


def test_the_real_eight_segment_cfg5_plan_recycled_slots_under_memory_pressure():
    """bench.py's cfg5_strong: 8 segments x 100 000 steps at S = 146 in ONE plan -- 78 GB of filtered covariances, nine (G, Delta) slots of
    their own and three recycled from PF.  (i) every output of every segment equals, bit for bit, the same plan without
    the recycled slots (NAGP_NO_RECYCLE=1: scratch-slot schedule, gains of the chunks without a slot computed twice); (ii) two of its
    segments equal their single-segment plans to rounding (a single segment picks other span lengths -- latency regime -- so the scans
    associate differently: 1e-9, not bit equality); (iii) the per-sweep nlZ of all segments are finite."""
    D, N, T, Tp = 32, 6, 100000, 12500
    probs, ys = [], []
    for q in range(8):
        pr = harness.nmf_problem(D, N, Tp, 5000 + q, 'constraints')
        blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'))
        probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(np.tile(pr['y'], T // Tp))      # (timing and memory do not depend on the numbers)
    mom = Mom('likModulatorNMFPower', p_cubature=7); d = 0.5 * np.ones(3)
    fields = ('Eft', 'Varft', 'ttau', 'tnu', 'lZ', 'nlZ', 'maxDiffM', 'maxDiffP')

    def run(pp, yy, env=None):
        old = {k: os.environ.get(k) for k in (env or {})}
        os.environ.update(env or {})
        try:
            plan = Plan(L.KIND_GF_EP, pp, T, mom=mom, ep_fraction=0.5, ep_damping=d, ep_itts=3)
        finally:
            for k, v in old.items():
                os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)
        plan.upload(yy); plan.execute(); o = plan.download(want_MS=False); nb = plan.device_bytes(); plan.close()
        return o, nb
    full, nbytes = run(probs, ys)
    assert nbytes > 200e9                                  # PF 78 GB + nine slots of 19.8 GiB: the plan fills the card
    assert all(np.all(np.isfinite(o.nlZ)) and np.all(np.isfinite(o.Eft)) and np.all(o.Varft > 0) and o.counters[0] == 0 for o in full)
    keep = {q: {f: np.array(getattr(full[q], f)) for f in fields} for q in range(8)}
    del full
    norec, _ = run(probs, ys, env={'NAGP_NO_RECYCLE': '1'})
    for q in range(8):
        for f in fields:
            assert np.array_equal(keep[q][f], getattr(norec[q], f), equal_nan=True), (q, f)
    del norec
    for q in (2, 5):
        (one,), _ = run([probs[q]], [ys[q]])
        assert rel(one.Eft, keep[q]['Eft']) < 1e-9 and rel(one.Varft, keep[q]['Varft']) < 1e-9
        assert rel(one.ttau, keep[q]['ttau']) < 1e-8 and rel(one.tnu, keep[q]['tnu']) < 1e-8
        assert relz(one.nlZ, keep[q]['nlZ']) < 1e-11
