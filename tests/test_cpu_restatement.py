"""CPU (-m "not gpu"): the compiled dense-as-written restatement (oracle/cpu/nagp_cpu.cpp, the timed CPU baseline of bench.py)
against the NumPy oracle -- through the committed golden vectors, which are oracle outputs (tools/make_golden.py), and directly
on small cases.  Two independent restatements of the reference loops agreeing to 1e-9 is the CPU<->CPU pin SURVEY 8(c)(6) asks for.
Both are test / measurement infrastructure; neither is on the product path."""
import os

import numpy as np
import pytest

from oracle import cpu as ocpu, gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik, ss as oss

GOLD = os.path.join(os.path.dirname(__file__), 'golden')
TOL = 1e-9            # CPU <-> CPU: means / variances relative to the array's largest magnitude
TOL_SITE = 1e-6       # site parameters (divisions by 1 + d2*v amplify rounding): the tolerance the GPU parity tests state


def gold(name):
    return np.load(os.path.join(GOLD, name + '.npz'))


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b))
    return float(np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-300))


def test_compiled_mom_equals_oracle_mom_for_the_three_likelihoods():
    rng = np.random.default_rng(3); D, N = 5, 3
    W = rng.uniform(0, 0.5, (D, N))
    from oracle import cubature as ocub
    wn, xn = ocub.sigma_points(7, N, True)
    cases = [(olik.Mom(olik.LIK_POWER_NMF, p=9), D + N, W), (olik.Mom(olik.LIK_POWER_NMF, p=5, link=olik.exp_link()), D + N, W),
             (olik.Mom(olik.LIK_POWER_NMF_SQRT, link=olik.softplus_link(1.0), wn=wn, xn_unscaled=xn), D + N, W),
             (olik.Mom(olik.LIK_POWER, p=7), 2 * D, None)]
    for om, M, Wc in cases:
        for trial in range(4):
            mu = rng.normal(0, 1, M); s2 = rng.uniform(0.05, 2.0, M); y = float(rng.normal()); alpha = [1.0, 0.5, 0.75, 0.3][trial]
            a = om(np.log(3e-2), mu, s2, Wc, alpha, np.array([y]), 0)
            b = ocpu.mom(om, np.log(3e-2), y, mu, s2, Wc, alpha)
            assert abs(a[0] - b[0]) < 1e-12 * max(1.0, abs(a[0])) and rel(b[1], a[1]) < 1e-11 and rel(b[2], a[2]) < 1e-10
    # a floored Z (max(Z, jitter)) and the NaN rule of MATLAB's max
    om = olik.Mom(olik.LIK_POWER_NMF, p=5); mu = np.zeros(D + N); s2 = 1e-4 * np.ones(D + N)
    a = om(np.log(1e-6), mu, s2, W, 1.0, np.array([50.0]), 0); b = ocpu.mom(om, np.log(1e-6), 50.0, mu, s2, W, 1.0)
    assert a[0] == b[0] == np.log(1e-10)


def test_cfg1_gf_ep_modulator_golden_full_size():
    g = gold('cfg1_gf_ep_modulator'); D = 4
    w = g['w']; param = np.exp(w[1:])
    model = ogf.assemble(w[:1], param[:3 * D], param[3 * D:], None, 'matern32', 'matern52', balance=True)
    r = ocpu.gf_predict(model, g['y'], olik.Mom(olik.LIK_POWER, p=9), 0.5, g['ep_damping'], 5, D, D, predict_at_k1=True)
    assert r['status'] == 0 and r['counters']['chol_retries'] == 0
    assert rel(r['Eft'], g['Eft']) < TOL and rel(r['Varft'], g['Varft']) < TOL and rel(r['nlZ'], g['nlZ']) < TOL
    assert rel(r['ttau'], g['ttau']) < TOL_SITE and rel(r['tnu'], g['tnu']) < TOL_SITE and rel(r['lZ'], g['lZ']) < 1e-8
    assert rel(r['maxDiffM'], g['maxDiffM']) < 1e-7 and rel(r['maxDiffP'], g['maxDiffP']) < 1e-7


def test_cfg2_gf_ep_modulator_nmf_golden_with_missing_data():
    g = gold('cfg2_gf_ep_modulator_nmf'); D, N = int(g['D']), int(g['N'])
    model = ogf.build_model_nmf(g['w'], 'matern32', 'matern52', 1, D, N, balance=False)
    r = ocpu.gf_predict(model, g['y'], olik.Mom(olik.LIK_POWER_NMF, p=9), 0.5, 0.5 * np.ones(3), 3, D, N)
    assert r['status'] == 0
    assert rel(r['Eft'], g['Eft']) < TOL and rel(r['Varft'], g['Varft']) < TOL and rel(r['nlZ'], g['nlZ']) < TOL
    assert rel(r['ttau'], g['ttau']) < TOL_SITE and rel(r['tnu'], g['tnu']) < TOL_SITE


def test_cfg3_ihgp_golden():
    g = gold('cfg3_ihgp_ep_modulator_nmf'); D, N = int(g['D']), int(g['N'])
    lik, p1, p2, W = oss.unpack_log(g['w'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', True, True)
    tabs = oih.build_tables(model)
    r = ocpu.ihgp_predict(model, g['y'], olik.Mom(olik.LIK_POWER_NMF, p=7), 0.5, 0.5 * np.ones(3), 3, D, N, tabs)
    assert r['status'] == 0
    assert rel(r['Eft'], g['Eft']) < TOL and rel(r['Varft'], g['Varft']) < TOL and rel(r['nlZ'], g['nlZ']) < TOL
    assert rel(r['ttau'], g['ttau']) < TOL_SITE
    assert np.array_equal(np.isinf(r['R']), np.isinf(g['R']))


def test_cfg4_giekf_golden_both_variants():
    g = gold('cfg4_gf_giekf_modulator_nmf'); D, N = int(g['D']), int(g['N'])
    lik, p1, p2, W = oss.unpack_constraints(g['w'], g['w_fixed'], list(g['tune_hypers']), g['constraints'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', balance=True)
    r = ocpu.giekf_predict(model, g['y'], D, N, 3, 1, constraints_variant=True)
    assert r['status'] == 0 and rel(r['Eft'], g['Eft']) < TOL and rel(r['Varft'], g['Varft']) < TOL and rel(r['maxDiffP'], g['maxDiffP']) < 1e-7
    lik, p1, p2, W = oss.unpack_log(g['w_log'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', balance=True)
    r = ocpu.giekf_predict(model, g['y'], D, N, 2, 2, constraints_variant=False)
    assert rel(r['Eft'], g['Eft_plain']) < TOL and rel(r['Varft'], g['Varft_plain']) < TOL


def test_cfg5_constraints_golden_S146():
    g = gold('cfg5_gf_ep_modulator_nmf_constraints'); D, N = int(g['D']), int(g['N'])
    lik, p1, p2, W = oss.unpack_constraints(g['w'], g['w_fixed'], list(g['tune_hypers']), g['constraints'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', balance=True)
    T = 200                                    # a prefix keeps the CPU suite short: compared with the oracle run on the same prefix
    y = g['y'][:T]
    r = ocpu.gf_predict(model, y, olik.Mom(olik.LIK_POWER_NMF, p=7), 0.5, 0.5 * np.ones(2), 2, D, N)
    o = ogf.run_predict(model, y, olik.Mom(olik.LIK_POWER_NMF, p=7), 0.5, 0.5 * np.ones(2), 2)
    assert model['A'].shape[0] == 146 and r['status'] == 0
    assert rel(r['Eft'], o['Eft']) < TOL and rel(r['Varft'], o['Varft']) < TOL and rel(r['nlZ'], o['nlZ']) < TOL
    assert rel(r['ttau'], o['ttau']) < TOL_SITE


def test_structured_form_equals_the_dense_as_written_form():
    """structured = block-diagonal A, selection H, P - K*W': the stronger CPU baseline of bench.py computes the same numbers as the
    dense-as-written form (and therefore as the oracle) to rounding, on the golden shapes of all three families"""
    g = gold('cfg2_gf_ep_modulator_nmf'); D, N = int(g['D']), int(g['N'])
    model = ogf.build_model_nmf(g['w'], 'matern32', 'matern52', 1, D, N, balance=False)
    r = ocpu.gf_predict(model, g['y'], olik.Mom(olik.LIK_POWER_NMF, p=9), 0.5, 0.5 * np.ones(3), 3, D, N, structured=True)
    assert rel(r['Eft'], g['Eft']) < TOL and rel(r['Varft'], g['Varft']) < TOL and rel(r['nlZ'], g['nlZ']) < TOL and rel(r['ttau'], g['ttau']) < TOL_SITE
    g = gold('cfg3_ihgp_ep_modulator_nmf'); D, N = int(g['D']), int(g['N'])
    lik, p1, p2, W = oss.unpack_log(g['w'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', True, True)
    r = ocpu.ihgp_predict(model, g['y'], olik.Mom(olik.LIK_POWER_NMF, p=7), 0.5, 0.5 * np.ones(3), 3, D, N, oih.build_tables(model), structured=True)
    assert rel(r['Eft'], g['Eft']) < TOL and rel(r['Varft'], g['Varft']) < TOL and rel(r['nlZ'], g['nlZ']) < TOL and rel(r['ttau'], g['ttau']) < TOL_SITE
    g = gold('cfg4_gf_giekf_modulator_nmf'); D, N = int(g['D']), int(g['N'])
    lik, p1, p2, W = oss.unpack_log(g['w_log'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', balance=True)
    r = ocpu.giekf_predict(model, g['y'], D, N, 2, 2, structured=True)
    assert rel(r['Eft'], g['Eft_plain']) < TOL and rel(r['Varft'], g['Varft_plain']) < TOL


def test_cholesky_retry_and_failure_follow_the_oracle():
    """the jitter branch (C-7) in the compiled restatement: same retry count as the NumPy oracle, same outputs; a matrix that
    fails twice returns the not-PD status where the oracle's chol throws"""
    from nagp import harness
    D, N, T = 3, 2, 40
    pr = harness.nmf_problem(D, N, T, 11)
    lik, p1, p2, W = oss.unpack_log(pr['w'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', False)
    om = olik.Mom(olik.LIK_POWER_NMF, p=5); d = 0.5 * np.ones(2)
    m2 = dict(model); P = model['Pinf'].copy(); P[17, 17] -= 1e-8; m2['Pinf'] = P
    o = ogf.run_predict(m2, pr['y'], om, 0.5, d, 2)
    r = ocpu.gf_predict(m2, pr['y'], om, 0.5, d, 2, D, N)
    assert r['counters']['chol_retries'] == o['counters']['chol_retries'] == 2 * (T - 1)
    assert rel(r['Eft'], o['Eft']) < TOL and rel(r['Varft'], o['Varft']) < TOL
    m3 = dict(model); P = model['Pinf'].copy(); P[14, 14] -= 1e-4; m3['Pinf'] = P
    assert ocpu.gf_predict(m3, pr['y'], om, 0.5, d, 2, D, N)['status'] == -6
    with pytest.raises(np.linalg.LinAlgError):
        ogf.run_predict(m3, pr['y'], om, 0.5, d, 2)


def test_segments_over_host_threads_equal_the_single_calls():
    """the OpenMP driver the bench times: independent segments, each equal to its own single call"""
    from nagp import harness
    D, N, T = 4, 2, 60
    pr = harness.nmf_problem(D, N, T, 21)
    lik, p1, p2, W = oss.unpack_log(pr['w'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', False)
    om = olik.Mom(olik.LIK_POWER_NMF, p=5); d = 0.5 * np.ones(2)
    ys = [harness.nmf_problem(D, N, T, 30 + q)['y'] for q in range(5)]
    nlz, st = ocpu.segments('gf', model, ys, om, 0.5, d, 2, D, N)
    assert st == 0 and ocpu.threads() >= 1
    for q in (0, 4):
        one = ocpu.gf_predict(model, ys[q], om, 0.5, d, 2, D, N)
        assert np.array_equal(nlz[q], one['nlZ'])
    mi = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', True, True)
    tabs = oih.build_tables(mi)
    nlz, st = ocpu.segments('ihgp', mi, ys[:3], om, 0.5, d, 2, D, N, tables=tabs)
    assert st == 0 and np.array_equal(nlz[1], ocpu.ihgp_predict(mi, ys[1], om, 0.5, d, 2, D, N, tabs)['nlZ'])
