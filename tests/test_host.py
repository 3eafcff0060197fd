"""CPU: host-side logic of the product (model construction, tables, unpacking, the C-ABI library's
exports, multi-process sharding) -- no GPU compute."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import nagp
from nagp import cubature as pc, harness, ihgp_tables, ss as pss
from nagp.api import _blocks_from_dense, _merge_inputs, _stack_sources, _unpack_constraints, _unpack_log, Mom, SSHandle
from oracle import cubature as oc, ss as oss, ihgp as oih, gf_ep as ogf, mixture as omx

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize('p', [3, 5, 7, 9])
@pytest.mark.parametrize('n', [1, 2, 3, 6])
def test_product_cubature_equals_reference_rule_as_a_set(p, n):
    W1, S1 = pc.utp_ws(p, n); W2, S2 = oc.utp_ws(p, n)
    a = sorted(zip(map(tuple, np.round(S1.T, 11)), W1)); b = sorted(zip(map(tuple, np.round(S2.T, 11)), W2))
    assert len(a) == len(b)
    assert all(x[0] == y[0] and abs(x[1] - y[1]) < 1e-11 * max(1, abs(y[1])) for x, y in zip(a, b))


def test_gauss_hermite_grid():
    w1, x1 = pc.mvhermgauss_unit(2, 5); w2, x2 = oc.mvhermgauss_unit(2, 5)
    a = sorted(zip(map(tuple, np.round(x1.T, 10)), w1)); b = sorted(zip(map(tuple, np.round(x2.T, 10)), w2))
    assert all(x[0] == y[0] and abs(x[1] - y[1]) < 1e-12 for x, y in zip(a, b))
    assert np.allclose(pc.gauher(20)[0], oc.gauher(20)[0])


@pytest.mark.parametrize('k1,k2,bal', [('matern32', 'matern52', False), ('matern32', 'matern52', True), ('exp', 'matern32', True),
                                       ('matern32', 'matern72', True), ('matern52', 'matern52', True), ('matern72', 'matern32', False)])
def test_blockwise_model_equals_dense_reference_construction(k1, k2, bal):
    pr = harness.nmf_problem(5, 2, 10, 11, kernel1=k1, kernel2=k2)
    F, L, Qc, H, Pinf = nagp.ss_modulators_nmf(pr['param1'], pr['param2'], k1, k2)
    Fo, Lo, Qo, Ho, Po = oss.ss_modulators_nmf(pr['param1'], pr['param2'], k1, k2)
    assert np.array_equal(F, Fo) and np.array_equal(H, Ho) and np.array_equal(Pinf, Po)
    assert np.allclose(L @ Qc @ L.T, Lo @ Qo @ Lo.T, rtol=0, atol=0)
    blk = _blocks_from_dense(F, L, Qc, H, Pinf, 5, 2)
    if bal:
        blk = pss.balance_blocks(blk)
        Fo, Lo, Ho, Po, _ = oss.balance_ss(Fo, Lo, Ho, Po)
    A, Q, P = pss.discretise(blk)
    Ao, Qo_ = oss.lti_disc(Fo, Lo, Qo, 1.0)
    assert np.allclose(A, Ao, rtol=1e-12, atol=1e-15) and np.allclose(Q, Qo_, rtol=1e-9, atol=1e-18)
    assert np.allclose(P, Po, rtol=1e-13) and np.allclose(blk.h_val, Ho[np.arange(7), blk.offsets[:-1]])


def test_ss_modulators_non_nmf_and_rejects_non_block_models():
    w = np.array([.1, .2, 30, 40, .5, .6, 2, 3, 300, 400.0])
    F, L, Qc, H, Pinf = nagp.ss_modulators(w, 'matern32', 'matern52')
    Fo, Lo, Qo, Ho, Po = oss.ss_modulators(w, 'matern32', 'matern52')
    assert np.array_equal(F, Fo) and np.array_equal(H, Ho)
    F[0, -1] = 1.0
    with pytest.raises(ValueError):
        _blocks_from_dense(F, L, Qc, H, Pinf, 2, 2)


def test_unpacking_and_input_merge():
    D, N = 4, 2
    pr = harness.nmf_problem(D, N, 5, 3, 'constraints')
    a = _unpack_log(pr['w'], 1, D, N); b = oss.unpack_log(pr['w'], 1, D, N)
    assert all(np.array_equal(x, y) for x, y in zip(a, b))
    cons = harness.CONSTRAINTS_DEMO(D)
    for tune in ([0, 0, 1, 0, 1, 1, 0], [1, 1, 1, 1, 1, 1, 1], [0, 0, 0, 0, 0, 0, 0], [1, 0, 0, 1, 0, 0, 1]):
        w, wf = harness.constrained_vectors(pr, cons, tune)
        a = _unpack_constraints(w, wf, tune, cons, 1, D, N); b = oss.unpack_constraints(w, wf, tune, cons, 1, D, N)
        assert all(np.allclose(x, y, rtol=1e-14) for x, y in zip(a, b))
        assert np.allclose(a[1], pr['param1'], rtol=1e-9) and np.allclose(a[3], pr['W'], rtol=1e-9)
    x = np.array([3.0, 1.0, 2.0]); y = np.array([30.0, 10.0, 20.0]); xt = np.array([2.0, 4.0, 1.0])
    ya, ra = _merge_inputs(x, y, xt); yo, ro = ogf.merge_inputs(x, y, xt)
    assert np.array_equal(ya, yo, equal_nan=True) and np.array_equal(ra, ro)
    assert np.array_equal(ya, [10, 20, 30, np.nan], equal_nan=True) and list(ra) == [1, 3, 0]


def test_stacked_source_models_equal_the_oracle_stacking():
    """experiments/gf_ep_mods_nmf_mixture.m:89-128: per-source kernels, sub-band blocks first, block-diagonal Wnmf."""
    mp = harness.mixture_problem([(2, 1), (3, 2), (1, 1)], 8, 11, ['exp', 'matern32', 'exp'], ['matern52', 'matern32', 'matern52'])
    blk, W, lik = _stack_sources(SSHandle(), None, mp['w'], mp['kernel1'], mp['kernel2'], 3)
    st = omx.stack_models(mp['w'], mp['kernel1'], mp['kernel2'], 3)
    F, LQL, H, Pinf = blk.dense()
    assert (blk.D, blk.N, blk.M) == (6, 4, 10) and np.array_equal(W, st['Wnmf']) and lik[0] == st['lik_param'][0]
    assert np.allclose(F, st['F'], rtol=1e-14, atol=0) and np.allclose(Pinf, st['Pinf'], rtol=1e-14, atol=0)
    assert np.allclose(LQL, st['L'] @ st['Qc'] @ st['L'].T, rtol=1e-14, atol=0) and np.array_equal(H, st['H'])
    A, Q, _ = pss.discretise(blk)
    Ao, Qo = oss.lti_disc(st['F'], st['L'], st['Qc'], 1.0)
    assert np.allclose(A, Ao, rtol=1e-11, atol=1e-13) and np.allclose(Q, Qo, rtol=1e-9, atol=1e-13)


def test_ihgp_tables_match_oracle_tables():
    pr = harness.nmf_problem(3, 2, 5, 9)
    lik, p1, p2, W = oss.unpack_log(pr['w'], 1, 3, 2)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', True, True)
    ilist, r, PPl, PGl = oih.build_tables(model)
    blk = pss.balance_blocks(pss.ss_blocks_nmf(p1, p2, 'matern32', 'matern52'))
    A, Q, P = pss.discretise(blk, symmetrize_Q=True)
    r2, PP, ppo, PG, pgo = ihgp_tables.build_tables(A, Q, blk.offsets, blk.h_val)
    assert np.allclose(r, r2)
    for n in range(5):
        b = blk.sizes[n]
        assert np.allclose(PP[ppo[n]:ppo[n] + 200 * b * b].reshape(200, -1), PPl[n], rtol=1e-7, atol=1e-12)
        assert np.allclose(PG[pgo[n]:pgo[n] + 400 * b * b].reshape(200, -1), PGl[n], rtol=1e-6, atol=1e-12)


@pytest.mark.parametrize('k1,tol_pp,tol_pg', [('matern52', 1e-6, 1e-5)])
def test_ihgp_tables_of_six_state_blocks_match_the_oracle_within_their_conditioning(k1, tol_pp, tol_pg):
    """Sub-band blocks of six states: the steady-state covariances are conditioned ~1e8, and the two DARE solvers (batched doubling on the host, SciPy's in
    the oracle) agree to 1e-8 .. 1e-6 -- both solve the equation to 1e-12 of the covariance's size (checked here for the host's tables)."""
    pr = harness.nmf_problem(3, 2, 5, 11, kernel1=k1)
    lik, p1, p2, W = oss.unpack_log(pr['w'], 1, 3, 2)
    model = ogf.assemble(lik, p1, p2, W, k1, 'matern52', True, True)
    ilist, r, PPl, PGl = oih.build_tables(model)
    blk = pss.balance_blocks(pss.ss_blocks_nmf(p1, p2, k1, 'matern52'))
    A, Q, P = pss.discretise(blk, symmetrize_Q=True)
    r2, PP, ppo, PG, pgo = ihgp_tables.build_tables(A, Q, blk.offsets, blk.h_val)
    assert np.allclose(r, r2) and list(blk.sizes[:3]) == [6, 6, 6]
    for n in range(5):
        b = blk.sizes[n]; o = blk.offsets[n]
        pp = PP[ppo[n]:ppo[n] + 200 * b * b].reshape(200, -1); pg = PG[pgo[n]:pgo[n] + 400 * b * b].reshape(200, -1)
        assert np.abs(pp - PPl[n]).max() < tol_pp * np.abs(PPl[n]).max() and np.abs(pg - PGl[n]).max() < tol_pg * np.abs(PGl[n]).max()
        Ab = A[o:o + b, o:o + b]; Qb = Q[o:o + b, o:o + b]; h = np.zeros((1, b)); h[0, 0] = blk.h_val[n]
        for g in (0, 199):      # residual of the predictive DARE  P = A (P - P h' (h P h' + r)^-1 h P) A' + Q  at the two grid points that are knots of the solver (the rest is interpolated)
            Pm = pp[g].reshape(b, b, order='F')
            K = Pm @ h.T / (h @ Pm @ h.T + r2[g])
            res = Ab @ (Pm - K @ h @ Pm) @ Ab.T + Qb - Pm
            assert np.abs(res).max() < 1e-11 * np.abs(Pm).max(), (n, g)


def test_mom_descriptor_validation():
    with pytest.raises(ValueError):
        Mom('likSomethingElse')
    with pytest.raises(ValueError):
        Mom('likModulatorPreCalcwn')
    wn, xn = Mom('likModulatorNMFPower', p_cubature=7).tables(3)
    assert wn.size == 45 and xn.shape == (3, 45)
    assert callable(SSHandle()) and len(SSHandle()(None, [.1, 30, .5], [2, 300], 'exp', 'matern32')) == 5


def test_library_builds_loads_and_exports_every_declared_symbol():
    """The C-ABI shared object cross-compiles without a GPU and exports exactly what include/nagp.h declares."""
    path = nagp.build()
    assert os.path.exists(path)
    hdr = open(os.path.join(ROOT, 'include', 'nagp.h')).read()
    declared = set(re.findall(r'^(?:int|void|int64_t|const char\*)\s+(nagp_[a-z0-9_]+)\s*\(', hdr, re.M))
    assert {'nagp_ep_run', 'nagp_ihgp_run', 'nagp_giekf_run', 'nagp_plan_create', 'nagp_plan_execute'} <= declared
    lib = ctypes.CDLL(path)
    for sym in declared:
        assert hasattr(lib, sym), sym
    L = nagp.lib()
    assert L.nagp_version() == 300
    assert 'ncclAllReduce' in subprocess.run(['nm', '-D', '--undefined-only', path], capture_output=True, text=True).stdout   # RCCL linked in
    assert L.nagp_strerror(-2).decode() == 'unsupported shape'
    out = subprocess.run(['nm', '-D', '--defined-only', path], capture_output=True, text=True).stdout
    assert set(re.findall(r' T (nagp_[a-z0-9_]+)', out)) == declared


def test_product_fails_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('a GPU is visible')
    pr = harness.nmf_problem(3, 2, 20, 1)
    t = np.arange(1, 21.0)
    with pytest.raises(nagp.NagpError):
        nagp.gf_ep_modulator_nmf(pr['w'], t, pr['y'], SSHandle(), Mom('likModulatorNMFPower', p_cubature=5), t, 'matern32', 'matern52',
                                 1, 3, 2, 0.5, [0.5], 1)


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'nonstationary-audio-gp_amd')
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(('.py', '.hip', '.hpp', '.h', '.cpp')):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r'^\s*(from|import)\s+oracle\b', src, re.M), f
                assert 'oracle/' not in src, f


def test_batch_partition_round_robin():
    """problem -> device map of nagp_batch_run (problem i -> GPU i mod G); pure host code of the C ABI."""
    assert list(nagp.batch_partition(8, 8)) == list(range(8))
    assert list(nagp.batch_partition(8, 2)) == [0, 1] * 4
    assert list(nagp.batch_partition(5, 3)) == [0, 1, 2, 0, 1]
    assert list(nagp.batch_partition(3, 1)) == [0, 0, 0]
    assert list(nagp.batch_partition(0, 4)) == []
    # the same map as the one-process-per-GPU layer
    from nagp import dist as nd
    for n, g in ((8, 2), (5, 3), (7, 4)):
        dev = nagp.batch_partition(n, g)
        for r in range(g):
            assert [i for i in range(n) if dev[i] == r] == nd.shard(n, r, g)
    with pytest.raises(nagp.NagpError):
        nagp.batch_partition(4, 0)


@pytest.mark.parametrize('G', [2, 3, 5, 8])
def test_batch_run_host_logic_for_several_devices_without_the_hardware(G, monkeypatch):
    """nagp_batch_run for G = 2 .. 8 devices on a machine with none (test hook NAGP_TEST_FAKE_DEVICES): the partition, one host
    thread per device, every worker's own host-side validation and packing up to its first device call, and the propagation of the
    first failing device's status and text.  With NAGP_TEST_FAIL_DEVICE the injected failure of that device is what comes back when
    it is the first in device order, and the call returns (no worker left behind) whichever device fails."""
    from nagp import harness, Mom, ss as pss, _lib as L
    if nagp.lib().nagp_device_count() > 0:
        pytest.skip('a GPU is visible: the GPU suite covers this path with real plans')
    monkeypatch.setenv('NAGP_TEST_FAKE_DEVICES', str(G))
    D, N, T = 3, 2, 12
    probs, ys = [], []
    for q in range(11):
        pr = harness.nmf_problem(D, N, T, 40 + q)
        probs.append((pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'), pr['W'], np.log(pr['w_lik']))); ys.append(pr['y'])
    kw = dict(mom=Mom('likModulatorNMFPower', p_cubature=5), ep_fraction=0.5, ep_damping=0.5 * np.ones(2), ep_itts=2)
    with pytest.raises(nagp.NagpError, match=r'no HIP device.*device 0: '):
        nagp.batch_run(L.KIND_GF_EP, probs, ys, T, n_gpus=G, **kw)
    with pytest.raises(nagp.NagpError, match='n_gpus'):
        nagp.batch_run(L.KIND_GF_EP, probs, ys, T, n_gpus=G + 1, **kw)          # more than the (fake) devices
    monkeypatch.setenv('NAGP_TEST_FAIL_DEVICE', '0')
    with pytest.raises(nagp.NagpError, match=r'HIP runtime error.*device 0: injected failure'):
        nagp.batch_run(L.KIND_GF_EP, probs, ys, T, n_gpus=G, **kw)
    monkeypatch.setenv('NAGP_TEST_FAIL_DEVICE', str(G - 1))                     # the last device fails differently: device 0's status still wins
    with pytest.raises(nagp.NagpError, match=r'no HIP device.*device 0: '):
        nagp.batch_run(L.KIND_GF_EP, probs, ys, T, n_gpus=G, **kw)
    monkeypatch.delenv('NAGP_TEST_FAIL_DEVICE')


def test_developer_switches_and_test_hooks_are_ignored_without_nagp_developer():
    """A stray NAGP_TEST_FAKE_DEVICES in the environment of a host process must not map 'devices' onto nothing: without NAGP_DEVELOPER=1
    the library reads none of its switches (here: on a machine without a GPU the fake devices do not exist and the call is refused with
    'no HIP device visible'; with NAGP_DEVELOPER=1 the same call gets as far as the workers' first device call)."""
    if nagp.lib().nagp_device_count() > 0:
        pytest.skip('a GPU is visible')
    code = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'nonstationary-audio-gp_amd'))
import numpy as np, nagp
from nagp import harness, Mom, ss as pss, _lib as L
pr = harness.nmf_problem(3, 2, 12, 40)
probs = [(pss.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52'), pr['W'], np.log(pr['w_lik']))] * 4
try:
    nagp.batch_run(L.KIND_GF_EP, probs, [pr['y']] * 4, 12, n_gpus=2, mom=Mom('likModulatorNMFPower', p_cubature=5), ep_fraction=0.5, ep_damping=0.5 * np.ones(2), ep_itts=2)
except nagp.NagpError as e:
    print('ERR', e)
"""
    outs = {}
    for dev in ('0', '1'):
        env = dict(os.environ, NAGP_TEST_FAKE_DEVICES='4'); env.pop('NAGP_DEVELOPER', None)
        if dev == '1':
            env['NAGP_DEVELOPER'] = '1'
        r = subprocess.run([sys.executable, '-c', code, ROOT], env=env, capture_output=True, text=True, timeout=300)
        outs[dev] = r.stdout + r.stderr
    assert 'no HIP device visible' in outs['0'] and 'device 0: ' not in outs['0'] and 'developer switch' not in outs['0'], outs['0']
    assert 'device 0: ' in outs['1'] and 'developer switch NAGP_TEST_FAKE_DEVICES=4 is active' in outs['1'], outs['1']


def test_measmodel_handle_raises_the_documented_error():
    H = np.zeros((5, 7)); H[np.arange(5), [0, 1, 2, 3, 5]] = 1.0
    mm = nagp.MeasModel(H, np.ones((3, 2)), 3, 2)
    with pytest.raises(nagp.NagpError):
        mm.h(np.zeros(7))
    with pytest.raises(nagp.NagpError):
        mm.dh(np.zeros(7))


_WORKER = r"""
import os, sys
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, os.path.join(sys.argv[1], 'nonstationary-audio-gp_amd'))
import numpy as np
from nagp import dist as nd, harness
from oracle import gf_ep as ogf, lik as olik, ss as oss
rank, lr, world = nd.init('gloo')
NSEG, I, D, N, T = 5, 2, 3, 2, 30
def seg_nlz(q):      # what a rank's Plan returns for segment q: the per-sweep nlZ of gf_ep_modulator_nmf (here from the oracle)
    pr = harness.nmf_problem(D, N, T, 700 + q)
    lik, p1, p2, W = oss.unpack_log(pr['w'], 1, D, N)
    model = ogf.assemble(lik, p1, p2, W, 'matern32', 'matern52', False)
    return ogf.run_predict(model, pr['y'], olik.Mom(olik.LIK_POWER_NMF, p=5), 0.5, 0.5 * np.ones(I), I)['nlZ']
mine = nd.shard(NSEG, rank, world)
part = np.array([seg_nlz(q) for q in mine]).reshape(len(mine), I)      # (n_local_problems, ep_itts), as Plan.download_nlz()
tot = nd.allreduce_nlz(part)
mx = nd.allreduce_max(float(rank + 1))
nd.barrier()
exp = np.array([seg_nlz(q) for q in range(NSEG)]).sum(axis=0)         # serial sum over all segments
assert tot.shape == (I,) and np.allclose(tot, exp, rtol=1e-13), (tot, exp)
assert mx == world and nd.world_size() == world
assert sorted(sum([nd.shard(NSEG, r, world) for r in range(world)], [])) == list(range(NSEG))
# a rank without segments (more ranks than segments) still takes part in the reduction
none = nd.allreduce_nlz(np.zeros((0, I)).reshape(-1, I) if rank == 1 else np.ones((1, I)))
assert np.allclose(none, 1.0)
print('rank', rank, 'ok', flush=True)
nd.finalize()
"""


def test_two_rank_gloo_sharding_and_nlz_allreduce(tmp_path):
    """N>1 path on CPU: segments sharded round-robin over ranks, the per-sweep nlZ of every rank's segments (real
    gf_ep_modulator_nmf values, here produced by the oracle) all-reduced and compared with the serial sum (gloo stands in for RCCL)."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER)
    from nagp import dist as nd
    env = dict(os.environ, **nd.file_rendezvous_env(str(tmp_path), 2))      # file store: no port to race for
    procs = [subprocess.Popen([sys.executable, str(script), ROOT], env=dict(env, RANK=str(r), LOCAL_RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=180)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs


def test_bench_refuses_a_world_size_mismatch_and_spawns_ranks_itself():
    """bench.py --gpus N: without RANK it is the launcher (N children); with RANK set and another WORLD_SIZE it refuses."""
    src = open(os.path.join(ROOT, 'bench.py')).read()
    assert "if a.gpus > 1 and 'RANK' not in os.environ:" in src and 'spawn_ranks' in src
    # the launcher decision comes before anything that could touch the GPU
    assert src.index("sys.exit(spawn_ranks(a))") < src.index('import torch\n    import nagp')
    env = dict(os.environ, RANK='0', LOCAL_RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29577')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--T', '50'], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and '--gpus 2 but WORLD_SIZE=1' in (r.stderr + r.stdout)


def test_c_callers_and_the_mex_gateway_compile_and_link():
    """tests/c/abi_golden.c (plain C caller of the ABI) and matlab/nagp_mex.c (against the mock MEX API of tests/c) build with
    gcc and link against libnagp.so; they run in the -m gpu tests."""
    import tempfile
    pkg = os.path.join(ROOT, 'nonstationary-audio-gp_amd'); c = os.path.join(ROOT, 'tests', 'c')
    nagp.build()
    with tempfile.TemporaryDirectory() as td:
        base = ['gcc', '-Wall', '-Werror', '-O1', '-std=c99', '-I', os.path.join(ROOT, 'include'), '-I', c]
        link = ['-L', pkg, '-lnagp', '-lm', '-Wl,-rpath,' + pkg, '-Wl,-rpath,/opt/rocm/lib', '-Wl,-rpath-link,/opt/rocm/lib']
        for out, srcs in (('abi_golden', ['abi_golden.c']), ('mex_driver', ['mex_driver.c', 'mex_mock.c', os.path.join(ROOT, 'matlab', 'nagp_mex.c')])):
            r = subprocess.run(base + ['-o', os.path.join(td, out)] + [os.path.join(c, f) for f in srcs] + link, capture_output=True, text=True)
            assert r.returncode == 0, r.stderr
    # every reference entry point has its MATLAB wrapper, with the reference's argument list
    sigs = {'gf_ep_modulator': 'w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,ep_fraction,ep_damping,ep_itts',
            'gf_ep_modulator_nmf': 'w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts',
            'gf_ep_modulator_nmf_constraints': 'w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts,constraints,w_fixed,tune_hypers',
            'ihgp_ep_modulator_nmf': 'w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts',
            'ihgp_ep_modulator_nmf_constraints': 'w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,ep_fraction,ep_damping,ep_itts,constraints,w_fixed,tune_hypers',
            'gf_giekf_modulator_nmf': 'w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,g_iter,l_iter,GradObj',
            'gf_giekf_modulator_nmf_constraints': 'w,x,y,ss,mom,xt,kernel1,kernel2,num_lik_params,D,N,g_iter,l_iter,constraints,w_fixed,tune_hypers,GradObj'}
    sigs.update({'gf_ep_mods_nmf_mixture': 'w,x,y,ss,mom,xt,kernel1,kernel2,J,ep_fraction,ep_damping,ep_itts',
                 'ihgp_ep_mods_nmf_mixture': 'w,x,y,ss,mom,xt,kernel1,kernel2,J,ep_fraction,ep_damping,ep_itts'})
    for fn, args in sigs.items():
        src = open(os.path.join(ROOT, 'matlab', fn + '.m')).read()
        m = re.match(r'function \[varargout\] = (\w+)\(([^)]*)\)', re.sub(r'\.\.\.\s*', '', src))
        assert m and m.group(1) == fn and re.sub(r'\s', '', m.group(2)) == args, fn
    # the entry points with named outputs: first lines equal to the reference's (ekf_update1.m:48, iekf_update1.m:48, kernel_ss_kalmanFastFB.m:1)
    for fn, line in (('ekf_update1', 'function [M,P,K,MU,S,LH] = ekf_update1(M,P,y,H,R,h,V,param)'),
                     ('iekf_update1', 'function [M,P,K,MU,S,LH] = iekf_update1(M,P,y,H,R,h,V,param,iters)'),
                     ('kernel_ss_kalmanFastFB', 'function [lik,Xfin,Pfin,varargout] = kernel_ss_kalmanFastFB(A,Q,C,P0,K,vary,y,varargin)')):
        assert open(os.path.join(ROOT, 'matlab', fn + '.m')).readline().strip() == line, fn
    # every command string the wrappers send exists in the gateway
    gw = open(os.path.join(ROOT, 'matlab', 'nagp_mex.c')).read()
    for f in os.listdir(os.path.join(ROOT, 'matlab')):
        if f.endswith('.m'):
            for cmd in re.findall(r"nagp_mex\('(\w+)'", open(os.path.join(ROOT, 'matlab', f)).read()):
                assert '!strcmp(cmd, "%s")' % cmd in gw, (f, cmd)


def test_abi_error_paths_under_asan():
    """The host code of the C ABI (nagp_api.hip) built with AddressSanitizer; tests/c/abi_errors.c walks the
    argument-validation paths (NULL pointers, bad shapes, mismatching problems) on this GPU-less machine: every call comes
    back with its status and ASan reports nothing."""
    import tempfile, glob
    lib = nagp.build(asan=True)
    clang = '/opt/rocm/lib/llvm/bin/clang'
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux') + glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/x86_64-unknown-linux-gnu')
    with tempfile.TemporaryDirectory() as td:
        exe = os.path.join(td, 'abi_errors')
        r = subprocess.run([clang, '-fsanitize=address', '-shared-libsan', '-g', '-O1', '-std=c99', '-I', os.path.join(ROOT, 'include'), '-o', exe,
                            os.path.join(ROOT, 'tests', 'c', 'abi_errors.c'), lib, '-Wl,-rpath,/opt/rocm/lib', '-Wl,-rpath-link,/opt/rocm/lib'] +
                           ['-Wl,-rpath,' + d for d in rt], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        env = dict(os.environ, ASAN_OPTIONS='detect_leaks=0:abort_on_error=0:exitcode=66', LD_LIBRARY_PATH=':'.join(rt + [os.environ.get('LD_LIBRARY_PATH', '')]))
        r = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and 'all error paths returned their status' in r.stdout, r.stdout + r.stderr
        assert 'AddressSanitizer' not in r.stderr, r.stderr
