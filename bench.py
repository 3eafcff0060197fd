#!/usr/bin/env python3
"""bench.py -- throughput of the Kalman/RTS/Power-EP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5|cfg3_batch|cfg2_batch|cfg5_fill] [--segments B] [--T T] [--extras ...]

A "step" is one full pass of the hot path (all EP sweeps: forward filter, smoother, site refresh) over one batch of
synthetic audio segments already resident in HBM.

The top-level line is BASELINE.json configs[2], the configuration north_star states its target on:
ihgp_ep_modulator_nmf, 200 000 samples, 32 channels / 6 NMF components -- ONE segment per GPU ("weak": the sequence is a
single recursion and does not shard; with N GPUs every rank filters its own 200 k-sample segment).  The same JSON line
carries, under `cfg5_strong`, the segment-sharded workload of configs[4] (gf_ep_modulator_nmf_constraints, 8 segments of
100 000 samples IN TOTAL, 8/N per rank, RCCL all-reduce of the per-sweep log marginal likelihood inside the timed region)
and, at N = 1, configs[1] under `cfg2`.

`--gpus N` with N > 1 and no RANK in the environment starts N fresh child processes (one per GPU) BEFORE anything in this
process touches the GPU; under `python -m torch.distributed.run` (RANK set) the process is one rank.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: function, D, N, T, cubature order, parameter recipe, balance, segments IN TOTAL (None: `per_gpu` segments on every GPU)
    'cfg2': dict(fn='gf_ep_modulator_nmf', D=16, N=3, T=84010, p=9, recipe='demo_nmf', balance=False, total_segments=None, audio='speech_74',
                 damping=0.1),      # the damping of the reference's speech drivers (experiments/noise_reduction_speech.m:29); at 0.5 the reference algorithm itself is chaotic on this file (tests/golden/audio_conditioning.json)
    'cfg3': dict(fn='ihgp_ep_modulator_nmf', D=32, N=6, T=200000, p=7, recipe='constraints', balance=True, total_segments=None),
    'cfg4': dict(fn='gf_giekf_modulator_nmf', D=24, N=3, T=88200, p=9, recipe='demo_nmf', balance=True, total_segments=None, audio='stim312_wind'),
    'cfg5': dict(fn='gf_ep_modulator_nmf_constraints', D=32, N=6, T=100000, p=7, recipe='constraints', balance=True, total_segments=8),
    # the fill-the-chip regime of the same three kernels families (the batch axes of the path: segments / hyper-parameter replicas):
    # every GPU runs `per_gpu` independent segments, so the series is weak-scaling by construction
    'cfg3_batch': dict(fn='ihgp_ep_modulator_nmf', D=32, N=6, T=4000, p=7, recipe='constraints', balance=True, total_segments=None, per_gpu=256),
    'cfg2_batch': dict(fn='gf_ep_modulator_nmf', D=16, N=3, T=4000, p=9, recipe='demo_nmf', balance=False, total_segments=None, per_gpu=128),
    'cfg5_fill': dict(fn='gf_ep_modulator_nmf_constraints', D=32, N=6, T=12500, p=7, recipe='constraints', balance=True, total_segments=None, per_gpu=32),
    # cfg3 with the likelihood every paper experiment uses (experiments/likModulatorPreCalcwn.m: amplitudes sqrt(W softplus(g - 1)),
    # train_model.m:38,55, noise_reduction_speech.m:41) on the same rule (ut7, 305 points, passed in precomputed as the drivers do)
    'cfg2_sqrt': dict(fn='gf_ep_modulator_nmf', D=16, N=3, T=84010, p=9, recipe='demo_nmf', balance=False, total_segments=None,
                      lik='likModulatorPreCalcwn', link_shift=1.0, damping=0.1),     # (not in the default line: the gf ADF launches with the sqrt likelihood)
    'cfg3_sqrt': dict(fn='ihgp_ep_modulator_nmf', D=32, N=6, T=200000, p=7, recipe='constraints', balance=True, total_segments=None,
                      lik='likModulatorPreCalcwn', link_shift=1.0, damping=0.1),     # damping of the drivers that use this likelihood (noise_reduction_speech.m:29: <= 0.1); at 0.5 the reference algorithm's own sites reach 1e14
}
EP_ITTS = 3
PEAK_FP64_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (MI355X_MICROARCH.md / SURVEY App. E)
PEAK_HBM_GBS = 8000.0


def named_audio(name, T):
    """The decoded samples of the audio file a BASELINE configuration names (tests/golden/audio_*.npz, tools/make_audio_fixtures.py), as the
    reference's drivers prepare them: int16 / 32768, divided by the standard deviation.  None when T is not the file's length."""
    try:
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'audio_%s.npz' % name))
    except OSError:
        return None
    x = z['samples'].astype(np.float64) / 32768.0
    return x / np.std(x) if x.size == T else None


def build_problems(wl, seeds):
    from nagp import harness
    from nagp import ss as ssm
    probs, ys = [], []
    for sd in seeds:
        ya = named_audio(wl['audio'], wl['T']) if wl.get('audio') else None
        pr = harness.nmf_problem(wl['D'], wl['N'], wl['T'] if ya is None else 8, sd, wl['recipe'],      # (--T overrides: a prior sample of that length)
                                 link_shift=wl.get('link_shift', 0.0), sqrt_amp=wl.get('lik') == 'likModulatorPreCalcwn')     # (data from the model that is inferred)
        if ya is not None:
            pr['y'] = ya
        blk = ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
        if wl['balance']:
            blk = ssm.balance_blocks(blk)
        probs.append((blk, pr['W'], np.log(pr['w_lik'])))
        ys.append(pr['y'])
    return probs, ys


def make_mom(wl):
    from nagp import Mom, cubature
    if wl.get('lik', 'likModulatorNMFPower') == 'likModulatorPreCalcwn':
        wn, xn = cubature.sigma_points(wl['p'], wl['N'], True)
        return Mom('likModulatorPreCalcwn', link_shift=wl.get('link_shift', 0.0), wn=wn, xn_unscaled=xn)
    return Mom('likModulatorNMFPower', p_cubature=wl['p'])


def usable_cores():
    """cores this process may use: the affinity mask, cut by the cgroup CPU quota when there is one"""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            q, per = fh.read().split()
        if q != 'max':
            n = max(1, min(n, int(float(q) / float(per))))
    except (OSError, ValueError):
        pass
    return n


def host_cpu():
    model = 'unknown'
    try:
        with open('/proc/cpuinfo') as fh:
            for ln in fh:
                if ln.lower().startswith('model name'):
                    model = ln.split(':', 1)[1].strip()
                    break
    except OSError:
        pass
    return model, os.cpu_count() or 1, usable_cores()


def cpu_baseline(wl, budget_s=4.0):
    """The reference algorithm on this host's cores (MATLAB / Octave do not exist here: kind "port").  Timed, on a bounded prefix of the
    same workload: the COMPILED restatement oracle/cpu/nagp_cpu.cpp (g++ -O3 -march=native, plain loops, checked against the oracle on
    the golden vectors) -- (i) dense as written, one thread; (ii) structured (block-diagonal A, selection H: what a careful CPU port
    does), one thread; (iii) structured, one independent segment per core on all usable cores (OpenMP; the path's own parallel axis) --
    and (iv) the NumPy oracle, one thread, for continuity with the earlier rounds.  `value` is (iii), the strongest of them."""
    from nagp import harness
    from oracle import cpu as ocpu, gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik, ss as oss
    D, N = wl['D'], wl['N']
    fam = 'ihgp' if wl['fn'].startswith('ihgp') else ('giekf' if wl['fn'].startswith('gf_giekf') else 'gf')
    om = None if fam == 'giekf' else olik.Mom(olik.LIK_POWER_NMF, p=wl['p'])
    d = 0.5 * np.ones(EP_ITTS)
    model_name, nproc, usable = host_cpu()
    Tmax = 20000 if fam == 'ihgp' else 4000
    pr = harness.nmf_problem(D, N, Tmax, 1000, wl['recipe'])
    lik_param, p1, p2, W = oss.unpack_log(pr['w'], 1, D, N)
    model = ogf.assemble(lik_param, p1, p2, W, 'matern32', 'matern52', wl['balance'] or fam != 'gf', fam == 'ihgp')
    tabs = oih.build_tables(model) if fam == 'ihgp' else None          # DARE tables are set-up, not the timed loop

    def one(Ts, structured):
        t0 = time.perf_counter()
        if fam == 'ihgp':
            r = ocpu.ihgp_predict(model, pr['y'][:Ts], om, 0.5, d, EP_ITTS, D, N, tabs, structured=structured)
        elif fam == 'giekf':
            r = ocpu.giekf_predict(model, pr['y'][:Ts], D, N, EP_ITTS, 1, structured=structured)
        else:
            r = ocpu.gf_predict(model, pr['y'][:Ts], om, 0.5, d, EP_ITTS, D, N, structured=structured)
        dt = time.perf_counter() - t0
        assert r['status'] == 0 and np.all(np.isfinite(r['Eft']))
        return dt

    def calibrated(structured, target_s):
        Ts = 50
        dt = one(Ts, structured)
        Ts2 = int(min(Tmax, max(Ts, Ts * target_s / max(dt, 1e-4))))
        if Ts2 > 2 * Ts:
            Ts, dt = Ts2, one(Ts2, structured)
        return Ts, dt, Ts * EP_ITTS / dt

    ocpu.build()
    Td, dtd, rate_dense = calibrated(False, budget_s * 0.5)
    Tst, dts, rate_struct = calibrated(True, budget_s * 0.7)
    threads = max(1, min(usable, 128))
    rate_all, dt_all = rate_struct, dts
    if threads > 1:
        ys = [harness.nmf_problem(D, N, Tst, 1000 + q, wl['recipe'])['y'] for q in range(min(threads, 8))]
        ys = [ys[q % len(ys)] for q in range(threads)]                   # the timing does not depend on the numbers
        t0 = time.perf_counter()
        _, st = ocpu.segments(fam, model, ys, om, 0.5, d, EP_ITTS, D, N, tables=tabs, threads=threads, structured=True)
        dt_all = time.perf_counter() - t0
        assert st == 0
        rate_all = threads * Tst * EP_ITTS / dt_all
    # the NumPy oracle (interpreter + BLAS), one thread, ~2 s
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    Tn = int(max(50, min(Tmax, 1.0 * rate_dense / EP_ITTS)))            # a second or two of interpreter time
    with ctx:
        t0 = time.perf_counter()
        if fam == 'ihgp':
            oih.run_predict(model, pr['y'][:Tn], om, 0.5, d, EP_ITTS, tables=tabs)
        elif fam == 'giekf':
            oek.run_predict(model, pr['y'][:Tn], D, N, EP_ITTS, 1)
        else:
            ogf.run_predict(model, pr['y'][:Tn], om, 0.5, d, EP_ITTS)
        dtn = time.perf_counter() - t0
    return dict(value=rate_all, unit='samples/s', cores=threads, kind='port',
                single_thread_value=rate_struct, dense_as_written_single_thread_value=rate_dense, numpy_oracle_single_thread_value=Tn * EP_ITTS / dtn,
                host_cpu_model=model_name, host_nproc=nproc, host_usable_cores=usable,
                sample='compiled restatement oracle/cpu/nagp_cpu.cpp (g++ -O3 -march=native, plain loops; MATLAB/Octave absent) on a prefix of the same '
                       'workload x %d sweeps: dense as written, 1 thread: first %d samples %.1f s; structured, 1 thread: first %d samples %.1f s; structured, '
                       '%d threads with one independent %d-sample segment each: %.1f s wall (= value); NumPy oracle, 1 thread: first %d samples %.1f s'
                       % (EP_ITTS, Td, dtd, Tst, dts, threads, Tst, dt_all, Tn, dtn))


def sig(x, n=5):
    """n significant digits (the extras of the JSON line)"""
    try:
        return float('%.*g' % (n, float(x)))
    except (TypeError, ValueError):
        return x


def source_hash():
    from nagp import _lib as L
    return L.source_hash()


def run_workload(name, a, rank, local_rank, world, dev, with_cpu, steps, warmup):
    """Times `steps` executes of one named workload; returns the JSON-able result dict."""
    import nagp
    from nagp import Mom, _lib as L, dist as nd
    import torch
    wl = dict(WORKLOADS[name])
    if a.T:
        wl['T'] = a.T
    if a.segments:                           # diagnostics: B segments on every GPU
        seeds = [1000 + 100 * rank + q for q in range(a.segments)]
        n_total = a.segments * world; scaling = 'weak'
    elif wl['total_segments']:               # a fixed set of segments sharded over the ranks (round robin)
        seeds = [5000 + q for q in nd.shard(wl['total_segments'], rank, world)]
        n_total = wl['total_segments']; scaling = 'strong'
    else:                                    # `per_gpu` segments on every GPU (1 for the BASELINE configurations)
        per_gpu = wl.get('per_gpu', 1)
        seeds = [1000 + 100 * rank + (q % 8) for q in range(per_gpu)]     # eight distinct prior samples, cycled: generating 256 on the host takes longer than the bench
        n_total = per_gpu * world; scaling = 'weak'
    n_seg = len(seeds)
    kind = {'gf_ep': L.KIND_GF_EP, 'ihgp_': L.KIND_IHGP, 'gf_gi': L.KIND_GIEKF}[wl['fn'][:5]]
    mom = None if kind == L.KIND_GIEKF else make_mom(wl)
    plan = None
    if n_seg:
        uniq = sorted(set(seeds))
        up, uy = build_problems(wl, uniq)
        probs = [up[uniq.index(sd)] for sd in seeds]; ys = [uy[uniq.index(sd)] for sd in seeds]
        plan = nagp.Plan(kind, probs, wl['T'], mom=mom, ep_fraction=0.5, ep_damping=wl.get('damping', 0.5) * np.ones(EP_ITTS), ep_itts=EP_ITTS,
                         l_iter=1, device=local_rank)
        plan.upload(ys)                                           # inputs resident in HBM before timing

    def step():
        if plan is not None:
            plan.execute()                                        # all sweeps, synchronous on the plan's stream
            part = plan.download_nlz()
        else:
            part = np.zeros((1, EP_ITTS))
        return nd.allreduce_nlz(part, dev)                        # RCCL all-reduce of nlZ[itt] (8*I bytes)

    for _ in range(warmup):
        step()
    nd.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern = {k: 0.0 for k in L.KERNEL_NAMES}; launches = {k: 0 for k in L.KERNEL_NAMES}
    nlz_total = None
    for _ in range(steps):
        nlz_total = step()
        if plan is not None:
            tm = plan.timings()
            for k in kern:
                kern[k] += tm['ms'][k]; launches[k] += tm['launches'][k]
    torch.cuda.synchronize(); nd.barrier()
    dt = nd.allreduce_max(time.perf_counter() - t0, dev)
    # (outside the timed region) the all-reduced log marginal likelihoods are finite and are the sum of the per-problem values: the ranks'
    # own sums, gathered on a second path, add up to what the all-reduce returned
    part = plan.download_nlz() if plan is not None else np.zeros((1, EP_ITTS))
    per_rank = nd.allgather_sums(part, dev)
    nlz_ok = bool(np.all(np.isfinite(nlz_total)) and np.all(np.isfinite(part)) and
                  np.allclose(per_rank.sum(axis=0), nlz_total, rtol=1e-12, atol=0.0) if kind != L.KIND_GIEKF else np.all(np.isfinite(nlz_total)))
    if not nlz_ok:
        raise SystemExit('bench.py: %s: nlZ_allreduced %s is not finite / not the sum of the per-problem values %s' % (name, nlz_total, per_rank.sum(axis=0)))

    D, N, T = wl['D'], wl['N'], wl['T']
    S = plan.S if plan else 0; M = plan.M if plan else 0
    samples_per_step = n_total * T
    value = samples_per_step * EP_ITTS * steps / dt
    n_pts = mom.tables(N)[0].size if mom is not None else 0
    f_mom = n_pts * (2 * N * D + 12 * D + 8 * N + 10)
    roof = None
    if plan is not None and kind == L.KIND_IHGP:
        dom = 'filter'
        per_sample = 8.0 * (3 * S + 9 * M + 1)                    # IHGP algorithmic bytes / sample / sweep
        units = n_seg * T * steps                                 # one ADF-sweep launch covers T samples of every segment
        per_launch = per_sample * n_seg * T
        achieved = per_sample * units / (kern[dom] * 1e-3) / 1e9
        roof = dict(bound='hbm', kernel='ihgp ADF sweep (ihgp_adf8_kernel / ihgp_adf8sq_kernel / ihgp_adf_kernel / ihgp_filter_kernel, one launch per execute)',
                    achieved=achieved, peak=PEAK_HBM_GBS, unit='GB/s', frac=achieved / PEAK_HBM_GBS, traffic=None,
                    algorithmic_bytes_per_sample=per_sample, algorithmic_bytes_per_launch=per_launch,
                    valu_gflops=(8.0 * S * (S / M) + 8 * S + f_mom) * units / (kern[dom] * 1e-3) / 1e9,
                    valu_peak_gflops=PEAK_FP64_TFLOPS * 1e3, valu_frac=(8.0 * S * (S / M) + 8 * S + f_mom) * units / (kern[dom] * 1e-3) / 1e9 / (PEAK_FP64_TFLOPS * 1e3),
                    workgroups=n_seg, cus_occupied=min(n_seg, 256),
                    # what the ADF launch itself moves (MF, five site / marginal arrays, y, lZ); SURVEY's figure above counts the whole sweep
                    adf_kernel_bytes_per_sample=8.0 * (S + 5 * M + 2),
                    avg_launch_ms=kern[dom] / max(launches[dom], 1), us_per_sample=kern[dom] * 1e3 / units)
    elif plan is not None:
        # gf_ep: the ADF launches (sweep 1: all T steps, later sweeps: the step k = T-1) are one kernel
        # (gf_filter_kernel<.., MV, 256>, timing slot 'filter'); the fixed-site steps of sweeps >= 2 run in the
        # mom-free instantiation (slot 'filter_lin').  giekf: every sweep is the EKF instantiation (slot 'filter').
        dom = 'filter'
        bbar = S / M
        per_step = (4 * bbar + 1) * S * S + 2.0 * M * S * S       # block-diagonal A P A' + Q, rank-M update
        if kind == L.KIND_GF_EP:
            adf_steps = n_seg * (T + (EP_ITTS - 1)) * steps       # steps the ADF kernel processed (3 launches per execute)
            lin_steps = n_seg * (T - 1) * (EP_ITTS - 1) * steps
            flops = (per_step + f_mom) * adf_steps
        else:
            per_step = (4 * bbar + 1) * S * S + 6.0 * S * S       # EKF: l_iter*4S^2 + 2S^2 instead of the rank-M update
            adf_steps = n_seg * T * EP_ITTS * steps; lin_steps = 0
            flops = per_step * adf_steps
        achieved = flops / (kern[dom] * 1e-3) / 1e12
        sm_ms = (kern['scan'] + kern['gain']) * 1e-3
        sm_steps = n_seg * (T - 1) * EP_ITTS * steps
        roof = dict(bound='fp64', kernel='gf_filter_kernel (ADF / EKF launches; VALU FP64, no MFMA issued)', achieved=achieved, peak=PEAK_FP64_TFLOPS,
                    unit='TFLOP/s', frac=achieved / PEAK_FP64_TFLOPS, traffic=None,
                    algorithmic_flops_per_step=per_step + (f_mom if kind == L.KIND_GF_EP else 0),
                    avg_launch_ms=kern[dom] / max(launches[dom], 1), launches_per_execute=launches[dom] / steps,
                    # a segment is one sequential recursion on one workgroup = one CU of 256: the per-CU ceiling it can reach
                    peak_one_cu=PEAK_FP64_TFLOPS / 256, frac_one_cu=achieved / (PEAK_FP64_TFLOPS / 256 * n_seg),
                    smoother_tflops_algorithmic=(19.0 / 3.0 * S ** 3) * sm_steps / sm_ms / 1e12,   # chol + 2 trsm + G dP G'
                    smoother_tflops_executed=(10.0 * S ** 3) * sm_steps / sm_ms / 1e12)            # parallel-in-time form: 2.5x the sequential count
        if lin_steps and kern['filter_lin'] > 0:
            roof['fixed_site_kernel'] = dict(kernel='gf_filter_kernel<.., MV=-1> (sweeps >= 2, k < T-1)',
                                             achieved=per_step * lin_steps / (kern['filter_lin'] * 1e-3) / 1e12, unit='TFLOP/s',
                                             avg_launch_ms=kern['filter_lin'] / max(launches['filter_lin'], 1),
                                             us_per_sample=kern['filter_lin'] * 1e3 / lin_steps)
        # whole call against the FP64 peak with SURVEY 8(d)'s algorithmic flops per sample and sweep:
        # F_gf = 19/3 S^3 (chol + 2 trsm + G dP G') + (10 b + 2M + 6) S^2 + c_mom F_mom   (giekf: no mom, 6 S^2 in place of 2M S^2)
        if kind == L.KIND_GF_EP:
            f_gf = 19.0 / 3.0 * S ** 3 + (10 * bbar + 2 * M + 6) * S * S + (4.0 / 3.0) * f_mom      # c_mom: 2 on sweep 1, 1 on sweep 2, 1 (filter only) on sweep 3
        else:
            f_gf = 19.0 / 3.0 * S ** 3 + (10 * bbar + 6 + 6) * S * S
        roof['whole_call'] = dict(bound='fp64', algorithmic_flops_per_sample_per_sweep=f_gf,
                                  achieved=f_gf * n_seg * T * EP_ITTS * steps / dt / 1e12, peak=PEAK_FP64_TFLOPS, unit='TFLOP/s',
                                  frac=f_gf * n_seg * T * EP_ITTS * steps / dt / 1e12 / PEAK_FP64_TFLOPS,
                                  note='this rank; wall time of the timed region (all kernels, both streams, host round trips)')
        roof['adf_us_per_sample'] = kern[dom] * 1e3 / adf_steps
        roof['algorithmic_bytes_per_launch'] = 8.0 * (8 * M * (M + 1) + S + 5 * M + 2) * adf_steps / max(launches[dom], 1)   # lower tiles + means + sites
    if roof is not None:
        # per-kernel table of THIS execute: kernel time (HIP events on the kernel's own stream; kernels of the two streams overlap, so the
        # shares are of the summed kernel time, not of the wall time), algorithmic work and the fraction of the chip's peak
        sm_steps_k = n_seg * (T - 1) * EP_ITTS * steps
        if kind == L.KIND_IHGP:
            work = {'filter': ('hbm', 8.0 * (S + 5 * M + 2) * n_seg * T * steps, 'ADF sweep: MF, five site / marginal arrays, y, lZ'),
                    'filter_lin': ('hbm', 8.0 * (2 * S + 4 * M) * n_seg * T * (EP_ITTS - 1) * steps, 'fixed-site scans of sweeps >= 2'),
                    'scan': ('hbm', 8.0 * (2 * S + 3 * M) * sm_steps_k, 'backward mean scans'),
                    'epsite': ('fp64', float(f_mom) * n_seg * (T - 1) * (EP_ITTS - 1) * steps, 'site refresh (cubature)')}
        else:
            work = {'filter': ('fp64', flops, 'ADF / EKF filter launches'),
                    'filter_lin': ('fp64', per_step * lin_steps, 'fixed-site filter launches'),
                    'gain': ('fp64', (7.0 / 3.0) * S ** 3 * sm_steps_k, 'chol + two triangular solves'),
                    'scan': ('fp64', 4.0 * S ** 3 * sm_steps_k, 'G dP G\' (algorithmic; the parallel-in-time passes execute 2.5x)'),
                    'epsite': ('fp64', float(f_mom) * n_seg * (T - 1) * (EP_ITTS - 1) * steps, 'site refresh (cubature)')}
        tot_ms = sum(kern[k] for k in work if launches[k]) or 1.0
        table = []
        for k, (bound, amount, what) in work.items():
            if not launches[k] or kern[k] <= 0:
                continue
            rate = amount / (kern[k] * 1e-3) / (1e9 if bound == 'hbm' else 1e12)
            peak = PEAK_HBM_GBS if bound == 'hbm' else PEAK_FP64_TFLOPS
            table.append(dict(kernel=k, what=what, ms=kern[k] / steps, share=kern[k] / tot_ms, bound=bound, achieved=rate,
                              unit='GB/s' if bound == 'hbm' else 'TFLOP/s', frac=rate / peak))
        table.sort(key=lambda r: -r['ms'])
        roof['per_kernel'] = table
        roof['dominant_kernel'] = table[0]['kernel'] if table else None
        if table and name != 'cfg3' and table[0]['kernel'] != dom:
            # the headline figures of this extra describe the kernel with the largest share of ITS execute
            t0_ = table[0]
            roof.update(kernel='%s (%s)' % (t0_['kernel'], t0_['what']), bound=t0_['bound'], achieved=t0_['achieved'], peak=PEAK_HBM_GBS if t0_['bound'] == 'hbm' else PEAK_FP64_TFLOPS,
                        unit=t0_['unit'], frac=t0_['frac'], avg_launch_ms=kern[t0_['kernel']] / max(launches[t0_['kernel']], 1))
            dom = t0_['kernel']
        # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_run.sh -> profiles/pmc_traffic.json); only the
        # figures collected on THIS build (same source hash) and this workload are reported
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
                pm = json.load(fh).get(name)
            pc = (pm or {}).get('by_category', {}).get(dom)        # the dominant timing category of THIS run, whatever kernels served it
            if pc and pm.get('T') == T and pm.get('segments') == n_seg and world == 1 and pm.get('source_hash') == source_hash():
                per_exec = max(launches[dom] / steps, 1.0)            # launches of the dominant kernel in one execute
                roof['traffic'] = (pc['fetch_bytes_per_execute'] + pc['write_bytes_per_execute']) / per_exec
                roof['traffic_kernels'] = pc['kernels']
                roof['traffic_source'] = pm['source'] + ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, this build)'
                if roof['bound'] == 'hbm':   # the same launch priced with the bytes it really moved instead of SURVEY's per-sample figure
                    roof['achieved_from_traffic'] = roof['traffic'] / (roof['avg_launch_ms'] * 1e-3) / 1e9
        except (OSError, ValueError, KeyError):
            pass
    res = {
        'metric': 'audio samples/sec filtered+smoothed (state dim %d, per EP sweep)' % S,
        'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
        'dtype': 'f64', 'data': ('audio/%s.wav (decoded samples, tests/golden)' % wl['audio']) if (wl.get('audio') and not a.T) else 'synthetic',
        'config': {'workload': '%s: %s%s, %d channels / %d NMF components, T=%d, %d segment(s) in total = %s per GPU, p=%d cubature (%d points), %d EP sweeps%s'
                   % (name, wl['fn'], (' with ' + wl['lik']) if wl.get('lik') else '', D, N, T, n_total, ('%d' % n_seg) if scaling == 'weak' else ('%d/%d' % (n_total, world)), wl['p'], n_pts, EP_ITTS,
                      ('; input = the decoded samples of audio/%s.wav (tests/golden/audio_%s.npz), std-normalised' % (wl['audio'], wl['audio'])) if (wl.get('audio') and not a.T) else ''),
                   'state_dim': S, 'sites_per_step': M, 'parallelism': 'segments sharded over %d GPU(s), no data-path collective; nlZ all-reduce (%s)' % (world, nd.backend_name())},
        'end_to_end_samples_per_s': samples_per_step * steps / dt,
        'kernel_ms_per_step': {k: kern[k] / steps for k in kern if launches[k]},
        'nlZ_allreduced': [float(v) for v in np.atleast_1d(nlz_total)],
        'roofline': roof,
    }
    if wl.get('audio') and not a.T:
        # how well-posed the timed run is: the reference algorithm (sequential CPU restatement) on y against itself on y (1 + 1e-13), and
        # the GPU against it, measured at full length by tools/full_length_parity.py --audio; the same input is asserted on in
        # tests/test_gpu_parity.py::test_full_length_all_sweeps_against_the_sequential_cpu_algorithm[cfg2audio / cfg4audio]
        try:
            with open(os.path.join(ROOT, 'tests', 'golden', 'audio_conditioning.json')) as fh:
                cc = json.load(fh)['cases'].get({'cfg2': 'cfg2audio', 'cfg4': 'cfg4audio'}.get(name, ''), None)
            if cc and abs(cc.get('ep_damping', 0.5) - wl.get('damping', 0.5)) < 1e-12:
                res['input_conditioning'] = {'oracle_self_sensitivity': cc['oracle_moves'], 'gpu_minus_oracle': cc['gpu_minus_cpu'],
                                             'source': 'tests/golden/audio_conditioning.json'}
        except (OSError, ValueError, KeyError):
            pass
    if plan is not None:
        plan.close()
    if rank == 0 and with_cpu:
        cb = cpu_baseline(wl)
        res['cpu_baseline'] = cb
        # this rank's whole GPU (all its segments) against: every usable host core running independent segments (structured compiled
        # port); one core, structured; one core, dense as written (the cost of the MATLAB text); one core, NumPy oracle
        res['speedup_vs_cpu_all_cores_structured'] = value / world / cb['value']
        res['speedup_vs_cpu_single_thread_structured'] = value / world / cb['single_thread_value']
        res['speedup_vs_cpu_single_thread_dense_as_written'] = value / world / cb['dense_as_written_single_thread_value']
        res['speedup_vs_cpu_single_thread_numpy_oracle'] = value / world / cb['numpy_oracle_single_thread_value']
    return res


def spawn_ranks(a):
    """--gpus N without a launcher: N fresh child processes, one per GPU.  Nothing in THIS process has touched the GPU."""
    import tempfile
    from nagp import dist as nd                                  # pure Python: no GPU call
    rdv = tempfile.mkdtemp(prefix='nagp_bench_')
    procs = []
    for r in range(a.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), NAGP_BENCH_SELF_SPAWNED='1',
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'), **nd.file_rendezvous_env(rdv, a.gpus))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    rc = 0
    pending = list(procs)
    while pending:
        for pr in list(pending):
            r = pr.poll()
            if r is None:
                continue
            pending.remove(pr)
            if r != 0:
                rc = r
                for other in pending:       # exact PIDs of our own children
                    other.terminate()
        time.sleep(0.2)
    import shutil
    shutil.rmtree(rdv, ignore_errors=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='cfg3', choices=sorted(WORKLOADS))
    ap.add_argument('--segments', type=int, default=0, help='diagnostics: this many segments on EVERY GPU instead of the named configuration')
    ap.add_argument('--T', type=int, default=0, help='override the segment length (diagnostics only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--extras', default='default', help="comma list of additional workloads measured into the same line ('default': cfg5 [+ cfg2 at N=1]; 'none')")
    a = ap.parse_args()

    if a.gpus > 1 and 'RANK' not in os.environ:
        sys.exit(spawn_ranks(a))

    import torch
    import nagp
    from nagp import dist as nd
    rehearsal = bool(os.environ.get('NAGP_BENCH_REHEARSAL'))     # several ranks on ONE card over gloo: exercises the launch path only
    rank, local_rank, world = nd.init('nccl' if (torch.cuda.is_available() and not rehearsal) else 'gloo')
    if a.gpus != world:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback)')
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = None if rehearsal else torch.device('cuda', local_rank)
    if rank == 0:
        nagp.build()                                              # one rank compiles (if the library is stale), the others wait
    nd.barrier()

    with_cpu = (not a.no_cpu_baseline) and world == 1             # the CPU baseline is timed at N = 1 only
    line = run_workload(a.workload, a, rank, local_rank, world, dev, with_cpu, a.steps, a.warmup)
    line['world_size'] = nd.world_size()                       # what the initialised process group reports
    line['collective_backend'] = nd.backend_id()                # 'nccl' (= RCCL on ROCm) | 'gloo' (rehearsal: several ranks on one card) | 'none'
    line['devices'] = nd.gather_device_ids(local_rank)          # one entry per rank: a SCALE record shows N distinct PCI bus ids
    line['launch'] = 'self-spawned ranks (file-store rendezvous)' if os.environ.get('NAGP_BENCH_SELF_SPAWNED') and world > 1 else ('torch.distributed.run' if world > 1 else 'single process')
    extras = []
    if a.extras == 'default':
        if a.workload == 'cfg3' and not a.T and not a.segments:
            # configs[4] sharded (strong) and in the fill-the-chip form (weak), the full-chip batch lines of both kernel families,
            # and at N = 1 the other single-GPU configurations
            extras = ['cfg5_fill', 'cfg3_batch', 'cfg2_batch', 'cfg5'] + (['cfg2', 'cfg4', 'cfg3_sqrt'] if world == 1 else [])     # (the weak series first: they are the ones that scale)
    elif a.extras != 'none':
        extras = [e for e in a.extras.split(',') if e]
    for e in extras:
        # extras: one warm-up + two timed steps (the contract's K and W apply to the top-level line); CPU baseline for the BASELINE configurations
        ex = run_workload(e, a, rank, local_rank, world, dev, with_cpu and e in ('cfg2', 'cfg4', 'cfg5'), 2, 1)
        key = 'cfg5_strong' if e == 'cfg5' else e
        if e == 'cfg5':
            ex['scaling_note'] = 'strong series capped at ~1.2x: one 100k-step segment alone = 3.7 s of sequential recursion, all 8 on one GPU 4.4 s; cfg5_fill / cfg3_batch are the series that scale'
        # The line has to survive the driver's tail (round 3: 14 KB, half of the extras cut off): an extra keeps its numbers -- value, time,
        # the roofline of its dominant kernel, the per-kernel table as [kernel, ms, fraction of peak] -- and its full record goes to stderr.
        if rank == 0:
            sys.stderr.write('[bench extra] ' + json.dumps({key: ex}) + '\n')
        rf = ex.get('roofline') or {}
        small = {'value': sig(ex['value']), 'ms_per_step': sig(ex['ms_per_step']), 'scaling': ex['scaling'], 'workload': ex['config']['workload'].split(';')[0][:160],
                 'state_dim': ex['config']['state_dim'], 'kernel_ms': {k: sig(v) for k, v in ex['kernel_ms_per_step'].items()},
                 'nlZ_allreduced': [sig(v, 8) for v in ex['nlZ_allreduced']],
                 'roofline': {k: (sig(rf[k]) if isinstance(rf.get(k), float) else rf.get(k)) for k in ('kernel', 'bound', 'achieved', 'peak', 'unit', 'frac', 'traffic') if k in rf},
                 'per_kernel': [[r['kernel'], sig(r['ms']), r['bound'], sig(r['frac'])] for r in rf.get('per_kernel', [])]}
        if rf.get('whole_call'):
            small['whole_call_frac'] = sig(rf['whole_call']['frac'])
        if ex.get('data') and ex['data'] != 'synthetic':
            small['data'] = ex['data']
        if 'input_conditioning' in ex:
            small['input_conditioning'] = ex['input_conditioning']
        if 'cpu_baseline' in ex:
            cb = ex['cpu_baseline']
            small['cpu_baseline'] = {'value': sig(cb['value']), 'unit': cb['unit'], 'cores': cb['cores'], 'kind': cb['kind'], 'single_thread_value': sig(cb['single_thread_value'])}
        if 'scaling_note' in ex:
            small['scaling_note'] = ex['scaling_note']
        line[key] = small
    if rank == 0:
        def compact(o):     # five significant digits are what the timings carry (the full figures of the headline are on stderr)
            if isinstance(o, float): return sig(o, 6)
            if isinstance(o, dict): return {k: (v if k in ('value', 'ms_per_step', 'nlZ_allreduced') else compact(v)) for k, v in o.items()}
            if isinstance(o, list): return [compact(v) for v in o]
            return o
        sys.stderr.write('[bench headline] ' + json.dumps({k: v for k, v in line.items() if k in ('roofline', 'cpu_baseline')}) + '\n')
        print(json.dumps(compact(line)))
        sys.stdout.flush()
    if world > 1:
        nd.finalize()                      # barrier + destroy: no rank leaves while a peer's threads still hold its sockets


if __name__ == '__main__':
    main()
