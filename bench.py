#!/usr/bin/env python3
"""bench.py -- throughput of the Kalman/RTS/Power-EP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5] [--segments B] [--T T]

A "step" is one full pass of the hot path (all EP sweeps: forward filter, RTS smoother, site refresh)
over one batch of synthetic audio segments already resident in HBM.  The default workload is
BASELINE.json configs[1] (gf_ep_modulator_nmf, 16 channels / 3 NMF components, 84 010 samples -- the
length of audio/speech_74.wav -- as synthetic audio of that shape).  One process per GPU; segments
are sharded over ranks with no data-path collective; the per-sweep log-marginal-likelihood vector is
all-reduced (RCCL) inside the timed region.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))

import numpy as np  # noqa: E402

WORKLOADS = {
    # name: (function, D, N, T, cubature order, parameter recipe, balance)
    'cfg2': dict(fn='gf_ep_modulator_nmf', D=16, N=3, T=84010, p=9, recipe='demo_nmf', balance=False, segments=1),
    'cfg3': dict(fn='ihgp_ep_modulator_nmf', D=32, N=6, T=200000, p=7, recipe='constraints', balance=True, segments=1),
    'cfg4': dict(fn='gf_giekf_modulator_nmf', D=24, N=3, T=88200, p=9, recipe='demo_nmf', balance=True, segments=1),
    'cfg5': dict(fn='gf_ep_modulator_nmf_constraints', D=32, N=6, T=100000, p=7, recipe='constraints', balance=True, segments=1),
}
EP_ITTS = 3
PEAK_FP64_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (spec)
PEAK_HBM_GBS = 8000.0


def build_problems(wl, n_seg, seed0):
    from nagp import harness
    from nagp import ss as ssm
    probs, ys = [], []
    for q in range(n_seg):
        pr = harness.nmf_problem(wl['D'], wl['N'], wl['T'], seed0 + q, wl['recipe'])
        blk = ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
        if wl['balance']:
            blk = ssm.balance_blocks(blk)
        probs.append((blk, pr['W'], np.log(pr['w_lik'])))
        ys.append(pr['y'])
    return probs, ys


def cpu_baseline(wl, budget_s=15.0):
    """The oracle (NumPy restatement of the reference algorithm, dense as written) timed on this host,
    one thread, on a bounded prefix of the same workload."""
    try:
        from threadpoolctl import threadpool_limits
    except Exception:
        threadpool_limits = None
    from nagp import harness
    from oracle import gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik, ss as oss
    import contextlib
    ctx = threadpool_limits(limits=1) if threadpool_limits else contextlib.nullcontext()
    D, N = wl['D'], wl['N']
    om = olik.Mom(olik.LIK_POWER_NMF, p=wl['p'])
    d = 0.5 * np.ones(EP_ITTS)
    with ctx:
        Ts, rate = 200, None
        while True:
            pr = harness.nmf_problem(D, N, Ts, 1000, wl['recipe'])
            t = np.arange(1, Ts + 1.0)
            t0 = time.perf_counter()
            lik_param, p1, p2, W = oss.unpack_log(pr['w'], 1, D, N)
            if wl['fn'].startswith('ihgp'):
                model = ogf.assemble(lik_param, p1, p2, W, 'matern32', 'matern52', True, True)
                tabs = oih.build_tables(model)   # DARE tables are set-up, not the timed loop
                t0 = time.perf_counter()
                oih.run_predict(model, pr['y'], om, 0.5, d, EP_ITTS, tables=tabs)
            elif wl['fn'].startswith('gf_giekf'):
                model = ogf.assemble(lik_param, p1, p2, W, 'matern32', 'matern52', True)
                t0 = time.perf_counter()
                oek.run_predict(model, pr['y'], D, N, EP_ITTS, 1)
            else:
                model = ogf.assemble(lik_param, p1, p2, W, 'matern32', 'matern52', wl['balance'])
                t0 = time.perf_counter()
                ogf.run_predict(model, pr['y'], om, 0.5, d, EP_ITTS)
            dt = time.perf_counter() - t0
            rate = Ts * EP_ITTS / dt
            if dt > budget_s / 3 or Ts >= 4000:
                break
            Ts = int(min(4000, max(Ts * 2, Ts * (budget_s / 1.5) / max(dt, 1e-3))))
    return dict(value=rate, unit='samples/s', cores=1, kind='port',
                sample='oracle (NumPy restatement, dense as written, 1 thread) on the first %d samples x %d sweeps of the same workload' % (Ts, EP_ITTS))


def run_workload(name, a, rank, local_rank, world, dev, with_cpu):
    """Times a.steps executes of one named workload; returns the JSON-able result dict."""
    import nagp
    from nagp import Mom, _lib as L, dist as nd
    import torch
    wl = dict(WORKLOADS[name])
    if a.T:
        wl['T'] = a.T
    n_seg = a.segments or wl['segments']
    kind = {'gf_ep': L.KIND_GF_EP, 'ihgp_': L.KIND_IHGP, 'gf_gi': L.KIND_GIEKF}[wl['fn'][:5]]
    probs, ys = build_problems(wl, n_seg, 1000 + 100 * rank)     # weak scaling: every rank has its own segments
    mom = None if kind == L.KIND_GIEKF else Mom('likModulatorNMFPower', p_cubature=wl['p'])
    plan = nagp.Plan(kind, probs, wl['T'], mom=mom, ep_fraction=0.5, ep_damping=0.5 * np.ones(EP_ITTS), ep_itts=EP_ITTS,
                     l_iter=1, device=local_rank)
    plan.upload(ys)                                               # inputs resident in HBM before timing

    def step():
        plan.execute()                                            # all sweeps, synchronous on the plan's stream
        return nd.allreduce_nlz(plan.download_nlz(), dev)         # RCCL all-reduce of nlZ[itt] (8*I bytes)

    for _ in range(a.warmup):
        step()
    nd.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    kern = {k: 0.0 for k in L.KERNEL_NAMES}; launches = {k: 0 for k in L.KERNEL_NAMES}
    for _ in range(a.steps):
        nlz_total = step()
        tm = plan.timings()
        for k in kern:
            kern[k] += tm['ms'][k]; launches[k] += tm['launches'][k]
    torch.cuda.synchronize(); nd.barrier()
    dt = nd.allreduce_max(time.perf_counter() - t0, dev)

    S, M, T, D, N = plan.S, plan.M, wl['T'], wl['D'], wl['N']
    samples_per_step = world * n_seg * T
    value = samples_per_step * EP_ITTS * a.steps / dt
    n_pts = mom.tables(N)[0].size if mom is not None else 0
    f_mom = n_pts * (2 * N * D + 12 * D + 8 * N + 10)
    if kind == L.KIND_IHGP:
        dom = 'filter'
        per_sample = 8.0 * (3 * S + 9 * M + 1)                    # IHGP algorithmic bytes / sample / sweep
        units = n_seg * T * a.steps                               # one ADF-filter launch covers T samples of every segment
        achieved = per_sample * units / (kern[dom] * 1e-3) / 1e9
        roof = dict(bound='hbm', kernel='ihgp_filter_kernel (ADF sweep)', achieved=achieved, peak=PEAK_HBM_GBS, unit='GB/s',
                    frac=achieved / PEAK_HBM_GBS, traffic=None, algorithmic_bytes_per_sample=per_sample,
                    valu_gflops=(8.0 * S * (S / M) + 8 * S + f_mom) * units / (kern[dom] * 1e-3) / 1e9,
                    avg_launch_ms=kern[dom] / max(launches[dom], 1), us_per_sample=kern[dom] * 1e3 / units)
    else:
        # gf_ep: the ADF launches (sweep 1: all T steps, later sweeps: the step k = T-1) are one kernel
        # (gf_filter_kernel<.., MV, 256>, timing slot 'filter'); the fixed-site steps of sweeps >= 2 run in the
        # mom-free instantiation (slot 'filter_lin').  giekf: every sweep is the EKF instantiation (slot 'filter').
        dom = 'filter'
        bbar = S / M
        per_step = (4 * bbar + 1) * S * S + 2.0 * M * S * S       # block-diagonal A P A' + Q, rank-M update
        if kind == L.KIND_GF_EP:
            adf_steps = n_seg * (T + (EP_ITTS - 1)) * a.steps     # steps the ADF kernel processed (3 launches per execute)
            lin_steps = n_seg * (T - 1) * (EP_ITTS - 1) * a.steps
            flops = (per_step + f_mom) * adf_steps
        else:
            per_step = (4 * bbar + 1) * S * S + 6.0 * S * S       # EKF: l_iter*4S^2 + 2S^2 instead of the rank-M update
            adf_steps = n_seg * T * EP_ITTS * a.steps; lin_steps = 0
            flops = per_step * adf_steps
        achieved = flops / (kern[dom] * 1e-3) / 1e12
        roof = dict(bound='mfma', kernel='gf_filter_kernel (ADF / EKF launches)', achieved=achieved, peak=PEAK_FP64_TFLOPS, unit='TFLOP/s',
                    frac=achieved / PEAK_FP64_TFLOPS, traffic=None, algorithmic_flops_per_step=per_step + (f_mom if kind == L.KIND_GF_EP else 0),
                    avg_launch_ms=kern[dom] / max(launches[dom], 1), launches_per_execute=launches[dom] / a.steps,
                    # a segment is one sequential recursion on one workgroup = one CU of 256: the per-CU ceiling it can reach
                    peak_one_cu=PEAK_FP64_TFLOPS / 256, frac_one_cu=achieved / (PEAK_FP64_TFLOPS / 256 * n_seg),
                    smoother_tflops=(10.0 * S ** 3) * n_seg * (T - 1) * EP_ITTS * a.steps / ((kern['scan'] + kern['gain']) * 1e-3) / 1e12)
        if lin_steps and kern['filter_lin'] > 0:
            roof['fixed_site_kernel'] = dict(kernel='gf_filter_kernel<.., MV=-1> (sweeps >= 2, k < T-1)',
                                             achieved=per_step * lin_steps / (kern['filter_lin'] * 1e-3) / 1e12, unit='TFLOP/s',
                                             avg_launch_ms=kern['filter_lin'] / max(launches['filter_lin'], 1),
                                             us_per_sample=kern['filter_lin'] * 1e3 / lin_steps)
        roof['adf_us_per_sample'] = kern[dom] * 1e3 / adf_steps
        algo_bytes = 8.0 * (8 * M * (M + 1) + S + 5 * M + 2) * adf_steps / max(launches[dom], 1)   # lower tiles + means + sites, per launch
        roof['algorithmic_bytes_per_launch'] = algo_bytes
    try:    # PMC traffic is collected in separate rocprofv3 --pmc passes (profiles/); copied here when it is this workload
        with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'profiles', 'pmc_traffic.json')) as fh:
            pm = json.load(fh).get(name)
        if pm and pm.get('T') == T and pm.get('segments') == n_seg and world == 1:
            roof['traffic'] = pm['fetch_bytes_per_launch'] + pm['write_bytes_per_launch']
            roof['traffic_source'] = pm['source']
    except (OSError, ValueError):
        pass
    res = {
        'metric': 'audio samples/sec filtered+smoothed (state dim %d, per EP sweep)' % S,
        'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
        'ms_per_step': dt / a.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: %s, %d channels / %d NMF components, T=%d, %d segment(s) per GPU, p=%d cubature (%d points), %d EP sweeps'
                   % (name, wl['fn'], D, N, T, n_seg, wl['p'], n_pts, EP_ITTS),
                   'state_dim': S, 'sites_per_step': M, 'parallelism': 'segments sharded over %d GPU(s)' % world},
        'end_to_end_samples_per_s': samples_per_step * a.steps / dt,
        'kernel_ms_per_step': {k: kern[k] / a.steps for k in kern if launches[k]},
        'nlZ_allreduced': [float(v) for v in np.atleast_1d(nlz_total)],
        'roofline': roof,
    }
    plan.close()
    if rank == 0 and with_cpu:
        res['cpu_baseline'] = cpu_baseline(wl)
        res['speedup_vs_cpu_baseline'] = value / world / res['cpu_baseline']['value']
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=2)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='cfg2', choices=sorted(WORKLOADS))
    ap.add_argument('--segments', type=int, default=0, help='segments per GPU (default: the named configuration)')
    ap.add_argument('--T', type=int, default=0, help='override the segment length (diagnostics only)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extra', action='store_true', help='skip the additional 200k-sample IHGP workload (cfg3) line item')
    a = ap.parse_args()

    import torch
    import nagp
    from nagp import dist as nd
    rehearsal = bool(os.environ.get('NAGP_BENCH_REHEARSAL'))     # several ranks on ONE card over gloo: exercises the launch path only
    rank, local_rank, world = nd.init('nccl' if (torch.cuda.is_available() and not rehearsal) else 'gloo')
    if a.gpus != world and rank == 0 and world > 1:
        print('warning: --gpus %d but WORLD_SIZE=%d' % (a.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a GPU (no CPU fallback)')
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if rank == 0:
        nagp.build()                                              # one rank compiles (if the library is stale), the others wait
    nd.barrier()

    with_cpu = (not a.no_cpu_baseline) and world == 1             # the CPU baseline is timed at N = 1 only
    line = run_workload(a.workload, a, rank, local_rank, world, dev, with_cpu)
    if a.workload == 'cfg2' and not a.no_extra and not a.T and not a.segments:
        # the north-star target is stated on the 200k-sample, 32-channel sweep (BASELINE.json configs[2])
        extra = run_workload('cfg3', a, rank, local_rank, world, dev, with_cpu)
        line['target_workload_cfg3'] = {k: extra[k] for k in ('value', 'unit', 'ms_per_step', 'config', 'kernel_ms_per_step', 'roofline',
                                                               'cpu_baseline', 'speedup_vs_cpu_baseline') if k in extra}
    if rank == 0:
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
