#!/usr/bin/env python3
"""bench.py -- throughput of the Kalman/RTS/Power-EP hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg3|cfg4|cfg5] [--segments B] [--T T] [--extras ...]

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
    # name: function, D, N, T, cubature order, parameter recipe, balance, segments IN TOTAL (None: one per GPU)
    'cfg2': dict(fn='gf_ep_modulator_nmf', D=16, N=3, T=84010, p=9, recipe='demo_nmf', balance=False, total_segments=None),
    'cfg3': dict(fn='ihgp_ep_modulator_nmf', D=32, N=6, T=200000, p=7, recipe='constraints', balance=True, total_segments=None),
    'cfg4': dict(fn='gf_giekf_modulator_nmf', D=24, N=3, T=88200, p=9, recipe='demo_nmf', balance=True, total_segments=None),
    'cfg5': dict(fn='gf_ep_modulator_nmf_constraints', D=32, N=6, T=100000, p=7, recipe='constraints', balance=True, total_segments=8),
}
EP_ITTS = 3
PEAK_FP64_TFLOPS = 78.6   # MI355X FP64 vector = matrix peak (MI355X_MICROARCH.md / SURVEY App. E)
PEAK_HBM_GBS = 8000.0


def build_problems(wl, seeds):
    from nagp import harness
    from nagp import ss as ssm
    probs, ys = [], []
    for sd in seeds:
        pr = harness.nmf_problem(wl['D'], wl['N'], wl['T'], sd, wl['recipe'])
        blk = ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
        if wl['balance']:
            blk = ssm.balance_blocks(blk)
        probs.append((blk, pr['W'], np.log(pr['w_lik'])))
        ys.append(pr['y'])
    return probs, ys


def _oracle_rate(args):
    """One oracle run on the first Ts samples of the workload (NumPy restatement of the reference, dense as written).
    Top-level so that a process pool can pickle it.  Returns (samples x sweeps per second, seconds)."""
    wl, Ts, seed = args
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
    try:
        from threadpoolctl import threadpool_limits
        ctx = threadpool_limits(limits=1)
    except Exception:
        import contextlib
        ctx = contextlib.nullcontext()
    from nagp import harness
    from oracle import gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik, ss as oss
    D, N = wl['D'], wl['N']
    om = olik.Mom(olik.LIK_POWER_NMF, p=wl['p'])
    d = 0.5 * np.ones(EP_ITTS)
    with ctx:
        pr = harness.nmf_problem(D, N, Ts, seed, wl['recipe'])
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
    return Ts * EP_ITTS / dt, dt


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
    try:
        usable = len(os.sched_getaffinity(0))
    except AttributeError:
        usable = os.cpu_count() or 1
    return model, os.cpu_count() or 1, usable


def cpu_baseline(wl, budget_s=12.0):
    """The oracle timed on this host: (i) one thread, (ii) one independent segment per core on min(usable cores, 16)
    worker processes (the path's own parallel axis).  Bounded prefix of the same workload."""
    Ts, rate1, dt = 100, None, 0.0
    while True:
        rate1, dt = _oracle_rate((wl, Ts, 1000))
        if dt > budget_s / 3 or Ts >= 4000:
            break
        Ts = int(min(4000, max(Ts * 2, Ts * (budget_s / 1.5) / max(dt, 1e-3))))
    model, nproc, usable = host_cpu()
    cores = max(1, min(usable, 16))
    rate_all, dt_all = rate1, dt
    if cores > 1:
        import multiprocessing as mp
        t0 = time.perf_counter()
        with mp.get_context('spawn').Pool(cores) as pool:      # fresh interpreters: nothing GPU-related is inherited
            res = pool.map(_oracle_rate, [(wl, Ts, 1000 + q) for q in range(cores)])
        wall = max(r[1] for r in res)                           # the slowest worker bounds the job
        rate_all = cores * Ts * EP_ITTS / wall
        dt_all = time.perf_counter() - t0
    return dict(value=rate_all, unit='samples/s', cores=cores, kind='port',
                single_thread_value=rate1, host_cpu_model=model, host_nproc=nproc, host_usable_cores=usable,
                sample='oracle (NumPy restatement of the reference loop, dense as written; MATLAB/Octave absent) on the first %d samples x %d sweeps '
                       'of the same workload: 1 thread %.1f s; %d worker processes with one independent segment each %.1f s wall' % (Ts, EP_ITTS, dt, cores, dt_all))


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
    else:                                    # one segment per GPU
        seeds = [1000 + 100 * rank]
        n_total = world; scaling = 'weak'
    n_seg = len(seeds)
    kind = {'gf_ep': L.KIND_GF_EP, 'ihgp_': L.KIND_IHGP, 'gf_gi': L.KIND_GIEKF}[wl['fn'][:5]]
    mom = None if kind == L.KIND_GIEKF else Mom('likModulatorNMFPower', p_cubature=wl['p'])
    plan = None
    if n_seg:
        probs, ys = build_problems(wl, seeds)
        plan = nagp.Plan(kind, probs, wl['T'], mom=mom, ep_fraction=0.5, ep_damping=0.5 * np.ones(EP_ITTS), ep_itts=EP_ITTS,
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
        roof = dict(bound='hbm', kernel='ihgp ADF sweep (ihgp_adf8_kernel / ihgp_adf_kernel / ihgp_filter_kernel, one launch per execute)',
                    achieved=achieved, peak=PEAK_HBM_GBS, unit='GB/s', frac=achieved / PEAK_HBM_GBS, traffic=None,
                    algorithmic_bytes_per_sample=per_sample, algorithmic_bytes_per_launch=per_launch,
                    valu_gflops=(8.0 * S * (S / M) + 8 * S + f_mom) * units / (kern[dom] * 1e-3) / 1e9,
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
        roof['adf_us_per_sample'] = kern[dom] * 1e3 / adf_steps
        roof['algorithmic_bytes_per_launch'] = 8.0 * (8 * M * (M + 1) + S + 5 * M + 2) * adf_steps / max(launches[dom], 1)   # lower tiles + means + sites
    if roof is not None:
        # HBM bytes per launch from separate rocprofv3 --pmc passes (tools/pmc_run.sh -> profiles/pmc_traffic.json); only the
        # figures collected on THIS build (same source hash) and this workload are reported
        try:
            with open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json')) as fh:
                pm = json.load(fh).get(name)
            if pm and pm.get('T') == T and pm.get('segments') == n_seg and world == 1 and pm.get('source_hash') == source_hash():
                per_exec = max(launches[dom] / steps, 1.0)            # launches of the dominant kernel in one execute
                roof['traffic'] = (pm['fetch_bytes_per_execute'] + pm['write_bytes_per_execute']) / per_exec
                roof['traffic_source'] = pm['source'] + ' (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, this build)'
        except (OSError, ValueError, KeyError):
            pass
    res = {
        'metric': 'audio samples/sec filtered+smoothed (state dim %d, per EP sweep)' % S,
        'value': value, 'unit': 'samples/s', 'n_gpus': world, 'steps': steps, 'warmup': warmup,
        'ms_per_step': dt / steps * 1e3, 'higher_is_better': True, 'scaling': scaling, 'vs_baseline': None,
        'dtype': 'f64', 'data': 'synthetic',
        'config': {'workload': '%s: %s, %d channels / %d NMF components, T=%d, %d segment(s) in total = %s per GPU, p=%d cubature (%d points), %d EP sweeps'
                   % (name, wl['fn'], D, N, T, n_total, ('%d' % n_seg) if scaling == 'weak' else ('%d/%d' % (n_total, world)), wl['p'], n_pts, EP_ITTS),
                   'state_dim': S, 'sites_per_step': M, 'parallelism': 'segments sharded over %d GPU(s), no data-path collective; nlZ all-reduce (%s)' % (world, nd.backend_name())},
        'end_to_end_samples_per_s': samples_per_step * steps / dt,
        'kernel_ms_per_step': {k: kern[k] / steps for k in kern if launches[k]},
        'nlZ_allreduced': [float(v) for v in np.atleast_1d(nlz_total)],
        'roofline': roof,
    }
    if plan is not None:
        plan.close()
    if rank == 0 and with_cpu:
        res['cpu_baseline'] = cpu_baseline(wl)
        res['speedup_vs_cpu_baseline'] = value / n_total / res['cpu_baseline']['value'] * 1.0          # one GPU segment stream vs ALL host cores
        res['speedup_vs_cpu_single_thread'] = value / n_total / res['cpu_baseline']['single_thread_value']
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
    line['rccl_world_size'] = nd.world_size()
    line['launch'] = 'self-spawned ranks (file-store rendezvous)' if os.environ.get('NAGP_BENCH_SELF_SPAWNED') and world > 1 else ('torch.distributed.run' if world > 1 else 'single process')
    extras = []
    if a.extras == 'default':
        if a.workload == 'cfg3' and not a.T and not a.segments:
            extras = ['cfg5'] + (['cfg2'] if world == 1 else [])
    elif a.extras != 'none':
        extras = [e for e in a.extras.split(',') if e]
    for e in extras:
        # extras: one warm-up + one timed step (the contract's K and W apply to the top-level line)
        ex = run_workload(e, a, rank, local_rank, world, dev, with_cpu and e != 'cfg5', 1, 1)
        key = 'cfg5_strong' if e == 'cfg5' else e
        line[key] = {k: ex[k] for k in ('metric', 'value', 'unit', 'scaling', 'ms_per_step', 'steps', 'warmup', 'config', 'end_to_end_samples_per_s', 'kernel_ms_per_step',
                                        'nlZ_allreduced', 'roofline', 'cpu_baseline', 'speedup_vs_cpu_baseline', 'speedup_vs_cpu_single_thread') if k in ex}
    if rank == 0:
        print(json.dumps(line))
        sys.stdout.flush()
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
