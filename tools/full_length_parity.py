#!/usr/bin/env python3
"""End-to-end parity at the sizes the contract is stated on: the exact bench.py workloads -- all sweeps, default chunking, the
chunk-pipelined schedule, the parallel-in-time scans -- against the COMPILED restatement of the reference's sequential loops
(oracle/cpu/nagp_cpu.cpp, structured form; held to the NumPy oracle at 1e-9 by tests/test_cpu_restatement.py).

    python tools/full_length_parity.py [--cases cfg3,cfg2,cfg4,cfg5seg] [--out profiles/r04_full_length_parity.txt]

cases (bench.py WORKLOADS, seed 1000 = the segment rank 0 times; cfg2audio / cfg4audio = bench.py's cfg2 / cfg4 themselves: the decoded audio
files BASELINE names, cfg2 with the drivers' damping 0.1):
  cfg3     ihgp_ep_modulator_nmf, T = 200 000, 32 ch / 6 comps, p = 7, 3 sweeps          (north_star's target sentence)
  cfg2     gf_ep_modulator_nmf, T = 84 010 (the length of audio/speech_74.wav; prior sample), 16 ch / 3 comps, p = 9, 3 sweeps
  cfg4     gf_giekf_modulator_nmf, T = 88 200 (the length of audio/stim312_wind.wav; prior sample), 24 ch / 3 comps, g_iter = 3, l_iter = 1
  cfg5seg  gf_ep_modulator_nmf_constraints model (S = 146), one segment cut to T = 20 000, p = 7, 3 sweeps
  cfg3sqrt cfg3 with experiments/likModulatorPreCalcwn.m (bench.py's cfg3_sqrt: ihgp_adf8sq_kernel), T = 200 000, 3 sweeps
The four CPU legs (one core each: about 35 / 90 / 200 / 115 s) run side by side in threads (ctypes releases the GIL), the GPU
runs beside them.  Used by tests/test_gpu_parity.py (which asserts the stated tolerances) and, as a script, writes the
measured differences to profiles/.

Reference loops: gf_ep_modulator_nmf.m:126-283, ihgp_ep_modulator_nmf.m:233-442, gf_giekf_modulator_nmf.m:126-221.
"""
import argparse
import os
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))

import numpy as np  # noqa: E402

CASES = {
    'cfg3': dict(fam='ihgp', D=32, N=6, T=200000, p=7, recipe='constraints', balance=True),
    'cfg2': dict(fam='gf', D=16, N=3, T=84010, p=9, recipe='demo_nmf', balance=False),
    'cfg4': dict(fam='giekf', D=24, N=3, T=88200, p=9, recipe='demo_nmf', balance=True),
    'cfg5seg': dict(fam='gf', D=32, N=6, T=20000, p=7, recipe='constraints', balance=True),
    # bench.py's cfg2 / cfg4 inputs: the decoded audio files BASELINE names (tests/golden/audio_*.npz).  cfg2 runs with the damping of the
    # reference's speech drivers (experiments/noise_reduction_speech.m:29: ep_damping = 0.1): with 0.5 the reference algorithm itself is
    # chaotic on this file (the oracle's Eft moves by 50 % under a 1e-13 relative change of y), with 0.1 it moves by 1e-13 -- so THIS is
    # the input bench.py times and the tests assert on.  The chaotic recipe stays as an informative case (script mode, --audio).
    'cfg2audio': dict(fam='gf', D=16, N=3, T=84010, p=9, recipe='demo_nmf', balance=False, audio='speech_74', damping=0.1),
    'cfg4audio': dict(fam='giekf', D=24, N=3, T=88200, p=9, recipe='demo_nmf', balance=True, audio='stim312_wind'),
    'cfg2audio_d05': dict(fam='gf', D=16, N=3, T=84010, p=9, recipe='demo_nmf', balance=False, audio='speech_74', informative=True),
    # bench.py's cfg3_sqrt: the same model with experiments/likModulatorPreCalcwn.m (sqrt amplitudes, softplus(g - 1), ut7 passed in precomputed)
    'cfg3sqrt': dict(fam='ihgp', D=32, N=6, T=200000, p=7, recipe='constraints', balance=True, lik='sqrt', link_shift=1.0, damping=0.1),
}
SWEEPS = 3
SEED = 1000
# stated tolerances (tests/test_gpu_parity.py): means / variances relative to the array's largest magnitude, sites, log Z
TOL_MEAN, TOL_SITE, TOL_LOGZ = 1e-7, 1e-6, 1e-8
SITE_MAX = 1e8       # tools/fuzz_conditioning.py


def problem(name, T=None):
    """bench.py's inputs: the model of seed 1000; y = the decoded samples of the audio file BASELINE names (cfg2, cfg4: tests/golden/audio_*.npz,
    int16 / 32768, std-normalised) or the prior sample of that seed"""
    from nagp import harness
    c = CASES[name]
    if c.get('audio') and not T:
        z = np.load(os.path.join(ROOT, 'tests', 'golden', 'audio_%s.npz' % c['audio']))
        x = z['samples'].astype(np.float64) / 32768.0
        pr = harness.nmf_problem(c['D'], c['N'], 8, SEED, c['recipe'])
        pr['y'] = x / np.std(x)
        assert pr['y'].size == c['T']
        return pr
    return harness.nmf_problem(c['D'], c['N'], T or c['T'], SEED, c['recipe'], link_shift=c.get('link_shift', 0.0), sqrt_amp=c.get('lik') == 'sqrt')


class CpuLegs:
    """The sequential reference algorithm (compiled oracle) of every case, started once, each on a thread of its own."""

    def __init__(self, names, problems=None):
        self.names = list(names); self.res = {}; self.err = {}; self.secs = {}; self.threads = {}
        self.problems = problems or {}
        self._started = False

    def start(self):
        if self._started:
            return self
        self._started = True
        from oracle import cpu as ocpu
        ocpu.build(); ocpu.lib()
        for n in self.names:
            if n not in self.problems:
                self.problems[n] = problem(n)
            th = threading.Thread(target=self._run, args=(n,), daemon=True)
            self.threads[n] = th; th.start()
        return self

    def _run(self, name):
        try:
            from oracle import cpu as ocpu, gf_ep as ogf, ihgp as oih, lik as olik, ss as oss
            c = CASES[name]; pr = self.problems[name]; D, N = c['D'], c['N']
            lik_param, p1, p2, W = oss.unpack_log(pr['w'], 1, D, N)
            model = ogf.assemble(lik_param, p1, p2, W, 'matern32', 'matern52', c['balance'], c['fam'] == 'ihgp')
            d = c.get('damping', 0.5) * np.ones(SWEEPS)
            if c.get('lik') == 'sqrt':
                from oracle import cubature as ocub
                wn, xn = ocub.sigma_points(c['p'], N, True)
                omom = olik.Mom(olik.LIK_POWER_NMF_SQRT, link=olik.softplus_link(c['link_shift']), wn=wn, xn_unscaled=xn)
            else:
                omom = olik.Mom(olik.LIK_POWER_NMF, p=c['p'])
            t0 = time.perf_counter()
            if c['fam'] == 'ihgp':
                r = ocpu.ihgp_predict(model, pr['y'], omom, 0.5, d, SWEEPS, D, N, oih.build_tables(model), structured=True)
            elif c['fam'] == 'giekf':
                r = ocpu.giekf_predict(model, pr['y'], D, N, SWEEPS, 1, structured=True)
            else:
                r = ocpu.gf_predict(model, pr['y'], omom, 0.5, d, SWEEPS, D, N, structured=True)
            self.secs[name] = time.perf_counter() - t0
            self.res[name] = r
        except BaseException as e:          # handed to the waiting test
            self.err[name] = e

    def result(self, name):
        self.start()
        self.threads[name].join()
        if name in self.err:
            raise self.err[name]
        return self.res[name]


def gpu_run(name, pr=None, env=None):
    """The bench.py workload of `name` through the product (Plan -> C ABI -> HIP): outputs of the one segment, seconds per execute."""
    from nagp import Mom, Plan, _lib as L, ss as ssm
    c = CASES[name]; pr = pr or problem(name)
    blk = ssm.ss_blocks_nmf(pr['param1'], pr['param2'], 'matern32', 'matern52')
    if c['balance']:
        blk = ssm.balance_blocks(blk)
    kind = {'gf': L.KIND_GF_EP, 'ihgp': L.KIND_IHGP, 'giekf': L.KIND_GIEKF}[c['fam']]
    if c['fam'] == 'giekf':
        mom = None
    elif c.get('lik') == 'sqrt':
        from nagp import cubature
        wn, xn = cubature.sigma_points(c['p'], c['N'], True)
        mom = Mom('likModulatorPreCalcwn', link_shift=c['link_shift'], wn=wn, xn_unscaled=xn)
    else:
        mom = Mom('likModulatorNMFPower', p_cubature=c['p'])
    old = {}
    for k, v in (env or {}).items():
        old[k] = os.environ.get(k); os.environ[k] = v
    try:
        plan = Plan(kind, [(blk, pr['W'], np.log(pr['w_lik']))], pr['y'].size, mom=mom, ep_fraction=0.5, ep_damping=c.get('damping', 0.5) * np.ones(SWEEPS), ep_itts=SWEEPS, l_iter=1)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    plan.upload([pr['y']])
    t0 = time.perf_counter(); plan.execute(); dt = time.perf_counter() - t0
    out = plan.download(want_MS=False)[0]; plan.close()
    return out, dt


def _rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return float('inf')
    return float(np.nanmax(np.abs(a - b)) / (np.nanmax(np.abs(b)) + 1e-300)) if a.size else 0.0


def compare(name, out, ref):
    """Measured differences GPU vs sequential CPU algorithm; every entry is (value, tolerance)."""
    c = CASES[name]
    m = {'Eft': (_rel(out.Eft, ref['Eft']), TOL_MEAN), 'Varft': (_rel(out.Varft, ref['Varft']), TOL_MEAN)}
    if c['fam'] != 'giekf':
        nl = np.abs(out.nlZ - ref['nlZ']) / np.abs(ref['nlZ'])
        for i in range(SWEEPS):
            m['dlogZ/logZ sweep %d' % (i + 1)] = (float(nl[i]), TOL_LOGZ)
        fin = np.isfinite(ref['ttau'])
        # A site update is -d2 / (1 + d2 v): where 1 + d2 v = O(1e-9 .. 1e-15) its size and sign are rounding noise IN THE REFERENCE ALGORITHM
        # (tools/fuzz_conditioning.py: SITE_MAX = 1e8 -- e.g. ONE of the 7.6 M sites of cfg3sqrt, 1.3e9 where its neighbours are 1e-3 .. 18, and
        # the oracle moves it by 1.3e-4 under a 1e-13 relative change of y).  Such elements are counted and left out of the site comparison;
        # what they do to the marginals is in Eft / Varft, which are compared in full.
        wild = (np.abs(ref['ttau']) > SITE_MAX) | (np.abs(out.ttau) > SITE_MAX)
        tt_o, tt_r = np.where(wild, 0.0, out.ttau), np.where(wild, 0.0, ref['ttau'])
        tn_o, tn_r = np.where(wild, 0.0, out.tnu), np.where(wild, 0.0, ref['tnu'])
        m['sites beyond 1e8 (rounding noise of the reference; excluded below)'] = (float(np.sum(wild)), None)
        m['ttau'] = (_rel(tt_o, tt_r), TOL_SITE)
        # tnu is NaN at missing observations in the IHGP path (SURVEY C-3): the NaN pattern is part of the comparison
        m['tnu'] = (_rel(tn_o, tn_r), TOL_SITE)
        # the worst single site relative to ITS OWN size (informative: the stated tolerance is relative to the array)
        big = fin & ~wild & (np.abs(ref['ttau']) > 1e-6 * np.nanmax(np.abs(tt_r)))
        m['ttau worst element, relative to itself (informative)'] = (float(np.max(np.abs(out.ttau[big] - ref['ttau'][big]) / np.abs(ref['ttau'][big]))), None)
        if c['fam'] == 'ihgp':
            m['R: Inf pattern equal'] = (0.0 if np.array_equal(np.isinf(out.R), np.isinf(ref['R'])) else float('inf'), 0.5)
    else:
        m['maxDiffP'] = (_rel(out.maxDiffP, ref['maxDiffP']), 1e-6)
    m['maxDiff (convergence diagnostics)'] = (_rel(out.maxDiffP, ref['maxDiffP']), None)
    return m


def passed(m):
    return all(v <= tol for (v, tol) in m.values() if tol is not None)


def bit_equal(a, b, fields=('Eft', 'Varft', 'ttau', 'tnu', 'lZ', 'nlZ', 'maxDiffM', 'maxDiffP')):
    return [f for f in fields if not np.array_equal(getattr(a, f), getattr(b, f), equal_nan=True)]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--cases', default='cfg3,cfg2,cfg4,cfg5seg,cfg3sqrt,cfg2audio,cfg4audio')
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r05_full_length_parity.txt'))
    ap.add_argument('--audio', action='store_true', help='also the informative audio-input section (four more CPU legs)')
    a = ap.parse_args()
    names = [n for n in a.cases.split(',') if n]
    import nagp
    nagp.build()
    t_all = time.perf_counter()
    probs = {n: problem(n) for n in names}
    legs = CpuLegs(names, probs).start()
    la = lb = None
    if a.audio:      # the informative audio legs run beside the others from the start (eight more minutes of CPU otherwise)
        an = ['cfg2audio_d05', 'cfg2audio', 'cfg4audio']
        pa = {n: problem(n) for n in an}
        pb = {}
        for n in an:
            q = dict(pa[n]); q['y'] = q['y'] * (1.0 + 1e-13); pb[n] = q
        la = CpuLegs(an, pa).start(); lb = CpuLegs(an, pb).start()
    print('CPU legs started', flush=True)
    lines = ['# GPU (libnagp.so, source hash %s) against the compiled sequential restatement oracle/cpu/nagp_cpu.cpp (structured form),' % nagp._lib.source_hash(),
             '# the exact bench.py workloads: all %d sweeps, default chunking, pipelined schedule, parallel-in-time scans.' % SWEEPS,
             '# differences are max |gpu - cpu| / max |cpu| over the whole array unless stated; tolerance in brackets.', '']
    ok = True
    gpu = {}
    for n in names:
        gpu[n] = gpu_run(n, probs[n])
    if 'cfg2' in names:
        ser, _ = gpu_run('cfg2', probs['cfg2'], env={'NAGP_NO_PIPELINE': '1'})
        diff = bit_equal(gpu['cfg2'][0], ser)
        lines += ['cfg2, 3 sweeps, T = 84 010: pipelined schedule == serial schedule (NAGP_NO_PIPELINE=1), every output bit for bit: %s' % ('yes' if not diff else 'NO: ' + ','.join(diff)), '']
        ok = ok and not diff
    for n in names:
        c = CASES[n]
        ref = legs.result(n)
        print('joined', n, flush=True)
        m = compare(n, gpu[n][0], ref)
        lines.append('%s  (%s, T = %d, %d ch / %d comps; GPU execute %.2f s, CPU one core %.0f s)' % (n, c['fam'], c['T'], c['D'], c['N'], gpu[n][1], legs.secs[n]))
        for k, (v, tol) in m.items():
            lines.append('    %-58s %.3e%s' % (k, v, '' if tol is None else '   [%.0e] %s' % (tol, 'ok' if v <= tol else 'FAIL')))
        if c['fam'] != 'giekf':
            lines.append('    nlZ gpu %s' % np.array2string(gpu[n][0].nlZ, precision=12))
            lines.append('    nlZ cpu %s' % np.array2string(ref['nlZ'], precision=12))
        lines.append('')
        ok = ok and passed(m)
    if a.audio:
        # informative: the audio files BASELINE names with the bench's untrained hyper-parameters.  GPU - CPU next to the ORACLE's own movement
        # under y -> y (1 + 1e-13) (tools/fuzz_conditioning.py's criterion): where the reference algorithm itself is chaotic nothing can be asserted.
        lines += ['# CONDITIONING of the reference algorithm on the audio inputs (cfg2audio_d05 = the round-4 bench recipe, damping 0.5: nothing can be asserted on it).',
                  '# "oracle moves" = the sequential CPU algorithm on y against itself on y (1 + 1e-13): the conditioning of the reference algorithm on this input.', '']
        cond = {}
        for n in an:
            c = CASES[n]
            out, dt = gpu_run(n, pa[n])
            ra, rb = la.result(n), lb.result(n)
            lines.append('%s  (%s on audio/%s.wav, T = %d; GPU execute %.2f s)' % (n, c['fam'], c['audio'], c['T'], dt))
            keys = ('Eft', 'Varft') + (() if c['fam'] == 'giekf' else ('ttau', 'nlZ'))
            for k in keys:
                g = getattr(out, k)
                lines.append('    %-8s gpu - cpu %.3e      oracle moves %.3e      largest |value| %.3e' % (k, _rel(g, ra[k]), _rel(rb[k], ra[k]), float(np.nanmax(np.abs(ra[k])))))
            lines.append('    outputs finite: %s' % bool(np.all(np.isfinite(out.Eft)) and np.all(np.isfinite(out.Varft))))
            lines.append('')
            cond[n] = dict(fn={'gf': 'gf_ep_modulator_nmf', 'giekf': 'gf_giekf_modulator_nmf'}[c['fam']], audio=c['audio'] + '.wav', T=c['T'], ep_damping=c.get('damping', 0.5),
                           oracle_moves={k: _rel(rb[k], ra[k]) for k in keys}, gpu_minus_cpu={k: _rel(getattr(out, k), ra[k]) for k in keys})
        import json
        with open(os.path.join(ROOT, 'tests', 'golden', 'audio_conditioning.json'), 'w') as fh:      # bench.py prints it next to cfg2 / cfg4
            json.dump(dict(note='sequential CPU algorithm (oracle/cpu) on y against itself on y (1 + 1e-13): tools/full_length_parity.py --audio',
                           source_hash=nagp._lib.source_hash(), cases=cond), fh, indent=1, default=lambda o: o.tolist() if hasattr(o, 'tolist') else float(o))
    lines.append('# wall time of this script: %.0f s; all within tolerance: %s' % (time.perf_counter() - t_all, ok))
    txt = '\n'.join(lines) + '\n'
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    with open(a.out, 'w') as fh:
        fh.write(txt)
    print(txt)
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
