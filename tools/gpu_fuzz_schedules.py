"""Randomised check of the chunk-pipelined smoother schedule (developer tool): random shape, family, number of segments, chunk length,
number of (G, Delta) buffers, sweeps, missing data; every output of the pipelined plan (with and without the cross-sweep form, NAGP_NO_XSWEEP=1) must equal the serial plan's bit for bit
(NAGP_NO_PIPELINE=1), and the serial plan is the one the oracle tests pin.
python tools/gpu_fuzz_schedules.py [n_cases] [seed] [--k1 matern52]"""
import os, sys, time
os.environ.setdefault('NAGP_DEVELOPER', '1')      # developer tool: libnagp.so reads its switches only with this set
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, Plan, _lib as L, ss as pss

def run_cases(n, seed, verbose=True, k1='matern32'):
    """-> number of cases whose pipelined outputs differ from the serial ones"""
    rng = np.random.default_rng(seed)
    bad = 0; t0 = time.time()
    for case in range(n):
        ekf = rng.random() < 0.25
        D = int(rng.choice([3, 8, 12, 16, 20, 24, 32, 36])); N = int(rng.integers(1, 7 if D >= 12 else 4))
        if k1 != 'matern32': D = min(D, 20)      # blocks of six / eight states are split over two tile rows: sites + split blocks fit the filter's LDS up to ~54
        if D == 36: N = min(N, 8)
        T = int(rng.integers(40, 700)); B = int(rng.choice([1, 1, 2, 3, 5, 8]))
        chunk = int(rng.choice([0, 8, 16, 24, 50, 100, 200]))
        slots = int(rng.choice([0, 2, 3, 4, 5, 6, 8])); no_recycle = rng.random() < 0.3
        itts = int(rng.integers(2, 4)); p = int(rng.choice([3, 5])) if N > 3 else int(rng.choice([3, 5, 7]))
        want_ps = (not ekf) and rng.random() < 0.25
        nlml = (not ekf) and (not want_ps) and rng.random() < 0.25      # energy mode: sweeps 1 .. I-1 filter + smooth + refresh, no filter in sweep I
        probs, ys = [], []
        for q in range(B):
            pr = harness.nmf_problem(D, N, T, int(rng.integers(1, 1 << 30)), 'constraints', kernel1=k1)
            blk = pss.balance_blocks(pss.ss_blocks_nmf(pr['param1'], pr['param2'], k1, 'matern52'))
            y = pr['y'].copy()
            if rng.random() < 0.5: y[rng.integers(0, T, size=max(1, T // 40))] = np.nan
            probs.append((blk, pr['W'], np.log(pr['w_lik']))); ys.append(y)
        kw = dict(ep_itts=itts, l_iter=2) if ekf else dict(mom=Mom('likModulatorNMFPower', p_cubature=p), ep_fraction=0.5, ep_damping=0.5 * np.ones(itts), ep_itts=itts,
                                                            flags=L.FLAG_WANT_PS if want_ps else 0, mode=L.MODE_NLML if nlml else L.MODE_PREDICT)
        res = {}
        penv = dict({'NAGP_PIPELINE_SLOTS': str(slots)} if slots else {}, **({'NAGP_NO_RECYCLE': '1'} if no_recycle else {}))
        # 'pipelined' = the default schedule (cross-sweep form whenever every chunk owns a buffer), 'one_sweep' = the same without it
        for name, env in (('pipelined', penv), ('one_sweep', dict(penv, NAGP_NO_XSWEEP='1')), ('serial', {'NAGP_NO_PIPELINE': '1'})):
            os.environ.update(env)
            try:
                plan = Plan(L.KIND_GIEKF if ekf else L.KIND_GF_EP, probs, T, chunk=chunk, **kw)
                plan.upload(ys)
                try:
                    plan.execute(); st = 'ok'
                except nagp.NagpError as e:
                    st = str(e)[:60]
                res[name] = (plan.download(want_PS=want_ps, want_MF=True), st); plan.close()
            finally:
                for k in env: os.environ.pop(k, None)
        fields = ('Eft', 'Varft', 'MS', 'MF', 'maxDiffP') + (() if ekf else ('ttau', 'tnu', 'R', 'lZ', 'nlZ', 'maxDiffM')) + (('PS',) if want_ps else ())
        diff = [(q, f, v) for v in ('pipelined', 'one_sweep') for q in range(B) for f in fields
                if not np.array_equal(getattr(res[v][0][q], f), getattr(res['serial'][0][q], f), equal_nan=True)]
        same_status = res['pipelined'][1] == res['serial'][1] == res['one_sweep'][1]
        ok = not diff and same_status
        bad += (not ok)
        if verbose: print('case %2d %s D=%2d N=%d T=%3d B=%d chunk=%3d slots=%d sweeps=%d p=%d PS=%d: %s%s' % (
            case, 'ekf' if ekf else ('nlm' if nlml else 'gf '), D, N, T, B, chunk, slots, itts, p, want_ps, 'bit-equal' if ok else 'DIFFERENT %s' % diff[:4],
            '' if res['serial'][1] == 'ok' else ' [status: %s / %s]' % (res['pipelined'][1], res['serial'][1]))); sys.stdout.flush()
    if verbose: print('%d cases, %d different, %.0f s' % (n, bad, time.time() - t0))
    return bad


if __name__ == '__main__':
    k1 = 'matern32'
    if '--k1' in sys.argv:      # python tools/gpu_fuzz_schedules.py 40 7 --k1 matern52 : the same draws with six-state sub-band blocks
        i = sys.argv.index('--k1'); k1 = sys.argv[i + 1]; del sys.argv[i:i + 2]
    sys.exit(1 if run_cases(int(sys.argv[1]) if len(sys.argv) > 1 else 20, int(sys.argv[2]) if len(sys.argv) > 2 else 0, k1=k1) else 0)
