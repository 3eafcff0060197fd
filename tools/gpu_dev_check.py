"""Developer diagnostic (not a test): run every path on the GPU against the oracle and print the
differences without asserting.  Usage on the GPU box: python tools/gpu_dev_check.py"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'nonstationary-audio-gp_amd'))
import numpy as np
import nagp
from nagp import harness, Mom, SSHandle
from oracle import gf_ep as ogf, ihgp as oih, giekf as oek, lik as olik

nagp.build()
print('lib version', nagp.lib().nagp_version(), 'devices', nagp.lib().nagp_device_count())


def rel(a, b):
    a = np.asarray(a, float); b = np.asarray(b, float)
    den = np.nanmax(np.abs(b)) + 1e-300
    return float(np.nanmax(np.abs(a - b)) / den)


def report(name, out, ref, keys):
    for k in keys:
        try:
            print('  %-10s rel.diff %.3e   (|ref|max %.3e)' % (k, rel(out[k], ref[k]), np.nanmax(np.abs(ref[k]))))
        except Exception as e:
            print('  %-10s ERR %s' % (k, e))


def run(name, f):
    print('== ' + name); t0 = time.time()
    try:
        f()
    except Exception:
        traceback.print_exc()
    print('   (%.1fs)' % (time.time() - t0)); sys.stdout.flush()


def t_cfg1():
    c = harness.cfg1(T=300)
    t = np.arange(1, c['y'].size + 1.0)
    mom = Mom('likModulatorPower', p_cubature=9); omom = olik.Mom(olik.LIK_POWER, p=9)
    r = nagp.gf_ep_modulator(c['w'], t, c['y'], SSHandle('ss_modulators'), mom, t, 'matern32', 'matern52', 1, 0.5, c['ep_damping'], 5, nargout=6)
    o = ogf.gf_ep_modulator(c['w'], t, c['y'], None, omom, t, 'matern32', 'matern52', 1, 0.5, c['ep_damping'], 5)
    out = r[5]; ref = o[5]; ref['PS'] = np.transpose(ref['PS'], (1, 2, 0))
    print('  nlZ gpu', out['nlZ']); print('  nlZ ref', ref['nlZ']); print('  counters', out['counters'])
    report('cfg1', dict(out, Eft=r[0], Varft=r[1]), dict(ref, Eft=o[0], Varft=o[1]),
           ['Eft', 'Varft', 'nlZ', 'ttau', 'tnu', 'lZ', 'MS', 'PS', 'maxDiffM', 'maxDiffP'])
    e, _ = nagp.gf_ep_modulator(c['w'], t, c['y'], SSHandle('ss_modulators'), mom, None, 'matern32', 'matern52', 1, 0.5, c['ep_damping'], 3)
    eo, _ = ogf.gf_ep_modulator(c['w'], t, c['y'], None, omom, None, 'matern32', 'matern52', 1, 0.5, c['ep_damping'], 3)
    print('  nlml I=3: gpu %.10g ref %.10g rel %.2e' % (e, eo, abs(e - eo) / abs(eo)))


def t_nmf(D=6, N=2, T=400, p=9, miss=True):
    pr = harness.nmf_problem(D, N, T, 100)
    y = pr['y'].copy()
    if miss:
        y[50:70] = np.nan
    t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorNMFPower', p_cubature=p); omom = olik.Mom(olik.LIK_POWER_NMF, p=p)
    d = 0.5 * np.ones(3)
    r = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3, nargout=6)
    o = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, omom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3)
    out = r[5]; ref = o[5]; ref['PS'] = np.transpose(ref['PS'], (1, 2, 0))
    print('  nlZ gpu', out['nlZ']); print('  nlZ ref', ref['nlZ']); print('  counters', out['counters'])
    report('nmf', dict(out, Eft=r[0], Varft=r[1]), dict(ref, Eft=o[0], Varft=o[1]),
           ['Eft', 'Varft', 'nlZ', 'ttau', 'tnu', 'lZ', 'MS', 'PS', 'maxDiffM', 'maxDiffP'])
    e, _ = nagp.gf_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, None, 'matern32', 'matern52', 1, D, N, 0.5, d, 1)
    eo, _ = ogf.gf_ep_modulator_nmf(pr['w'], t, y, None, omom, None, 'matern32', 'matern52', 1, D, N, 0.5, d, 1)
    print('  nlml I=1: gpu %.10g ref %.10g rel %.2e' % (e, eo, abs(e - eo) / abs(eo)))


def t_ihgp(D=6, N=2, T=400, p=7):
    pr = harness.nmf_problem(D, N, T, 101)
    y = pr['y'].copy(); y[100:110] = np.nan
    t = np.arange(1, T + 1.0)
    mom = Mom('likModulatorNMFPower', p_cubature=p); omom = olik.Mom(olik.LIK_POWER_NMF, p=p)
    d = 0.5 * np.ones(3)
    r = nagp.ihgp_ep_modulator_nmf(pr['w'], t, y, SSHandle(), mom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3, nargout=6)
    o = oih.ihgp_ep_modulator_nmf(pr['w'], t, y, None, omom, t, 'matern32', 'matern52', 1, D, N, 0.5, d, 3)
    out = r[5]; ref = o[5]
    print('  nlZ gpu', out['nlZ']); print('  nlZ ref', ref['nlZ']); print('  counters', out['counters'])
    Rg = np.where(np.isinf(out['R']), 1e300, out['R']); Rr = np.where(np.isinf(ref['R']), 1e300, ref['R'])
    report('ihgp', dict(out, Eft=r[0], Varft=r[1], R=np.log10(np.abs(Rg) + 1e-300)), dict(ref, Eft=o[0], Varft=o[1], R=np.log10(np.abs(Rr) + 1e-300)),
           ['Eft', 'Varft', 'nlZ', 'ttau', 'tnu', 'R', 'MS', 'maxDiffM', 'maxDiffP'])


def t_giekf(D=6, N=2, T=400):
    pr = harness.nmf_problem(D, N, T, 102)
    y = pr['y'].copy(); y[30:40] = np.nan
    t = np.arange(1, T + 1.0)
    r = nagp.gf_giekf_modulator_nmf(pr['w'], t, y, SSHandle(), None, t, 'matern32', 'matern52', 1, D, N, 3, 2, nargout=6)
    o = oek.gf_giekf_modulator_nmf(pr['w'], t, y, None, None, t, 'matern32', 'matern52', 1, D, N, 3, 2)
    out = r[5]; ref = o[5]; ref['PS'] = np.transpose(ref['PS'], (1, 2, 0))
    report('giekf', dict(out, Eft=r[0], Varft=r[1]), dict(ref, Eft=o[0], Varft=o[1]), ['Eft', 'Varft', 'MS', 'PS', 'maxDiffP'])


run('cfg1 gf_ep_modulator', t_cfg1)
run('gf_ep_modulator_nmf D6 N2', t_nmf)
run('gf_ep_modulator_nmf D16 N3 p7', lambda: t_nmf(16, 3, 300, 7))
run('ihgp D6 N2', t_ihgp)
run('giekf D6 N2', t_giekf)
