"""Synthetic inputs for the BASELINE configs (the build's counterpart of the reference's demo drivers:
matlab/demo_toy_modulators.m:34-49, demo_toy_modulators_nmf.m:27-53, demo_toy_modulators_nmf_constraints.m:43-70).
NumPy PCG64 streams (MATLAB's randn stream cannot be reproduced), same generative equations:
  z_1 = chol(Pinf)' eps ; z_k = A z_{k-1} + chol(Q)' eps ; y_k = (H_z z)' W link(H_g z)   (noise free)
"""
import math

import numpy as np

from . import ss as ssm


def sample_prior(blk, W, T, rng, link_shift=0.0, sqrt_amp=False):
    """Block-wise prior sample of the (unbalanced) model; returns y (T,).  sqrt_amp: the generative model of
    experiments/likModulatorPreCalcwn.m (:44: amplitudes sqrt(W link(g))) instead of likModulatorNMFPower's (W link(g))."""
    A, Q, P = ssm.discretise(blk)
    S = blk.S; D, N = blk.D, blk.N
    cP = np.zeros((S, S)); cQ = np.zeros((S, S))
    for n in range(blk.M):
        o, e = blk.offsets[n], blk.offsets[n + 1]
        cP[o:e, o:e] = np.linalg.cholesky(P[o:e, o:e])
        q = Q[o:e, o:e]
        cQ[o:e, o:e] = np.linalg.cholesky((q + q.T) / 2)
    zi = blk.offsets[:D]; gi = blk.offsets[D:D + N]
    y = np.empty(T)
    z = cP @ rng.standard_normal(S)
    eps = rng.standard_normal((T, S))
    Wm = np.eye(D) if W is None else W
    for k in range(T):
        if k > 0:
            z = A @ z + cQ @ eps[k]
        amp = Wm @ np.log1p(np.exp(z[gi] - link_shift))
        y[k] = z[zi] @ (np.sqrt(amp) if sqrt_amp else amp)
    return y


def cfg1(T=1000, seed=123):
    """demo_toy_modulators.m hyper-parameters cycled to D=4: gf_ep_modulator."""
    D = 4
    var_fast = np.full(D, 0.1); len_fast = np.array([50.0, 40.0, 50.0, 40.0])
    omega = np.array([math.pi / 4, math.pi / 6, math.pi / 8, math.pi / 10])
    var_slow = np.array([2.0, 3.0, 2.0, 3.0]); len_slow = np.array([500.0, 700.0, 500.0, 700.0])
    param = np.concatenate([var_fast, len_fast, omega, var_slow, len_slow])
    blk = ssm.ss_blocks_nmf(param[:3 * D], param[3 * D:], 'matern32', 'matern52')
    y = sample_prior(blk, None, T, np.random.default_rng(seed))
    w = np.log(np.concatenate([[1e-5], param]))
    return dict(w=w, y=y, D=D, kernel1='matern32', kernel2='matern52', p=9, ep_fraction=0.5, ep_itts=5,
                ep_damping=0.3 * np.ones(5))


def nmf_params(D, N, seed, recipe='demo_nmf'):
    rng = np.random.default_rng(seed)
    if recipe == 'demo_nmf':      # demo_toy_modulators_nmf.m:27-33
        len_fast = 150 + 400 * rng.random(D); var_fast = 0.01 * np.ones(D)
        omega = np.linspace(math.pi / 3, math.pi / 50, D); len_slow = np.linspace(200, 1500, N)
        var_slow = 5 + 5 * rng.random(N); W = 0.1 * np.abs((2 * rng.random((D, N))) ** 2 - 0.2)
    else:                         # demo_toy_modulators_nmf_constraints.m:27-49
        var_fast = 0.055 * np.ones(D); len_fast = 20 + 480 * rng.random(D)
        omega = np.linspace(math.pi / 4, math.pi / 50, D); var_slow = 3.5 * np.ones(N)
        len_slow = np.linspace(202, 1998, N); W = (2.0 / D) * rng.random((D, N))
    return var_fast, len_fast, omega, var_slow, len_slow, W


def nmf_problem(D, N, T, seed, recipe='demo_nmf', w_lik=1e-4, kernel1='matern32', kernel2='matern52', link_shift=0.0, sqrt_amp=False):
    """Synthetic GT-NMF problem of a given shape: log-parameter vector w (gf_ep_modulator_nmf.m:72-75
    packing) and a prior-sampled signal normalised to unit variance."""
    vf, lf, om, vs, ls, W = nmf_params(D, N, seed, recipe)
    blk = ssm.ss_blocks_nmf(np.concatenate([vf, lf, om]), np.concatenate([vs, ls]), kernel1, kernel2)
    y = sample_prior(blk, W, T, np.random.default_rng(seed + 7919), link_shift=link_shift, sqrt_amp=sqrt_amp)
    w = np.log(np.concatenate([[w_lik], vf, lf, om, vs, ls, W.flatten(order='F')]))
    return dict(w=w, y=y, D=D, N=N, W=W, kernel1=kernel1, kernel2=kernel2,
                param1=np.concatenate([vf, lf, om]), param2=np.concatenate([vs, ls]), w_lik=w_lik)


def mixture_problem(shapes, T, seed, kernel1, kernel2, w_lik=1e-4):
    """J sources of shapes [(D_j, N_j)]: the cell w = {log sn2, {param1_j}, {param2_j}, {W_j}} of
    experiments/source_sep_piano.m:128 (natural units inside the cells) and the sum of one prior sample per source,
    normalised to unit variance."""
    p1s, p2s, Ws = [], [], []
    y = np.zeros(T)
    for j, (D_, N_) in enumerate(shapes):
        vf, lf, om, vs, ls, W = nmf_params(D_, N_, seed + 31 * j)
        p1s.append(np.concatenate([vf, lf, om])); p2s.append(np.concatenate([vs, ls])); Ws.append(W)
        blk = ssm.ss_blocks_nmf(p1s[-1], p2s[-1], kernel1[j], kernel2[j])
        y = y + sample_prior(blk, W, T, np.random.default_rng(seed + 7919 + j))
    y = y / np.sqrt(np.var(y))
    return dict(w=[np.array([math.log(w_lik)]), p1s, p2s, Ws], y=y, J=len(shapes), kernel1=list(kernel1), kernel2=list(kernel2))


CONSTRAINTS_DEMO = lambda D: np.array([[0.01, 0.1], [20.0, 500.0], [0.0, 2 * math.pi], [2.0, 5.0], [200.0, 2000.0], [0.0, 2.0 / D]])
TUNE_DEMO = [0, 0, 1, 0, 1, 1, 0]


def constrained_vectors(prob, constraints, tune_hypers):
    """w / w_fixed split of demo_toy_modulators_nmf_constraints.m:96-104."""
    D, N = prob['D'], prob['N']
    p1, p2, W = prob['param1'], prob['param2'], prob['W']
    w_all = [np.array([math.log(prob['w_lik'])]), ssm.inv_sigmoid(p1[:D], constraints[0]), ssm.inv_sigmoid(p1[D:2 * D], constraints[1]),
             ssm.inv_sigmoid(p1[2 * D:], constraints[2]), ssm.inv_sigmoid(p2[:N], constraints[3]),
             ssm.inv_sigmoid(p2[N:], constraints[4]), ssm.inv_sigmoid(W.flatten(order='F'), constraints[5])]
    w = [v for v, t in zip(w_all, tune_hypers) if t]
    wf = [v for v, t in zip(w_all, tune_hypers) if not t]
    return (np.concatenate(w) if w else np.zeros(0)), (np.concatenate(wf) if wf else np.zeros(0))
