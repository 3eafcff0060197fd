"""Batched objective evaluation for hyper-parameter training.

The reference tunes hyper-parameters with `fminunc(@(w) gf_ep_modulator_nmf_constraints(w,t,y,ss,mom,[],...), w0, opts)` and
`GradObj = 'off'` (experiments/train_GTFNMF.m:186-201): every optimiser iteration evaluates the negative log marginal
likelihood at numel(w)+1 parameter vectors, one after the other.  Here all replicas of one iteration are ONE device-resident
plan (`nagp_plan_create(n_problems = replicas)`): the per-replica state-space models are built on the host, the sequential
ADF/EP recursions of the replicas run concurrently on different compute units.

    f  = nlml_batch(ws, t, y, ss, mom, kernel1, kernel2, num_lik_params, D, N, ep_fraction, ep_damping, ep_itts,
                    constraints, w_fixed, tune_hypers)                 # f[i] = gf_ep_modulator_nmf_constraints(ws[i], ...)
    f0, g = fd_value_and_gradient(w, ...same arguments...)             # forward differences, fminunc's default scheme
"""
import numpy as np

from . import _lib as L
from . import ss as ssm
from .api import _blocks_from_dense, _merge_inputs, _unpack_constraints, _unpack_log
from .plan import Plan


def nlml_batch(ws, x, y, ss, mom, kernel1, kernel2, num_lik_params, D, N, ep_fraction, ep_damping, ep_itts,
               constraints=None, w_fixed=None, tune_hypers=None, device=0, inference='EP'):
    """Negative log marginal likelihoods of gf_ep_modulator_nmf_constraints (constraints given) or gf_ep_modulator_nmf
    (constraints None) at every parameter vector of `ws`, in one batched GPU call (likelihood mode, xt = []).
    inference='EKF': the objective of the 'EKF' case of train_GTFNMF.m:198-201, gf_giekf_modulator_nmf_constraints with
    GradObj='off' (mom, ep_fraction, ep_damping, ep_itts unused)."""
    ws = [np.asarray(w, float).ravel() for w in ws]
    yall, _ = _merge_inputs(x, y, None)
    probs = []
    for w in ws:
        if constraints is not None:
            lik_param, p1, p2, Wnmf = _unpack_constraints(w, w_fixed, tune_hypers, constraints, num_lik_params, D, N)
            blk = ssm.balance_blocks(_blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N))     # balance ON (:115)
        else:
            lik_param, p1, p2, Wnmf = _unpack_log(w, num_lik_params, D, N)
            blk = _blocks_from_dense(*ss(x, p1, p2, kernel1, kernel2), D, N)
        probs.append((blk, Wnmf, lik_param))
    if inference == 'EKF':
        if constraints is None:
            raise NotImplementedError('the nlml branch of gf_giekf_modulator_nmf.m does not run in the reference')
        plan = Plan(L.KIND_GIEKF, probs, yall.size, ep_itts=1, l_iter=1, mode=L.MODE_NLML, flags=L.FLAG_EKF_RESET_P, device=device)
    else:
        plan = Plan(L.KIND_GF_EP, probs, yall.size, mom=mom, ep_fraction=ep_fraction, ep_damping=ep_damping, ep_itts=ep_itts,
                    mode=L.MODE_NLML, device=device)
    try:
        plan.upload([yall] * len(ws))
        plan.execute()
        return plan.download_nlz()[:, 0].copy()
    finally:
        plan.close()


def fd_value_and_gradient(w, *args, rel_step=None, **kw):
    """f(w) and its forward-difference gradient from ONE batched evaluation of numel(w)+1 replicas
    (fminunc 'forward' differences: step sqrt(eps)*max(|w_i|, 1) in coordinate i)."""
    w = np.asarray(w, float).ravel()
    h = (np.sqrt(np.finfo(float).eps) if rel_step is None else rel_step) * np.maximum(np.abs(w), 1.0)
    ws = [w] + [w + h[i] * (np.arange(w.size) == i) for i in range(w.size)]
    f = nlml_batch(ws, *args, **kw)
    return float(f[0]), (f[1:] - f[0]) / h
