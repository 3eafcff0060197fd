/*
 * nagp.h -- C ABI of libnagp.so: MI355X-native (gfx950, HIP) Kalman-filter / RTS-smoother /
 * Power-EP / iterated-EKF inference loops of the GT-NMF audio model.
 *
 * The reference (AaltoML/nonstationary-audio-gp) is 100 % MATLAB and has no FFI; the entry points
 * below are what a MEX gateway (or ctypes / any FFI) binds to replace the per-time-step loops of
 *
 *   nagp_*_KIND_GF_EP   -> matlab/gf_ep_modulator.m:113-283 (predict) / :383-522 (nlml)
 *                          matlab/gf_ep_modulator_nmf.m:113-283 / :384-522
 *                          matlab/gf_ep_modulator_nmf_constraints.m:148-334 / :436-573
 *   nagp_*_KIND_IHGP    -> matlab/ihgp_ep_modulator_nmf.m:223-454,
 *                          matlab/ihgp_ep_modulator_nmf_constraints.m:257-480
 *   nagp_*_KIND_GIEKF   -> matlab/gf_giekf_modulator_nmf.m:126-221 (+ iekf_update1.m:110-117,
 *                          ekf_update1.m:106-109), matlab/gf_giekf_modulator_nmf_constraints.m:162-257
 *   mom callbacks       -> matlab/likModulatorPower.m:25-100, likModulatorNMFPower.m:28-87,
 *                          experiments/likModulatorPreCalcwn.m:28-86 (selected by lik_kind)
 *
 * Everything outside those loops (parameter unpacking, ss_modulators*, balance, lti_disc, DARE
 * tables, sigma-point tables) stays on the host side of the boundary and arrives here as plain
 * arrays.  All matrices are column-major IEEE doubles (MATLAB layout); all pointers are HOST
 * pointers owned by the caller; inputs are read-only; outputs are caller-allocated and may be
 * NULL (= not wanted).  No exceptions cross the ABI: every function returns 0 or a negative
 * nagp_status.  Single caller thread per plan; safe to call repeatedly from a long-lived process.
 */
#ifndef NAGP_H
#define NAGP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NAGP_VERSION 300 /* 0.3.0 */

typedef enum nagp_status {
  NAGP_OK = 0,
  NAGP_EINVAL = -1,       /* bad argument / inconsistent sizes */
  NAGP_EUNSUPPORTED = -2, /* shape outside what the kernels handle (e.g. block size > 4) */
  NAGP_EHIP = -3,         /* HIP runtime error (see nagp_last_error) */
  NAGP_ENOMEM = -4,       /* device memory exhausted */
  NAGP_ENODEVICE = -5,    /* no gfx950 device visible */
  NAGP_ENOTPD = -6,       /* Cholesky failed even after the jitter retry (where MATLAB's chol throws, gf_ep_modulator_nmf.m:219-222);
                             returned by the execute / run calls after the sweeps have finished: outputs can still be downloaded */
  NAGP_ERCCL = -7         /* RCCL failure in nagp_batch_run (see nagp_last_error) */
} nagp_status;

typedef enum nagp_kind { NAGP_KIND_GF_EP = 0, NAGP_KIND_IHGP = 1, NAGP_KIND_GIEKF = 2 } nagp_kind;
typedef enum nagp_mode { NAGP_MODE_PREDICT = 0, NAGP_MODE_NLML = 1 } nagp_mode;
/* mom callbacks of the reference, by file */
typedef enum nagp_lik {
  NAGP_LIK_POWER = 0,          /* likModulatorPower.m        (W = I, jitter 1e-8)  */
  NAGP_LIK_POWER_NMF = 1,      /* likModulatorNMFPower.m     (jitter 1e-10, pEP_const = 1) */
  NAGP_LIK_POWER_NMF_SQRT = 2  /* experiments/likModulatorPreCalcwn.m (sqrt amplitude, true pEP_const) */
} nagp_lik;
/* link functions seen in the drivers: log(1+exp(g-shift)) and exp(g) */
typedef enum nagp_link { NAGP_LINK_SOFTPLUS = 0, NAGP_LINK_EXP = 1 } nagp_link;

/* flags (nagp_opts.flags) */
#define NAGP_FLAG_IHGP_CONSTRAINTS 0x1u /* ihgp_ep_modulator_nmf_constraints.m: R starts at 0, no abs(Varft) */
#define NAGP_FLAG_EKF_RESET_P      0x2u /* gf_giekf_modulator_nmf_constraints.m:168: P=Pinf every global iteration */
#define NAGP_FLAG_MIXTURE_RULE     0x8u /* experiments/{gf,ihgp}_ep_mods_nmf_mixture.m: mom at power ep_fraction in the filter too,
                                           site <- (1-d) site + d/ep_fraction (...), clamp in the filter pass only, R starts at 0 */
#define NAGP_FLAG_WANT_PS          0x4u /* keep the smoothed covariances so that nagp_out.PS can be filled */

/* Discrete-time model of ONE problem (segment / hyper-parameter replica).
 * State ordering and block structure as built by ss_modulators_nmf.m:128-132: M diagonal blocks,
 * block n spanning states block_offsets[n] .. block_offsets[n+1]-1 (0-based); A, Q, Pinf are
 * block-diagonal with these blocks (only the diagonal blocks are read); row n of H has its single
 * non-zero h_val[n] at column block_offsets[n] (1 before `balance`, a power of two after).
 * Blocks hold 1 .. 8 states: 1 .. 4 for the kernels every driver uses (exp / Matern-3/2 sub-bands, Matern-5/2 modulators), 6 or 8 for
 * Matern-5/2 / -7/2 sub-bands (ss_modulators_nmf.m:13-33, cf_matern52_to_ss.m:93-121, cf_matern72_to_ss.m:93-124).  The state order of
 * every input and output is the caller's; how blocks of more than four states are laid out on the device is DESIGN.md section 3.
 * Larger blocks (the SE kernel's 12-state form) are refused with NAGP_EUNSUPPORTED. */
typedef struct nagp_model {
  int32_t S;                    /* state dimension */
  int32_t M;                    /* sites per step: D+N (NMF) or 2*D (gf_ep_modulator) */
  int32_t D;                    /* sub-bands */
  int32_t N;                    /* modulators / NMF components */
  const int32_t* block_offsets; /* M+1 */
  const double* A;              /* S x S */
  const double* Q;              /* S x S */
  const double* Pinf;           /* S x S */
  const double* h_val;          /* M */
  const double* Wnmf;           /* D x N, or NULL for NAGP_LIK_POWER */
  double lik_param;             /* w(1): log observation-noise variance */
} nagp_model;

/* IHGP look-up tables of one problem (ihgp_ep_modulator_nmf.m:107-134, 150-191), n_grid rows each,
 * channel-major: PPlist[n] is n_grid x b_n^2 (column-major b x b per row, as MATLAB's PP(:)'),
 * PGlist[n] is n_grid x 2 b_n^2 = [PS2(:)' G(:)'].  Stored flat: channel n starts at
 * pp_offsets[n] / pg_offsets[n] (in doubles), row stride b_n^2 / 2 b_n^2. */
typedef struct nagp_ihgp_tables {
  int32_t n_grid;          /* 200 in the reference */
  const double* r_grid;    /* n_grid ascending (logspace(-2,4,200)) */
  const double* PPlist;
  const int64_t* pp_offsets; /* M */
  const double* PGlist;
  const int64_t* pg_offsets; /* M */
} nagp_ihgp_tables;

typedef struct nagp_opts {
  int32_t kind;            /* nagp_kind */
  int32_t mode;            /* nagp_mode (NLML: GF_EP; GIEKF = the GradObj=off energy of gf_giekf_modulator_nmf_constraints.m:332-480, model Q = Pinf-A*Pinf*A') */
  int32_t lik_kind;        /* nagp_lik */
  int32_t link_kind;       /* nagp_link */
  double link_shift;       /* softplus shift (mod_sparsity) */
  int32_t n_pts;           /* sigma points */
  int32_t cub_dim;         /* N (NMF) or D (POWER) */
  const double* wn;        /* n_pts weights  (utp_ws / mvhermgauss) */
  const double* xn_unscaled; /* cub_dim x n_pts unit sigma points, column-major */
  double ep_fraction;      /* power-EP alpha */
  int32_t ep_itts;         /* EP sweeps (GIEKF: g_iter) */
  const double* ep_damping; /* ep_itts values */
  int32_t l_iter;          /* GIEKF inner iterations (iekf_update1 `iters`) */
  int32_t predict_at_k1;   /* gf_ep_modulator.m:131-133 predicts at k=1 in predict mode */
  uint32_t flags;
  int32_t device;          /* HIP device ordinal */
  int32_t chunk;           /* smoother chunk length (0 = default) */
  /* Warm start (one-shot entry points; plans: nagp_plan_upload_sites): initial site parameters, M x T column-major, in
   * place of the reference's zeros (gf_ep_modulator_nmf.m:96-97) -- e.g. the ttau / tnu a previous call returned
   * (the `out` struct of the reference carries them for this purpose).  NULL = zeros. */
  const double* ttau0;
  const double* tnu0;
} nagp_opts;

/* Caller-allocated outputs of ONE problem; any pointer may be NULL. */
typedef struct nagp_out {
  double* Eft;      /* M x T   H*MS                                  */
  double* Varft;    /* M x T   diag(H*PS_k*H')  (IHGP: time-constant) */
  double* MS;       /* S x T   smoothed means                        */
  double* PS;       /* S x S x T smoothed covariances (GF_EP / GIEKF) */
  double* ttau;     /* M x T */
  double* tnu;      /* M x T */
  double* R;        /* M x T */
  double* lZ;       /* T      (GF_EP: per-step log Z as left by the last pass) */
  double* nlZ;      /* ep_itts (predict) ; nlZ[0] = edata in NLML mode */
  double* maxDiffM; /* ep_itts */
  double* maxDiffP; /* ep_itts */
  int64_t* counters; /* NAGP_N_COUNTERS: chol retries, clamped sites, NaN observations, not-PD */
  double* MF;       /* S x T   filtered means of the last forward pass (the reference's out.MF, gf_ep_modulator_nmf.m:191-198) */
} nagp_out;

#define NAGP_N_COUNTERS 4
#define NAGP_CNT_CHOL_RETRY 0
#define NAGP_CNT_CLAMPED 1
#define NAGP_CNT_NAN_OBS 2
#define NAGP_CNT_NOTPD 3

/* Per-kernel device time of the last nagp_plan_execute, measured with HIP events on the plan's
 * own stream (ms, summed over launches) and launch counts. */
#define NAGP_N_KERNELS 8
#define NAGP_K_FILTER 0      /* gf/ekf forward filter  | ihgp ADF filter          */
#define NAGP_K_GAIN 1        /* RTS gain (PSkp, Cholesky, G)                       */
#define NAGP_K_SCAN 2        /* RTS backward recursion | ihgp backward mean scan  */
#define NAGP_K_EPSITE 3      /* cavity + mom + site update (parallel over k)      */
#define NAGP_K_REDUCE 4      /* nlZ / maxDiff reductions                          */
#define NAGP_K_FILTER_LIN 5  /* ihgp fixed-site (linear) filter of sweeps >= 2    */
#define NAGP_K_OUTPUT 6      /* output formatting (tile-major -> column-major)    */
#define NAGP_K_OTHER 7
typedef struct nagp_timings {
  double ms[NAGP_N_KERNELS];
  int64_t launches[NAGP_N_KERNELS];
  double total_ms; /* whole execute, events on the same stream */
} nagp_timings;

typedef struct nagp_plan nagp_plan; /* opaque */

int nagp_version(void);
int nagp_device_count(void);
const char* nagp_strerror(int status);
const char* nagp_last_error(void); /* text of the last HIP failure on this thread */

/* One-shot host-buffer entry points (create + upload + execute + download + destroy).  The three
 * names mirror the reference's function families; `tables` only for IHGP. */
int nagp_ep_run(const nagp_model* model, const double* y, int64_t T, const nagp_opts* opts, nagp_out* out);
int nagp_ihgp_run(const nagp_model* model, const nagp_ihgp_tables* tables, const double* y, int64_t T,
                  const nagp_opts* opts, nagp_out* out);
int nagp_giekf_run(const nagp_model* model, const double* y, int64_t T, const nagp_opts* opts, nagp_out* out);

/* The `mom` callback itself for n_eval independent inputs -- replaces likModulatorPower.m:25-100,
 * likModulatorNMFPower.m:28-87 and experiments/likModulatorPreCalcwn.m:28-86 as called through the handles of
 * demo_toy_modulators.m:81 / demo_toy_modulators_nmf.m:81 / train_GTFNMF.m:149:
 *   [lZ, dlZ, d2lZ] = mom(hyp, mu, s2, [Wnmf,] ep_frac, yall, k)
 * opts supplies lik_kind, link, cubature (n_pts, cub_dim, wn, xn_unscaled), ep_fraction and device; D, N as in
 * nagp_model (M = D+N for the NMF likelihoods, 2*D for NAGP_LIK_POWER); Wnmf is D x N column-major (NULL for POWER);
 * lik_param = hyp = log observation-noise variance.  y[n_eval]; mu, s2, dlZ, d2lZ are M x n_eval column-major. */
int nagp_mom_eval(const nagp_opts* opts, int32_t D, int32_t N, const double* Wnmf, double lik_param, int64_t n_eval,
                  const double* y, const double* mu, const double* s2, double* lZ, double* dlZ, double* d2lZ);

/* [M,P,K,MU,S] = iekf_update1(M,P,y,H,R,h,V,param,iters) -- iekf_update1.m:110-117 (ekf_update1.m:106-109 is
 * iters = 1) for the measurement model the reference passes through its handles (funh / funhd,
 * gf_giekf_modulator_nmf_constraints.m:492-502): h(x) = (H_z x)' W softplus(H_g x).  Row n of H has its single
 * non-zero h_val[n] at (0-based) column h_col[n]; n < D: sub-bands, then N modulators.  m (S) and P (S x S
 * column-major) are updated in place; K (S), *MU, *Sinn receive the last iteration's gain, prediction and innovation
 * variance (any of the three may be NULL). */
int nagp_iekf_update1(int32_t S, int32_t D, int32_t N, const int32_t* h_col, const double* h_val, const double* Wnmf,
                      double R, double y, int32_t iters, double* m, double* P, double* K, double* MU, double* Sinn,
                      int32_t device);

/* The EKF training objective WITH its gradient recursion -- matlab/gf_giekf_modulator_nmf_constraints.m:332-480 with GradObj = 'on'
 * (gf_giekf_modulator_nmf.m:296-439): one plain EKF pass (as NAGP_MODE_NLML of the GIEKF kind: prediction at the first step too,
 * one update per step, no isnan guard) and, per parameter slice j, the sensitivity recursion of (m, P) (:387-401, :437-466).
 * models[q].A = expm(F), models[q].Q = Pinf - A*Pinf*A' (:377-378).  Per problem and slice, S x S column-major, block diagonal with
 * the blocks of the model (only those blocks are read): dA[q][j] = lower-left block of expm([F 0; dF_j F]) (:355-366),
 * dQ[q][j] = dPinf_j - dA_j*Pinf*A' - A*dPinf_j*A' - (dA_j*Pinf*A')' (:392-394), dPinf[q][j]; dR[j] (:124).
 * What the Jacobian derivative `dmdJH` of slice j is made of is data: hess[j] != 0 adds dm_j' * d2h (:439), w_index[j] >= 0 adds
 * dh(.; W_) with W_ the unit matrix at column-major position w_index[j] of Wnmf (:441-443), w_direct[j] != 0 adds h(.; W_) to the
 * derivative of the predicted measurement (a term the reference's statements leave out).  The reference as written:
 * hess = 1, w_index = -1 for j < n_param - D*N and hess = 0, w_index = j - (n_param - D*N) for the last D*N slices, w_direct = 0.
 * edata[q], gdata[q * n_param + j]; NaN for a problem whose innovation variance is not positive even with the jitter (:423-426).
 * Shapes: M = D + N <= 32 sites, blocks of <= 4 states. */
int nagp_giekf_nlml_grad(int32_t n_problems, const nagp_model* models, const double* const* ys, int64_t T, int32_t n_param,
                         const double* const* dA, const double* const* dQ, const double* const* dPinf, const double* dR,
                         const int32_t* hess, const int32_t* w_index, const int32_t* w_direct, double* edata, double* gdata,
                         int32_t device);

/* Stationary filterbank (the step before the hot path in every real-audio script, e.g. train_GTFNMF.m:56-65):
 * the two loops of unifying_prob_tf/kernel_ss_kalmanFastFB.m -- infinite-horizon Kalman filter (:83-110)
 *     if ~isnan(y_k): v = y_k - HA*m; m = AKHA*m + K*y_k; else m = A*m;   MS(:,k) = m
 * and steady-state RTS smoother (:134-151)   m = MS(:,k) + G*(m - A*MS(:,k)),  k = T-1 .. 1.
 * The caller keeps the set-up lines of the .m (dare, K, AKHA = A-K*H*A, HA = H*A, G = PF2*A'/PP) and passes the constant
 * matrices: A, AKHA, G are S x S column-major (G = NULL: filter only, the KF = 1 option); HA, K have S entries.
 * MS (S x T, column-major) receives the filtered / smoothed means, *sum_v2 the sum of squared innovations of the observed
 * steps (lik = -( T/2 log(2 pi S_inn) + sum_v2 / (2 S_inn) ), :80,:101,:158).  S <= 256 (a thread per state; up to S = 96 both matrices of a pass live in the LDS and long series run parallel in time,
 * beyond that they are read from global memory and the passes run sequentially: 32 Matern-3/2 channels = S 128). */
int nagp_fastfb_run(int32_t S, const double* A, const double* AKHA, const double* HA, const double* K, const double* G,
                    const double* y, int64_t T, double* MS, double* sum_v2, int32_t device);

/* What the drivers do next with Eft / Varft (demo_toy_modulators_nmf.m:119-158, the same block in the other
 * demos): the reconstructed signal sig = sum_d (W link(g))_d z_d and the modulator amplitudes link(g_n) under the independent
 * posterior marginals z_d ~ N(Eft_d, Varft_d), g_n ~ N(Eft_{D+n}, Varft_{D+n}) of every time step:
 *   Eft_mod (N x T) = mean link(g_n), Varft_mod = var link(g_n), Esig (T) = mean sig, Vsig = var sig.
 * n_samples = 0: the population values (Gauss-Hermite rule gh_x, gh_w of n_gh points for the standard normal weight per
 *   modulator -- exp link: closed form -- then closed-form combination);
 * n_samples >= 2: the reference's estimator (s = 250 there): sample mean and variance (s-1) over draws of a counter-based
 *   generator (Philox4x32-10 keyed by `seed`, Box-Muller), reproducible on the host.
 * Eft, Varft: M x T column-major as returned by the *_run calls; Wnmf D x N column-major. */
int nagp_reconstruct(int32_t D, int32_t N, int64_t T, const double* Eft, const double* Varft, const double* Wnmf,
                     int32_t link_kind, double link_shift, int32_t n_gh, const double* gh_x, const double* gh_w,
                     int32_t n_samples, uint64_t seed, double* Esig, double* Vsig, double* Eft_mod, double* Varft_mod, int32_t device);

/* Batched / device-resident form: n_problems independent problems of identical shape
 * (S, M, block structure, T) -- audio segments or hyper-parameter replicas -- run concurrently. */
int nagp_plan_create(nagp_plan** plan, int32_t n_problems, const nagp_model* models,
                     const nagp_ihgp_tables* tables /* n_problems or NULL */, int64_t T, const nagp_opts* opts);
int nagp_plan_upload_y(nagp_plan* plan, const double* const* ys); /* n_problems pointers to T doubles (NaN = missing) */
int nagp_plan_execute(nagp_plan* plan);                          /* enqueue all sweeps; returns after stream sync */
int nagp_plan_timings(const nagp_plan* plan, nagp_timings* t);
int nagp_plan_download(nagp_plan* plan, nagp_out* outs);          /* n_problems outs */
int nagp_plan_upload_sites(nagp_plan* plan, const double* const* ttau0, const double* const* tnu0); /* warm start: n_problems pointers to
                                                                     M x T doubles each, or NULL/NULL to return to cold starts */
int64_t nagp_plan_device_bytes(const nagp_plan* plan);
void nagp_plan_destroy(nagp_plan* plan);

/* Multi-GPU form of the batched call (one process, the GPUs of one node) -- what a MEX caller uses to spread audio segments
 * or the numel(w)+1 objective evaluations of a fminunc iteration (train_GTFNMF.m:186-201) over the node:
 * problem i runs on device i mod n_gpus (nagp_batch_partition), one host thread, plan and stream per device, host buffers in
 * and out as in nagp_plan_*; the only exchange is the sum over ALL problems of the per-sweep negative log marginal likelihood
 * nlZ[itt] = -sum_k lZ_k (gf_ep_modulator_nmf.m:187, 277, 525), all-reduced over the devices with RCCL
 * (ncclAllReduce, ncclDouble, ncclSum, count = ep_itts) and returned in nlZ_total (ep_itts doubles, may be NULL).
 * opts->device is ignored; opts->ttau0 / tnu0 must be NULL (they describe ONE problem: NAGP_EINVAL otherwise -- warm-started
 * batches go through nagp_plan_create + nagp_plan_upload_sites).  n_gpus = 1 involves no collective unless the environment sets NAGP_FORCE_RCCL.  The RCCL
 * communicators are created on first use and kept until nagp_shutdown(). */
int nagp_batch_partition(int32_t n_problems, int32_t n_gpus, int32_t* device_of_problem /* n_problems */);
int nagp_batch_run(int32_t n_problems, const nagp_model* models, const nagp_ihgp_tables* tables /* n_problems or NULL */,
                   const double* const* ys, int64_t T, const nagp_opts* opts, nagp_out* outs, int32_t n_gpus,
                   double* nlZ_total);
void nagp_shutdown(void); /* releases cached RCCL communicators; never resets a device */

#ifdef __cplusplus
}
#endif
#endif /* NAGP_H */
