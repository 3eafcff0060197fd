/* abi_errors.c -- the argument-validation paths of the C ABI under AddressSanitizer (libnagp_asan.so: the host code of
 * nagp_api.hip and nagp_grad.hip instrumented), on a machine without a GPU: every bad call returns a status, nothing is read out of bounds.
 * Built and run by tests/test_host.py::test_abi_error_paths_under_asan. */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "nagp.h"

#define EXPECT(call, code)                                                                         \
  do { int _s = (call); if (_s != (code)) { fprintf(stderr, "%s -> %d (%s), expected %d\n", #call, _s, nagp_last_error(), (code)); ++bad; } } while (0)

int main(void) {
  int bad = 0, i;
  enum { S = 7, M = 3, D = 2, N = 1, T = 5 };
  double A[S * S], Q[S * S], P[S * S], h[M] = {1, 1, 1}, W[D * N] = {0.3, 0.2}, y[T] = {0.1, 0.2, 0.3, 0.4, 0.5};
  int32_t off[M + 1] = {0, 2, 4, 7}, dev[8];
  double wn[3] = {0.5, 0.25, 0.25}, xn[3] = {0, 1, -1}, damp[2] = {0.5, 0.5};
  nagp_model m; nagp_opts o; nagp_out out; nagp_plan* p = NULL;
  memset(A, 0, sizeof A); memset(Q, 0, sizeof Q); memset(P, 0, sizeof P);
  for (i = 0; i < S; ++i) { A[i * S + i] = 0.9; Q[i * S + i] = 0.1; P[i * S + i] = 1.0; }
  memset(&m, 0, sizeof m); memset(&o, 0, sizeof o); memset(&out, 0, sizeof out);
  m.S = S; m.M = M; m.D = D; m.N = N; m.block_offsets = off; m.A = A; m.Q = Q; m.Pinf = P; m.h_val = h; m.Wnmf = W;
  o.kind = NAGP_KIND_GF_EP; o.lik_kind = NAGP_LIK_POWER_NMF; o.n_pts = 3; o.cub_dim = N; o.wn = wn; o.xn_unscaled = xn;
  o.ep_fraction = 0.5; o.ep_itts = 2; o.ep_damping = damp;
  printf("version %d, devices %d, strerror(-7) = %s\n", nagp_version(), nagp_device_count(), nagp_strerror(-7));

  EXPECT(nagp_plan_create(NULL, 1, &m, NULL, T, &o), NAGP_EINVAL);
  EXPECT(nagp_plan_create(&p, 0, &m, NULL, T, &o), NAGP_EINVAL);
  EXPECT(nagp_plan_create(&p, 1, &m, NULL, 0, &o), NAGP_EINVAL);
  { nagp_opts q = o; q.ep_itts = 0; EXPECT(nagp_plan_create(&p, 1, &m, NULL, T, &q), NAGP_EINVAL); }
  { nagp_opts q = o; q.wn = NULL; EXPECT(nagp_plan_create(&p, 1, &m, NULL, T, &q), NAGP_EINVAL); }
  { nagp_opts q = o; q.kind = 7; EXPECT(nagp_plan_create(&p, 1, &m, NULL, T, &q), NAGP_EINVAL); }
  { nagp_opts q = o; q.kind = NAGP_KIND_IHGP; EXPECT(nagp_plan_create(&p, 1, &m, NULL, T, &q), NAGP_EINVAL); }        /* tables missing */
  { nagp_model q = m; q.A = NULL; EXPECT(nagp_plan_create(&p, 1, &q, NULL, T, &o), NAGP_EINVAL); }
  { nagp_model q = m; q.block_offsets = NULL; EXPECT(nagp_plan_create(&p, 1, &q, NULL, T, &o), NAGP_EINVAL); }
  { nagp_model q = m; q.Wnmf = NULL; EXPECT(nagp_plan_create(&p, 1, &q, NULL, T, &o), NAGP_EINVAL); }
  { nagp_model q = m; q.M = 65; EXPECT(nagp_plan_create(&p, 1, &q, NULL, T, &o), NAGP_EUNSUPPORTED); }
  { int32_t big[M + 1] = {0, 9, 10, 11}; nagp_model q = m; q.block_offsets = big; EXPECT(nagp_plan_create(&p, 1, &q, NULL, T, &o), NAGP_EUNSUPPORTED); }  /* block of 9 (blocks of 5 .. 8 states are split over two tile rows) */
  { int32_t gap[M + 1] = {1, 2, 4, 7}; nagp_model q = m; q.block_offsets = gap; EXPECT(nagp_plan_create(&p, 1, &q, NULL, T, &o), NAGP_EINVAL); }
  { nagp_model two[2]; int32_t other[M + 1] = {0, 3, 4, 7}; two[0] = m; two[1] = m; two[1].block_offsets = other;
    EXPECT(nagp_plan_create(&p, 2, two, NULL, T, &o), NAGP_EINVAL); }                                                    /* shapes differ */
  { nagp_model two[2]; two[0] = m; two[1] = m; two[1].Q = NULL; EXPECT(nagp_plan_create(&p, 2, two, NULL, T, &o), NAGP_EINVAL); }
  /* a well-formed request passes every host check and stops at the missing device */
  if (nagp_device_count() == 0) EXPECT(nagp_plan_create(&p, 1, &m, NULL, T, &o), NAGP_ENODEVICE);
  /* ... and so does one with a block of five states: the host builds the device view of the model (a tail row behind the sites, permuted
   * A / Q / Pinf, cross tiles) before it looks at the device -- under the sanitizer */
  if (nagp_device_count() == 0) {
    int32_t five[M + 1] = {0, 5, 6, 7}; nagp_model two[2]; two[0] = m; two[0].block_offsets = five; two[1] = two[0];
    EXPECT(nagp_plan_create(&p, 2, two, NULL, T, &o), NAGP_ENODEVICE);
    { nagp_opts q = o; q.kind = NAGP_KIND_GIEKF; q.l_iter = 2; EXPECT(nagp_plan_create(&p, 2, two, NULL, T, &q), NAGP_ENODEVICE); }
  }
  EXPECT(nagp_ep_run(NULL, y, T, &o, &out), NAGP_EINVAL);
  EXPECT(nagp_ep_run(&m, NULL, T, &o, &out), NAGP_EINVAL);
  { nagp_opts q = o; q.kind = NAGP_KIND_IHGP; EXPECT(nagp_ep_run(&m, y, T, &q, &out), NAGP_EINVAL); }
  EXPECT(nagp_plan_upload_y(NULL, NULL), NAGP_EINVAL);
  EXPECT(nagp_plan_upload_sites(NULL, NULL, NULL), NAGP_EINVAL);
  EXPECT(nagp_plan_execute(NULL), NAGP_EINVAL);
  EXPECT(nagp_plan_download(NULL, &out), NAGP_EINVAL);
  EXPECT(nagp_batch_partition(8, 3, dev), NAGP_OK);
  for (i = 0; i < 8; ++i) if (dev[i] != i % 3) { fprintf(stderr, "partition[%d] = %d\n", i, dev[i]); ++bad; }
  EXPECT(nagp_batch_partition(4, 0, dev), NAGP_EINVAL);
  { const double* ys[1] = {y}; EXPECT(nagp_batch_run(0, &m, NULL, ys, T, &o, &out, 1, NULL), NAGP_EINVAL);
    EXPECT(nagp_batch_run(1, &m, NULL, ys, T, &o, &out, 0, NULL), NAGP_EINVAL);
    { nagp_opts q = o; q.ttau0 = y; q.tnu0 = y; EXPECT(nagp_batch_run(1, &m, NULL, ys, T, &q, &out, 1, NULL), NAGP_EINVAL); } }   /* warm start: plans only */
  EXPECT(nagp_mom_eval(NULL, D, N, W, 0.0, 1, y, y, y, A, A, A), NAGP_EINVAL);
  EXPECT(nagp_iekf_update1(S, D, N, NULL, h, W, 0.1, 0.2, 1, A, P, NULL, NULL, NULL, 0), NAGP_EINVAL);
  EXPECT(nagp_fastfb_run(S, NULL, A, A, A, NULL, y, T, A, NULL, 0), NAGP_EINVAL);
  /* nagp_giekf_nlml_grad (nagp_grad.hip, instrumented as well): NULL, shape and w_index errors; a well-formed call runs the whole
     host packing (every block of dA, dQ, dPinf of every slice is read) and stops at the missing device */
  { enum { NP = 3 };
    static double dA[NP * S * S], dQ[NP * S * S], dP[NP * S * S];
    double dR[NP] = {1.0, 0.0, 0.0}, ed[1], gd[NP];
    int32_t hess[NP] = {1, 1, 0}, widx[NP] = {-1, -1, 1}, wdir[NP] = {0, 0, 0};
    const double* ys[1] = {y}; const double* pa[1] = {dA}; const double* pq[1] = {dQ}; const double* pp[1] = {dP};
    EXPECT(nagp_giekf_nlml_grad(0, &m, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL);
    EXPECT(nagp_giekf_nlml_grad(1, NULL, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL);
    EXPECT(nagp_giekf_nlml_grad(1, &m, NULL, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL);
    EXPECT(nagp_giekf_nlml_grad(1, &m, ys, 0, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL);
    EXPECT(nagp_giekf_nlml_grad(1, &m, ys, T, 0, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL);
    EXPECT(nagp_giekf_nlml_grad(1, &m, ys, T, NP, NULL, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL);
    EXPECT(nagp_giekf_nlml_grad(1, &m, ys, T, NP, pa, pq, pp, dR, hess, NULL, wdir, ed, gd, 0), NAGP_EINVAL);
    EXPECT(nagp_giekf_nlml_grad(1, &m, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, NULL, gd, 0), NAGP_EINVAL);
    { const double* none[1] = {NULL}; EXPECT(nagp_giekf_nlml_grad(1, &m, ys, T, NP, pa, none, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL);
      EXPECT(nagp_giekf_nlml_grad(1, &m, none, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL); }
    { int32_t wbad[NP] = {-1, -1, D * N}; EXPECT(nagp_giekf_nlml_grad(1, &m, ys, T, NP, pa, pq, pp, dR, hess, wbad, wdir, ed, gd, 0), NAGP_EINVAL); }   /* outside Wnmf */
    { nagp_model q = m; q.M = 4; EXPECT(nagp_giekf_nlml_grad(1, &q, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL); }          /* M != D + N */
    { nagp_model q = m; q.Wnmf = NULL; EXPECT(nagp_giekf_nlml_grad(1, &q, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL); }
    { int32_t big[M + 1] = {0, 5, 6, 7}; nagp_model q = m; q.block_offsets = big; EXPECT(nagp_giekf_nlml_grad(1, &q, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EUNSUPPORTED); }
    { int32_t gap[M + 1] = {1, 2, 4, 7}; nagp_model q = m; q.block_offsets = gap; EXPECT(nagp_giekf_nlml_grad(1, &q, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL); }
    { nagp_model two[2]; int32_t other[M + 1] = {0, 3, 4, 7}; const double* y2[2] = {y, y}; const double* a2[2] = {dA, dA};
      two[0] = m; two[1] = m; two[1].block_offsets = other;
      EXPECT(nagp_giekf_nlml_grad(2, two, y2, T, NP, a2, a2, a2, dR, hess, widx, wdir, ed, gd, 0), NAGP_EINVAL); }                                      /* shapes differ */
    if (nagp_device_count() == 0) EXPECT(nagp_giekf_nlml_grad(1, &m, ys, T, NP, pa, pq, pp, dR, hess, widx, wdir, ed, gd, 0), NAGP_ENODEVICE);
  }
  nagp_plan_destroy(NULL);
  nagp_shutdown();
  printf("%s\n", bad ? "FAILED" : "all error paths returned their status");
  return bad ? 1 : 0;
}
