/*
 * nagp_mex.c -- MEX gateway between the reference-named MATLAB wrappers in this directory and the C ABI of
 * include/nagp.h (libnagp.so: HIP kernels for gfx950).  Logic-free by design: it maps mxArray fields to the structs of
 * nagp.h, allocates the outputs and turns status codes into MATLAB errors.
 *
 *   mex -R2017b -I<repo>/include nagp_mex.c -L<repo>/nonstationary-audio-gp_amd -lnagp
 *
 *   [Eft,Varft,ttau,tnu,R,lZ,nlZ,maxDiffM,maxDiffP,counters,MS,PS] = nagp_mex(model, y, opts)
 *   [Eft,...] = nagp_mex(model, y, opts, tables)            % infinite-horizon kind
 *
 * model  struct: A, Q, Pinf (S x S double), block_offsets (int32, M+1, 0-based), h_val (M x 1), Wnmf (D x N or []),
 *                D, N, lik_param                                         -> nagp_model
 * y      double vector (NaN = missing)
 * opts   struct: kind, mode, lik_kind, link_kind, link_shift, wn, xn_unscaled, ep_fraction, ep_damping, l_iter,
 *                predict_at_k1, flags, device [, chunk, ttau0, tnu0, ep_itts]  -> nagp_opts
 *                (ep_itts defaults to numel(ep_damping); the EKF kind passes it explicitly as g_iter)
 * tables struct: r (n_grid x 1), PP, PG (double vectors), pp_off, pg_off (int64, M)          -> nagp_ihgp_tables
 *
 * The kernels replace the loops of matlab/gf_ep_modulator_nmf.m:113-283 / :384-522 (and the other functions listed in
 * include/nagp.h); everything before those loops stays in the .m wrappers.
 */
#include <string.h>

#include "mex.h"
#include "nagp.h"

static const mxArray* field(const mxArray* s, const char* name, int required) {
  const mxArray* f = mxIsStruct(s) ? mxGetField(s, 0, name) : NULL;
  if (!f && required) mexErrMsgIdAndTxt("nagp:arg", "missing field '%s'", name);
  return f;
}
static double scalar_or(const mxArray* s, const char* name, double dflt) {
  const mxArray* f = field(s, name, 0);
  return (f && !mxIsEmpty(f)) ? mxGetScalar(f) : dflt;
}
static const double* doubles(const mxArray* s, const char* name, int required, size_t* n) {
  const mxArray* f = field(s, name, required);
  if (n) *n = 0;
  if (!f || mxIsEmpty(f)) return NULL;
  if (!mxIsDouble(f) || mxIsComplex(f)) mexErrMsgIdAndTxt("nagp:arg", "field '%s' must be real double", name);
  if (n) *n = mxGetNumberOfElements(f);
  return mxGetPr(f);
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  nagp_model m;
  nagp_opts o;
  nagp_out out;
  nagp_ihgp_tables tb;
  const mxArray *sm, *so, *f;
  size_t n, nd, T;
  int st;
  mwSize dims3[3];

  if (nrhs < 3 || nrhs > 4) mexErrMsgIdAndTxt("nagp:arg", "usage: [...] = nagp_mex(model, y, opts [, tables])");
  if (nlhs > 12) mexErrMsgIdAndTxt("nagp:arg", "at most 12 outputs");
  sm = prhs[0]; so = prhs[2];
  memset(&m, 0, sizeof m); memset(&o, 0, sizeof o); memset(&out, 0, sizeof out); memset(&tb, 0, sizeof tb);

  /* ---- model */
  f = field(sm, "A", 1);
  m.S = (int32_t)mxGetM(f);
  m.A = doubles(sm, "A", 1, &n);
  if (n != (size_t)m.S * m.S) mexErrMsgIdAndTxt("nagp:arg", "A must be S x S");
  m.Q = doubles(sm, "Q", 1, &n);
  if (n != (size_t)m.S * m.S) mexErrMsgIdAndTxt("nagp:arg", "Q must be S x S");
  m.Pinf = doubles(sm, "Pinf", 1, &n);
  if (n != (size_t)m.S * m.S) mexErrMsgIdAndTxt("nagp:arg", "Pinf must be S x S");
  m.h_val = doubles(sm, "h_val", 1, &n);
  m.M = (int32_t)n;
  f = field(sm, "block_offsets", 1);
  if (!mxIsInt32(f) || mxGetNumberOfElements(f) != (size_t)m.M + 1) mexErrMsgIdAndTxt("nagp:arg", "block_offsets must be int32 with M+1 entries");
  m.block_offsets = (const int32_t*)mxGetData(f);
  m.Wnmf = doubles(sm, "Wnmf", 0, &n);
  m.D = (int32_t)scalar_or(sm, "D", 0);
  m.N = (int32_t)scalar_or(sm, "N", 0);
  if (m.Wnmf && n != (size_t)m.D * m.N) mexErrMsgIdAndTxt("nagp:arg", "Wnmf must be D x N");
  m.lik_param = scalar_or(sm, "lik_param", 0);

  /* ---- observations */
  if (!mxIsDouble(prhs[1]) || mxIsComplex(prhs[1])) mexErrMsgIdAndTxt("nagp:arg", "y must be real double");
  T = mxGetNumberOfElements(prhs[1]);

  /* ---- options */
  o.kind = (int32_t)scalar_or(so, "kind", NAGP_KIND_GF_EP);
  o.mode = (int32_t)scalar_or(so, "mode", NAGP_MODE_PREDICT);
  o.lik_kind = (int32_t)scalar_or(so, "lik_kind", NAGP_LIK_POWER_NMF);
  o.link_kind = (int32_t)scalar_or(so, "link_kind", NAGP_LINK_SOFTPLUS);
  o.link_shift = scalar_or(so, "link_shift", 0.0);
  o.wn = doubles(so, "wn", 0, &n);
  o.n_pts = (int32_t)n;
  o.xn_unscaled = doubles(so, "xn_unscaled", 0, &nd);
  if (o.wn) {
    if (!o.xn_unscaled || nd % n) mexErrMsgIdAndTxt("nagp:arg", "xn_unscaled must be cub_dim x n_pts");
    o.cub_dim = (int32_t)(nd / n);
  }
  o.ep_fraction = scalar_or(so, "ep_fraction", 0.5);
  o.ep_damping = doubles(so, "ep_damping", 0, &n);
  o.ep_itts = (int32_t)scalar_or(so, "ep_itts", (double)n);
  if (o.ep_damping && (size_t)o.ep_itts > n) mexErrMsgIdAndTxt("nagp:arg", "ep_damping has fewer than ep_itts entries");
  o.l_iter = (int32_t)scalar_or(so, "l_iter", 0);
  o.predict_at_k1 = (int32_t)scalar_or(so, "predict_at_k1", 0);
  o.flags = (uint32_t)scalar_or(so, "flags", 0);
  o.device = (int32_t)scalar_or(so, "device", 0);
  o.chunk = (int32_t)scalar_or(so, "chunk", 0);
  o.ttau0 = doubles(so, "ttau0", 0, &n);
  if (o.ttau0 && n != (size_t)m.M * T) mexErrMsgIdAndTxt("nagp:arg", "ttau0 must be M x T");
  o.tnu0 = doubles(so, "tnu0", 0, &n);
  if (o.tnu0 && n != (size_t)m.M * T) mexErrMsgIdAndTxt("nagp:arg", "tnu0 must be M x T");
  if (o.ep_itts < 1) mexErrMsgIdAndTxt("nagp:arg", "ep_itts < 1");
  if (nlhs > 11) o.flags |= NAGP_FLAG_WANT_PS;

  /* ---- outputs (column-major M x T etc.: exactly the library's layout) */
  /* MATLAB guarantees max(nlhs, 1) slots in plhs and no more: only those are touched */
  plhs[0] = mxCreateDoubleMatrix(m.M, T, mxREAL); out.Eft = mxGetPr(plhs[0]);
  if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(m.M, T, mxREAL); out.Varft = mxGetPr(plhs[1]); }
  if (nlhs > 2) { plhs[2] = mxCreateDoubleMatrix(m.M, T, mxREAL); out.ttau = mxGetPr(plhs[2]); }
  if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(m.M, T, mxREAL); out.tnu = mxGetPr(plhs[3]); }
  if (nlhs > 4) { plhs[4] = mxCreateDoubleMatrix(m.M, T, mxREAL); out.R = mxGetPr(plhs[4]); }
  if (nlhs > 5) { plhs[5] = mxCreateDoubleMatrix(1, T, mxREAL); out.lZ = mxGetPr(plhs[5]); }
  if (nlhs > 6) { plhs[6] = mxCreateDoubleMatrix(1, o.ep_itts, mxREAL); out.nlZ = mxGetPr(plhs[6]); }
  if (nlhs > 7) { plhs[7] = mxCreateDoubleMatrix(1, o.ep_itts, mxREAL); out.maxDiffM = mxGetPr(plhs[7]); }
  if (nlhs > 8) { plhs[8] = mxCreateDoubleMatrix(1, o.ep_itts, mxREAL); out.maxDiffP = mxGetPr(plhs[8]); }
  if (nlhs > 9) { plhs[9] = mxCreateNumericMatrix(1, NAGP_N_COUNTERS, mxINT64_CLASS, mxREAL); out.counters = (int64_t*)mxGetData(plhs[9]); }
  if (nlhs > 10) { plhs[10] = mxCreateDoubleMatrix(m.S, T, mxREAL); out.MS = mxGetPr(plhs[10]); }
  if (nlhs > 11) {
    dims3[0] = (mwSize)m.S; dims3[1] = (mwSize)m.S; dims3[2] = (mwSize)T;
    plhs[11] = mxCreateNumericArray(3, dims3, mxDOUBLE_CLASS, mxREAL); out.PS = mxGetPr(plhs[11]);
  }

  /* ---- the call */
  if (o.kind == NAGP_KIND_IHGP) {
    const mxArray* stb;
    if (nrhs < 4) mexErrMsgIdAndTxt("nagp:arg", "the infinite-horizon kind needs the look-up tables");
    stb = prhs[3];
    tb.r_grid = doubles(stb, "r", 1, &n);
    tb.n_grid = (int32_t)n;
    tb.PPlist = doubles(stb, "PP", 1, NULL);
    tb.PGlist = doubles(stb, "PG", 1, NULL);
    f = field(stb, "pp_off", 1);
    if (!mxIsInt64(f) || mxGetNumberOfElements(f) != (size_t)m.M) mexErrMsgIdAndTxt("nagp:arg", "pp_off must be int64 with M entries");
    tb.pp_offsets = (const int64_t*)mxGetData(f);
    f = field(stb, "pg_off", 1);
    if (!mxIsInt64(f) || mxGetNumberOfElements(f) != (size_t)m.M) mexErrMsgIdAndTxt("nagp:arg", "pg_off must be int64 with M entries");
    tb.pg_offsets = (const int64_t*)mxGetData(f);
    st = nagp_ihgp_run(&m, &tb, mxGetPr(prhs[1]), (int64_t)T, &o, &out);
  } else if (o.kind == NAGP_KIND_GIEKF) {
    st = nagp_giekf_run(&m, mxGetPr(prhs[1]), (int64_t)T, &o, &out);
  } else {
    st = nagp_ep_run(&m, mxGetPr(prhs[1]), (int64_t)T, &o, &out);
  }
  if (st != NAGP_OK) mexErrMsgIdAndTxt("nagp:fail", "%s (%d): %s", nagp_strerror(st), st, nagp_last_error());
}
