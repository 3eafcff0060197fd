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
 * Further entry points of include/nagp.h, selected by a command string in the first argument:
 *   [M,P,K,MU,S] = nagp_mex('iekf_update1', M, P, y, h_col(int32, 0-based), h_val, Wnmf, R, iters [,device])   nagp_iekf_update1
 *   [MS, sum_v2] = nagp_mex('fastfb', A, AKHA, HA, K, G ([] = filter only), y [,device])                       nagp_fastfb_run
 *   [Esig,Vsig,Eft_mod,Varft_mod] = nagp_mex('reconstruct', Eft, Varft, Wnmf, link_kind, link_shift, gh_x, gh_w, n_samples, seed [,device])
 *                                                                                                              nagp_reconstruct
 *   [nlZ_total, Eft, Varft, nlZ] = nagp_mex('batch', models {cell of model structs}, ys {cell}, opts, n_gpus [, tables {cell}])
 *                                    Eft, Varft, nlZ: cells, one entry per problem                             nagp_batch_run
 *   [e, g] = nagp_mex('giekf_grad', model, y, dA, dQ, dPinf, dR, hess, w_index, w_direct [,device])            nagp_giekf_nlml_grad
 *
 * The kernels replace the loops of matlab/gf_ep_modulator_nmf.m:113-283 / :384-522 (and the other functions listed in
 * include/nagp.h); everything before those loops stays in the .m wrappers.
 */
#include <stdlib.h>
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

static const double* dvec(const mxArray* a, const char* what, size_t* n) {
  if (n) *n = 0;
  if (!a || mxIsEmpty(a)) return NULL;
  if (!mxIsDouble(a) || mxIsComplex(a)) mexErrMsgIdAndTxt("nagp:arg", "%s must be real double", what);
  if (n) *n = mxGetNumberOfElements(a);
  return mxGetPr(a);
}
static void fail_if(int st) {
  if (st != NAGP_OK) mexErrMsgIdAndTxt("nagp:fail", "%s (%d): %s", nagp_strerror(st), st, nagp_last_error());
}

static void read_model(const mxArray* sm, nagp_model* m) {
  const mxArray* f; size_t n;
  memset(m, 0, sizeof *m);
  f = field(sm, "A", 1);
  m->S = (int32_t)mxGetM(f);
  m->A = doubles(sm, "A", 1, &n);
  if (n != (size_t)m->S * m->S) mexErrMsgIdAndTxt("nagp:arg", "A must be S x S");
  m->Q = doubles(sm, "Q", 1, &n);
  if (n != (size_t)m->S * m->S) mexErrMsgIdAndTxt("nagp:arg", "Q must be S x S");
  m->Pinf = doubles(sm, "Pinf", 1, &n);
  if (n != (size_t)m->S * m->S) mexErrMsgIdAndTxt("nagp:arg", "Pinf must be S x S");
  m->h_val = doubles(sm, "h_val", 1, &n);
  m->M = (int32_t)n;
  f = field(sm, "block_offsets", 1);
  if (!mxIsInt32(f) || mxGetNumberOfElements(f) != (size_t)m->M + 1) mexErrMsgIdAndTxt("nagp:arg", "block_offsets must be int32 with M+1 entries");
  m->block_offsets = (const int32_t*)mxGetData(f);
  m->Wnmf = doubles(sm, "Wnmf", 0, &n);
  m->D = (int32_t)scalar_or(sm, "D", 0);
  m->N = (int32_t)scalar_or(sm, "N", 0);
  if (m->Wnmf && n != (size_t)m->D * m->N) mexErrMsgIdAndTxt("nagp:arg", "Wnmf must be D x N");
  m->lik_param = scalar_or(sm, "lik_param", 0);
}

static void read_opts(const mxArray* so, nagp_opts* o, int M, size_t T) {
  size_t n, nd;
  memset(o, 0, sizeof *o);
  o->kind = (int32_t)scalar_or(so, "kind", NAGP_KIND_GF_EP);
  o->mode = (int32_t)scalar_or(so, "mode", NAGP_MODE_PREDICT);
  o->lik_kind = (int32_t)scalar_or(so, "lik_kind", NAGP_LIK_POWER_NMF);
  o->link_kind = (int32_t)scalar_or(so, "link_kind", NAGP_LINK_SOFTPLUS);
  o->link_shift = scalar_or(so, "link_shift", 0.0);
  o->wn = doubles(so, "wn", 0, &n);
  o->n_pts = (int32_t)n;
  o->xn_unscaled = doubles(so, "xn_unscaled", 0, &nd);
  if (o->wn) {
    if (!o->xn_unscaled || nd % n) mexErrMsgIdAndTxt("nagp:arg", "xn_unscaled must be cub_dim x n_pts");
    o->cub_dim = (int32_t)(nd / n);
  }
  o->ep_fraction = scalar_or(so, "ep_fraction", 0.5);
  o->ep_damping = doubles(so, "ep_damping", 0, &n);
  o->ep_itts = (int32_t)scalar_or(so, "ep_itts", (double)n);
  if (o->ep_damping && (size_t)o->ep_itts > n) mexErrMsgIdAndTxt("nagp:arg", "ep_damping has fewer than ep_itts entries");
  o->l_iter = (int32_t)scalar_or(so, "l_iter", 0);
  o->predict_at_k1 = (int32_t)scalar_or(so, "predict_at_k1", 0);
  o->flags = (uint32_t)scalar_or(so, "flags", 0);
  o->device = (int32_t)scalar_or(so, "device", 0);
  o->chunk = (int32_t)scalar_or(so, "chunk", 0);
  o->ttau0 = doubles(so, "ttau0", 0, &n);
  if (o->ttau0 && n != (size_t)M * T) mexErrMsgIdAndTxt("nagp:arg", "ttau0 must be M x T");
  o->tnu0 = doubles(so, "tnu0", 0, &n);
  if (o->tnu0 && n != (size_t)M * T) mexErrMsgIdAndTxt("nagp:arg", "tnu0 must be M x T");
  if (o->ep_itts < 1) mexErrMsgIdAndTxt("nagp:arg", "ep_itts < 1");
}

static void read_tables(const mxArray* stb, nagp_ihgp_tables* tb, int M) {
  const mxArray* f; size_t n;
  memset(tb, 0, sizeof *tb);
  tb->r_grid = doubles(stb, "r", 1, &n);
  tb->n_grid = (int32_t)n;
  tb->PPlist = doubles(stb, "PP", 1, NULL);
  tb->PGlist = doubles(stb, "PG", 1, NULL);
  f = field(stb, "pp_off", 1);
  if (!mxIsInt64(f) || mxGetNumberOfElements(f) != (size_t)M) mexErrMsgIdAndTxt("nagp:arg", "pp_off must be int64 with M entries");
  tb->pp_offsets = (const int64_t*)mxGetData(f);
  f = field(stb, "pg_off", 1);
  if (!mxIsInt64(f) || mxGetNumberOfElements(f) != (size_t)M) mexErrMsgIdAndTxt("nagp:arg", "pg_off must be int64 with M entries");
  tb->pg_offsets = (const int64_t*)mxGetData(f);
}

/* ---- the command forms (first argument a string) */
static void cmd_giekf_grad(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  /* ('giekf_grad', model, y, dA, dQ, dPinf (S x S x n_param each), dR, hess, w_index, w_direct (int32, n_param each) [,device]) -> [e, g]
     the EKF energy with its gradient recursion, gf_giekf_modulator_nmf_constraints.m:332-480 with GradObj = 'on' */
  nagp_model m; size_t T, n, np_; const double *y, *dA, *dQ, *dP, *dR; const double* ys[1]; const double* a1[1]; const double* a2[1]; const double* a3[1];
  double e = 0; int32_t dev;
  if (nrhs < 10 || nrhs > 11) mexErrMsgIdAndTxt("nagp:arg", "usage: [e,g] = nagp_mex('giekf_grad',model,y,dA,dQ,dPinf,dR,hess,w_index,w_direct[,device])");
  read_model(prhs[1], &m);
  y = dvec(prhs[2], "y", &T);
  dR = dvec(prhs[6], "dR", &np_);
  dA = dvec(prhs[3], "dA", &n); if (n != (size_t)m.S * m.S * np_) mexErrMsgIdAndTxt("nagp:arg", "dA must be S x S x numel(dR)");
  dQ = dvec(prhs[4], "dQ", &n); if (n != (size_t)m.S * m.S * np_) mexErrMsgIdAndTxt("nagp:arg", "dQ must be S x S x numel(dR)");
  dP = dvec(prhs[5], "dPinf", &n); if (n != (size_t)m.S * m.S * np_) mexErrMsgIdAndTxt("nagp:arg", "dPinf must be S x S x numel(dR)");
  if (!mxIsInt32(prhs[7]) || !mxIsInt32(prhs[8]) || !mxIsInt32(prhs[9]) || mxGetNumberOfElements(prhs[7]) != np_ ||
      mxGetNumberOfElements(prhs[8]) != np_ || mxGetNumberOfElements(prhs[9]) != np_) mexErrMsgIdAndTxt("nagp:arg", "hess, w_index, w_direct must be int32 with numel(dR) entries");
  dev = nrhs > 10 ? (int32_t)mxGetScalar(prhs[10]) : 0;
  plhs[0] = mxCreateDoubleMatrix(1, 1, mxREAL);
  { mxArray* g = mxCreateDoubleMatrix(1, np_, mxREAL);
    ys[0] = y; a1[0] = dA; a2[0] = dQ; a3[0] = dP;
    fail_if(nagp_giekf_nlml_grad(1, &m, ys, (int64_t)T, (int32_t)np_, a1, a2, a3, dR, (const int32_t*)mxGetData(prhs[7]), (const int32_t*)mxGetData(prhs[8]),
                                 (const int32_t*)mxGetData(prhs[9]), &e, mxGetPr(g), dev));
    mxGetPr(plhs[0])[0] = e;
    if (nlhs > 1) plhs[1] = g; }
}

static void cmd_iekf_update1(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  /* ('iekf_update1', M, P, y, h_col, h_val, Wnmf, R, iters [,device]) -> [M,P,K,MU,S]  (iekf_update1.m:48, :110-117) */
  size_t S, n, nm; int32_t D, N, iters, dev; const double *m0, *P0, *hv, *W; double *m, *P, *K, MU = 0, Sx = 0;
  if (nrhs < 9 || nrhs > 10) mexErrMsgIdAndTxt("nagp:arg", "usage: [M,P,K,MU,S] = nagp_mex('iekf_update1',M,P,y,h_col,h_val,Wnmf,R,iters[,device])");
  m0 = dvec(prhs[1], "M", &S); P0 = dvec(prhs[2], "P", &n);
  if (!m0 || n != S * S) mexErrMsgIdAndTxt("nagp:arg", "P must be S x S for M of S entries");
  hv = dvec(prhs[5], "h_val", &nm);
  if (!mxIsInt32(prhs[4]) || mxGetNumberOfElements(prhs[4]) != nm) mexErrMsgIdAndTxt("nagp:arg", "h_col must be int32 with as many entries as h_val");
  W = dvec(prhs[6], "Wnmf", &n);
  D = (int32_t)mxGetM(prhs[6]); N = D ? (int32_t)(n / (size_t)D) : 0;
  if ((size_t)(D + N) != nm) mexErrMsgIdAndTxt("nagp:arg", "Wnmf must be D x N with D + N = numel(h_val)");
  iters = (int32_t)mxGetScalar(prhs[8]); dev = nrhs > 9 ? (int32_t)mxGetScalar(prhs[9]) : 0;
  plhs[0] = mxCreateDoubleMatrix(S, 1, mxREAL); m = mxGetPr(plhs[0]); memcpy(m, m0, S * sizeof(double));
  { mxArray* Pa = mxCreateDoubleMatrix(S, S, mxREAL); mxArray* Ka = mxCreateDoubleMatrix(S, 1, mxREAL);
    P = mxGetPr(Pa); memcpy(P, P0, S * S * sizeof(double)); K = mxGetPr(Ka);
    fail_if(nagp_iekf_update1((int32_t)S, D, N, (const int32_t*)mxGetData(prhs[4]), hv, W, mxGetScalar(prhs[7]), mxGetScalar(prhs[3]), iters, m, P, K, &MU, &Sx, dev));
    if (nlhs > 1) plhs[1] = Pa;
    if (nlhs > 2) plhs[2] = Ka; }
  if (nlhs > 3) { plhs[3] = mxCreateDoubleMatrix(1, 1, mxREAL); mxGetPr(plhs[3])[0] = MU; }
  if (nlhs > 4) { plhs[4] = mxCreateDoubleMatrix(1, 1, mxREAL); mxGetPr(plhs[4])[0] = Sx; }
}

static void cmd_fastfb(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  /* ('fastfb', A, AKHA, HA, K, G, y [,device]) -> [MS, sum_v2]  (kernel_ss_kalmanFastFB.m:83-110, :134-151) */
  size_t n, S, T; const double *A, *AK, *HA, *K, *G, *y; double sv2 = 0; int32_t dev;
  if (nrhs < 7 || nrhs > 8) mexErrMsgIdAndTxt("nagp:arg", "usage: [MS,sum_v2] = nagp_mex('fastfb',A,AKHA,HA,K,G,y[,device])");
  A = dvec(prhs[1], "A", &n); S = mxGetM(prhs[1]);
  if (!A || n != S * S) mexErrMsgIdAndTxt("nagp:arg", "A must be S x S");
  AK = dvec(prhs[2], "AKHA", &n); if (n != S * S) mexErrMsgIdAndTxt("nagp:arg", "AKHA must be S x S");
  HA = dvec(prhs[3], "HA", &n); if (n != S) mexErrMsgIdAndTxt("nagp:arg", "HA must have S entries");
  K = dvec(prhs[4], "K", &n); if (n != S) mexErrMsgIdAndTxt("nagp:arg", "K must have S entries");
  G = dvec(prhs[5], "G", &n); if (G && n != S * S) mexErrMsgIdAndTxt("nagp:arg", "G must be S x S or []");
  y = dvec(prhs[6], "y", &T); dev = nrhs > 7 ? (int32_t)mxGetScalar(prhs[7]) : 0;
  plhs[0] = mxCreateDoubleMatrix(S, T, mxREAL);
  fail_if(nagp_fastfb_run((int32_t)S, A, AK, HA, K, G, y, (int64_t)T, mxGetPr(plhs[0]), &sv2, dev));
  if (nlhs > 1) { plhs[1] = mxCreateDoubleMatrix(1, 1, mxREAL); mxGetPr(plhs[1])[0] = sv2; }
}

static void cmd_reconstruct(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  /* ('reconstruct', Eft, Varft, Wnmf, link_kind, link_shift, gh_x, gh_w, n_samples, seed [,device])  (demo_toy_modulators_nmf.m:119-158) */
  size_t n, nv, ngx, ngw, D, N, M, T; const double *E, *V, *W, *gx, *gw; int32_t dev; mxArray* o[4]; int i;
  if (nrhs < 10 || nrhs > 11) mexErrMsgIdAndTxt("nagp:arg", "usage: [Esig,Vsig,Eft_mod,Varft_mod] = nagp_mex('reconstruct',Eft,Varft,Wnmf,link_kind,link_shift,gh_x,gh_w,n_samples,seed[,device])");
  E = dvec(prhs[1], "Eft", &n); V = dvec(prhs[2], "Varft", &nv); W = dvec(prhs[3], "Wnmf", NULL);
  M = mxGetM(prhs[1]); D = mxGetM(prhs[3]); N = D ? mxGetNumberOfElements(prhs[3]) / D : 0;
  if (!E || !V || !W || n != nv || M != D + N) mexErrMsgIdAndTxt("nagp:arg", "Eft, Varft must be (D+N) x T for Wnmf of D x N");
  T = n / M;
  gx = dvec(prhs[6], "gh_x", &ngx); gw = dvec(prhs[7], "gh_w", &ngw);
  if (ngx != ngw) mexErrMsgIdAndTxt("nagp:arg", "gh_x and gh_w must have the same length");
  dev = nrhs > 10 ? (int32_t)mxGetScalar(prhs[10]) : 0;
  o[0] = mxCreateDoubleMatrix(1, T, mxREAL); o[1] = mxCreateDoubleMatrix(1, T, mxREAL);
  o[2] = mxCreateDoubleMatrix(N, T, mxREAL); o[3] = mxCreateDoubleMatrix(N, T, mxREAL);
  fail_if(nagp_reconstruct((int32_t)D, (int32_t)N, (int64_t)T, E, V, W, (int32_t)mxGetScalar(prhs[4]), mxGetScalar(prhs[5]), (int32_t)ngx, gx, gw,
                           (int32_t)mxGetScalar(prhs[8]), (uint64_t)mxGetScalar(prhs[9]), mxGetPr(o[0]), mxGetPr(o[1]), mxGetPr(o[2]), mxGetPr(o[3]), dev));
  plhs[0] = o[0];
  for (i = 1; i < 4; ++i) if (nlhs > i) plhs[i] = o[i];
}

static void cmd_batch(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  /* ('batch', models {cell}, ys {cell}, opts, n_gpus [, tables {cell}]) -> [nlZ_total, Eft {cell}, Varft {cell}, nlZ {cell}]
     segments / objective replicas spread over the GPUs of the node, nlZ all-reduced with RCCL (nagp_batch_run) */
  size_t B, T = 0, q; nagp_model* ms; nagp_ihgp_tables* ts = NULL; const double** ys; nagp_out* outs; nagp_opts o; int32_t ng;
  mxArray *cE, *cV, *cZ;
  if (nrhs < 5 || nrhs > 6 || !mxIsCell(prhs[1]) || !mxIsCell(prhs[2])) mexErrMsgIdAndTxt("nagp:arg", "usage: [nlZ_total,Eft,Varft,nlZ] = nagp_mex('batch',models,ys,opts,n_gpus[,tables])");
  B = mxGetNumberOfElements(prhs[1]);
  if (B < 1 || mxGetNumberOfElements(prhs[2]) != B || (nrhs > 5 && (!mxIsCell(prhs[5]) || mxGetNumberOfElements(prhs[5]) != B)))
    mexErrMsgIdAndTxt("nagp:arg", "models, ys (and tables) must be cell arrays of the same length");
  /* mxCalloc: MATLAB frees these when an argument error below leaves the MEX function through mexErrMsgIdAndTxt (a longjmp) */
  ms = (nagp_model*)mxCalloc(B, sizeof *ms); ys = (const double**)mxCalloc(B, sizeof *ys); outs = (nagp_out*)mxCalloc(B, sizeof *outs);
  if (nrhs > 5) ts = (nagp_ihgp_tables*)mxCalloc(B, sizeof *ts);
  for (q = 0; q < B; ++q) {
    size_t n;
    read_model(mxGetCell(prhs[1], q), &ms[q]);
    ys[q] = dvec(mxGetCell(prhs[2], q), "y", &n);
    if (q == 0) T = n;
    if (!ys[q] || n != T) mexErrMsgIdAndTxt("nagp:arg", "every y must be a real double vector of the same length");
    if (ts) read_tables(mxGetCell(prhs[5], q), &ts[q], ms[q].M);
  }
  read_opts(prhs[3], &o, ms[0].M, T);
  if (o.ttau0 || o.tnu0) mexErrMsgIdAndTxt("nagp:arg", "the batch call takes no warm-start sites");
  ng = (int32_t)mxGetScalar(prhs[4]);
  plhs[0] = mxCreateDoubleMatrix(1, o.ep_itts, mxREAL);
  cE = mxCreateCellMatrix(1, B); cV = mxCreateCellMatrix(1, B); cZ = mxCreateCellMatrix(1, B);
  for (q = 0; q < B; ++q) {
    mxArray* e = mxCreateDoubleMatrix(ms[q].M, T, mxREAL); mxArray* v = mxCreateDoubleMatrix(ms[q].M, T, mxREAL); mxArray* z = mxCreateDoubleMatrix(1, o.ep_itts, mxREAL);
    outs[q].Eft = mxGetPr(e); outs[q].Varft = mxGetPr(v); outs[q].nlZ = mxGetPr(z);
    mxSetCell(cE, q, e); mxSetCell(cV, q, v); mxSetCell(cZ, q, z);
  }
  { const int st = nagp_batch_run((int32_t)B, ms, ts, ys, (int64_t)T, &o, outs, ng, mxGetPr(plhs[0]));
    mxFree(ms); mxFree((void*)ys); mxFree(outs); if (ts) mxFree(ts);
    fail_if(st); }
  if (nlhs > 1) plhs[1] = cE;
  if (nlhs > 2) plhs[2] = cV;
  if (nlhs > 3) plhs[3] = cZ;
}

void mexFunction(int nlhs, mxArray* plhs[], int nrhs, const mxArray* prhs[]) {
  nagp_model m;
  nagp_opts o;
  nagp_out out;
  nagp_ihgp_tables tb;
  size_t T;
  int st;
  mwSize dims3[3];

  if (nrhs >= 1 && mxIsChar(prhs[0])) {
    char cmd[32];
    if (mxGetString(prhs[0], cmd, sizeof cmd)) mexErrMsgIdAndTxt("nagp:arg", "command string too long");
    if (!strcmp(cmd, "iekf_update1")) cmd_iekf_update1(nlhs, plhs, nrhs, prhs);
    else if (!strcmp(cmd, "fastfb")) cmd_fastfb(nlhs, plhs, nrhs, prhs);
    else if (!strcmp(cmd, "reconstruct")) cmd_reconstruct(nlhs, plhs, nrhs, prhs);
    else if (!strcmp(cmd, "batch")) cmd_batch(nlhs, plhs, nrhs, prhs);
    else if (!strcmp(cmd, "giekf_grad")) cmd_giekf_grad(nlhs, plhs, nrhs, prhs);
    else mexErrMsgIdAndTxt("nagp:arg", "unknown command '%s'", cmd);
    return;
  }
  if (nrhs < 3 || nrhs > 4) mexErrMsgIdAndTxt("nagp:arg", "usage: [...] = nagp_mex(model, y, opts [, tables])");
  if (nlhs > 12) mexErrMsgIdAndTxt("nagp:arg", "at most 12 outputs");
  memset(&out, 0, sizeof out); memset(&tb, 0, sizeof tb);
  read_model(prhs[0], &m);

  /* ---- observations */
  if (!mxIsDouble(prhs[1]) || mxIsComplex(prhs[1])) mexErrMsgIdAndTxt("nagp:arg", "y must be real double");
  T = mxGetNumberOfElements(prhs[1]);
  read_opts(prhs[2], &o, m.M, T);
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
    if (nrhs < 4) mexErrMsgIdAndTxt("nagp:arg", "the infinite-horizon kind needs the look-up tables");
    read_tables(prhs[3], &tb, m.M);
    st = nagp_ihgp_run(&m, &tb, mxGetPr(prhs[1]), (int64_t)T, &o, &out);
  } else if (o.kind == NAGP_KIND_GIEKF) {
    st = nagp_giekf_run(&m, mxGetPr(prhs[1]), (int64_t)T, &o, &out);
  } else {
    st = nagp_ep_run(&m, mxGetPr(prhs[1]), (int64_t)T, &o, &out);
  }
  fail_if(st);
}
