/* mex_driver.c -- drives matlab/nagp_mex.c (compiled against the mock mex.h of this directory) with a golden fixture:
 * builds the model / opts / tables structs the .m wrappers build, calls mexFunction, compares the outputs.
 *   mex_driver <dump dir>        exit 0 = within tolerance */
#include "dump.h"
#include "mex.h"

static mxArray* dbl(const char* d, const char* name, size_t rows) {
  size_t n; double* p = (double*)dump_load(d, name, 8, &n);
  mxArray* a = mock_numeric(mxDOUBLE_CLASS, rows ? rows : n, rows ? n / rows : 1, p);
  free(p); return a;
}

int main(int argc, char** argv) {
  const char* d = argc > 1 ? argv[1] : ".";
  size_t n, T; double worst = 0.0, r;
  const int S = (int)dump_scalar(d, "S"), kind = (int)dump_scalar(d, "kind");
  mxArray *model = mock_struct(), *opts = mock_struct(), *tables = NULL, *y;
  mxArray **plhs, **plhs2; const mxArray* prhs[4];
  int32_t* bo = (int32_t*)dump_load(d, "block_offsets", 4, &n);
  mock_set(model, "A", dbl(d, "A", S)); mock_set(model, "Q", dbl(d, "Q", S)); mock_set(model, "Pinf", dbl(d, "Pinf", S));
  mock_set(model, "block_offsets", mock_numeric(mxINT32_CLASS, 1, n, bo));
  mock_set(model, "h_val", dbl(d, "h_val", 0));
  mock_set(model, "Wnmf", dbl(d, "Wnmf", (size_t)dump_scalar(d, "D")));
  mock_set(model, "D", mock_scalar(dump_scalar(d, "D"))); mock_set(model, "N", mock_scalar(dump_scalar(d, "N")));
  mock_set(model, "lik_param", mock_scalar(dump_scalar(d, "lik_param")));
  y = dbl(d, "y", 0); T = mxGetNumberOfElements(y);
  mock_set(opts, "kind", mock_scalar(kind)); mock_set(opts, "mode", mock_scalar(0));
  mock_set(opts, "lik_kind", mock_scalar(dump_scalar(d, "lik_kind"))); mock_set(opts, "link_kind", mock_scalar(0));
  mock_set(opts, "link_shift", mock_scalar(0));
  mock_set(opts, "ep_fraction", mock_scalar(dump_scalar(d, "ep_fraction")));
  mock_set(opts, "ep_itts", mock_scalar(dump_scalar(d, "ep_itts")));
  mock_set(opts, "l_iter", mock_scalar(dump_scalar(d, "l_iter")));
  mock_set(opts, "flags", mock_scalar(dump_scalar(d, "flags")));
  if (kind != 2) {
    size_t npts = dump_count(d, "wn");
    mock_set(opts, "wn", dbl(d, "wn", 1));
    mock_set(opts, "xn_unscaled", dbl(d, "xn_unscaled", dump_count(d, "xn_unscaled") / npts));
    mock_set(opts, "ep_damping", dbl(d, "ep_damping", 1));
  }
  prhs[0] = model; prhs[1] = y; prhs[2] = opts;
  if (kind == 1) {
    int64_t* po = (int64_t*)dump_load(d, "pp_off", 8, &n); int64_t* pg = (int64_t*)dump_load(d, "pg_off", 8, &n);
    tables = mock_struct();
    mock_set(tables, "r", dbl(d, "r", 0)); mock_set(tables, "PP", dbl(d, "PP", 0)); mock_set(tables, "PG", dbl(d, "PG", 0));
    mock_set(tables, "pp_off", mock_numeric(mxINT64_CLASS, n, 1, po)); mock_set(tables, "pg_off", mock_numeric(mxINT64_CLASS, n, 1, pg));
    prhs[3] = tables;
  }
  /* plhs has EXACTLY nlhs slots (heap, so that a sanitizer build sees a gateway that writes past them): MATLAB promises no more */
  plhs = (mxArray**)malloc(11 * sizeof *plhs);
  mexFunction(11, plhs, kind == 1 ? 4 : 3, prhs);       /* ... counters, MS */
  plhs2 = (mxArray**)malloc(2 * sizeof *plhs2);           /* the wrappers' predict call: [Eft, Varft] = nagp_mex(...) */
  mexFunction(2, plhs2, kind == 1 ? 4 : 3, prhs);
  {
    size_t q; const size_t ne = mxGetNumberOfElements(plhs[0]);
    if (mxGetNumberOfElements(plhs2[0]) != ne || mxGetNumberOfElements(plhs2[1]) != ne) { printf("nlhs=2 call: wrong output sizes\n"); return 1; }
    q = ne * sizeof(double);
    if (memcmp(mxGetPr(plhs2[0]), mxGetPr(plhs[0]), q) || memcmp(mxGetPr(plhs2[1]), mxGetPr(plhs[1]), q)) { printf("nlhs=2 call differs from the nlhs=11 call\n"); return 1; }
  }
  {
    double* e = (double*)dump_load(d, "exp_Eft", 8, &n);
    r = rel_diff(mxGetPr(plhs[0]), e, n, "Eft"); if (r > worst) worst = r;
    e = (double*)dump_load(d, "exp_Varft", 8, &n);
    r = rel_diff(mxGetPr(plhs[1]), e, n, "Varft"); if (r > worst) worst = r;
    if (kind != 2) {
      e = (double*)dump_load(d, "exp_nlZ", 8, &n);
      r = rel_diff(mxGetPr(plhs[6]), e, n, "nlZ"); if (r > worst) worst = r;
      e = (double*)dump_load(d, "exp_tnu", 8, &n);
      r = 0.1 * rel_diff(mxGetPr(plhs[3]), e, n, "tnu"); if (r > worst) worst = r;
    }
  }
  if (kind == 0) {
    /* the command forms of the gateway on the same fixture:
       'batch' = nagp_batch_run over two copies of the problem (one GPU): every copy equals the single call, nlZ_total = 2 x nlZ;
       'reconstruct' on the marginals just computed (moments form): finite outputs of the right sizes;
       'iekf_update1' with iters = 1 on a small state: K = P J'/S, M moves along K */
    mxArray *models = mxCreateCellMatrix(1, 2), *ysc = mxCreateCellMatrix(1, 2), *o4[4]; const mxArray* a5[5]; size_t q; const size_t ne = mxGetNumberOfElements(plhs[0]);
    mxSetCell(models, 0, model); mxSetCell(models, 1, model); mxSetCell(ysc, 0, y); mxSetCell(ysc, 1, y);
    a5[0] = mock_string("batch"); a5[1] = models; a5[2] = ysc; a5[3] = opts; a5[4] = mock_scalar(1);
    mexFunction(4, o4, 5, a5);
    for (q = 0; q < 2; ++q)
      if (memcmp(mxGetPr(mxGetCell(o4[1], q)), mxGetPr(plhs[0]), ne * sizeof(double)) || memcmp(mxGetPr(mxGetCell(o4[2], q)), mxGetPr(plhs[1]), ne * sizeof(double))) {
        printf("batch call: problem %zu differs from the single call\n", q); return 1; }
    for (q = 0; q < mxGetNumberOfElements(o4[0]); ++q)
      if (fabs(mxGetPr(o4[0])[q] - 2.0 * mxGetPr(plhs[6])[q]) > 1e-12 * fabs(mxGetPr(plhs[6])[q])) { printf("batch call: nlZ_total is not the sum over the problems\n"); return 1; }
    {
      const mxArray* r[10]; mxArray* ro[4]; const double gx[3] = {-1.7320508075688772, 0.0, 1.7320508075688772}, gw[3] = {1.0 / 6, 2.0 / 3, 1.0 / 6};
      const size_t Dn = (size_t)dump_scalar(d, "D"), Nn = (size_t)dump_scalar(d, "N");
      r[0] = mock_string("reconstruct"); r[1] = plhs[0]; r[2] = plhs[1]; r[3] = mxGetField(model, 0, "Wnmf"); r[4] = mock_scalar(0); r[5] = mock_scalar(0);
      r[6] = mock_numeric(mxDOUBLE_CLASS, 1, 3, gx); r[7] = mock_numeric(mxDOUBLE_CLASS, 1, 3, gw); r[8] = mock_scalar(0); r[9] = mock_scalar(1);
      mexFunction(4, ro, 10, r);
      if (mxGetNumberOfElements(ro[0]) != T || mxGetNumberOfElements(ro[2]) != Nn * T || mxGetM(ro[2]) != Nn || Dn + Nn != mxGetM(plhs[0])) { printf("reconstruct: wrong output sizes\n"); return 1; }
      for (q = 0; q < T; ++q) if (!isfinite(mxGetPr(ro[0])[q]) || !(mxGetPr(ro[1])[q] >= 0)) { printf("reconstruct: bad value at %zu\n", q); return 1; }
    }
    {
      const double m0[3] = {0.3, -0.2, 0.1}, P0[9] = {1, 0, 0, 0, 2, 0, 0, 0, 0.5}, hv[2] = {1.0, 1.0}, W1[1] = {0.7}; const int32_t hc[2] = {0, 2};
      const mxArray* r[9]; mxArray* ro[5]; double e0, J0, J2, Sx, K0;
      r[0] = mock_string("iekf_update1"); r[1] = mock_numeric(mxDOUBLE_CLASS, 3, 1, m0); r[2] = mock_numeric(mxDOUBLE_CLASS, 3, 3, P0); r[3] = mock_scalar(0.4);
      r[4] = mock_numeric(mxINT32_CLASS, 1, 2, hc); r[5] = mock_numeric(mxDOUBLE_CLASS, 2, 1, hv); r[6] = mock_numeric(mxDOUBLE_CLASS, 1, 1, W1); r[7] = mock_scalar(0.05); r[8] = mock_scalar(1);
      mexFunction(5, ro, 9, r);
      /* h = z W softplus(g), z = m(1), g = m(3):  J = [W softplus(g), 0, z W sigmoid(g)] */
      e0 = exp(m0[2]); J0 = 0.7 * log(1.0 + e0); J2 = m0[0] * 0.7 * e0 / (1.0 + e0);
      Sx = 0.05 + J0 * J0 * 1.0 + J2 * J2 * 0.5; K0 = 1.0 * J0 / Sx;
      if (fabs(mxGetPr(ro[4])[0] - Sx) > 1e-13 || fabs(mxGetPr(ro[2])[0] - K0) > 1e-13 || fabs(mxGetPr(ro[3])[0] - m0[0] * J0) > 1e-13) {
        printf("iekf_update1: S %.15g (%.15g) K1 %.15g (%.15g) MU %.15g\n", mxGetPr(ro[4])[0], Sx, mxGetPr(ro[2])[0], K0, mxGetPr(ro[3])[0]); return 1; }
    }
    printf("command forms ok: batch, reconstruct, iekf_update1\n");
  }
  printf("counters: %lld %lld %lld %lld; MS is %zu x %zu; T = %zu\n", (long long)((int64_t*)mxGetData(plhs[9]))[0], (long long)((int64_t*)mxGetData(plhs[9]))[1],
         (long long)((int64_t*)mxGetData(plhs[9]))[2], (long long)((int64_t*)mxGetData(plhs[9]))[3], mxGetM(plhs[10]), mxGetNumberOfElements(plhs[10]) / mxGetM(plhs[10]), T);
  printf("worst %.3e\n", worst);
  return worst < 1e-7 ? 0 : 1;
}
