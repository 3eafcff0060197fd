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
  printf("counters: %lld %lld %lld %lld; MS is %zu x %zu; T = %zu\n", (long long)((int64_t*)mxGetData(plhs[9]))[0], (long long)((int64_t*)mxGetData(plhs[9]))[1],
         (long long)((int64_t*)mxGetData(plhs[9]))[2], (long long)((int64_t*)mxGetData(plhs[9]))[3], mxGetM(plhs[10]), mxGetNumberOfElements(plhs[10]) / mxGetM(plhs[10]), T);
  printf("worst %.3e\n", worst);
  return worst < 1e-7 ? 0 : 1;
}
