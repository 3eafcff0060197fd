/* abi_golden.c -- a plain C caller of the C ABI (include/nagp.h): loads a golden fixture dumped by the Python test,
 * calls nagp_ep_run / nagp_ihgp_run / nagp_giekf_run and compares with the fixture's expected outputs.
 *   abi_golden <dump dir>        exit 0 = within tolerance
 * Built with gcc against libnagp.so (tests/test_gpu_parity.py::test_c_abi_from_plain_c). */
#include "dump.h"
#include "nagp.h"

int main(int argc, char** argv) {
  const char* d = argc > 1 ? argv[1] : ".";
  nagp_model m; nagp_opts o; nagp_out out; nagp_ihgp_tables tb;
  size_t n, T, I; int st; double worst = 0.0, r;
  memset(&m, 0, sizeof m); memset(&o, 0, sizeof o); memset(&out, 0, sizeof out); memset(&tb, 0, sizeof tb);
  m.S = (int32_t)dump_scalar(d, "S"); m.M = (int32_t)dump_scalar(d, "M"); m.D = (int32_t)dump_scalar(d, "D"); m.N = (int32_t)dump_scalar(d, "N");
  m.A = (double*)dump_load(d, "A", 8, &n); m.Q = (double*)dump_load(d, "Q", 8, &n); m.Pinf = (double*)dump_load(d, "Pinf", 8, &n);
  m.block_offsets = (int32_t*)dump_load(d, "block_offsets", 4, &n);
  m.h_val = (double*)dump_load(d, "h_val", 8, &n);
  m.Wnmf = (double*)dump_load(d, "Wnmf", 8, &n);
  m.lik_param = dump_scalar(d, "lik_param");
  const double* y = (double*)dump_load(d, "y", 8, &T);
  o.kind = (int32_t)dump_scalar(d, "kind"); o.mode = NAGP_MODE_PREDICT;
  o.lik_kind = (int32_t)dump_scalar(d, "lik_kind"); o.link_kind = NAGP_LINK_SOFTPLUS; o.link_shift = 0.0;
  o.ep_fraction = dump_scalar(d, "ep_fraction");
  o.ep_itts = (int32_t)dump_scalar(d, "ep_itts"); o.l_iter = (int32_t)dump_scalar(d, "l_iter");
  o.flags = (uint32_t)dump_scalar(d, "flags");
  if (o.kind != NAGP_KIND_GIEKF) {
    o.wn = (double*)dump_load(d, "wn", 8, &n); o.n_pts = (int32_t)n;
    o.xn_unscaled = (double*)dump_load(d, "xn_unscaled", 8, &n); o.cub_dim = (int32_t)(n / o.n_pts);
    o.ep_damping = (double*)dump_load(d, "ep_damping", 8, &I);
  }
  I = (size_t)o.ep_itts;
  out.Eft = (double*)calloc((size_t)m.M * T, 8); out.Varft = (double*)calloc((size_t)m.M * T, 8);
  out.ttau = (double*)calloc((size_t)m.M * T, 8); out.tnu = (double*)calloc((size_t)m.M * T, 8);
  out.nlZ = (double*)calloc(I, 8); out.maxDiffP = (double*)calloc(I, 8);
  int64_t counters[NAGP_N_COUNTERS]; out.counters = counters;
  printf("C ABI v%d, %d device(s); kind %d, S=%d M=%d T=%zu\n", nagp_version(), nagp_device_count(), o.kind, m.S, m.M, T);
  if (o.kind == NAGP_KIND_IHGP) {
    tb.r_grid = (double*)dump_load(d, "r", 8, &n); tb.n_grid = (int32_t)n;
    tb.PPlist = (double*)dump_load(d, "PP", 8, &n); tb.PGlist = (double*)dump_load(d, "PG", 8, &n);
    tb.pp_offsets = (int64_t*)dump_load(d, "pp_off", 8, &n); tb.pg_offsets = (int64_t*)dump_load(d, "pg_off", 8, &n);
    st = nagp_ihgp_run(&m, &tb, y, (int64_t)T, &o, &out);
  } else if (o.kind == NAGP_KIND_GIEKF) {
    st = nagp_giekf_run(&m, y, (int64_t)T, &o, &out);
  } else {
    st = nagp_ep_run(&m, y, (int64_t)T, &o, &out);
  }
  if (st != NAGP_OK) { fprintf(stderr, "nagp: %s (%d): %s\n", nagp_strerror(st), st, nagp_last_error()); return 1; }
  {
    double* e = (double*)dump_load(d, "exp_Eft", 8, &n);
    r = rel_diff(out.Eft, e, n, "Eft"); if (r > worst) worst = r;
    e = (double*)dump_load(d, "exp_Varft", 8, &n);
    r = rel_diff(out.Varft, e, n, "Varft"); if (r > worst) worst = r;
    if (o.kind != NAGP_KIND_GIEKF) {
      e = (double*)dump_load(d, "exp_nlZ", 8, &n);
      r = rel_diff(out.nlZ, e, n, "nlZ"); if (r > worst) worst = r;
      e = (double*)dump_load(d, "exp_ttau", 8, &n);
      r = 0.1 * rel_diff(out.ttau, e, n, "ttau"); if (r > worst) worst = r;     /* site tolerance is 10x the mean tolerance */
    }
  }
  /* an invalid call must come back as a status, never crash */
  o.ep_itts = 0;
  st = nagp_ep_run(&m, y, (int64_t)T, &o, &out);
  printf("invalid call -> %d (%s)\n", st, nagp_strerror(st));
  nagp_shutdown();
  printf("worst %.3e\n", worst);
  return (worst < 1e-7 && st != NAGP_OK) ? 0 : 1;
}
