// nagp_api_entry.hpp -- part of the ONE translation unit nagp_api.hip (included there, in this order: nagp_api_plan.hpp, nagp_api_sweep.hpp,
// nagp_api_entry.hpp; the plan struct, the error helpers and the developer-switch accessor live in nagp_api.hip itself).
// The one-shot entry points (ep / ihgp / giekf run), mom on its own, iekf_update1, the stationary filterbank, nagp_batch_run (RCCL), reconstruction.

// ---------------------------------------------------------------------------------------------
static int run_one(const nagp_model* model, const nagp_ihgp_tables* tables, const double* y, int64_t T,
                   const nagp_opts* opts, nagp_out* out) {
  if (!model || !y || !opts || !out) FAIL(NAGP_EINVAL, "null argument");
  nagp_opts o = *opts;
  if (out->PS) o.flags |= 0x4u;
  nagp_plan* p = nullptr;
  int st = nagp_plan_create(&p, 1, model, tables, T, &o);
  if (st != NAGP_OK) return st;
  const double* ys[1] = {y};
  st = nagp_plan_upload_y(p, ys);
  if (st == NAGP_OK && (o.ttau0 || o.tnu0)) {
    const double* t0[1] = {o.ttau0}; const double* n0[1] = {o.tnu0};
    st = nagp_plan_upload_sites(p, t0, n0);
  }
  if (st == NAGP_OK) st = nagp_plan_execute(p);
  if (st == NAGP_OK) st = nagp_plan_download(p, out);
  nagp_plan_destroy(p);
  return st;
}

extern "C" int nagp_ep_run(const nagp_model* model, const double* y, int64_t T, const nagp_opts* opts, nagp_out* out) {
  if (opts && opts->kind != NAGP_KIND_GF_EP) FAIL(NAGP_EINVAL, "nagp_ep_run needs kind = NAGP_KIND_GF_EP");
  return run_one(model, nullptr, y, T, opts, out);
}
extern "C" int nagp_ihgp_run(const nagp_model* model, const nagp_ihgp_tables* tables, const double* y, int64_t T,
                             const nagp_opts* opts, nagp_out* out) {
  if (opts && opts->kind != NAGP_KIND_IHGP) FAIL(NAGP_EINVAL, "nagp_ihgp_run needs kind = NAGP_KIND_IHGP");
  return run_one(model, tables, y, T, opts, out);
}
extern "C" int nagp_giekf_run(const nagp_model* model, const double* y, int64_t T, const nagp_opts* opts, nagp_out* out) {
  if (opts && opts->kind != NAGP_KIND_GIEKF) FAIL(NAGP_EINVAL, "nagp_giekf_run needs kind = NAGP_KIND_GIEKF");
  return run_one(model, nullptr, y, T, opts, out);
}

// ---------------------------------------------------------------------------------------------
// mom on its own (see include/nagp.h)
extern "C" int nagp_mom_eval(const nagp_opts* o, int32_t D, int32_t N, const double* Wnmf, double lik_param, int64_t n,
                             const double* y, const double* mu, const double* s2, double* lZ, double* dlZ, double* d2lZ) {
  if (!o || !y || !mu || !s2 || !lZ || !dlZ || !d2lZ || n < 0) FAIL(NAGP_EINVAL, "null argument");
  if (o->lik_kind < NAGP_LIK_POWER || o->lik_kind > NAGP_LIK_POWER_NMF_SQRT) FAIL(NAGP_EINVAL, "unknown likelihood");
  const bool power = o->lik_kind == NAGP_LIK_POWER;
  const int M = power ? 2 * D : D + N;
  if (D < 1 || M > MAXM || o->n_pts < 1 || !o->wn || !o->xn_unscaled) FAIL(NAGP_EINVAL, "bad sizes / cubature");
  if (power ? (o->cub_dim != D) : (o->cub_dim != N || N < 1 || N > MOM_MAXCD || !Wnmf)) FAIL(NAGP_EUNSUPPORTED, "cub_dim / N / Wnmf");
  if (n == 0) return NAGP_OK;
  if (hipSetDevice(o->device) != hipSuccess) FAIL(NAGP_EHIP, "hipSetDevice(%d)", o->device);
  std::vector<double> xd;
  std::vector<unsigned char> code((size_t)o->n_pts * o->cub_dim);
  for (int pt = 0; pt < o->n_pts; ++pt)
    for (int j = 0; j < o->cub_dim; ++j) {
      const double v = o->xn_unscaled[j + (size_t)o->cub_dim * pt];
      size_t ci = 0;
      while (ci < xd.size() && xd[ci] != v) ++ci;
      if (ci == xd.size()) {
        if (xd.size() == 64) FAIL(NAGP_EUNSUPPORTED, "sigma-point rule has more than 64 distinct coordinate values");
        xd.push_back(v);
      }
      code[(size_t)pt * o->cub_dim + j] = (unsigned char)ci;
    }
  // one device block: wn | xd | code | W | y | mu | s2 | lZ | dl | d2l
  const size_t n_code = (code.size() + 7) / 8 + 1, nW = power ? 0 : (size_t)D * N;
  const size_t o_wn = 0, o_xd = o_wn + o->n_pts, o_code = o_xd + xd.size(), o_W = o_code + n_code, o_y = o_W + nW,
               o_mu = o_y + n, o_s2 = o_mu + (size_t)n * M, o_lZ = o_s2 + (size_t)n * M, o_dl = o_lZ + n, o_d2 = o_dl + (size_t)n * M,
               total = o_d2 + (size_t)n * M;
  double* dev = nullptr;
  if (hipMalloc(&dev, total * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); FAIL(NAGP_ENOMEM, "hipMalloc(%zu)", total * sizeof(double)); }
  std::vector<double> Wr(nW);
  for (int dd = 0; dd < (power ? 0 : D); ++dd)
    for (int j = 0; j < N; ++j) Wr[(size_t)dd * N + j] = Wnmf[dd + (size_t)D * j];
  int st = NAGP_OK;
#define ME_HIP(x) do { if (st == NAGP_OK) { hipError_t _e = (x); if (_e != hipSuccess) { g_last_error = std::string("nagp_mom_eval: " #x " -> ") + hipGetErrorString(_e); st = NAGP_EHIP; } } } while (0)
  ME_HIP(hipMemcpy(dev + o_wn, o->wn, (size_t)o->n_pts * 8, hipMemcpyHostToDevice));
  ME_HIP(hipMemcpy(dev + o_xd, xd.data(), xd.size() * 8, hipMemcpyHostToDevice));
  ME_HIP(hipMemcpy(dev + o_code, code.data(), code.size(), hipMemcpyHostToDevice));
  if (nW) ME_HIP(hipMemcpy(dev + o_W, Wr.data(), nW * 8, hipMemcpyHostToDevice));
  ME_HIP(hipMemcpy(dev + o_y, y, (size_t)n * 8, hipMemcpyHostToDevice));
  ME_HIP(hipMemcpy(dev + o_mu, mu, (size_t)n * M * 8, hipMemcpyHostToDevice));
  ME_HIP(hipMemcpy(dev + o_s2, s2, (size_t)n * M * 8, hipMemcpyHostToDevice));
  MomCfg mc{};
  mc.lik_kind = o->lik_kind; mc.link_kind = o->link_kind; mc.link_shift = o->link_shift;
  mc.n_pts = o->n_pts; mc.cdim = o->cub_dim; mc.D = D; mc.nd = (int)xd.size();
  mc.wn = dev + o_wn; mc.xd = dev + o_xd; mc.code = reinterpret_cast<const unsigned char*>(dev + o_code);
  mc.jitter = power ? 1e-8 : 1e-10; mc.stamps = nullptr;
  mc.DG = pick_DG(o->lik_kind, o->n_pts, 256, D, o->cub_dim);
  mc.cache_tabs = 1; mc.store_a = (o->lik_kind == NAGP_LIK_POWER_NMF_SQRT) ? 1 : 0;
  if (momk_lds_doubles(D, power ? D : N, M, mc) * sizeof(double) > 150 * 1024) mc.store_a = 0;
  if (momk_lds_doubles(D, power ? D : N, M, mc) * sizeof(double) > 150 * 1024) mc.cache_tabs = 0;
  const size_t lds = momk_lds_doubles(D, power ? D : N, M, mc) * sizeof(double);
  if (lds > 160 * 1024) { (void)hipFree(dev); FAIL(NAGP_EUNSUPPORTED, "mom workspace of %zu B exceeds the LDS", lds); }
  MomPar mp{D, power ? 0 : N, M, std::exp(lik_param), o->ep_fraction, nW ? dev + o_W : nullptr, dev + o_y, dev + o_mu, dev + o_s2,
            dev + o_lZ, dev + o_dl, dev + o_d2, n};
  const int grid = (int)std::min<int64_t>(n, 1024);
#define LM(V) do { if (st == NAGP_OK) st = set_lds(mom_kernel<V>, lds); if (st == NAGP_OK) hipLaunchKernelGGL(mom_kernel<V>, dim3(grid), dim3(256), lds, 0, mc, mp); } while (0)
  NAGP_MV_SWITCH9(mom_variant(mc), LM)
#undef LM
  ME_HIP(hipGetLastError());
  ME_HIP(hipDeviceSynchronize());
  ME_HIP(hipMemcpy(lZ, dev + o_lZ, (size_t)n * 8, hipMemcpyDeviceToHost));
  ME_HIP(hipMemcpy(dlZ, dev + o_dl, (size_t)n * M * 8, hipMemcpyDeviceToHost));
  ME_HIP(hipMemcpy(d2lZ, dev + o_d2, (size_t)n * M * 8, hipMemcpyDeviceToHost));
#undef ME_HIP
  (void)hipFree(dev);
  return st;
}

// ---------------------------------------------------------------------------------------------
// iekf_update1 / ekf_update1 on their own (see include/nagp.h)
extern "C" int nagp_iekf_update1(int32_t S, int32_t D, int32_t N, const int32_t* h_col, const double* h_val, const double* Wnmf,
                                 double R, double y, int32_t iters, double* m, double* P, double* K, double* MU, double* Sinn,
                                 int32_t device) {
  if (!h_col || !h_val || !Wnmf || !m || !P) FAIL(NAGP_EINVAL, "null argument");
  const int M = D + N;
  if (S < 1 || S > 512 || D < 1 || N < 1 || M > S || iters < 1) FAIL(NAGP_EINVAL, "bad sizes (S=%d D=%d N=%d iters=%d)", S, D, N, iters);
  for (int n = 0; n < M; ++n)
    if (h_col[n] < 0 || h_col[n] >= S) FAIL(NAGP_EINVAL, "h_col[%d] = %d outside the state", n, h_col[n]);
  if (hipSetDevice(device) != hipSuccess) FAIL(NAGP_EHIP, "hipSetDevice(%d)", device);
  // one device block: m | P | K | MU,S | hval | W | hcol(int)
  const size_t o_m = 0, o_P = o_m + S, o_K = o_P + (size_t)S * S, o_ms = o_K + S, o_hv = o_ms + 2, o_W = o_hv + M,
               o_hc = o_W + (size_t)D * N, total = o_hc + (M + 1) / 2 + 1;
  double* dev = nullptr;
  if (hipMalloc(&dev, total * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); FAIL(NAGP_ENOMEM, "hipMalloc(%zu)", total * sizeof(double)); }
  std::vector<double> Wr((size_t)D * N);
  for (int dd = 0; dd < D; ++dd)
    for (int j = 0; j < N; ++j) Wr[(size_t)dd * N + j] = Wnmf[dd + (size_t)D * j];
  int st = NAGP_OK;
#define EK_HIP(x) do { if (st == NAGP_OK) { hipError_t _e = (x); if (_e != hipSuccess) { g_last_error = std::string("nagp_iekf_update1: " #x " -> ") + hipGetErrorString(_e); st = NAGP_EHIP; } } } while (0)
  EK_HIP(hipMemcpy(dev + o_m, m, (size_t)S * 8, hipMemcpyHostToDevice));
  EK_HIP(hipMemcpy(dev + o_P, P, (size_t)S * S * 8, hipMemcpyHostToDevice));
  EK_HIP(hipMemcpy(dev + o_hv, h_val, (size_t)M * 8, hipMemcpyHostToDevice));
  EK_HIP(hipMemcpy(dev + o_W, Wr.data(), Wr.size() * 8, hipMemcpyHostToDevice));
  EK_HIP(hipMemcpy(dev + o_hc, h_col, (size_t)M * sizeof(int32_t), hipMemcpyHostToDevice));
  EkfPar ep{S, D, N, iters, R, y, reinterpret_cast<const int*>(dev + o_hc), dev + o_hv, dev + o_W, dev + o_m, dev + o_P, dev + o_K, dev + o_ms};
  const size_t lds = (2 * (size_t)S + 2 * M + 2) * sizeof(double);
  if (st == NAGP_OK) hipLaunchKernelGGL(iekf_update1_kernel, dim3(1), dim3(256), lds, 0, ep);
  EK_HIP(hipGetLastError());
  EK_HIP(hipDeviceSynchronize());
  double ms[2] = {0, 0};
  EK_HIP(hipMemcpy(m, dev + o_m, (size_t)S * 8, hipMemcpyDeviceToHost));
  EK_HIP(hipMemcpy(P, dev + o_P, (size_t)S * S * 8, hipMemcpyDeviceToHost));
  if (K) EK_HIP(hipMemcpy(K, dev + o_K, (size_t)S * 8, hipMemcpyDeviceToHost));
  EK_HIP(hipMemcpy(ms, dev + o_ms, 16, hipMemcpyDeviceToHost));
#undef EK_HIP
  if (MU) *MU = ms[0];
  if (Sinn) *Sinn = ms[1];
  (void)hipFree(dev);
  return st;
}

// ---------------------------------------------------------------------------------------------
// stationary filterbank: kernel_ss_kalmanFastFB (see include/nagp.h)
extern "C" int nagp_fastfb_run(int32_t S, const double* A, const double* AKHA, const double* HA, const double* K, const double* G,
                               const double* y, int64_t T, double* MS, double* sum_v2, int32_t device) {
  if (!A || !AKHA || !HA || !K || !y || !MS) FAIL(NAGP_EINVAL, "null argument");
  if (S < 1 || T < 1) FAIL(NAGP_EINVAL, "bad sizes (S=%d T=%lld)", S, (long long)T);
  // S <= 96: both constant S x S matrices of a pass live in the LDS; 96 < S <= 256 (a thread per state): they stay in global memory (L2-resident)
  const int mat_global = (fb_lds_doubles(S) * sizeof(double) > 160 * 1024) ? 1 : 0;
  const size_t lds = fb_lds_doubles(S, mat_global) * sizeof(double);
  if (S > 256) FAIL(NAGP_EUNSUPPORTED, "S=%d: the stationary filterbank runs a thread per state (S <= 256)", S);
  if (hipSetDevice(device) != hipSuccess) FAIL(NAGP_EHIP, "hipSetDevice(%d)", device);
  // spans of the parallel-in-time form (needs two more S x S work matrices in LDS: S <= 64); short series run as one span
  const size_t lds_c = fb_compose_lds_doubles(S) * sizeof(double);
  int ns = 1;
  if (lds_c <= 160 * 1024 && T >= 2048 && !dev_env("NAGP_FB_SEQUENTIAL")) ns = (int)std::min<int64_t>(512, T / 128);
  const int64_t L = (T + ns - 1) / ns;
  ns = (int)((T + L - 1) / L);
  const size_t SS = (size_t)S * S, SP = (size_t)S + 4;
  const size_t o_A = 0, o_B = o_A + SS, o_G = o_B + SS, o_ha = o_G + SS, o_k = o_ha + S, o_y = o_k + S, o_ms = o_y + T,
               o_sv = o_ms + (size_t)T * S, o_phi = o_sv + ns + 1, o_st = o_phi + (size_t)ns * S * SP, total = o_st + (size_t)ns * S + 2;
  double* dev = nullptr;
  if (hipMalloc(&dev, total * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); FAIL(NAGP_ENOMEM, "hipMalloc(%zu)", total * sizeof(double)); }
  int st = NAGP_OK;
#define FB_HIP(x) do { if (st == NAGP_OK) { hipError_t _e = (x); if (_e != hipSuccess) { g_last_error = std::string("nagp_fastfb_run: " #x " -> ") + hipGetErrorString(_e); st = NAGP_EHIP; } } } while (0)
  FB_HIP(hipMemcpy(dev + o_A, A, SS * 8, hipMemcpyHostToDevice));
  FB_HIP(hipMemcpy(dev + o_B, AKHA, SS * 8, hipMemcpyHostToDevice));
  if (G) FB_HIP(hipMemcpy(dev + o_G, G, SS * 8, hipMemcpyHostToDevice));
  FB_HIP(hipMemcpy(dev + o_ha, HA, (size_t)S * 8, hipMemcpyHostToDevice));
  FB_HIP(hipMemcpy(dev + o_k, K, (size_t)S * 8, hipMemcpyHostToDevice));
  FB_HIP(hipMemcpy(dev + o_y, y, (size_t)T * 8, hipMemcpyHostToDevice));
  const int NT = std::max(64, roundup64(S));
  if (st == NAGP_OK) st = set_lds(fastfb_filter_kernel, lds);
  if (st == NAGP_OK) st = set_lds(fastfb_smoother_kernel, lds);
  if (st == NAGP_OK && ns > 1) st = set_lds(fastfb_compose_kernel<false>, lds_c);
  if (st == NAGP_OK && ns > 1) st = set_lds(fastfb_compose_kernel<true>, lds_c);
  if (st == NAGP_OK) {
    FbPar fp{S, T, dev + o_A, dev + o_B, dev + o_ha, dev + o_k, dev + o_y, dev + o_ms, dev + o_sv, L, ns, dev + o_phi,
             ns > 1 ? dev + o_st : nullptr, mat_global};
    if (ns > 1) {
      hipLaunchKernelGGL(fastfb_compose_kernel<false>, dim3(ns), dim3(256), lds_c, 0, fp);
      hipLaunchKernelGGL(fastfb_boundary_kernel<false>, dim3(1), dim3(256), 0, 0, fp);
    }
    hipLaunchKernelGGL(fastfb_filter_kernel, dim3(ns), dim3(NT), lds, 0, fp);
    if (G && T > 1) {
      fp.B = dev + o_G;
      // the T-1 smoothing steps are partitioned with the same span length
      const int nss = (int)((T - 1 + L - 1) / L);
      fp.ns = nss;
      if (ns > 1) {
        hipLaunchKernelGGL(fastfb_compose_kernel<true>, dim3(nss), dim3(256), lds_c, 0, fp);
        hipLaunchKernelGGL(fastfb_boundary_kernel<true>, dim3(1), dim3(256), 0, 0, fp);
      }
      hipLaunchKernelGGL(fastfb_smoother_kernel, dim3(nss), dim3(NT), lds, 0, fp);
    }
  }
  FB_HIP(hipGetLastError());
  FB_HIP(hipDeviceSynchronize());
  FB_HIP(hipMemcpy(MS, dev + o_ms, (size_t)T * S * 8, hipMemcpyDeviceToHost));
  if (sum_v2) {
    std::vector<double> part((size_t)ns);
    FB_HIP(hipMemcpy(part.data(), dev + o_sv, (size_t)ns * 8, hipMemcpyDeviceToHost));
    double acc = 0.0;
    for (int j = 0; j < ns; ++j) acc += part[j];      // fixed order
    *sum_v2 = acc;
  }
#undef FB_HIP
  (void)hipFree(dev);
  return st;
}

// ---------------------------------------------------------------------------------------------
// Multi-GPU batched call (see include/nagp.h): problems round robin over the devices, one host thread + plan per device,
// RCCL all-reduce of the per-sweep nlZ sums.
extern "C" int nagp_batch_partition(int32_t n_problems, int32_t n_gpus, int32_t* dev_of) {
  if (n_problems < 0 || n_gpus < 1 || (n_problems > 0 && !dev_of)) FAIL(NAGP_EINVAL, "bad partition arguments");
  for (int i = 0; i < n_problems; ++i) dev_of[i] = i % n_gpus;      // problem i -> GPU i mod G
  return NAGP_OK;
}

namespace {
struct CommCache {
  std::mutex mu;
  int n = 0;
  std::vector<ncclComm_t> comms;
  std::vector<hipStream_t> streams;
  std::vector<double*> bufs;      // per device: [2 * 64] send | recv
};
CommCache g_cc;

void cc_release_locked() {
  for (size_t d = 0; d < g_cc.comms.size(); ++d) {
    (void)hipSetDevice((int)d);
    if (g_cc.bufs[d]) (void)hipFree(g_cc.bufs[d]);
    if (g_cc.streams[d]) (void)hipStreamDestroy(g_cc.streams[d]);
    if (g_cc.comms[d]) (void)ncclCommDestroy(g_cc.comms[d]);
  }
  g_cc.comms.clear(); g_cc.streams.clear(); g_cc.bufs.clear(); g_cc.n = 0;
}

constexpr int NLZ_MAX = 4096;      // EP sweeps of one call (the reference's drivers use 1 .. 30)
// sum over devices of part[d][0..cnt) with ncclAllReduce; every device ends with the total, device 0's copy is returned
int allreduce_nlz(int G, int cnt, const std::vector<std::vector<double>>& part, std::vector<double>& total) {
  std::lock_guard<std::mutex> lk(g_cc.mu);
  if (cnt > NLZ_MAX) FAIL(NAGP_EUNSUPPORTED, "more than %d EP sweeps in the nlZ reduction", NLZ_MAX);
  if (g_cc.n != G) {
    cc_release_locked();
    g_cc.comms.assign(G, nullptr); g_cc.streams.assign(G, nullptr); g_cc.bufs.assign(G, nullptr);
    std::vector<int> devs(G);
    for (int d = 0; d < G; ++d) devs[d] = d;
    ncclResult_t r = ncclCommInitAll(g_cc.comms.data(), G, devs.data());
    if (r != ncclSuccess) { cc_release_locked(); FAIL(NAGP_ERCCL, "ncclCommInitAll(%d) -> %s", G, ncclGetErrorString(r)); }
    // the cache counts as initialised (g_cc.n = G) only once every per-device stream and buffer exists; a failure on the
    // way releases what was created, so that the next call starts over instead of using null streams / buffers
    for (int d = 0; d < G; ++d) {
      hipError_t e = hipSetDevice(d);
      if (e == hipSuccess) e = hipStreamCreateWithFlags(&g_cc.streams[d], hipStreamNonBlocking);
      if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&g_cc.bufs[d]), 2 * NLZ_MAX * sizeof(double));
      if (e != hipSuccess) {
        cc_release_locked();
        FAIL(e == hipErrorOutOfMemory ? NAGP_ENOMEM : NAGP_EHIP, "per-device resources of the nlZ all-reduce (device %d) -> %s", d, hipGetErrorString(e));
      }
    }
    g_cc.n = G;
  }
  for (int d = 0; d < G; ++d) {
    HIP_TRY(hipSetDevice(d));
    HIP_TRY(hipMemcpyAsync(g_cc.bufs[d], part[d].data(), cnt * sizeof(double), hipMemcpyHostToDevice, g_cc.streams[d]));
  }
  ncclResult_t r = ncclGroupStart();
  for (int d = 0; d < G && r == ncclSuccess; ++d)
    r = ncclAllReduce(g_cc.bufs[d], g_cc.bufs[d] + NLZ_MAX, (size_t)cnt, ncclDouble, ncclSum, g_cc.comms[d], g_cc.streams[d]);
  ncclResult_t r2 = ncclGroupEnd();
  if (r == ncclSuccess) r = r2;
  if (r != ncclSuccess) FAIL(NAGP_ERCCL, "ncclAllReduce -> %s", ncclGetErrorString(r));
  total.assign(cnt, 0.0);
  for (int d = 0; d < G; ++d) {
    HIP_TRY(hipSetDevice(d));
    HIP_TRY(hipStreamSynchronize(g_cc.streams[d]));
  }
  HIP_TRY(hipSetDevice(0));
  HIP_TRY(hipMemcpy(total.data(), g_cc.bufs[0] + NLZ_MAX, cnt * sizeof(double), hipMemcpyDeviceToHost));
  return NAGP_OK;
}
}  // namespace

extern "C" void nagp_shutdown(void) {
  std::lock_guard<std::mutex> lk(g_cc.mu);
  cc_release_locked();
}

extern "C" int nagp_batch_run(int32_t n_problems, const nagp_model* models, const nagp_ihgp_tables* tables, const double* const* ys,
                              int64_t T, const nagp_opts* opts, nagp_out* outs, int32_t n_gpus, double* nlZ_total) {
  if (n_problems < 1 || !models || !ys || !opts || !outs || n_gpus < 1) FAIL(NAGP_EINVAL, "null/empty argument");
  if (opts->kind == NAGP_KIND_IHGP && !tables) FAIL(NAGP_EINVAL, "IHGP tables missing");
  if (opts->ttau0 || opts->tnu0)
    FAIL(NAGP_EINVAL, "nagp_batch_run takes no warm-start sites (opts.ttau0 / tnu0 describe ONE problem): use nagp_plan_create + nagp_plan_upload_sites");
  int ndev = 0, ndev_real = 0;
  if (hipGetDeviceCount(&ndev_real) != hipSuccess) ndev_real = 0;
  (void)hipGetLastError();
  // Test hooks (multi-GPU host logic without the hardware): NAGP_TEST_FAKE_DEVICES=n -- the partition, the per-device threads and the error
  // propagation run for n devices; device d's plan lives on physical device d mod (real devices) (every worker stops at its first device
  // call on a machine without one) and the nlZ sums are added on the host in device order instead of by RCCL (one card cannot hold two
  // ranks of a communicator).  NAGP_TEST_FAIL_DEVICE=d -- worker d reports NAGP_EHIP before it creates its plan.
  const int fake = dev_env("NAGP_TEST_FAKE_DEVICES") ? std::max(0, atoi(dev_env("NAGP_TEST_FAKE_DEVICES"))) : 0;
  const int fail_dev = dev_env("NAGP_TEST_FAIL_DEVICE") ? atoi(dev_env("NAGP_TEST_FAIL_DEVICE")) : -1;
  ndev = fake ? fake : ndev_real;
  if (ndev < 1) FAIL(NAGP_ENODEVICE, "no HIP device visible");
  if (n_gpus > ndev) FAIL(NAGP_EINVAL, "n_gpus = %d but %d device(s) visible", n_gpus, ndev);
  const int G = std::min<int>(n_gpus, n_problems);     // a device without a problem takes no part
  const int I = opts->ep_itts;
  if (I < 1) FAIL(NAGP_EINVAL, "ep_itts < 1");
  std::vector<int32_t> dev_of(n_problems);
  (void)nagp_batch_partition(n_problems, G, dev_of.data());
  std::vector<int> status(G, NAGP_OK);
  std::vector<std::string> errs(G);
  std::vector<std::vector<double>> part(G, std::vector<double>(I, 0.0));
  auto worker = [&](int d) {
    std::vector<int> idx;
    for (int i = 0; i < n_problems; ++i) if (dev_of[i] == d) idx.push_back(i);
    std::vector<nagp_model> ms; std::vector<nagp_ihgp_tables> ts; std::vector<const double*> yv; std::vector<nagp_out> os;
    std::vector<std::vector<double>> nlz(idx.size(), std::vector<double>(I, 0.0));
    for (size_t a = 0; a < idx.size(); ++a) {
      ms.push_back(models[idx[a]]);
      if (tables) ts.push_back(tables[idx[a]]);
      yv.push_back(ys[idx[a]]);
      nagp_out o = outs[idx[a]];
      if (!o.nlZ) o.nlZ = nlz[a].data();        // the reduction needs them whether or not the caller wants them
      os.push_back(o);
    }
    nagp_opts o = *opts;
    o.device = fake ? (ndev_real > 0 ? d % ndev_real : 0) : d; o.ttau0 = nullptr; o.tnu0 = nullptr;
    bool wantPS = false;
    for (const nagp_out& q : os) wantPS = wantPS || q.PS;
    if (wantPS) o.flags |= NAGP_FLAG_WANT_PS;
    nagp_plan* p = nullptr;
    int st = NAGP_OK;
    if (d == fail_dev) { g_last_error = "injected failure (NAGP_TEST_FAIL_DEVICE)"; st = NAGP_EHIP; }
    if (st == NAGP_OK) st = nagp_plan_create(&p, (int32_t)idx.size(), ms.data(), tables ? ts.data() : nullptr, T, &o);
    if (st == NAGP_OK) st = nagp_plan_upload_y(p, yv.data());
    if (st == NAGP_OK) st = nagp_plan_execute(p);
    if (st == NAGP_OK) st = nagp_plan_download(p, os.data());
    if (st == NAGP_OK)
      for (size_t a = 0; a < idx.size(); ++a)
        for (int i = 0; i < I; ++i) part[d][i] += os[a].nlZ[i];     // fixed order: ascending problem index
    if (st != NAGP_OK) errs[d] = g_last_error;                       // thread-local text of this worker
    nagp_plan_destroy(p);
    status[d] = st;
  };
  if (G == 1) {
    worker(0);
  } else {
    std::vector<std::thread> th;
    for (int d = 0; d < G; ++d) th.emplace_back(worker, d);
    for (auto& t : th) t.join();
  }
  for (int d = 0; d < G; ++d)
    if (status[d] != NAGP_OK) { g_last_error = "device " + std::to_string(d) + ": " + errs[d]; return status[d]; }
  std::vector<double> total(I, 0.0);
  if (fake && G > 1) {
    for (int d = 0; d < G; ++d) for (int i = 0; i < I; ++i) total[i] += part[d][i];
  } else if (G > 1 || dev_env("NAGP_FORCE_RCCL")) {
    const int st = allreduce_nlz(G, I, part, total);
    if (st != NAGP_OK) return st;
  } else {
    total = part[0];
  }
  if (nlZ_total) for (int i = 0; i < I; ++i) nlZ_total[i] = total[i];
  return NAGP_OK;
}

// ---------------------------------------------------------------------------------------------
// posterior reconstruction of the signal and the modulator amplitudes (see include/nagp.h, nagp_recon.hpp)
extern "C" int nagp_reconstruct(int32_t D, int32_t N, int64_t T, const double* Eft, const double* Varft, const double* Wnmf,
                                int32_t link_kind, double link_shift, int32_t n_gh, const double* gh_x, const double* gh_w,
                                int32_t n_samples, uint64_t seed, double* Esig, double* Vsig, double* Eft_mod, double* Varft_mod, int32_t device) {
  if (!Eft || !Varft || !Wnmf || !Esig || !Vsig || !Eft_mod || !Varft_mod) FAIL(NAGP_EINVAL, "null argument");
  if (D < 1 || N < 1 || N > MOM_MAXCD || D + N > MAXM || T < 1) FAIL(NAGP_EINVAL, "bad sizes (D=%d N=%d T=%lld)", D, N, (long long)T);
  if (link_kind != NAGP_LINK_SOFTPLUS && link_kind != NAGP_LINK_EXP) FAIL(NAGP_EINVAL, "unknown link");
  const bool sampling = n_samples > 0;
  if (sampling && n_samples < 2) FAIL(NAGP_EINVAL, "sampling needs at least two draws");
  if (!sampling && link_kind == NAGP_LINK_SOFTPLUS && (n_gh < 1 || n_gh > 256 || !gh_x || !gh_w)) FAIL(NAGP_EINVAL, "Gauss-Hermite rule missing");
  if (hipSetDevice(device) != hipSuccess) FAIL(NAGP_EHIP, "hipSetDevice(%d)", device);
  const int M = D + N;
  const size_t nW = (size_t)D * N, nMT = (size_t)M * T, ngh = sampling ? 0 : (size_t)std::max(n_gh, 0);
  const size_t o_W = 0, o_E = o_W + nW, o_V = o_E + nMT, o_gx = o_V + nMT, o_gw = o_gx + ngh, o_es = o_gw + ngh, o_vs = o_es + T,
               o_em = o_vs + T, o_vm = o_em + (size_t)N * T, total = o_vm + (size_t)N * T;
  double* dev = nullptr;
  if (hipMalloc(&dev, total * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); FAIL(NAGP_ENOMEM, "hipMalloc(%zu)", total * sizeof(double)); }
  std::vector<double> Wr(nW);
  for (int d = 0; d < D; ++d)
    for (int j = 0; j < N; ++j) Wr[(size_t)d * N + j] = Wnmf[d + (size_t)D * j];
  int st = NAGP_OK;
#define RC_HIP(x) do { if (st == NAGP_OK) { hipError_t _e = (x); if (_e != hipSuccess) { g_last_error = std::string("nagp_reconstruct: " #x " -> ") + hipGetErrorString(_e); st = NAGP_EHIP; } } } while (0)
  RC_HIP(hipMemcpy(dev + o_W, Wr.data(), nW * 8, hipMemcpyHostToDevice));
  RC_HIP(hipMemcpy(dev + o_E, Eft, nMT * 8, hipMemcpyHostToDevice));      // M x T column-major = [T][M]
  RC_HIP(hipMemcpy(dev + o_V, Varft, nMT * 8, hipMemcpyHostToDevice));
  if (ngh) { RC_HIP(hipMemcpy(dev + o_gx, gh_x, ngh * 8, hipMemcpyHostToDevice)); RC_HIP(hipMemcpy(dev + o_gw, gh_w, ngh * 8, hipMemcpyHostToDevice)); }
  ReconPar rp{D, N, M, T, link_kind, link_shift, dev + o_W, dev + o_E, dev + o_V, (int)ngh, dev + o_gx, dev + o_gw, n_samples, seed,
              dev + o_es, dev + o_vs, dev + o_em, dev + o_vm};
  if (st == NAGP_OK) {
    if (sampling) {
      const unsigned grid = (unsigned)std::min<int64_t>(T, 65536);
      hipLaunchKernelGGL(recon_sample_kernel, dim3(grid), dim3(64), nW * sizeof(double), 0, rp);
    } else {
      hipLaunchKernelGGL(recon_moments_kernel, dim3((unsigned)((T + 255) / 256)), dim3(256), (nW + 2 * ngh) * sizeof(double), 0, rp);
    }
  }
  RC_HIP(hipGetLastError());
  RC_HIP(hipDeviceSynchronize());
  RC_HIP(hipMemcpy(Esig, dev + o_es, (size_t)T * 8, hipMemcpyDeviceToHost));
  RC_HIP(hipMemcpy(Vsig, dev + o_vs, (size_t)T * 8, hipMemcpyDeviceToHost));
  RC_HIP(hipMemcpy(Eft_mod, dev + o_em, (size_t)N * T * 8, hipMemcpyDeviceToHost));
  RC_HIP(hipMemcpy(Varft_mod, dev + o_vm, (size_t)N * T * 8, hipMemcpyDeviceToHost));
#undef RC_HIP
  (void)hipFree(dev);
  return st;
}

