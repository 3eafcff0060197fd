// nagp_api.hip -- C ABI (include/nagp.h) and host orchestration of the HIP kernels.
// One plan = n_problems independent problems of identical shape; everything between
// nagp_plan_upload_y and nagp_plan_download stays in HBM.  The sweep structure follows
// matlab/gf_ep_modulator_nmf.m:113-283 (predict) / :384-522 (nlml), ihgp_ep_modulator_nmf.m:223-454
// and gf_giekf_modulator_nmf.m:126-221.
#include "nagp_inst.hpp"
#include "nagp_recon.hpp"
#include "../../include/nagp.h"

// every templated kernel is instantiated in one of the inst_*.hip translation units
NAGP_LIST_ALL(extern template __global__)

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <mutex>
#include <set>
#include <string>
#include <functional>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

using namespace nagp;

static thread_local std::string g_last_error;

// Developer switches and test hooks.  The library is meant to live inside a long-running host process (MATLAB): a stray NAGP_* variable
// in that environment must not change what it computes, so NONE of them is read unless NAGP_DEVELOPER=1 is set as well (tests/conftest.py
// and the scripts under tools/ set it; bench.py and the MEX gateway never do).  The switches that make results meaningless (phase-skipping
// timing probes) and the test hooks that replace devices or fail allocations say so on stderr once when they are active.
static const char* dev_env(const char* name) {
  static const bool on = [] { const char* d = getenv("NAGP_DEVELOPER"); return d && d[0] == '1' && d[1] == 0; }();
  if (!on) return nullptr;
  const char* v = getenv(name);
  if (v && (!strcmp(name, "NAGP_FILTER_DBG") || !strncmp(name, "NAGP_TEST_", 10))) {
    static std::mutex mu; static std::set<std::string> said;
    std::lock_guard<std::mutex> lk(mu);
    if (said.insert(name).second)
      fprintf(stderr, "[nagp] developer switch %s=%s is active (%s)\n", name, v,
              !strcmp(name, "NAGP_FILTER_DBG") ? "phases of the filter step are skipped: results are garbage, timing only" : "test hook: devices / allocations are not the real ones");
  }
  return v;
}

#define HIP_TRY(expr)                                                                         \
  do {                                                                                        \
    hipError_t _e = (expr);                                                                   \
    if (_e != hipSuccess) {                                                                   \
      char _b[512];                                                                           \
      snprintf(_b, sizeof _b, "%s:%d %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
      g_last_error = _b;                                                                      \
      (void)hipGetLastError();   /* the runtime's last-error slot is sticky: a later hipGetLastError() after a launch must not see this one */ \
      return (_e == hipErrorOutOfMemory) ? NAGP_ENOMEM : NAGP_EHIP;                           \
    }                                                                                         \
  } while (0)

#define FAIL(code, ...)                          \
  do {                                           \
    char _b[512];                                \
    snprintf(_b, sizeof _b, __VA_ARGS__);        \
    g_last_error = _b;                           \
    return (code);                               \
  } while (0)

struct EvRec { int kid; hipEvent_t a, b; };

struct nagp_plan {
  Shape sh{};
  nagp_opts opts{};
  std::vector<double> damping;
  int B = 0;
  int TPT = 1, TPT_f = 1, NT = 256, NT_f = 256, NT_ih = 256;
  int TPT_a = 1, NT_a = 256, LB_a = 256;   // ADF (mom) launches of the gf filter
  int NT_fl = 256;                         // threads of the (not wide) fixed-site launches
  int wide_l = 0, NT_l = 256;              // fixed-site launches of models with 512 < tiles <= 1024: one tile per thread, 1024-thread bound
  int chunk = 2048, LP1 = 1, LP2 = 1, ns_max = 1;
  SpanPar spar{};
  int mfma_sp = 0;      // > 0: FP64-MFMA smoother passes on dense Sp x Sp matrices
  int big_sp = 0;       // 1: Sp > 96, the column-owner kernels of nagp_mfma_big.hpp
  int gain768 = 0;      // rts_gain_kernel<2, 768>: 1025..1536 tiles with a lower triangle of <= 768 tiles
  int gain_mfma = 0;    // dense (G, Delta) output through rts_gain_mfma_kernel<Sp/16> (nagp_gain_mfma.hpp)
  int gain_inv = 0;     // ... in its explicit-inverse form G = A^-1 - (A^-1 Q) PSkp^-1 (every block of A comfortably invertible)
  double* d_ainv = nullptr;      // [B][M][32]: per block A^-1 and A^-1 Q (GainPar::ainv)
  int lin_mfma = 0;     // fixed-site filter launches through gf_filter_lin_mfma_kernel<NTL> (nagp_filter_mfma.hpp); = NTL
  size_t lds_lin = 0;
  size_t gbuf_doubles = 0;
  MfmaPar mpar{};
  size_t lds_mfma = 0;
  int hph_lds = 0, sta_f = 0, sta_ep = 0, DG_f = 1, DG_ep = 1, cache_f = 0, cache_ep = 0, kb_f = 16;
  bool want_PS = false;
  bool need_PF = false;
  MomSrc src_all{};     // block structure of Wnmf (n_src >= 2) and which kernels use it
  int src_f = 0, src_ep = 0, kb_ih = 16, chunk_cap_f = 0;
  MomSp sp{};           // sparse-point form of likModulatorNMFPower (nagp_momsp.hpp); sp_ih: the IHGP ADF sweep uses it
  int sp_ih = 0, sp_gf = 0, kb_sp = 16, hph_sp = 1; size_t lds_sp = 0;
  int sp_ep = 0; size_t lds_ep_sp = 0;      // site refresh (ep_site_sp_kernel) in the sparse-point form
  int sq_c0 = -1, sq_ok = 0, sq_ih = 0, kb_sq = 16, hph_sq = 1; size_t lds_sq = 0; int sq_ep = 0; size_t lds_ep_sq = 0; int sq_gf = 0;   // likModulatorPreCalcwn in the staged form (nagp_momsq.hpp): centre code, rule fits, IHGP ADF sweep uses it
  int a8_gf = 0, a8_pack = 0, a8_tpt = 1, a8_st = 0, kb_a8 = 16; size_t lds_a8 = 0;   // ADF sweep of the gf filter with role-specialised waves (gf_adf8_kernel)
  int sp_ih8 = 0, sp_pack = 0, sp_maxmem = 0; size_t lds_sp8 = 0;   // sp_maxmem: most points sharing one non-centre (dimension, coordinate)   // the role-specialised 512-thread form of the same sweep (ihgp_adf8_kernel)
  hipStream_t stream = nullptr;
  // chunk-pipelined smoother (gf / giekf): while the sequential filter occupies one CU per problem, the parallel smoother kernels
  // of the chunks it has finished (rts_gain + the compose pass) run on `stream2` on the rest of the chip
  hipStream_t stream2 = nullptr;
  bool pipeline = false;
  int nc = 1;                                   // smoother chunks per sweep
  std::vector<double*> slotG, slotD;            // (G, Delta) / delta chunk buffers; slot 0 doubles as the scratch of non-retained chunks
  std::vector<char> slot_tiled;                 // the slot last held tile-major matrices: zero it before the next dense use (padding rows)
  std::vector<int> slot_cap;                    // capacity of a slot in steps (= stride between its problems): `chunk`, except the small
                                                // last slot that belongs to the short chunk of the latest steps
  std::vector<size_t> slot_gps;                 // Bufs::gpstride of the slot: 0, or the PF stride of a slot recycled from PF
  int n_full_slots = 0, n_recycled = 0;         // full-size slots of their own; slots inside PF (behind the small slot in the vectors)
  size_t mat_doubles = 0;                       // doubles of one dense / tile-major matrix in a slot
  size_t gstep = 0; int dpacked = 0;            // doubles of (G, Delta) of one step in a slot; Delta as packed lower 16x16 tiles (GainPar::dpacked)
  std::vector<double*> c_spanbuf, c_spanvec, c_mspanbuf, c_mspanvec;   // compose results per chunk (VALU / MFMA layouts)
  std::vector<double*> c_bnd, c_mbnd;           // boundary values (E_top, e_top of every span) per chunk: the apply passes of several chunks
                                                // run side by side on `s_apply` once the (sequential) boundary chain has passed them
  std::vector<hipStream_t> s_apply;             // [0]: the merged apply launch of the chunks with their own (G, Delta) buffer
  std::vector<hipEvent_t> ev_bnd, ev_app;       // [0]: boundary chain of those chunks done (main stream) / merged apply done (side stream)
  ChunkTab* h_tab = nullptr;                    // pinned host memory [nc]: chunk table of the merged apply launch
  std::vector<double*> c_xbuf;                  // per chunk as well: the VALU compose pass uses it as per-workgroup scratch, and the compose
                                                // passes of two chunks may run at the same time (one on each stream)
  unsigned long long* h_progress = nullptr;     // pinned host memory [B]: steps the filter has finished (FilterPar::progress)
  hipEvent_t ev_filter = nullptr, ev_s2 = nullptr;
  // cross-sweep schedule (gf): the apply pass and the site refresh of a sweep run chunk by chunk, earliest steps first, and the next
  // sweep's filter follows them chunk by chunk (ev_chunk[c]); the per-sweep reductions go to their own records of red_all
  bool xsweep = false;
  std::vector<hipEvent_t> ev_chunk; hipEvent_t ev_red = nullptr;
  double* red_all = nullptr; double* red0 = nullptr;     // [ep_itts + 2][B][8] ; the plan's single record (giekf, ihgp)
  Bufs b{};
  MomCfg mc{};
  IhgpTabs tb{};
  double* d_tt0 = nullptr; double* d_tn0 = nullptr; bool warm = false;   // warm start (nagp_plan_upload_sites)
  double* d_model = nullptr; double* d_y = nullptr; double* d_wn = nullptr; double* d_xi = nullptr;
  double* d_gstamps = nullptr;   // per-phase cycle counters of rts_gain_kernel (GainPar::stamps)
  double* d_stamps = nullptr; double* d_lZs = nullptr; double* d_affspan = nullptr; double* d_affbnd = nullptr; int aff_L = 128, aff_ns = 1; double* d_vprev = nullptr; double* d_tab = nullptr; double* d_r = nullptr;
  std::vector<void*> allocs;
  std::vector<size_t> alloc_bytes;     // bytes of allocs[i]
  int64_t dev_bytes = 0;
  std::vector<double> nlZ, mdM, mdP;   // [B][ep_itts]
  std::vector<EvRec> evs;
  std::vector<hipEvent_t> ev_pool; size_t ev_next = 0;
  hipEvent_t ev_t0 = nullptr, ev_t1 = nullptr;
  nagp_timings tim{};
  std::vector<double> h_hval;          // [B][M]
  size_t lds_filter = 0, lds_gain = 0, lds_scan = 0, lds_ep = 0, lds_ih = 0;
};

extern "C" int nagp_version(void) { return NAGP_VERSION; }
extern "C" int nagp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}
extern "C" const char* nagp_last_error(void) { return g_last_error.c_str(); }
// (other translation units of the library record their error text here; not part of the ABI)
extern "C" __attribute__((visibility("hidden"))) void nagp_internal_set_error(const char* msg) { g_last_error = msg ? msg : ""; }
extern "C" const char* nagp_strerror(int s) {
  switch (s) {
    case NAGP_OK: return "ok";
    case NAGP_EINVAL: return "invalid argument";
    case NAGP_EUNSUPPORTED: return "unsupported shape";
    case NAGP_EHIP: return "HIP runtime error";
    case NAGP_ENOMEM: return "out of device memory";
    case NAGP_ENODEVICE: return "no HIP device";
    case NAGP_ENOTPD: return "matrix not positive definite";
    case NAGP_ERCCL: return "RCCL error";
    default: return "unknown";
  }
}

// length of the smoother chunk that ends at step k1 (exclusive); `latest`: the chunk of the latest steps, cut short (see sweep_begin)
static int chunk_len(const nagp_plan* p, int64_t k1, bool latest) {
  const bool c0 = latest && (p->sh.T - 1) > p->chunk;
  return (int)std::min<int64_t>(c0 ? std::min(p->chunk, std::max(64, p->chunk / 8)) : p->chunk, k1);
}

static int dalloc(nagp_plan* p, double** ptr, size_t n_doubles, bool zero = true) {
  void* v = nullptr;
  const size_t bytes = (n_doubles ? n_doubles : 1) * sizeof(double);
  HIP_TRY(hipMalloc(&v, bytes));
  p->allocs.push_back(v);
  p->alloc_bytes.push_back(bytes);
  p->dev_bytes += (int64_t)bytes;
  if (zero) HIP_TRY(hipMemsetAsync(v, 0, bytes, p->stream));
  *ptr = static_cast<double*>(v);
  return NAGP_OK;
}

// give back one allocation of the plan (best-effort buffers that turned out not to fit)
static void dfree(nagp_plan* p, double* ptr) {
  if (!ptr) return;
  for (size_t i = p->allocs.size(); i-- > 0;)
    if (p->allocs[i] == static_cast<void*>(ptr)) {
      p->dev_bytes -= (int64_t)p->alloc_bytes[i];
      p->allocs.erase(p->allocs.begin() + (long)i); p->alloc_bytes.erase(p->alloc_bytes.begin() + (long)i);
      break;
    }
  (void)hipStreamSynchronize(p->stream);          // (a memset of the buffer may still be queued)
  (void)hipFree(ptr);
}

// run CALL(MV) for the mom variant mv (0 = POWER, 1..9 = NMF cubature dimension; 9 = three sources x three components of the
// source-separation mixtures, experiments/source_sep_piano.m:78-90)
#define NAGP_MV_SWITCH(mv, CALL)                                                                         \
  switch (mv) {                                                                                          \
    case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;      \
    case 4: CALL(4); break; case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break;      \
    case 8: CALL(8); break; default: CALL(9); break;                                                     \
  }

// the kernels without covariance tiles (IHGP filter, site refresh, mom) also exist for N = 9 (three sources x three
// components of the source-separation mixtures, experiments/source_sep_piano.m:78-90)
#define NAGP_MV_SWITCH9(mv, CALL)                                                                        \
  switch (mv) {                                                                                          \
    case 0: CALL(0); break; case 1: CALL(1); break; case 2: CALL(2); break; case 3: CALL(3); break;      \
    case 4: CALL(4); break; case 5: CALL(5); break; case 6: CALL(6); break; case 7: CALL(7); break;      \
    case 8: CALL(8); break; default: CALL(9); break;                                                     \
  }

// Block structure of Wnmf (source-separation mixtures): the finest partition into contiguous (sub-band range,
// component range) blocks that holds the non-zeros of every problem of the plan; the tables of MomSrc (nagp_dev.hpp).
// Returns false (unstructured) whenever anything does not fit the table formats.
static bool build_mom_src(int B, const nagp_model* models, int D, int N, int n_pts, const std::vector<unsigned char>& code,
                          MomSrc& sc, std::vector<unsigned char>& blob) {
  sc = MomSrc{};
  if (N < 2 || D < 2 || n_pts > 65535 || dev_env("NAGP_NO_SRC")) return false;
  std::vector<int> par(D + N);
  for (int i = 0; i < D + N; ++i) par[i] = i;
  auto find = [&](int x) { while (par[x] != x) { par[x] = par[par[x]]; x = par[x]; } return x; };
  for (int q = 0; q < B; ++q)
    for (int n = 0; n < N; ++n)
      for (int d = 0; d < D; ++d)
        if (models[q].Wnmf[d + (size_t)D * n] != 0.0) par[find(d)] = find(D + n);
  // sources in order of appearance along d; both index sets must be contiguous and in the same order
  std::vector<int> root;
  int d0[MOM_MAXSRC + 1] = {0}, n0[MOM_MAXSRC + 1] = {0};
  for (int d = 0; d < D; ++d) {
    const int r = find(d);
    if (root.empty() || root.back() != r) {
      for (int x : root) if (x == r) return false;          // came back to an earlier source
      if ((int)root.size() == MOM_MAXSRC) return false;
      d0[root.size()] = d; root.push_back(r);
    }
  }
  const int ns = (int)root.size();
  if (ns < 2) return false;
  d0[ns] = D;
  int cur = -1;
  for (int n = 0; n < N; ++n) {
    const int r = find(D + n);
    int j = -1;
    for (int x = 0; x < ns; ++x) if (root[x] == r) j = x;
    if (j < 0 || j < cur || j > cur + 1) return false;      // a component without sub-bands, or out of order
    if (j == cur + 1) { n0[j] = n; cur = j; }
  }
  if (cur != ns - 1) return false;
  n0[ns] = N;
  // tuples
  std::vector<unsigned char> tup((size_t)ns * n_pts);
  std::vector<std::vector<std::vector<unsigned char>>> tuples(ns);
  for (int j = 0; j < ns; ++j) {
    const int nc = n0[j + 1] - n0[j];
    for (int pt = 0; pt < n_pts; ++pt) {
      std::vector<unsigned char> key(code.begin() + (size_t)pt * N + n0[j], code.begin() + (size_t)pt * N + n0[j] + nc);
      size_t b = 0;
      while (b < tuples[j].size() && tuples[j][b] != key) ++b;
      if (b == tuples[j].size()) { if (b == 255) return false; tuples[j].push_back(key); }
      tup[(size_t)j * n_pts + pt] = (unsigned char)b;
    }
  }
  int nbmax = 0, dlmax = 0;
  for (int j = 0; j < ns; ++j) { nbmax = std::max(nbmax, (int)tuples[j].size()); dlmax = std::max(dlmax, d0[j + 1] - d0[j]); }
  std::vector<unsigned char> tcode((size_t)ns * nbmax * N, 0);
  for (int j = 0; j < ns; ++j)
    for (size_t b = 0; b < tuples[j].size(); ++b)
      for (size_t k = 0; k < tuples[j][b].size(); ++k) tcode[((size_t)j * nbmax + b) * N + k] = tuples[j][b][k];
  // points ordered by tuple, bins cut into slices of <= 64 points (4 trips of a 16-lane group)
  std::vector<unsigned short> perm((size_t)ns * n_pts);
  struct Item { int j, b, start, len, bin_order; };
  std::vector<Item> items;
  for (int j = 0; j < ns; ++j) {
    int pos = 0;
    for (int b = 0; b < (int)tuples[j].size(); ++b) {
      const int start = pos;
      for (int pt = 0; pt < n_pts; ++pt) if (tup[(size_t)j * n_pts + pt] == b) perm[(size_t)j * n_pts + pos++] = (unsigned short)pt;
      for (int s0 = start; s0 < pos; s0 += 64) items.push_back({j, b, s0, std::min(64, pos - s0), (int)items.size()});
    }
  }
  std::stable_sort(items.begin(), items.end(), [](const Item& a, const Item& b) { return a.len > b.len; });
  std::vector<int> it4(items.size() * 4), bin_first((size_t)ns * nbmax + 1, 0), bitem(items.size());
  for (size_t e = 0; e < items.size(); ++e) {
    it4[4 * e] = items[e].j; it4[4 * e + 1] = items[e].b; it4[4 * e + 2] = items[e].start; it4[4 * e + 3] = items[e].len;
    ++bin_first[(size_t)items[e].j * nbmax + items[e].b + 1];
  }
  for (size_t t = 0; t < (size_t)ns * nbmax; ++t) bin_first[t + 1] += bin_first[t];
  {   // items of a bin in slice order (ascending start)
    std::vector<int> fill(bin_first.begin(), bin_first.end() - 1);
    std::vector<int> order(items.size());
    for (size_t e = 0; e < items.size(); ++e) order[e] = (int)e;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return items[a].bin_order < items[b].bin_order; });
    for (int e : order) bitem[fill[(size_t)items[e].j * nbmax + items[e].b]++] = e;
  }
  sc.n_src = ns; sc.nbmax = nbmax; sc.dlmax = dlmax | 1; sc.n_items = (int)items.size();   // odd row stride: lanes = tuples hit distinct LDS banks
  for (int j = 0; j <= ns; ++j) { sc.d0[j] = d0[j]; sc.n0[j] = n0[j]; }
  for (int j = 0; j < ns; ++j) sc.nb[j] = (int)tuples[j].size();
  auto put = [&](const void* src, size_t bytes) { const size_t o = blob.size(); blob.resize(o + ((bytes + 15) / 16) * 16); std::memcpy(blob.data() + o, src, bytes); return o; };
  sc.off[0] = (int)put(tup.data(), tup.size());
  sc.off[1] = (int)put(tcode.data(), tcode.size());
  sc.off[2] = (int)put(perm.data(), perm.size() * sizeof(unsigned short));
  sc.off[3] = (int)put(it4.data(), it4.size() * sizeof(int));
  sc.off[4] = (int)put(bin_first.data(), bin_first.size() * sizeof(int));
  sc.off[5] = (int)put(bitem.data(), bitem.size() * sizeof(int));
  sc.blob_bytes = (int)blob.size();
  return true;
}

static int roundup64(int x) { return ((x + 63) / 64) * 64; }

// lanes per sigma point in mom (see nagp_dev.hpp).  POWER: as many as keep one trip over the points.
// NMF: every lane keeps the W rows of <= MOM_NDM sub-bands in registers, and the lane sets 0..cdim of
// phase 2 own the modulator outputs and Z.
static int pick_DG(int lik_kind, int n_pts, int NT, int D, int cdim) {
  if (lik_kind == NAGP_LIK_POWER) {
    int dg = 1;
    while (dg * 2 <= 16 && dg * 2 <= D && (long long)n_pts * dg * 2 <= NT) dg *= 2;
    return dg;
  }
  int dg = 4;
  while (dg < 16 && (dg * MOM_NDM < D || dg < cdim + 1)) dg *= 2;
  return dg;
}

static hipEvent_t next_event(nagp_plan* p) {
  if (p->ev_next == p->ev_pool.size()) {
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return nullptr;
    p->ev_pool.push_back(e);
  }
  return p->ev_pool[p->ev_next++];
}
struct Timed {
  nagp_plan* p; int kid; hipEvent_t a, b; hipStream_t st;
  Timed(nagp_plan* p_, int kid_, hipStream_t st_ = nullptr) : p(p_), kid(kid_), st(st_ ? st_ : p_->stream) {
    a = next_event(p); b = next_event(p);
    if (a) (void)hipEventRecord(a, st);
  }
  ~Timed() {
    if (b) (void)hipEventRecord(b, st);
    if (a && b) p->evs.push_back({kid, a, b});
  }
};

// The dynamic-LDS limit of a kernel is a per-process attribute: it is raised to the full 160 KiB once and never lowered, so that
// plans with different LDS needs can be alive at the same time (a later, smaller plan must not shrink it under a live one).
template <typename K>
static int set_lds(K kernel, size_t bytes) {
  if (bytes > 160 * 1024) FAIL(NAGP_EUNSUPPORTED, "kernel needs %zu B of LDS (> 160 KiB)", bytes);
  if (bytes > 48 * 1024)
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  return NAGP_OK;
}

// ---------------------------------------------------------------------------------------------
extern "C" int nagp_plan_create(nagp_plan** out, int32_t B, const nagp_model* models, const nagp_ihgp_tables* tables,
                                int64_t T, const nagp_opts* o) {
  if (!out || !models || !o || B < 1 || T < 1) FAIL(NAGP_EINVAL, "null/empty argument");
  *out = nullptr;
  const nagp_model& m0 = models[0];
  if (!m0.block_offsets) FAIL(NAGP_EINVAL, "problem 0: NULL block_offsets");
  if (m0.M < 1 || m0.M > MAXM) FAIL(NAGP_EUNSUPPORTED, "M=%d outside 1..%d", m0.M, MAXM);
  if (m0.S < m0.M || m0.S > 1024) FAIL(NAGP_EUNSUPPORTED, "S=%d unsupported", m0.S);
  if (o->kind != NAGP_KIND_GF_EP && o->kind != NAGP_KIND_IHGP && o->kind != NAGP_KIND_GIEKF) FAIL(NAGP_EINVAL, "kind");
  if (o->kind == NAGP_KIND_IHGP && o->mode != NAGP_MODE_PREDICT)
    FAIL(NAGP_EUNSUPPORTED, "the reference's IHGP nlml mode is broken (SURVEY C-11)");
  if (o->ep_itts < 1) FAIL(NAGP_EINVAL, "ep_itts < 1");
  const bool ekf = (o->kind == NAGP_KIND_GIEKF);
  if (!ekf) {
    if (!o->wn || !o->xn_unscaled || o->n_pts < 1 || !o->ep_damping) FAIL(NAGP_EINVAL, "cubature/damping missing");
    if (o->lik_kind == NAGP_LIK_POWER) {
      if (m0.M != 2 * m0.D || o->cub_dim != m0.D) FAIL(NAGP_EINVAL, "POWER likelihood needs M=2D, cub_dim=D");
    } else {
      if (m0.M != m0.D + m0.N || o->cub_dim != m0.N || !m0.Wnmf) FAIL(NAGP_EINVAL, "NMF likelihood needs M=D+N, cub_dim=N, Wnmf");
      const int nmax = (o->kind == NAGP_KIND_IHGP) ? MOM_MAXCD : MOM_MAXCD_GF;
      if (m0.N > nmax) FAIL(NAGP_EUNSUPPORTED, "N=%d > %d NMF components", m0.N, nmax);
    }
  } else {
    if (m0.M != m0.D + m0.N || !m0.Wnmf || o->l_iter < 1) FAIL(NAGP_EINVAL, "EKF needs M=D+N, Wnmf, l_iter>=1");
  }
  if ((o->flags & NAGP_FLAG_MIXTURE_RULE) && (ekf || o->mode != NAGP_MODE_PREDICT || o->lik_kind == NAGP_LIK_POWER))
    FAIL(NAGP_EINVAL, "the mixture EP rule exists for the NMF likelihoods in predict mode only (gf_ep_mods_nmf_mixture.m:376)");
  if (o->kind == NAGP_KIND_IHGP && !tables) FAIL(NAGP_EINVAL, "IHGP tables missing");

  nagp_plan* p = new nagp_plan();
  p->opts = *o;
  p->B = B;
  Shape& sh = p->sh;
  sh.S = m0.S; sh.M = m0.M; sh.D = m0.D; sh.N = (o->lik_kind == NAGP_LIK_POWER && !ekf) ? m0.D : m0.N;
  sh.T = T; sh.ntiles = m0.M * m0.M;
  for (int n = 0; n <= m0.M; ++n) sh.off[n] = m0.block_offsets[n];
  for (int n = 0; n < m0.M; ++n) {
    sh.bsz[n] = sh.off[n + 1] - sh.off[n];
    if (sh.bsz[n] < 1 || sh.bsz[n] > 4) { const int bsn = sh.bsz[n]; delete p; FAIL(NAGP_EUNSUPPORTED, "block %d has size %d (supported: 1..4)", n, bsn); }
  }
  if (sh.off[0] != 0 || sh.off[m0.M] != m0.S) { delete p; FAIL(NAGP_EINVAL, "block_offsets do not span 0..S"); }
  for (int q = 0; q < B; ++q) {   // every pointer the packing below dereferences
    const nagp_model& mq = models[q];
    if (!mq.A || !mq.Q || !mq.Pinf || !mq.h_val || !mq.block_offsets) { delete p; FAIL(NAGP_EINVAL, "problem %d: NULL A / Q / Pinf / h_val / block_offsets", q); }
    if ((ekf || o->lik_kind != NAGP_LIK_POWER) && !mq.Wnmf) { delete p; FAIL(NAGP_EINVAL, "problem %d: Wnmf missing", q); }
    if (o->kind == NAGP_KIND_IHGP && (!tables[q].r_grid || !tables[q].PPlist || !tables[q].PGlist || !tables[q].pp_offsets || !tables[q].pg_offsets)) {
      delete p; FAIL(NAGP_EINVAL, "problem %d: NULL IHGP table pointer", q);
    }
  }
  for (int q = 1; q < B; ++q) {
    const nagp_model& mq = models[q];
    bool same = mq.S == m0.S && mq.M == m0.M && mq.D == m0.D && mq.N == m0.N;
    for (int n = 0; same && n <= m0.M; ++n) same = mq.block_offsets[n] == m0.block_offsets[n];
    if (!same) { delete p; FAIL(NAGP_EINVAL, "problem %d has a different shape", q); }
  }
  if (!ekf) p->damping.assign(o->ep_damping, o->ep_damping + o->ep_itts);
  p->want_PS = true;   // smoothed covariances are cheap to keep only if asked; decided at download (see below)

#define PLAN_TRY(expr) do { int _s = (expr); if (_s != NAGP_OK) { nagp_plan_destroy(p); return _s; } } while (0)
#define PLAN_HIP(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { char _b[512]; snprintf(_b, sizeof _b, "%s -> %s", #expr, hipGetErrorString(_e)); g_last_error = _b; (void)hipGetLastError(); nagp_plan_destroy(p); return _e == hipErrorOutOfMemory ? NAGP_ENOMEM : NAGP_EHIP; } } while (0)

  // the device is looked at only after every pure-host check has passed (those run under ASan on GPU-less machines)
  {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1) { delete p; FAIL(NAGP_ENODEVICE, "no HIP device visible"); }
    if (o->device < 0 || o->device >= ndev) { delete p; FAIL(NAGP_EINVAL, "device ordinal %d out of range", o->device); }
  }
  PLAN_HIP(hipSetDevice(o->device));
  PLAN_HIP(hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking));
  PLAN_HIP(hipEventCreate(&p->ev_t0));
  PLAN_HIP(hipEventCreate(&p->ev_t1));

  // ---- launch geometry
  const int nt = sh.ntiles;
  p->TPT = (nt + 511) / 512;   // <= 512 threads per workgroup: 256 VGPRs per lane for the register-resident tiles
  const bool ih = (o->kind == NAGP_KIND_IHGP);   // no covariance tiles: the tile-count limits below do not apply
  if (ih) p->TPT = std::min(p->TPT, 4);
  if (!ih && p->TPT > 4 && p->TPT <= 8) p->TPT = 8;     // instantiated: 1 .. 4 and 8 tiles per thread (8: scratch-resident tiles, 46 .. 64 sites)
  if (!ih && (p->TPT > 8 || sh.S > 512)) { const int Mx = sh.M, Sx = sh.S; nagp_plan_destroy(p); FAIL(NAGP_EUNSUPPORTED, "M=%d / S=%d: more than 4096 tiles or 512 states", Mx, Sx); }
  p->NT = std::max(roundup64((nt + p->TPT - 1) / p->TPT), std::max(roundup64(sh.S), 128));
  {   // filter: one thread per lower-triangular tile
    const int slots = sh.M * (sh.M + 1) / 2;
    p->TPT_f = (slots + 511) / 512;
    if (p->TPT_f > 4 && !ih) { const int Mx = sh.M; nagp_plan_destroy(p); FAIL(NAGP_EUNSUPPORTED, "M=%d too large for the filter kernel", Mx); }
    if (ih) p->TPT_f = std::min(p->TPT_f, 4);
    if (p->TPT_f == 3) p->TPT_f = 4;   // instantiated: 1, 2, 4 tiles per thread
    p->NT_f = std::max(roundup64((slots + p->TPT_f - 1) / p->TPT_f), std::max(roundup64(sh.S), ekf ? 128 : 384));
    // ADF launches: <= 256 threads (512 registers per lane) whenever the tiles fit
    if (slots <= 1024 && sh.S <= 256) { p->TPT_a = slots <= 256 ? 1 : (slots <= 512 ? 2 : (slots <= 768 ? 3 : 4)); p->NT_a = 256; p->LB_a = 256; }
    else { p->TPT_a = 4; p->NT_a = std::max(roundup64((slots + 3) / 4), roundup64(sh.S)); p->LB_a = 512; }
    p->wide_l = (!ekf && slots > 512 && slots <= 1024 && !dev_env("NAGP_NO_WIDE")) ? 1 : 0;
    p->NT_l = p->wide_l ? roundup64(slots) : p->NT_f;
    // fixed-site launches with one tile per thread: whole waves beyond the tile threads for the state lanes (gf_filter_kernel: soff)
    p->NT_fl = p->NT_f;
    if (!ekf && p->TPT_f == 1 && roundup64(slots) + roundup64(sh.S) <= 512) p->NT_fl = std::max(p->NT_f, roundup64(slots) + roundup64(sh.S));
  }
  p->want_PS = (o->flags & 0x4u) != 0;
  p->need_PF = (o->kind != NAGP_KIND_IHGP) && !(o->mode == NAGP_MODE_NLML && (o->ep_itts == 1 || ekf));

  // ---- model packing
  const size_t msz = mdl_size(sh);
  std::vector<double> hm((size_t)B * msz, 0.0);
  p->h_hval.resize((size_t)B * sh.M);
  for (int q = 0; q < B; ++q) {
    const nagp_model& mq = models[q];
    double* d = hm.data() + (size_t)q * msz;
    const int S = sh.S;
    for (int n = 0; n < sh.M; ++n) {
      const int o0 = sh.off[n], bs = sh.bsz[n];
      for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) {
          const size_t src = (size_t)(o0 + i) + (size_t)S * (o0 + j);   // column-major
          d[mdl_A(sh) + (size_t)n * 16 + 4 * i + j] = mq.A[src];
          d[mdl_Q(sh) + (size_t)n * 16 + 4 * i + j] = mq.Q[src];
          d[mdl_P(sh) + (size_t)n * 16 + 4 * i + j] = mq.Pinf[src];
        }
      d[mdl_h(sh) + n] = mq.h_val[n];
      p->h_hval[(size_t)q * sh.M + n] = mq.h_val[n];
    }
    if (mq.Wnmf && (ekf || o->lik_kind != NAGP_LIK_POWER))
      for (int dd = 0; dd < sh.D; ++dd)
        for (int j = 0; j < sh.N; ++j) d[mdl_W(sh) + (size_t)dd * sh.N + j] = mq.Wnmf[dd + (size_t)sh.D * j];
    d[mdl_sn2(sh)] = std::exp(mq.lik_param);
  }
  PLAN_TRY(dalloc(p, &p->d_model, hm.size(), false));
  PLAN_HIP(hipMemcpyAsync(p->d_model, hm.data(), hm.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
  PLAN_HIP(hipStreamSynchronize(p->stream));

  // ---- cubature tables (point-major)
  MomCfg& mc = p->mc;
  if (!ekf) {
    // distinct unit coordinates + per-point byte codes (see nagp_dev.hpp: mom)
    std::vector<double> xd;
    std::vector<unsigned char> code((size_t)o->n_pts * o->cub_dim);
    for (int pt = 0; pt < o->n_pts; ++pt)
      for (int j = 0; j < o->cub_dim; ++j) {
        const double v = o->xn_unscaled[j + (size_t)o->cub_dim * pt];
        size_t ci = 0;
        while (ci < xd.size() && xd[ci] != v) ++ci;
        if (ci == xd.size()) {
          if (xd.size() == 64) { nagp_plan_destroy(p); FAIL(NAGP_EUNSUPPORTED, "sigma-point rule has more than 64 distinct coordinate values"); }
          xd.push_back(v);
        }
        code[(size_t)pt * o->cub_dim + j] = (unsigned char)ci;
      }
    PLAN_TRY(dalloc(p, &p->d_wn, o->n_pts, false));
    PLAN_TRY(dalloc(p, &p->d_xi, xd.size() + (code.size() + 7) / 8 + 1, false));
    PLAN_HIP(hipMemcpyAsync(p->d_wn, o->wn, (size_t)o->n_pts * sizeof(double), hipMemcpyHostToDevice, p->stream));
    PLAN_HIP(hipMemcpyAsync(p->d_xi, xd.data(), xd.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
    PLAN_HIP(hipMemcpyAsync(p->d_xi + xd.size(), code.data(), code.size(), hipMemcpyHostToDevice, p->stream));
    PLAN_HIP(hipStreamSynchronize(p->stream));
    mc.nd = (int)xd.size(); mc.xd = p->d_xi; mc.code = reinterpret_cast<const unsigned char*>(p->d_xi + xd.size());
    mc.lik_kind = o->lik_kind; mc.link_kind = o->link_kind; mc.link_shift = o->link_shift;
    mc.n_pts = o->n_pts; mc.cdim = o->cub_dim; mc.D = sh.D; mc.wn = p->d_wn;
    mc.jitter = (o->lik_kind == NAGP_LIK_POWER) ? 1e-8 : 1e-10;
    mc.DG = 1; mc.cache_tabs = 0; mc.store_a = 0; mc.stamps = nullptr;
    if (o->lik_kind == NAGP_LIK_POWER_NMF && o->cub_dim <= MSP_MAXCD && !dev_env("NAGP_NO_SPARSE")) {
      // sparse-point form: needs the coordinate value 0 and <= MSP_NZ non-centre coordinates per sigma point
      int c0 = -1;
      for (size_t ci = 0; ci < xd.size(); ++ci) if (xd[ci] == 0.0) c0 = (int)ci;
      int nzmax = 0;
      std::vector<int> pdesc((size_t)o->n_pts * MSP_NZ, -1);
      bool okp = c0 >= 0 && (int)xd.size() * o->cub_dim <= MSP_TS - 1;
      for (int pt = 0; okp && pt < o->n_pts; ++pt) {
        int nz = 0;
        for (int j = 0; j < o->cub_dim; ++j) {
          const int cc = code[(size_t)pt * o->cub_dim + j];
          if (cc == c0) continue;
          if (nz == MSP_NZ) { okp = false; break; }
          pdesc[(size_t)pt * MSP_NZ + nz++] = j * (int)xd.size() + cc;
        }
        nzmax = std::max(nzmax, nz);
      }
      if (okp) {
        double* dd = nullptr;
        PLAN_TRY(dalloc(p, &dd, (pdesc.size() + 1) / 2 + 1, false));
        PLAN_HIP(hipMemcpyAsync(dd, pdesc.data(), pdesc.size() * sizeof(int), hipMemcpyHostToDevice, p->stream));
        PLAN_HIP(hipStreamSynchronize(p->stream));
        p->sp.enabled = 1; p->sp.c0 = c0; p->sp.nzmax = nzmax; p->sp.pdesc = reinterpret_cast<const int*>(dd);
        for (int j = 0; j < o->cub_dim; ++j)
          for (int cc = 0; cc < (int)xd.size(); ++cc) {
            if (cc == c0) continue;
            int cnt = 0;
            for (int q = 0; q < o->n_pts; ++q) cnt += (code[(size_t)q * o->cub_dim + j] == cc) ? 1 : 0;
            p->sp_maxmem = std::max(p->sp_maxmem, cnt);
          }
      }
    }
    if (o->lik_kind == NAGP_LIK_POWER_NMF_SQRT && o->cub_dim <= MSQ_MAXCD && sh.D <= MSQ_MAXD && !dev_env("NAGP_NO_SPARSE")) {
      // staged form of the square-root amplitudes: needs the coordinate value 0 (the marginal sums leave the centre to a difference),
      // the marginal lists of the packed form (<= 16 per marginal wave, <= 64 members each) and <= 320 sigma points
      int c0 = -1;
      const int ndp = (int)xd.size(), CDp = o->cub_dim;
      for (size_t ci = 0; ci < xd.size(); ++ci) if (xd[ci] == 0.0) c0 = (int)ci;
      int maxmem = 0;
      for (int j = 0; j < CDp; ++j)
        for (int cc = 0; cc < ndp; ++cc) {
          if (cc == c0) continue;
          int cnt = 0;
          for (int q = 0; q < o->n_pts; ++q) cnt += (code[(size_t)q * CDp + j] == cc) ? 1 : 0;
          maxmem = std::max(maxmem, cnt);
        }
      if (c0 >= 0 && ndp >= 2 && ndp * CDp <= MSP_TS - 1 && (ndp - 1) * ((CDp + 1) / 2) <= 16 && (ndp - 1) * CDp <= MSR_NMARG &&
          maxmem <= 4 * MSR_NMEM && o->n_pts <= MSQ_MAXPTS) { p->sq_ok = 1; p->sq_c0 = c0; }
    }
    if (o->lik_kind != NAGP_LIK_POWER) {
      std::vector<unsigned char> blob;
      MomSrc sc;
      if (build_mom_src(B, models, sh.D, sh.N, o->n_pts, code, sc, blob)) {
        double* dsrc = nullptr;
        PLAN_TRY(dalloc(p, &dsrc, (blob.size() + 7) / 8, false));
        PLAN_HIP(hipMemcpyAsync(dsrc, blob.data(), blob.size(), hipMemcpyHostToDevice, p->stream));
        PLAN_HIP(hipStreamSynchronize(p->stream));
        sc.blob = reinterpret_cast<const unsigned char*>(dsrc);
        p->src_all = sc;
      }
    }
  }

  // ---- buffers
  const size_t BT = (size_t)B * T;
  Bufs& b = p->b;
  b.model = p->d_model;
  PLAN_TRY(dalloc(p, &p->d_y, BT)); b.y = p->d_y;
  PLAN_TRY(dalloc(p, &b.ttau, BT * sh.M));
  PLAN_TRY(dalloc(p, &b.tnu, BT * sh.M));
  PLAN_TRY(dalloc(p, &b.R, BT * sh.M));
  PLAN_TRY(dalloc(p, &b.lZ, BT));
  PLAN_TRY(dalloc(p, &b.MF, BT * sh.S));
  PLAN_TRY(dalloc(p, &b.MS, BT * sh.S));
  PLAN_TRY(dalloc(p, &b.fm, BT * sh.M));
  PLAN_TRY(dalloc(p, &b.fv, BT * sh.M));
  PLAN_TRY(dalloc(p, &b.sm, BT * sh.M));
  PLAN_TRY(dalloc(p, &b.sv, BT * sh.M));
  PLAN_TRY(dalloc(p, &b.state, (size_t)B * ((size_t)nt * 16 + sh.S)));
  PLAN_TRY(dalloc(p, &b.red, (size_t)B * 8));
  p->red0 = b.red;
  if (p->opts.kind == NAGP_KIND_GF_EP) PLAN_TRY(dalloc(p, &p->red_all, (size_t)(p->opts.ep_itts + 2) * B * 8));
  { double* c = nullptr; PLAN_TRY(dalloc(p, &c, (size_t)B * 4)); b.counters = reinterpret_cast<unsigned long long*>(c); }
  // smoother chunk: the unit of the (G, Delta) buffers and of the filter -> smoother pipeline.  About a dozen chunks per sweep
  // (the tail the pipeline cannot hide is one chunk's gain + compose), at least 2048 steps each, at most what one buffer may take.
  {
    const int64_t n_ch = dev_env("NAGP_CHUNKS") ? std::max(1, atoi(dev_env("NAGP_CHUNKS"))) : 12;     // developer switch
    p->chunk = (o->chunk > 0) ? o->chunk : (int)std::min<int64_t>(T, std::max<int64_t>(2048, (T + n_ch - 1) / n_ch));
  }
  if (p->chunk > T) p->chunk = (int)T;
  PLAN_TRY(dalloc(p, &p->d_stamps, 24));
  PLAN_TRY(dalloc(p, &p->d_gstamps, 32));
  if (o->kind != NAGP_KIND_IHGP) {
    {   // FP64 MFMA smoother for padded dimensions up to 96 (set NAGP_NO_MFMA=1 to force the VALU passes)
      const int Sp = ((4 * sh.M + 15) / 16) * 16;
      if (Sp <= 96 && !dev_env("NAGP_NO_MFMA")) p->mfma_sp = Sp;
      // 96 < Sp <= 160: state and G no longer fit LDS side by side; column-owner kernels
      // (a sweep that stores the smoothed covariances runs the VALU passes instead: see run_smoother)
      else if (Sp <= 160 && !dev_env("NAGP_NO_MFMA") && !dev_env("NAGP_NO_MFMA_BIG")) { p->mfma_sp = Sp; p->big_sp = 1; }
    }
    const size_t mat = p->mfma_sp ? (size_t)p->mfma_sp * p->mfma_sp : (size_t)nt * 16;     // (4M)^2 <= Sp^2: the tile-major form fits the dense slot
    // column-owner passes (96 < Sp <= 160) read the symmetric Delta through its lower 16x16 tiles only: the slots hold it packed
    // (Sp = 160: 315 KB per step instead of 410 -- eight chunks of the 8-segment cfg5 plan keep their slot where six did).  Not when a
    // sweep stores smoothed covariances (its VALU passes use the tile-major layout of the same slots) or with the opt-in MFMA gain kernel.
    p->dpacked = (p->big_sp && !p->want_PS && !dev_env("NAGP_DENSE_DELTA")) ? 1 : 0;
    p->gstep = p->mfma_sp ? gd_step_doubles(p->mfma_sp, p->dpacked) : 2 * mat;
    const double per_step = (double)B * ((double)p->gstep + sh.S) * 8.0;                   // one step of a (G, Delta, delta) chunk buffer
    {
      // one chunk buffer: at most 24 GiB and at most a quarter of the device memory that is free once the per-step arrays
      // (filtered covariances, means, sites) of this plan are counted
      double cap_bytes = 24.0 * 1073741824.0;
      size_t free_b = 0, total_b = 0;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const double fixed = (double)BT * ((p->need_PF ? (double)pf_step_doubles(sh) : 0.0) + (p->want_PS ? nt * 16.0 : 0.0) + 2.0 * sh.S) * 8.0;
        cap_bytes = std::min(cap_bytes, std::max(0.25 * ((double)free_b - fixed), 64.0 * per_step));
      }
      while (p->chunk > 64 && per_step * p->chunk > cap_bytes) p->chunk = (p->chunk + 1) / 2;
    }
    p->nc = (int)std::max<int64_t>(1, (T - 1 + p->chunk - 1) / p->chunk);
    if (p->nc >= 2) p->nc += 1;      // the chunk of the latest steps is cut short (chunk0_len): one chunk more
    if (p->need_PF) PLAN_TRY(dalloc(p, &b.PF, BT * pf_step_doubles(sh), false));   // lower-triangular tiles only (layout: pf_off)
    if (p->want_PS) PLAN_TRY(dalloc(p, &b.PSs, BT * nt * 16, false));
    // panel widths: one tile per thread per operand panel, panels (double buffered) within 72 KiB of LDS
    const double cap = 72.0 * 1024.0;
    p->LP1 = std::max(1, std::min(std::min(sh.M, p->NT / (3 * sh.M)), (int)(cap / (2.0 * 3 * sh.M * TS * 8))));
    p->LP2 = std::max(1, std::min(std::min(sh.M, p->NT / (2 * sh.M)), (int)(cap / (2.0 * 2 * sh.M * TS * 8))));
    // spans per chunk: pass 2 is sequential in the span count, passes 1+3 in the span length
    {
      const int per_prob = std::max(1, 1024 / std::min(B, 1024));
      int ns = (int)std::lround(std::sqrt(2.5 * (double)p->chunk));
      ns = std::max(1, std::min(std::min(ns, per_prob), (p->chunk + 7) / 8));
      p->ns_max = ns;
    }
    const bool need_valu = !p->mfma_sp || (p->big_sp && p->want_PS);   // the column-owner kernels have no smoothed-covariance output
    const size_t SS = (size_t)p->mfma_sp * p->mfma_sp;
    // boundary values / scratch of the span passes: one set (boundary and apply of a chunk run back to back on the main stream)
    if (p->mfma_sp) {
      PLAN_TRY(dalloc(p, &p->mpar.stateD, (size_t)B * (SS + sh.S), true));
      p->lds_mfma = (p->big_sp ? big_lds_doubles(p->mfma_sp / 16) : mfma_lds_doubles(p->mfma_sp)) * sizeof(double);
    }
    // Chunk-pipelined schedule: needs >= 2 chunks and >= 2 chunk buffers.  The compose results (Phi, C, c of every span) are kept
    // per chunk; the (G, Delta, delta) buffers are kept for as many chunks as the free memory holds, the rest recompute their
    // gains after the filter (slot 0 is the scratch).
    p->pipeline = p->need_PF && p->nc >= 2 && !dev_env("NAGP_NO_PIPELINE");
    const int n_sets = p->pipeline ? p->nc : 1;
    for (int c = 0; c < n_sets; ++c) {
      double *a1 = nullptr, *a2 = nullptr, *a3 = nullptr, *a4 = nullptr, *a5 = nullptr;
      if (need_valu) {
        PLAN_TRY(dalloc(p, &a1, (size_t)B * p->ns_max * 2 * nt * 16, false));
        PLAN_TRY(dalloc(p, &a2, (size_t)B * p->ns_max * sh.S, false));
        PLAN_TRY(dalloc(p, &a5, (size_t)B * p->ns_max * nt * 16, false));
      }
      p->c_xbuf.push_back(a5);
      {
        double *b1 = nullptr, *b2 = nullptr;
        if (need_valu) PLAN_TRY(dalloc(p, &b1, (size_t)B * p->ns_max * ((size_t)nt * 16 + sh.S), false));
        if (p->mfma_sp) PLAN_TRY(dalloc(p, &b2, (size_t)B * p->ns_max * (SS + sh.S), false));
        p->c_bnd.push_back(b1); p->c_mbnd.push_back(b2);
      }
      if (p->mfma_sp) {
        PLAN_TRY(dalloc(p, &a3, (size_t)B * p->ns_max * 2 * SS, false));
        PLAN_TRY(dalloc(p, &a4, (size_t)B * p->ns_max * sh.S, false));
      }
      p->c_spanbuf.push_back(a1); p->c_spanvec.push_back(a2); p->c_mspanbuf.push_back(a3); p->c_mspanvec.push_back(a4);
    }
    int n_slots = 1;
    if (p->pipeline) {
      size_t free_b = 0, total_b = 0;
      const double slot_bytes = per_step * p->chunk;
      if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
        const double avail = (double)free_b - 2.0 * 1073741824.0 - 0.03 * (double)total_b;     // head-room for the runtime, RCCL, other plans
        n_slots = (int)std::max(1.0, std::min((double)(p->nc - 1), std::floor((avail - slot_bytes / 8.0) / slot_bytes)));   // full chunks: nc - 1
      }
      if (const char* e = dev_env("NAGP_PIPELINE_SLOTS")) n_slots = std::max(1, std::min(p->nc - 1, atoi(e)));   // developer switch (tests: partial retention)
      if (n_slots < 2) { p->pipeline = false; n_slots = 1; }
    }
    p->mat_doubles = mat;
    auto add_slot = [&](int cap_steps) -> int {
      if (const char* e = dev_env("NAGP_TEST_SLOT_ENOMEM"))          // test hook: the (n+1)-th slot allocation of a plan fails
        if ((int)p->slotG.size() >= atoi(e)) return NAGP_ENOMEM;
      double *g = nullptr, *d = nullptr;
      int st = dalloc(p, &g, (size_t)B * cap_steps * p->gstep, true);
      if (st == NAGP_OK) st = dalloc(p, &d, (size_t)B * cap_steps * sh.S, false);
      if (st == NAGP_OK) { p->slotG.push_back(g); p->slotD.push_back(d); p->slot_tiled.push_back(0); p->slot_cap.push_back(cap_steps); p->slot_gps.push_back(0); }
      else if (g) dfree(p, g);
      return st;
    };
    // The slots beyond the first are an optimisation sized from ONE hipMemGetInfo snapshot: fragmentation, a second plan or another
    // process may have taken the memory since.  Best effort -- a slot that cannot be had is done without (fewer retained chunks, or the
    // serial schedule with the one scratch slot); only slot 0 is indispensable.
    auto drop_last_slot = [&]() {
      dfree(p, p->slotG.back()); dfree(p, p->slotD.back());
      p->slotG.pop_back(); p->slotD.pop_back(); p->slot_tiled.pop_back(); p->slot_cap.pop_back(); p->slot_gps.pop_back();
    };
    PLAN_TRY(add_slot(p->chunk));
    {
      int got = 1;
      for (; got < n_slots; ++got) {
        const int st = add_slot(p->chunk);
        if (st == NAGP_ENOMEM) { g_last_error.clear(); break; }
        PLAN_TRY(st);
      }
      n_slots = got;
    }
    if (p->pipeline) {   // the short chunk of the latest steps has its own small slot
      const int small = std::min(p->chunk, std::max(64, p->chunk / 8));
      int st = (n_slots >= 2) ? add_slot(small) : NAGP_ENOMEM;
      while (st == NAGP_ENOMEM && n_slots > 2) { drop_last_slot(); --n_slots; st = add_slot(small); }
      if (st == NAGP_ENOMEM) {     // fewer than two full slots beside the small one: serial schedule, slot 0 only
        while (n_slots > 1) { drop_last_slot(); --n_slots; }
        p->pipeline = false; g_last_error.clear();
      } else PLAN_TRY(st);
    }
    p->n_full_slots = n_slots;
    // Recycled slots.  The column-owner passes read PF_k for k = 0 only (the restart state), the gain kernel of a chunk reads the PF of
    // its own steps and of the step behind them, and the gains of the chunks are enqueued in time order on one stream: once the gains
    // of the earliest chunks exist, their part of PF is free until the next sweep's filter.  When the free memory does not hold a slot
    // per chunk, the chunks the filter finishes LAST take theirs from there -- recycled slot j (the chunk with n_slots + j full chunks
    // before it in time) occupies doubles [pf_step + j * chunk * gstep, pf_step + (j+1) * chunk * gstep) of every problem's PF; all of
    // it must lie below the first step of that chunk.  All-or-nothing: a chunk left without a slot would read PF again.
    if (p->pipeline && p->dpacked && n_slots < p->nc - 1 && !dev_env("NAGP_NO_RECYCLE")) {
      std::vector<int64_t> k0s;      // first step of the chunks, latest first (the cuts of sweep_begin)
      for (int64_t k1 = T - 1; k1 > 0;) { const int nk = chunk_len(p, k1, k0s.empty()); k0s.push_back(k1 - nk); k1 -= nk; }
      const int ncs = (int)k0s.size(), need = (ncs - 1) - n_slots;
      const size_t pfs = pf_step_doubles(sh);
      bool ok = need > 0;
      for (int j = 0; ok && j < need; ++j) {
        const size_t end = pfs + (size_t)(j + 1) * p->chunk * p->gstep;
        const int c = ncs - 1 - (n_slots + j);               // time-chunk n_slots + j
        ok = (end + pfs - 1) / pfs <= (size_t)k0s[c];
      }
      if (ok) {
        std::vector<double*> ds;                         // the delta vectors of the recycled slots are memory of their own: all or none
        for (int j = 0; ok && j < need; ++j) {
          double* d = nullptr;
          const int st = dalloc(p, &d, (size_t)B * p->chunk * sh.S, false);
          if (st == NAGP_ENOMEM) { ok = false; break; }
          PLAN_TRY(st);
          ds.push_back(d);
        }
        if (!ok) { for (double* d : ds) dfree(p, d); g_last_error.clear(); }
        else {
          for (int j = 0; j < need; ++j) {
            p->slotG.push_back(b.PF + pfs + (size_t)j * p->chunk * p->gstep); p->slotD.push_back(ds[j]);
            p->slot_tiled.push_back(0); p->slot_cap.push_back(p->chunk); p->slot_gps.push_back((size_t)T * pfs);
          }
          p->n_recycled = need;
        }
      }
    }
    p->gbuf_doubles = (size_t)B * p->chunk * p->gstep;
    if (p->pipeline) {
      PLAN_HIP(hipStreamCreateWithFlags(&p->stream2, hipStreamNonBlocking));
      { hipStream_t st = nullptr; PLAN_HIP(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); p->s_apply.push_back(st); }
      {
        hipEvent_t e1 = nullptr, e2 = nullptr;
        PLAN_HIP(hipEventCreateWithFlags(&e1, hipEventDisableTiming)); p->ev_bnd.push_back(e1);
        PLAN_HIP(hipEventCreateWithFlags(&e2, hipEventDisableTiming)); p->ev_app.push_back(e2);
      }
      PLAN_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->h_tab), (size_t)(p->nc + 1) * sizeof(ChunkTab), hipHostMallocMapped | hipHostMallocCoherent));
      PLAN_HIP(hipEventCreateWithFlags(&p->ev_filter, hipEventDisableTiming));
      PLAN_HIP(hipEventCreateWithFlags(&p->ev_s2, hipEventDisableTiming));
      if (p->opts.kind == NAGP_KIND_GF_EP && !dev_env("NAGP_NO_XSWEEP")) {
        for (int c = 0; c < p->nc + 1; ++c) { hipEvent_t e = nullptr; PLAN_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->ev_chunk.push_back(e); }
        PLAN_HIP(hipEventCreateWithFlags(&p->ev_red, hipEventDisableTiming));
        p->xsweep = true;
      }
      PLAN_HIP(hipHostMalloc(reinterpret_cast<void**>(&p->h_progress), (size_t)B * sizeof(unsigned long long), hipHostMallocMapped | hipHostMallocCoherent));
      std::memset(p->h_progress, 0, (size_t)B * sizeof(unsigned long long));
    }
    if (dev_env("NAGP_STAMPS")) fprintf(stderr, "[nagp plan] smoother: chunk %d, %d chunk(s) per sweep, %d (G,Delta) buffer(s) of %.2f GiB (+ %d recycled from PF), pipelined %d\n", p->chunk, p->nc, n_slots, per_step * p->chunk / 1073741824.0, p->n_recycled, (int)p->pipeline);
  } else {
    PLAN_TRY(dalloc(p, &p->d_lZs, BT));
    PLAN_TRY(dalloc(p, &p->d_vprev, (size_t)B * sh.M));
    p->aff_L = 128; p->aff_ns = (int)((T + p->aff_L - 1) / p->aff_L);
    PLAN_TRY(dalloc(p, &p->d_affspan, (size_t)B * p->aff_ns * sh.M * 20, false));
    PLAN_TRY(dalloc(p, &p->d_affbnd, (size_t)B * p->aff_ns * sh.M * 4, false));
    // ---- IHGP tables: MATLAB layout -> device layout (see nagp_ihgp.hpp)
    const int NG = tables[0].n_grid;
    if (NG < 2) { nagp_plan_destroy(p); FAIL(NAGP_EINVAL, "n_grid < 2"); }
    const size_t tsz = itab_size(sh, NG);
    std::vector<double> ht((size_t)B * tsz, 0.0);
    for (int q = 0; q < B; ++q) {
      const nagp_ihgp_tables& tq = tables[q];
      if (tq.n_grid != NG) { nagp_plan_destroy(p); FAIL(NAGP_EINVAL, "n_grid differs between problems"); }
      double* d = ht.data() + (size_t)q * tsz;
      const nagp_model& mq = models[q];
      for (int n = 0; n < sh.M; ++n) {
        const int bs = sh.bsz[n], o0 = sh.off[n];
        const double h = mq.h_val[n];
        const double* pp = tq.PPlist + tq.pp_offsets[n];
        const double* pg = tq.PGlist + tq.pg_offsets[n];
        for (int g = 0; g < NG; ++g) {
          const double* ppr = pp + (size_t)g * bs * bs;          // column-major bs x bs
          d[itab_hph(sh, NG) + (size_t)n * NG + g] = h * h * ppr[0];
          for (int i = 0; i < bs; ++i) d[itab_wcol(sh, NG) + ((size_t)n * NG + g) * 4 + i] = h * ppr[i];
          const double* pgr = pg + (size_t)g * 2 * bs * bs;      // [PS2(:)' G(:)']
          d[itab_v(sh, NG) + (size_t)n * NG + g] = h * h * pgr[0];
          for (int i = 0; i < bs; ++i)
            for (int j = 0; j < bs; ++j)
              d[itab_g(sh, NG) + ((size_t)n * NG + g) * 16 + 4 * i + j] = pgr[bs * bs + i + bs * j];
        }
        d[itab_hph0(sh, NG) + n] = h * h * mq.Pinf[(size_t)o0 + (size_t)sh.S * o0];
        for (int i = 0; i < bs; ++i) d[itab_wcol0(sh, NG) + (size_t)n * 4 + i] = h * mq.Pinf[(size_t)(o0 + i) + (size_t)sh.S * o0];
      }
    }
    PLAN_TRY(dalloc(p, &p->d_tab, ht.size(), false));
    PLAN_TRY(dalloc(p, &p->d_r, NG, false));
    PLAN_HIP(hipMemcpyAsync(p->d_tab, ht.data(), ht.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
    PLAN_HIP(hipMemcpyAsync(p->d_r, tables[0].r_grid, (size_t)NG * sizeof(double), hipMemcpyHostToDevice, p->stream));
    PLAN_HIP(hipStreamSynchronize(p->stream));
    p->tb.NG = NG; p->tb.r = p->d_r; p->tb.base = p->d_tab;
    p->tb.lr0 = std::log10(tables[0].r_grid[0]);
    p->tb.inv_dlr = (double)(NG - 1) / (std::log10(tables[0].r_grid[NG - 1]) - p->tb.lr0);
  }

  // ---- LDS sizes / kernel attributes
  if (o->kind == NAGP_KIND_IHGP) {
    // one wave per SIMD: 512 registers per lane (the cubature's pressure stays out of scratch memory); the N = 9 instantiation
    // (thousands of sigma points per step, W rows in registers) runs two waves per SIMD
    p->NT_ih = (mom_variant(mc) >= 9) ? 512 : 256;
    p->DG_f = pick_DG(o->lik_kind, o->n_pts, p->NT_ih, sh.D, o->cub_dim);
    MomCfg t = mc; t.DG = p->DG_f; t.cache_tabs = 1; t.store_a = (o->lik_kind == NAGP_LIK_POWER_NMF_SQRT) ? 1 : 0;
    p->kb_ih = IH_KB;
    if (p->src_all.n_src >= 2) {   // block-structured Wnmf: the tuple tables must be resident (a shorter I/O ring makes room)
      t.src = p->src_all;
      while (p->kb_ih > 4 && ihgp_filter_lds_doubles(sh, t, p->tb.NG, 0, p->kb_ih) * sizeof(double) > 156 * 1024) p->kb_ih /= 2;
      if (ihgp_filter_lds_doubles(sh, t, p->tb.NG, 0, p->kb_ih) * sizeof(double) <= 156 * 1024) p->src_f = 1;
      else { t.src = MomSrc{}; p->kb_ih = IH_KB; }
    }
    p->hph_lds = 1;   // LDS budget, least valuable resident first: H PP H' table, a[d][p], cubature tables
    if (ihgp_filter_lds_doubles(sh, t, p->tb.NG, p->hph_lds, p->kb_ih) * sizeof(double) > 156 * 1024) p->hph_lds = 0;
    if (ihgp_filter_lds_doubles(sh, t, p->tb.NG, p->hph_lds, p->kb_ih) * sizeof(double) > 156 * 1024) t.store_a = 0;
    if (ihgp_filter_lds_doubles(sh, t, p->tb.NG, p->hph_lds, p->kb_ih) * sizeof(double) > 156 * 1024) t.cache_tabs = 0;
    p->cache_f = t.cache_tabs; p->sta_f = t.store_a;
    p->lds_ih = ihgp_filter_lds_doubles(sh, t, p->tb.NG, p->hph_lds, p->kb_ih) * sizeof(double);
    if (dev_env("NAGP_STAMPS")) fprintf(stderr, "[nagp plan] ihgp filter: LDS %zu B, hph table in LDS %d, cubature tables in LDS %d, block-structured mom %d, mom LDS %zu B\n", p->lds_ih, p->hph_lds, p->cache_f, p->src_f, mom_lds_doubles(t) * sizeof(double));
    // the ADF sweep in the sparse-point form (ihgp_adf_kernel): plain NMF likelihood, <= 320 sigma points, unstructured Wnmf
    if (p->sp.enabled && !p->src_f && sh.M <= 64 && sh.D <= 4 * MSP_DT && o->n_pts <= MSP_NT + 64 && (o->n_pts + 3) / 4 <= MSP_NW * MSP_NST) {
      p->kb_sp = IH_KB; p->hph_sp = 1;
      if (const char* e = dev_env("NAGP_IH_KB")) p->kb_sp = std::max(1, std::min(IH_KB, atoi(e)));   // developer switch: steps per I/O block
      auto need = [&]() { return ihgp_adf_lds_doubles(sh, o->cub_dim, p->tb.NG, p->hph_sp, p->kb_sp) * sizeof(double) + 16; };
      if (need() > 156 * 1024) p->kb_sp = 8;
      if (need() > 156 * 1024) p->hph_sp = 0;
      if (need() <= 156 * 1024) {
        p->sp_ih = 1; p->lds_sp = need();
        switch (o->cub_dim) {
          case 1: PLAN_TRY(set_lds(ihgp_adf_kernel<1>, p->lds_sp)); break; case 2: PLAN_TRY(set_lds(ihgp_adf_kernel<2>, p->lds_sp)); break;
          case 3: PLAN_TRY(set_lds(ihgp_adf_kernel<3>, p->lds_sp)); break; case 4: PLAN_TRY(set_lds(ihgp_adf_kernel<4>, p->lds_sp)); break;
          case 5: PLAN_TRY(set_lds(ihgp_adf_kernel<5>, p->lds_sp)); break; case 6: PLAN_TRY(set_lds(ihgp_adf_kernel<6>, p->lds_sp)); break;
          default: PLAN_TRY(set_lds(ihgp_adf_kernel<7>, p->lds_sp)); break;
        }
        // role-specialised waves: two serial + six worker waves, one sigma point per worker lane, <= 80 MFMA steps
        const size_t need8 = ihgp_adf8_lds_doubles(sh, o->cub_dim, p->tb.NG, p->hph_sp, p->kb_sp) * sizeof(double) + 16;
        const char* er = dev_env("NAGP_IH_ROLES");
        if (o->n_pts <= 64 * MSR_NWK && (o->n_pts + 3) / 4 <= 4 * MSR_NST && need8 <= 156 * 1024 && !(er && er[0] == '0')) {
          p->sp_ih8 = 1; p->lds_sp8 = need8;
          // packed form (eight points per MFMA step, g1 / g2 from marginal sums): <= 6 components, <= 16 marginals per marginal wave, each of <= 64 members
          {
            const int CDp = o->cub_dim, ndp = mc.nd;
            bool pk = CDp <= 6 && (ndp - 1) * ((CDp + 1) / 2) <= 16 && (ndp - 1) * CDp <= MSR_NMARG && (o->n_pts + 7) / 8 <= 40 && p->sp_maxmem <= 4 * MSR_NMEM;   // 3 of 8 slots <= MSR_NSTP steps
            const char* ep = dev_env("NAGP_IH_PACK");
            if (ep && ep[0] == '0') pk = false;
            p->sp_pack = pk ? 1 : 0;
          }
#define SL8(V, PK) PLAN_TRY(set_lds(ihgp_adf8_kernel<V, PK>, need8))
          if (p->sp_pack) switch (o->cub_dim) { case 1: SL8(1, true); break; case 2: SL8(2, true); break; case 3: SL8(3, true); break; case 4: SL8(4, true); break; case 5: SL8(5, true); break; default: SL8(6, true); break; }
          else switch (o->cub_dim) { case 1: SL8(1, false); break; case 2: SL8(2, false); break; case 3: SL8(3, false); break; case 4: SL8(4, false); break; case 5: SL8(5, false); break; case 6: SL8(6, false); break; default: SL8(7, false); break; }
#undef SL8
        }
      }
    }
    // likModulatorPreCalcwn: the role-specialised sweep of nagp_momsq.hpp
    if (p->sq_ok && !p->src_f && sh.M <= 64) {
      p->kb_sq = IH_KB; p->hph_sq = 1;
      if (const char* e = dev_env("NAGP_IH_KB")) p->kb_sq = std::max(1, std::min(IH_KB, atoi(e)));
      auto needq = [&]() { return ihgp_adf8sq_lds_doubles(sh, o->cub_dim, p->tb.NG, p->hph_sq, p->kb_sq) * sizeof(double) + 16; };
      if (needq() > 156 * 1024) p->kb_sq = 8;
      if (needq() > 156 * 1024) p->hph_sq = 0;
      if (needq() <= 156 * 1024) {
        p->sq_ih = 1; p->lds_sq = needq();
        switch (o->cub_dim) {
          case 1: PLAN_TRY(set_lds(ihgp_adf8sq_kernel<1>, p->lds_sq)); break; case 2: PLAN_TRY(set_lds(ihgp_adf8sq_kernel<2>, p->lds_sq)); break;
          case 3: PLAN_TRY(set_lds(ihgp_adf8sq_kernel<3>, p->lds_sq)); break; case 4: PLAN_TRY(set_lds(ihgp_adf8sq_kernel<4>, p->lds_sq)); break;
          case 5: PLAN_TRY(set_lds(ihgp_adf8sq_kernel<5>, p->lds_sq)); break; default: PLAN_TRY(set_lds(ihgp_adf8sq_kernel<6>, p->lds_sq)); break;
        }
      }
    }
    if (dev_env("NAGP_STAMPS")) fprintf(stderr, "[nagp plan] ihgp ADF sweep, square-root amplitudes in the staged form: %d (LDS %zu B, ring %d steps, hph table in LDS %d)\n", p->sq_ih, p->lds_sq, p->kb_sq, p->hph_sq);
    if (dev_env("NAGP_STAMPS")) fprintf(stderr, "[nagp plan] ihgp ADF sweep in the sparse-point form: %d (LDS %zu B, ring %d steps, hph table in LDS %d), role-specialised waves %d (LDS %zu B), packed MFMA steps %d\n", p->sp_ih, p->lds_sp, p->kb_sp, p->hph_sp, p->sp_ih8, p->lds_sp8, p->sp_pack);
#define SL(V) PLAN_TRY(set_lds(ihgp_filter_kernel<V, false>, p->lds_ih))
#define SLS(V) PLAN_TRY(set_lds(ihgp_filter_kernel<V, true>, p->lds_ih))
    if (p->src_f) { NAGP_MV_SWITCH9(mom_variant(mc), SLS) } else { NAGP_MV_SWITCH9(mom_variant(mc), SL) }
#undef SL
#undef SLS
  } else {
    if (!ekf) p->DG_f = pick_DG(o->lik_kind, o->n_pts, p->NT_a, sh.D, o->cub_dim);
    MomCfg t = mc; t.DG = p->DG_f; t.cache_tabs = ekf ? 0 : 1; t.store_a = (!ekf && o->lik_kind == NAGP_LIK_POWER_NMF_SQRT) ? 1 : 0;
    // ADF launches in the sparse-point form (256-thread launches, <= 320 sigma points)
    if (!ekf && p->sp.enabled && p->LB_a == 256 && p->NT_a == MSP_NT && sh.M <= 64 && sh.D <= 4 * MSP_DT && o->n_pts <= MSP_NT + 64 &&
        (o->n_pts + 3) / 4 <= MSP_NW * MSP_NST) {
      p->sp_gf = 1; t.sp = p->sp;
    }
    // ... or with likModulatorPreCalcwn in the staged form of nagp_momsq.hpp
    if (!ekf && p->sq_ok && !p->src_all.n_src && p->LB_a == 256 && p->NT_a == 256 && sh.M <= 64) { p->sq_gf = 1; t.sq_form = 1; t.store_a = 0; }
    const size_t cap = 156 * 1024;
    p->kb_f = 16;
    if (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double) > cap) p->kb_f = 8;
    if (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double) > cap) p->kb_f = 4;
    if (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double) > cap) t.store_a = 0;
    if (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double) > cap) t.cache_tabs = 0;
    if (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double) > cap) p->kb_f = 2;      // 59 .. 63 sites: the W panel alone is 110 - 127 KB
    if (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double) > cap) p->kb_f = 1;
    if (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double) > cap) t.chunk_cap = 256;   // ut7 / ut9 in nine dimensions at 57 sites: 256 points per pass
    if (const char* e = dev_env("NAGP_MOM_CHUNK")) t.chunk_cap = std::max(64, atoi(e));      // developer switch
    p->chunk_cap_f = t.chunk_cap;
    if (const char* e = dev_env("NAGP_KB_F")) p->kb_f = std::max(1, std::min(16, atoi(e)));      // developer switches: the fall-backs of LDS-tight shapes
    if (dev_env("NAGP_NO_CACHE_TABS")) { t.cache_tabs = 0; t.store_a = 0; }
    p->cache_f = t.cache_tabs; p->sta_f = t.store_a;
    p->lds_filter = filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) * sizeof(double);
    // pipelined plans: the filter's workgroup asks for the whole LDS of its CU, so that no workgroup of the smoother kernels running
    // beside it on the second stream is placed on the same CU (the filter is the critical path; its time is latency, not occupancy)
    if (p->pipeline && B <= 128 && p->lds_filter < 160 * 1024) p->lds_filter = 160 * 1024;     // (never BELOW what the kernel needs: set_lds refuses > 160 KiB)
    if (dev_env("NAGP_STAMPS")) fprintf(stderr, "[nagp plan] gf filter: LDS %zu B, ring %d steps, cubature tables in LDS %d, mom LDS %zu B, sparse-point ADF %d\n", p->lds_filter, p->kb_f, p->cache_f, ekf ? (size_t)0 : mom_lds_doubles(t) * sizeof(double), p->sp_gf);
    // ADF sweep with role-specialised waves (nagp_gfadf8.hpp): 512 threads, <= 2 lower tiles per thread, the role layout's limits
    // (one sigma point per worker lane, <= 80 MFMA steps; packed form as in the IHGP sweep)
    if (p->sp_gf && sh.M * (sh.M + 1) / 2 <= 2 * MSR_NT && sh.S <= MSR_NT && o->n_pts <= 64 * MSR_NWK && (o->n_pts + 3) / 4 <= 4 * MSR_NST &&
        !dev_env("NAGP_NO_GF_ROLES")) {
      // tiles per thread / who owns them: 1 or 2 on the six worker waves (<= 384 / 768 lower tiles), else 2 on all eight waves
      const int nlow8 = sh.M * (sh.M + 1) / 2, ntw = MSR_NT - 64 * MSR_W0;
      p->a8_tpt = (nlow8 <= ntw) ? 1 : 2; p->a8_st = (nlow8 <= 2 * ntw) ? 0 : 1;
      if (dev_env("NAGP_A8_ST")) { p->a8_tpt = 2; p->a8_st = 1; }       // developer switch: tiles on all eight waves
      p->kb_a8 = 16;
      while (p->kb_a8 > 2 && gf_adf8_lds_doubles(sh, o->cub_dim, p->kb_a8) * sizeof(double) > cap) p->kb_a8 /= 2;
      if (const char* e = dev_env("NAGP_KB_A8")) p->kb_a8 = std::max(2, std::min(16, atoi(e) & ~1));
      const size_t need = gf_adf8_lds_doubles(sh, o->cub_dim, p->kb_a8) * sizeof(double);
      if (need <= cap) {
        const int CDp = o->cub_dim, ndp = mc.nd;
        bool pk = CDp <= 6 && (ndp - 1) * ((CDp + 1) / 2) <= 16 && (ndp - 1) * CDp <= MSR_NMARG && (o->n_pts + 7) / 8 <= 40 && p->sp_maxmem <= 4 * MSR_NMEM;
        const char* ep = dev_env("NAGP_IH_PACK");
        if (ep && ep[0] == '0') pk = false;
        p->a8_gf = 1; p->a8_pack = pk ? 1 : 0; p->lds_a8 = need;
        if (p->pipeline && B <= 128) p->lds_a8 = 160 * 1024;      // (the whole LDS of the CU, as for the other filter launches below)
#define SA8(TP, V, PK) do { if (p->a8_st) PLAN_TRY(set_lds((gf_adf8_kernel<2, V, PK, true>), p->lds_a8)); else PLAN_TRY(set_lds((gf_adf8_kernel<TP, V, PK, false>), p->lds_a8)); } while (0)
#define SA8V(TP, PK) switch (o->cub_dim) { case 1: SA8(TP, 1, PK); break; case 2: SA8(TP, 2, PK); break; case 3: SA8(TP, 3, PK); break; \
          case 4: SA8(TP, 4, PK); break; case 5: SA8(TP, 5, PK); break; default: SA8(TP, 6, PK); break; }
        if (!pk && o->cub_dim == 7) { if (p->a8_tpt == 1) SA8(1, 7, false); else SA8(2, 7, false); }
        else if (pk) { if (p->a8_tpt == 1) SA8V(1, true) else SA8V(2, true) }
        else { if (p->a8_tpt == 1) SA8V(1, false) else SA8V(2, false) }
#undef SA8V
#undef SA8
      }
    }
    if (dev_env("NAGP_STAMPS")) fprintf(stderr, "[nagp plan] gf ADF sweep with role-specialised waves: %d (tiles per thread %d, ring %d steps, LDS %zu B, packed MFMA steps %d)\n", p->a8_gf, p->a8_tpt, p->kb_a8, p->lds_a8, p->a8_pack);
    p->lds_gain = ((p->TPT == 1) ? gain_lds_doubles_staged(sh) : gain_lds_doubles(sh)) * sizeof(double);     // (rts_gain_kernel: STAGE)
    p->lds_scan = span_lds_doubles(sh, p->LP1, p->LP2) * sizeof(double);
    if (ekf) {
      switch (p->TPT_f) {
        case 1: PLAN_TRY(set_lds(gf_filter_kernel<1, 1, 0>, p->lds_filter)); break;
        case 2: PLAN_TRY(set_lds(gf_filter_kernel<2, 1, 0>, p->lds_filter)); break;
        default: PLAN_TRY(set_lds(gf_filter_kernel<4, 1, 0>, p->lds_filter)); break;
      }
    } else {
#define SL1(V) PLAN_TRY(set_lds(gf_filter_kernel<1, 0, V, 256>, p->lds_filter))
#define SL2(V) PLAN_TRY(set_lds(gf_filter_kernel<2, 0, V, 256>, p->lds_filter))
#define SL3(V) PLAN_TRY(set_lds(gf_filter_kernel<3, 0, V, 256>, p->lds_filter))
#define SL4(V) PLAN_TRY(set_lds(gf_filter_kernel<4, 0, V, 256>, p->lds_filter))
#define SL5(V) PLAN_TRY(set_lds(gf_filter_kernel<4, 0, V, 512>, p->lds_filter))
#define NAGP_SP_SWITCH(TP, CALLSP) switch (mom_variant(mc)) { case 1: CALLSP(TP, 1); break; case 2: CALLSP(TP, 2); break; case 3: CALLSP(TP, 3); break; \
        case 4: CALLSP(TP, 4); break; case 5: CALLSP(TP, 5); break; case 6: CALLSP(TP, 6); break; default: CALLSP(TP, 7); break; }
#define SLSP(TP, V) PLAN_TRY(set_lds(gf_filter_kernel<TP, 0, V, 256, 1>, p->lds_filter))
#define NAGP_SQ_SWITCH(TP, CALLSQ) switch (mom_variant(mc)) { case 1: CALLSQ(TP, 1); break; case 2: CALLSQ(TP, 2); break; case 3: CALLSQ(TP, 3); break; \
        case 4: CALLSQ(TP, 4); break; case 5: CALLSQ(TP, 5); break; default: CALLSQ(TP, 6); break; }
#define SLSQ(TP, V) PLAN_TRY(set_lds(gf_filter_kernel<TP, 0, V, 256, 2>, p->lds_filter))
      if (p->sq_gf) {
        switch (p->TPT_a) { case 1: NAGP_SQ_SWITCH(1, SLSQ) break; case 2: NAGP_SQ_SWITCH(2, SLSQ) break; case 3: NAGP_SQ_SWITCH(3, SLSQ) break; default: NAGP_SQ_SWITCH(4, SLSQ) break; }
      } else
#undef SLSQ
      if (p->sp_gf) {
        switch (p->TPT_a) { case 1: NAGP_SP_SWITCH(1, SLSP) break; case 2: NAGP_SP_SWITCH(2, SLSP) break; case 3: NAGP_SP_SWITCH(3, SLSP) break; default: NAGP_SP_SWITCH(4, SLSP) break; }
      } else if (p->LB_a == 512) { NAGP_MV_SWITCH(mom_variant(mc), SL5) }
      else switch (p->TPT_a) {
        case 1: NAGP_MV_SWITCH(mom_variant(mc), SL1) break;
        case 2: NAGP_MV_SWITCH(mom_variant(mc), SL2) break;
        case 3: NAGP_MV_SWITCH(mom_variant(mc), SL3) break;
        default: NAGP_MV_SWITCH(mom_variant(mc), SL4) break;
      }
#undef SLSP
      // 768-thread bound when the tiles fit: three waves per SIMD = 168 registers per lane (no spills; 30 spilled at the 1024 bound)
      if (p->wide_l && p->NT_l <= 768) PLAN_TRY(set_lds(gf_filter_kernel<1, 0, -1, 768>, p->lds_filter));
      else if (p->wide_l) PLAN_TRY(set_lds(gf_filter_kernel<1, 0, -1, 1024>, p->lds_filter));
      else switch (p->TPT_f) {   // mom-free kernel of the fixed-site steps
        case 1: PLAN_TRY(set_lds(gf_filter_kernel<1, 0, -1>, p->lds_filter)); break;
        case 2: PLAN_TRY(set_lds(gf_filter_kernel<2, 0, -1>, p->lds_filter)); break;
        default: PLAN_TRY(set_lds(gf_filter_kernel<4, 0, -1>, p->lds_filter)); break;
      }
#undef SL1
#undef SL2
#undef SL3
#undef SL4
#undef SL5
    }
    if (nt > 1024 && nt <= 1536 && sh.M * (sh.M + 1) / 2 <= 768 && sh.S <= 768 && !dev_env("NAGP_NO_GAIN768")) {
      p->gain768 = 1;
      p->lds_gain = gain_lds_doubles_staged(sh) * sizeof(double);
      PLAN_TRY(set_lds(rts_gain_kernel<2, 768>, p->lds_gain));
    }
    switch (p->TPT) {
      case 1: PLAN_TRY(set_lds(rts_gain_kernel<1>, p->lds_gain)); PLAN_TRY(set_lds(rts_compose_kernel<1>, p->lds_scan)); PLAN_TRY(set_lds(rts_boundary_kernel<1>, p->lds_scan)); PLAN_TRY(set_lds(rts_apply_kernel<1>, p->lds_scan)); break;
      case 2: PLAN_TRY(set_lds(rts_gain_kernel<2>, p->lds_gain)); PLAN_TRY(set_lds(rts_compose_kernel<2>, p->lds_scan)); PLAN_TRY(set_lds(rts_boundary_kernel<2>, p->lds_scan)); PLAN_TRY(set_lds(rts_apply_kernel<2>, p->lds_scan)); break;
      case 3: PLAN_TRY(set_lds(rts_gain_kernel<3>, p->lds_gain)); PLAN_TRY(set_lds(rts_compose_kernel<3>, p->lds_scan)); PLAN_TRY(set_lds(rts_boundary_kernel<3>, p->lds_scan)); PLAN_TRY(set_lds(rts_apply_kernel<3>, p->lds_scan)); break;
      case 4: PLAN_TRY(set_lds(rts_gain_kernel<4>, p->lds_gain)); PLAN_TRY(set_lds(rts_compose_kernel<4>, p->lds_scan)); PLAN_TRY(set_lds(rts_boundary_kernel<4>, p->lds_scan)); PLAN_TRY(set_lds(rts_apply_kernel<4>, p->lds_scan)); break;
      default: PLAN_TRY(set_lds(rts_gain_kernel<8>, p->lds_gain)); PLAN_TRY(set_lds(rts_compose_kernel<8>, p->lds_scan)); PLAN_TRY(set_lds(rts_boundary_kernel<8>, p->lds_scan)); PLAN_TRY(set_lds(rts_apply_kernel<8>, p->lds_scan)); break;
    }
  }
  if (!ekf && o->kind == NAGP_KIND_GF_EP && o->mode == NAGP_MODE_PREDICT && !(o->flags & NAGP_FLAG_MIXTURE_RULE) && 4 * sh.M <= 160 && dev_env("NAGP_LIN_MFMA")) {
    // fixed-site steps (sweeps >= 2) on the matrix cores: the plain predict-mode rule only.  Opt-in: measured on MI355X the step is
    // 13.7 us against 10.4 us of the 4x4-tile VALU kernel at S = 146 (6.2 against 3.85 at S = 73) -- DESIGN section 8
    const int ntl = (4 * sh.M + 15) / 16;
    p->lin_mfma = ntl;
    p->lds_lin = flm_lds_doubles(sh, ntl, 16) * sizeof(double);
    if (p->pipeline && B <= 128 && p->lds_lin < 160 * 1024) p->lds_lin = 160 * 1024;      // (the CU to itself, as for the other filter launches)
#define SETF(N, W) PLAN_TRY(set_lds((gf_filter_lin_mfma_kernel<N, W>), p->lds_lin))
    switch (ntl) { case 1: SETF(1, 4); break; case 2: SETF(2, 4); break; case 3: SETF(3, 4); break; case 4: SETF(4, 4); break; case 5: SETF(5, 4); break;
                   case 6: SETF(6, 8); break; case 7: SETF(7, 8); break; case 8: SETF(8, 8); break; case 9: SETF(9, 8); break; default: SETF(10, 8); break; }
#undef SETF
  }
  // rts_gain_mfma_kernel (16x16 tiles on the matrix cores, the dependence chain of the blocked Cholesky on a wave of its own) serves every
  // plan whose smoother passes take dense (G, Delta); NAGP_NO_GAIN_MFMA=1 (developer switch) keeps the 4x4-tile VALU kernel
  if (p->mfma_sp && !dev_env("NAGP_NO_GAIN_MFMA")) {
    p->gain_mfma = 1;
    const size_t lg = gainm_lds_doubles(p->mfma_sp / 16, sh) * sizeof(double);
    // The explicit-inverse form needs A^-1 per block.  It is used when EVERY block of every problem of the plan is comfortably
    // invertible -- |A_b^-1|_inf <= 8: A_b = expm(F_b dt) with dt = one sample is a slightly damped rotation for every kernel the
    // drivers use; a block with a length-scale below a sample would amplify the rounding error of PSkp^-1 by its |A_b^-1|, and such
    // plans keep the solve form (G = PS_k A' / L' / L, no inverse of A anywhere).  NAGP_GAIN_FORM=solve|inv (developer switch) forces one.
    {
      std::vector<double> ha((size_t)B * sh.M * 32, 0.0);
      double worst = 0.0; bool singular = false;
      for (int q = 0; q < B; ++q)
        for (int n = 0; n < sh.M; ++n) {
          const int bs = sh.bsz[n];
          const double* A0 = hm.data() + (size_t)q * msz + mdl_A(sh) + (size_t)n * 16;
          const double* Q0 = hm.data() + (size_t)q * msz + mdl_Q(sh) + (size_t)n * 16;
          double a[16], x[16];
          for (int e = 0; e < 16; ++e) { a[e] = A0[e]; x[e] = 0.0; }
          for (int i = 0; i < bs; ++i) x[4 * i + i] = 1.0;
          for (int col = 0; col < bs; ++col) {      // Gauss-Jordan, partial pivoting
            int piv = col;
            for (int r = col + 1; r < bs; ++r) if (std::fabs(a[4 * r + col]) > std::fabs(a[4 * piv + col])) piv = r;
            if (!(std::fabs(a[4 * piv + col]) > 0.0)) { singular = true; break; }
            for (int j = 0; j < 4; ++j) { std::swap(a[4 * col + j], a[4 * piv + j]); std::swap(x[4 * col + j], x[4 * piv + j]); }
            const double d = 1.0 / a[4 * col + col];
            for (int j = 0; j < 4; ++j) { a[4 * col + j] *= d; x[4 * col + j] *= d; }
            for (int r = 0; r < bs; ++r)
              if (r != col) { const double f = a[4 * r + col]; for (int j = 0; j < 4; ++j) { a[4 * r + j] -= f * a[4 * col + j]; x[4 * r + j] -= f * x[4 * col + j]; } }
          }
          double* o = ha.data() + ((size_t)q * sh.M + n) * 32;
          for (int i = 0; i < bs; ++i) {
            double rs = 0.0;
            for (int j = 0; j < bs; ++j) {
              o[4 * i + j] = x[4 * i + j]; rs += std::fabs(x[4 * i + j]);
              double wv = 0.0;
              for (int l = 0; l < bs; ++l) wv += x[4 * i + l] * Q0[4 * l + j];
              o[16 + 4 * i + j] = wv;
            }
            worst = std::max(worst, rs);
          }
        }
      const char* form = dev_env("NAGP_GAIN_FORM");
      p->gain_inv = (!singular && std::isfinite(worst) && (worst <= 8.0 || (form && !strcmp(form, "inv")))) ? 1 : 0;
      if (form && !strcmp(form, "solve")) p->gain_inv = 0;
      if (p->gain_inv) {
        PLAN_TRY(dalloc(p, &p->d_ainv, ha.size(), false));
        PLAN_HIP(hipMemcpyAsync(p->d_ainv, ha.data(), ha.size() * sizeof(double), hipMemcpyHostToDevice, p->stream));
        PLAN_HIP(hipStreamSynchronize(p->stream));
      }
    }
#define SETG(N) PLAN_TRY(set_lds((rts_gain_mfma_kernel<N, false>), lg)); PLAN_TRY(set_lds((rts_gain_mfma_kernel<N, true>), lg))
    switch (p->mfma_sp / 16) { case 1: SETG(1); break; case 2: SETG(2); break; case 3: SETG(3); break; case 4: SETG(4); break; case 5: SETG(5); break;
                               case 6: SETG(6); break; case 7: SETG(7); break; case 8: SETG(8); break; case 9: SETG(9); break; default: SETG(10); break; }
#undef SETG
  }
  if (p->big_sp) {
#define SETB(N) PLAN_TRY(set_lds(rts_big_kernel<N, 0>, p->lds_mfma)); PLAN_TRY(set_lds(rts_big_kernel<N, 1>, p->lds_mfma)); \
    PLAN_TRY(set_lds(rts_big_kernel<N, 2>, p->lds_mfma)); PLAN_TRY(set_lds(rts_big_phi_kernel<N>, p->lds_mfma))
    switch (p->mfma_sp / 16) { case 7: SETB(7); break; case 8: SETB(8); break; case 9: SETB(9); break; default: SETB(10); break; }
#undef SETB
  } else if (p->mfma_sp) {
#define SETM(N) PLAN_TRY(set_lds(rts_compose_mfma_kernel<N>, p->lds_mfma)); PLAN_TRY(set_lds(rts_boundary_mfma_kernel<N>, p->lds_mfma)); PLAN_TRY(set_lds(rts_apply_mfma_kernel<N>, p->lds_mfma))
    switch (p->mfma_sp / 16) { case 1: SETM(1); break; case 2: SETM(2); break; case 3: SETM(3); break; case 4: SETM(4); break; case 5: SETM(5); break; default: SETM(6); break; }
#undef SETM
  }
  if (!ekf) {
    p->DG_ep = pick_DG(o->lik_kind, o->n_pts, 256, sh.D, o->cub_dim);
    MomCfg t = mc; t.DG = p->DG_ep; t.cache_tabs = 1; t.store_a = (o->lik_kind == NAGP_LIK_POWER_NMF_SQRT) ? 1 : 0;
    if (p->src_all.n_src >= 2) {
      t.src = p->src_all;
      if (ep_lds_doubles(sh, t) * sizeof(double) <= 150 * 1024) p->src_ep = 1; else t.src = MomSrc{};
    }
    if (!p->src_ep && ep_lds_doubles(sh, t) * sizeof(double) > 64 * 1024) t.store_a = 0;
    if (!p->src_ep && ep_lds_doubles(sh, t) * sizeof(double) > 64 * 1024) t.cache_tabs = 0;
    p->cache_ep = t.cache_tabs; p->sta_ep = t.store_a;
    p->lds_ep = ep_lds_doubles(sh, t) * sizeof(double);
#define SL(V) PLAN_TRY(set_lds(ep_site_kernel<V>, p->lds_ep))
    NAGP_MV_SWITCH9(mom_variant(mc), SL)
#undef SL
    // site refresh in the staged sparse-point form (the conditions of the ADF launches: likModulatorNMFPower on a fully symmetric rule)
    if (p->sp.enabled && !p->src_ep && sh.M <= 64 && sh.D <= 4 * MSP_DT && o->cub_dim <= MSP_MAXCD && o->n_pts <= MSP_NT + 64 &&
        (o->n_pts + 3) / 4 <= MSP_NW * MSP_NST && !dev_env("NAGP_NO_SPARSE_EP")) {
      p->sp_ep = 1;
      p->lds_ep_sp = ep_sp_lds_doubles(sh, o->cub_dim) * sizeof(double);
#define SLS(V) PLAN_TRY(set_lds(ep_site_sp_kernel<V>, p->lds_ep_sp))
      switch (o->cub_dim) { case 1: SLS(1); break; case 2: SLS(2); break; case 3: SLS(3); break; case 4: SLS(4); break; case 5: SLS(5); break; case 6: SLS(6); break; default: SLS(7); break; }
#undef SLS
    }
    // ... and with likModulatorPreCalcwn in the staged form of nagp_momsq.hpp
    if (p->sq_ok && !p->src_ep && sh.M <= 64 && !dev_env("NAGP_NO_SPARSE_EP")) {
      p->sq_ep = 1;
      p->lds_ep_sq = ep_sq_lds_doubles(sh, o->cub_dim) * sizeof(double);
#define SLQ(V) PLAN_TRY(set_lds(ep_site_sq_kernel<V>, p->lds_ep_sq))
      switch (o->cub_dim) { case 1: SLQ(1); break; case 2: SLQ(2); break; case 3: SLQ(3); break; case 4: SLQ(4); break; case 5: SLQ(5); break; default: SLQ(6); break; }
#undef SLQ
    }
  }
  p->nlZ.assign((size_t)B * o->ep_itts, 0.0);
  p->mdM.assign((size_t)B * o->ep_itts, 0.0);
  p->mdP.assign((size_t)B * o->ep_itts, 0.0);
  PLAN_HIP(hipStreamSynchronize(p->stream));
  *out = p;
  return NAGP_OK;
}

extern "C" void nagp_plan_destroy(nagp_plan* p) {
  if (!p) return;
  // every stream of the plan is drained BEFORE its memory goes (an execute that failed half way may have left launches on the side streams)
  if (p->stream) (void)hipStreamSynchronize(p->stream);
  if (p->stream2) (void)hipStreamSynchronize(p->stream2);
  for (hipStream_t st : p->s_apply) (void)hipStreamSynchronize(st);
  for (void* v : p->allocs) (void)hipFree(v);
  for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
  if (p->ev_t0) (void)hipEventDestroy(p->ev_t0);
  if (p->ev_t1) (void)hipEventDestroy(p->ev_t1);
  for (hipStream_t st : p->s_apply) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  for (hipEvent_t e : p->ev_bnd) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_app) (void)hipEventDestroy(e);
  for (hipEvent_t e : p->ev_chunk) (void)hipEventDestroy(e);
  if (p->ev_red) (void)hipEventDestroy(p->ev_red);
  if (p->ev_filter) (void)hipEventDestroy(p->ev_filter);
  if (p->ev_s2) (void)hipEventDestroy(p->ev_s2);
  if (p->h_progress) (void)hipHostFree(p->h_progress);
  if (p->h_tab) (void)hipHostFree(p->h_tab);
  if (p->stream2) { (void)hipStreamSynchronize(p->stream2); (void)hipStreamDestroy(p->stream2); }
  if (p->stream) (void)hipStreamDestroy(p->stream);
  delete p;
}

extern "C" int64_t nagp_plan_device_bytes(const nagp_plan* p) { return p ? p->dev_bytes : 0; }

extern "C" int nagp_plan_upload_sites(nagp_plan* p, const double* const* ttau0, const double* const* tnu0) {
  if (!p) FAIL(NAGP_EINVAL, "null plan");
  if (p->opts.kind == NAGP_KIND_GIEKF) FAIL(NAGP_EINVAL, "the EKF path has no sites");
  if (!ttau0 && !tnu0) { p->warm = false; return NAGP_OK; }
  if (!ttau0 || !tnu0) FAIL(NAGP_EINVAL, "ttau0 and tnu0 come together");
  HIP_TRY(hipSetDevice(p->opts.device));
  const size_t n = (size_t)p->sh.T * p->sh.M;
  if (!p->d_tt0 || !p->d_tn0) {   // (a first call whose second allocation failed leaves d_tt0 set: test both)
    int st = p->d_tt0 ? NAGP_OK : dalloc(p, &p->d_tt0, (size_t)p->B * n, false);
    if (st == NAGP_OK) st = dalloc(p, &p->d_tn0, (size_t)p->B * n, false);
    if (st != NAGP_OK) return st;
  }
  for (int q = 0; q < p->B; ++q) {
    if (!ttau0[q] || !tnu0[q]) FAIL(NAGP_EINVAL, "problem %d: NULL site array", q);
    HIP_TRY(hipMemcpyAsync(p->d_tt0 + (size_t)q * n, ttau0[q], n * sizeof(double), hipMemcpyHostToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->d_tn0 + (size_t)q * n, tnu0[q], n * sizeof(double), hipMemcpyHostToDevice, p->stream));
  }
  HIP_TRY(hipStreamSynchronize(p->stream));
  p->warm = true;
  return NAGP_OK;
}

extern "C" int nagp_plan_upload_y(nagp_plan* p, const double* const* ys) {
  if (!p || !ys) FAIL(NAGP_EINVAL, "null argument");
  for (int q = 0; q < p->B; ++q)
    if (!ys[q]) FAIL(NAGP_EINVAL, "problem %d: NULL observation array", q);
  HIP_TRY(hipSetDevice(p->opts.device));
  for (int q = 0; q < p->B; ++q)
    HIP_TRY(hipMemcpyAsync(p->d_y + (size_t)q * p->sh.T, ys[q], (size_t)p->sh.T * sizeof(double), hipMemcpyHostToDevice, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return NAGP_OK;
}

// ---------------------------------------------------------------------------------------------
static int launch_filter(nagp_plan* p, const FilterPar& fp_in) {
  FilterPar fp = fp_in;
  fp.kb = p->kb_f;
  if (p->pipeline && fp.store_PF) { fp.progress = p->h_progress; fp.progress_every = 256; }
  if (const char* e = dev_env("NAGP_FILTER_DBG")) fp.dbg = atoi(e);   // developer switch: see FilterPar::dbg
  const bool ekf = p->opts.kind == NAGP_KIND_GIEKF;
  MomCfg mc = p->mc; mc.DG = p->DG_f; mc.cache_tabs = p->cache_f; mc.store_a = p->sta_f; mc.chunk_cap = p->chunk_cap_f;
  mc.sp = p->sp_gf ? p->sp : MomSp{};
  if (p->sq_gf) { mc.sq_form = 1; mc.sp.c0 = p->sq_c0; mc.store_a = 0; }
  if (dev_env("NAGP_STAMPS")) mc.stamps = reinterpret_cast<unsigned long long*>(p->d_stamps);   // developer diagnostics
  const bool adf = ekf || fp.mom_all || fp.k_end == p->sh.T;   // launches that may call mom (or the EKF filter)
  int nt_ekf = p->NT_f;
  if (ekf && p->NT_f + 64 <= 512 && p->sh.N <= 64) { nt_ekf = p->NT_f + 64; fp.spl_wave = 1; }   // one extra wave for the link
  Timed t(p, adf ? NAGP_K_FILTER : NAGP_K_FILTER_LIN);
  dim3 g(p->B), bl(p->NT_f);
  if (ekf) {
#define LF(TP) hipLaunchKernelGGL((gf_filter_kernel<TP, 1, 0>), g, dim3(nt_ekf), p->lds_filter, p->stream, p->sh, p->b, mc, fp)
    switch (p->TPT_f) { case 1: LF(1); break; case 2: LF(2); break; default: LF(4); break; }
#undef LF
  } else {
    if (adf && p->a8_gf && fp.mom_all && fp.k_end - fp.k_begin > 1) {
      // sweep 1 (mom at every step): role-specialised waves
      FilterPar fa = fp; fa.kb = p->kb_a8;
      MomCfg ma = mc; ma.sp = p->sp;
#define LA8(TP, V, PK) do { if (p->a8_st) hipLaunchKernelGGL((gf_adf8_kernel<2, V, PK, true>), g, dim3(MSR_NT), p->lds_a8, p->stream, p->sh, p->b, ma, fa); \
        else hipLaunchKernelGGL((gf_adf8_kernel<TP, V, PK, false>), g, dim3(MSR_NT), p->lds_a8, p->stream, p->sh, p->b, ma, fa); } while (0)
#define LA8V(TP, PK) switch (mc.cdim) { case 1: LA8(TP, 1, PK); break; case 2: LA8(TP, 2, PK); break; case 3: LA8(TP, 3, PK); break; \
        case 4: LA8(TP, 4, PK); break; case 5: LA8(TP, 5, PK); break; default: LA8(TP, 6, PK); break; }
      if (!p->a8_pack && mc.cdim == 7) { if (p->a8_tpt == 1) LA8(1, 7, false); else LA8(2, 7, false); }
      else if (p->a8_pack) { if (p->a8_tpt == 1) LA8V(1, true) else LA8V(2, true) }
      else { if (p->a8_tpt == 1) LA8V(1, false) else LA8V(2, false) }
#undef LA8V
#undef LA8
    } else
    if (adf) {
      dim3 ba(p->NT_a);
#define LF1(V) hipLaunchKernelGGL((gf_filter_kernel<1, 0, V, 256>), g, ba, p->lds_filter, p->stream, p->sh, p->b, mc, fp)
#define LF2(V) hipLaunchKernelGGL((gf_filter_kernel<2, 0, V, 256>), g, ba, p->lds_filter, p->stream, p->sh, p->b, mc, fp)
#define LF3(V) hipLaunchKernelGGL((gf_filter_kernel<3, 0, V, 256>), g, ba, p->lds_filter, p->stream, p->sh, p->b, mc, fp)
#define LF4(V) hipLaunchKernelGGL((gf_filter_kernel<4, 0, V, 256>), g, ba, p->lds_filter, p->stream, p->sh, p->b, mc, fp)
#define LF5(V) hipLaunchKernelGGL((gf_filter_kernel<4, 0, V, 512>), g, ba, p->lds_filter, p->stream, p->sh, p->b, mc, fp)
#define LFSP(TP, V) hipLaunchKernelGGL((gf_filter_kernel<TP, 0, V, 256, 1>), g, ba, p->lds_filter, p->stream, p->sh, p->b, mc, fp)
#define LFSQ(TP, V) hipLaunchKernelGGL((gf_filter_kernel<TP, 0, V, 256, 2>), g, ba, p->lds_filter, p->stream, p->sh, p->b, mc, fp)
      if (p->sq_gf) {
        switch (p->TPT_a) { case 1: NAGP_SQ_SWITCH(1, LFSQ) break; case 2: NAGP_SQ_SWITCH(2, LFSQ) break; case 3: NAGP_SQ_SWITCH(3, LFSQ) break; default: NAGP_SQ_SWITCH(4, LFSQ) break; }
      } else
#undef LFSQ
      if (p->sp_gf) {
        switch (p->TPT_a) { case 1: NAGP_SP_SWITCH(1, LFSP) break; case 2: NAGP_SP_SWITCH(2, LFSP) break; case 3: NAGP_SP_SWITCH(3, LFSP) break; default: NAGP_SP_SWITCH(4, LFSP) break; }
      } else
#undef LFSP
      if (p->LB_a == 512) { NAGP_MV_SWITCH(mom_variant(mc), LF5) }
      else switch (p->TPT_a) {
        case 1: NAGP_MV_SWITCH(mom_variant(mc), LF1) break;
        case 2: NAGP_MV_SWITCH(mom_variant(mc), LF2) break;
        case 3: NAGP_MV_SWITCH(mom_variant(mc), LF3) break;
        default: NAGP_MV_SWITCH(mom_variant(mc), LF4) break;
      }
#undef LF1
#undef LF2
#undef LF3
#undef LF4
#undef LF5
    } else if (p->lin_mfma && !fp.legacy_update && !fp.clamp_always && !fp.R_raw) {
      FilterPar fl = fp; fl.kb = 16;
#define LFM(N, W) hipLaunchKernelGGL((gf_filter_lin_mfma_kernel<N, W>), g, dim3(64 * W), p->lds_lin, p->stream, p->sh, p->b, fl)
      switch (p->lin_mfma) { case 1: LFM(1, 4); break; case 2: LFM(2, 4); break; case 3: LFM(3, 4); break; case 4: LFM(4, 4); break; case 5: LFM(5, 4); break;
                             case 6: LFM(6, 8); break; case 7: LFM(7, 8); break; case 8: LFM(8, 8); break; case 9: LFM(9, 8); break; default: LFM(10, 8); break; }
#undef LFM
    } else if (p->wide_l) {
      if (p->NT_l <= 768) hipLaunchKernelGGL((gf_filter_kernel<1, 0, -1, 768>), g, dim3(p->NT_l), p->lds_filter, p->stream, p->sh, p->b, mc, fp);
      else hipLaunchKernelGGL((gf_filter_kernel<1, 0, -1, 1024>), g, dim3(p->NT_l), p->lds_filter, p->stream, p->sh, p->b, mc, fp);
    } else {   // no step of this launch calls mom
      switch (p->TPT_f) {
        case 1: hipLaunchKernelGGL((gf_filter_kernel<1, 0, -1>), g, dim3(p->NT_fl), p->lds_filter, p->stream, p->sh, p->b, mc, fp); break;
        case 2: hipLaunchKernelGGL((gf_filter_kernel<2, 0, -1>), g, bl, p->lds_filter, p->stream, p->sh, p->b, mc, fp); break;
        default: hipLaunchKernelGGL((gf_filter_kernel<4, 0, -1>), g, bl, p->lds_filter, p->stream, p->sh, p->b, mc, fp); break;
      }
    }
  }
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

#define RUN(expr) do { int _s = (expr); if (_s != NAGP_OK) return _s; } while (0)

// ---- smoother of one sweep, chunk by chunk (chunks are cut from the END of the sequence: chunk 0 holds the latest steps)
struct ChunkGeom { int64_t k0; int nk; int L, ns; };
enum SmMode { SM_VALU = 0, SM_MFMA = 1, SM_BIG = 2 };

struct SweepCtx {
  bool write_PSs = false;
  SmMode mode = SM_VALU;
  std::vector<ChunkGeom> ch;     // [0] = latest steps
  std::vector<int> slot_of;      // chunk -> its own (G, Delta) buffer, or -1: not retained (slot 0 = scratch, gains recomputed)
  std::vector<char> composed;    // gain + compose of the chunk were enqueued on the second stream while the filter ran
  int next = -1;                 // next chunk the pump may start (counts down to 1; chunk 0 needs the complete filter)
  bool s2_used = false;
  bool xs = false;               // sweep_finish ran the cross-sweep form: apply + site refresh per chunk, ev_chunk[c] recorded behind each
};

static void sweep_begin(nagp_plan* p, SweepCtx& sc, bool write_PSs) {
  const Shape& sh = p->sh;
  sc.write_PSs = write_PSs;
  sc.mode = (p->big_sp && !write_PSs) ? SM_BIG : ((p->mfma_sp && !p->big_sp) ? SM_MFMA : SM_VALU);
  sc.ch.clear();
  // Chunk 0 (the latest steps) is all the pipeline cannot hide: its gains and compose pass need the complete filter.  It is cut
  // short (an eighth of a chunk, spans a quarter as long) whenever there is more than one chunk.
  const bool many = (sh.T - 1) > p->chunk;
  for (int64_t k1 = sh.T - 1; k1 > 0;) {
    const int nk = chunk_len(p, k1, sc.ch.empty());   // never beyond the buffer's capacity
    ChunkGeom g{k1 - nk, nk, 1, 1};
    // Spans.  The boundary pass is one sequential chain over ALL spans of the sweep (one workgroup per problem, a step per span); a
    // compose / apply launch costs a span LENGTH of latency, and the apply passes of the chunks run as one merged grid.
    //  * few workgroups (B * spans of the whole sweep <= 512: single sequences): latency decides -- one span length for the whole
    //    backward recursion, L* = sqrt((T-1) r) with r = boundary step : apply step (2.5 VALU passes; MFMA passes 0.5: 54 us per
    //    span against 109 us per step at Sp = 160, 13 against 31 at Sp = 80, profiles/r03_pipeline_timeline_*); per-chunk sqrt
    //    rules would multiply the boundary chain by sqrt(#chunks);
    //  * many workgroups (segments x spans fill the chip): throughput decides -- a chunk's launch should be whole rounds of the CUs the
    //    filter leaves free, spans as long as that allows (column-owner kernels: ~90 us per step of the three span passes, ~47 us
    //    per boundary span, measured at Sp = 160); the other kernels keep the sqrt(2.5 nk) rule under the workgroup cap.
    const double r_ba = (sc.mode == SM_VALU) ? 2.5 : 0.5;
    double Lstar = std::max(8.0, std::sqrt((double)(sh.T - 1) / (sc.mode == SM_VALU ? 2.5 : 1.0)));   // (regime test only)
    const bool latency_regime = (double)p->B * (double)(sh.T - 1) / Lstar <= 512.0;
    int ns;
    if (latency_regime) {
      // one span length for the sweep: the boundary chain costs (T/L) r, the merged apply grid ceil(B (T/L) / 256) rounds of L steps
      const double r = r_ba;
      double best = 1e300;
      for (int L = 8; L <= std::max<int64_t>(8, sh.T - 1); L += std::max(1, L / 64)) {
        const double spans = std::ceil((double)(sh.T - 1) / L);
        const double cost = spans * r + std::ceil(spans * p->B / 256.0) * L;
        if (cost < best) { best = cost; Lstar = L; }
      }
      // the two chunks of the latest steps are what the pipeline cannot hide (chunk 0 needs the complete filter, chunk 1's compose
      // pass is still running when the filter ends): shorter spans there -- a few more steps of the boundary chain for a quarter
      // of the compose latency
      if (many && sc.ch.size() <= 1) Lstar = std::max(8.0, Lstar / 4.0);
      ns = (int)std::lround((double)nk / Lstar);
      ns = std::max(1, std::min(std::min(ns, p->ns_max), (nk + 7) / 8));
    } else if (sc.mode == SM_BIG) {
      const int n_cu = std::max(32, 256 - p->B);
      double best = 1e300; ns = 1;
      for (int c = 1; c <= std::max(1, std::min(p->ns_max, (nk + 7) / 8)); ++c) {
        const int L = (nk + c - 1) / c, cc = (nk + L - 1) / L;
        const double rounds = std::ceil((double)cc * p->B / n_cu);
        const double cost = rounds * L * 90.0 + cc * 47.0;
        if (cost < best) { best = cost; ns = cc; }
      }
    } else {
      ns = (int)std::lround(std::sqrt(2.5 * (double)nk));
      ns = std::max(1, std::min(std::min(ns, p->ns_max), (nk + 7) / 8));
    }
    g.L = (nk + ns - 1) / ns;
    g.ns = (nk + g.L - 1) / g.L;
    sc.ch.push_back(g);
    k1 = g.k0;
  }
  const int nc = (int)sc.ch.size();
  sc.slot_of.assign(nc, -1);
  sc.composed.assign(nc, 0);
  if (p->pipeline) {
    // the short chunk 0 owns the small last slot; full slots 1 .. n_full-1 belong to the chunks the filter finishes last (slot 0 is
    // the scratch of the others) -- or, with a full slot for every other chunk, slot c-1 to chunk c
    const int n_full = p->n_full_slots;
    sc.slot_of[0] = n_full;
    if (p->n_recycled > 0) {
      // every chunk owns a slot: the earliest n_full chunks the full ones, the later ones the slots recycled from PF (plan creation)
      for (int c = 1; c < nc; ++c) { const int tau = nc - 1 - c; sc.slot_of[c] = tau < n_full ? tau : n_full + 1 + (tau - n_full); }
    } else
    if (n_full >= nc - 1) for (int c = 1; c < nc; ++c) sc.slot_of[c] = c - 1;
    else for (int c = 1; c < n_full; ++c) sc.slot_of[c] = c;
    sc.next = nc - 1;
    std::memset(p->h_progress, 0, (size_t)p->B * sizeof(unsigned long long));
  } else {
    sc.next = 0;
  }
  sc.s2_used = false;
}

// Ownership map of rts_gain_kernel<2, 768> (GainPar::gmapB / gmapL): the B groups (64 column-major tiles each) are paired early with
// late -- group g with group nB-1-g -- so that every wave's two slots together take part in about M trailing updates of the
// factorisation AND about M of the backward solve; the lower-triangle groups (their cost grows with the column) go heaviest first to the
// wave with the lightest load of its SIMD (waves w, w+4, w+8 share one).
static void gain_map(const Shape& sh, GainPar& gp) {
  const int M = sh.M, nB = (sh.ntiles + 63) / 64, nlow = M * (M + 1) / 2, nL = (nlow + 63) / 64;
  gp.use_map = 0;
  if (nB > 24 || nL > 12 || !dev_env("NAGP_GAIN_MAP")) return;      // opt-in: measured without effect (profiles/r04_gain_phases.txt)
  for (int w = 0; w < 12; ++w) { gp.gmapB[0][w] = gp.gmapB[1][w] = gp.gmapL[w] = -1; }
  double load[12];
  auto colB = [&](int g) { return ((double)g * 64 + 32) / M; };                 // column of the middle tile of a B group
  auto colL = [&](int g) {                                                        // ... of a lower-triangle group
    const int t = std::min(g * 64 + 32, nlow - 1);
    int J = 0;
    while (J + 1 < M && (J + 1) * M - (J + 1) * J / 2 <= t) ++J;
    return (double)J;
  };
  int lo = 0, hi = nB - 1, w = 0;
  for (; lo < hi && w < 12; ++lo, --hi, ++w) { gp.gmapB[0][w] = (signed char)lo; gp.gmapB[1][w] = (signed char)hi; load[w] = colB(lo) + colB(hi); }
  if (lo == hi && w < 12) { gp.gmapB[0][w] = (signed char)lo; load[w] = colB(lo); ++w; }
  for (; w < 12; ++w) load[w] = 0.0;
  for (int g = nL - 1; g >= 0; --g) {                                             // heaviest lower group first
    int best = -1; double bl = 0.0;
    for (int v = 0; v < 12; ++v) {
      if (gp.gmapL[v] >= 0) continue;
      const double simd = load[v] + load[(v + 4) % 12] + load[(v + 8) % 12];    // the SIMD's load decides, the wave's own breaks ties
      const double key = simd * 16.0 + load[v];
      if (best < 0 || key < bl) { best = v; bl = key; }
    }
    gp.gmapL[best] = (signed char)g; load[best] += colL(g);
  }
  gp.use_map = 1;
}

static int launch_gain_chunk(nagp_plan* p, const SweepCtx& sc, int c, int slot, hipStream_t st) {
  const Shape& sh = p->sh; const ChunkGeom& g = sc.ch[c];
  GainPar gp{};
  gp.k0 = g.k0; gp.nk = g.nk; gp.chunk = p->slot_cap[slot]; gp.dense_sp = (sc.mode != SM_VALU) ? p->mfma_sp : 0;
  gp.dbg = dev_env("NAGP_GAINM_DBG") ? atoi(dev_env("NAGP_GAINM_DBG")) : 0;
  if (dev_env("NAGP_STAMPS") && p->d_gstamps) gp.stamps = reinterpret_cast<unsigned long long*>(p->d_gstamps);
  gp.use_map = 0;
  if (p->gain768) gain_map(sh, gp);
  gp.dpacked = (sc.mode == SM_BIG) ? p->dpacked : 0;
  if (gp.dense_sp && p->slot_tiled[slot]) {
    HIP_TRY(hipMemsetAsync(p->slotG[slot], 0, (size_t)p->B * p->slot_cap[slot] * p->gstep * sizeof(double), st)); p->slot_tiled[slot] = 0;
  }
  if (!gp.dense_sp && p->mfma_sp) p->slot_tiled[slot] = 1;
  Bufs b = p->b; b.Gbuf = p->slotG[slot]; b.dbuf = p->slotD[slot]; b.gpstride = p->slot_gps[slot];
  Timed t(p, NAGP_K_GAIN, st);
  dim3 gr(g.nk, p->B), bl(p->NT);
  if (gp.dense_sp && p->gain_mfma) {
    const int ntl = p->mfma_sp / 16;
    const size_t lg = gainm_lds_doubles(ntl, sh) * sizeof(double);
    const dim3 gr8((unsigned)((g.nk + 7) / 8 * 8), (unsigned)p->B);      // (the steps of one XCD contiguous: nagp_gain_mfma.hpp)
    gp.ainv = p->gain_inv ? p->d_ainv : nullptr;
#define LG(N) do { if (p->gain_inv) hipLaunchKernelGGL((rts_gain_mfma_kernel<N, true>), gr8, dim3(64 * (N + 1)), lg, st, sh, b, gp); \
                   else hipLaunchKernelGGL((rts_gain_mfma_kernel<N, false>), gr8, dim3(64 * (N + 1)), lg, st, sh, b, gp); } while (0)
    switch (ntl) { case 1: LG(1); break; case 2: LG(2); break; case 3: LG(3); break; case 4: LG(4); break; case 5: LG(5); break;
                   case 6: LG(6); break; case 7: LG(7); break; case 8: LG(8); break; case 9: LG(9); break; default: LG(10); break; }
#undef LG
  } else
  if (p->gain768) hipLaunchKernelGGL((rts_gain_kernel<2, 768>), gr, dim3(768), p->lds_gain, st, sh, b, gp);
  else switch (p->TPT) {
    case 1: hipLaunchKernelGGL((rts_gain_kernel<1>), gr, bl, p->lds_gain, st, sh, b, gp); break;
    case 2: hipLaunchKernelGGL((rts_gain_kernel<2>), gr, bl, p->lds_gain, st, sh, b, gp); break;
    case 3: hipLaunchKernelGGL((rts_gain_kernel<3>), gr, bl, p->lds_gain, st, sh, b, gp); break;
    case 4: hipLaunchKernelGGL((rts_gain_kernel<4>), gr, bl, p->lds_gain, st, sh, b, gp); break;
    default: hipLaunchKernelGGL((rts_gain_kernel<8>), gr, bl, p->lds_gain, st, sh, b, gp); break;
  }
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

static SpanPar span_par(nagp_plan* p, const SweepCtx& sc, int c, int slot) {
  const ChunkGeom& g = sc.ch[c];
  const int set = p->pipeline ? c : 0;
  SpanPar sp = p->spar;
  sp.k0 = g.k0; sp.nk = g.nk; sp.chunk = p->slot_cap[slot]; sp.ns_max = p->ns_max; sp.LP1 = p->LP1; sp.LP2 = p->LP2;
  sp.first = (c == 0) ? 1 : 0; sp.write_PSs = sc.write_PSs ? 1 : 0; sp.L = g.L; sp.ns = g.ns;
  sp.spanbuf = p->c_spanbuf[set]; sp.spanvec = p->c_spanvec[set]; sp.xbuf = p->c_xbuf[set]; sp.bnd = p->c_bnd[set];
  sp.tab = nullptr; sp.ntab = 0;
  return sp;
}
static MfmaPar mfma_par(nagp_plan* p, const SweepCtx& sc, int c, int slot) {
  const ChunkGeom& g = sc.ch[c];
  const int set = p->pipeline ? c : 0;
  MfmaPar mp = p->mpar;
  mp.k0 = g.k0; mp.nk = g.nk; mp.chunk = p->slot_cap[slot]; mp.L = g.L; mp.ns = g.ns; mp.ns_max = p->ns_max; mp.Sp = p->mfma_sp;
  mp.first = (c == 0) ? 1 : 0; mp.write_PSs = (sc.mode == SM_MFMA && sc.write_PSs) ? 1 : 0;
  mp.spanbuf = p->c_mspanbuf[set]; mp.spanvec = p->c_mspanvec[set]; mp.bnd = p->c_mbnd[set];
  mp.tab = nullptr; mp.ntab = 0; mp.xbuf = nullptr;
  mp.dpacked = (sc.mode == SM_BIG) ? p->dpacked : 0;
  return mp;
}

// pass 1 of the span scheme (one workgroup per span): reads the chunk's (G, Delta, delta), writes its (Phi, C, c)
static int launch_compose_chunk(nagp_plan* p, const SweepCtx& sc, int c, int slot, hipStream_t st) {
  const Shape& sh = p->sh; const ChunkGeom& g = sc.ch[c];
  Bufs b = p->b; b.Gbuf = p->slotG[slot]; b.dbuf = p->slotD[slot]; b.gpstride = p->slot_gps[slot];
  Timed t(p, NAGP_K_SCAN, st);
  if (sc.mode == SM_BIG) {
    MfmaPar mp = mfma_par(p, sc, c, slot);
    const int ntl = p->mfma_sp / 16;
    dim3 gr(g.ns, p->B), bl(64 * ntl);
#define LB(N) do { \
      hipLaunchKernelGGL((rts_big_phi_kernel<N>), gr, bl, p->lds_mfma, st, sh, b, mp); \
      hipLaunchKernelGGL((rts_big_kernel<N, 0>), gr, bl, p->lds_mfma, st, sh, b, mp); } while (0)
    switch (ntl) { case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
#undef LB
  } else if (sc.mode == SM_MFMA) {
    MfmaPar mp = mfma_par(p, sc, c, slot);
    dim3 gr(g.ns, p->B), bl(256);
#define LM(N) hipLaunchKernelGGL((rts_compose_mfma_kernel<N>), gr, bl, p->lds_mfma, st, sh, b, mp)
    switch (p->mfma_sp / 16) { case 1: LM(1); break; case 2: LM(2); break; case 3: LM(3); break; case 4: LM(4); break; case 5: LM(5); break; default: LM(6); break; }
#undef LM
  } else {
    SpanPar sp = span_par(p, sc, c, slot);
    dim3 gr(g.ns, p->B), bl(p->NT);
#define LS(TP) hipLaunchKernelGGL((rts_compose_kernel<TP>), gr, bl, p->lds_scan, st, sh, b, sp)
    switch (p->TPT) { case 1: LS(1); break; case 2: LS(2); break; case 3: LS(3); break; case 4: LS(4); break; default: LS(8); break; }
#undef LS
  }
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

// pass 2: boundary values over the spans of the chunk (sequential; continues from the chunk behind it in time through the carry
// state), pass 3: the reference recursion inside every span from its boundary value
static int launch_boundary_chunk(nagp_plan* p, const SweepCtx& sc, int c, int slot, hipStream_t st) {
  const Shape& sh = p->sh;
  Bufs b = p->b; b.Gbuf = p->slotG[slot]; b.dbuf = p->slotD[slot]; b.gpstride = p->slot_gps[slot];
  Timed t(p, NAGP_K_SCAN, st);
  dim3 g2(p->B);
  if (sc.mode == SM_BIG) {
    MfmaPar mp = mfma_par(p, sc, c, slot);
    const int ntl = p->mfma_sp / 16;
    dim3 bl(64 * ntl);
#define LB(N) hipLaunchKernelGGL((rts_big_kernel<N, 1>), g2, bl, p->lds_mfma, st, sh, b, mp)
    switch (ntl) { case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
#undef LB
  } else if (sc.mode == SM_MFMA) {
    MfmaPar mp = mfma_par(p, sc, c, slot);
    dim3 bl(256);
#define LM(N) hipLaunchKernelGGL((rts_boundary_mfma_kernel<N>), g2, bl, p->lds_mfma, st, sh, b, mp)
    switch (p->mfma_sp / 16) { case 1: LM(1); break; case 2: LM(2); break; case 3: LM(3); break; case 4: LM(4); break; case 5: LM(5); break; default: LM(6); break; }
#undef LM
  } else {
    SpanPar sp = span_par(p, sc, c, slot);
    dim3 bl(p->NT);
#define LS(TP) hipLaunchKernelGGL((rts_boundary_kernel<TP>), g2, bl, p->lds_scan, st, sh, b, sp)
    switch (p->TPT) { case 1: LS(1); break; case 2: LS(2); break; case 3: LS(3); break; case 4: LS(4); break; default: LS(8); break; }
#undef LS
  }
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

static int launch_apply_chunk(nagp_plan* p, const SweepCtx& sc, int c, int slot, hipStream_t st) {
  const Shape& sh = p->sh; const ChunkGeom& g = sc.ch[c];
  Bufs b = p->b; b.Gbuf = p->slotG[slot]; b.dbuf = p->slotD[slot]; b.gpstride = p->slot_gps[slot];
  Timed t(p, NAGP_K_SCAN, st);
  if (sc.mode == SM_BIG) {
    MfmaPar mp = mfma_par(p, sc, c, slot);
    const int ntl = p->mfma_sp / 16;
    dim3 gr(g.ns, p->B), bl(64 * ntl);
#define LB(N) hipLaunchKernelGGL((rts_big_kernel<N, 2>), gr, bl, p->lds_mfma, st, sh, b, mp)
    switch (ntl) { case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
#undef LB
  } else if (sc.mode == SM_MFMA) {
    MfmaPar mp = mfma_par(p, sc, c, slot);
    dim3 gr(g.ns, p->B), bl(256);
#define LM(N) hipLaunchKernelGGL((rts_apply_mfma_kernel<N>), gr, bl, p->lds_mfma, st, sh, b, mp)
    switch (p->mfma_sp / 16) { case 1: LM(1); break; case 2: LM(2); break; case 3: LM(3); break; case 4: LM(4); break; case 5: LM(5); break; default: LM(6); break; }
#undef LM
  } else {
    SpanPar sp = span_par(p, sc, c, slot);
    dim3 gr(g.ns, p->B), bl(p->NT);
#define LS(TP) hipLaunchKernelGGL((rts_apply_kernel<TP>), gr, bl, p->lds_scan, st, sh, b, sp)
    switch (p->TPT) { case 1: LS(1); break; case 2: LS(2); break; case 3: LS(3); break; case 4: LS(4); break; default: LS(8); break; }
#undef LS
  }
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

// While the filter launches of this sweep run on the main stream: start gain + compose of every chunk whose steps (and the one
// behind them, for Delta_k = PS_{k+1} - ...) the filter has published, on the second stream.  Returns when the filter has finished.
static int sweep_pump(nagp_plan* p, SweepCtx& sc) {
  if (!p->pipeline) return NAGP_OK;
  HIP_TRY(hipEventRecord(p->ev_filter, p->stream));
  auto try_start = [&]() -> int {
    unsigned long long done = ~0ull;
    for (int q = 0; q < p->B; ++q) {
      const unsigned long long v = __atomic_load_n(&p->h_progress[q], __ATOMIC_ACQUIRE);
      done = std::min(done, v);
    }
    while (sc.next >= 1) {
      const ChunkGeom& g = sc.ch[sc.next];
      if (done < (unsigned long long)(g.k0 + g.nk + 1)) break;       // steps k0 .. k0+nk (inclusive) are needed
      const int c = sc.next;
      const int slot = sc.slot_of[c] >= 0 ? sc.slot_of[c] : 0;
      RUN(launch_gain_chunk(p, sc, c, slot, p->stream2));
      RUN(launch_compose_chunk(p, sc, c, slot, p->stream2));
      sc.composed[c] = 1; sc.s2_used = true;
      --sc.next;
    }
    return NAGP_OK;
  };
  for (;;) {
    const hipError_t e = hipEventQuery(p->ev_filter);
    if (e == hipSuccess) break;
    if (e != hipErrorNotReady) HIP_TRY(e);
    RUN(try_start());
    std::this_thread::sleep_for(std::chrono::microseconds(50));
  }
  RUN(try_start());      // whatever the last poll missed (chunk 0 stays with the main stream)
  if (sc.s2_used) HIP_TRY(hipEventRecord(p->ev_s2, p->stream2));
  return NAGP_OK;
}

// The apply passes of chunks [0, n_own) -- those with their own (G, Delta) buffer -- as ONE grid on the side stream.
static int launch_apply_merged(nagp_plan* p, const SweepCtx& sc, int n_own, hipStream_t st) {
  const Shape& sh = p->sh;
  int tot = 0;
  for (int c = 0; c < n_own; ++c) {
    const ChunkGeom& g = sc.ch[c];
    ChunkTab& t = p->h_tab[c];
    const int slot = sc.slot_of[c];
    t.k0 = g.k0; t.nk = g.nk; t.L = g.L; t.ns = g.ns; t.first = (c == 0) ? 1 : 0; t.span0 = tot; t.cap = p->slot_cap[slot];
    t.G = p->slotG[slot]; t.d = p->slotD[slot]; t.gps = p->slot_gps[slot];
    if (sc.mode == SM_VALU) { t.spanbuf = p->c_spanbuf[c]; t.spanvec = p->c_spanvec[c]; t.bnd = p->c_bnd[c]; t.xbuf = p->c_xbuf[c]; }
    else { t.spanbuf = p->c_mspanbuf[c]; t.spanvec = p->c_mspanvec[c]; t.bnd = p->c_mbnd[c]; t.xbuf = nullptr; }
    tot += g.ns;
  }
  Bufs b = p->b; b.Gbuf = p->slotG[sc.slot_of[0]]; b.dbuf = p->slotD[sc.slot_of[0]]; b.gpstride = p->slot_gps[sc.slot_of[0]];
  Timed t(p, NAGP_K_SCAN, st);
  dim3 gr(tot, p->B);
  if (sc.mode == SM_BIG) {
    MfmaPar mp = mfma_par(p, sc, 0, sc.slot_of[0]); mp.tab = p->h_tab; mp.ntab = n_own;
    const int ntl = p->mfma_sp / 16;
    dim3 bl(64 * ntl);
#define LB(N) hipLaunchKernelGGL((rts_big_kernel<N, 2>), gr, bl, p->lds_mfma, st, sh, b, mp)
    switch (ntl) { case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
#undef LB
  } else if (sc.mode == SM_MFMA) {
    MfmaPar mp = mfma_par(p, sc, 0, sc.slot_of[0]); mp.tab = p->h_tab; mp.ntab = n_own;
    dim3 bl(256);
#define LM(N) hipLaunchKernelGGL((rts_apply_mfma_kernel<N>), gr, bl, p->lds_mfma, st, sh, b, mp)
    switch (p->mfma_sp / 16) { case 1: LM(1); break; case 2: LM(2); break; case 3: LM(3); break; case 4: LM(4); break; case 5: LM(5); break; default: LM(6); break; }
#undef LM
  } else {
    SpanPar sp = span_par(p, sc, 0, sc.slot_of[0]); sp.tab = p->h_tab; sp.ntab = n_own;
    dim3 bl(p->NT);
#define LS(TP) hipLaunchKernelGGL((rts_apply_kernel<TP>), gr, bl, p->lds_scan, st, sh, b, sp)
    switch (p->TPT) { case 1: LS(1); break; case 2: LS(2); break; case 3: LS(3); break; case 4: LS(4); break; default: LS(8); break; }
#undef LS
  }
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

// After the filter: chunk by chunk from the end of the sequence -- (gain, compose unless they ran beside the filter,) boundary pass on
// the main stream (the carry between chunks is its order).  The apply passes of the chunks that own a (G, Delta) buffer run as ONE
// merged grid on a side stream once the boundary chain has passed them (a span length of latency instead of one per chunk);
// the chunks without a buffer follow on the main stream: gains again into the scratch buffer, boundary, apply.
using EpRange = std::function<int(int64_t, int64_t, hipStream_t)>;
static int sweep_finish(nagp_plan* p, SweepCtx& sc, const EpRange* ep_chunk = nullptr) {
  const int nc = (int)sc.ch.size();
  bool waited = !sc.s2_used;       // the main stream has to wait ONCE for the second stream's work (one event behind all of it)
  auto wait_s2 = [&]() -> int {
    if (!waited) { HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_s2, 0)); waited = true; }
    return NAGP_OK;
  };
  int n_own = 0;                   // chunks 0 .. n_own-1 own a buffer (pipelined plans; slot_of is a prefix by construction)
  if (p->pipeline) while (n_own < nc && sc.slot_of[n_own] >= 0) ++n_own;
  // Cross-sweep form (every chunk owns a buffer, another sweep follows): behind the boundary chain the apply pass and the site refresh
  // run chunk by chunk on the side stream, the chunk of the EARLIEST steps first, and an event behind each lets the next sweep's filter
  // follow them chunk by chunk -- only the first chunk's apply + refresh stays exposed.  Chunks whose buffer is recycled from PF (it lies
  // in the PF of early steps, which that filter overwrites first) go before all others.
  sc.xs = ep_chunk && p->xsweep && n_own == nc && nc > 1;
  for (int c = 0; c < nc; ++c) {
    const int slot = sc.slot_of[c] >= 0 ? sc.slot_of[c] : 0;
    if (!sc.composed[c]) {
      // chunk 0 (it needs the complete filter), and every chunk of a serial plan
      if (sc.slot_of[c] < 0) RUN(wait_s2());                 // slot 0 may still be the second stream's scratch
      RUN(launch_gain_chunk(p, sc, c, slot, p->stream));
      RUN(launch_compose_chunk(p, sc, c, slot, p->stream));
    } else {
      RUN(wait_s2());                                        // its compose results (and gains) come from the second stream
      if (sc.slot_of[c] < 0) RUN(launch_gain_chunk(p, sc, c, slot, p->stream));   // gains dropped after the compose pass: recompute
    }
    RUN(launch_boundary_chunk(p, sc, c, slot, p->stream));
    if (c < n_own) {
      if (c == n_own - 1) {
        hipStream_t st = p->s_apply[0];
        HIP_TRY(hipEventRecord(p->ev_bnd[0], p->stream));
        HIP_TRY(hipStreamWaitEvent(st, p->ev_bnd[0], 0));
        if (sc.xs) {
          while ((int)p->ev_chunk.size() < nc) { hipEvent_t e = nullptr; HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->ev_chunk.push_back(e); }
          std::vector<int> order;
          for (int c2 = 0; c2 < nc; ++c2) if (p->slot_gps[sc.slot_of[c2]] != 0) order.push_back(c2);
          for (int c2 = nc - 1; c2 >= 0; --c2) if (p->slot_gps[sc.slot_of[c2]] == 0) order.push_back(c2);
          for (int c2 : order) {
            RUN(launch_apply_chunk(p, sc, c2, sc.slot_of[c2], st));
            RUN((*ep_chunk)(sc.ch[c2].k0, sc.ch[c2].k0 + sc.ch[c2].nk, st));
            HIP_TRY(hipEventRecord(p->ev_chunk[c2], st));
          }
        } else {
          RUN(launch_apply_merged(p, sc, n_own, st));
          HIP_TRY(hipEventRecord(p->ev_app[0], st));
        }
      }
    } else {
      RUN(launch_apply_chunk(p, sc, c, slot, p->stream));    // (scratch buffer: the next chunk's gains overwrite it)
    }
  }
  RUN(wait_s2());
  if (n_own > 0 && !sc.xs) HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_app[0], 0));
  return NAGP_OK;
}

static bool mixture_rule(const nagp_plan* p) { return (p->opts.flags & NAGP_FLAG_MIXTURE_RULE) != 0; }

static int launch_ep(nagp_plan* p, double alpha, double damp, int clamp, int write_R, double* lZ_out, int64_t k_lo = 0, int64_t k_hi = -1, hipStream_t st = nullptr) {
  const Shape& sh = p->sh;
  if (sh.T < 2) return NAGP_OK;
  MomCfg mc = p->mc; mc.DG = p->DG_ep; mc.cache_tabs = p->cache_ep; mc.store_a = p->sta_ep;
  if (p->src_ep) mc.src = p->src_all;
  EpPar ep{};
  if (!st) st = p->stream;
  ep.k_begin = k_lo;
  ep.k_end = (k_hi < 0) ? sh.T - 1 : k_hi;          // steps [k_lo, k_end): the whole sequence, or one smoother chunk (cross-sweep schedule)
  if (ep.k_end <= ep.k_begin) return NAGP_OK;
  // ~8192 workgroups over all problems (32 per CU): enough to fill the chip, and the per-workgroup set-up (cubature tables, the static
  // addresses of the sparse-point stages) is amortised over the steps of a workgroup when many problems share the launch
  ep.steps_per_wg = (int)std::max<int64_t>(1, ((int64_t)p->B * (sh.T - 1) + 8191) / 8192);
  ep.alpha = alpha; ep.clamp = clamp; ep.write_R = write_R; ep.lZ_out = lZ_out;
  if (mixture_rule(p)) { ep.w_old = 1.0 - damp; ep.w_new = damp / alpha; }
  else { ep.w_old = 1.0 - damp * alpha; ep.w_new = damp; }
  Timed t(p, NAGP_K_EPSITE, st);
  dim3 g((unsigned)((ep.k_end - ep.k_begin + ep.steps_per_wg - 1) / ep.steps_per_wg), p->B), bl(256);
  if (p->sq_ep) {
    MomCfg ms = mc; ms.sp = MomSp{}; ms.sp.c0 = p->sq_c0; ms.src = MomSrc{};
#define LEQ(V) hipLaunchKernelGGL(ep_site_sq_kernel<V>, g, dim3(256), p->lds_ep_sq, st, sh, p->b, ms, ep)
    switch (ms.cdim) { case 1: LEQ(1); break; case 2: LEQ(2); break; case 3: LEQ(3); break; case 4: LEQ(4); break; case 5: LEQ(5); break; default: LEQ(6); break; }
#undef LEQ
    HIP_TRY(hipGetLastError());
    return NAGP_OK;
  }
  if (p->sp_ep) {
    MomCfg ms = mc; ms.sp = p->sp; ms.src = MomSrc{};
#define LES(V) hipLaunchKernelGGL(ep_site_sp_kernel<V>, g, dim3(MSP_NT), p->lds_ep_sp, st, sh, p->b, ms, ep)
    switch (ms.cdim) { case 1: LES(1); break; case 2: LES(2); break; case 3: LES(3); break; case 4: LES(4); break; case 5: LES(5); break; case 6: LES(6); break; default: LES(7); break; }
#undef LES
    HIP_TRY(hipGetLastError());
    return NAGP_OK;
  }
#define LE(V) hipLaunchKernelGGL(ep_site_kernel<V>, g, bl, p->lds_ep, st, sh, p->b, mc, ep)
  NAGP_MV_SWITCH9(mom_variant(mc), LE)
#undef LE
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

static int reduce_sum(nagp_plan* p, const double* v, int64_t k_lo, int64_t k_hi, int slot, hipStream_t st = nullptr, double* out = nullptr) {
  if (!st) st = p->stream;
  Timed t(p, NAGP_K_REDUCE, st);
  hipLaunchKernelGGL(sum_kernel, dim3(p->B), dim3(1024), 0, st, v, p->sh.T, k_lo, k_hi, out ? out : p->b.red, slot);
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

static int fetch_red(nagp_plan* p, std::vector<double>& h) {
  h.resize((size_t)p->B * 8);
  HIP_TRY(hipMemcpyAsync(h.data(), p->b.red, h.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  return NAGP_OK;
}

// copy the filtered marginals / mean of the last step into the smoothed arrays (the smoother never
// visits k = T-1: gf_ep_modulator_nmf.m:207)
static __global__ void seed_last_kernel(Bufs b, int64_t T, int M, int S, int with_sv) {
  const size_t o = (size_t)blockIdx.x * T + (T - 1);
  for (int i = threadIdx.x; i < M; i += blockDim.x) { b.sm[o * M + i] = b.fm[o * M + i]; if (with_sv) b.sv[o * M + i] = b.fv[o * M + i]; }
  for (int i = threadIdx.x; i < S; i += blockDim.x) b.MS[o * S + i] = b.MF[o * S + i];
}
static int seed_last_step(nagp_plan* p) {      // one launch for all problems (a batch of 256 segments made 768 small copies of it)
  const Shape& sh = p->sh;
  hipLaunchKernelGGL(seed_last_kernel, dim3(p->B), dim3(256), 0, p->stream, p->b, sh.T, sh.M, sh.S, p->opts.kind != NAGP_KIND_IHGP ? 1 : 0);
  HIP_TRY(hipGetLastError());
  return NAGP_OK;
}

// red[q][1], red[q][2] (maxDiffM, maxDiffP) <- 0 for every problem
static __global__ void zero_maxdiff_kernel(double* red, int B) {
  const int q = blockIdx.x * blockDim.x + threadIdx.x;
  if (q < B) { red[(size_t)q * 8 + 1] = 0.0; red[(size_t)q * 8 + 2] = 0.0; }
}

static int zero_async(nagp_plan* p, void* ptr, size_t bytes) {
  HIP_TRY(hipMemsetAsync(ptr, 0, bytes, p->stream));
  return NAGP_OK;
}

static int exec_gf(nagp_plan* p) {
  const Shape& sh = p->sh; const nagp_opts& o = p->opts; const int I = o.ep_itts, B = p->B;
  const bool nlml = (o.mode == NAGP_MODE_NLML);
  // Reduction records, one per sweep, fetched once at the end (no host synchronisation between the sweeps):
  // record 0 = sum of the filter's lZ (sweep 1; nlml: the final sum), record itt = (lZ sum after the refresh, maxDiffM, maxDiffP) of sweep itt
  const size_t RR = (size_t)B * 8;
  RUN(zero_async(p, p->red_all, (size_t)(I + 2) * RR * sizeof(double)));
  struct RestoreRed { nagp_plan* p; ~RestoreRed() { p->b.red = p->red0; } } restore{p};
  bool xs_pending = false;             // the previous sweep ended in the cross-sweep form: ev_chunk[c] per chunk, ev_red behind everything
  std::vector<ChunkGeom> xs_ch;
  for (int itt = 1; itt <= I; ++itt) {
    p->b.red = p->red_all + (size_t)itt * RR;
    const bool run_filter = !nlml || itt == 1 || itt < I;
    const bool run_smoother = !nlml || itt < I;
    SweepCtx sc;
    const bool smooth = run_smoother && run_filter && sh.T > 1;
    if (smooth) sweep_begin(p, sc, p->want_PS && itt == I);
    if (run_filter) {
      FilterPar fp{};
      fp.itt = itt; fp.ep_damp = p->damping[itt - 1]; fp.mom_all = (itt == 1);
      const bool mix = mixture_rule(p);
      fp.legacy_update = nlml || mix; fp.clamp_always = nlml || mix; fp.write_R = !nlml; fp.R_raw = mix;
      fp.w_old = 1.0 - fp.ep_damp; fp.w_new = mix ? fp.ep_damp / o.ep_fraction : fp.ep_damp; fp.mom_alpha = mix ? o.ep_fraction : 1.0;
      fp.predict_k1 = (!nlml && o.predict_at_k1) ? 1 : 0;
      fp.store_PF = p->need_PF ? 1 : 0; fp.l_iter = 0;
      fp.k_begin = 0; fp.k_end = sh.T;
      if (!fp.mom_all && p->need_PF && sh.T > 1) {   // fixed sites for k < T-1: lean kernel, then the ADF step at k = T-1
        if (xs_pending) {
          // one launch per chunk of the previous sweep's smoother, each behind that chunk's apply pass and site refresh
          for (int c = (int)xs_ch.size() - 1; c >= 0; --c) {
            HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_chunk[c], 0));
            fp.k_begin = xs_ch[c].k0; fp.k_end = xs_ch[c].k0 + xs_ch[c].nk;
            RUN(launch_filter(p, fp));
          }
          HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_red, 0));      // (the lZ sum reads lZ[T-1], which the ADF step rewrites; the fixed-site launches in front of it, k_end < T, do not write lZ at all)
          xs_pending = false;
        } else {
          fp.k_end = sh.T - 1;
          RUN(launch_filter(p, fp));
        }
        fp.k_begin = sh.T - 1; fp.k_end = sh.T;
      }
      if (xs_pending) { HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_red, 0)); xs_pending = false; }
      RUN(launch_filter(p, fp));
      if (smooth) RUN(sweep_pump(p, sc));      // gain + compose of the finished chunks on the second stream while the filter runs
    }
    if (xs_pending) { HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_red, 0)); xs_pending = false; }
    if (itt == 1 && !nlml) RUN(reduce_sum(p, p->b.lZ, 0, sh.T, 0, nullptr, p->red_all));
    if (run_smoother && run_filter) {
      RUN(seed_last_step(p));
      const double ep_damp = (itt < I) ? p->damping[itt] : 0.0;
      const int ep_clamp = (nlml || mixture_rule(p)) ? 0 : 1, ep_wR = nlml ? 0 : 1;
      double* ep_lZ = mixture_rule(p) ? nullptr : p->b.lZ;   // the mixture variant leaves the clamp to the next filter pass (gf_ep_mods_nmf_mixture.m:195, 280-284)
      const EpRange ep_range = [&](int64_t lo, int64_t hi, hipStream_t st) { return launch_ep(p, o.ep_fraction, ep_damp, ep_clamp, ep_wR, ep_lZ, lo, hi, st); };
      if (smooth) RUN(sweep_finish(p, sc, itt < I ? &ep_range : nullptr));
      if (itt < I) {
        if (sc.xs) {
          hipStream_t st = p->s_apply[0];
          if (!nlml) RUN(reduce_sum(p, p->b.lZ, 0, sh.T, 0, st));
          HIP_TRY(hipEventRecord(p->ev_red, st));
          xs_pending = true; xs_ch = sc.ch;
        } else {
          RUN(ep_range(0, -1, p->stream));
          if (!nlml) RUN(reduce_sum(p, p->b.lZ, 0, sh.T, 0));
        }
      }
    }
  }
  if (xs_pending) HIP_TRY(hipStreamWaitEvent(p->stream, p->ev_red, 0));
  if (nlml) RUN(reduce_sum(p, p->b.lZ, 0, sh.T, 0, nullptr, p->red_all));
  std::vector<double> red((size_t)(I + 2) * RR);
  HIP_TRY(hipMemcpyAsync(red.data(), p->red_all, red.size() * sizeof(double), hipMemcpyDeviceToHost, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  for (int q = 0; q < B; ++q) {
    p->nlZ[(size_t)q * I] = -red[(size_t)q * 8];
    for (int itt = 1; itt <= I; ++itt) {
      const double* r = &red[(size_t)itt * RR + (size_t)q * 8];
      const bool ran = (!nlml || itt < I) && (!nlml || itt == 1 || itt < I);
      if (!ran) continue;
      if (itt < I && !nlml) p->nlZ[(size_t)q * I + itt] = -r[0];
      p->mdM[(size_t)q * I + itt - 1] = r[1];
      p->mdP[(size_t)q * I + itt - 1] = r[2];
    }
  }
  return NAGP_OK;
}

static int exec_giekf(nagp_plan* p) {
  const Shape& sh = p->sh; const nagp_opts& o = p->opts; const int I = o.ep_itts, B = p->B;
  std::vector<double> red;
  if (o.mode == NAGP_MODE_NLML) {
    // gf_giekf_modulator_nmf_constraints.m:385-472 with GradObj='off': ONE plain EKF pass (prediction at k=1 too, a single
    // update per step whatever l_iter says, no smoother), edata = sum of the per-step energies
    FilterPar fp{};
    fp.itt = 1; fp.store_PF = 0; fp.l_iter = 1; fp.predict_k1 = 1; fp.ekf_energy = 1;
    fp.k_begin = 0; fp.k_end = sh.T;
    RUN(launch_filter(p, fp));
    RUN(reduce_sum(p, p->b.lZ, 0, sh.T, 0));
    RUN(fetch_red(p, red));
    for (int q = 0; q < B; ++q) p->nlZ[(size_t)q * I] = -red[(size_t)q * 8];
    return NAGP_OK;
  }
  for (int itt = 1; itt <= I; ++itt) {
    FilterPar fp{};
    fp.itt = itt; fp.store_PF = 1; fp.l_iter = o.l_iter;
    fp.init_from_state = (itt > 1); fp.reset_P = (o.flags & NAGP_FLAG_EKF_RESET_P) ? 1 : 0;
    fp.k_begin = 0; fp.k_end = sh.T;
    SweepCtx sc;
    if (sh.T > 1) sweep_begin(p, sc, p->want_PS && itt == I);
    RUN(launch_filter(p, fp));
    if (sh.T > 1) RUN(sweep_pump(p, sc));
    RUN(zero_async(p, p->b.red, (size_t)B * 8 * sizeof(double)));
    RUN(seed_last_step(p));
    if (sh.T == 1) {   // no smoothing step: the restart state is the filtered one
      for (int q = 0; q < B; ++q) {
        double* st = p->b.state + (size_t)q * ((size_t)sh.ntiles * 16 + sh.S);
        std::vector<double> lo(pf_step_doubles(sh)), full((size_t)sh.ntiles * 16);
        HIP_TRY(hipMemcpyAsync(lo.data(), p->b.PF + (size_t)q * pf_step_doubles(sh), lo.size() * 8, hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (int Ib = 0; Ib < sh.M; ++Ib)
          for (int Jb = 0; Jb < sh.M; ++Jb)
            for (int i = 0; i < 4; ++i)
              for (int j = 0; j < 4; ++j)
                full[((size_t)Ib * sh.M + Jb) * 16 + 4 * i + j] = (Ib >= Jb) ? lo[pf_off(Ib * (Ib + 1) / 2 + Jb, 4 * i + j)]
                                                                            : lo[pf_off(Jb * (Jb + 1) / 2 + Ib, 4 * j + i)];
        HIP_TRY(hipMemcpyAsync(st, full.data(), full.size() * 8, hipMemcpyHostToDevice, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        HIP_TRY(hipMemcpyAsync(st + (size_t)sh.ntiles * 16, p->b.MF + (size_t)q * sh.S, sh.S * sizeof(double), hipMemcpyDeviceToDevice, p->stream));
      }
    }
    if (sh.T > 1) RUN(sweep_finish(p, sc));
    RUN(fetch_red(p, red));
    for (int q = 0; q < B; ++q) {
      p->mdM[(size_t)q * I + itt - 1] = red[(size_t)q * 8 + 1];
      p->mdP[(size_t)q * I + itt - 1] = red[(size_t)q * 8 + 2];
    }
  }
  return NAGP_OK;
}


static __global__ void ihgp_init_kernel(double* R, size_t n_per, const double* model, size_t msz, size_t sn2_off, int zero_R,
                                        double* vprev, const double* tab, size_t tab_sz, size_t hph0_off, int M) {
  const int q = blockIdx.y;
  const double v = zero_R ? 0.0 : model[(size_t)q * msz + sn2_off];
  double* r = R + (size_t)q * n_per;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_per; i += (size_t)gridDim.x * blockDim.x) r[i] = v;
  if (blockIdx.x == 0)
    for (int i = threadIdx.x; i < M; i += blockDim.x) vprev[(size_t)q * M + i] = tab[(size_t)q * tab_sz + hph0_off + i];
}

static int exec_ihgp(nagp_plan* p) {
  const Shape& sh = p->sh; const nagp_opts& o = p->opts; const int I = o.ep_itts, B = p->B;
  std::vector<double> red;
  const bool mix = mixture_rule(p);
  const bool cv = (o.flags & NAGP_FLAG_IHGP_CONSTRAINTS) != 0 || mix;   // the mixture variant also starts from R = 0 (:248)
  // R = exp(lik) .* ones (ihgp_ep_modulator_nmf.m:209) or zeros (constraints variant :243), problem-wise value; PSP of sweep 1 = Pinf
  // -> vprev = h^2 Pinf(c,c).  One launch for all problems (the values are read from the packed models and tables on the device).
  {
    const size_t n_per = (size_t)sh.T * sh.M;
    const unsigned gx = (unsigned)std::max<size_t>(1, std::min<size_t>(64, (n_per + 4095) / 4096));
    hipLaunchKernelGGL(ihgp_init_kernel, dim3(gx, B), dim3(256), 0, p->stream, p->b.R, n_per, p->d_model, mdl_size(sh), mdl_sn2(sh), cv ? 1 : 0,
                       p->d_vprev, p->d_tab, itab_size(sh, p->tb.NG), itab_hph0(sh, p->tb.NG), sh.M);
    HIP_TRY(hipGetLastError());
  }
  MomCfg mcf = p->mc; mcf.DG = p->DG_f; mcf.cache_tabs = p->cache_f; mcf.store_a = p->sta_f;
  if (p->src_f) mcf.src = p->src_all;
  if (dev_env("NAGP_STAMPS")) mcf.stamps = reinterpret_cast<unsigned long long*>(p->d_stamps);   // developer diagnostics
  auto affine = [&](int mode, int64_t kend, int itt) -> int {
    if (kend <= 0) return NAGP_OK;
    AffPar ap{};
    ap.mode = mode; ap.kend = kend; ap.L = p->aff_L; ap.ns = (int)((kend + ap.L - 1) / ap.L);
    ap.spanbuf = p->d_affspan; ap.bnd = p->d_affbnd; ap.vprev = p->d_vprev;
    const dim3 g((unsigned)((ap.ns * sh.M + 255) / 256), B), bl(256);
    Timed t(p, mode == 0 ? NAGP_K_FILTER_LIN : NAGP_K_SCAN);
    if (mode == 0) {
      hipLaunchKernelGGL((ihgp_aff_compose_kernel<0>), g, bl, 0, p->stream, sh, p->b, p->tb, ap);
      hipLaunchKernelGGL((ihgp_aff_boundary_kernel<0>), dim3(B), dim3(64), 0, p->stream, sh, p->b, ap, itt);
      hipLaunchKernelGGL((ihgp_aff_apply_kernel<0>), g, bl, 0, p->stream, sh, p->b, p->tb, ap);
    } else {
      hipLaunchKernelGGL((ihgp_aff_compose_kernel<1>), g, bl, 0, p->stream, sh, p->b, p->tb, ap);
      hipLaunchKernelGGL((ihgp_aff_boundary_kernel<1>), dim3(B), dim3(64), 0, p->stream, sh, p->b, ap, itt);
      hipLaunchKernelGGL((ihgp_aff_apply_kernel<1>), g, bl, 0, p->stream, sh, p->b, p->tb, ap);
    }
    HIP_TRY(hipGetLastError());
    return NAGP_OK;
  };
  for (int itt = 1; itt <= I; ++itt) {
    // forward: sweep 1 is the sequential ADF filter; later sweeps have fixed sites for k < T-1 (an affine
    // recursion, run parallel in time) and one ADF step at k = T-1
    if (itt > 1) RUN(affine(0, sh.T - 1, itt));
    IhgpPar ip{itt, p->damping[itt - 1], itt == 1 ? 1 : 0, (itt == 1) ? (int64_t)0 : (int64_t)(sh.T - 1)};
    ip.hph_lds = p->hph_lds; ip.kb = p->kb_ih;
    if (const char* e = dev_env("NAGP_STAMP_WORKER")) ip.dbg_wave = atoi(e);
    ip.w_old = 1.0 - ip.ep_damp; ip.w_new = mix ? ip.ep_damp / o.ep_fraction : ip.ep_damp; ip.mom_alpha = mix ? o.ep_fraction : 1.0;
    {
      Timed t(p, itt == 1 ? NAGP_K_FILTER : NAGP_K_FILTER_LIN);
#define LI(V) hipLaunchKernelGGL((ihgp_filter_kernel<V, false>), dim3(B), dim3(p->NT_ih), p->lds_ih, p->stream, sh, p->b, mcf, p->tb, ip)
#define LIS(V) hipLaunchKernelGGL((ihgp_filter_kernel<V, true>), dim3(B), dim3(p->NT_ih), p->lds_ih, p->stream, sh, p->b, mcf, p->tb, ip)
      if (p->sq_ih) {
        IhgpPar ia = ip; ia.hph_lds = p->hph_sq; ia.kb = p->kb_sq;
        MomSp sq{}; sq.c0 = p->sq_c0;
#define LQ(V) hipLaunchKernelGGL((ihgp_adf8sq_kernel<V>), dim3(B), dim3(MSQ_NT), p->lds_sq, p->stream, sh, p->b, mcf, sq, p->tb, ia)
        switch (mcf.cdim) { case 1: LQ(1); break; case 2: LQ(2); break; case 3: LQ(3); break; case 4: LQ(4); break; case 5: LQ(5); break; default: LQ(6); break; }
#undef LQ
      } else if (p->sp_ih) {
        IhgpPar ia = ip; ia.hph_lds = p->hph_sp; ia.kb = p->kb_sp;
#define LA(V) hipLaunchKernelGGL((ihgp_adf_kernel<V>), dim3(B), dim3(MSP_NT), p->lds_sp, p->stream, sh, p->b, mcf, p->sp, p->tb, ia)
#define LA8(V, PK) hipLaunchKernelGGL((ihgp_adf8_kernel<V, PK>), dim3(B), dim3(MSR_NT), p->lds_sp8, p->stream, sh, p->b, mcf, p->sp, p->tb, ia)
        if (p->sp_ih8 && p->sp_pack) switch (mcf.cdim) { case 1: LA8(1, true); break; case 2: LA8(2, true); break; case 3: LA8(3, true); break; case 4: LA8(4, true); break; case 5: LA8(5, true); break; default: LA8(6, true); break; }
        else if (p->sp_ih8) switch (mcf.cdim) { case 1: LA8(1, false); break; case 2: LA8(2, false); break; case 3: LA8(3, false); break; case 4: LA8(4, false); break; case 5: LA8(5, false); break; case 6: LA8(6, false); break; default: LA8(7, false); break; }
        else switch (mcf.cdim) { case 1: LA(1); break; case 2: LA(2); break; case 3: LA(3); break; case 4: LA(4); break; case 5: LA(5); break; case 6: LA(6); break; default: LA(7); break; }
#undef LA
#undef LA8
      } else if (p->src_f) { NAGP_MV_SWITCH9(mom_variant(mcf), LIS) } else { NAGP_MV_SWITCH9(mom_variant(mcf), LI) }
#undef LI
#undef LIS
    }
    HIP_TRY(hipGetLastError());
    RUN(reduce_sum(p, p->b.lZ, itt == 1 ? 0 : sh.T - 1, sh.T, 0));
    RUN(seed_last_step(p));
    // backward mean recursion (parallel in time); red[1], red[2] = maxDiffM, maxDiffP
    hipLaunchKernelGGL(zero_maxdiff_kernel, dim3((B + 255) / 256), dim3(256), 0, p->stream, p->b.red, B);
    if (sh.T > 1) RUN(affine(1, sh.T - 1, itt));
    else {   // no smoothing step: P = zeros (ihgp_ep_modulator_nmf.m:364) -> maxDiffP = |H PSP H'|
      Timed t(p, NAGP_K_SCAN);
      hipLaunchKernelGGL(ihgp_scan_kernel, dim3(B), dim3(64), 0, p->stream, sh, p->b, p->tb, p->d_vprev);
    }
    if (itt < I) {
      RUN(zero_async(p, p->d_lZs, (size_t)B * sh.T * sizeof(double)));
      RUN(launch_ep(p, o.ep_fraction, p->damping[itt], 0, 2, p->d_lZs));
      RUN(reduce_sum(p, p->d_lZs, 0, sh.T, 3));
    }
    RUN(fetch_red(p, red));
    for (int q = 0; q < B; ++q) {
      const double sumF = red[(size_t)q * 8], sumS = red[(size_t)q * 8 + 3];
      if (itt == 1) p->nlZ[(size_t)q * I] = -sumF;
      if (itt < I) p->nlZ[(size_t)q * I + itt] = -(sumF + (itt > 1 ? sumS : 0.0));
      p->mdM[(size_t)q * I + itt - 1] = red[(size_t)q * 8 + 1];
      p->mdP[(size_t)q * I + itt - 1] = red[(size_t)q * 8 + 2];
    }
  }
  return NAGP_OK;
}

extern "C" int nagp_plan_execute(nagp_plan* p) {
  if (!p) FAIL(NAGP_EINVAL, "null plan");
  HIP_TRY(hipSetDevice(p->opts.device));
  const Shape& sh = p->sh; const size_t BT = (size_t)p->B * sh.T;
  p->evs.clear(); p->ev_next = 0;
  HIP_TRY(hipEventRecord(p->ev_t0, p->stream));
  // every call starts from the reference's initial state (sites zero, MS zero, ...)
  if (p->warm) {   // warm start: the sites a previous call returned instead of zeros
    HIP_TRY(hipMemcpyAsync(p->b.ttau, p->d_tt0, BT * sh.M * 8, hipMemcpyDeviceToDevice, p->stream));
    HIP_TRY(hipMemcpyAsync(p->b.tnu, p->d_tn0, BT * sh.M * 8, hipMemcpyDeviceToDevice, p->stream));
  } else {
    RUN(zero_async(p, p->b.ttau, BT * sh.M * 8)); RUN(zero_async(p, p->b.tnu, BT * sh.M * 8));
  }
  RUN(zero_async(p, p->b.R, BT * sh.M * 8)); RUN(zero_async(p, p->b.lZ, BT * 8));
  RUN(zero_async(p, p->b.sm, BT * sh.M * 8)); RUN(zero_async(p, p->b.sv, BT * sh.M * 8));
  RUN(zero_async(p, p->b.MS, BT * sh.S * 8)); RUN(zero_async(p, p->b.red, (size_t)p->B * 64));
  RUN(zero_async(p, p->b.counters, (size_t)p->B * 32));
  RUN(zero_async(p, p->b.state, (size_t)p->B * ((size_t)sh.ntiles * 16 + sh.S) * 8));
  if (p->d_stamps) RUN(zero_async(p, p->d_stamps, 24 * 8));
  if (p->d_gstamps) RUN(zero_async(p, p->d_gstamps, 32 * 8));
  std::fill(p->nlZ.begin(), p->nlZ.end(), 0.0);
  std::fill(p->mdM.begin(), p->mdM.end(), 0.0);
  std::fill(p->mdP.begin(), p->mdP.end(), 0.0);
  int st;
  switch (p->opts.kind) {
    case NAGP_KIND_GF_EP: st = exec_gf(p); break;
    case NAGP_KIND_IHGP: st = exec_ihgp(p); break;
    default: st = exec_giekf(p); break;
  }
  if (st != NAGP_OK) {   // leave no launch of the failed call behind: the next upload / execute / destroy starts from idle streams
    const std::string keep = g_last_error;
    if (p->stream) (void)hipStreamSynchronize(p->stream);
    if (p->stream2) (void)hipStreamSynchronize(p->stream2);
    for (hipStream_t s2 : p->s_apply) (void)hipStreamSynchronize(s2);
    (void)hipGetLastError();
    g_last_error = keep;
    return st;
  }
  HIP_TRY(hipEventRecord(p->ev_t1, p->stream));
  HIP_TRY(hipStreamSynchronize(p->stream));
  if (dev_env("NAGP_STAMPS") && p->d_gstamps && p->opts.kind != NAGP_KIND_IHGP) {
    unsigned long long g[32];
    if (hipMemcpy(g, p->d_gstamps, sizeof g, hipMemcpyDeviceToHost) == hipSuccess && p->gain_mfma && g[12]) {
      static const char* nm[12] = {"staging", "prologue barriers", "B' | delta_k", "PSkp", "Delta | tile 0", "trailing | 4 products", "factor+invert", "forward row", "interval barrier", "retry check", "backward", "G store"};
      for (int r = 0; r < 2; ++r) {
        if (!g[16 * r + 12]) continue;
        fprintf(stderr, "[nagp stamps] rts_gain_mfma_kernel, %s wave, cycles per workgroup (%llu sampled):", r ? "chain" : "column", g[16 * r + 12]);
        for (int q = 0; q < 12; ++q) fprintf(stderr, " %s %llu |", nm[q], g[16 * r + q] / g[16 * r + 12]);
        fprintf(stderr, "\n");
      }
    } else if (!p->gain_mfma && g[6])
      fprintf(stderr, "[nagp stamps] rts_gain_kernel, cycles per workgroup (thread 0 of %llu sampled): prologue %llu | diagonal tiles %llu | column solves %llu | trailing updates %llu | backward solve %llu | G store %llu\n",
              g[6], g[0] / g[6], g[1] / g[6], g[2] / g[6], g[3] / g[6], g[4] / g[6], g[5] / g[6]);
  }
  if (dev_env("NAGP_STAMPS") && p->d_stamps) {
    unsigned long long st[24];
    if (hipMemcpy(st, p->d_stamps, sizeof st, hipMemcpyDeviceToHost) == hipSuccess) {
      if (p->opts.kind == NAGP_KIND_IHGP)
        for (int w = 0; w < 2; ++w)
          fprintf(stderr, "[nagp stamps] %s: wait at B1 %llu | Q/v %llu | B2..B3 %llu | weights %llu | wait at B4 %llu | marginal sums %llu | MFMA steps %llu | wait at B5 %llu\n",
                  w ? "last worker wave " : "first worker wave", st[8 + 8 * w], st[9 + 8 * w], st[10 + 8 * w], st[11 + 8 * w], st[12 + 8 * w], st[13 + 8 * w], st[14 + 8 * w], st[15 + 8 * w]);
    }
    if (p->opts.kind == NAGP_KIND_GF_EP && hipMemcpy(st, p->d_stamps, 64, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[nagp stamps] fixed-site step (thread 0; the ADF launches add their cubature stamps to the same slots): loop top + mean prediction %llu | congruence + panel %llu | wait at B1 %llu | mean update %llu | rank-M update %llu | outputs, PF stores, B5 %llu\n", st[4], st[5], st[0], st[1], st[6], st[7]);
    if (p->opts.kind == NAGP_KIND_GIEKF && hipMemcpy(st, p->d_stamps, 64, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[nagp stamps] EKF step: loop top + mean prediction %llu | congruence + panel of wave 0 %llu | wait at B1 %llu | Jacobian partials %llu | P J' %llu | wave sums, gain, mean %llu | P -= K S K' %llu | outputs, PF stores, B5 %llu\n", st[4], st[5], st[0], st[1], st[2], st[3], st[6], st[7]);
    else if (hipMemcpy(st, p->d_stamps, 64, hipMemcpyDeviceToHost) == hipSuccess)
      fprintf(stderr, "[nagp stamps] mom: p1a %llu p1b %llu p2 %llu p3 %llu | pre-mom %llu post-mom %llu | aux %llu %llu  (sparse-point IHGP sweep: p1a..p3 = A, B+1b, 2, wait at B1 ; pre..aux = reduce+outputs, site+state+ring, look-up, A m)\n", st[0], st[1], st[2], st[3], st[4], st[5], st[6], st[7]);
  }

  HIP_TRY(hipGetLastError());
  memset(&p->tim, 0, sizeof p->tim);
  for (const EvRec& e : p->evs) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, e.a, e.b) == hipSuccess) { p->tim.ms[e.kid] += ms; p->tim.launches[e.kid] += 1; }
  }
  float tot = 0.f;
  (void)hipEventElapsedTime(&tot, p->ev_t0, p->ev_t1);
  p->tim.total_ms = tot;
  if (p->opts.kind != NAGP_KIND_IHGP) {
    // a step whose PSkp failed the Cholesky even with the jitter: the reference stops there (chol throws inside the catch block,
    // gf_ep_modulator_nmf.m:219-222).  The sweeps have run to the end (the outputs can be downloaded and will hold NaN).
    std::vector<unsigned long long> c((size_t)p->B * 4);
    HIP_TRY(hipMemcpy(c.data(), p->b.counters, c.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    for (int q = 0; q < p->B; ++q)
      if (c[(size_t)q * 4 + NAGP_CNT_NOTPD])
        FAIL(NAGP_ENOTPD, "problem %d: A*PS_k*A'+Q not positive definite at %llu smoother step(s) even with the jitter of the retry", q, c[(size_t)q * 4 + NAGP_CNT_NOTPD]);
  }
  return NAGP_OK;
}

extern "C" int nagp_plan_timings(const nagp_plan* p, nagp_timings* t) {
  if (!p || !t) FAIL(NAGP_EINVAL, "null argument");
  *t = p->tim;
  return NAGP_OK;
}

extern "C" int nagp_plan_download(nagp_plan* p, nagp_out* outs) {
  if (!p || !outs) FAIL(NAGP_EINVAL, "null argument");
  HIP_TRY(hipSetDevice(p->opts.device));
  const Shape& sh = p->sh; const int64_t T = sh.T; const int M = sh.M, S = sh.S, I = p->opts.ep_itts;
  const bool ihgp = p->opts.kind == NAGP_KIND_IHGP;
  std::vector<double> tmp;
  for (int q = 0; q < p->B; ++q) {
    nagp_out& o = outs[q];
    const size_t oM = (size_t)q * T * M, oS = (size_t)q * T * S;
#define D2H(dst, src, n) do { if (dst) HIP_TRY(hipMemcpyAsync(dst, src, (size_t)(n) * sizeof(double), hipMemcpyDeviceToHost, p->stream)); } while (0)
    D2H(o.Eft, p->b.sm + oM, T * M);
    D2H(o.MS, p->b.MS + oS, T * S);
    D2H(o.MF, p->b.MF + oS, T * S);
    D2H(o.ttau, p->b.ttau + oM, T * M);
    D2H(o.tnu, p->b.tnu + oM, T * M);
    D2H(o.R, p->b.R + oM, T * M);
    D2H(o.lZ, p->b.lZ + (size_t)q * T, T);
    if (!ihgp) D2H(o.Varft, p->b.sv + oM, T * M);
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (ihgp && o.Varft) {
      // Varft = repmat(diag(H*P*H')) with the blocks last looked up (k = 0); abs() unless constraints variant
      std::vector<double> v0(M, 0.0);
      if (T > 1) { HIP_TRY(hipMemcpy(v0.data(), p->b.sv + oM, M * sizeof(double), hipMemcpyDeviceToHost)); }
      const bool cv = (p->opts.flags & (NAGP_FLAG_IHGP_CONSTRAINTS | NAGP_FLAG_MIXTURE_RULE)) != 0;   // neither takes abs(Varft)
      for (int64_t k = 0; k < T; ++k)
        for (int n = 0; n < M; ++n) o.Varft[(size_t)k * M + n] = cv ? v0[n] : std::fabs(v0[n]);
    }
    if (o.PS) {
      if (ihgp || !p->want_PS) FAIL(NAGP_EINVAL, "PS requested but the plan was created without NAGP flag 0x4 (or IHGP)");
      const size_t tl = (size_t)sh.ntiles * 16;
      const int64_t KB = 256;
      tmp.resize((size_t)KB * tl);
      for (int64_t k0 = 0; k0 < T; k0 += KB) {
        const int64_t nk = std::min<int64_t>(KB, T - k0);
        // smoothed tiles for k < T-1; the last step is the filtered one
        const int64_t nsm = std::min<int64_t>(nk, std::max<int64_t>(0, (T - 1) - k0));
        if (nsm > 0) HIP_TRY(hipMemcpy(tmp.data(), p->b.PSs + ((size_t)q * T + k0) * tl, (size_t)nsm * tl * 8, hipMemcpyDeviceToHost));
        if (nsm < nk) {
          std::vector<double> lo(pf_step_doubles(sh));
          HIP_TRY(hipMemcpy(lo.data(), p->b.PF + ((size_t)q * T + (T - 1)) * pf_step_doubles(sh), lo.size() * 8, hipMemcpyDeviceToHost));
          double* full = tmp.data() + (size_t)nsm * tl;
          for (int Ib = 0; Ib < M; ++Ib)
            for (int Jb = 0; Jb < M; ++Jb)
              for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j)
                  full[((size_t)Ib * M + Jb) * 16 + 4 * i + j] = (Ib >= Jb) ? lo[pf_off(Ib * (Ib + 1) / 2 + Jb, 4 * i + j)]
                                                                           : lo[pf_off(Jb * (Jb + 1) / 2 + Ib, 4 * j + i)];
        }
        for (int64_t kk = 0; kk < nk; ++kk) {
          double* dst = o.PS + (size_t)(k0 + kk) * S * S;
          const double* src = tmp.data() + (size_t)kk * tl;
          for (int Ib = 0; Ib < M; ++Ib)
            for (int Jb = 0; Jb < M; ++Jb) {
              const double* t16 = src + ((size_t)Ib * M + Jb) * 16;
              for (int i = 0; i < sh.bsz[Ib]; ++i)
                for (int j = 0; j < sh.bsz[Jb]; ++j) dst[(size_t)(sh.off[Ib] + i) + (size_t)S * (sh.off[Jb] + j)] = t16[4 * i + j];
            }
        }
      }
    }
    if (o.nlZ) for (int i = 0; i < I; ++i) o.nlZ[i] = p->nlZ[(size_t)q * I + i];
    if (o.maxDiffM) for (int i = 0; i < I; ++i) o.maxDiffM[i] = p->mdM[(size_t)q * I + i];
    if (o.maxDiffP) for (int i = 0; i < I; ++i) o.maxDiffP[i] = p->mdP[(size_t)q * I + i];
    if (o.counters) {
      unsigned long long c[4];
      HIP_TRY(hipMemcpy(c, p->b.counters + (size_t)q * 4, sizeof c, hipMemcpyDeviceToHost));
      for (int i = 0; i < 4; ++i) o.counters[i] = (int64_t)c[i];
    }
  }
  return NAGP_OK;
}

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
  const size_t lds = fb_lds_doubles(S) * sizeof(double);
  if (S > 256 || lds > 160 * 1024) FAIL(NAGP_EUNSUPPORTED, "S=%d: the two constant S x S matrices do not fit the LDS (S <= 96)", S);
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
             ns > 1 ? dev + o_st : nullptr};
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
