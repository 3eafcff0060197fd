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
  std::vector<int> perm;   // plans with split blocks: device state index -> the caller's state index (empty otherwise)
  int Mu = 0;              // ... and the caller's number of sites (sh.Ms)
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

// ---- the rest of this translation unit, in three parts (one object file: the parts share the plan struct and the static helpers above)
#include "nagp_api_plan.hpp"
#include "nagp_api_sweep.hpp"
#include "nagp_api_entry.hpp"
