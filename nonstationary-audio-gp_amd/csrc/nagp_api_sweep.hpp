// nagp_api_sweep.hpp -- part of the ONE translation unit nagp_api.hip (included there, in this order: nagp_api_plan.hpp, nagp_api_sweep.hpp,
// nagp_api_entry.hpp; the plan struct, the error helpers and the developer-switch accessor live in nagp_api.hip itself).
// The sweep scheduler: filter launches, the chunk-pipelined smoother (gain, compose, boundary, apply), site refresh, the three execute loops, download.

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
  if (p->sh.Ms < p->sh.M) {      // split blocks: one geometry (the fixed-site one) for every launch, the general mom code
    fp.cpl_doubles = (int)filter_cpl_doubles(p->sh); fp.cpl_chunk = filter_cpl_chunk(p->sh);
    mc.sp = MomSp{};
#define LFC(TP, ME, V) hipLaunchKernelGGL((gf_filter_kernel<TP, ME, V, 512, 0, true>), g, dim3(ekf ? nt_ekf : p->NT_f), p->lds_filter, p->stream, p->sh, p->b, mc, fp)
#define LFC1(V) LFC(1, 0, V)
#define LFC2(V) LFC(2, 0, V)
#define LFC4(V) LFC(4, 0, V)
    if (ekf) switch (p->TPT_f) { case 1: LFC(1, 1, 0); break; case 2: LFC(2, 1, 0); break; default: LFC(4, 1, 0); break; }
    else if (adf) switch (p->TPT_f) { case 1: NAGP_MV_SWITCH(mom_variant(mc), LFC1) break; case 2: NAGP_MV_SWITCH(mom_variant(mc), LFC2) break; default: NAGP_MV_SWITCH(mom_variant(mc), LFC4) break; }
    else if (p->wide_l && p->NT_l <= 768) hipLaunchKernelGGL((gf_filter_kernel<1, 0, -1, 768, 0, true>), g, dim3(p->NT_l), p->lds_filter, p->stream, p->sh, p->b, mc, fp);
    else if (p->wide_l) hipLaunchKernelGGL((gf_filter_kernel<1, 0, -1, 1024, 0, true>), g, dim3(p->NT_l), p->lds_filter, p->stream, p->sh, p->b, mc, fp);
    else switch (p->TPT_f) { case 1: LFC1(-1); break; case 2: LFC2(-1); break; default: LFC4(-1); break; }
#undef LFC
#undef LFC1
#undef LFC2
#undef LFC4
  } else
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
      const int per_cu = std::max(1, (int)((160 * 1024) / std::max<size_t>(p->lds_mfma, 1)));      // workgroups of the column-owner passes a CU holds (LDS)
      // resident workgroups a launch can count on: the CUs the filter's workgroups do not hold (one workgroup per CU), or -- where two fit a CU --
      // two on every CU (measured at 128 segments: 243 -> 180 ms against 225 with 384; the filter launches are short beside the passes there)
      int n_cu = (per_cu >= 2) ? 256 * per_cu : std::max(32, 256 - p->B);
      if (dev_env("NAGP_BIG_NCU")) n_cu = std::max(1, atoi(dev_env("NAGP_BIG_NCU")));      // developer switch
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
  if (sh.Ms < sh.M) {      // split blocks
    gp.cpl_doubles = (int)gain_cpl_doubles(sh);
    switch (p->TPT) {
      case 1: hipLaunchKernelGGL((rts_gain_kernel<1, 512, true>), gr, bl, p->lds_gain, st, sh, b, gp); break;
      case 2: hipLaunchKernelGGL((rts_gain_kernel<2, 512, true>), gr, bl, p->lds_gain, st, sh, b, gp); break;
      case 3: hipLaunchKernelGGL((rts_gain_kernel<3, 512, true>), gr, bl, p->lds_gain, st, sh, b, gp); break;
      case 4: hipLaunchKernelGGL((rts_gain_kernel<4, 512, true>), gr, bl, p->lds_gain, st, sh, b, gp); break;
      default: hipLaunchKernelGGL((rts_gain_kernel<8, 512, true>), gr, bl, p->lds_gain, st, sh, b, gp); break;
    }
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
    switch (ntl) { case 5: LB(5); break; case 6: LB(6); break; case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
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
    switch (ntl) { case 5: LB(5); break; case 6: LB(6); break; case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
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
    switch (ntl) { case 5: LB(5); break; case 6: LB(6); break; case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
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
    switch (ntl) { case 5: LB(5); break; case 6: LB(6); break; case 7: LB(7); break; case 8: LB(8); break; case 9: LB(9); break; default: LB(10); break; }
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
  // (sized from the steps of THIS launch: in the cross-sweep form a launch covers one smoother chunk)
  ep.steps_per_wg = (int)std::max<int64_t>(1, ((int64_t)p->B * (ep.k_end - ep.k_begin) + 8191) / 8192);
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
#define LAFF(MO, BSV) do { hipLaunchKernelGGL((ihgp_aff_compose_kernel<MO, BSV>), g, bl, 0, p->stream, sh, p->b, p->tb, ap); \
                           hipLaunchKernelGGL((ihgp_aff_boundary_kernel<MO, BSV>), dim3(B), dim3(64), 0, p->stream, sh, p->b, ap, itt); \
                           hipLaunchKernelGGL((ihgp_aff_apply_kernel<MO, BSV>), g, bl, 0, p->stream, sh, p->b, p->tb, ap); } while (0)
    if (sh.BS == 8) { if (mode == 0) LAFF(0, 8); else LAFF(1, 8); }      // (blocks of 5 .. 8 states: 8 x 8 maps per thread)
    else { if (mode == 0) LAFF(0, 4); else LAFF(1, 4); }
#undef LAFF
    HIP_TRY(hipGetLastError());
    return NAGP_OK;
  };
  for (int itt = 1; itt <= I; ++itt) {
    // forward: sweep 1 is the sequential ADF filter; later sweeps have fixed sites for k < T-1 (an affine
    // recursion, run parallel in time) and one ADF step at k = T-1
    const bool seq8 = sh.BS == 8;     // blocks of 5 .. 8 states: the general ADF kernel at block stride 8
    const bool seq = dev_env("NAGP_IH_SEQ") && !p->sq_ih && !p->sp_ih && !p->src_f;      // developer switch: the sequential kernels (general ADF filter, ihgp_scan_kernel) for every sweep instead of the affine scans
    if (itt > 1 && !seq) RUN(affine(0, sh.T - 1, itt));
    IhgpPar ip{itt, p->damping[itt - 1], itt == 1 ? 1 : 0, (itt == 1 || seq) ? (int64_t)0 : (int64_t)(sh.T - 1)};
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
      } else if (seq8) {
#define LI8(V) hipLaunchKernelGGL((ihgp_filter_kernel<V, false, 8>), dim3(B), dim3(p->NT_ih), p->lds_ih, p->stream, sh, p->b, mcf, p->tb, ip)
        NAGP_MV_SWITCH9(mom_variant(mcf), LI8)
#undef LI8
      } else if (p->src_f) { NAGP_MV_SWITCH9(mom_variant(mcf), LIS) } else { NAGP_MV_SWITCH9(mom_variant(mcf), LI) }
#undef LI
#undef LIS
    }
    HIP_TRY(hipGetLastError());
    RUN(reduce_sum(p, p->b.lZ, itt == 1 ? 0 : sh.T - 1, sh.T, 0));
    RUN(seed_last_step(p));
    // backward mean recursion (parallel in time); red[1], red[2] = maxDiffM, maxDiffP
    hipLaunchKernelGGL(zero_maxdiff_kernel, dim3((B + 255) / 256), dim3(256), 0, p->stream, p->b.red, B);
    if (sh.T > 1 && !seq) RUN(affine(1, sh.T - 1, itt));
    else {   // T = 1, no smoothing step: P = zeros (ihgp_ep_modulator_nmf.m:364) -> maxDiffP = |H PSP H'|   (or a plan of the sequential kernels)
      Timed t(p, NAGP_K_SCAN);
      if (seq8) hipLaunchKernelGGL(ihgp_scan_kernel<8>, dim3(B), dim3(64), 0, p->stream, sh, p->b, p->tb, p->d_vprev);
      else hipLaunchKernelGGL(ihgp_scan_kernel<4>, dim3(B), dim3(64), 0, p->stream, sh, p->b, p->tb, p->d_vprev);
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
    if (!p->perm.empty()) {
      // split blocks: the device keeps Md = M tile rows of which the caller's Mu sites are the first, and numbers the states tile row by
      // tile row (perm: device index -> the caller's)
      const int Mu = p->Mu;
      std::vector<double> w;
      auto sites = [&](double* dst, const double* src) -> int {
        if (!dst) return NAGP_OK;
        w.resize((size_t)T * M);
        HIP_TRY(hipMemcpy(w.data(), src, w.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < T; ++k)
          for (int i = 0; i < Mu; ++i) dst[(size_t)k * Mu + i] = w[(size_t)k * M + i];
        return NAGP_OK;
      };
      auto states = [&](double* dst, const double* src) -> int {
        if (!dst) return NAGP_OK;
        w.resize((size_t)T * S);
        HIP_TRY(hipMemcpy(w.data(), src, w.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int64_t k = 0; k < T; ++k)
          for (int i = 0; i < S; ++i) dst[(size_t)k * S + p->perm[i]] = w[(size_t)k * S + i];
        return NAGP_OK;
      };
      HIP_TRY(hipStreamSynchronize(p->stream));
      RUN(sites(o.Eft, p->b.sm + oM)); RUN(sites(o.Varft, p->b.sv + oM)); RUN(sites(o.ttau, p->b.ttau + oM)); RUN(sites(o.tnu, p->b.tnu + oM)); RUN(sites(o.R, p->b.R + oM));
      RUN(states(o.MS, p->b.MS + oS)); RUN(states(o.MF, p->b.MF + oS));
      if (o.lZ) HIP_TRY(hipMemcpy(o.lZ, p->b.lZ + (size_t)q * T, (size_t)T * sizeof(double), hipMemcpyDeviceToHost));
    } else {
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
    }
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
                for (int j = 0; j < sh.bsz[Jb]; ++j) {
                  const int r = sh.off[Ib] + i, c = sh.off[Jb] + j;
                  if (p->perm.empty()) dst[(size_t)r + (size_t)S * c] = t16[4 * i + j];
                  else dst[(size_t)p->perm[r] + (size_t)S * p->perm[c]] = t16[4 * i + j];
                }
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

