// nagp_api_plan.hpp -- part of the ONE translation unit nagp_api.hip (included there, in this order: nagp_api_plan.hpp, nagp_api_sweep.hpp,
// nagp_api_entry.hpp; the plan struct, the error helpers and the developer-switch accessor live in nagp_api.hip itself).
// Plan creation: shape checks, kernel / LDS / slot policy, model packing, cubature tables, device buffers; destroy, upload.

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
  sh.BS = 4; sh.Ms = m0.M;
  for (int n = 0; n < MAXM; ++n) sh.part[n] = -1;
  const bool ih_kind = (o->kind == NAGP_KIND_IHGP);
  for (int n = 0; n < m0.M; ++n) {
    sh.bsz[n] = sh.off[n + 1] - sh.off[n];
    // blocks of 5 .. 8 states (Matern-5/2 and -7/2 sub-bands): the infinite-horizon plans keep them whole (BS = 8)
    // (and the full-covariance plans split them over two tile rows, below)
    if (sh.bsz[n] < 1 || sh.bsz[n] > 8) { const int bsn = sh.bsz[n]; delete p; FAIL(NAGP_EUNSUPPORTED, "block %d has size %d (supported: 1..8)", n, bsn); }
    if (sh.bsz[n] > 4 && ih_kind) sh.BS = 8;
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
  // ---- full-covariance plans with blocks of 5 .. 8 states: the DEVICE VIEW of the models.  Block n keeps its first four states in tile
  // row n; the rest become tile row Mu + e behind the Mu real sites (Shape::part pairs the two).  The states are renumbered tile row by
  // tile row (perm: device index -> caller's index); A, Q, Pinf are permuted accordingly, the tail rows get h = 0, and everything below
  // this point -- packing, buffers, kernels -- sees Md = Mu + E "sites" of at most four states whose last E never carry a measurement.
  std::vector<nagp_model> dmods;
  std::vector<std::vector<double>> dstore;
  std::vector<int32_t> doff;
  {
    int E = 0;
    for (int n = 0; n < m0.M; ++n) if (sh.bsz[n] > 4) ++E;
    if (!ih_kind && E > 0) {
      const int Mu = m0.M, Md = Mu + E, S = m0.S;
      if (Md > MAXM) { delete p; FAIL(NAGP_EUNSUPPORTED, "M=%d sites with %d blocks of more than four states: more than %d tile rows", Mu, E, MAXM); }
      std::vector<int> start(Md), size(Md);
      int e = 0;
      for (int n = 0; n < Mu; ++n) {
        const int bs = sh.bsz[n];
        start[n] = sh.off[n]; size[n] = std::min(bs, 4);
        if (bs > 4) { const int t = Mu + e++; start[t] = sh.off[n] + 4; size[t] = bs - 4; sh.part[n] = (signed char)t; sh.part[t] = (signed char)n; }
      }
      p->perm.resize(S); p->Mu = Mu;
      doff.resize(Md + 1);
      int pos = 0;
      for (int r = 0; r < Md; ++r) { doff[r] = pos; for (int i = 0; i < size[r]; ++i) p->perm[pos++] = start[r] + i; }
      doff[Md] = pos;
      sh.M = Md; sh.Ms = Mu; sh.ntiles = Md * Md;
      for (int r = 0; r <= Md; ++r) sh.off[r] = doff[r];
      for (int r = 0; r < Md; ++r) sh.bsz[r] = size[r];
      dmods.assign(models, models + B);
      dstore.resize((size_t)B * 4);
      for (int q = 0; q < B; ++q) {
        const nagp_model& mq = models[q];
        auto permuted = [&](const double* X, std::vector<double>& out) {
          out.resize((size_t)S * S);
          for (int j = 0; j < S; ++j)
            for (int i = 0; i < S; ++i) out[(size_t)i + (size_t)S * j] = X[(size_t)p->perm[i] + (size_t)S * p->perm[j]];
        };
        permuted(mq.A, dstore[(size_t)q * 4 + 0]); permuted(mq.Q, dstore[(size_t)q * 4 + 1]); permuted(mq.Pinf, dstore[(size_t)q * 4 + 2]);
        std::vector<double>& hv = dstore[(size_t)q * 4 + 3];
        hv.assign(Md, 0.0);
        for (int n = 0; n < Mu; ++n) hv[n] = mq.h_val[n];
        dmods[q].M = Md; dmods[q].block_offsets = doff.data();
        dmods[q].A = dstore[(size_t)q * 4 + 0].data(); dmods[q].Q = dstore[(size_t)q * 4 + 1].data(); dmods[q].Pinf = dstore[(size_t)q * 4 + 2].data();
        dmods[q].h_val = hv.data();
      }
      models = dmods.data();
    }
  }
  const bool split = sh.Ms < sh.M;
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
    // split blocks: every filter launch has the geometry of the fixed-site one (gf_filter_kernel<TPT_f, ., ., 512, 0, true>)
    // (fixed-site launches of 513 .. 1024 lower tiles: one tile per thread under the 768- / 1024-thread bound, as for unsplit models)
    if (split) { p->TPT_a = p->TPT_f; p->NT_a = p->NT_f; p->LB_a = 512; p->NT_l = p->wide_l ? roundup64(slots) : p->NT_f; p->NT_fl = p->NT_f; }
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
    const int S = sh.S, BSr = sh.BS, BS2 = sh.BS * sh.BS;
    for (int n = 0; n < sh.M; ++n) {
      const int o0 = sh.off[n], bs = sh.bsz[n];
      for (int i = 0; i < bs; ++i)
        for (int j = 0; j < bs; ++j) {
          const size_t src = (size_t)(o0 + i) + (size_t)S * (o0 + j);   // column-major
          d[mdl_A(sh) + (size_t)n * BS2 + BSr * i + j] = mq.A[src];
          d[mdl_Q(sh) + (size_t)n * BS2 + BSr * i + j] = mq.Q[src];
          d[mdl_P(sh) + (size_t)n * BS2 + BSr * i + j] = mq.Pinf[src];
        }
      d[mdl_h(sh) + n] = mq.h_val[n];
      p->h_hval[(size_t)q * sh.M + n] = mq.h_val[n];
      if (split && sh.part[n] >= 0) {      // cross tiles of the pair: X(rows of n, columns of part[n])
        const int pn = sh.part[n], op = sh.off[pn], bp = sh.bsz[pn];
        for (int i = 0; i < bs; ++i)
          for (int j = 0; j < bp; ++j) {
            const size_t src = (size_t)(o0 + i) + (size_t)S * (op + j);
            d[mdl_Ax(sh) + (size_t)n * 16 + 4 * i + j] = mq.A[src];
            d[mdl_Qx(sh) + (size_t)n * 16 + 4 * i + j] = mq.Q[src];
            d[mdl_Px(sh) + (size_t)n * 16 + 4 * i + j] = mq.Pinf[src];
          }
      }
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
      // Sp >= 80 (five tile columns and more): the column-owner kernels of nagp_mfma_big.hpp -- one wave per tile column, the symmetric state
      // in LDS as its lower tiles, G streamed, two workgroups per CU at Sp = 80 / 96; below that the four-wave kernels with G and the state
      // side by side in LDS.  (Round 5: at Sp = 80 the column-owner passes take cfg2_batch from 508 to 444 ms, span passes 428 -> 235 ms, once
      // their span count knows that a CU holds two of their workgroups; NAGP_BIG_MIN_SP=97 restores the round-4 split.)
      const int big_min = dev_env("NAGP_BIG_MIN_SP") ? std::max(80, atoi(dev_env("NAGP_BIG_MIN_SP"))) : 80;
      if (Sp < big_min && Sp <= 96 && !dev_env("NAGP_NO_MFMA")) p->mfma_sp = Sp;
      // Sp <= 160: state and G do not fit LDS side by side (or, from Sp = 80, are better off apart); column-owner kernels
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
    PLAN_TRY(dalloc(p, &p->d_affspan, (size_t)B * p->aff_ns * sh.M * (sh.BS * sh.BS + sh.BS), false));
    PLAN_TRY(dalloc(p, &p->d_affbnd, (size_t)B * p->aff_ns * sh.M * sh.BS, false));
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
          for (int i = 0; i < bs; ++i) d[itab_wcol(sh, NG) + ((size_t)n * NG + g) * sh.BS + i] = h * ppr[i];
          const double* pgr = pg + (size_t)g * 2 * bs * bs;      // [PS2(:)' G(:)']
          d[itab_v(sh, NG) + (size_t)n * NG + g] = h * h * pgr[0];
          for (int i = 0; i < bs; ++i)
            for (int j = 0; j < bs; ++j)
              d[itab_g(sh, NG) + ((size_t)n * NG + g) * sh.BS * sh.BS + sh.BS * i + j] = pgr[bs * bs + i + bs * j];
        }
        d[itab_hph0(sh, NG) + n] = h * h * mq.Pinf[(size_t)o0 + (size_t)sh.S * o0];
        for (int i = 0; i < bs; ++i) d[itab_wcol0(sh, NG) + (size_t)n * sh.BS + i] = h * mq.Pinf[(size_t)(o0 + i) + (size_t)sh.S * o0];
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
    if (p->src_all.n_src >= 2 && sh.BS == 4) {   // block-structured Wnmf: the tuple tables must be resident (a shorter I/O ring makes room)
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
    // (plans with a block of 5 .. 8 states, BS = 8: the general ADF kernel ihgp_filter_kernel<MV, false, 8>; the affine scans are instantiated for both strides)
    if (sh.BS == 4 && p->sp.enabled && !p->src_f && sh.M <= 64 && sh.D <= 4 * MSP_DT && o->n_pts <= MSP_NT + 64 && (o->n_pts + 3) / 4 <= MSP_NW * MSP_NST) {
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
    if (sh.BS == 4 && p->sq_ok && !p->src_f && sh.M <= 64) {
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
#define SL8(V) PLAN_TRY(set_lds(ihgp_filter_kernel<V, false, 8>, p->lds_ih))
    if (sh.BS == 8) { NAGP_MV_SWITCH9(mom_variant(mc), SL8) } else if (p->src_f) { NAGP_MV_SWITCH9(mom_variant(mc), SLS) } else { NAGP_MV_SWITCH9(mom_variant(mc), SL) }
#undef SL
#undef SLS
#undef SL8
  } else {
    if (!ekf) p->DG_f = pick_DG(o->lik_kind, o->n_pts, p->NT_a, sh.D, o->cub_dim);
    MomCfg t = mc; t.DG = p->DG_f; t.cache_tabs = ekf ? 0 : 1; t.store_a = (!ekf && o->lik_kind == NAGP_LIK_POWER_NMF_SQRT) ? 1 : 0;
    // ADF launches in the sparse-point form (256-thread launches, <= 320 sigma points)
    if (!split && !ekf && p->sp.enabled && p->LB_a == 256 && p->NT_a == MSP_NT && sh.M <= 64 && sh.D <= 4 * MSP_DT && o->n_pts <= MSP_NT + 64 &&
        (o->n_pts + 3) / 4 <= MSP_NW * MSP_NST) {
      p->sp_gf = 1; t.sp = p->sp;
    }
    // ... or with likModulatorPreCalcwn in the staged form of nagp_momsq.hpp
    if (!split && !ekf && p->sq_ok && !p->src_all.n_src && p->LB_a == 256 && p->NT_a == 256 && sh.M <= 64) { p->sq_gf = 1; t.sq_form = 1; t.store_a = 0; }
    const size_t cap = 156 * 1024 - filter_cpl_doubles(sh) * sizeof(double);      // (split blocks: the exchange buffer and the cross tiles in front)
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
    p->lds_filter = (filter_lds_doubles(sh, t, ekf ? 1 : 0, p->kb_f) + filter_cpl_doubles(sh)) * sizeof(double);
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
    p->lds_gain = (((p->TPT == 1) ? gain_lds_doubles_staged(sh) : gain_lds_doubles(sh)) + gain_cpl_doubles(sh)) * sizeof(double);     // (rts_gain_kernel: STAGE)
    p->lds_scan = span_lds_doubles(sh, p->LP1, p->LP2) * sizeof(double);
    if (split) {
      if (ekf) {
        switch (p->TPT_f) {
          case 1: PLAN_TRY(set_lds((gf_filter_kernel<1, 1, 0, 512, 0, true>), p->lds_filter)); break;
          case 2: PLAN_TRY(set_lds((gf_filter_kernel<2, 1, 0, 512, 0, true>), p->lds_filter)); break;
          default: PLAN_TRY(set_lds((gf_filter_kernel<4, 1, 0, 512, 0, true>), p->lds_filter)); break;
        }
      } else {
#define SLC1(V) PLAN_TRY(set_lds((gf_filter_kernel<1, 0, V, 512, 0, true>), p->lds_filter))
#define SLC2(V) PLAN_TRY(set_lds((gf_filter_kernel<2, 0, V, 512, 0, true>), p->lds_filter))
#define SLC4(V) PLAN_TRY(set_lds((gf_filter_kernel<4, 0, V, 512, 0, true>), p->lds_filter))
        switch (p->TPT_f) {
          case 1: NAGP_MV_SWITCH(mom_variant(mc), SLC1) SLC1(-1); break;
          case 2: NAGP_MV_SWITCH(mom_variant(mc), SLC2) SLC2(-1); break;
          default: NAGP_MV_SWITCH(mom_variant(mc), SLC4) SLC4(-1); break;
        }
        if (p->wide_l && p->NT_l <= 768) PLAN_TRY(set_lds((gf_filter_kernel<1, 0, -1, 768, 0, true>), p->lds_filter));
        else if (p->wide_l) PLAN_TRY(set_lds((gf_filter_kernel<1, 0, -1, 1024, 0, true>), p->lds_filter));
#undef SLC1
#undef SLC2
#undef SLC4
      }
      switch (p->TPT) {
        case 1: PLAN_TRY(set_lds((rts_gain_kernel<1, 512, true>), p->lds_gain)); break;
        case 2: PLAN_TRY(set_lds((rts_gain_kernel<2, 512, true>), p->lds_gain)); break;
        case 3: PLAN_TRY(set_lds((rts_gain_kernel<3, 512, true>), p->lds_gain)); break;
        case 4: PLAN_TRY(set_lds((rts_gain_kernel<4, 512, true>), p->lds_gain)); break;
        default: PLAN_TRY(set_lds((rts_gain_kernel<8, 512, true>), p->lds_gain)); break;      // (46 .. 64 tile rows: tiles in scratch, as for unsplit models)
      }
    }
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
    if (!split && nt > 1024 && nt <= 1536 && sh.M * (sh.M + 1) / 2 <= 768 && sh.S <= 768 && !dev_env("NAGP_NO_GAIN768")) {
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
  if (!split && !ekf && o->kind == NAGP_KIND_GF_EP && o->mode == NAGP_MODE_PREDICT && !(o->flags & NAGP_FLAG_MIXTURE_RULE) && 4 * sh.M <= 160 && dev_env("NAGP_LIN_MFMA")) {
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
  if (p->mfma_sp && !split && !dev_env("NAGP_NO_GAIN_MFMA")) {      // (split blocks: the VALU kernel knows the cross tiles of A)
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
    switch (p->mfma_sp / 16) { case 5: SETB(5); break; case 6: SETB(6); break; case 7: SETB(7); break; case 8: SETB(8); break; case 9: SETB(9); break; default: SETB(10); break; }
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
    if (!split && p->sp.enabled && !p->src_ep && sh.M <= 64 && sh.D <= 4 * MSP_DT && o->cub_dim <= MSP_MAXCD && o->n_pts <= MSP_NT + 64 &&
        (o->n_pts + 3) / 4 <= MSP_NW * MSP_NST && !dev_env("NAGP_NO_SPARSE_EP")) {
      p->sp_ep = 1;
      p->lds_ep_sp = ep_sp_lds_doubles(sh, o->cub_dim) * sizeof(double);
#define SLS(V) PLAN_TRY(set_lds(ep_site_sp_kernel<V>, p->lds_ep_sp))
      switch (o->cub_dim) { case 1: SLS(1); break; case 2: SLS(2); break; case 3: SLS(3); break; case 4: SLS(4); break; case 5: SLS(5); break; case 6: SLS(6); break; default: SLS(7); break; }
#undef SLS
    }
    // ... and with likModulatorPreCalcwn in the staged form of nagp_momsq.hpp
    if (!split && p->sq_ok && !p->src_ep && sh.M <= 64 && !dev_env("NAGP_NO_SPARSE_EP")) {
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
  std::vector<double> wide;      // split blocks: the caller's Mu columns per step, zero sites on the tail rows
  for (int q = 0; q < p->B; ++q) {
    if (!ttau0[q] || !tnu0[q]) FAIL(NAGP_EINVAL, "problem %d: NULL site array", q);
    if (!p->perm.empty()) {
      const int Md = p->sh.M, Mu = p->Mu;
      wide.assign(2 * n, 0.0);
      for (int64_t k = 0; k < p->sh.T; ++k)
        for (int i = 0; i < Mu; ++i) { wide[(size_t)k * Md + i] = ttau0[q][(size_t)k * Mu + i]; wide[n + (size_t)k * Md + i] = tnu0[q][(size_t)k * Mu + i]; }
      HIP_TRY(hipMemcpy(p->d_tt0 + (size_t)q * n, wide.data(), n * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(p->d_tn0 + (size_t)q * n, wide.data() + n, n * sizeof(double), hipMemcpyHostToDevice));
      continue;
    }
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

