// nagp_inst.hpp -- the kernel instantiations libnagp.so is built from, grouped by translation unit.
// The host code (nagp_api.hip) sees every list as `extern template`; each inst_*.hip file instantiates one group, so the
// groups compile in parallel (the cubature inlined into the sequential filters makes them the expensive part of the build).
#pragma once
#include "nagp_ihgp.hpp"
#include "nagp_mfma.hpp"
#include "nagp_mfma_big.hpp"
#include "nagp_gain_mfma.hpp"
#include "nagp_filter_mfma.hpp"
#include "nagp_gfadf8.hpp"

#define NAGP_SIG_GF (nagp::Shape, nagp::Bufs, nagp::MomCfg, nagp::FilterPar)
#define NAGP_LIST_GF_ADF(P, TPT, LB)                                                                                       \
  P void nagp::gf_filter_kernel<TPT, 0, 0, LB> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 1, LB> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 2, LB> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 3, LB> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 4, LB> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 5, LB> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 6, LB> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 7, LB> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 8, LB> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 9, LB> NAGP_SIG_GF;
#define NAGP_LIST_GF_ADF1(P) NAGP_LIST_GF_ADF(P, 1, 256)
#define NAGP_LIST_GF_ADF2(P) NAGP_LIST_GF_ADF(P, 2, 256)
#define NAGP_LIST_GF_ADF3(P) NAGP_LIST_GF_ADF(P, 3, 256)
#define NAGP_LIST_GF_ADF4(P) NAGP_LIST_GF_ADF(P, 4, 256)
#define NAGP_LIST_GF_ADF5(P) NAGP_LIST_GF_ADF(P, 4, 512)

// plans with split blocks (Shape::part): one geometry for the ADF, fixed-site and EKF launches; the VALU gain kernel with the cross tiles
#define NAGP_LIST_GF_CPL(P, TPT)                                                                                                            \
  P void nagp::gf_filter_kernel<TPT, 0, 0, 512, 0, true> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 1, 512, 0, true> NAGP_SIG_GF;   \
  P void nagp::gf_filter_kernel<TPT, 0, 2, 512, 0, true> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 3, 512, 0, true> NAGP_SIG_GF;   \
  P void nagp::gf_filter_kernel<TPT, 0, 4, 512, 0, true> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 5, 512, 0, true> NAGP_SIG_GF;   \
  P void nagp::gf_filter_kernel<TPT, 0, 6, 512, 0, true> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 7, 512, 0, true> NAGP_SIG_GF;   \
  P void nagp::gf_filter_kernel<TPT, 0, 8, 512, 0, true> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 9, 512, 0, true> NAGP_SIG_GF;   \
  P void nagp::gf_filter_kernel<TPT, 0, -1, 512, 0, true> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 1, 0, 512, 0, true> NAGP_SIG_GF;
#define NAGP_LIST_GF_CPL1(P) NAGP_LIST_GF_CPL(P, 1)
#define NAGP_LIST_GF_CPL2(P) NAGP_LIST_GF_CPL(P, 2)
#define NAGP_LIST_GF_CPL4(P) NAGP_LIST_GF_CPL(P, 4)
#define NAGP_LIST_GF_CPLW(P)                                                                                                               \
  P void nagp::gf_filter_kernel<1, 0, -1, 768, 0, true> NAGP_SIG_GF; P void nagp::gf_filter_kernel<1, 0, -1, 1024, 0, true> NAGP_SIG_GF;
#define NAGP_LIST_GAIN_CPL(P)                                                                                                               \
  P void nagp::rts_gain_kernel<1, 512, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); P void nagp::rts_gain_kernel<2, 512, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_kernel<3, 512, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); P void nagp::rts_gain_kernel<4, 512, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_kernel<8, 512, true>(nagp::Shape, nagp::Bufs, nagp::GainPar);

// ADF launches in the sparse-point form (likModulatorNMFPower, 1..7 components), 256-thread launches
#define NAGP_LIST_GF_SP(P, TPT)                                                                                            \
  P void nagp::gf_filter_kernel<TPT, 0, 1, 256, 1> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 2, 256, 1> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 3, 256, 1> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 4, 256, 1> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 5, 256, 1> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 6, 256, 1> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 7, 256, 1> NAGP_SIG_GF;
#define NAGP_LIST_GF_SP12(P) NAGP_LIST_GF_SP(P, 1) NAGP_LIST_GF_SP(P, 2)
#define NAGP_LIST_GF_SP34(P) NAGP_LIST_GF_SP(P, 3) NAGP_LIST_GF_SP(P, 4)

// ADF launches with likModulatorPreCalcwn in the staged form (nagp_momsq.hpp; 1..6 components), 256-thread launches
#define NAGP_LIST_GF_SQ(P, TPT)                                                                                            \
  P void nagp::gf_filter_kernel<TPT, 0, 1, 256, 2> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 2, 256, 2> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 3, 256, 2> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 4, 256, 2> NAGP_SIG_GF;    \
  P void nagp::gf_filter_kernel<TPT, 0, 5, 256, 2> NAGP_SIG_GF; P void nagp::gf_filter_kernel<TPT, 0, 6, 256, 2> NAGP_SIG_GF;
#define NAGP_LIST_GF_SQ12(P) NAGP_LIST_GF_SQ(P, 1) NAGP_LIST_GF_SQ(P, 2)
#define NAGP_LIST_GF_SQ34(P) NAGP_LIST_GF_SQ(P, 3) NAGP_LIST_GF_SQ(P, 4)

// EKF and fixed-site filters, smoother kernels
#define NAGP_LIST_GF_REST(P)                                                                                               \
  P void nagp::gf_filter_kernel<1, 1, 0> NAGP_SIG_GF; P void nagp::gf_filter_kernel<2, 1, 0> NAGP_SIG_GF;                \
  P void nagp::gf_filter_kernel<4, 1, 0> NAGP_SIG_GF; P void nagp::gf_filter_kernel<1, 0, -1> NAGP_SIG_GF;               \
  P void nagp::gf_filter_kernel<2, 0, -1> NAGP_SIG_GF; P void nagp::gf_filter_kernel<4, 0, -1> NAGP_SIG_GF;              \
  P void nagp::gf_filter_kernel<1, 0, -1, 768> NAGP_SIG_GF; P void nagp::gf_filter_kernel<1, 0, -1, 1024> NAGP_SIG_GF;
#define NAGP_LIST_SMOOTH_T(P, TPT)                                                                                         \
  P void nagp::rts_gain_kernel<TPT>(nagp::Shape, nagp::Bufs, nagp::GainPar);                                             \
  P void nagp::rts_compose_kernel<TPT>(nagp::Shape, nagp::Bufs, nagp::SpanPar);                                          \
  P void nagp::rts_boundary_kernel<TPT>(nagp::Shape, nagp::Bufs, nagp::SpanPar);                                         \
  P void nagp::rts_apply_kernel<TPT>(nagp::Shape, nagp::Bufs, nagp::SpanPar);
// eight tiles per thread (2 049 .. 4 096 tiles: 46 .. 64 sites, the source-separation mixture at 48 channels / 9 modulators): the tiles
// spill to scratch (~3 KB per lane in the gain kernel) -- served, not tuned
#define NAGP_LIST_SMOOTH8(P) NAGP_LIST_SMOOTH_T(P, 8)
#define NAGP_LIST_SMOOTH_M(P, NTL)                                                                                         \
  P void nagp::rts_compose_mfma_kernel<NTL>(nagp::Shape, nagp::Bufs, nagp::MfmaPar);                                     \
  P void nagp::rts_boundary_mfma_kernel<NTL>(nagp::Shape, nagp::Bufs, nagp::MfmaPar);                                    \
  P void nagp::rts_apply_mfma_kernel<NTL>(nagp::Shape, nagp::Bufs, nagp::MfmaPar);
#define NAGP_LIST_SMOOTH(P)                                                                                                \
  P void nagp::rts_gain_kernel<2, 768>(nagp::Shape, nagp::Bufs, nagp::GainPar);                                         \
  NAGP_LIST_SMOOTH_T(P, 1) NAGP_LIST_SMOOTH_T(P, 2) NAGP_LIST_SMOOTH_T(P, 3) NAGP_LIST_SMOOTH_T(P, 4)                    \
  NAGP_LIST_SMOOTH_M(P, 1) NAGP_LIST_SMOOTH_M(P, 2) NAGP_LIST_SMOOTH_M(P, 3) NAGP_LIST_SMOOTH_M(P, 4)                    \
  NAGP_LIST_SMOOTH_M(P, 5) NAGP_LIST_SMOOTH_M(P, 6)

// fixed-site filter step on the matrix cores
#define NAGP_SIG_FLM (nagp::Shape, nagp::Bufs, nagp::FilterPar)
#define NAGP_LIST_FLM(P)                                                                                                   \
  P void nagp::gf_filter_lin_mfma_kernel<1, 4> NAGP_SIG_FLM; P void nagp::gf_filter_lin_mfma_kernel<2, 4> NAGP_SIG_FLM;   \
  P void nagp::gf_filter_lin_mfma_kernel<3, 4> NAGP_SIG_FLM; P void nagp::gf_filter_lin_mfma_kernel<4, 4> NAGP_SIG_FLM;   \
  P void nagp::gf_filter_lin_mfma_kernel<5, 4> NAGP_SIG_FLM; P void nagp::gf_filter_lin_mfma_kernel<6, 8> NAGP_SIG_FLM;   \
  P void nagp::gf_filter_lin_mfma_kernel<7, 8> NAGP_SIG_FLM; P void nagp::gf_filter_lin_mfma_kernel<8, 8> NAGP_SIG_FLM;   \
  P void nagp::gf_filter_lin_mfma_kernel<9, 8> NAGP_SIG_FLM; P void nagp::gf_filter_lin_mfma_kernel<10, 8> NAGP_SIG_FLM;

// RTS gain on the matrix cores (dense output), Sp = 16 .. 160
#define NAGP_LIST_GAINM(P)                                                                                                 \
  P void nagp::rts_gain_mfma_kernel<1, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<2, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<3, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<4, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<5, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<6, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<7, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<8, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<9, false>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<10, false>(nagp::Shape, nagp::Bufs, nagp::GainPar);
#define NAGP_LIST_GAINI(P)                                                                                                 \
  P void nagp::rts_gain_mfma_kernel<1, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<2, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<3, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<4, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<5, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<6, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<7, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<8, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<9, true>(nagp::Shape, nagp::Bufs, nagp::GainPar); \
  P void nagp::rts_gain_mfma_kernel<10, true>(nagp::Shape, nagp::Bufs, nagp::GainPar);

// MFMA smoother passes for 96 < Sp <= 160
#define NAGP_LIST_BIG_N(P, NTL)                                                                                            \
  P void nagp::rts_big_kernel<NTL, 0>(nagp::Shape, nagp::Bufs, nagp::MfmaPar);                                           \
  P void nagp::rts_big_kernel<NTL, 1>(nagp::Shape, nagp::Bufs, nagp::MfmaPar);                                           \
  P void nagp::rts_big_kernel<NTL, 2>(nagp::Shape, nagp::Bufs, nagp::MfmaPar);                                           \
  P void nagp::rts_big_phi_kernel<NTL>(nagp::Shape, nagp::Bufs, nagp::MfmaPar);
#define NAGP_LIST_BIG(P) NAGP_LIST_BIG_N(P, 5) NAGP_LIST_BIG_N(P, 6) NAGP_LIST_BIG_N(P, 7) NAGP_LIST_BIG_N(P, 8) NAGP_LIST_BIG_N(P, 9) NAGP_LIST_BIG_N(P, 10)

// site refresh and mom on its own
#define NAGP_LIST_EP_V(P, V)                                                                                               \
  P void nagp::ep_site_kernel<V>(nagp::Shape, nagp::Bufs, nagp::MomCfg, nagp::EpPar);                                    \
  P void nagp::mom_kernel<V>(nagp::MomCfg, nagp::MomPar);
#define NAGP_LIST_EP(P)                                                                                                    \
  NAGP_LIST_EP_V(P, 0) NAGP_LIST_EP_V(P, 1) NAGP_LIST_EP_V(P, 2) NAGP_LIST_EP_V(P, 3) NAGP_LIST_EP_V(P, 4)               \
  NAGP_LIST_EP_V(P, 5) NAGP_LIST_EP_V(P, 6) NAGP_LIST_EP_V(P, 7) NAGP_LIST_EP_V(P, 8) NAGP_LIST_EP_V(P, 9)

// site refresh in the sparse-point form
#define NAGP_SIG_EPS (nagp::Shape, nagp::Bufs, nagp::MomCfg, nagp::EpPar)
#define NAGP_LIST_EPS(P)                                                                                                   \
  P void nagp::ep_site_sp_kernel<1> NAGP_SIG_EPS; P void nagp::ep_site_sp_kernel<2> NAGP_SIG_EPS; P void nagp::ep_site_sp_kernel<3> NAGP_SIG_EPS;   \
  P void nagp::ep_site_sp_kernel<4> NAGP_SIG_EPS; P void nagp::ep_site_sp_kernel<5> NAGP_SIG_EPS; P void nagp::ep_site_sp_kernel<6> NAGP_SIG_EPS;   \
  P void nagp::ep_site_sp_kernel<7> NAGP_SIG_EPS;

// site refresh with likModulatorPreCalcwn in the staged form (nagp_momsq.hpp, flat layout)
#define NAGP_LIST_EPQ(P)                                                                                                   \
  P void nagp::ep_site_sq_kernel<1> NAGP_SIG_EPS; P void nagp::ep_site_sq_kernel<2> NAGP_SIG_EPS; P void nagp::ep_site_sq_kernel<3> NAGP_SIG_EPS;   \
  P void nagp::ep_site_sq_kernel<4> NAGP_SIG_EPS; P void nagp::ep_site_sq_kernel<5> NAGP_SIG_EPS; P void nagp::ep_site_sq_kernel<6> NAGP_SIG_EPS;

// infinite-horizon filters
#define NAGP_SIG_IH (nagp::Shape, nagp::Bufs, nagp::MomCfg, nagp::IhgpTabs, nagp::IhgpPar)
#define NAGP_LIST_IH_S(P, SRC)                                                                                             \
  P void nagp::ihgp_filter_kernel<0, SRC> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<1, SRC> NAGP_SIG_IH;              \
  P void nagp::ihgp_filter_kernel<2, SRC> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<3, SRC> NAGP_SIG_IH;              \
  P void nagp::ihgp_filter_kernel<4, SRC> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<5, SRC> NAGP_SIG_IH;              \
  P void nagp::ihgp_filter_kernel<6, SRC> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<7, SRC> NAGP_SIG_IH;              \
  P void nagp::ihgp_filter_kernel<8, SRC> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<9, SRC> NAGP_SIG_IH;
#define NAGP_LIST_IH0(P) NAGP_LIST_IH_S(P, false)
// ... with blocks of 5 .. 8 states
#define NAGP_LIST_IH8(P)                                                                                                   \
  P void nagp::ihgp_filter_kernel<0, false, 8> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<1, false, 8> NAGP_SIG_IH;    \
  P void nagp::ihgp_filter_kernel<2, false, 8> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<3, false, 8> NAGP_SIG_IH;    \
  P void nagp::ihgp_filter_kernel<4, false, 8> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<5, false, 8> NAGP_SIG_IH;    \
  P void nagp::ihgp_filter_kernel<6, false, 8> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<7, false, 8> NAGP_SIG_IH;    \
  P void nagp::ihgp_filter_kernel<8, false, 8> NAGP_SIG_IH; P void nagp::ihgp_filter_kernel<9, false, 8> NAGP_SIG_IH;
#define NAGP_LIST_IH1(P) NAGP_LIST_IH_S(P, true)
#define NAGP_SIG_IHA (nagp::Shape, nagp::Bufs, nagp::MomCfg, nagp::MomSp, nagp::IhgpTabs, nagp::IhgpPar)
#define NAGP_LIST_IHA(P)                                                                                                   \
  P void nagp::ihgp_adf_kernel<1> NAGP_SIG_IHA; P void nagp::ihgp_adf_kernel<2> NAGP_SIG_IHA;                            \
  P void nagp::ihgp_adf_kernel<3> NAGP_SIG_IHA; P void nagp::ihgp_adf_kernel<4> NAGP_SIG_IHA;                            \
  P void nagp::ihgp_adf_kernel<5> NAGP_SIG_IHA; P void nagp::ihgp_adf_kernel<6> NAGP_SIG_IHA;                            \
  P void nagp::ihgp_adf_kernel<7> NAGP_SIG_IHA;

#define NAGP_LIST_IHA8(P)                                                                                                  \
  P void nagp::ihgp_adf8_kernel<1, false> NAGP_SIG_IHA; P void nagp::ihgp_adf8_kernel<2, false> NAGP_SIG_IHA;            \
  P void nagp::ihgp_adf8_kernel<3, false> NAGP_SIG_IHA; P void nagp::ihgp_adf8_kernel<4, false> NAGP_SIG_IHA;            \
  P void nagp::ihgp_adf8_kernel<5, false> NAGP_SIG_IHA; P void nagp::ihgp_adf8_kernel<6, false> NAGP_SIG_IHA;            \
  P void nagp::ihgp_adf8_kernel<7, false> NAGP_SIG_IHA;                                                                  \
  P void nagp::ihgp_adf8_kernel<1, true> NAGP_SIG_IHA; P void nagp::ihgp_adf8_kernel<2, true> NAGP_SIG_IHA;              \
  P void nagp::ihgp_adf8_kernel<3, true> NAGP_SIG_IHA; P void nagp::ihgp_adf8_kernel<4, true> NAGP_SIG_IHA;              \
  P void nagp::ihgp_adf8_kernel<5, true> NAGP_SIG_IHA; P void nagp::ihgp_adf8_kernel<6, true> NAGP_SIG_IHA;

// the role-specialised sweep for likModulatorPreCalcwn (nagp_momsq.hpp)
#define NAGP_LIST_IHA8Q(P)                                                                                                 \
  P void nagp::ihgp_adf8sq_kernel<1> NAGP_SIG_IHA; P void nagp::ihgp_adf8sq_kernel<2> NAGP_SIG_IHA;                      \
  P void nagp::ihgp_adf8sq_kernel<3> NAGP_SIG_IHA; P void nagp::ihgp_adf8sq_kernel<4> NAGP_SIG_IHA;                      \
  P void nagp::ihgp_adf8sq_kernel<5> NAGP_SIG_IHA; P void nagp::ihgp_adf8sq_kernel<6> NAGP_SIG_IHA;

// the ADF sweep of the full-covariance filter with role-specialised waves (nagp_gfadf8.hpp): one or two lower tiles per thread
#define NAGP_LIST_GF_A8T(P, TPT, ST)                                                                                       \
  P void nagp::gf_adf8_kernel<TPT, 1, false, ST> NAGP_SIG_GF; P void nagp::gf_adf8_kernel<TPT, 2, false, ST> NAGP_SIG_GF;        \
  P void nagp::gf_adf8_kernel<TPT, 3, false, ST> NAGP_SIG_GF; P void nagp::gf_adf8_kernel<TPT, 4, false, ST> NAGP_SIG_GF;        \
  P void nagp::gf_adf8_kernel<TPT, 5, false, ST> NAGP_SIG_GF; P void nagp::gf_adf8_kernel<TPT, 6, false, ST> NAGP_SIG_GF;        \
  P void nagp::gf_adf8_kernel<TPT, 7, false, ST> NAGP_SIG_GF;                                                                \
  P void nagp::gf_adf8_kernel<TPT, 1, true, ST> NAGP_SIG_GF; P void nagp::gf_adf8_kernel<TPT, 2, true, ST> NAGP_SIG_GF;          \
  P void nagp::gf_adf8_kernel<TPT, 3, true, ST> NAGP_SIG_GF; P void nagp::gf_adf8_kernel<TPT, 4, true, ST> NAGP_SIG_GF;          \
  P void nagp::gf_adf8_kernel<TPT, 5, true, ST> NAGP_SIG_GF; P void nagp::gf_adf8_kernel<TPT, 6, true, ST> NAGP_SIG_GF;
#define NAGP_LIST_GF_A81(P) NAGP_LIST_GF_A8T(P, 1, false)
#define NAGP_LIST_GF_A82(P) NAGP_LIST_GF_A8T(P, 2, false)
#define NAGP_LIST_GF_A83(P) NAGP_LIST_GF_A8T(P, 2, true)

#define NAGP_LIST_ALL(P)                                                                                                   \
  NAGP_LIST_GF_ADF1(P) NAGP_LIST_GF_ADF2(P) NAGP_LIST_GF_ADF3(P) NAGP_LIST_GF_ADF4(P) NAGP_LIST_GF_ADF5(P)               \
  NAGP_LIST_GF_SP12(P) NAGP_LIST_GF_SP34(P) NAGP_LIST_GF_SQ12(P) NAGP_LIST_GF_SQ34(P) NAGP_LIST_GF_REST(P) NAGP_LIST_SMOOTH(P) NAGP_LIST_SMOOTH8(P) NAGP_LIST_BIG(P) NAGP_LIST_GAINM(P) NAGP_LIST_GAINI(P) NAGP_LIST_FLM(P) NAGP_LIST_EP(P) NAGP_LIST_EPS(P) NAGP_LIST_EPQ(P) NAGP_LIST_IH0(P) NAGP_LIST_IH8(P) NAGP_LIST_IH1(P) NAGP_LIST_IHA(P) NAGP_LIST_IHA8(P) NAGP_LIST_IHA8Q(P) NAGP_LIST_GF_A81(P) NAGP_LIST_GF_A82(P) NAGP_LIST_GF_A83(P) NAGP_LIST_GF_CPL1(P) NAGP_LIST_GF_CPL2(P) NAGP_LIST_GF_CPL4(P) NAGP_LIST_GF_CPLW(P) NAGP_LIST_GAIN_CPL(P)
