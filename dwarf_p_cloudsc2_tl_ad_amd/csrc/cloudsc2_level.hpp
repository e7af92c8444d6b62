// cloudsc2_level.hpp -- one model level of one grid column of CLOUDSC2: trajectory, tangent-linear
// and adjoint, as inline device functions.  All three HIP kernels (cloudsc2_kernels.hip) are built from
// these, so the TL and the AD see bit-identical trajectory values and branch decisions
// (the reference's linearisation freezes every branch on the trajectory).
//
// Design (not a transliteration of the Fortran):
//   * The reference sweeps (KLON,KLEV) work arrays level by level in ~10 fissioned JL loops
//     (src/cloudsc2_nl/cloudsc2.F90:339-725).  Here one GPU lane owns one column; a level is a pure
//     function (inputs at JK, 3 carried scalars) -> (outputs at JK, 3 carried scalars); all
//     intermediates live in registers (struct LevelTraj).
//   * The TL (src/cloudsc2_tl/cloudsc2tl.F90:453-1101) recomputes the trajectory statement by statement
//     next to each perturbation statement; here the level's trajectory is evaluated once into LevelTraj
//     and level_tl() consumes it.
//   * The AD (src/cloudsc2_ad/cloudsc2ad.F90) stores 117 (KLON,KLEV) arrays in a forward sweep
//     (:366-866) and unwinds them (:934-1668).  Here only the 3 carried scalars are checkpointed per
//     level; the reverse sweep re-evaluates level_forward() for level JK and level_ad() applies the
//     transposed statements to register-resident adjoints.
// Operation order inside branch-deciding expressions follows the reference so that branch decisions
// agree with it to rounding.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <cmath>
#include "../../include/cloudsc2_hip.h"

// -DC2_EVAP_FAST=0: the evaporation block (LEVAPLS2 / LDRAIN1D variants) keeps IEEE divisions and pow() in fast mode too (A/B builds)
#ifndef C2_EVAP_FAST
#define C2_EVAP_FAST 1
#endif
// -DC2_EVAP_FAST_TLAD=0: only the tangent / adjoint blocks of level_tl / level_ad keep their divisions, pow() and sqrt() (A/B builds)
#ifndef C2_EVAP_FAST_TLAD
#define C2_EVAP_FAST_TLAD C2_EVAP_FAST
#endif

namespace cloudsc2 {

// JPRB (src/common/module/parkind1.F90:40-44): fp64, or fp32 when the library is built with -DCLOUDSC2_SINGLE (the
// reference's -DSINGLE).  Every literal of the column code is written RC(x) so that no fp32 expression is promoted.
typedef cloudsc2_real real_t;
#define RC(x) ((real_t)(x))
// <cmath>'s overload sets, so that the same call is the fp32 or the fp64 function
using std::cosh; using std::exp; using std::fabs; using std::fma; using std::fmax; using std::fmin; using std::ldexp;
using std::pow; using std::sqrt; using std::tanh;

#define C2_HD __host__ __device__ __forceinline__

// Launch-invariant data (constants, field pointers) is read through the SCALAR cache straight from the kernel-
// argument segment, inside the level loop, instead of being held in SGPRs for the whole kernel: ~40 fp64 constants
// + 26..68 field pointers do not fit the 102 SGPRs of a wave and were being spilled to VGPR lanes
// (v_writelane/v_readlane).  C2_LAUNDER makes the base pointer opaque at stage boundaries so that the s_load of a
// constant is emitted next to its use and its SGPRs die right after.  On the host both are plain C++.
#if defined(__HIP_DEVICE_COMPILE__)
#define C2_CONST_AS __attribute__((address_space(4)))
#define C2_LAUNDER(p) asm volatile("" : "+s"(p))
#else
#define C2_CONST_AS
#define C2_LAUNDER(p) ((void)0)
#endif

// ---------------------------------------------------------------------------------------------------------
// fp64 math building blocks.  The level costs ~50 divisions, ~12 exp and a tanh in the reference formulation
// (SURVEY.md 8d); on CDNA4 an IEEE fp64 division is ~13 instructions around a quarter-rate v_rcp_f64, so the
// arithmetic, not HBM, bounds the kernel.  Hence: reciprocals are taken with v_rcp_f64 + one third-order step and
// shared / batch-inverted (Montgomery) between quotients, exp is a 15-instruction branch-free kernel (c2_exp),
// tanh comes from one exp.  Results differ from correctly rounded ones by a few ulp (parity tolerance: 1e-10).
// ---------------------------------------------------------------------------------------------------------
#if defined(CLOUDSC2_SINGLE)
C2_HD real_t c2_rcp(real_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_amdgcn_rcpf(x);  // v_rcp_f32: 1 ulp
#else
  return 1.0f / x;
#endif
}
#else
C2_HD real_t c2_rcp(real_t x) {
#if defined(__HIP_DEVICE_COMPILE__)
  // v_rcp_f64 is good to ~2^-23; one third-order step r(1 + e + e^2), e = 1 - x r, takes that to 2^-69 < 1/2 ulp
  // in three fma (two Newton steps need four)
  real_t r = __builtin_amdgcn_rcp(x);
  real_t e = __builtin_fma(-x, r, 1.0);
  real_t t = __builtin_fma(e, e, e);
  return __builtin_fma(r, t, r);
#else
  return 1.0 / x;
#endif
}

#endif

C2_HD void c2_rcp2(real_t a, real_t b, real_t& ra, real_t& rb) {
  real_t r = c2_rcp(a * b);
  ra = b * r;
  rb = a * r;
}

// 1/a, 1/b, 1/c from one reciprocal
C2_HD void c2_rcp3(real_t a, real_t b, real_t c, real_t& ra, real_t& rb, real_t& rc) {
  real_t ab = a * b;
  real_t r = c2_rcp(ab * c);
  rc = ab * r;
  real_t rab = c * r;
  ra = b * rab;
  rb = a * rab;
}

// PRECISE = the reference's own operation order with IEEE division and libm exp/tanh (used by the Taylor-test
// driver, whose V-shape verdict is decided by round-off noise at lambda <= 1e-7); otherwise shared reciprocals.
template <bool PRECISE>
C2_HD real_t quot(real_t num, real_t den, real_t rden) { return PRECISE ? num / den : num * rden; }
template <bool PRECISE>
C2_HD real_t recip(real_t den) { return PRECISE ? 1.0 / den : c2_rcp(den); }

#if defined(CLOUDSC2_SINGLE)
// fp32 exp: n = round(x log2 e) by the 1.5*2^23 trick, two-constant Cody-Waite reduction (the high part of ln2 has 9
// trailing zero bits, so n*ln2_hi is exact for |n| < 512), degree-6 near-minimax polynomial on |r| <= ln2/2 (1e-8), v_ldexp_f32.
C2_HD real_t c2_exp(real_t x) {
  const float log2e = 1.44269502e+00f, ln2_hi = 6.93145752e-01f, ln2_lo = 1.42860677e-06f;
  const float magic = 12582912.0f;  // 1.5 * 2^23
  x = fmaxf(x, -200.0f);            // exp(-200) underflows to 0 in fp32 like exp(-inf)
  const float nb = fmaf(x, log2e, magic);
  const float n = nb - magic;
  float r = fmaf(-n, ln2_hi, x);
  r = fmaf(-n, ln2_lo, r);
  float p = 1.393365208e-03f;
  p = fmaf(p, r, 8.363181725e-03f);
  p = fmaf(p, r, 4.166646674e-02f);
  p = fmaf(p, r, 1.666657627e-01f);
  p = fmaf(p, r, 5.0e-01f);
  p = fmaf(p, r, 1.0f);
  p = fmaf(p, r, 1.0f);
  return ldexpf(p, (int)n);
}
#else
// exp: n = round(x log2 e) by the 1.5*2^52 trick (the low dword of the biased sum IS the integer n, no convert),
// r = x - n ln2 in one fma (ln2 rounded once: the error n*7.6e-17 stays below 3e-15 for |x| < 60, 2.4e-14 at the
// underflow end), degree-10 near-minimax polynomial on |r| <= ln2/2 (3.3e-16), one v_ldexp_f64: 15 instructions.
C2_HD real_t c2_exp(real_t x) {
  const real_t log2e = 1.44269504088896338700e+00, ln2 = 6.93147180559945286227e-01;
  const real_t magic = 6755399441055744.0;  // 1.5 * 2^52
  x = fmax(x, -1000.0);                     // keeps n inside 32 bits; exp(-1000) underflows to 0 like exp(-inf)
  const real_t nb = fma(x, log2e, magic);
  const real_t n = nb - magic;
  const real_t r = fma(-n, ln2, x);
  real_t p = 0x1.28a2c6cc1d7acp-22;
  p = fma(p, r, 0x1.72faf086f80f0p-19);
  p = fma(p, r, 0x1.a019a6617b7afp-16);
  p = fma(p, r, 0x1.a01978c64c250p-13);
  p = fma(p, r, 0x1.6c16c17f4783ep-10);
  p = fma(p, r, 0x1.1111112dd6a6dp-7);
  p = fma(p, r, 0x1.55555555520a2p-5);
  p = fma(p, r, 0x1.555555554b755p-3);
  p = fma(p, r, 0x1.0000000000005p-1);
  p = fma(p, r, 0x1.000000000001ep+0);
  p = fma(p, r, 1.0);
  return ldexp(p, (int)(unsigned)__builtin_bit_cast(unsigned long long, nb));
}
#endif

struct StageBlock { real_t v[8]; };
// k0: first guess, entry reciprocals, stage A (saturation pressure)
enum { K0_PTSPHY, K0_R4LES, K0_R4IES, K0_RTT, K0_RLPTRC, K0_R3IES, K0_R3LES, K0_R2ES };
// k1: stage A (dqs/dT), B, D
enum { K1_R5LES, K1_R5IES, K1_RETV, K1_RD, K1_ZCONS3, K1_RTICE, K1_RG, K1_RLMIN };
// k2: stages D..H
enum { K2_PTSPHY, K2_RCPD_R, K2_ZQTMST, K2_ZCONS2, K2_ZLFDCP0_R, K2_ZMELTP2, K2_ZCONS2_R, K2_RG };
// k6: latent heats over cp when RVTMP2 == 0
enum { K6_ZLVDCP0, K6_ZLSDCP0, K6_ZLFDCP0, K6_SPARE0, K6_SPARE1, K6_SPARE2, K6_SPARE3, K6_SPARE4 };
// k3: stage I (autoconversion)
enum { K3_ZLCRIT_L_R, K3_ZCKCODTL, K3_ZLCRIT_I_R, K3_ZCKCODTI, K3_RTT, K3_ZCONS2, K3_ZLCRIT_L, K3_ZLCRIT_I };
// k4: stages K, L (saturation adjustment)
enum { K4_RTT, K4_R3LES, K4_R4LES, K4_R3IES, K4_R4IES, K4_R2ES, K4_RETV, K4_PTSPHY };
// k5: stages L, M, enthalpy fluxes
enum { K5_R5ALVCP, K5_RALVDCP, K5_R5ALSCP, K5_RALSDCP, K5_ZCONS2, K5_ZQTMST, K5_RLVTT, K5_RLSTT };
// ks: SATUR
enum { KS_R4LES, KS_R4IES, KS_RTT, KS_R3LES, KS_R3IES, KS_R2ES, KS_RETV, KS_SPARE };
// kt0..kt2: level_tl / level_ad
enum { KT0_PTSPHY, KT0_RTT, KT0_R3IES, KT0_R4IES, KT0_R3LES, KT0_R4LES, KT0_R5LES, KT0_R5IES };
enum { KT1_RETV, KT1_ZCONS3, KT1_RG, KT1_ZQTMST, KT1_ZCONS2, KT1_ZMELTP2, KT1_ZLCRIT_L_R2, KT1_ZLCRIT_I_R2 };
enum { KT2_CK_L, KT2_CK_I, KT2_RLVTT, KT2_RLSTT, KT2_SPARE0, KT2_SPARE1, KT2_SPARE2, KT2_SPARE3 };
// kf: FOEALFA
enum { KF_RTICE, KF_RTWAT, KF_RTWAT_RTICE_R, KF_SPARE0, KF_SPARE1, KF_SPARE2, KF_SPARE3, KF_SPARE4 };

// Launch-invariant scalars: the module constants plus what CLOUDSC2 derives from them at entry
// (cloudsc2.F90:235-244, cloudsc2tl.F90:321-332).
struct Consts {
  real_t rg, rd, rcpd, retv, rlvtt, rlstt, rlmlt, rtt;
  real_t r2es, r3les, r3ies, r4les, r4ies, r5les, r5ies, r5alvcp, r5alscp, ralvdcp, ralsdcp;
  real_t rtwat, rtice, rtwat_rtice_r, rvtmp2;
  real_t rlmin, rpecons, rlptrc;
  real_t ptsphy, zckcodtl, zckcodti, zckcodtla, zckcodtia, zcons2, zcons3, zmeltp2, zqtmst;
  real_t zlcrit_l, zlcrit_i;   // autoconversion thresholds (cloudsc2.F90:505-509,522-526)
  real_t rcpd_r;               // 1/RCPD
  real_t zzz0;                 // 1/(RCPD+RCPD*RVTMP2*q) when RVTMP2 == 0
  real_t zlcrit_l_r, zlcrit_i_r;  // 1/ZLCRIT
  real_t zlfdcp0_r;               // 1/ZLFDCP when RVTMP2 == 0
  real_t zcons2_r;                // PTSPHY*RG = 1/ZCONS2
  // the same values grouped by stage of use (see C2_PIN8); index names below
  StageBlock k0, k1, k2, k3, k4, k5, k6, ks, kf;
  StageBlock kt0, kt1, kt2;  // constants of level_tl / level_ad
  int evap;                    // LEVAPLS2 .OR. LDRAIN1D
  int lregcl;
  int rvtmp2_zero;
  int nlev;
};

// host side: derive the per-stage blocks from the flat members (call last when building a Consts)
inline void fill_stage_blocks(Consts& c) {
  real_t* k;
  k = c.k0.v; k[K0_PTSPHY] = c.ptsphy; k[K0_R4LES] = c.r4les; k[K0_R4IES] = c.r4ies; k[K0_RTT] = c.rtt;
  k[K0_RLPTRC] = c.rlptrc; k[K0_R3IES] = c.r3ies; k[K0_R3LES] = c.r3les; k[K0_R2ES] = c.r2es;
  k = c.k1.v; k[K1_R5LES] = c.r5les; k[K1_R5IES] = c.r5ies; k[K1_RETV] = c.retv; k[K1_RD] = c.rd;
  k[K1_ZCONS3] = c.zcons3; k[K1_RTICE] = c.rtice; k[K1_RG] = c.rg; k[K1_RLMIN] = c.rlmin;
  k = c.k2.v; k[K2_PTSPHY] = c.ptsphy; k[K2_RCPD_R] = c.rcpd_r; k[K2_ZQTMST] = c.zqtmst; k[K2_ZCONS2] = c.zcons2;
  k[K2_ZLFDCP0_R] = c.zlfdcp0_r; k[K2_ZMELTP2] = c.zmeltp2; k[K2_ZCONS2_R] = c.zcons2_r; k[K2_RG] = c.rg;
  k = c.k6.v; k[K6_ZLVDCP0] = c.rlvtt * c.zzz0; k[K6_ZLSDCP0] = c.rlstt * c.zzz0; k[K6_ZLFDCP0] = c.rlmlt * c.zzz0;
  k[K6_SPARE0] = c.zzz0;
  k[K6_SPARE1] = k[K6_SPARE2] = k[K6_SPARE3] = k[K6_SPARE4] = RC(0.0);
  k = c.k3.v; k[K3_ZLCRIT_L_R] = c.zlcrit_l_r; k[K3_ZCKCODTL] = c.zckcodtl; k[K3_ZLCRIT_I_R] = c.zlcrit_i_r;
  k[K3_ZCKCODTI] = c.zckcodti; k[K3_RTT] = c.rtt; k[K3_ZCONS2] = c.zcons2; k[K3_ZLCRIT_L] = c.zlcrit_l;
  k[K3_ZLCRIT_I] = c.zlcrit_i;
  k = c.k4.v; k[K4_RTT] = c.rtt; k[K4_R3LES] = c.r3les; k[K4_R4LES] = c.r4les; k[K4_R3IES] = c.r3ies;
  k[K4_R4IES] = c.r4ies; k[K4_R2ES] = c.r2es; k[K4_RETV] = c.retv; k[K4_PTSPHY] = c.ptsphy;
  k = c.k5.v; k[K5_R5ALVCP] = c.r5alvcp; k[K5_RALVDCP] = c.ralvdcp; k[K5_R5ALSCP] = c.r5alscp;
  k[K5_RALSDCP] = c.ralsdcp; k[K5_ZCONS2] = c.zcons2; k[K5_ZQTMST] = c.zqtmst; k[K5_RLVTT] = c.rlvtt;
  k[K5_RLSTT] = c.rlstt;
  k = c.ks.v; k[KS_R4LES] = c.r4les; k[KS_R4IES] = c.r4ies; k[KS_RTT] = c.rtt; k[KS_R3LES] = c.r3les;
  k[KS_R3IES] = c.r3ies; k[KS_R2ES] = c.r2es; k[KS_RETV] = c.retv; k[KS_SPARE] = RC(0.0);
  k = c.kt0.v; k[KT0_PTSPHY] = c.ptsphy; k[KT0_RTT] = c.rtt; k[KT0_R3IES] = c.r3ies; k[KT0_R4IES] = c.r4ies;
  k[KT0_R3LES] = c.r3les; k[KT0_R4LES] = c.r4les; k[KT0_R5LES] = c.r5les; k[KT0_R5IES] = c.r5ies;
  k = c.kt1.v; k[KT1_RETV] = c.retv; k[KT1_ZCONS3] = c.zcons3; k[KT1_RG] = c.rg; k[KT1_ZQTMST] = c.zqtmst;
  k[KT1_ZCONS2] = c.zcons2; k[KT1_ZMELTP2] = c.zmeltp2; k[KT1_ZLCRIT_L_R2] = c.zlcrit_l_r * c.zlcrit_l_r;
  k[KT1_ZLCRIT_I_R2] = c.zlcrit_i_r * c.zlcrit_i_r;
  // regularised autoconversion coefficients (cloudsc2tl.F90:754-760,794-800)
  k = c.kt2.v; k[KT2_CK_L] = c.lregcl ? c.zckcodtla : c.zckcodtl; k[KT2_CK_I] = c.lregcl ? c.zckcodtia : c.zckcodti;
  k[KT2_RLVTT] = c.rlvtt; k[KT2_RLSTT] = c.rlstt; k[KT2_SPARE0] = k[KT2_SPARE1] = k[KT2_SPARE2] = k[KT2_SPARE3] = RC(0.0);
  k = c.kf.v; k[KF_RTICE] = c.rtice; k[KF_RTWAT] = c.rtwat; k[KF_RTWAT_RTICE_R] = c.rtwat_rtice_r;
  k[KF_SPARE0] = k[KF_SPARE1] = k[KF_SPARE2] = k[KF_SPARE3] = k[KF_SPARE4] = RC(0.0);
}

// host side: everything the kernels need besides fields and level tables, from the caller's parameter block
inline Consts make_consts(const cloudsc2_params& p, double ptsphy) {
  Consts c;
  c.rg = p.rg; c.rd = p.rd; c.rcpd = p.rcpd; c.retv = p.retv; c.rlvtt = p.rlvtt; c.rlstt = p.rlstt;
  c.rlmlt = p.rlmlt; c.rtt = p.rtt;
  c.r2es = p.r2es; c.r3les = p.r3les; c.r3ies = p.r3ies; c.r4les = p.r4les; c.r4ies = p.r4ies;
  c.r5les = p.r5les; c.r5ies = p.r5ies; c.r5alvcp = p.r5alvcp; c.r5alscp = p.r5alscp;
  c.ralvdcp = p.ralvdcp; c.ralsdcp = p.ralsdcp;
  c.rtwat = p.rtwat; c.rtice = p.rtice; c.rtwat_rtice_r = p.rtwat_rtice_r; c.rvtmp2 = p.rvtmp2;
  c.rlmin = p.rlmin; c.rpecons = p.rpecons; c.rlptrc = p.rlptrc;
  c.ptsphy = ptsphy;
  // cloudsc2.F90:235-240, cloudsc2tl.F90:321-328
  c.zckcodtl = RC(2.0) * p.rkconv * ptsphy;
  c.zckcodti = RC(5.0) * p.rkconv * ptsphy;
  c.zckcodtla = c.zckcodtl / RC(100.0);
  c.zckcodtia = c.zckcodti / RC(100.0);
  c.zcons2 = RC(1.0) / (ptsphy * p.rg);
  c.zcons3 = p.rlvtt / p.rcpd;
  c.zmeltp2 = p.rtt + RC(2.0);
  c.zqtmst = RC(1.0) / ptsphy;
  c.evap = (p.levapls2 || p.ldrain1d) ? 1 : 0;
  // cloudsc2.F90:505-509,522-526
  c.zlcrit_l = c.evap ? RC(1.9) * p.rclcrit : p.rclcrit * RC(2.0);
  c.zlcrit_i = c.evap ? RC(1.e-04) : p.rclcrit * RC(2.0);
  c.rcpd_r = RC(1.0) / p.rcpd;
  c.zlcrit_l_r = RC(1.0) / c.zlcrit_l;
  c.zlcrit_i_r = RC(1.0) / c.zlcrit_i;
  c.zcons2_r = ptsphy * p.rg;
  c.rvtmp2_zero = (p.rvtmp2 == RC(0.0)) ? 1 : 0;
  c.zzz0 = RC(1.0) / (p.rcpd + p.rcpd * p.rvtmp2 * RC(0.0));
  c.zlfdcp0_r = RC(1.0) / (p.rlmlt * c.zzz0);
  c.lregcl = p.lregcl ? 1 : 0;
  c.nlev = p.nlev;
  fill_stage_blocks(c);
  return c;
}

// Raw inputs of one level (dummy arguments of CLOUDSC2 at (JL,JK); cloudsc2.F90:124-143).
struct LevelIn {
  real_t paph_k, paph_k1;  // PAPHP1(JK), PAPHP1(JK+1)
  real_t pap, q, qs, t, l, i, lude, lu_k1, mfu, mfd, gt, gq, gl, gi, supsat;  // lu_k1 = PLU(JK+1)
  real_t paph_surf;        // PAPHP1(KLEV+1), only read when evap
};

// Carried top->bottom (cloudsc2.F90:305-312,720-723).
struct Carry {
  real_t rfl, sfl, covptot;
};

// Outputs of one level (cloudsc2.F90:709-715,732-733; PCOVPTOT :582).
struct LevelOut {
  real_t tent, tenq, tenl, teni, clc, covptot, fplsl, fplsn;  // fluxes at half level JK+1
  real_t fhpsl, fhpsn;                                        // enthalpy fluxes (cloudsc2.F90:732-733)
};

// Per-level, column-independent values prepared on the host.
typedef const C2_CONST_AS Consts* ConstsP;

#define C2_CONSTS(a) (&(a)->c)

// A wave has 102 SGPRs; the ~40 fp64 constants of a level plus the field pointers do not fit, and left to itself the
// compiler sinks every constant's s_load next to its use (one exposed scalar-cache latency per constant: 42
// `s_waitcnt lgkmcnt(0)` per level, half of the wave's lifetime).  The constants are therefore grouped by the stage that
// uses them into 64-byte blocks; a stage copies its block with ONE s_load_dwordx16 and pins it (C2_PIN8), i.e. one
// wait per stage, and the SGPRs are free again afterwards.
#if defined(__HIP_DEVICE_COMPILE__)
// The pins are "s" INPUTS of an empty asm statement: the values stay rematerialisable, i.e. under SGPR pressure the
// allocator re-issues the s_load inside a divergent branch instead of spilling.  ("+s" operands, which turn the values
// into definitions that get parked in VGPR lanes under pressure, measured the same within noise.)
#define C2_PIN8(b) asm volatile("" ::"s"((b).v[0]), "s"((b).v[1]), "s"((b).v[2]), "s"((b).v[3]), "s"((b).v[4]), "s"((b).v[5]), \
                                 "s"((b).v[6]), "s"((b).v[7]))
#define C2_PIN2(x, y) asm volatile("" ::"s"(x), "s"(y))
#else
#define C2_PIN8(b) ((void)0)
#define C2_PIN2(x, y) ((void)0)
#endif
template <class B>
C2_HD B c2_block(const C2_CONST_AS B* p) {
  B b = *p;
  C2_PIN8(b);
  return b;
}
// three blocks with one wait
template <class B>
C2_HD void c2_block3(const C2_CONST_AS B* p, const C2_CONST_AS B* q, const C2_CONST_AS B* r, B& a, B& b, B& cc) {
  a = *p;
  b = *q;
  cc = *r;
  C2_PIN8(a);
  C2_PIN8(b);
  C2_PIN8(cc);
}
// two blocks with one wait: both loads are issued before the first pin
template <class B>
C2_HD void c2_block2(const C2_CONST_AS B* p, const C2_CONST_AS B* q, B& a, B& b) {
  a = *p;
  b = *q;
  C2_PIN8(a);
  C2_PIN8(b);
}

struct LevelCst {
  real_t ceta, zscalm;  // CETA(JK); ZSCALM(JK) (cloudsc2.F90:266)
  int last;             // JK == KLEV
};

// Per-column critical-RH set-up, depends only on ZTRPAUS (cloudsc2.F90:384-390).
struct RhCrit {
  real_t zeta3, zrh2, zdeta1;
};

C2_HD void rhcrit_setup(real_t ztrpaus, RhCrit& r) {
  r.zeta3 = ztrpaus;
  real_t d = ztrpaus - RC(0.25);
  real_t dq = d / RC(0.15);
  r.zrh2 = RC(0.35) + RC(0.14) * (dq * dq) + RC(0.04) * fmin(d, RC(0.0)) / RC(0.15);
  r.zdeta1 = RC(0.09) + RC(0.16) * (RC(0.4) - ztrpaus) / RC(0.3);
}

C2_HD real_t rhcrit_level(const RhCrit& r, real_t ceta) {
  // cloudsc2.F90:391-399 (ZRH1 = ZRH3 = 1, ZDETA2 = 0.3)
  const real_t zdeta2 = RC(0.3);
  real_t zcrh2 = RC(1.0);
  if (ceta < r.zeta3) {
    zcrh2 = RC(1.0);
  } else if (ceta < (r.zeta3 + zdeta2)) {
    zcrh2 = RC(1.0) + (r.zrh2 - RC(1.0)) * ((ceta - r.zeta3) / zdeta2);
  } else if (ceta < (RC(1.0) - r.zdeta1)) {
    zcrh2 = r.zrh2;
  } else {
    zcrh2 = RC(1.0) + (r.zrh2 - RC(1.0)) * sqrt((RC(1.0) - ceta) / r.zdeta1);
  }
  return zcrh2;
}

template <bool PRECISE>
C2_HD real_t ex(real_t x) { return PRECISE ? exp(x) : c2_exp(x); }

// FOEALFA (src/common/include/fcttre.func.h:74-75)
C2_HD real_t foealfa(ConstsP c, real_t t) {
  real_t x = (fmax(c->rtice, fmin(c->rtwat, t)) - c->rtice) * c->rtwat_rtice_r;
  return fmin(RC(1.0), x * x);
}

// SATUR, LDPHYLIN branch (src/cloudsc2_nl/satur.F90:106-123)
template <bool P>
C2_HD real_t satur_point(ConstsP c, real_t pap, real_t t) {
  StageBlock kf, ks;
  c2_block2(&c->kf, &c->ks, kf, ks);
  real_t zalfa;
  {  // FOEALFA (fcttre.func.h:74-75)
    real_t xa = (fmax(kf.v[KF_RTICE], fmin(kf.v[KF_RTWAT], t)) - kf.v[KF_RTICE]) * kf.v[KF_RTWAT_RTICE_R];
    zalfa = fmin(RC(1.0), xa * xa);
  }
  const real_t r4les = ks.v[KS_R4LES], r4ies = ks.v[KS_R4IES], rtt = ks.v[KS_RTT], r3les = ks.v[KS_R3LES],
               r3ies = ks.v[KS_R3IES], r2es = ks.v[KS_R2ES], retv = ks.v[KS_RETV];
  real_t zfoeewl, zfoeewi, zqs, zcor;
  if (P) {
    zfoeewl = r2es * exp(r3les * (t - rtt) / (t - r4les));
    zfoeewi = r2es * exp(r3ies * (t - rtt) / (t - r4ies));
    real_t zfoeew = zalfa * zfoeewl + (RC(1.0) - zalfa) * zfoeewi;
    zqs = zfoeew / pap;
    if (zqs > RC(0.5)) zqs = RC(0.5);
    zcor = RC(1.0) / (RC(1.0) - retv * zqs);
  } else {
    real_t rl, ri, rp;
    c2_rcp3(t - r4les, t - r4ies, pap, rl, ri, rp);
    real_t dt = t - rtt;
    // both branches unconditionally: a lane-divergent skip of the zero-weight exp costs more (the constant block is
    // re-fetched inside every branch) than the ~20 instructions it saves
    zfoeewl = r2es * c2_exp(r3les * dt * rl);
    zfoeewi = r2es * c2_exp(r3ies * dt * ri);
    real_t zfoeew = zalfa * zfoeewl + (RC(1.0) - zalfa) * zfoeewi;
    zqs = zfoeew * rp;
    if (zqs > RC(0.5)) zqs = RC(0.5);
    zcor = c2_rcp(RC(1.0) - retv * zqs);
  }
  return zqs * zcor;
}

// Everything level_tl / level_ad need from the trajectory of one level.  Names follow the "...5" variables of
// cloudsc2tl.F90 / cloudsc2ad.F90 where one exists.
struct LevelTraj {
  // first guess and thermodynamics
  real_t ztp2, zqp2, zl, zi, zdp, zzz, zlfdcp, zlsdcp, zlvdcp;  // ZTP25, ZQP25
  // stage A
  real_t zfwat, zfoeew, zesdp, zfacw, zfaci, zfac, zcor, zdqsdtemp, zcorqs, zqlim, tm4l, tm4i;
  real_t zcosh2r;  // 1/cosh^2(0.17 (T-RLPTRC)) (only set when cold)
  real_t rdp;      // 1/(PAPHP1(JK+1)-PAPHP1(JK))
  real_t rden, rlu, rclc, rcons;  // 1/zden, 1/PLU(JK+1), 1/PCLC, 1/ZCONS
  real_t rdt;      // RD*T
  real_t rl, ri, rtp2, rzsqrt, rlfdcp;  // 1/(T-R4LES), 1/(T-R4IES), 1/T, 1/zsqrt, 1/ZLFDCP
  int cold, esdp_clip, qlim_is_qs;
  // stage B
  real_t zcrh2, zsupsat, zqsat, zqcrit;
  int below_rtice;
  // stage C
  real_t zqt, zqpd, zqcd, zden, zsqrt, zclc, zqc1;
  int regime;  // 0 clear, 1 overcast, 2 partial
  // stage D
  real_t zgdp, zlude, zexpl, clc, zqc2;  // clc = PCLC5 (final)
  int llo1;
  // stage E
  real_t zfac1, zrho, zfac2, zrodqsdp, zldcp, zfac3, dtdzmo, zdqsdz, zfac4, zdqc, zqc3;
  int llo3;
  // stage F
  real_t zqlwc1, zqiwc1, zcondl1, zcondi1;
  // stage G
  real_t covptot_in, covptot1, covpclr1, covpclr;
  int newmax;
  // stage H
  real_t rfl_in, sfl_in, zcons, zz2s, zsnmlt, ztp1;  // ztp1 = ZTP15 (after melting)
  int melt, warm2, melt_all;
  // stage I
  real_t zcldl, zexp3, zdl, zexpdl, zprr, zqlwc, zcldi, zexp1, zexp2, zdi, zexpdi, zprs, zqiwc;
  real_t zdr1, zrfreeze1, zfwatr1, rfln2, sfln2;
  int cloudy, frz1;
  // stage J
  real_t zprtot, zpreclr1, zqe, zbeta, zb, zdtgdp, zdpr1, zdpr, zpreclr, zevapr, zevaps, omc, zsqp;
  int llo2, dpr_clip, reset;
  // stage K
  real_t ztpb, zqpb;  // ZTPB5, ZQPB5 (= ZQOLD5)
  // stage L (two adjustment iterations, cuadjtqs.F90:212-244)
  real_t z3es, z4es, z5alcp, zaldcp, zqp;
  real_t a_t[2], a_q[2], a_foeew[2], a_qsatu[2], a_cor[2], a_qsat[2], a_z2s[2], a_tm4[2], a_den[2], a_rtm4[2], a_rden[2];
  int a_clip[2];
  real_t ztp3, zqp1;  // ZTP35, ZQP15 after adjustment
  // stage M
  real_t zdq, zdr2, zfwatr2, zcondl2, zcondi2, zrfreeze3;
  int dq_pos, frz2;
};

// ---------------------------------------------------------------------------------------------------------
// Trajectory of one level.  cloudsc2.F90:253-279 (first guess) + :343-723.
// Same statements as the reference, with the quotients rewritten on shared reciprocals (see c2_rcp above).
// ---------------------------------------------------------------------------------------------------------
// EVAP is Consts::evap lifted to compile time (the launchers dispatch on it): LEVAPLS2/LDRAIN1D are off in every
// shipped configuration, and with the evaporation branch merely skipped at run time its live ranges still cost
// ~60 VGPRs (one wave per SIMD less in the NL kernel).  With EVAP=false t.llo2 is a compile-time false, which also
// removes the branch's TL and AD statements.
// LIN is `LPHYLIN .OR. LDRAIN1D` (cloudsc2.F90:349), true in every shipped configuration (the three mains force LPHYLIN=.true.,
// dwarf_cloudsc.F90:107) and always for the trajectories of CLOUDSC2TL / CLOUDSC2AD, which have no other form; false selects the
// FOEALFA / FOEEWM form of stage A (:365-369) in the NL sweep.
template <bool P, bool EVAP, bool LIN = true>
C2_HD void level_forward(ConstsP c, const LevelCst& k, const RhCrit& rh, const LevelIn& x, Carry& cy,
                         LevelTraj& t, LevelOut& o) {
  const real_t zqmax = RC(0.5), zeps2 = RC(1.e-10);
  const int rvtmp2_zero = c->rvtmp2_zero;
  const bool evap = EVAP;

  // ---- blocks k0 (first guess, entry reciprocals, saturation pressure) and k1 (dqs/dT, critical RH, convection) ----
  StageBlock k0, k1;
  c2_block2(&c->k0, &c->k1, k0, k1);
  const real_t ptsphy = k0.v[K0_PTSPHY], rtt = k0.v[K0_RTT];

  // first guess (cloudsc2.F90:255-258) and thermodynamic constants (:272-276)
  t.ztp2 = x.t + ptsphy * x.gt;
  t.zqp2 = x.q + ptsphy * x.gq + x.supsat;
  t.zl = x.l + ptsphy * x.gl;
  t.zi = x.i + ptsphy * x.gi;
  t.zdp = x.paph_k1 - x.paph_k;

  // reciprocals known at level entry: 1/(T-R4LES), 1/(T-R4IES), 1/p, 1/dp from ONE v_rcp_f64
  t.tm4l = t.ztp2 - k0.v[K0_R4LES];
  t.tm4i = t.ztp2 - k0.v[K0_R4IES];
  real_t rl = RC(0.0), ri = RC(0.0), rp, rdp;
  if (P) {
    rp = RC(1.0) / x.pap;
    rdp = RC(1.0) / t.zdp;
    rl = RC(1.0) / t.tm4l;  // TL/AD only
    ri = RC(1.0) / t.tm4i;
  } else {
    real_t li = t.tm4l * t.tm4i;
    real_t pd = x.pap * t.zdp;
    real_t r = c2_rcp(li * pd);
    real_t rli = pd * r, rpd = li * r;
    rl = t.tm4i * rli;
    ri = t.tm4l * rli;
    rp = t.zdp * rpd;
    rdp = x.pap * rpd;
  }
  t.zqp = rp;
  t.rdp = rdp;
  t.rl = rl;
  t.ri = ri;

  // A. mixed phase and dqs/dT (cloudsc2.F90:350-375, LPHYLIN branch)
  t.cold = t.ztp2 < rtt;
  if (!LIN) {
    // ZFWAT = FOEALFA(T), ZFOEEW = FOEEWM(T) (cloudsc2.F90:366-367; fcttre.func.h:74-75,81-83); ZESDP is not clipped (:368)
    const StageBlock kf = c2_block(&c->kf);
    const real_t xa = (fmax(kf.v[KF_RTICE], fmin(kf.v[KF_RTWAT], t.ztp2)) - kf.v[KF_RTICE]) * kf.v[KF_RTWAT_RTICE_R];
    t.zfwat = fmin(RC(1.0), xa * xa);
    t.zcosh2r = RC(0.0);
    const real_t dt = t.ztp2 - rtt;
    const real_t el = ex<P>(quot<P>(k0.v[K0_R3LES] * dt, t.tm4l, rl));
    const real_t ei = ex<P>(quot<P>(k0.v[K0_R3IES] * dt, t.tm4i, ri));
    t.zfoeew = k0.v[K0_R2ES] * (t.zfwat * el + (RC(1.0) - t.zfwat) * ei);
  } else {
    real_t z3es, z4es, r4;
    if (P) {
      real_t u = RC(0.17) * (t.ztp2 - k0.v[K0_RLPTRC]);
      real_t ch = cosh(u);  // TL/AD only
      t.zcosh2r = RC(1.0) / (ch * ch);
      real_t zoealfaw = RC(0.545) * (tanh(u) + RC(1.0));
      if (t.cold) { t.zfwat = zoealfaw; z3es = k0.v[K0_R3IES]; z4es = k0.v[K0_R4IES]; }
      else        { t.zfwat = RC(1.0);      z3es = k0.v[K0_R3LES]; z4es = k0.v[K0_R4LES]; }
      r4 = RC(0.0);
    } else if (t.cold) {
      // tanh(u)+1 = 2 e^{2u}/(e^{2u}+1),  1/cosh^2(u) = 4 e^{2u}/(e^{2u}+1)^2,  u = 0.17 (T - RLPTRC)
      real_t e2 = c2_exp(RC(0.34) * (t.ztp2 - k0.v[K0_RLPTRC]));
      real_t re = c2_rcp(e2 + RC(1.0));
      real_t th1 = RC(2.0) * e2 * re;
      t.zcosh2r = RC(2.0) * th1 * re;
      t.zfwat = RC(0.545) * th1;
      z3es = k0.v[K0_R3IES]; z4es = k0.v[K0_R4IES]; r4 = ri;
    } else {
      t.zcosh2r = RC(0.0);  // only read when cold
      t.zfwat = RC(1.0);
      z3es = k0.v[K0_R3LES]; z4es = k0.v[K0_R4LES]; r4 = rl;
    }
    t.zfoeew = k0.v[K0_R2ES] * ex<P>(quot<P>(z3es * (t.ztp2 - rtt), t.ztp2 - z4es, r4));
  }

  const real_t retv = k1.v[K1_RETV], rg = k1.v[K1_RG];
  {
    real_t zesdp1 = quot<P>(t.zfoeew, x.pap, rp);
    t.esdp_clip = LIN && zesdp1 > zqmax;
    t.zesdp = t.esdp_clip ? zqmax : zesdp1;
    t.zfacw = quot<P>(k1.v[K1_R5LES], t.tm4l * t.tm4l, rl * rl);
    t.zfaci = quot<P>(k1.v[K1_R5IES], t.tm4i * t.tm4i, ri * ri);
    t.zfac = t.zfwat * t.zfacw + (RC(1.0) - t.zfwat) * t.zfaci;
    t.rdt = k1.v[K1_RD] * t.ztp2;
    if (P) {
      t.zcor = RC(1.0) / (RC(1.0) - retv * t.zesdp);
      t.zfac1 = RC(1.0) / t.rdt;
      t.zfac2 = RC(1.0) / (x.pap - retv * t.zfoeew);
    } else {
      // 1/(1-RETV*esdp), 1/(RD*T), 1/(p-RETV*es) share one reciprocal (used in A and E)
      c2_rcp3(RC(1.0) - retv * t.zesdp, t.rdt, x.pap - retv * t.zfoeew, t.zcor, t.zfac1, t.zfac2);
    }
    t.rtp2 = t.zfac1 * k1.v[K1_RD];  // 1/T (TL/AD only)
    t.zdqsdtemp = t.zfac * t.zcor * x.qs;
    t.zcorqs = RC(1.0) + k1.v[K1_ZCONS3] * t.zdqsdtemp;
    t.qlim_is_qs = t.zqp2 > x.qs;
    t.zqlim = t.qlim_is_qs ? x.qs : t.zqp2;
  }

  // B. critical relative humidity (cloudsc2.F90:384-407)
  t.zcrh2 = rhcrit_level(rh, k.ceta);
  t.below_rtice = t.ztp2 < k1.v[K1_RTICE];
  t.zsupsat = t.below_rtice ? (RC(1.8) - RC(3.e-03) * t.ztp2) : RC(1.0);
  t.zqsat = x.qs * t.zsupsat;
  t.zqcrit = t.zcrh2 * t.zqsat;

  // C. uniform-PDF cloud cover (cloudsc2.F90:413-426)
  t.zqt = t.zqp2 + t.zl + t.zi;
  t.zqpd = RC(0.0); t.zqcd = RC(0.0); t.zden = RC(1.0); t.zsqrt = RC(1.0); t.rden = RC(1.0); t.rzsqrt = RC(1.0);
  if (t.zqt <= t.zqcrit) {
    t.regime = 0; t.zclc = RC(0.0); t.zqc1 = RC(0.0);
  } else if (t.zqt >= t.zqsat) {
    t.regime = 1; t.zclc = RC(1.0); t.zqc1 = (RC(1.0) - k.zscalm) * (t.zqsat - t.zqcrit);
  } else {
    t.regime = 2;
    t.zqpd = t.zqsat - t.zqt;
    t.zqcd = t.zqsat - t.zqcrit;
    t.zden = t.zqcd - k.zscalm * (t.zqt - t.zqcrit);
    t.rden = recip<P>(t.zden);
    t.zsqrt = sqrt(quot<P>(t.zqpd, t.zden, t.rden));
    t.rzsqrt = recip<P>(t.zsqrt);  // TL/AD only
    t.zclc = RC(1.0) - t.zsqrt;
    t.zqc1 = (k.zscalm * t.zqpd + (RC(1.0) - k.zscalm) * t.zqcd) * (t.zclc * t.zclc);
  }

  // D. convective component (cloudsc2.F90:432-443)
  t.zgdp = quot<P>(rg, x.paph_k1 - x.paph_k, rdp);
  t.zlude = x.lude * ptsphy * t.zgdp;
  t.llo1 = (!k.last) && (t.zlude >= k1.v[K1_RLMIN]) && (x.lu_k1 >= zeps2);
  t.zexpl = RC(1.0); t.rlu = RC(0.0);
  if (t.llo1) {
    t.rlu = recip<P>(x.lu_k1);
    t.zexpl = ex<P>(quot<P>(-t.zlude, x.lu_k1, t.rlu));
    t.clc = t.zclc + (RC(1.0) - t.zclc) * (RC(1.0) - t.zexpl);
    t.zqc2 = t.zqc1 + t.zlude;
  } else {
    t.clc = t.zclc;
    t.zqc2 = t.zqc1;
  }

  // ---- blocks k2, k6: subsidence, condensation rates, melting ----
  StageBlock k2, k6;
  c2_block2(&c->k2, &c->k6, k2, k6);
  if (rvtmp2_zero) {
    t.zzz = k6.v[K6_SPARE0];  // = zzz0
    t.zlvdcp = k6.v[K6_ZLVDCP0]; t.zlsdcp = k6.v[K6_ZLSDCP0]; t.zlfdcp = k6.v[K6_ZLFDCP0];
    t.rlfdcp = k2.v[K2_ZLFDCP0_R];
  } else {
    t.zzz = recip<P>(c->rcpd + c->rcpd * c->rvtmp2 * t.zqp2);
    t.zlfdcp = c->rlmlt * t.zzz; t.zlsdcp = c->rlstt * t.zzz; t.zlvdcp = c->rlvtt * t.zzz;
    t.rlfdcp = recip<P>(t.zlfdcp);
  }

  // E. compensating subsidence (cloudsc2.F90:449-459)
  t.zrho = x.pap * t.zfac1;
  t.zrodqsdp = -t.zrho * x.qs * t.zfac2;
  t.zldcp = t.zfwat * t.zlvdcp + (RC(1.0) - t.zfwat) * t.zlsdcp;
  t.zfac3 = recip<P>(RC(1.0) + t.zldcp * t.zdqsdtemp);
  t.dtdzmo = k2.v[K2_RG] * (k2.v[K2_RCPD_R] - t.zldcp * t.zrodqsdp) * t.zfac3;
  t.zdqsdz = t.zdqsdtemp * t.dtdzmo - k2.v[K2_RG] * t.zrodqsdp;
  t.zfac4 = P ? RC(1.0) / t.zrho : t.rdt * rp;  // 1/rho
  {
    real_t xdq = t.zdqsdz * (x.mfu + x.mfd) * k2.v[K2_PTSPHY] * t.zfac4;
    t.llo3 = xdq < t.zqc2;
    t.zdqc = t.llo3 ? xdq : t.zqc2;
  }
  t.zqc3 = t.zqc2 - t.zdqc;

  // F. condensate partition and condensation rates (cloudsc2.F90:465-468)
  t.zqlwc1 = t.zqc3 * t.zfwat;
  t.zqiwc1 = t.zqc3 * (RC(1.0) - t.zfwat);
  t.zcondl1 = (t.zqlwc1 - t.zl) * k2.v[K2_ZQTMST];
  t.zcondi1 = (t.zqiwc1 - t.zi) * k2.v[K2_ZQTMST];

  // G. maximum overlap of precipitation (cloudsc2.F90:476-480)
  t.covptot_in = cy.covptot;
  t.newmax = t.clc > cy.covptot;
  t.covptot1 = t.newmax ? t.clc : cy.covptot;
  t.covpclr1 = t.covptot1 - t.clc;
  t.covpclr = (t.covpclr1 < RC(0.0)) ? RC(0.0) : t.covpclr1;

  // H. melting of incoming snow (cloudsc2.F90:488-497)
  t.rfl_in = cy.rfl;
  t.sfl_in = cy.sfl;
  t.melt = cy.sfl != RC(0.0);
  real_t rfln, sfln;
  t.zcons = RC(1.0); t.rcons = RC(1.0); t.zz2s = RC(0.0); t.zsnmlt = RC(0.0); t.warm2 = 0; t.melt_all = 0;
  if (t.melt) {
    const real_t zmeltp2 = k2.v[K2_ZMELTP2];
    if (P) t.zcons = k2.v[K2_ZCONS2] * t.zdp / t.zlfdcp;
    else t.zcons = k2.v[K2_ZCONS2] * t.zdp * t.rlfdcp;
    t.warm2 = (t.ztp2 - zmeltp2) > RC(0.0);
    t.zz2s = t.warm2 ? t.zcons * (t.ztp2 - zmeltp2) : RC(0.0);
    t.melt_all = cy.sfl <= t.zz2s;
    t.zsnmlt = t.melt_all ? cy.sfl : t.zz2s;
    rfln = cy.rfl + t.zsnmlt;
    sfln = cy.sfl - t.zsnmlt;
    // dT = -snmlt/zcons = -snmlt * ZLFDCP / (ZCONS2*dp)
    t.rcons = P ? RC(1.0) / t.zcons : t.zlfdcp * (k2.v[K2_ZCONS2_R] * rdp);
    t.ztp1 = t.ztp2 - quot<P>(t.zsnmlt, t.zcons, t.rcons);
  } else {
    rfln = cy.rfl;
    sfln = cy.sfl;
    t.ztp1 = t.ztp2;
  }

  // I. autoconversion to rain and snow (cloudsc2.F90:504-552)
  t.cloudy = t.clc > zeps2;
  real_t zcons2dp = k2.v[K2_ZCONS2] * t.zdp;
  real_t rtt3 = rtt;
  if (t.cloudy) {
    // ---- block k3 ----
    const StageBlock k3 = c2_block(&c->k3);
    rtt3 = k3.v[K3_RTT];
    t.rclc = recip<P>(t.clc);
    t.zcldl = quot<P>(t.zqlwc1, t.clc, t.rclc);
    real_t ql = quot<P>(t.zcldl, k3.v[K3_ZLCRIT_L], k3.v[K3_ZLCRIT_L_R]);
    t.zexp3 = ex<P>(-(ql * ql));
    t.zdl = k3.v[K3_ZCKCODTL] * (RC(1.0) - t.zexp3);
    t.zexpdl = ex<P>(-t.zdl);
    real_t zlnew = t.clc * t.zcldl * t.zexpdl;
    t.zprr = t.zqlwc1 - zlnew;
    t.zqlwc = t.zqlwc1 - t.zprr;

    t.zcldi = quot<P>(t.zqiwc1, t.clc, t.rclc);
    real_t qi = quot<P>(t.zcldi, k3.v[K3_ZLCRIT_I], k3.v[K3_ZLCRIT_I_R]);
    t.zexp1 = ex<P>(RC(0.025) * (t.ztp1 - rtt3));
    t.zexp2 = ex<P>(-(qi * qi));
    t.zdi = k3.v[K3_ZCKCODTI] * t.zexp1 * (RC(1.0) - t.zexp2);
    t.zexpdi = ex<P>(-t.zdi);
    real_t zinew = t.clc * t.zcldi * t.zexpdi;
    t.zprs = t.zqiwc1 - zinew;
    t.zqiwc = t.zqiwc1 - t.zprs;
  } else {
    t.rclc = RC(0.0);
    t.zcldl = RC(0.0); t.zexp3 = RC(1.0); t.zdl = RC(0.0); t.zexpdl = RC(1.0); t.zprr = RC(0.0); t.zqlwc = t.zqlwc1;
    t.zcldi = RC(0.0); t.zexp1 = RC(1.0); t.zexp2 = RC(1.0); t.zdi = RC(0.0); t.zexpdi = RC(1.0); t.zprs = RC(0.0); t.zqiwc = t.zqiwc1;
  }

  // ---- blocks k4 (first guess after the cloud processes, saturation adjustment) and k5 (adjustment, tendencies) ----
  StageBlock k4, k5;
  c2_block2(&c->k4, &c->k5, k4, k5);
  const real_t rtt4 = k4.v[K4_RTT], ptsphy4 = k4.v[K4_PTSPHY], retv4 = k4.v[K4_RETV], r2es4 = k4.v[K4_R2ES];
  t.zdr1 = zcons2dp * (t.zprr + t.zprs);
  t.frz1 = t.ztp1 < rtt4;
  if (t.frz1) { t.zrfreeze1 = zcons2dp * t.zprr; t.zfwatr1 = RC(0.0); }
  else        { t.zrfreeze1 = RC(0.0);               t.zfwatr1 = RC(1.0); }
  rfln = rfln + t.zfwatr1 * t.zdr1;
  sfln = sfln + (RC(1.0) - t.zfwatr1) * t.zdr1;
  t.rfln2 = rfln;
  t.sfln2 = sfln;

  // J. evaporation of precipitation (cloudsc2.F90:556-591); dead unless LEVAPLS2 .OR. LDRAIN1D
  t.zprtot = rfln + sfln;
  t.llo2 = evap && (t.zprtot > zeps2) && (t.covpclr > zeps2);
  real_t covptot = t.covptot1;
  real_t pcovptot = RC(0.0);
  t.zevapr = RC(0.0); t.zevaps = RC(0.0);
  t.dpr_clip = 0; t.reset = 0;
  t.zpreclr1 = RC(0.0); t.zqe = RC(0.0); t.zbeta = RC(0.0); t.zb = RC(0.0); t.zdtgdp = RC(1.0); t.zdpr1 = RC(0.0); t.zdpr = RC(0.0);
  t.zpreclr = RC(0.0); t.omc = RC(1.0); t.zsqp = RC(1.0);
  if (t.llo2) {
    t.omc = RC(1.0) - t.clc;
#if C2_EVAP_FAST
    if (!P) {
      // fast arithmetic (round 5): the block's seven IEEE divisions on three refined v_rcp_f64 (1/covptot1 and 1/omc^2 from one,
      // 1/covpclr, 1/prtot and 1/p_surf from one, 1/(1 + beta dt corqs) alone), the division by ZDTGDP = dt g / dp as a product
      // with dp / (dt g), which the level has anyway, and x^0.5777 as exp(0.5777 ln x) on the branch-free exp -- a few ulp from
      // the reference's order, like the rest of the fast path; the precise mode below keeps that order
      real_t r_cov1, r_omc2, r_clr, r_prtot, r_psurf;
      c2_rcp2(t.covptot1, t.omc * t.omc, r_cov1, r_omc2);
      c2_rcp3(t.covpclr, t.zprtot, x.paph_surf, r_clr, r_prtot, r_psurf);
      t.zpreclr1 = t.zprtot * t.covpclr * r_cov1;
      t.zqe = x.qs - (x.qs - t.zqlim) * t.covpclr * r_omc2;
      t.zsqp = sqrt(x.pap * r_psurf);
      t.zbeta = c->rg * c->rpecons * c2_exp(RC(0.5777) * log(t.zsqp * RC(196.46365422396858) * t.zpreclr1 * r_clr));  // 1 / 5.09e-3
      t.zb = ptsphy4 * t.zbeta * (x.qs - t.zqe) * c2_rcp(RC(1.0) + t.zbeta * ptsphy4 * t.zcorqs);
      t.zdtgdp = c->zcons2_r * t.rdp;
      t.zdpr1 = t.covpclr * t.zb * zcons2dp;
      t.dpr_clip = t.zdpr1 > t.zpreclr1;
      t.zdpr = t.dpr_clip ? t.zpreclr1 : t.zdpr1;
      // ZPRECLR - MIN(ZDPR, ZPRECLR) is EXACTLY zero when everything evaporates, and the reset of the cover hangs on that zero
      // (cloudsc2.F90:578-580): written as a subtraction, the contraction of zpreclr1 = (prtot covpclr) r into an fma leaves the
      // product's rounding error (1e-25) instead and the reset is missed
      t.zpreclr = t.dpr_clip ? RC(0.0) : t.zpreclr1 - t.zdpr1;
      t.reset = t.zpreclr <= RC(0.0);
      if (t.reset) covptot = t.clc;
      pcovptot = covptot;
      const real_t dq = t.zdpr * r_prtot;
      t.zevapr = dq * t.rfln2;
      t.zevaps = dq * t.sfln2;
      rfln = rfln - t.zevapr;
      sfln = sfln - t.zevaps;
    } else
#endif
    {
    t.zpreclr1 = t.zprtot * t.covpclr / t.covptot1;
    t.zqe = x.qs - (x.qs - t.zqlim) * t.covpclr / (t.omc * t.omc);
    t.zsqp = sqrt(x.pap / x.paph_surf);
    t.zbeta = c->rg * c->rpecons * pow(t.zsqp / RC(5.09e-3) * t.zpreclr1 / t.covpclr, RC(0.5777));
    t.zb = ptsphy4 * t.zbeta * (x.qs - t.zqe) / (RC(1.0) + t.zbeta * ptsphy4 * t.zcorqs);
    t.zdtgdp = ptsphy4 * c->rg / (x.paph_k1 - x.paph_k);
    t.zdpr1 = t.covpclr * t.zb / t.zdtgdp;
    t.dpr_clip = t.zdpr1 > t.zpreclr1;
    t.zdpr = t.dpr_clip ? t.zpreclr1 : t.zdpr1;
    t.zpreclr = t.zpreclr1 - t.zdpr;
    t.reset = t.zpreclr <= RC(0.0);
    if (t.reset) covptot = t.clc;
    pcovptot = covptot;
    t.zevapr = t.zdpr * t.rfln2 / t.zprtot;
    rfln = rfln - t.zevapr;
    t.zevaps = t.zdpr * t.sfln2 / t.zprtot;
    sfln = sfln - t.zevaps;
    }
  }

  // K. first-guess T and q after the cloud processes (cloudsc2.F90:602-617)
  const real_t w5 = t.zfwat * t.zlvdcp + (RC(1.0) - t.zfwat) * t.zlsdcp;
  const real_t ev = x.lude + t.zevapr + t.zevaps;
  const real_t lsv = t.zlsdcp - t.zlvdcp;
  {
    real_t zdqdt = -(t.zcondl1 + t.zcondi1) + ev * t.zgdp;
    real_t zdtdt = t.zlvdcp * t.zcondl1 + t.zlsdcp * t.zcondi1 -
                   (t.zlvdcp * t.zevapr + t.zlsdcp * t.zevaps + x.lude * w5 - lsv * t.zrfreeze1) * t.zgdp;
    t.ztpb = t.ztp1 + ptsphy4 * zdtdt;
    t.zqpb = t.zqp2 + ptsphy4 * zdqdt;
  }

  // L. saturation adjustment, two iterations (cloudsc2.F90:630-669 == cuadjtqs.F90:212-244).
  // With a = 1 - RETV*qs:  cond = (q - qs/a) / (1 + qs*z2s/a^2) = (q*a - qs)*a / (a^2 + qs*z2s): one reciprocal
  // for the exp argument and z2s, one for the quotient.  a_cor/a_qsat/a_den (TL/AD only) are derived from them.
  {
    if (t.ztpb > rtt4) { t.z3es = k4.v[K4_R3LES]; t.z4es = k4.v[K4_R4LES]; t.z5alcp = k5.v[K5_R5ALVCP]; t.zaldcp = k5.v[K5_RALVDCP]; }
    else               { t.z3es = k4.v[K4_R3IES]; t.z4es = k4.v[K4_R4IES]; t.z5alcp = k5.v[K5_R5ALSCP]; t.zaldcp = k5.v[K5_RALSDCP]; }
    real_t tt = t.ztpb, qq = t.zqpb;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      t.a_t[it] = tt;
      t.a_q[it] = qq;
      t.a_tm4[it] = tt - t.z4es;
      real_t zcond1;
      if (P) {
        t.a_rtm4[it] = RC(1.0) / t.a_tm4[it];
        t.a_foeew[it] = r2es4 * exp(t.z3es * (tt - rtt4) / t.a_tm4[it]);
        real_t qs1 = t.zqp * t.a_foeew[it];
        t.a_clip[it] = qs1 > zqmax;
        if (t.a_clip[it]) qs1 = zqmax;
        t.a_qsatu[it] = qs1;
        t.a_cor[it] = RC(1.0) / (RC(1.0) - retv4 * qs1);
        t.a_qsat[it] = qs1 * t.a_cor[it];
        t.a_z2s[it] = t.z5alcp / (t.a_tm4[it] * t.a_tm4[it]);
        t.a_den[it] = RC(1.0) + t.a_qsat[it] * t.a_cor[it] * t.a_z2s[it];
        t.a_rden[it] = RC(1.0) / t.a_den[it];
        zcond1 = (qq - t.a_qsat[it]) / t.a_den[it];
      } else {
        real_t r4 = c2_rcp(t.a_tm4[it]);
        t.a_rtm4[it] = r4;
        t.a_foeew[it] = r2es4 * c2_exp(t.z3es * (tt - rtt4) * r4);
        real_t qs1 = t.zqp * t.a_foeew[it];
        t.a_clip[it] = qs1 > zqmax;
        if (t.a_clip[it]) qs1 = zqmax;
        t.a_qsatu[it] = qs1;
        real_t a = RC(1.0) - retv4 * qs1;
        t.a_z2s[it] = t.z5alcp * (r4 * r4);
        real_t d = a * a + qs1 * t.a_z2s[it];
        real_t rd = c2_rcp(d);
        zcond1 = (qq * a - qs1) * a * rd;
        // TL/AD coefficients (dead code in the NL kernel)
        t.a_cor[it] = c2_rcp(a);
        t.a_qsat[it] = qs1 * t.a_cor[it];
        t.a_den[it] = RC(1.0) + t.a_qsat[it] * t.a_cor[it] * t.a_z2s[it];
        t.a_rden[it] = (a * a) * rd;  // = 1/a_den
      }
      tt = tt + t.zaldcp * zcond1;
      qq = qq - zcond1;
    }
    t.ztp3 = tt;
    t.zqp1 = qq;
  }

  // M. extra condensation goes to precipitation; final tendencies (cloudsc2.F90:673-716)
  {
    const real_t zqtmst = k5.v[K5_ZQTMST];
    real_t d = t.zqpb - t.zqp1;
    t.dq_pos = d >= RC(0.0);
    t.zdq = t.dq_pos ? d : RC(0.0);
    t.zdr2 = k5.v[K5_ZCONS2] * t.zdp * t.zdq;
    t.frz2 = t.ztp3 < rtt4;
    real_t zrfreeze2;
    if (t.frz2) { zrfreeze2 = t.zfwat * t.zdr2; t.zfwatr2 = RC(0.0); }
    else        { zrfreeze2 = RC(0.0);              t.zfwatr2 = RC(1.0); }
    t.zcondl2 = t.zcondl1 + t.zfwatr2 * t.zdq * zqtmst;
    t.zcondi2 = t.zcondi1 + (RC(1.0) - t.zfwatr2) * t.zdq * zqtmst;
    rfln = rfln + t.zfwatr2 * t.zdr2;
    sfln = sfln + (RC(1.0) - t.zfwatr2) * t.zdr2;
    t.zrfreeze3 = t.zrfreeze1 + zrfreeze2;

    o.tenq = -(t.zcondl2 + t.zcondi2) + ev * t.zgdp;
    o.tent = t.zlvdcp * t.zcondl2 + t.zlsdcp * t.zcondi2 -
             (t.zlvdcp * t.zevapr + t.zlsdcp * t.zevaps + x.lude * w5 - lsv * t.zrfreeze3) * t.zgdp;
    o.tenl = (t.zqlwc - t.zl) * zqtmst;
    o.teni = (t.zqiwc - t.zi) * zqtmst;
    o.clc = t.clc;
    o.covptot = pcovptot;
    o.fplsl = rfln;
    o.fplsn = sfln;
    o.fhpsl = -rfln * k5.v[K5_RLVTT];
    o.fhpsn = -sfln * k5.v[K5_RLSTT];
  }

  // N. carry (cloudsc2.F90:720-723)
  cy.rfl = rfln;
  cy.sfl = sfln;
  cy.covptot = covptot;
}

// Regularisation of the cloud-fraction perturbation (cloudsc2tl.F90:575-580, cloudsc2ad.F90:1554-1559)
C2_HD real_t regcl_factor(real_t zqpd5, real_t zqcd5, real_t zscalm) {
  real_t zrat = zqpd5 * c2_rcp(zqcd5);
  real_t w = RC(1.0) - zscalm * (RC(1.0) - zrat);
  return fmin(RC(0.3), RC(3.5) * sqrt(zrat * (w * w * w)) * c2_rcp(RC(1.0) - zscalm));
}

// ---------------------------------------------------------------------------------------------------------
// Tangent-linear of one level about LevelTraj.  cloudsc2tl.F90:343-373 (first guess) + :457-1099,
// CUADJTQSTL KCALL=0 (cuadjtqstl.F90:333-405).  dx = perturbation inputs, dcy = perturbation carries.
// ---------------------------------------------------------------------------------------------------------
C2_HD void level_tl(ConstsP c, const LevelCst& k, const LevelIn& x, const LevelTraj& t, const LevelIn& dx,
                    Carry& dcy, LevelOut& dout) {
  // constants of this function from three 64-byte blocks (one scalar-cache wait)
  StageBlock kt0, kt1, kt2;
  c2_block3(&c->kt0, &c->kt1, &c->kt2, kt0, kt1, kt2);
  const real_t ptsphy = kt0.v[KT0_PTSPHY], rtt = kt0.v[KT0_RTT], r3ies = kt0.v[KT0_R3IES], r4ies = kt0.v[KT0_R4IES],
               r3les = kt0.v[KT0_R3LES], r4les = kt0.v[KT0_R4LES], r5les = kt0.v[KT0_R5LES], r5ies = kt0.v[KT0_R5IES];
  const real_t retv = kt1.v[KT1_RETV], zcons3 = kt1.v[KT1_ZCONS3], rg = kt1.v[KT1_RG], zqtmst = kt1.v[KT1_ZQTMST],
               zcons2 = kt1.v[KT1_ZCONS2], zmeltp2 = kt1.v[KT1_ZMELTP2], zlcrit_l_r2 = kt1.v[KT1_ZLCRIT_L_R2],
               zlcrit_i_r2 = kt1.v[KT1_ZLCRIT_I_R2];
  const real_t ck_l = kt2.v[KT2_CK_L], ck_i = kt2.v[KT2_CK_I];
  const int lregcl = c->lregcl, rvtmp2_zero = c->rvtmp2_zero;
  (void)r4ies; (void)r4les; (void)zcons3; (void)zmeltp2; (void)rvtmp2_zero;
  // first guess
  real_t ztp1 = dx.t + ptsphy * dx.gt;
  real_t zqp1 = dx.q + ptsphy * dx.gq + dx.supsat;
  real_t zl = dx.l + ptsphy * dx.gl;
  real_t zi = dx.i + ptsphy * dx.gi;
  real_t zdp = dx.paph_k1 - dx.paph_k;
  real_t zlfdcp = RC(0.0), zlsdcp = RC(0.0), zlvdcp = RC(0.0);
  if (!rvtmp2_zero) {  // cloudsc2tl.F90:366-373
    real_t zzz = -c->rcpd * c->rvtmp2 * zqp1 * (t.zzz * t.zzz);
    zlfdcp = c->rlmlt * zzz; zlsdcp = c->rlstt * zzz; zlvdcp = c->rlvtt * zzz;
  }

  // A (cloudsc2tl.F90:463-501)
  // quotients of the reference are products with the trajectory's reciprocals (LevelTraj::rl, ri, zqp, rdp, ...)
  real_t zfwat, z3es, z4es, r4;
  if (t.cold) { zfwat = RC(0.545) * RC(0.17) * ztp1 * t.zcosh2r; z3es = r3ies; z4es = r4ies; r4 = t.ri; }
  else        { zfwat = RC(0.0);                            z3es = r3les; z4es = r4les; r4 = t.rl; }
  const real_t rp = t.zqp;
  real_t zfoeew = z3es * (rtt - z4es) * ztp1 * t.zfoeew * (r4 * r4);
  real_t zesdp = zfoeew * rp - dx.pap * t.zfoeew * (rp * rp);
  if (t.esdp_clip) zesdp = RC(0.0);
  real_t zfacw = -RC(2.0) * r5les * ztp1 * (t.rl * t.rl * t.rl);
  real_t zfaci = -RC(2.0) * r5ies * ztp1 * (t.ri * t.ri * t.ri);
  real_t zfac = t.zfwat * zfacw + t.zfacw * zfwat + (RC(1.0) - t.zfwat) * zfaci - t.zfaci * zfwat;
  real_t zcor = retv * zesdp * (t.zcor * t.zcor);
  real_t zdqsdtemp = t.zfac * t.zcor * dx.qs + t.zfac * x.qs * zcor + t.zcor * x.qs * zfac;
  real_t zcorqs = zcons3 * zdqsdtemp;
  real_t zqlim = t.qlim_is_qs ? dx.qs : zqp1;

  // B (cloudsc2tl.F90:532-543)
  real_t zsupsat = t.below_rtice ? (-RC(3.e-03) * ztp1) : RC(0.0);
  real_t zqsat = dx.qs * t.zsupsat + x.qs * zsupsat;
  real_t zqcrit = t.zcrh2 * zqsat;

  // C (cloudsc2tl.F90:549-589)
  real_t zqt = zqp1 + zl + zi;
  real_t pclc, zqc;
  if (t.regime == 0) {
    pclc = RC(0.0); zqc = RC(0.0);
  } else if (t.regime == 1) {
    pclc = RC(0.0); zqc = (RC(1.0) - k.zscalm) * (zqsat - zqcrit);
  } else {
    real_t zqpd = zqsat - zqt;
    real_t zqcd = zqsat - zqcrit;
    pclc = -(RC(0.5) * t.rzsqrt) * (zqpd * t.zden - t.zqpd * (zqcd - k.zscalm * (zqt - zqcrit))) * (t.rden * t.rden);
    if (lregcl) pclc = regcl_factor(t.zqpd, t.zqcd, k.zscalm) * pclc;
    zqc = (k.zscalm * zqpd + (RC(1.0) - k.zscalm) * zqcd) * (t.zclc * t.zclc) +
          (k.zscalm * t.zqpd + (RC(1.0) - k.zscalm) * t.zqcd) * RC(2.0) * t.zclc * pclc;
  }

  // D (cloudsc2tl.F90:595-622)
  const real_t rdp2 = t.rdp * t.rdp;
  real_t zgdp = -rg * (dx.paph_k1 - dx.paph_k) * rdp2;
  real_t zlude = ptsphy * t.zgdp * dx.lude + ptsphy * x.lude * zgdp;
  if (t.llo1) {
    pclc = pclc - pclc * (RC(1.0) - t.zexpl) + ((RC(1.0) - t.zclc) * t.rlu) * t.zexpl * zlude -
           ((RC(1.0) - t.zclc) * t.zlude * (t.rlu * t.rlu)) * t.zexpl * dx.lu_k1;
    zqc = zqc + zlude;
  }

  // E (cloudsc2tl.F90:628-664)
  {
    real_t zrho = (dx.pap - ztp1 * x.pap * t.rtp2) * t.zfac1;
    real_t zrodqsdp = (-zrho * x.qs - t.zrho * dx.qs + t.zrho * x.qs * (dx.pap - retv * zfoeew) * t.zfac2) * t.zfac2;
    real_t zldcp = zfwat * t.zlvdcp + t.zfwat * zlvdcp + (RC(1.0) - t.zfwat) * zlsdcp - zfwat * t.zlsdcp;
    real_t dtdzmo = -(rg * (zldcp * t.zrodqsdp + t.zldcp * zrodqsdp) +
                      t.dtdzmo * (t.zldcp * zdqsdtemp + zldcp * t.zdqsdtemp)) * t.zfac3;
    real_t zdqsdz = t.zdqsdtemp * dtdzmo + zdqsdtemp * t.dtdzmo - rg * zrodqsdp;
    real_t zdqc;
    if (t.llo3) {
      zdqc = (ptsphy * (zdqsdz * (x.mfu + x.mfd) + t.zdqsdz * (dx.mfu + dx.mfd)) - t.zdqc * zrho) * t.zfac4;
      if (lregcl) zdqc = zdqc * RC(0.1);
    } else {
      zdqc = zqc;
    }
    zqc = zqc - zdqc;
  }

  // F (cloudsc2tl.F90:670-680)
  real_t zqlwc = zqc * t.zfwat + t.zqc3 * zfwat;
  real_t zqiwc = zqc * (RC(1.0) - t.zfwat) - t.zqc3 * zfwat;
  real_t zcondl = (zqlwc - zl) * zqtmst;
  real_t zcondi = (zqiwc - zi) * zqtmst;

  // G (cloudsc2tl.F90:687-696)
  real_t zcovptot = t.newmax ? pclc : dcy.covptot;
  real_t zcovpclr = zcovptot - pclc;
  if (t.covpclr1 < RC(0.0)) zcovpclr = RC(0.0);

  // H (cloudsc2tl.F90:704-733)
  real_t zrfln, zsfln;
  if (t.melt) {
    real_t zcons = zcons2 * (zdp * t.zlfdcp - t.zdp * zlfdcp) * (t.rlfdcp * t.rlfdcp);
    real_t zz2s = t.warm2 ? (t.zcons * ztp1 + zcons * (t.ztp2 - zmeltp2)) : RC(0.0);
    real_t zsnmlt = t.melt_all ? dcy.sfl : zz2s;
    zrfln = dcy.rfl + zsnmlt;
    zsfln = dcy.sfl - zsnmlt;
    ztp1 = ztp1 - (zsnmlt * t.zcons - zcons * t.zsnmlt) * (t.rcons * t.rcons);
  } else {
    zrfln = dcy.rfl;
    zsfln = dcy.sfl;
  }

  // I (cloudsc2tl.F90:739-840)
  real_t zprr = RC(0.0), zprs = RC(0.0);
  if (t.cloudy) {
    const real_t rclc2 = t.rclc * t.rclc;
    real_t zcldl = zqlwc * t.rclc - t.zqlwc1 * pclc * rclc2;
    real_t ck = ck_l;
    real_t zd = (RC(2.0) * ck * zlcrit_l_r2) * t.zexp3 * t.zcldl * zcldl;
    real_t zlnew = t.zcldl * t.zexpdl * pclc + t.clc * t.zexpdl * zcldl - t.clc * t.zcldl * t.zexpdl * zd;
    zprr = zqlwc - zlnew;
    zqlwc = zqlwc - zprr;

    real_t zcldi = zqiwc * t.rclc - t.zqiwc1 * pclc * rclc2;
    real_t cki = ck_i;
    real_t zdi = cki * t.zexp1 *
                 (t.zexp2 * (RC(2.0) * t.zcldi * zcldi * zlcrit_i_r2 - RC(0.025) * ztp1) + RC(0.025) * ztp1);
    real_t zinew = t.zcldi * t.zexpdi * pclc + t.clc * t.zexpdi * zcldi - t.clc * t.zcldi * t.zexpdi * zdi;
    zprs = zqiwc - zinew;
    zqiwc = zqiwc - zprs;
  }
  real_t zdr = zcons2 * (t.zdp * (zprr + zprs) + zdp * (t.zprr + t.zprs));
  real_t zrfreeze = RC(0.0);
  if (t.frz1) zrfreeze = zcons2 * (zdp * t.zprr + t.zdp * zprr);
  zrfln = zrfln + t.zfwatr1 * zdr;
  zsfln = zsfln + (RC(1.0) - t.zfwatr1) * zdr;

  // J (cloudsc2tl.F90:844-936)
  real_t zevapr = RC(0.0), zevaps = RC(0.0), pcovptot = RC(0.0);
  if (t.llo2) {
    real_t zprtot = zrfln + zsfln;
#if C2_EVAP_FAST_TLAD
    // The block's seventeen divisions, its pow() and its sqrt() on six reciprocals (two refined v_rcp_f64), and the tangent of
    // ZBETA = RG RPECONS (B / 5.09e-3)^0.5777, B = sqrt(p / p_surf) ZPRECLR / ZCOVPCLR, as 0.5777 ZBETA5 dB / B with
    //   dB / B = dZPRECLR / ZPRECLR5 + dp / (2 p) - dp_surf / (2 p_surf) - dZCOVPCLR / ZCOVPCLR5
    // -- the same linear map as cloudsc2tl.F90:871-882, with the power and the root cancelled instead of evaluated.
    real_t r_cov1, r_omc, r_clr, r_prtot, r_psurf, r_den;
    const real_t den = RC(1.0) + t.zbeta * ptsphy * t.zcorqs;
    c2_rcp3(t.covptot1, t.omc, t.covpclr, r_cov1, r_omc, r_clr);
    c2_rcp3(t.zprtot, x.paph_surf, den, r_prtot, r_psurf, r_den);
    const real_t r_omc2 = r_omc * r_omc, iz = zcons2 * t.zdp;  // iz = 1 / ZDTGDP5
    real_t zpreclr = (t.zprtot * zcovpclr + t.covpclr * zprtot) * r_cov1 - t.zprtot * t.covpclr * zcovptot * (r_cov1 * r_cov1);
    real_t zqe = dx.qs - ((x.qs - t.zqlim) * zcovpclr + t.covpclr * dx.qs - t.covpclr * zqlim) * r_omc2 -
                 RC(2.0) * (x.qs - t.zqlim) * t.covpclr * pclc * (r_omc2 * r_omc);
    real_t zbeta = RC(0.5777) * t.zbeta * (zpreclr * (t.covptot1 * r_prtot * r_clr) + RC(0.5) * dx.pap * t.zqp -
                                           RC(0.5) * dx.paph_surf * r_psurf - zcovpclr * r_clr);
    real_t zb = ptsphy * ((x.qs - t.zqe) * zbeta + t.zbeta * dx.qs - t.zbeta * zqe) * r_den -
                (ptsphy * ptsphy) * t.zbeta * (x.qs - t.zqe) * (t.zbeta * zcorqs + t.zcorqs * zbeta) * (r_den * r_den);
    real_t zdtgdp = -ptsphy * rg * (dx.paph_k1 - dx.paph_k) * rdp2;
    real_t zdpr = (t.covpclr * zb + t.zb * zcovpclr) * iz - t.covpclr * t.zb * zdtgdp * (iz * iz);
    if (t.dpr_clip) zdpr = zpreclr;
    zpreclr = zpreclr - zdpr;
    if (t.reset) zcovptot = pclc;
    pcovptot = zcovptot;
    const real_t q2 = t.zdpr * zprtot * (r_prtot * r_prtot);
    zevapr = (t.zdpr * zrfln + t.rfln2 * zdpr) * r_prtot - t.rfln2 * q2;
    zevaps = (t.zdpr * zsfln + t.sfln2 * zdpr) * r_prtot - t.sfln2 * q2;
#else
    real_t zpreclr = (t.zprtot * zcovpclr + t.covpclr * zprtot) / t.covptot1 -
                     t.zprtot * t.covpclr * zcovptot / (t.covptot1 * t.covptot1);
    real_t omc2 = t.omc * t.omc;
    real_t zqe = dx.qs - ((x.qs - t.zqlim) * zcovpclr + t.covpclr * dx.qs - t.covpclr * zqlim) / omc2 -
                 RC(2.0) * (x.qs - t.zqlim) * t.covpclr * pclc / (omc2 * t.omc);
    real_t zbeta = RC(0.5777) * (rg * c->rpecons / RC(5.09e-3)) *
                   pow(RC(5.09e-3) * t.covpclr / (t.zpreclr1 * t.zsqp), RC(0.4223)) *
                   ((t.zsqp * zpreclr + RC(0.5) * t.zpreclr1 * dx.pap / sqrt(x.pap * x.paph_surf) -
                     RC(0.5) * t.zpreclr1 * t.zsqp * dx.paph_surf / x.paph_surf) / t.covpclr -
                    t.zpreclr1 * t.zsqp * zcovpclr / (t.covpclr * t.covpclr));
    real_t den = RC(1.0) + t.zbeta * ptsphy * t.zcorqs;
    real_t zb = ptsphy * ((x.qs - t.zqe) * zbeta + t.zbeta * dx.qs - t.zbeta * zqe) / den -
                (ptsphy * ptsphy) * t.zbeta * (x.qs - t.zqe) * (t.zbeta * zcorqs + t.zcorqs * zbeta) / (den * den);
    real_t zdtgdp = -ptsphy * rg * (dx.paph_k1 - dx.paph_k) * rdp2;
    real_t zdpr = (t.covpclr * zb + t.zb * zcovpclr) / t.zdtgdp - t.covpclr * t.zb * zdtgdp / (t.zdtgdp * t.zdtgdp);
    if (t.dpr_clip) zdpr = zpreclr;
    zpreclr = zpreclr - zdpr;
    if (t.reset) zcovptot = pclc;
    pcovptot = zcovptot;
    zevapr = (t.zdpr * zrfln + t.rfln2 * zdpr) / t.zprtot - t.zdpr * t.rfln2 * zprtot / (t.zprtot * t.zprtot);
    zevaps = (t.zdpr * zsfln + t.sfln2 * zdpr) / t.zprtot - t.zdpr * t.sfln2 * zprtot / (t.zprtot * t.zprtot);
#endif
    zrfln = zrfln - zevapr;
    zsfln = zsfln - zevaps;
  }

  // K (cloudsc2tl.F90:943-982)
  real_t w5 = t.zfwat * t.zlvdcp + (RC(1.0) - t.zfwat) * t.zlsdcp;
  real_t w = zfwat * (t.zlvdcp - t.zlsdcp) + (t.zfwat * zlvdcp + (RC(1.0) - t.zfwat) * zlsdcp);
  {
    real_t zdqdt = -(zcondl + zcondi) + (dx.lude + zevapr + zevaps) * t.zgdp + (x.lude + t.zevapr + t.zevaps) * zgdp;
    real_t zdtdt = zlvdcp * t.zcondl1 + zlsdcp * t.zcondi1 + t.zlvdcp * zcondl + t.zlsdcp * zcondi -
                   (zlvdcp * t.zevapr + zlsdcp * t.zevaps + t.zlvdcp * zevapr + t.zlsdcp * zevaps +
                    dx.lude * w5 + x.lude * w - (zlsdcp - zlvdcp) * t.zrfreeze1 -
                    (t.zlsdcp - t.zlvdcp) * zrfreeze) * t.zgdp -
                   (t.zlvdcp * t.zevapr + t.zlsdcp * t.zevaps + x.lude * w5 -
                    (t.zlsdcp - t.zlvdcp) * t.zrfreeze1) * zgdp;
    ztp1 = ztp1 + ptsphy * zdtdt;
    zqp1 = zqp1 + ptsphy * zdqdt;
  }
  real_t zqold = zqp1;

  // L (cuadjtqstl.F90:339-402)
  {
    real_t zqp = -dx.pap * (t.zqp * t.zqp);
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const real_t r4a = t.a_rtm4[it], r4a2 = r4a * r4a;
      real_t dfoeew = t.z3es * (rtt - t.z4es) * ztp1 * t.a_foeew[it] * r4a2;
      real_t dqsat = t.zqp * dfoeew + zqp * t.a_foeew[it];
      if (t.a_clip[it]) dqsat = RC(0.0);
      real_t dcor = (retv * dqsat) * (t.a_cor[it] * t.a_cor[it]);
      dqsat = t.a_qsatu[it] * dcor + dqsat * t.a_cor[it];
      real_t dz2s = -RC(2.0) * ztp1 * t.z5alcp * (r4a2 * r4a);
      const real_t rden = t.a_rden[it];
      real_t dcond = (zqp1 - dqsat) * rden -
                     (t.a_q[it] - t.a_qsat[it]) *
                         (dqsat * t.a_cor[it] * t.a_z2s[it] + t.a_qsat[it] * dcor * t.a_z2s[it] +
                          t.a_qsat[it] * t.a_cor[it] * dz2s) * (rden * rden);
      ztp1 = ztp1 + t.zaldcp * dcond;
      zqp1 = zqp1 - dcond;
    }
  }

  // M (cloudsc2tl.F90:994-1091)
  {
    real_t zdq = RC(0.0);
    if (t.dq_pos) {
      zdq = zqold - zqp1;
      if (lregcl) zdq = zdq * RC(0.7);
    }
    real_t zdr2 = zcons2 * (t.zdp * zdq + t.zdq * zdp);
    real_t zrfreeze2 = RC(0.0);
    if (t.frz2) zrfreeze2 = zfwat * t.zdr2 + t.zfwat * zdr2;
    zcondl = zcondl + (t.zfwatr2 * zdq) * zqtmst;
    zcondi = zcondi + ((RC(1.0) - t.zfwatr2) * zdq) * zqtmst;
    zrfln = zrfln + t.zfwatr2 * zdr2;
    zsfln = zsfln + (RC(1.0) - t.zfwatr2) * zdr2;
    zrfreeze = zrfreeze + zrfreeze2;

    dout.tenq = -(zcondl + zcondi) + (dx.lude + zevapr + zevaps) * t.zgdp + (x.lude + t.zevapr + t.zevaps) * zgdp;
    dout.tent = zlvdcp * t.zcondl2 + zlsdcp * t.zcondi2 + t.zlvdcp * zcondl + t.zlsdcp * zcondi -
                (zlvdcp * t.zevapr + zlsdcp * t.zevaps + t.zlvdcp * zevapr + t.zlsdcp * zevaps +
                 dx.lude * w5 + x.lude * w - (zlsdcp - zlvdcp) * t.zrfreeze3 -
                 (t.zlsdcp - t.zlvdcp) * zrfreeze) * t.zgdp -
                (t.zlvdcp * t.zevapr + t.zlsdcp * t.zevaps + x.lude * w5 -
                 (t.zlsdcp - t.zlvdcp) * t.zrfreeze3) * zgdp;
    dout.tenl = (zqlwc - zl) * zqtmst;
    dout.teni = (zqiwc - zi) * zqtmst;
    dout.clc = pclc;
    dout.covptot = pcovptot;
    dout.fplsl = zrfln;
    dout.fplsn = zsfln;
    dout.fhpsl = -zrfln * kt2.v[KT2_RLVTT];  // cloudsc2tl.F90:1108-1111
    dout.fhpsn = -zsfln * kt2.v[KT2_RLSTT];
  }

  dcy.rfl = zrfln;
  dcy.sfl = zsfln;
  dcy.covptot = zcovptot;
}

// ---------------------------------------------------------------------------------------------------------
// Adjoint of one level.  cloudsc2ad.F90:940-1666 + the per-level parts of its epilogue (:1701-1738),
// CUADJTQSAD KCALL=0 (cuadjtqsad.F90:542-641).
//   ya   : output adjoints of this level (tent,tenq,tenl,teni,clc,covptot and fplsl/fplsn at JK+1 with the
//          enthalpy-flux adjoints already folded in, cloudsc2ad.F90:914-921)
//   acy  : adjoint carries.  In: adjoints of (rfl,sfl,covptot) leaving this level downward.
//          Out: adjoints of the values entering this level from above.
//   ax   : adjoint increments of the level's inputs (to be added to the input-adjoint arrays);
//          ax.paph_k / ax.paph_k1 / ax.lu_k1 / ax.paph_surf are contributions to neighbouring indices.
// ---------------------------------------------------------------------------------------------------------
C2_HD void level_ad(ConstsP c, const LevelCst& k, const LevelIn& x, const LevelTraj& t, const LevelOut& ya,
                    Carry& acy, LevelIn& ax) {
  // constants of this function from three 64-byte blocks (one scalar-cache wait)
  StageBlock kt0, kt1, kt2;
  c2_block3(&c->kt0, &c->kt1, &c->kt2, kt0, kt1, kt2);
  const real_t ptsphy = kt0.v[KT0_PTSPHY], rtt = kt0.v[KT0_RTT], r3ies = kt0.v[KT0_R3IES], r4ies = kt0.v[KT0_R4IES],
               r3les = kt0.v[KT0_R3LES], r4les = kt0.v[KT0_R4LES], r5les = kt0.v[KT0_R5LES], r5ies = kt0.v[KT0_R5IES];
  const real_t retv = kt1.v[KT1_RETV], zcons3 = kt1.v[KT1_ZCONS3], rg = kt1.v[KT1_RG], zqtmst = kt1.v[KT1_ZQTMST],
               zcons2 = kt1.v[KT1_ZCONS2], zmeltp2 = kt1.v[KT1_ZMELTP2], zlcrit_l_r2 = kt1.v[KT1_ZLCRIT_L_R2],
               zlcrit_i_r2 = kt1.v[KT1_ZLCRIT_I_R2];
  const real_t ck_l = kt2.v[KT2_CK_L], ck_i = kt2.v[KT2_CK_I];
  const int lregcl = c->lregcl, rvtmp2_zero = c->rvtmp2_zero;
  (void)r4ies; (void)r4les; (void)zcons3; (void)zmeltp2; (void)rvtmp2_zero;
  // adjoints of level-local quantities
  real_t a_tp1 = RC(0.0), a_qp1 = RC(0.0), a_l = RC(0.0), a_i = RC(0.0), a_dp = RC(0.0);
  real_t a_lvdcp = RC(0.0), a_lsdcp = RC(0.0), a_lfdcp = RC(0.0);
  real_t a_qlwc = RC(0.0), a_qiwc = RC(0.0), a_condl = RC(0.0), a_condi = RC(0.0), a_evapr = RC(0.0), a_evaps = RC(0.0);
  real_t a_rfreeze = RC(0.0), a_gdp = RC(0.0), a_fwat = RC(0.0), a_clc = ya.clc, a_lude_in = RC(0.0);
  real_t a_pap = RC(0.0), a_qs = RC(0.0), a_mfu = RC(0.0), a_mfd = RC(0.0), a_lu_k1 = RC(0.0), a_paph_k = RC(0.0), a_paph_k1 = RC(0.0);
  real_t a_paph_surf = RC(0.0);
  real_t a_covptot = acy.covptot, a_covpclr = RC(0.0), a_qlim = RC(0.0), a_corqs = RC(0.0), a_dqsdtemp = RC(0.0);

  const real_t w5 = t.zfwat * t.zlvdcp + (RC(1.0) - t.zfwat) * t.zlsdcp;

  // fluxes leaving the level (cloudsc2ad.F90:941-957)
  real_t a_sfln = acy.sfl + ya.fplsn;
  real_t a_rfln = acy.rfl + ya.fplsl;

  // M^T: final tendencies (cloudsc2ad.F90:959-1012)
  {
    real_t zdidt = ya.teni, zdldt = ya.tenl, zdtdt = ya.tent, zdqdt = ya.tenq;
    a_i -= zqtmst * zdidt;  a_qiwc += zqtmst * zdidt;
    a_l -= zqtmst * zdldt;  a_qlwc += zqtmst * zdldt;
    a_gdp -= zdtdt * (t.zlvdcp * t.zevapr + t.zlsdcp * t.zevaps + x.lude * w5 - (t.zlsdcp - t.zlvdcp) * t.zrfreeze3);
    a_condl += zdtdt * t.zlvdcp;
    a_condi += zdtdt * t.zlsdcp;
    a_evapr -= zdtdt * t.zlvdcp * t.zgdp;
    a_evaps -= zdtdt * t.zlsdcp * t.zgdp;
    a_lvdcp += zdtdt * (t.zcondl2 - t.zevapr * t.zgdp);
    a_lsdcp += zdtdt * (t.zcondi2 - t.zevaps * t.zgdp);
    a_lude_in -= zdtdt * t.zgdp * w5;
    a_lvdcp -= zdtdt * x.lude * t.zgdp * t.zfwat;
    a_lsdcp -= zdtdt * x.lude * t.zgdp * (RC(1.0) - t.zfwat);
    a_fwat -= zdtdt * x.lude * t.zgdp * (t.zlvdcp - t.zlsdcp);
    a_lsdcp += zdtdt * t.zrfreeze3 * t.zgdp;
    a_lvdcp -= zdtdt * t.zrfreeze3 * t.zgdp;
    a_rfreeze += zdtdt * (t.zlsdcp - t.zlvdcp) * t.zgdp;
    a_gdp += zdqdt * (x.lude + t.zevapr + t.zevaps);
    a_lude_in += zdqdt * t.zgdp;
    a_evapr += zdqdt * t.zgdp;
    a_evaps += zdqdt * t.zgdp;
    a_condl -= zdqdt;
    a_condi -= zdqdt;
  }

  // M^T: extra condensation to precipitation (cloudsc2ad.F90:1017-1063)
  real_t a_qold = RC(0.0);
  {
    real_t zrfreeze2 = a_rfreeze;
    real_t zsn = a_sfln, zrn = a_rfln;
    real_t a_dq = a_condi * (RC(1.0) - t.zfwatr2) * zqtmst + a_condl * t.zfwatr2 * zqtmst;
    real_t zdr2 = (RC(1.0) - t.zfwatr2) * zsn + t.zfwatr2 * zrn;
    if (t.frz2) {
      a_fwat += t.zdr2 * zrfreeze2;
      zdr2 += t.zfwat * zrfreeze2;
    }
    a_dq += zcons2 * t.zdp * zdr2;
    a_dp += zcons2 * t.zdq * zdr2;
    if (t.dq_pos) {
      if (lregcl) a_dq *= RC(0.7);
      a_qold += a_dq;
      a_qp1 -= a_dq;
    }
  }

  // L^T: saturation adjustment (cuadjtqsad.F90:542-641)
  {
    real_t a_zqp = RC(0.0);
#pragma unroll
    for (int it = 1; it >= 0; --it) {
      const real_t rden = t.a_rden[it];
      const real_t r4a = t.a_rtm4[it], r4a2 = r4a * r4a;
      real_t zcond1 = -a_qp1 + t.zaldcp * a_tp1;
      a_qp1 += zcond1 * rden;
      real_t zqsat = -zcond1 * rden;
      real_t dqmq = (t.a_q[it] - t.a_qsat[it]) * (rden * rden);
      zqsat -= zcond1 * dqmq * t.a_cor[it] * t.a_z2s[it];
      real_t zcor = -zcond1 * dqmq * t.a_qsat[it] * t.a_z2s[it];
      real_t z2s = -zcond1 * dqmq * t.a_qsat[it] * t.a_cor[it];
      real_t ztarg = -RC(2.0) * z2s * t.z5alcp * (r4a2 * r4a);
      zcor += zqsat * t.a_qsatu[it];
      zqsat = zqsat * t.a_cor[it];
      zqsat += zcor * retv * (t.a_cor[it] * t.a_cor[it]);
      if (t.a_clip[it]) zqsat = RC(0.0);
      real_t zfoeew = zqsat * t.zqp;
      a_zqp += zqsat * t.a_foeew[it];
      ztarg += zfoeew * t.z3es * (rtt - t.z4es) * t.a_foeew[it] * r4a2;
      a_tp1 += ztarg;
    }
    a_pap -= a_zqp * (t.zqp * t.zqp);
  }

  // K^T: first-guess T and q (cloudsc2ad.F90:1076-1125)
  {
    a_qp1 += a_qold;
    real_t zdqdt = ptsphy * a_qp1;
    real_t zdtdt = ptsphy * a_tp1;
    a_gdp -= zdtdt * (t.zlvdcp * t.zevapr + t.zlsdcp * t.zevaps + x.lude * w5 - (t.zlsdcp - t.zlvdcp) * t.zrfreeze1);
    a_condl += zdtdt * t.zlvdcp;
    a_condi += zdtdt * t.zlsdcp;
    a_evapr -= zdtdt * t.zlvdcp * t.zgdp;
    a_evaps -= zdtdt * t.zlsdcp * t.zgdp;
    a_lvdcp += zdtdt * (t.zcondl1 - t.zevapr * t.zgdp);
    a_lsdcp += zdtdt * (t.zcondi1 - t.zevaps * t.zgdp);
    a_lude_in -= zdtdt * t.zgdp * w5;
    a_lvdcp -= zdtdt * x.lude * t.zgdp * t.zfwat;
    a_lsdcp -= zdtdt * x.lude * t.zgdp * (RC(1.0) - t.zfwat);
    a_fwat -= zdtdt * x.lude * t.zgdp * (t.zlvdcp - t.zlsdcp);
    a_lsdcp += zdtdt * t.zrfreeze1 * t.zgdp;
    a_lvdcp -= zdtdt * t.zrfreeze1 * t.zgdp;
    a_rfreeze += zdtdt * (t.zlsdcp - t.zlvdcp) * t.zgdp;
    a_gdp += zdqdt * (x.lude + t.zevapr + t.zevaps);
    a_lude_in += zdqdt * t.zgdp;
    a_evapr += zdqdt * t.zgdp;
    a_evaps += zdqdt * t.zgdp;
    a_condl -= zdqdt;
    a_condi -= zdqdt;
  }

  // J^T: evaporation of precipitation (cloudsc2ad.F90:1152-1261)
  real_t a_prtot = RC(0.0);
  if (t.llo2) {
    real_t zdpr = RC(0.0), zpreclr = RC(0.0), zb = RC(0.0), zbeta = RC(0.0), zqe = RC(0.0), a_dtgdp = RC(0.0);
#if C2_EVAP_FAST_TLAD
    // the transpose of level_tl's fast block, term by term: six reciprocals from two refined v_rcp_f64, 1 / ZDTGDP5 = dp / (dt g),
    // and the adjoint of ZBETA through dB / B (w = 0.5777 ZBETA5 zbeta*), the power and the root cancelled
    real_t r_cov1, r_omc, r_clr, r_prtot, r_psurf, r_den;
    const real_t den = RC(1.0) + t.zbeta * ptsphy * t.zcorqs;
    c2_rcp3(t.covptot1, t.omc, t.covpclr, r_cov1, r_omc, r_clr);
    c2_rcp3(t.zprtot, x.paph_surf, den, r_prtot, r_psurf, r_den);
    const real_t r_omc2 = r_omc * r_omc, iz = zcons2 * t.zdp, r_prtot2 = r_prtot * r_prtot;
    // ice proportion
    a_evaps -= a_sfln;
    a_sfln += t.zdpr * a_evaps * r_prtot;
    zdpr += t.sfln2 * a_evaps * r_prtot;
    a_prtot -= t.zdpr * t.sfln2 * a_evaps * r_prtot2;
    // warm proportion
    a_evapr -= a_rfln;
    a_rfln += t.zdpr * a_evapr * r_prtot;
    zdpr += t.rfln2 * a_evapr * r_prtot;
    a_prtot -= t.zdpr * t.rfln2 * a_evapr * r_prtot2;
    // clear-sky flux
    a_covptot += ya.covptot;
    if (t.reset) { a_clc += a_covptot; a_covptot = RC(0.0); }
    zdpr -= zpreclr;
    if (t.dpr_clip) { zpreclr += zdpr; zdpr = RC(0.0); }
    zb += t.covpclr * zdpr * iz;
    a_covpclr += t.zb * zdpr * iz;
    a_dtgdp -= t.covpclr * t.zb * zdpr * (iz * iz);
    {
      real_t g = ptsphy * rg * a_dtgdp * (t.rdp * t.rdp);
      a_paph_k1 -= g;
      a_paph_k += g;
    }
    // implicit solution
    const real_t pz = ptsphy * zb * r_den;
    zbeta += (x.qs - t.zqe) * pz;
    a_qs += t.zbeta * pz;
    zqe -= t.zbeta * pz;
    const real_t p2 = ptsphy * t.zbeta * (x.qs - t.zqe) * pz * r_den;
    a_corqs -= p2 * t.zbeta;
    zbeta -= p2 * t.zcorqs;
    // zbeta
    const real_t w = RC(0.5777) * t.zbeta * zbeta;
    zpreclr += w * (t.covptot1 * r_prtot * r_clr);
    a_pap += RC(0.5) * w * t.zqp;
    a_paph_surf -= RC(0.5) * w * r_psurf;
    a_covpclr -= w * r_clr;
    // zqe
    a_qs += zqe;
    a_covpclr -= (x.qs - t.zqlim) * zqe * r_omc2;
    a_qs -= t.covpclr * zqe * r_omc2;
    a_qlim += t.covpclr * zqe * r_omc2;
    a_clc -= RC(2.0) * (x.qs - t.zqlim) * t.covpclr * zqe * (r_omc2 * r_omc);
    // zpreclr
    a_covpclr += t.zprtot * zpreclr * r_cov1;
    a_prtot += t.covpclr * zpreclr * r_cov1;
    a_covptot -= t.zprtot * t.covpclr * zpreclr * (r_cov1 * r_cov1);
#else
    // ice proportion
    a_evaps -= a_sfln;
    a_sfln += t.zdpr * a_evaps / t.zprtot;
    zdpr += t.sfln2 * a_evaps / t.zprtot;
    a_prtot -= t.zdpr * t.sfln2 * a_evaps / (t.zprtot * t.zprtot);
    // warm proportion
    a_evapr -= a_rfln;
    a_rfln += t.zdpr * a_evapr / t.zprtot;
    zdpr += t.rfln2 * a_evapr / t.zprtot;
    a_prtot -= t.zdpr * t.rfln2 * a_evapr / (t.zprtot * t.zprtot);
    // clear-sky flux
    a_covptot += ya.covptot;
    if (t.reset) { a_clc += a_covptot; a_covptot = RC(0.0); }
    zdpr -= zpreclr;
    if (t.dpr_clip) { zpreclr += zdpr; zdpr = RC(0.0); }
    zb += t.covpclr * zdpr / t.zdtgdp;
    a_covpclr += t.zb * zdpr / t.zdtgdp;
    a_dtgdp -= t.covpclr * t.zb * zdpr / (t.zdtgdp * t.zdtgdp);
    {
      real_t g = ptsphy * rg * a_dtgdp * (t.rdp * t.rdp);
      a_paph_k1 -= g;
      a_paph_k += g;
    }
    // implicit solution
    real_t den = RC(1.0) + t.zbeta * ptsphy * t.zcorqs;
    zbeta += ptsphy * (x.qs - t.zqe) * zb / den;
    a_qs += ptsphy * t.zbeta * zb / den;
    zqe -= ptsphy * t.zbeta * zb / den;
    a_corqs -= (ptsphy * ptsphy) * t.zbeta * (x.qs - t.zqe) * t.zbeta * zb / (den * den);
    zbeta -= (ptsphy * ptsphy) * t.zbeta * (x.qs - t.zqe) * t.zcorqs * zb / (den * den);
    // zbeta
    real_t zxx = RC(0.5777) * (rg * c->rpecons / RC(5.09e-3)) * pow(RC(5.09e-3) * t.covpclr / (t.zpreclr1 * t.zsqp), RC(0.4223));
    zpreclr += zxx * t.zsqp * zbeta / t.covpclr;
    a_pap += (zxx * RC(0.5) * t.zpreclr1 * zbeta / sqrt(x.pap * x.paph_surf)) / t.covpclr;
    a_paph_surf -= (zxx * RC(0.5) * t.zpreclr1 * t.zsqp * zbeta / x.paph_surf) / t.covpclr;
    a_covpclr -= zxx * t.zpreclr1 * t.zsqp * zbeta / (t.covpclr * t.covpclr);
    // zqe
    real_t omc2 = t.omc * t.omc;
    a_qs += zqe;
    a_covpclr -= (x.qs - t.zqlim) * zqe / omc2;
    a_qs -= t.covpclr * zqe / omc2;
    a_qlim += t.covpclr * zqe / omc2;
    a_clc -= RC(2.0) * (x.qs - t.zqlim) * t.covpclr * zqe / (omc2 * t.omc);
    // zpreclr
    a_covpclr += t.zprtot * zpreclr / t.covptot1;
    a_prtot += t.covpclr * zpreclr / t.covptot1;
    a_covptot -= t.zprtot * t.covpclr * zpreclr / (t.covptot1 * t.covptot1);
#endif
    a_evapr = RC(0.0);
    a_evaps = RC(0.0);
  }

  // I^T: new precipitation and autoconversion (cloudsc2ad.F90:1265-1356)
  {
    a_rfln += a_prtot;
    a_sfln += a_prtot;
    real_t zdr = (RC(1.0) - t.zfwatr1) * a_sfln + t.zfwatr1 * a_rfln;
    real_t zprr = RC(0.0), zprs = RC(0.0);
    if (t.frz1) {
      a_dp += a_rfreeze * zcons2 * t.zprr;
      zprr += a_rfreeze * zcons2 * t.zdp;
    }
    zprr += zcons2 * t.zdp * zdr;
    zprs += zcons2 * t.zdp * zdr;
    a_dp += zcons2 * (t.zprr + t.zprs) * zdr;
    if (t.cloudy) {
      // ice
      zprs -= a_qiwc;
      a_qiwc += zprs;
      real_t zinew = -zprs;
      a_clc += zinew * t.zcldi * t.zexpdi;
      real_t zcldi = zinew * t.clc * t.zexpdi;
      real_t zdi = -zinew * t.clc * t.zcldi * t.zexpdi;
      real_t cki = ck_i;
      a_tp1 += cki * t.zexp1 * (RC(1.0) - t.zexp2) * RC(0.025) * zdi;
      zcldi += (cki * t.zexp1 * t.zexp2 * RC(2.0) * t.zcldi * zlcrit_i_r2) * zdi;
      a_qiwc += zcldi * t.rclc;
      a_clc -= t.zqiwc1 * zcldi * (t.rclc * t.rclc);
      // liquid
      zprr -= a_qlwc;
      a_qlwc += zprr;
      real_t zlnew = -zprr;
      a_clc += zlnew * t.zcldl * t.zexpdl;
      real_t zcldl = zlnew * t.clc * t.zexpdl;
      real_t zdl = -zlnew * t.clc * t.zcldl * t.zexpdl;
      real_t ck = ck_l;
      zcldl += (RC(2.0) * ck * zlcrit_l_r2) * t.zexp3 * t.zcldl * zdl;
      a_qlwc += zcldl * t.rclc;
      a_clc -= t.zqlwc1 * zcldl * (t.rclc * t.rclc);
    }
  }

  // H^T: melting of incoming snow (cloudsc2ad.F90:1362-1400)
  real_t a_sfl = RC(0.0), a_rfl = RC(0.0);
  if (t.melt) {
    real_t zsnmlt = -a_tp1 * t.rcons;
    real_t zcons = (a_tp1 * t.zsnmlt) * (t.rcons * t.rcons);
    a_sfl += a_sfln;
    zsnmlt -= a_sfln;
    a_rfl += a_rfln;
    zsnmlt += a_rfln;
    real_t zz2s = RC(0.0);
    if (t.melt_all) a_sfl += zsnmlt; else zz2s += zsnmlt;
    if (t.warm2) {
      a_tp1 += t.zcons * zz2s;
      zcons += (t.ztp2 - zmeltp2) * zz2s;
    }
    a_dp += zcons2 * zcons * t.rlfdcp;
    a_lfdcp -= zcons2 * t.zdp * zcons * (t.rlfdcp * t.rlfdcp);
  } else {
    a_sfl += a_sfln;
    a_rfl += a_rfln;
  }

  // G^T: overlap (cloudsc2ad.F90:1407-1418)
  if (t.covpclr1 < RC(0.0)) a_covpclr = RC(0.0);
  a_covptot += a_covpclr;
  a_clc -= a_covpclr;
  if (t.newmax) { a_clc += a_covptot; a_covptot = RC(0.0); }

  // F^T (cloudsc2ad.F90:1425-1441)
  real_t a_qc = RC(0.0);
  a_qiwc += a_condi * zqtmst;  a_i -= a_condi * zqtmst;
  a_qlwc += a_condl * zqtmst;  a_l -= a_condl * zqtmst;
  a_qc += a_qiwc * (RC(1.0) - t.zfwat);
  a_fwat -= a_qiwc * t.zqc3;
  a_qc += a_qlwc * t.zfwat;
  a_fwat += a_qlwc * t.zqc3;

  // E^T: subsidence (cloudsc2ad.F90:1447-1495)
  real_t a_foeew = RC(0.0);
  {
    real_t zdqc = -a_qc, zdqsdz = RC(0.0), zrho = RC(0.0);
    if (t.llo3) {
      if (lregcl) zdqc *= RC(0.1);
      zdqsdz += zdqc * ptsphy * (x.mfu + x.mfd) * t.zfac4;
      a_mfu += zdqc * ptsphy * t.zdqsdz * t.zfac4;
      a_mfd += zdqc * ptsphy * t.zdqsdz * t.zfac4;
      zrho -= zdqc * t.zdqc * t.zfac4;
    } else {
      a_qc += zdqc;
    }
    real_t dtdzmo = zdqsdz * t.zdqsdtemp;
    a_dqsdtemp += zdqsdz * t.dtdzmo;
    real_t zrodqsdp = -zdqsdz * rg;
    real_t zldcp = -dtdzmo * (rg * t.zrodqsdp + t.dtdzmo * t.zdqsdtemp) * t.zfac3;
    zrodqsdp -= dtdzmo * rg * t.zldcp * t.zfac3;
    a_dqsdtemp -= dtdzmo * t.dtdzmo * t.zldcp * t.zfac3;
    a_fwat += zldcp * (t.zlvdcp - t.zlsdcp);
    a_lvdcp += zldcp * t.zfwat;
    a_lsdcp += zldcp * (RC(1.0) - t.zfwat);
    zrho -= zrodqsdp * x.qs * t.zfac2;
    a_qs -= zrodqsdp * t.zrho * t.zfac2;
    a_pap += zrodqsdp * t.zrho * x.qs * (t.zfac2 * t.zfac2);
    a_foeew -= zrodqsdp * t.zrho * x.qs * retv * (t.zfac2 * t.zfac2);
    a_pap += zrho * t.zfac1;
    a_tp1 -= zrho * x.pap * t.rtp2 * t.zfac1;
  }

  // D^T: convective component (cloudsc2ad.F90:1501-1526)
  {
    real_t zlude = RC(0.0);
    if (t.llo1) {
      zlude += a_qc;
      zlude += ((RC(1.0) - t.zclc) * t.rlu) * t.zexpl * a_clc;
      a_lu_k1 -= ((RC(1.0) - t.zclc) * t.zlude * (t.rlu * t.rlu)) * t.zexpl * a_clc;
      a_clc = a_clc * (RC(1.0) - (RC(1.0) - t.zexpl));
    }
    a_lude_in += ptsphy * t.zgdp * zlude;
    a_gdp += ptsphy * x.lude * zlude;
    real_t g = rg * a_gdp * (t.rdp * t.rdp);
    a_paph_k1 -= g;
    a_paph_k += g;
  }

  // C^T: cloud cover (cloudsc2ad.F90:1532-1582)
  real_t a_qsat = RC(0.0), a_qcrit = RC(0.0);
  {
    real_t zqt = RC(0.0);
    if (t.regime == 0) {
      // nothing propagates
    } else if (t.regime == 1) {
      a_qsat += (RC(1.0) - k.zscalm) * a_qc;
      a_qcrit -= (RC(1.0) - k.zscalm) * a_qc;
    } else {
      real_t zqpd = k.zscalm * a_qc * (t.zclc * t.zclc);
      real_t zqcd = (RC(1.0) - k.zscalm) * a_qc * (t.zclc * t.zclc);
      a_clc += (k.zscalm * t.zqpd + (RC(1.0) - k.zscalm) * t.zqcd) * RC(2.0) * t.zclc * a_qc;
      if (lregcl) a_clc = regcl_factor(t.zqpd, t.zqcd, k.zscalm) * a_clc;
      const real_t h = RC(0.5) * t.rzsqrt, rden2 = t.rden * t.rden;
      zqpd -= h * a_clc * t.rden;
      zqcd += h * (t.zqpd * a_clc) * rden2;
      zqt -= h * (t.zqpd * k.zscalm * a_clc) * rden2;
      a_qcrit += h * (t.zqpd * k.zscalm * a_clc) * rden2;
      a_qsat += zqcd;
      a_qcrit -= zqcd;
      a_qsat += zqpd;
      zqt -= zqpd;
    }
    a_qp1 += zqt;
    a_l += zqt;
    a_i += zqt;
  }

  // B^T and A^T (cloudsc2ad.F90:1596-1664)
  {
    a_qsat += a_qcrit * t.zcrh2;
    a_qs += a_qsat * t.zsupsat;
    real_t zsupsat = a_qsat * x.qs;
    if (t.below_rtice) a_tp1 -= zsupsat * RC(3.e-03);
    if (t.qlim_is_qs) a_qs += a_qlim; else a_qp1 += a_qlim;

    a_dqsdtemp += zcons3 * a_corqs;
    a_qs += t.zfac * t.zcor * a_dqsdtemp;
    real_t zcor = t.zfac * x.qs * a_dqsdtemp;
    real_t zfac = t.zcor * x.qs * a_dqsdtemp;
    real_t zesdp = retv * zcor * (t.zcor * t.zcor);
    real_t zfacw = t.zfwat * zfac;
    a_fwat += t.zfacw * zfac;
    real_t zfaci = (RC(1.0) - t.zfwat) * zfac;
    a_fwat -= t.zfaci * zfac;
    a_tp1 -= RC(2.0) * r5ies * zfaci * (t.ri * t.ri * t.ri);
    a_tp1 -= RC(2.0) * r5les * zfacw * (t.rl * t.rl * t.rl);
    if (t.esdp_clip) zesdp = RC(0.0);
    a_foeew += zesdp * t.zqp;
    a_pap -= zesdp * t.zfoeew * (t.zqp * t.zqp);
    real_t z3es, z4es, r4;
    if (t.cold) { z3es = r3ies; z4es = r4ies; r4 = t.ri; }
    else        { z3es = r3les; z4es = r4les; r4 = t.rl; }
    a_tp1 += z3es * (rtt - z4es) * a_foeew * t.zfoeew * (r4 * r4);
    if (t.cold) a_tp1 += RC(0.545) * RC(0.17) * a_fwat * t.zcosh2r;
  }

  // thermodynamic constants and first guess (cloudsc2ad.F90:1701-1738)
  {
    if (!rvtmp2_zero) {
      real_t zzz = c->rlvtt * a_lvdcp + c->rlstt * a_lsdcp + c->rlmlt * a_lfdcp;
      a_qp1 -= zzz * c->rcpd * c->rvtmp2 * (t.zzz * t.zzz);
    }
    a_paph_k1 += a_dp;
    a_paph_k -= a_dp;
  }

  ax.paph_k = a_paph_k;
  ax.paph_k1 = a_paph_k1;
  ax.paph_surf = a_paph_surf;
  ax.pap = a_pap;
  ax.q = a_qp1;
  ax.qs = a_qs;
  ax.t = a_tp1;
  ax.l = a_l;
  ax.i = a_i;
  ax.lude = a_lude_in;
  ax.lu_k1 = a_lu_k1;
  ax.mfu = a_mfu;
  ax.mfd = a_mfd;
  ax.gt = ptsphy * a_tp1;
  ax.gq = ptsphy * a_qp1;
  ax.gl = ptsphy * a_l;
  ax.gi = ptsphy * a_i;
  ax.supsat = ptsphy * a_qp1;  // the reference ASSIGNS PTSPHY*zqp1 (cloudsc2ad.F90:1733)

  acy.rfl = a_rfl;
  acy.sfl = a_sfl;
  acy.covptot = a_covptot;
}

}  // namespace cloudsc2
