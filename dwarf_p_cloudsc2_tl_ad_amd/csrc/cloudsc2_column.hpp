// cloudsc2_column.hpp -- what one GPU lane does for its grid column: the 137-level sweeps of SATUR+CLOUDSC2,
// CLOUDSC2TL and CLOUDSC2AD built from the per-level functions of cloudsc2_level.hpp, and the Taylor test's lambda loop as one
// sweep (taylor_column: lane = (column, lambda)).  The __global__ kernels in cloudsc2_kernels.hip are thin wrappers
// (gcol = blockIdx*blockDim + threadIdx) around these functions.
//
// Memory access: lane g reads f[jl + NPROMA*(jk + NLEVx*ibl)] with g = ibl*NPROMA + jl, i.e. consecutive lanes
// read consecutive doubles of every plane -> a wave64 fetches 512 contiguous bytes per plane and level.  Inputs of
// level JK+1 are requested before level JK is evaluated (register double buffer), which is what hides HBM latency
// at the 1-3 waves/SIMD this fp64-heavy code runs at (NL 3, TL and AD 1).
#pragma once
#include <type_traits>

#include "cloudsc2_level.hpp"

#ifndef CLOUDSC2_MAX_NLEV
#define CLOUDSC2_MAX_NLEV 200
#endif

namespace cloudsc2 {

// per-level, column-independent tables (device copy)
struct LevelTab {
  struct { real_t ceta, zscalm; } lev[CLOUDSC2_MAX_NLEV];  // interleaved: one 16-byte scalar load per level
};

struct Geom {
  int nproma, nlev, ngptot;
  long long ncols_pad;  // NBLOCKS*NPROMA
  int kb0, kb1;         // tropopause band [kb0,kb1): levels that can satisfy 0.1 < CETA < 0.4 (cloudsc2.F90:320)
  int fair;             // NL: the waves of a SIMD yield to each other by progress (progress_priority); set by the launcher
  // TL / AD: pacing of a launch of a few partial rounds of workgroups (struct Pace below); pace_recip_q16 = 0: off
  int pace_slots = 0, pace_first = 0, pace_recip_q16 = 0;
};

// Fields are grouped by the stride between NPROMA blocks.  `full` = (NPROMA,NLEV,NBLOCKS) arrays,
// `half` = (NPROMA,NLEV+1,NBLOCKS), `cml` = PGTEN* planes of B_CML, `clv` = PL/PI planes of PCLV,
// `loc` = PTEN* planes of B_LOC.  The launchers check that every field of a group has the group's stride.
struct Strides {
  long long full, half, cml, clv, loc;
};

struct InPtrs {
  const real_t *paph, *pap, *q, *qsat, *t, *l, *i, *lude, *lu, *mfu, *mfd, *gt, *gq, *gl, *gi, *supsat;
};
struct OutPtrs {
  real_t *tent, *tenq, *tenl, *teni, *clc, *fplsl, *fplsn, *fhpsl, *fhpsn, *covptot;
};
struct InPtrsRW {
  real_t *paph, *pap, *q, *qsat, *t, *l, *i, *lude, *lu, *mfu, *mfd, *gt, *gq, *gl, *gi, *supsat;
};

// Kernel argument blocks.  Each kernel takes exactly one of these by value; on the device the column functions read
// it in place from the kernel-argument segment through a constant-address-space pointer (see C2_LAUNDER).
struct NlArgs {
  Consts c; Geom g; Strides s; InPtrs in; OutPtrs out; const LevelTab* tab;
  real_t* zero_plane; long long zero_stride; real_t lam;
  real_t* ckpt;  // CKPT + EVAP kernels only: (NPROMA,NLEV,NBLOCKS) plane receiving the precipitation cover carried INTO each
                 // level.  Without the evaporation branch the carried cover feeds nothing but itself (level_forward stage G/J:
                 // covpclr is only read under llo2; level_ad: its adjoint stays zero), so it is neither stored nor re-read.
};
struct TlArgs {
  Consts c; Geom g; Strides s, sp; InPtrs in; OutPtrs out; InPtrs din; OutPtrs dout; const LevelTab* tab;
  real_t supsat_inc;  // C2F_SELFINC: the PSUPSAT increment is supsat_inc*PSUPSAT (0.01 in the Taylor test, 0 in the adjoint test)
  double* yy;         // C2F_SELFINC: NULL, or (ncols_pad) doubles receiving <y,y> of each column's TL outputs (the adjoint test's norm1)
};
// The adjoint's trajectory pass IS the NL sweep (with carry checkpoints, nl.ckpt = the scratch plane), so its
// argument block embeds the NL one.
struct AdArgs {
  NlArgs nl; Strides sa; InPtrsRW ain; OutPtrs aout;
  double* norms;  // C2F_ADNORM: (3, ncols_pad) doubles: norm1 (read), norm2 and norm3 (written)
  double* gmax;   // C2F_ADNORM: one double, raised to the largest |norm3| (the kernel's wave maxima, atomically)
};
typedef const C2_CONST_AS NlArgs* NlArgsP;
typedef const C2_CONST_AS TlArgs* TlArgsP;
typedef const C2_CONST_AS AdArgs* AdArgsP;
typedef const C2_CONST_AS InPtrs* InPtrsP;
typedef const C2_CONST_AS OutPtrs* OutPtrsP;
typedef const C2_CONST_AS InPtrsRW* InPtrsRWP;
typedef const C2_CONST_AS LevelTab* LevelTabP;
typedef const C2_CONST_AS Geom* GeomP;
typedef const C2_CONST_AS Strides* StridesP;

// Per-lane offsets of the column inside each layout group.  LaneOff: element offsets (64 bit).  LaneOff32: BYTE offsets
// in 32 bits, usable when every buffer is smaller than 4 GiB (C2F_OFF32): the accesses then take the
// `global_load v, v_off32, s[base]` form -- no 64-bit address arithmetic per access, half the offset registers.
template <class OT>
struct LaneOffT {
  OT full, half, cml, clv, loc;
};
typedef LaneOffT<long long> LaneOff;
typedef LaneOffT<unsigned> LaneOff32;
// offset of level jk / of one level row, in the units of the offset type
C2_HD long long level_off(long long, int jk, int nproma) { return (long long)jk * nproma; }
C2_HD unsigned level_off(unsigned, int jk, int nproma) { return (unsigned)jk * (unsigned)nproma * (unsigned)sizeof(real_t); }
C2_HD long long row_off(long long, int nproma) { return nproma; }
C2_HD unsigned row_off(unsigned, int nproma) { return (unsigned)nproma * (unsigned)sizeof(real_t); }

// element offsets -> the offset type of a kernel variant (bytes for LaneOff32)
template <class OT>
C2_HD LaneOffT<OT> lane_off_as(const LaneOff& o) {
  const long long m = sizeof(OT) == 4 ? (long long)sizeof(real_t) : 1;
  LaneOffT<OT> r;
  r.full = (OT)(o.full * m); r.half = (OT)(o.half * m); r.cml = (OT)(o.cml * m); r.clv = (OT)(o.clv * m); r.loc = (OT)(o.loc * m);
  return r;
}

C2_HD bool lane_setup(GeomP g, StridesP s, long long gcol, LaneOff& o, bool& active) {
  if (gcol >= g->ncols_pad) return false;
  long long ibl = gcol / g->nproma;
  long long jl = gcol - ibl * g->nproma;
  o.full = ibl * s->full + jl;
  o.half = ibl * s->half + jl;
  o.cml = ibl * s->cml + jl;
  o.clv = ibl * s->clv + jl;
  o.loc = ibl * s->loc + jl;
  active = gcol < g->ngptot;
  return true;
}

// Everything level jk needs from the input planes EXCEPT PAPHP1(JK), which is the previous level's PAPHP1(JK+1):
// the 14 full-level planes at jk plus the two look-ahead values PAPHP1(JK+1) and PLU(JK+1) (cloudsc2.F90:272,435).
// Loading the look-ahead values together with the level they belong to lets the whole set be requested one full
// level ahead of its use.
struct RawLevel {
  real_t paph_k1, pap, q, qsat, t, l, i, lude, lu_k1, mfu, mfd, gt, gq, gl, gi, supsat;
};

// Streaming accesses: every plane element is read once and written once per launch.  C2_NT_LOAD / C2_NT_STORE = 1
// mark them non-temporal (`nt`), so they do not displace the little that is re-read (tropopause band, level tables).
#ifndef C2_NT_LOAD
#define C2_NT_LOAD 1
#endif
#ifndef C2_NT_STORE
#define C2_NT_STORE 1
#endif


C2_HD real_t ldg(const real_t* p, long long i) {
#if C2_NT_LOAD && defined(__HIP_DEVICE_COMPILE__)
  return __builtin_nontemporal_load(p + i);
#else
  return p[i];
#endif
}
C2_HD void stg(real_t* p, long long i, real_t v) {
#if C2_NT_STORE && defined(__HIP_DEVICE_COMPILE__)
  __builtin_nontemporal_store(v, p + i);
#else
  p[i] = v;
#endif
}

C2_HD real_t ldg(const real_t* p, unsigned byte_off) {
  const real_t* q = (const real_t*)((const char*)p + byte_off);
#if C2_NT_LOAD && defined(__HIP_DEVICE_COMPILE__)
  return __builtin_nontemporal_load(q);
#else
  return *q;
#endif
}
C2_HD void stg(real_t* p, unsigned byte_off, real_t v) {
  real_t* q = (real_t*)((char*)p + byte_off);
#if C2_NT_STORE && defined(__HIP_DEVICE_COMPILE__)
  __builtin_nontemporal_store(v, q);
#else
  *q = v;
#endif
}

// STREAM = false: plain loads for data that other waves read too (the Taylor sweep, whose waves share cache lines)
template <bool STREAM>
C2_HD real_t ldx(const real_t* p, long long i) { return STREAM ? ldg(p, i) : p[i]; }
template <bool STREAM>
C2_HD real_t ldx(const real_t* p, unsigned byte_off) { return STREAM ? ldg(p, byte_off) : *(const real_t*)((const char*)p + byte_off); }

template <bool HAS_QSAT, class OT, bool STREAM = true>
C2_HD void load_level(InPtrsP pp, const LaneOffT<OT>& o, int nproma, int nlev, int jk, RawLevel& r) {
  const InPtrs p = *pp;
  const OT d = level_off(OT(), jk, nproma), d1 = d + row_off(OT(), nproma);
  r.paph_k1 = ldx<STREAM>(p.paph, o.half + d1);
  r.lu_k1 = (jk + 1 < nlev) ? ldx<STREAM>(p.lu, o.full + d1) : RC(0.0);
  r.pap = ldx<STREAM>(p.pap, o.full + d);
  r.q = ldx<STREAM>(p.q, o.full + d);
  r.t = ldx<STREAM>(p.t, o.full + d);
  r.l = ldx<STREAM>(p.l, o.clv + d);
  r.i = ldx<STREAM>(p.i, o.clv + d);
  r.lude = ldx<STREAM>(p.lude, o.full + d);
  r.mfu = ldx<STREAM>(p.mfu, o.full + d);
  r.mfd = ldx<STREAM>(p.mfd, o.full + d);
  r.gt = ldx<STREAM>(p.gt, o.cml + d);
  r.gq = ldx<STREAM>(p.gq, o.cml + d);
  r.gl = ldx<STREAM>(p.gl, o.cml + d);
  r.gi = ldx<STREAM>(p.gi, o.cml + d);
  r.supsat = ldx<STREAM>(p.supsat, o.full + d);
  if (HAS_QSAT) r.qsat = ldx<STREAM>(p.qsat, o.full + d);
}

// Perturbed state of the Taylor test: x5 = x + lambda*(0.01*x) (cloudsc_driver_tl_mod.F90:156-171,200-215).
C2_HD real_t pert(real_t x, real_t lam) { return x + lam * (x * RC(0.01)); }

C2_HD void perturb_raw(RawLevel& r, real_t lam) {
  r.paph_k1 = pert(r.paph_k1, lam);
  r.pap = pert(r.pap, lam); r.q = pert(r.q, lam); r.qsat = pert(r.qsat, lam); r.t = pert(r.t, lam);
  r.l = pert(r.l, lam); r.i = pert(r.i, lam); r.lude = pert(r.lude, lam); r.lu_k1 = pert(r.lu_k1, lam);
  r.mfu = pert(r.mfu, lam); r.mfd = pert(r.mfd, lam); r.gt = pert(r.gt, lam); r.gq = pert(r.gq, lam);
  r.gl = pert(r.gl, lam); r.gi = pert(r.gi, lam); r.supsat = pert(r.supsat, lam);
}

C2_HD void make_level_in(const RawLevel& cur, real_t paph_k, real_t paph_surf, LevelIn& x) {
  x.paph_k = paph_k; x.paph_k1 = cur.paph_k1;
  x.pap = cur.pap; x.q = cur.q; x.qs = cur.qsat; x.t = cur.t; x.l = cur.l; x.i = cur.i;
  x.lude = cur.lude; x.lu_k1 = cur.lu_k1; x.mfu = cur.mfu; x.mfd = cur.mfd;
  x.gt = cur.gt; x.gq = cur.gq; x.gl = cur.gl; x.gi = cur.gi; x.supsat = cur.supsat;
  x.paph_surf = paph_surf;
}

// Tropopause pre-scan (cloudsc2.F90:315-326): the last band level whose first-guess T exceeds the one below.
// The band is ~40 levels; its 2 x 40 loads are requested in batches of C2_TROP_BATCH levels -- one HBM latency per
// batch instead of one per level at the start of every wave.
#ifndef C2_TROP_BATCH
#define C2_TROP_BATCH 16
#endif
template <bool PERT>
C2_HD real_t tropopause(ConstsP c, LevelTabP tab, InPtrsP p, const LaneOff& o, GeomP g, real_t lam) {
  real_t ztrpaus = RC(0.1);
  const int kb0 = g->kb0, kb1 = g->kb1, nproma = g->nproma;
  if (kb1 > kb0) {
    const real_t ptsphy = c->ptsphy;
    const real_t* pt = p->t;
    const real_t* pg = p->gt;
    long long d = (long long)kb0 * nproma;
    real_t t0 = pt[o.full + d], g0 = pg[o.cml + d];
    if (PERT) { t0 = pert(t0, lam); g0 = pert(g0, lam); }
    real_t tup = t0 + ptsphy * g0;
    for (int jb = kb0; jb < kb1; jb += C2_TROP_BATCH) {
      real_t tb[C2_TROP_BATCH], gb[C2_TROP_BATCH];
#pragma unroll
      for (int u = 0; u < C2_TROP_BATCH; ++u) {
        const int jk1 = (jb + u + 1 < kb1) ? jb + u + 1 : kb1;  // clamped: level kb1 <= nlev-1 always exists
        const long long d1 = (long long)jk1 * nproma;
        tb[u] = pt[o.full + d1];
        gb[u] = pg[o.cml + d1];
      }
#pragma unroll
      for (int u = 0; u < C2_TROP_BATCH; ++u) {
        const int jk = jb + u;
        if (jk < kb1) {
          real_t t1 = tb[u], g1 = gb[u];
          if (PERT) { t1 = pert(t1, lam); g1 = pert(g1, lam); }
          const real_t tdn = t1 + ptsphy * g1;
          const real_t ce = tab->lev[jk].ceta;
          if (ce > RC(0.1) && ce < RC(0.4) && tup > tdn) ztrpaus = ce;
          tup = tdn;
        }
      }
    }
  }
  return ztrpaus;
}

// All ten output pointers must be valid (the launchers substitute nothing: a skipped trajectory store is a
// template flag of the TL kernel) -- no per-pointer branches, the pointer block is read with two wide scalar loads.
template <class OT>
C2_HD void store_out(OutPtrsP pp, const LaneOffT<OT>& o, int nproma, int jk, const LevelOut& v) {
  const OutPtrs p = *pp;
  const OT d = level_off(OT(), jk, nproma);
  stg(p.tent, o.loc + d, v.tent);
  stg(p.tenq, o.loc + d, v.tenq);
  stg(p.tenl, o.loc + d, v.tenl);
  stg(p.teni, o.loc + d, v.teni);
  stg(p.clc, o.full + d, v.clc);
  stg(p.covptot, o.full + d, v.covptot);
  const OT d1 = d + row_off(OT(), nproma);
  stg(p.fplsl, o.half + d1, v.fplsl);
  stg(p.fplsn, o.half + d1, v.fplsn);
  stg(p.fhpsl, o.half + d1, v.fhpsl);
  stg(p.fhpsn, o.half + d1, v.fhpsn);
}

C2_HD void store_top(OutPtrsP p, const LaneOff& o, ConstsP c) {
  // fluxes at the model top are zero (cloudsc2.F90:308-309); enthalpy fluxes -0*RLVTT (:732-733)
  const real_t z = RC(0.0);
  p->fplsl[o.half] = z;
  p->fplsn[o.half] = z;
  p->fhpsl[o.half] = -z * c->rlvtt;
  p->fhpsn[o.half] = -z * c->rlstt;
}

C2_HD void level_cst(LevelTabP tab, int jk, bool last, LevelCst& k) {
  k.ceta = tab->lev[jk].ceta;
  k.zscalm = tab->lev[jk].zscalm;
  k.last = last;
  C2_PIN2(k.ceta, k.zscalm);
}

// Pacing of the TL / AD sweeps when a launch is a few PARTIAL rounds of workgroups (round 4; profiles/EXPERIMENTS.md section 7).
// 160 000 columns are 1250 workgroups on 512 workgroup slots (256 CUs x 4 SIMDs at one wave per SIMD, two waves per workgroup): 226
// slots process three workgroups, 286 two, and the launch lasts as long as the three -- the third alone on a machine the others have
// left, at a lone wave's latency-bound pace.  While all slots are busy the memory system is saturated and every wave is slowed alike,
// but only the three-workgroup slots are on the critical path.  With in-order dispatch the workgroups that will share a slot with
// k others (instead of k - 1) are known in advance: position p = blockIdx mod slots of a round is in the FAST class if
// p < (workgroups mod slots).  The others nap at every level for 1/k of the time the level took them: their slot then needs the time
// of k + 1 unpaced workgroups for its k, they leave their share of the bandwidth to the fast class, and both classes end together
// (TL at 160 000 columns: 1.65 -> 1.57 ms, AD 3.01 -> 2.83 ms).  The nap is measured, not tabulated: it follows the clock, the
// variant and the contention by itself.
// How many waves of a launch share a SIMD (observed dispatch of two-wave workgroups on gfx950, profiles/EXPERIMENTS.md "What the
// dispatcher does", checked wave by wave with tools/wave_times.py: 2500 of 2500 in 5 of 5 launches): workgroup i runs on CU
// i mod #CUs; inside a CU the waves of its consecutive workgroups go to the SIMDs in the cyclic order 0,2 | 2,1 | 1,3 | 3,0, whatever
// SIMD the CU starts with -- so wave m of a CU (m = 2 x (i / #CUs) + its index in its workgroup) shares its SIMD with exactly the
// waves m' of that CU with the same ((m' + 1) / 2) mod 4.  `wgs_on_cu` workgroups on the CU; `mine`: waves on this wave's SIMD,
// `most`: on the CU's fullest SIMD.
C2_HD void simd_population(unsigned wgs_on_cu, unsigned j, unsigned wave_in_wg, unsigned& mine, unsigned& most) {
  unsigned pop[4] = {0u, 0u, 0u, 0u};
  for (unsigned k = 0; k < 2u * wgs_on_cu; ++k) pop[((k + 1u) >> 1) & 3u] += 1u;
  const unsigned m = 2u * j + wave_in_wg;
  mine = pop[((m + 1u) >> 1) & 3u];
  most = pop[0];
  for (int g = 1; g < 4; ++g) most = pop[g] > most ? pop[g] : most;
}

// Pace: state of one wave (wave-uniform scalars); begin() before the level loop, nap() once per level with the next loads in flight.
struct Pace {
  unsigned recip_q16 = 0;  // 65536 / k for the slow class, 0 = this workgroup does not nap
  unsigned mark = 0;       // shader clock (low 32 bits) when the previous nap ended
  C2_HD void begin(GeomP g) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned r = (unsigned)g->pace_recip_q16;
    if (r == 0) return;
    unsigned b = blockIdx.x;
    const unsigned s = (unsigned)g->pace_slots;
    while (b >= s) b -= s;  // (at most eight rounds: the launcher paces short launches only)
    if (b >= (unsigned)g->pace_first) { recip_q16 = r; mark = (unsigned)__builtin_amdgcn_s_memtime(); }
#else
    (void)g;
#endif
  }
  // One-round NL launch (g.fair & 4): inside the CUs the launch ends with -- those that carry the most workgroups -- the waves on SIMDs
  // that carry fewer waves than the CU's fullest SIMD nap a share of every level (the CU's memory pipeline is shared by its four
  // SIMDs: what the lighter ones leave, the fuller ones get).  Which SIMD a wave shares with how many others follows from blockIdx
  // alone (simd_population below); 160 000 columns: -2.1 % (profiles/r04_nl_light_ab.txt).
  C2_HD void begin_light(GeomP g) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned cus = (unsigned)g->pace_slots, q = (unsigned)g->pace_first, r = (unsigned)g->fair >> 8;
    unsigned c = blockIdx.x, j = 0;
    while (c >= cus) { c -= cus; ++j; }
    if (r != 0u && c >= r) return;  // only inside the CUs that carry the most workgroups: the launch ends with them
    unsigned mine, most;
    simd_population(q + (r != 0u ? 1u : 0u), j, (unsigned)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), mine, most);
    if (mine < most) { recip_q16 = (unsigned)g->pace_recip_q16; mark = (unsigned)__builtin_amdgcn_s_memtime(); }
#else
    (void)g;
#endif
  }
  C2_HD void nap() {
#if defined(__HIP_DEVICE_COMPILE__)
    if (recip_q16 == 0) return;
    const unsigned work = (unsigned)__builtin_amdgcn_s_memtime() - mark;             // clocks since the previous nap ended
    const unsigned naps = (unsigned)(((unsigned long long)work * recip_q16) >> 22);    // work / k, in units of 64 clocks
    for (unsigned i = 0; i < naps && i < 512u; ++i) __builtin_amdgcn_s_sleep(1);
    mark = (unsigned)__builtin_amdgcn_s_memtime();
#endif
  }
};

// Compile-time variant flags of the column sweeps (template argument F of the functions below and of the kernels)
enum : unsigned {
  C2F_QSAT = 1u,     // PQSAT is an input (otherwise SATUR is evaluated in the sweep)
  C2F_PRECISE = 2u,  // IEEE divisions / libm transcendentals in reference operation order
  C2F_EVAP = 4u,     // LEVAPLS2 .OR. LDRAIN1D
  C2F_PERT = 8u,     // NL only: inputs perturbed by lambda*0.01*x (Taylor test)
  C2F_CKPT = 16u,    // NL only: trajectory pass of the adjoint (carry checkpoints)
  C2F_TRAJ = 8u,     // TL only: trajectory outputs are stored
  C2F_SELFINC = 16u, // TL only: the increments are 0.01*x of the trajectory inputs themselves, as in both test drivers
                     // (cloudsc_driver_tl_mod.F90:156-171, cloudsc_driver_ad_mod.F90:124-139): no perturbation inputs are read
  C2F_ASSIGN = 8u,   // AD only: the input adjoints are ASSIGNED (x = A^T y) instead of accumulated (x += A^T y): their old
                     // values are neither read nor needed to be zero (the adjoint test zeroes them first, cloudsc_driver_ad_mod.F90:198-213)
  C2F_ADNORM = 16u,  // AD reverse sweep, assign form only: the adjoint test's <x, x_adj> and norm3 are formed in the sweep
                     // (cloudsc_driver_ad_mod.F90:240-264) instead of re-reading the 32 planes afterwards
  C2F_OFF32 = 32u,   // every buffer of the launch < 4 GiB: 32-bit byte offsets (LaneOff32)
  C2F_NOLIN = 64u,   // NL only: .NOT.(LPHYLIN .OR. LDRAIN1D), the FOEALFA / FOEEWM form of stage A (cloudsc2.F90:365-369)
};

// ---------------------------------------------------------------------------------------------------------
// SATUR for one column
// ---------------------------------------------------------------------------------------------------------
struct SaturArgs {
  Consts c; Geom g; Strides s; const real_t* pap; const real_t* t; real_t* qsat;
};
typedef const C2_CONST_AS SaturArgs* SaturArgsP;

template <bool P>
C2_HD void satur_column(long long gcol, SaturArgsP a) {
  LaneOff o; bool active;
  if (!lane_setup(&a->g, &a->s, gcol, o, active)) return;
  if (!active) return;
  const int nlev = a->g.nlev, nproma = a->g.nproma;
  const real_t* pap = a->pap;
  const real_t* t = a->t;
  real_t* qsat = a->qsat;
  for (int jk = 0; jk < nlev; ++jk) {
    long long d = (long long)jk * nproma;
    stg(qsat, o.full + d, satur_point<P>(C2_CONSTS(a), ldg(pap, o.full + d), ldg(t, o.full + d)));
  }
}

// Waves that share a SIMD are served oldest first: of the three NL waves on a SIMD the oldest runs as if alone and finishes a
// 160 000-column launch after 540 us, the second after 670, the youngest after 800 (tools/wave_times.py,
// profiles/r03_wave_times.txt) -- and the launch ends with its youngest waves, each alone on its SIMD and latency-bound, on a
// machine that has been half empty for 250 us.  With `fair` a wave's issue priority FALLS as it advances (s_setprio 3, 2, 1, 0,
// 3, ... per group of C2_PRIO_GROUP levels), so a wave that has got ahead of its SIMD's other waves yields to them until they
// have caught up: the waves of a SIMD stay within about one group of each other and finish together, whatever their age and
// wherever the dispatcher put them.  It pays where a launch is ONE partial round of waves, i.e. the SIMDs carry unequal numbers
// of them (160 000 columns: 2500 waves on 1024 SIMDs, 2 or 3 each: -5 %); with equal loads or several rounds the age order is
// as good or better (waves of different age are naturally staggered), so the launcher sets it by launch size (nl launch).
#ifndef C2_PRIO_GROUP
#define C2_PRIO_GROUP 8
#endif
C2_HD void progress_priority(int jk, int fair) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (fair && jk % C2_PRIO_GROUP == 0) {
    switch ((jk / C2_PRIO_GROUP) & 3) {  // (wave-uniform: scalar branches; s_setprio takes an immediate)
      case 0: __builtin_amdgcn_s_setprio(3); break;
      case 1: __builtin_amdgcn_s_setprio(2); break;
      case 2: __builtin_amdgcn_s_setprio(1); break;
      default: __builtin_amdgcn_s_setprio(0); break;
    }
  }
#else
  (void)jk; (void)fair;
#endif
}

// ---------------------------------------------------------------------------------------------------------
// NL: SATUR (optionally fused) + CLOUDSC2 for one column
// ---------------------------------------------------------------------------------------------------------
// CKPT: the sweep is the trajectory pass of the adjoint.  Rain and snow flux carries are the outputs PFPLSL5/PFPLSN5
// themselves; the one carry that is not an output (ZCOVPTOT5(JK-1)) is checkpointed -- only with the evaporation branch
// (EVAP), the one place that reads it.
template <unsigned F>
C2_HD void nl_column(long long gcol, NlArgsP a) {
  constexpr bool HAS_QSAT = (F & C2F_QSAT) != 0, PERT = (F & C2F_PERT) != 0, P = (F & C2F_PRECISE) != 0, CKPT = (F & C2F_CKPT) != 0, EVAP = (F & C2F_EVAP) != 0;
  constexpr bool OFF32 = (F & C2F_OFF32) != 0, LIN = (F & C2F_NOLIN) == 0;
  typedef typename std::conditional<OFF32, unsigned, long long>::type OT;
  static_assert(!(PERT && CKPT), "the adjoint's trajectory pass is never perturbed");
  static_assert(LIN || !CKPT, "CLOUDSC2AD's trajectory has the LPHYLIN form only");
  LaneOff o; bool active;
  if (!lane_setup(&a->g, &a->s, gcol, o, active)) return;
  const int nlev = a->g.nlev, nproma = a->g.nproma;
  const int fair = CKPT ? 0 : a->g.fair;  // (the adjoint's trajectory pass runs one wave per SIMD: compiled without the priority code)
  const real_t lam = PERT ? a->lam : RC(0.0);
  real_t* zero_plane = a->zero_plane;
  long long ozero = 0;
  if (zero_plane) {
    long long ibl = gcol / nproma;
    ozero = ibl * a->zero_stride + (gcol - ibl * nproma);
  }
  if (!active) {
    // padded tail of the last block: the driver zeroes the whole block's PCOVPTOT and CLD(:,:,NCLV)
    // (cloudsc_driver_mod.F90:87-88); nothing else is touched.
    real_t* cov = a->out.covptot;
    for (int jk = 0; jk < nlev; ++jk) {
      long long d = (long long)jk * nproma;
      cov[o.full + d] = RC(0.0);
      if (zero_plane) zero_plane[ozero + d] = RC(0.0);
    }
    return;
  }
  LevelTabP tab = (LevelTabP)a->tab;
  ConstsP c = C2_CONSTS(a);
  InPtrsP in = &a->in;
  OutPtrsP out = &a->out;
  real_t* ckpt = (CKPT && EVAP) ? a->ckpt : nullptr;
  const long long osc = (CKPT && EVAP) ? (gcol / nproma) * ((long long)nproma * nlev) + (gcol % nproma) : 0;

  // (ZTRPAUS is only read from the first band level on -- above it ZCRH2 = 1 whatever it is, cloudsc2.F90:391-399 -- so the pre-scan
  // could run when the sweep reaches the band instead of before level 1: measured equal at 160 000 columns and 1.5 % slower at 1 M,
  // like two other instruction-count candidates -- profiles/r03_b_nl_instruction_candidates_ab.txt.)
  real_t ztrpaus = tropopause<PERT>(c, tab, in, o, &a->g, lam);
  RhCrit rh;
  rhcrit_setup(ztrpaus, rh);

  real_t paph_surf = RC(0.0);
  if (EVAP) {
    paph_surf = in->paph[o.half + (long long)nlev * nproma];
    if (PERT) paph_surf = pert(paph_surf, lam);
  }

  store_top(out, o, c);

  Carry cy; cy.rfl = RC(0.0); cy.sfl = RC(0.0); cy.covptot = RC(0.0);
  real_t paph_k = in->paph[o.half];
  if (PERT) paph_k = pert(paph_k, lam);
  // offsets used inside the level loop, in the variant's offset type
  const LaneOffT<OT> ol = lane_off_as<OT>(o);
  const OT ozl = (OT)(ozero * (OFF32 ? (long long)sizeof(real_t) : 1)), oscl = (OT)(osc * (OFF32 ? (long long)sizeof(real_t) : 1));

  Pace pace;
  if (!CKPT && (fair & 4)) pace.begin_light(&a->g);

  // one level: `cur` holds the raw inputs of level jk (requested one level ago), `nxt` receives those of level jk+1
  auto step = [&](int jk, RawLevel& cur, RawLevel& nxt) {
    const bool last = (jk == nlev - 1);
    if (!CKPT) progress_priority(jk, fair & 1);
    NlArgsP ap = a;  // field pointers are re-read from the kernel-argument segment every level (transient SGPRs);
    C2_LAUNDER(ap);  // the physical constants stay resident
    // request everything level jk+1 needs now; nothing below touches `nxt` before the end of this level, so the
    // HBM latency is covered by the whole level's arithmetic
    if (!last) load_level<HAS_QSAT>(&ap->in, ol, nproma, nlev, jk + 1, nxt);
    if (!CKPT) pace.nap();

    if (!HAS_QSAT) cur.qsat = satur_point<P>(c, cur.pap, cur.t);  // SATUR on the unperturbed PAP, PT
    if (PERT) perturb_raw(cur, lam);

    LevelCst k;
    level_cst(tab, jk, last, k);
    LevelIn x;
    make_level_in(cur, paph_k, paph_surf, x);
    if (CKPT && EVAP) stg(ckpt, oscl + level_off(OT(), jk, nproma), cy.covptot);  // ZCOVPTOT5(JK-1)
    LevelTraj tr;
    LevelOut lo;
#ifdef C2_SKELETON  // diagnostic build only: the memory pattern of the sweep with the physics replaced by a few adds
    {
      const real_t s1 = x.paph_k1 + x.pap + x.q + x.qs + x.t + x.l + x.i + x.lude;
      const real_t s2 = x.lu_k1 + x.mfu + x.mfd + x.gt + x.gq + x.gl + x.gi + x.supsat + cy.rfl;
      cy.rfl = s1 * RC(1e-9) + s2 * RC(1e-9);
      lo.tent = s1; lo.tenq = s2; lo.tenl = s1 + s2; lo.teni = s1 - s2; lo.clc = s1 * RC(0.5); lo.covptot = s2 * RC(0.5);
      lo.fplsl = cy.rfl; lo.fplsn = s1 * RC(0.25); lo.fhpsl = s2 * RC(0.25); lo.fhpsn = s1 * RC(0.125);
      (void)tr; (void)k; (void)rh;
    }
#else
    level_forward<P, EVAP, LIN>(c, k, rh, x, cy, tr, lo);
#endif
    C2_LAUNDER(ap);
    store_out(&ap->out, ol, nproma, jk, lo);
    if (zero_plane) stg(zero_plane, ozl + level_off(OT(), jk, nproma), RC(0.0));
    paph_k = cur.paph_k1;
  };

  // (round 3, with the waves of a SIMD kept abreast by progress_priority: a third register set for two levels of look-ahead -- 165
  // VGPRs, still three waves per SIMD -- 0.790 against 0.784 ms at 160 000 columns and 4.98 against 4.91 at 1 M: the latency a
  // deeper look-ahead would hide is not what is left.  Earlier:
  // two levels of look-ahead with three register sets in rotation need 208 VGPRs, i.e. two waves per SIMD: 0.93 instead of 0.81 ms
  // at 160 000 columns, equal at 1 M -- profiles/r02_ab_experiments.txt; with three waves it spills 148 bytes per lane.  TOUCHING the
  // rows of level jk+2 instead -- one plain 32-bit load per lane and plane, 16 dwords held for a level, 167 VGPRs, no spill -- so that
  // the lines are in L2 / the Infinity Cache when the real load comes: 0.92 instead of 0.82 ms, 5.94 instead of 4.90 ms at 1 M; the
  // sweep is limited by the requests it makes, not by their latency.  What the physics costs on top of its own memory pattern, with
  // the state placed alike: -DC2_SKELETON 0.769 ms against 0.818 at 160 000 columns (6 %), 4.93 against 4.98 at 1 M (1 %).)
  RawLevel ra, rb;
  load_level<HAS_QSAT>(in, ol, nproma, nlev, 0, ra);
  rb = ra;
#pragma clang loop unroll(disable)
  for (int jk = 0; jk < nlev; ++jk) {
    step(jk, ra, rb);
    ra = rb;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Taylor test: the ten perturbed NL runs of CLOUDSC_DRIVER_TL and the level sums of its ERROR_NORM calls
// (cloudsc_driver_tl_mod.F90:21-31,197-244) in ONE sweep -- the lambdas lie on the lanes of the wave
// ---------------------------------------------------------------------------------------------------------
// A wave holds kTaylorCols columns x kTaylorLambdas lambdas: lane (c, k) runs the NL sweep of column c on the state perturbed by
// lambda_k = 10^-(k+1), exactly what nl_column<F | C2F_PERT> does, but stores nothing: it reads the BASE run's outputs of the
// level instead and keeps sum_levels(F - F5) of the ten compared fields in registers; the lane also sums TL output field k of its
// column (the denominators).  The ten lanes of a column request the same addresses (one 8-byte word per plane and level), so the
// state is read from HBM once for all ten runs instead of once per run, no perturbed outputs are written and re-read, and the
// code is the NL sweep's (one level_forward per lane and level; 60 of 64 lanes work).
//   colsum[(k*10 + f)*ncols_pad + g] = sum_levels(F_f - F5_f(lambda_k)) of column g     (k, f = 0..9; order of ERROR_NORM calls)
//   colsum[(100 + f)*ncols_pad + g]  = sum_levels(TL_f) of column g
constexpr int kTaylorLambdas = 10, kTaylorCols = 64 / kTaylorLambdas;
struct TenPtrs { const real_t* p[10]; long long stride[10]; int nlevx[10]; };  // the ten compared fields, cloudsc_driver_tl_mod.F90:233-242
struct TaylorArgs {
  NlArgs nl;      // nl.in: the unperturbed state; nl.out: the outputs of the BASE run (read here); nl.lam, zero_plane, ckpt unused
  TenPtrs tl;     // the TL outputs
  real_t lam[kTaylorLambdas];
  double* colsum;
};
typedef const C2_CONST_AS TaylorArgs* TaylorArgsP;

template <class OT>
C2_HD void load_out(OutPtrsP pp, const LaneOffT<OT>& o, int nproma, int jk, LevelOut& v) {
  const OutPtrs p = *pp;
  const OT d = level_off(OT(), jk, nproma);
  v.tent = ldx<false>(p.tent, o.loc + d);
  v.tenq = ldx<false>(p.tenq, o.loc + d);
  v.tenl = ldx<false>(p.tenl, o.loc + d);
  v.teni = ldx<false>(p.teni, o.loc + d);
  v.clc = ldx<false>(p.clc, o.full + d);
  v.covptot = ldx<false>(p.covptot, o.full + d);
  const OT d1 = d + row_off(OT(), nproma);
  v.fplsl = ldx<false>(p.fplsl, o.half + d1);
  v.fplsn = ldx<false>(p.fplsn, o.half + d1);
  v.fhpsl = ldx<false>(p.fhpsl, o.half + d1);
  v.fhpsn = ldx<false>(p.fhpsn, o.half + d1);
}

// column and lambda index of a thread: wave w holds columns [w*kTaylorCols, (w+1)*kTaylorCols), lane = k*kTaylorCols + c
C2_HD bool taylor_lane(long long gthread, long long& gcol, int& k) {
  const int lane = (int)(gthread & 63);
  k = lane / kTaylorCols;
  gcol = (gthread >> 6) * kTaylorCols + (lane - k * kTaylorCols);
  return k < kTaylorLambdas;
}

template <unsigned F>
C2_HD void taylor_column(long long gthread, TaylorArgsP ta) {
  constexpr bool HAS_QSAT = (F & C2F_QSAT) != 0, P = (F & C2F_PRECISE) != 0, EVAP = (F & C2F_EVAP) != 0, OFF32 = (F & C2F_OFF32) != 0;
  static_assert(!(F & (C2F_PERT | C2F_CKPT | C2F_NOLIN)), "flags of the Taylor sweep: QSAT, PRECISE, EVAP, OFF32");
  typedef typename std::conditional<OFF32, unsigned, long long>::type OT;
  long long gcol; int k;
  if (!taylor_lane(gthread, gcol, k)) return;
  NlArgsP a = &ta->nl;
  LaneOff o; bool active;
  if (!lane_setup(&a->g, &a->s, gcol, o, active) || !active) return;
  const int nlev = a->g.nlev, nproma = a->g.nproma;
  const real_t lam = ta->lam[k];
  LevelTabP tab = (LevelTabP)a->tab;
  ConstsP c = C2_CONSTS(a);
  InPtrsP in = &a->in;

  real_t ztrpaus = tropopause<true>(c, tab, in, o, &a->g, lam);
  RhCrit rh;
  rhcrit_setup(ztrpaus, rh);
  real_t paph_surf = RC(0.0);
  if (EVAP) paph_surf = pert(in->paph[o.half + (long long)nlev * nproma], lam);
  Carry cy; cy.rfl = RC(0.0); cy.sfl = RC(0.0); cy.covptot = RC(0.0);
  real_t paph_k = pert(in->paph[o.half], lam);
  const LaneOffT<OT> ol = lane_off_as<OT>(o);

  // TL output field k of this column (the fluxes live on half levels: level jk's flux is the one at jk+1, like LevelOut's;
  // their top value is zero in every run)
  const long long ibl = gcol / nproma;
  const real_t* tlk = ta->tl.p[k] + ibl * ta->tl.stride[k] + (gcol - ibl * nproma) + (ta->tl.nlevx[k] > nlev ? nproma : 0);
  double s[10] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0}, stl = 0.0;

  auto step = [&](int jk, RawLevel& cur, RawLevel& nxt) {
    const bool last = (jk == nlev - 1);
    NlArgsP ap = a;
    C2_LAUNDER(ap);
    if (!last) load_level<HAS_QSAT, OT, false>(&ap->in, ol, nproma, nlev, jk + 1, nxt);
    LevelOut base;
    load_out(&ap->out, ol, nproma, jk, base);
    const real_t tlv = tlk[(long long)jk * nproma];

    if (!HAS_QSAT) cur.qsat = satur_point<P>(c, cur.pap, cur.t);  // SATUR on the unperturbed PAP, PT
    perturb_raw(cur, lam);
    LevelCst kc;
    level_cst(tab, jk, last, kc);
    LevelIn x;
    make_level_in(cur, paph_k, paph_surf, x);
    LevelTraj tr;
    LevelOut lo;
    level_forward<P, EVAP, true>(c, kc, rh, x, cy, tr, lo);
    s[0] += (double)(base.tent - lo.tent); s[1] += (double)(base.tenq - lo.tenq);
    s[2] += (double)(base.tenl - lo.tenl); s[3] += (double)(base.teni - lo.teni);
    s[4] += (double)(base.clc - lo.clc);
    s[5] += (double)(base.fplsl - lo.fplsl); s[6] += (double)(base.fplsn - lo.fplsn);
    s[7] += (double)(base.fhpsl - lo.fhpsl); s[8] += (double)(base.fhpsn - lo.fhpsn);
    s[9] += (double)(base.covptot - lo.covptot);
    stl += (double)tlv;
    paph_k = cur.paph_k1;
  };

  RawLevel ra, rb;
  load_level<HAS_QSAT, OT, false>(in, ol, nproma, nlev, 0, ra);
  rb = ra;
#pragma clang loop unroll(disable)
  for (int jk = 0; jk < nlev; ++jk) {
    step(jk, ra, rb);
    ra = rb;
  }
  double* cs = ta->colsum;
  const long long np = a->g.ncols_pad;
#pragma unroll
  for (int f = 0; f < 10; ++f) cs[(long long)(k * 10 + f) * np + gcol] = s[f];
  cs[(long long)(100 + k) * np + gcol] = stl;
}

// The increments of the reference's test drivers: dx = 0.01*x for every input (ZSUPSAT: 0.01*PSUPSAT in the Taylor test, 0 in the
// adjoint test), taken from the trajectory inputs the sweep holds anyway.
C2_HD void self_increment(const RawLevel& r, real_t supsat_inc, RawLevel& d) {
  const real_t e = RC(0.01);
  d.paph_k1 = r.paph_k1 * e; d.pap = r.pap * e; d.q = r.q * e; d.qsat = r.qsat * e; d.t = r.t * e; d.l = r.l * e; d.i = r.i * e;
  d.lude = r.lude * e; d.lu_k1 = r.lu_k1 * e; d.mfu = r.mfu * e; d.mfd = r.mfd * e; d.gt = r.gt * e; d.gq = r.gq * e;
  d.gl = r.gl * e; d.gi = r.gi * e; d.supsat = r.supsat * supsat_inc;
}

// ---------------------------------------------------------------------------------------------------------
// TL: SATUR (optionally fused) + CLOUDSC2TL for one column
// ---------------------------------------------------------------------------------------------------------
template <unsigned F>
C2_HD void tl_column(long long gcol, TlArgsP a) {
  constexpr bool HAS_QSAT = (F & C2F_QSAT) != 0, P = (F & C2F_PRECISE) != 0, STORE_TRAJ = (F & C2F_TRAJ) != 0, EVAP = (F & C2F_EVAP) != 0;
  constexpr bool SELFINC = (F & C2F_SELFINC) != 0;
  typedef typename std::conditional<(F & C2F_OFF32) != 0, unsigned, long long>::type OT;
  LaneOff o, op; bool active;
  if (!lane_setup(&a->g, &a->s, gcol, o, active)) return;
  lane_setup(&a->g, &a->sp, gcol, op, active);
  if (!active) return;
  const int nlev = a->g.nlev, nproma = a->g.nproma;
  // only the fp32 variants that run three waves per SIMD (tl_kernel's launch bounds) share their SIMDs; the others -- all fp64
  // ones -- are compiled without the priority code (one more live SGPR costs the register-tight fp64 kernel 4 %)
  constexpr bool FAIRV = sizeof(real_t) == 4 && (F & C2F_OFF32) != 0 && !EVAP;
  const int fair = FAIRV ? a->g.fair : 0;
  LevelTabP tab = (LevelTabP)a->tab;
  ConstsP c = C2_CONSTS(a);
  InPtrsP in = &a->in, din = &a->din;
  OutPtrsP out = &a->out, dout = &a->dout;

  real_t ztrpaus = tropopause<false>(c, tab, in, o, &a->g, RC(0.0));
  RhCrit rh;
  rhcrit_setup(ztrpaus, rh);

  real_t paph_surf = RC(0.0), dpaph_surf = RC(0.0);
  if (EVAP) {
    paph_surf = in->paph[o.half + (long long)nlev * nproma];
    dpaph_surf = SELFINC ? paph_surf * RC(0.01) : din->paph[op.half + (long long)nlev * nproma];
  }

  if (STORE_TRAJ) store_top(out, o, c);
  store_top(dout, op, c);

  Carry cy; cy.rfl = RC(0.0); cy.sfl = RC(0.0); cy.covptot = RC(0.0);
  Carry dcy; dcy.rfl = RC(0.0); dcy.sfl = RC(0.0); dcy.covptot = RC(0.0);
  RawLevel cur, nxt, dcur, dnxt;
  double yy = 0.0;  // SELFINC: <y,y> of the column (cloudsc_driver_ad_mod.F90:184-195; the fluxes' top values are zero)
  real_t paph_k = in->paph[o.half], dpaph_k = SELFINC ? paph_k * RC(0.01) : din->paph[op.half];
  const LaneOffT<OT> ol = lane_off_as<OT>(o), opl = lane_off_as<OT>(op);  // offsets used inside the level loop
  load_level<HAS_QSAT>(in, ol, nproma, nlev, 0, cur);
  if (!SELFINC) load_level<true>(din, opl, nproma, nlev, 0, dcur);
  Pace pace;
  pace.begin(&a->g);

  // Both input sets of level jk+1 are requested at the top of level jk.  Measured alternatives (profiles/r02_ab_experiments.txt):
  // requesting the perturbation inputs at the top of their own level (no second register set for them: 280 instead of 311
  // registers) is 2 % slower, requesting the trajectory inputs between level_forward and level_tl 12 % slower, two levels of
  // look-ahead 3 % slower at 160 000 columns and equal at 1 M: the full level of distance is what hides HBM latency here.
  for (int jk = 0; jk < nlev; ++jk) {
    const bool last = (jk == nlev - 1);
    if (FAIRV) progress_priority(jk, fair);
    TlArgsP ap = a;
    C2_LAUNDER(ap);
    in = &ap->in; din = &ap->din;
    nxt = cur;
    if (!SELFINC) dnxt = dcur;
    if (!last) {
      load_level<HAS_QSAT>(in, ol, nproma, nlev, jk + 1, nxt);
      if (!SELFINC) load_level<true>(din, opl, nproma, nlev, jk + 1, dnxt);
    }
    pace.nap();  // (with the next level's loads in flight)
    if (!HAS_QSAT) cur.qsat = satur_point<P>(c, cur.pap, cur.t);
    if (SELFINC) self_increment(cur, ap->supsat_inc, dcur);

    LevelCst k;
    level_cst(tab, jk, last, k);
    LevelIn x, dx;
    make_level_in(cur, paph_k, paph_surf, x);
    make_level_in(dcur, dpaph_k, dpaph_surf, dx);
    LevelTraj tr;
    LevelOut lo, dlo;
    level_forward<P, EVAP>(c, k, rh, x, cy, tr, lo);
    level_tl(c, k, x, tr, dx, dcy, dlo);
    C2_LAUNDER(ap);
    out = &ap->out; dout = &ap->dout;
    if (STORE_TRAJ) store_out(out, ol, nproma, jk, lo);
    store_out(dout, opl, nproma, jk, dlo);
    if (SELFINC) {
      yy += (double)dlo.tent * dlo.tent + (double)dlo.tenq * dlo.tenq + (double)dlo.tenl * dlo.tenl + (double)dlo.teni * dlo.teni +
            (double)dlo.clc * dlo.clc + (double)dlo.covptot * dlo.covptot + (double)dlo.fplsl * dlo.fplsl + (double)dlo.fplsn * dlo.fplsn +
            (double)dlo.fhpsl * dlo.fhpsl + (double)dlo.fhpsn * dlo.fhpsn;
    }
    paph_k = cur.paph_k1; dpaph_k = dcur.paph_k1;
    cur = nxt;
    if (!SELFINC) dcur = dnxt;
  }
  if (SELFINC) {
    double* p = a->yy;
    if (p) p[gcol] = yy;
  }
}

// -DC2_TL_DMA=1 (experiment, profiles/EXPERIMENTS.md section 6; default off; NPROMA 128 only): the TL sweep with its look-ahead in
// LDS instead of registers.  A 128-thread workgroup is one NPROMA block row, so a level's row of any plane is 1 KiB contiguous: ONE
// 16-byte-per-lane LDS-DMA load (global_load_lds_dwordx4, no VGPR destination) by one wave fetches it for both waves.  Wave 0 requests
// the 16 trajectory rows of level jk+1, wave 1 the 16 perturbation rows, at the top of level jk; after the level's stores both waves
// meet, read their columns' 32 values (ds_read_b64) and meet again before the buffer is overwritten.  32 KiB per workgroup (four
// workgroups = 2 waves per SIMD fit a CU's 160 KiB), no second register set: the kernel is meant to be built with -DC2_TL_WAVES=2.
#ifndef C2_TL_DMA
#define C2_TL_DMA 0
#endif
#if C2_TL_DMA && defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void global_cvoid_t;
// one 1 KiB row: lane l of the calling wave moves 16 bytes from src (+ 16 l, already in `src`) to row[16 l]
__device__ __forceinline__ void dma_row(const real_t* plane, long long elem_off, real_t* row) {
  __builtin_amdgcn_global_load_lds((global_cvoid_t*)(plane + elem_off), (lds_void_t*)row, 16, 0, 2 /* nt */);
}
// the 16 rows of one input set for level jk (RawLevel order), `o`: offsets of column (block start + 2 * lane) in each layout group
template <bool HAS_QSAT>
__device__ __forceinline__ void dma_level(InPtrsP pp, const LaneOff& o, int nproma, int nlev, int jk, real_t (*rows)[128], int first, int count) {
  const InPtrs p = *pp;
  const long long d = (long long)jk * nproma, d1 = d + nproma;
  const real_t* src[16] = {p.paph, p.pap, p.q, p.qsat, p.t, p.l, p.i, p.lude, p.lu, p.mfu, p.mfd, p.gt, p.gq, p.gl, p.gi, p.supsat};
  const long long off[16] = {o.half + d1, o.full + d, o.full + d, o.full + d, o.full + d, o.clv + d, o.clv + d, o.full + d, o.full + d1,
                             o.full + d, o.full + d, o.cml + d, o.cml + d, o.cml + d, o.cml + d, o.full + d};
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    if (k < first || k >= first + count) continue;
    if (k == 3 && !HAS_QSAT) continue;
    if (k == 8 && jk + 1 >= nlev) continue;  // PLU(JK+1) below the last level: 0 (load_level)
    dma_row(src[k], off[k], rows[k]);
  }
}
template <bool HAS_QSAT>
__device__ __forceinline__ void rows_get(real_t (*rows)[128], int nlev, int jk, RawLevel& r) {
  const int t = threadIdx.x;
  r.paph_k1 = rows[0][t]; r.pap = rows[1][t]; r.q = rows[2][t]; if (HAS_QSAT) r.qsat = rows[3][t]; r.t = rows[4][t]; r.l = rows[5][t];
  r.i = rows[6][t]; r.lude = rows[7][t]; r.lu_k1 = (jk + 1 < nlev) ? rows[8][t] : RC(0.0); r.mfu = rows[9][t]; r.mfd = rows[10][t];
  r.gt = rows[11][t]; r.gq = rows[12][t]; r.gl = rows[13][t]; r.gi = rows[14][t]; r.supsat = rows[15][t];
}

template <unsigned F>
__device__ __forceinline__ void tl_column_dma(TlArgsP a) {
  constexpr bool HAS_QSAT = (F & C2F_QSAT) != 0, P = (F & C2F_PRECISE) != 0, STORE_TRAJ = (F & C2F_TRAJ) != 0, EVAP = (F & C2F_EVAP) != 0;
  constexpr bool SELFINC = (F & C2F_SELFINC) != 0;
  typedef typename std::conditional<(F & C2F_OFF32) != 0, unsigned, long long>::type OT;
  __shared__ real_t rows_t[16][128], rows_p[16][128];
  const long long gcol = (long long)blockIdx.x * 128 + threadIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  LaneOff o, op, orow, oprow; bool active, dummy;
  if (!lane_setup(&a->g, &a->s, gcol, o, active)) return;  // (whole workgroups only: ncols_pad is a multiple of NPROMA = 128)
  lane_setup(&a->g, &a->sp, gcol, op, active);
  lane_setup(&a->g, &a->s, (long long)blockIdx.x * 128 + 2 * lane, orow, dummy);   // the 16 bytes this lane moves of every row
  lane_setup(&a->g, &a->sp, (long long)blockIdx.x * 128 + 2 * lane, oprow, dummy);
  const int nlev = a->g.nlev, nproma = a->g.nproma;
  LevelTabP tab = (LevelTabP)a->tab;
  ConstsP c = C2_CONSTS(a);
  InPtrsP in = &a->in, din = &a->din;
  OutPtrsP out = &a->out, dout = &a->dout;

  real_t ztrpaus = tropopause<false>(c, tab, in, o, &a->g, RC(0.0));
  RhCrit rh;
  rhcrit_setup(ztrpaus, rh);
  real_t paph_surf = RC(0.0), dpaph_surf = RC(0.0);
  if (EVAP) {
    paph_surf = in->paph[o.half + (long long)nlev * nproma];
    dpaph_surf = SELFINC ? paph_surf * RC(0.01) : din->paph[op.half + (long long)nlev * nproma];
  }
  if (active) {
    if (STORE_TRAJ) store_top(out, o, c);
    store_top(dout, op, c);
  }
  Carry cy; cy.rfl = RC(0.0); cy.sfl = RC(0.0); cy.covptot = RC(0.0);
  Carry dcy; dcy.rfl = RC(0.0); dcy.sfl = RC(0.0); dcy.covptot = RC(0.0);
  RawLevel cur, dcur;
  double yy = 0.0;
  real_t paph_k = in->paph[o.half], dpaph_k = SELFINC ? paph_k * RC(0.01) : din->paph[op.half];
  const LaneOffT<OT> ol = lane_off_as<OT>(o), opl = lane_off_as<OT>(op);

  // who requests what: with a perturbation set wave 0 takes the trajectory rows and wave 1 the perturbation rows; without
  // (SELFINC: the increments are formed from the trajectory) the two waves take eight trajectory rows each
  auto request = [&](int jk) {
    if (SELFINC) dma_level<HAS_QSAT>(in, orow, nproma, nlev, jk, rows_t, wave * 8, 8);
    else if (wave == 0) dma_level<HAS_QSAT>(in, orow, nproma, nlev, jk, rows_t, 0, 16);
    else dma_level<true>(din, oprow, nproma, nlev, jk, rows_p, 0, 16);
  };
  request(0);
  __syncthreads();
  rows_get<HAS_QSAT>(rows_t, nlev, 0, cur);
  if (!SELFINC) rows_get<true>(rows_p, nlev, 0, dcur);
  __syncthreads();

  for (int jk = 0; jk < nlev; ++jk) {
    const bool last = (jk == nlev - 1);
    TlArgsP ap = a;
    C2_LAUNDER(ap);
    in = &ap->in; din = &ap->din;
    if (!last) request(jk + 1);
    if (!HAS_QSAT) cur.qsat = satur_point<P>(c, cur.pap, cur.t);
    if (SELFINC) self_increment(cur, ap->supsat_inc, dcur);

    LevelCst k;
    level_cst(tab, jk, last, k);
    LevelIn x, dx;
    make_level_in(cur, paph_k, paph_surf, x);
    make_level_in(dcur, dpaph_k, dpaph_surf, dx);
    LevelTraj tr;
    LevelOut lo, dlo;
    level_forward<P, EVAP>(c, k, rh, x, cy, tr, lo);
    level_tl(c, k, x, tr, dx, dcy, dlo);
    C2_LAUNDER(ap);
    out = &ap->out; dout = &ap->dout;
    if (active) {
      if (STORE_TRAJ) store_out(out, ol, nproma, jk, lo);
      store_out(dout, opl, nproma, jk, dlo);
    }
    if (SELFINC) {
      yy += (double)dlo.tent * dlo.tent + (double)dlo.tenq * dlo.tenq + (double)dlo.tenl * dlo.tenl + (double)dlo.teni * dlo.teni +
            (double)dlo.clc * dlo.clc + (double)dlo.covptot * dlo.covptot + (double)dlo.fplsl * dlo.fplsl + (double)dlo.fplsn * dlo.fplsn +
            (double)dlo.fhpsl * dlo.fhpsl + (double)dlo.fhpsn * dlo.fhpsn;
    }
    paph_k = cur.paph_k1; dpaph_k = dcur.paph_k1;
    if (!last) {
      __syncthreads();  // (waits for this wave's own requests, then for the other wave's)
      rows_get<HAS_QSAT>(rows_t, nlev, jk + 1, cur);
      if (!SELFINC) rows_get<true>(rows_p, nlev, jk + 1, dcur);
      __syncthreads();  // everybody has read: the rows may be overwritten
    }
  }
  if (SELFINC && active) {
    double* p = a->yy;
    if (p) p[gcol] = yy;
  }
}
#endif

// ---------------------------------------------------------------------------------------------------------
// AD for one column.  Two passes: the trajectory pass is nl_column<.., CKPT=true> -- it writes the trajectory
// outputs and checkpoints the carries (rain and snow flux live in the outputs PFPLSL5/PFPLSN5 that have to be
// written anyway; the precipitation cover goes to the scratch plane when the evaporation branch, its only reader, is
// compiled in).  The reverse pass below re-evaluates each level's trajectory from its checkpoint and applies the
// transposed level.  It needs nothing from the trajectory pass but PFPLSL5/PFPLSN5 (and the cover plane with EVAP), so a
// caller that already has them -- any earlier NL or TL sweep over the same state -- can run it alone
// (cloudsc2_ad_launch_reverse).
// ---------------------------------------------------------------------------------------------------------

// Everything the reverse pass reads for level jk: trajectory inputs, the three checkpointed carries, the output
// adjoints, and the OLD values of the input adjoints that are accumulated into.  (Written as `a[i] += x` after the
// compute, each read-modify-write would wait for its own HBM round trip: the compiler cannot move a load above a
// store that might alias.)
struct AdLevelLoads {
  RawLevel cur;   // trajectory inputs; paph_k1 is NOT loaded (it is the paph_k of the level below, already in a register)
  real_t paph_k;
  Carry cy;
  LevelOut ya;
  RawLevel xo;    // old input adjoints (PSUPSAT is assigned, not accumulated: not read)
};

template <bool HAS_QSAT, bool ASSIGN, bool EVAP, class OT>
C2_HD void ad_load_level(AdArgsP ap, const LaneOffT<OT>& o, const LaneOffT<OT>& oa, OT osc, int nproma, int nlev, int jk,
                         AdLevelLoads& L) {
  const bool last = (jk == nlev - 1);
  const OT d = level_off(OT(), jk, nproma);
  const OT d1 = d + row_off(OT(), nproma);
  {
    const InPtrs p = ap->nl.in;
    L.paph_k = ldg(p.paph, o.half + d);
    L.cur.lu_k1 = last ? RC(0.0) : ldg(p.lu, o.full + d1);
    L.cur.pap = ldg(p.pap, o.full + d);
    L.cur.q = ldg(p.q, o.full + d);
    L.cur.t = ldg(p.t, o.full + d);
    L.cur.l = ldg(p.l, o.clv + d);
    L.cur.i = ldg(p.i, o.clv + d);
    L.cur.lude = ldg(p.lude, o.full + d);
    L.cur.mfu = ldg(p.mfu, o.full + d);
    L.cur.mfd = ldg(p.mfd, o.full + d);
    L.cur.gt = ldg(p.gt, o.cml + d);
    L.cur.gq = ldg(p.gq, o.cml + d);
    L.cur.gl = ldg(p.gl, o.cml + d);
    L.cur.gi = ldg(p.gi, o.cml + d);
    L.cur.supsat = ldg(p.supsat, o.full + d);
    if (HAS_QSAT) L.cur.qsat = ldg(p.qsat, o.full + d);
    const OutPtrs po = ap->nl.out;
    L.cy.rfl = ldg(po.fplsl, o.half + d);  // ZRFL5(JK) = PFPLSL5(JK)
    L.cy.sfl = ldg(po.fplsn, o.half + d);
    // the carried cover matters to the evaporation branch alone; without it any value gives the same results (0 here)
    L.cy.covptot = EVAP ? ldg(ap->nl.ckpt, osc + d) : RC(0.0);
  }
  const OutPtrs pa = ap->aout;
  L.ya.tent = ldg(pa.tent, oa.loc + d);
  L.ya.tenq = ldg(pa.tenq, oa.loc + d);
  L.ya.tenl = ldg(pa.tenl, oa.loc + d);
  L.ya.teni = ldg(pa.teni, oa.loc + d);
  L.ya.clc = ldg(pa.clc, oa.full + d);
  L.ya.covptot = ldg(pa.covptot, oa.full + d);
  L.ya.fplsn = ldg(pa.fplsn, oa.half + d1);
  L.ya.fplsl = ldg(pa.fplsl, oa.half + d1);
  L.ya.fhpsn = ldg(pa.fhpsn, oa.half + d1);
  L.ya.fhpsl = ldg(pa.fhpsl, oa.half + d1);
  if (ASSIGN) {
    // assign form: the old input adjoints are not read (16 planes of traffic less per level)
    L.xo.pap = L.xo.q = L.xo.qsat = L.xo.t = L.xo.l = L.xo.i = L.xo.lude = L.xo.mfu = L.xo.mfd = RC(0.0);
    L.xo.gt = L.xo.gq = L.xo.gl = L.xo.gi = L.xo.lu_k1 = L.xo.paph_k1 = RC(0.0);
    return;
  }
  const InPtrsRW px = ap->ain;
  L.xo.pap = ldg(px.pap, oa.full + d);
  L.xo.q = ldg(px.q, oa.full + d);
  L.xo.qsat = ldg(px.qsat, oa.full + d);
  L.xo.t = ldg(px.t, oa.full + d);
  L.xo.l = ldg(px.l, oa.clv + d);
  L.xo.i = ldg(px.i, oa.clv + d);
  L.xo.lude = ldg(px.lude, oa.full + d);
  L.xo.mfu = ldg(px.mfu, oa.full + d);
  L.xo.mfd = ldg(px.mfd, oa.full + d);
  L.xo.gt = ldg(px.gt, oa.cml + d);
  L.xo.gq = ldg(px.gq, oa.cml + d);
  L.xo.gl = ldg(px.gl, oa.cml + d);
  L.xo.gi = ldg(px.gi, oa.cml + d);
  L.xo.lu_k1 = last ? RC(0.0) : ldg(px.lu, oa.full + d1);
  L.xo.paph_k1 = last ? RC(0.0) : ldg(px.paph, oa.half + d1);
}

// reverse sweep (cloudsc2ad.F90:877-1740); the trajectory pass has run before
C2_HD double adjoint_norm3(double n1, double n2);

template <unsigned F>
C2_HD double ad_reverse_column(long long gcol, AdArgsP a) {  // returns |norm3| of the column with C2F_ADNORM (+inf for NaN), else 0
  constexpr bool HAS_QSAT = (F & C2F_QSAT) != 0, P = (F & C2F_PRECISE) != 0, EVAP = (F & C2F_EVAP) != 0;
  constexpr bool OFF32 = (F & C2F_OFF32) != 0, ASSIGN = (F & C2F_ASSIGN) != 0, ADNORM = (F & C2F_ADNORM) != 0;
  static_assert(!ADNORM || ASSIGN, "the fused norms are those of the adjoint test: assign form");
  typedef typename std::conditional<OFF32, unsigned, long long>::type OT;
  LaneOff o, oa64; bool active;
  if (!lane_setup(&a->nl.g, &a->nl.s, gcol, o, active)) return 0.0;
  lane_setup(&a->nl.g, &a->sa, gcol, oa64, active);
  if (!active) return 0.0;
  const int nlev = a->nl.g.nlev, nproma = a->nl.g.nproma;
  LevelTabP tab = (LevelTabP)a->nl.tab;
  ConstsP c = C2_CONSTS(&a->nl);
  InPtrsP in = &a->nl.in;

  // scratch: (NPROMA, NLEV, NBLOCKS) contiguous
  const long long osc64 = (gcol / nproma) * ((long long)nproma * nlev) + (gcol % nproma);
  // offsets used inside the level loop, in the variant's offset type
  const LaneOffT<OT> ol = lane_off_as<OT>(o), oa = lane_off_as<OT>(oa64);
  const OT osc = (OT)(osc64 * (OFF32 ? (long long)sizeof(real_t) : 1));

  real_t ztrpaus = tropopause<false>(c, tab, in, o, &a->nl.g, RC(0.0));
  RhCrit rh;
  rhcrit_setup(ztrpaus, rh);
  const real_t paph_bottom = in->paph[o.half + (long long)nlev * nproma];
  const real_t paph_surf = EVAP ? paph_bottom : RC(0.0);

  Carry acy; acy.rfl = RC(0.0); acy.sfl = RC(0.0); acy.covptot = RC(0.0);
  real_t paph_pending = RC(0.0);  // contribution of level jk+1 to the PAPHP1 adjoint at half level jk+1
  real_t surf_acc = RC(0.0);      // PAPHP1(KLEV+1) adjoint, written once at the end
  real_t paph_k1 = paph_bottom;
  double n2 = 0.0;  // ADNORM: <x0, x_adj>, x0 = 0.01 * trajectory inputs, ZSUPSAT0 = 0 (cloudsc_driver_ad_mod.F90:139,240-256)
  AdLevelLoads L;
  Pace pace;
  pace.begin(&a->nl.g);
  for (int jk = nlev - 1; jk >= 0; --jk) {
    const bool last = (jk == nlev - 1);
    pace.nap();
    const OT d = level_off(OT(), jk, nproma);
    const OT d1 = d + row_off(OT(), nproma);
    AdArgsP ap = a;
    C2_LAUNDER(ap);
    // all 44 loads of the level at its top; requesting the trajectory part (19 values) one level ahead was measured: +1.6 % time
    // at 160 000 columns, -1 % at 1 M (profiles/r02_ab_experiments.txt)
    ad_load_level<HAS_QSAT, ASSIGN, EVAP>(ap, ol, oa, osc, nproma, nlev, jk, L);
    RawLevel& cur = L.cur;
    cur.paph_k1 = paph_k1;
    const RawLevel& xo = L.xo;
    LevelOut ya = L.ya;  // enthalpy-flux adjoints folded in below (cloudsc2ad.F90:914-921)

    if (!HAS_QSAT) cur.qsat = satur_point<P>(c, cur.pap, cur.t);
    LevelCst k;
    level_cst(tab, jk, last, k);
    LevelIn x;
    make_level_in(cur, L.paph_k, paph_surf, x);
    LevelTraj tr;
    LevelOut lo;
    level_forward<P, EVAP>(c, k, rh, x, L.cy, tr, lo);

    ya.fplsn = ya.fplsn - ya.fhpsn * c->rlstt;
    ya.fplsl = ya.fplsl - ya.fhpsl * c->rlvtt;
    LevelIn ax;
    level_ad(c, k, x, tr, ya, acy, ax);

    C2_LAUNDER(ap);
    const InPtrsRW px = ap->ain;
    const OutPtrs pa = ap->aout;
    // accumulate input adjoints (cloudsc2ad.F90:1723-1738; PSUPSAT assigned, :1733)
    stg(px.pap, oa.full + d, xo.pap + ax.pap);
    stg(px.q, oa.full + d, xo.q + ax.q);
    stg(px.qsat, oa.full + d, xo.qsat + ax.qs);
    stg(px.t, oa.full + d, xo.t + ax.t);
    stg(px.l, oa.clv + d, xo.l + ax.l);
    stg(px.i, oa.clv + d, xo.i + ax.i);
    stg(px.lude, oa.full + d, xo.lude + ax.lude);
    stg(px.mfu, oa.full + d, xo.mfu + ax.mfu);
    stg(px.mfd, oa.full + d, xo.mfd + ax.mfd);
    stg(px.gt, oa.cml + d, xo.gt + ax.gt);
    stg(px.gq, oa.cml + d, xo.gq + ax.gq);
    stg(px.gl, oa.cml + d, xo.gl + ax.gl);
    stg(px.gi, oa.cml + d, xo.gi + ax.gi);
    stg(px.supsat, oa.full + d, ax.supsat);
    if (!last) stg(px.lu, oa.full + d1, xo.lu_k1 + ax.lu_k1);
    surf_acc += ax.paph_surf;
    if (last) {
      surf_acc += ax.paph_k1;
    } else {
      stg(px.paph, oa.half + d1, xo.paph_k1 + (ax.paph_k1 + paph_pending));
    }
    if (ADNORM) {  // (assign form: what has just been stored is ax itself)
      const double e = 0.01;
      n2 += ((double)cur.pap * e) * ax.pap + ((double)cur.q * e) * ax.q + ((double)cur.qsat * e) * ax.qs + ((double)cur.t * e) * ax.t +
            ((double)cur.l * e) * ax.l + ((double)cur.i * e) * ax.i + ((double)cur.lude * e) * ax.lude + ((double)cur.mfu * e) * ax.mfu +
            ((double)cur.mfd * e) * ax.mfd + ((double)cur.gt * e) * ax.gt + ((double)cur.gq * e) * ax.gq + ((double)cur.gl * e) * ax.gl +
            ((double)cur.gi * e) * ax.gi;
      if (!last) n2 += ((double)cur.lu_k1 * e) * ax.lu_k1 + ((double)cur.paph_k1 * e) * (ax.paph_k1 + paph_pending);
    }
    paph_pending = ax.paph_k;

    // output adjoints are consumed (cloudsc2ad.F90:917-919,955-966,1173,1572)
    stg(pa.tent, oa.loc + d, RC(0.0));
    stg(pa.tenq, oa.loc + d, RC(0.0));
    stg(pa.tenl, oa.loc + d, RC(0.0));
    stg(pa.teni, oa.loc + d, RC(0.0));
    stg(pa.clc, oa.full + d, RC(0.0));
    stg(pa.covptot, oa.full + d, RC(0.0));
    stg(pa.fplsl, oa.half + d1, RC(0.0));
    stg(pa.fplsn, oa.half + d1, RC(0.0));
    stg(pa.fhpsl, oa.half + d1, RC(0.0));
    stg(pa.fhpsn, oa.half + d1, RC(0.0));

    paph_k1 = L.paph_k;
  }
  InPtrsRWP ain = &a->ain;
  OutPtrsP aout = &a->aout;
  if (ASSIGN) {
    ain->paph[oa64.half] = paph_pending;
    ain->paph[oa64.half + (long long)nlev * nproma] = surf_acc;
    ain->lu[oa64.full] = RC(0.0);  // PLU(1) has no adjoint contribution (only PLU(JK+1) is read, cloudsc2.F90:435)
  } else {
    ain->paph[oa64.half] += paph_pending;
    ain->paph[oa64.half + (long long)nlev * nproma] += surf_acc;
  }
  // the adjoint of the (constant zero) top fluxes is discarded (cloudsc2ad.F90:1678-1679,917-919)
  aout->fplsl[oa64.half] = RC(0.0);
  aout->fplsn[oa64.half] = RC(0.0);
  aout->fhpsl[oa64.half] = RC(0.0);
  aout->fhpsn[oa64.half] = RC(0.0);
  if (ADNORM) {
    // the two half levels the loop does not store: PAPHP1(1) (paph_k1 now holds the trajectory's value there) and PAPHP1(KLEV+1)
    n2 += ((double)paph_k1 * 0.01) * paph_pending + ((double)paph_bottom * 0.01) * surf_acc;
    double* norms = a->norms;
    const long long np = a->nl.g.ncols_pad;
    double n3 = adjoint_norm3(norms[gcol], n2);
    norms[np + gcol] = n2;
    norms[2 * np + gcol] = n3;
    n3 = fabs(n3);
    if (!(n3 == n3)) n3 = (double)INFINITY;  // NaN counts as failure
    return n3;
  }
  return 0.0;
}

// ---------------------------------------------------------------------------------------------------------
// Adjoint-test norms per column (cloudsc_driver_ad_mod.F90:184-195,240-264)
// ---------------------------------------------------------------------------------------------------------
// norm1 = <y,y>, field by field like the reference's SUMs (:185-194)
C2_HD double adjoint_norm1_column(int nlev, int nproma, const LaneOff& oa, const OutPtrs& y) {
  double st = 0, sq = 0, sl = 0, si = 0, sc = 0, sfl = 0, sfn = 0, shl = 0, shn = 0, scv = 0;
  for (int jk = 0; jk < nlev; ++jk) {
    long long d = (long long)jk * nproma;
    double v;
    v = y.tent[oa.loc + d]; st += v * v;
    v = y.tenq[oa.loc + d]; sq += v * v;
    v = y.tenl[oa.loc + d]; sl += v * v;
    v = y.teni[oa.loc + d]; si += v * v;
    v = y.clc[oa.full + d]; sc += v * v;
    v = y.covptot[oa.full + d]; scv += v * v;
  }
  for (int jk = 0; jk <= nlev; ++jk) {
    long long d = (long long)jk * nproma;
    double v;
    v = y.fplsl[oa.half + d]; sfl += v * v;
    v = y.fplsn[oa.half + d]; sfn += v * v;
    v = y.fhpsl[oa.half + d]; shl += v * v;
    v = y.fhpsn[oa.half + d]; shn += v * v;
  }
  return st + sq + sl + si + sc + sfl + sfn + shl + shn + scv;
}

// norm2 = <x0, x_adj> with x0 = 0.01 * trajectory inputs; ZSUPSAT0 = 0 (:139,157) so that term vanishes
C2_HD double adjoint_norm2_column(int nlev, int nproma, const LaneOff& o, const LaneOff& oa, long long oq, const InPtrs& in,
                                  const real_t* qsat, const InPtrs& xa) {
  double s_aph = 0, s_ap = 0, s_q = 0, s_qs = 0, s_t = 0, s_l = 0, s_i = 0, s_lude = 0, s_lu = 0, s_mfu = 0, s_mfd = 0, s_gt = 0,
         s_gq = 0, s_gl = 0, s_gi = 0;
  for (int jk = 0; jk <= nlev; ++jk) {
    long long d = (long long)jk * nproma;
    s_aph += (in.paph[o.half + d] * 0.01) * xa.paph[oa.half + d];
  }
  for (int jk = 0; jk < nlev; ++jk) {
    long long d = (long long)jk * nproma;
    s_ap += (in.pap[o.full + d] * 0.01) * xa.pap[oa.full + d];
    s_q += (in.q[o.full + d] * 0.01) * xa.q[oa.full + d];
    s_qs += (qsat[oq + d] * 0.01) * xa.qsat[oa.full + d];
    s_t += (in.t[o.full + d] * 0.01) * xa.t[oa.full + d];
    s_l += (in.l[o.clv + d] * 0.01) * xa.l[oa.clv + d];
    s_i += (in.i[o.clv + d] * 0.01) * xa.i[oa.clv + d];
    s_lude += (in.lude[o.full + d] * 0.01) * xa.lude[oa.full + d];
    s_lu += (in.lu[o.full + d] * 0.01) * xa.lu[oa.full + d];
    s_mfu += (in.mfu[o.full + d] * 0.01) * xa.mfu[oa.full + d];
    s_mfd += (in.mfd[o.full + d] * 0.01) * xa.mfd[oa.full + d];
    s_gt += (in.gt[o.cml + d] * 0.01) * xa.gt[oa.cml + d];
    s_gq += (in.gq[o.cml + d] * 0.01) * xa.gq[oa.cml + d];
    s_gl += (in.gl[o.cml + d] * 0.01) * xa.gl[oa.cml + d];
    s_gi += (in.gi[o.cml + d] * 0.01) * xa.gi[oa.cml + d];
  }
  return s_aph + s_ap + s_q + s_qs + s_t + s_l + s_i + s_lude + s_lu + s_mfu + s_mfd + s_gt + s_gq + s_gl + s_gi + 0.0;
}

// "machine precision is defined here as strictly 64bits" (cloudsc_driver_ad_mod.F90:258-264): EPSILON(1._8) whatever
// JPRB is, so the fp32 build reports its error in the same unit as the reference's -DSINGLE binary does.
C2_HD double adjoint_norm3(double n1, double n2) {
  const double eps = 2.220446049250313e-16;  // EPSILON(1._8)
  if (n2 == 0.0) return fabs(n1 - n2) / eps;
  return fabs(n1 - n2) / eps / n2;
}

}  // namespace cloudsc2
