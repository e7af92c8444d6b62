// TEST-ONLY: compiles the per-column device functions of csrc/cloudsc2_column.hpp for the HOST (hipcc
// --cuda-host-only) so that their logic can be unit-tested against the oracle in a container without a GPU.
// This library is never loaded by the package, the bench or the C ABI; the product has no CPU path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "../../include/cloudsc2_hip.h"
#include "../../dwarf_p_cloudsc2_tl_ad_amd/csrc/cloudsc2_column.hpp"

using namespace cloudsc2;

static Consts hc_consts(const cloudsc2_params& p, double ptsphy) { return make_consts(p, ptsphy); }

// run-time flag word -> compile-time variant
template <template <unsigned> class Fn, unsigned N, unsigned F = 0, class A>
static void hc_dispatch(unsigned f, long long gc, const A* a) {
  if constexpr (F < N) {
    if (f == F) Fn<F>::run(gc, a);
    else hc_dispatch<Fn, N, F + 1>(f, gc, a);
  }
}
template <unsigned F> struct HcNl {
  static void run(long long gc, const NlArgs* a) {
    if constexpr (!(F & C2F_CKPT)) nl_column<F>(gc, a);
  }
};
// only the flag words the launchers can produce are instantiated (TL: QSAT|PRECISE|EVAP|TRAJ|OFF32, AD: no TRAJ)
template <unsigned F> struct HcTl {
  static void run(long long gc, const TlArgs* a) {
    if constexpr ((F & ~(C2F_QSAT | C2F_PRECISE | C2F_EVAP | C2F_TRAJ | C2F_OFF32 | C2F_SELFINC)) == 0) tl_column<F>(gc, a);
  }
};
static int g_hc_ad_sweep = 0;  // 0: both sweeps of the adjoint, 1: forward sweep only, 2: reverse sweep only
static double g_hc_norm3_max = 0.0;  // largest |norm3| the C2F_ADNORM sweep returned (what the kernel's wave maxima feed)
template <unsigned F> struct HcAd {
  static void run(long long gc, const AdArgs* a) {
    if constexpr ((F & ~(C2F_QSAT | C2F_PRECISE | C2F_EVAP | C2F_OFF32 | C2F_ASSIGN)) == 0) {
      if (g_hc_ad_sweep != 2) nl_column<(F & ~C2F_ASSIGN) | C2F_CKPT>(gc, &a->nl);
      if (g_hc_ad_sweep != 1) ad_reverse_column<F>(gc, a);
    } else if constexpr ((F & C2F_ADNORM) && (F & C2F_ASSIGN) && !(F & C2F_EVAP)) {  // cloudsc2_ad_launch_reverse_norms
      g_hc_norm3_max = fmax(g_hc_norm3_max, ad_reverse_column<F>(gc, a));
    }
  }
};

template <unsigned F> struct HcTaylor {
  static void run(long long gthread, const TaylorArgs* a) {
    if constexpr ((F & ~(C2F_QSAT | C2F_PRECISE | C2F_EVAP | C2F_OFF32)) == 0) taylor_column<F>(gthread, a);
  }
};

// same derivation as get_tables() in cloudsc2_kernels.hip

static void hc_tables(const cloudsc2_params& p, LevelTab& tab, Geom& g) {
  memset(&tab, 0, sizeof(tab));
  g.kb0 = p.nlev; g.kb1 = 0;
  for (int jk = 0; jk < p.nlev; ++jk) {
    tab.lev[jk].ceta = p.ceta[jk];
    tab.lev[jk].zscalm = 0.9 * pow(fmax(p.ceta[jk] - 0.2, 1.e-12), 0.2);
    if (jk < p.nlev - 1 && p.ceta[jk] > 0.1 && p.ceta[jk] < 0.4) {
      if (jk < g.kb0) g.kb0 = jk;
      if (jk + 1 > g.kb1) g.kb1 = jk + 1;
    }
  }
  if (g.kb1 <= g.kb0) { g.kb0 = 0; g.kb1 = 0; }
}

static void hc_in(const cloudsc2_inputs& in, Strides& s, InPtrs& p) {
  s.full = in.pap.block_stride; s.half = in.paph.block_stride; s.cml = in.gtent.block_stride; s.clv = in.l.block_stride;
  p.paph = in.paph.ptr; p.pap = in.pap.ptr; p.q = in.q.ptr; p.qsat = in.qsat.ptr; p.t = in.t.ptr; p.l = in.l.ptr;
  p.i = in.i.ptr; p.lude = in.lude.ptr; p.lu = in.lu.ptr; p.mfu = in.mfu.ptr; p.mfd = in.mfd.ptr;
  p.gt = in.gtent.ptr; p.gq = in.gtenq.ptr; p.gl = in.gtenl.ptr; p.gi = in.gteni.ptr; p.supsat = in.supsat.ptr;
}
static void hc_out(const cloudsc2_outputs& out, Strides& s, OutPtrs& p) {
  s.loc = out.tent.block_stride;
  p.tent = out.tent.ptr; p.tenq = out.tenq.ptr; p.tenl = out.tenl.ptr; p.teni = out.teni.ptr; p.clc = out.clc.ptr;
  p.fplsl = out.fplsl.ptr; p.fplsn = out.fplsn.ptr; p.fhpsl = out.fhpsl.ptr; p.fhpsn = out.fhpsn.ptr;
  p.covptot = out.covptot.ptr;
}

static Geom hc_geom(int nproma, int nlev, int ngptot) {
  Geom g;
  long long nb = ((long long)ngptot + nproma - 1) / nproma;
  g.nproma = nproma; g.nlev = nlev; g.ngptot = ngptot; g.ncols_pad = nb * nproma; g.kb0 = g.kb1 = 0; g.fair = 0;
  return g;
}

static int g_hc_precise = 0;
static int g_hc_off32 = 0;
static int g_hc_assign = 0;
static double* g_hc_yy = nullptr;      // TL self-increment form: receives <y,y> per column
static double* g_hc_ad_norms = nullptr;  // AD: (3, ncols_pad), row 0 = norm1; reverse sweep alone in the assign form with the norms formed in it
static double g_hc_supsat_inc = -1.0;  // >= 0: hostcheck_tl runs the self-increment form with this PSUPSAT factor

extern "C" {

void hostcheck_set_precise(int p) { g_hc_precise = p; }
void hostcheck_set_assign(int v) { g_hc_assign = v; }  // AD: assign the input adjoints instead of accumulating (C2F_ASSIGN)
void hostcheck_set_ad_sweep(int v) { g_hc_ad_sweep = v; }  // what cloudsc2_ad_launch_forward / _reverse run
void hostcheck_set_self_increment(double v) { g_hc_supsat_inc = v; }  // TL: cloudsc2_tl_launch_self (negative: increments from din)
void hostcheck_set_yy(double* p) { g_hc_yy = p; }
void hostcheck_set_ad_norms(double* p) { g_hc_ad_norms = p; g_hc_norm3_max = 0.0; }
double hostcheck_get_norm3_max(void) { return g_hc_norm3_max; }
void hostcheck_set_off32(int v) { g_hc_off32 = v; }  // 32-bit byte offsets (C2F_OFF32) in all three sweeps

int hostcheck_satur(const cloudsc2_params* prm, int nproma, int nlev, int ngptot, cloudsc2_field pap, cloudsc2_field t,
                    cloudsc2_field qsat) {
  SaturArgs a;
  a.g = hc_geom(nproma, nlev, ngptot);
  a.c = hc_consts(*prm, 1.0);
  a.s = Strides{pap.block_stride, 0, 0, 0, 0};
  a.pap = pap.ptr; a.t = t.ptr; a.qsat = qsat.ptr;
  for (long long gc = 0; gc < a.g.ncols_pad; ++gc) { if (g_hc_precise) satur_column<true>(gc, &a); else satur_column<false>(gc, &a); }
  return 0;
}

int hostcheck_nl(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot, const cloudsc2_inputs* in,
                 const cloudsc2_outputs* out, cloudsc2_field zero_plane, double lam) {
  NlArgs a;
  a.g = hc_geom(nproma, nlev, ngptot);
  a.c = hc_consts(*prm, ptsphy);
  LevelTab tab; hc_tables(*prm, tab, a.g);
  a.tab = &tab;
  a.s = Strides{0, 0, 0, 0, 0};
  hc_in(*in, a.s, a.in); hc_out(*out, a.s, a.out);
  a.zero_plane = zero_plane.ptr; a.zero_stride = zero_plane.block_stride; a.lam = lam; a.ckpt = nullptr;
  unsigned f = (in->qsat.ptr ? C2F_QSAT : 0u) | (lam != 0.0 ? C2F_PERT : 0u) | (g_hc_precise ? C2F_PRECISE : 0u) |
               (a.c.evap ? C2F_EVAP : 0u) | (g_hc_off32 ? C2F_OFF32 : 0u) | ((!prm->lphylin && !prm->ldrain1d) ? C2F_NOLIN : 0u);
  for (long long gc = 0; gc < a.g.ncols_pad; ++gc) hc_dispatch<HcNl, 128>(f, gc, &a);
  return 0;
}

int hostcheck_tl(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot, const cloudsc2_inputs* in,
                 const cloudsc2_outputs* out, const cloudsc2_inputs* din, const cloudsc2_outputs* dout) {
  TlArgs a;
  a.g = hc_geom(nproma, nlev, ngptot);
  a.c = hc_consts(*prm, ptsphy);
  LevelTab tab; hc_tables(*prm, tab, a.g);
  a.tab = &tab;
  a.s = Strides{0, 0, 0, 0, 0}; a.sp = Strides{0, 0, 0, 0, 0};
  hc_in(*in, a.s, a.in); hc_out(*out, a.s, a.out);
  if (g_hc_supsat_inc >= 0.0) {  // cloudsc2_tl_launch_self: the increments are 0.01*x, formed in the sweep; din is not read
    memset(&a.din, 0, sizeof(a.din));
    a.sp.full = dout->clc.block_stride; a.sp.half = dout->fplsl.block_stride;
  } else {
    hc_in(*din, a.sp, a.din);
  }
  hc_out(*dout, a.sp, a.dout);
  a.supsat_inc = (real_t)(g_hc_supsat_inc >= 0.0 ? g_hc_supsat_inc : 0.0);
  a.yy = g_hc_yy;
  unsigned f = (in->qsat.ptr ? C2F_QSAT : 0u) | C2F_TRAJ | (g_hc_precise ? C2F_PRECISE : 0u) | (a.c.evap ? C2F_EVAP : 0u) |
               (g_hc_off32 ? C2F_OFF32 : 0u) | (g_hc_supsat_inc >= 0.0 ? C2F_SELFINC : 0u);
  for (long long gc = 0; gc < a.g.ncols_pad; ++gc) hc_dispatch<HcTl, 64>(f, gc, &a);
  return 0;
}

int hostcheck_ad(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot, const cloudsc2_inputs* in,
                 const cloudsc2_outputs* out, const cloudsc2_inputs* ain, const cloudsc2_outputs* aout, cloudsc2_real* scratch) {
  AdArgs a;
  a.nl.g = hc_geom(nproma, nlev, ngptot);
  a.nl.c = hc_consts(*prm, ptsphy);
  LevelTab tab; hc_tables(*prm, tab, a.nl.g);
  a.nl.tab = &tab;
  a.nl.s = Strides{0, 0, 0, 0, 0}; a.sa = Strides{0, 0, 0, 0, 0};
  InPtrs aip_c;
  hc_in(*in, a.nl.s, a.nl.in); hc_out(*out, a.nl.s, a.nl.out); hc_in(*ain, a.sa, aip_c); hc_out(*aout, a.sa, a.aout);
  a.ain.paph = ain->paph.ptr; a.ain.pap = ain->pap.ptr; a.ain.q = ain->q.ptr; a.ain.qsat = ain->qsat.ptr; a.ain.t = ain->t.ptr;
  a.ain.l = ain->l.ptr; a.ain.i = ain->i.ptr; a.ain.lude = ain->lude.ptr; a.ain.lu = ain->lu.ptr; a.ain.mfu = ain->mfu.ptr;
  a.ain.mfd = ain->mfd.ptr; a.ain.gt = ain->gtent.ptr; a.ain.gq = ain->gtenq.ptr; a.ain.gl = ain->gtenl.ptr;
  a.ain.gi = ain->gteni.ptr; a.ain.supsat = ain->supsat.ptr;
  a.nl.zero_plane = nullptr; a.nl.zero_stride = 0; a.nl.lam = 0.0; a.nl.ckpt = scratch;
  unsigned f = (in->qsat.ptr ? C2F_QSAT : 0u) | (g_hc_precise ? C2F_PRECISE : 0u) | (a.nl.c.evap ? C2F_EVAP : 0u) |
               (g_hc_off32 ? C2F_OFF32 : 0u) | (g_hc_assign ? C2F_ASSIGN : 0u);
  a.norms = g_hc_ad_norms; a.gmax = nullptr;
  if (g_hc_ad_norms) {
    if (g_hc_ad_sweep != 2 || !g_hc_assign || a.nl.c.evap) return -1;
    f |= C2F_ADNORM;
  }
  for (long long gc = 0; gc < a.nl.g.ncols_pad; ++gc) hc_dispatch<HcAd, 64>(f, gc, &a);
  return 0;
}

// the lambda sweep of the Taylor test (cloudsc2_taylor_sweep_launch's first kernel): colsum[(10*10 + 10) * ncols_pad]
int hostcheck_taylor_sweep(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot, const cloudsc2_inputs* in,
                           const cloudsc2_outputs* out, const cloudsc2_outputs* tl, double* colsum) {
  TaylorArgs a;
  a.nl.g = hc_geom(nproma, nlev, ngptot);
  a.nl.c = hc_consts(*prm, ptsphy);
  LevelTab tab; hc_tables(*prm, tab, a.nl.g);
  a.nl.tab = &tab;
  a.nl.s = Strides{0, 0, 0, 0, 0};
  hc_in(*in, a.nl.s, a.nl.in); hc_out(*out, a.nl.s, a.nl.out);
  a.nl.zero_plane = nullptr; a.nl.zero_stride = 0; a.nl.lam = 0; a.nl.ckpt = nullptr;
  const cloudsc2_field* f[10] = {&tl->tent, &tl->tenq, &tl->tenl, &tl->teni, &tl->clc, &tl->fplsl, &tl->fplsn, &tl->fhpsl, &tl->fhpsn, &tl->covptot};
  const int half[10] = {0, 0, 0, 0, 0, 1, 1, 1, 1, 0};
  for (int i = 0; i < 10; ++i) { a.tl.p[i] = f[i]->ptr; a.tl.stride[i] = f[i]->block_stride; a.tl.nlevx[i] = nlev + half[i]; }
  for (int il = 0; il < kTaylorLambdas; ++il) a.lam[il] = (real_t)pow(10.0, -(double)(il + 1));
  a.colsum = colsum;
  unsigned fl = (in->qsat.ptr ? C2F_QSAT : 0u) | (g_hc_precise ? C2F_PRECISE : 0u) | (a.nl.c.evap ? C2F_EVAP : 0u) | (g_hc_off32 ? C2F_OFF32 : 0u);
  const long long nthreads = ((a.nl.g.ncols_pad + kTaylorCols - 1) / kTaylorCols) * 64;
  for (long long t = 0; t < nthreads; ++t) hc_dispatch<HcTaylor, 64>(fl, t, &a);
  return 0;
}

}  // extern "C"
