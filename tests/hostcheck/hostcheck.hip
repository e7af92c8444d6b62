// TEST-ONLY: compiles the per-column device functions of csrc/cloudsc2_column.hpp for the HOST (hipcc
// --cuda-host-only) so that their logic can be unit-tested against the oracle in a container without a GPU.
// This library is never loaded by the package, the bench or the C ABI; the product has no CPU path.
#include <hip/hip_runtime.h>
#include <math.h>
#include <string.h>

#include "../../include/cloudsc2_hip.h"
#include "../../dwarf_p_cloudsc2_tl_ad_amd/csrc/cloudsc2_column.hpp"

using namespace cloudsc2;

// same derivation as make_consts()/get_tables() in cloudsc2_kernels.hip
static Consts hc_consts(const cloudsc2_params& p, double ptsphy) {
  Consts c;
  c.rg = p.rg; c.rd = p.rd; c.rcpd = p.rcpd; c.retv = p.retv; c.rlvtt = p.rlvtt; c.rlstt = p.rlstt;
  c.rlmlt = p.rlmlt; c.rtt = p.rtt;
  c.r2es = p.r2es; c.r3les = p.r3les; c.r3ies = p.r3ies; c.r4les = p.r4les; c.r4ies = p.r4ies;
  c.r5les = p.r5les; c.r5ies = p.r5ies; c.r5alvcp = p.r5alvcp; c.r5alscp = p.r5alscp;
  c.ralvdcp = p.ralvdcp; c.ralsdcp = p.ralsdcp;
  c.rtwat = p.rtwat; c.rtice = p.rtice; c.rtwat_rtice_r = p.rtwat_rtice_r; c.rvtmp2 = p.rvtmp2;
  c.rlmin = p.rlmin; c.rpecons = p.rpecons; c.rlptrc = p.rlptrc;
  c.ptsphy = ptsphy;
  c.zckcodtl = 2.0 * p.rkconv * ptsphy;
  c.zckcodti = 5.0 * p.rkconv * ptsphy;
  c.zckcodtla = c.zckcodtl / 100.0;
  c.zckcodtia = c.zckcodti / 100.0;
  c.zcons2 = 1.0 / (ptsphy * p.rg);
  c.zcons3 = p.rlvtt / p.rcpd;
  c.zmeltp2 = p.rtt + 2.0;
  c.zqtmst = 1.0 / ptsphy;
  c.evap = (p.levapls2 || p.ldrain1d) ? 1 : 0;
  c.zlcrit_l = c.evap ? 1.9 * p.rclcrit : p.rclcrit * 2.0;
  c.zlcrit_i = c.evap ? 1.e-04 : p.rclcrit * 2.0;
  c.rcpd_r = 1.0 / p.rcpd;
  c.zlcrit_l_r = 1.0 / c.zlcrit_l;
  c.zlcrit_i_r = 1.0 / c.zlcrit_i;
  c.zcons2_r = ptsphy * p.rg;
  c.rvtmp2_zero = (p.rvtmp2 == 0.0) ? 1 : 0;
  c.zzz0 = 1.0 / (p.rcpd + p.rcpd * p.rvtmp2 * 0.0);
  c.zlfdcp0_r = 1.0 / (p.rlmlt * c.zzz0);
  c.lregcl = p.lregcl ? 1 : 0;
  c.nlev = p.nlev;
  fill_stage_blocks(c);
  return c;
}

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
  g.nproma = nproma; g.nlev = nlev; g.ngptot = ngptot; g.ncols_pad = nb * nproma; g.kb0 = g.kb1 = 0;
  return g;
}

static int g_hc_precise = 0;

extern "C" {

void hostcheck_set_precise(int p) { g_hc_precise = p; }

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
  a.zero_plane = zero_plane.ptr; a.zero_stride = zero_plane.block_stride; a.lam = lam;
  const bool hq = in->qsat.ptr != nullptr, pt = lam != 0.0;
  for (long long gc = 0; gc < a.g.ncols_pad; ++gc) {
    if (g_hc_precise) {
      if (hq && pt) nl_column<true, true, true>(gc, &a);
      else if (hq) nl_column<true, false, true>(gc, &a);
      else if (pt) nl_column<false, true, true>(gc, &a);
      else nl_column<false, false, true>(gc, &a);
    } else {
      if (hq && pt) nl_column<true, true, false>(gc, &a);
      else if (hq) nl_column<true, false, false>(gc, &a);
      else if (pt) nl_column<false, true, false>(gc, &a);
      else nl_column<false, false, false>(gc, &a);
    }
  }
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
  hc_in(*in, a.s, a.in); hc_out(*out, a.s, a.out); hc_in(*din, a.sp, a.din); hc_out(*dout, a.sp, a.dout);
  for (long long gc = 0; gc < a.g.ncols_pad; ++gc) {
    if (g_hc_precise) { if (in->qsat.ptr) tl_column<true, true, true>(gc, &a); else tl_column<false, true, true>(gc, &a); }
    else { if (in->qsat.ptr) tl_column<true, false, true>(gc, &a); else tl_column<false, false, true>(gc, &a); }
  }
  return 0;
}

int hostcheck_ad(const cloudsc2_params* prm, double ptsphy, int nproma, int nlev, int ngptot, const cloudsc2_inputs* in,
                 const cloudsc2_outputs* out, const cloudsc2_inputs* ain, const cloudsc2_outputs* aout, double* scratch) {
  AdArgs a;
  a.g = hc_geom(nproma, nlev, ngptot);
  a.c = hc_consts(*prm, ptsphy);
  LevelTab tab; hc_tables(*prm, tab, a.g);
  a.tab = &tab;
  a.s = Strides{0, 0, 0, 0, 0}; a.sa = Strides{0, 0, 0, 0, 0};
  InPtrs aip_c;
  hc_in(*in, a.s, a.in); hc_out(*out, a.s, a.out); hc_in(*ain, a.sa, aip_c); hc_out(*aout, a.sa, a.aout);
  a.ain.paph = ain->paph.ptr; a.ain.pap = ain->pap.ptr; a.ain.q = ain->q.ptr; a.ain.qsat = ain->qsat.ptr; a.ain.t = ain->t.ptr;
  a.ain.l = ain->l.ptr; a.ain.i = ain->i.ptr; a.ain.lude = ain->lude.ptr; a.ain.lu = ain->lu.ptr; a.ain.mfu = ain->mfu.ptr;
  a.ain.mfd = ain->mfd.ptr; a.ain.gt = ain->gtent.ptr; a.ain.gq = ain->gtenq.ptr; a.ain.gl = ain->gtenl.ptr;
  a.ain.gi = ain->gteni.ptr; a.ain.supsat = ain->supsat.ptr;
  a.scratch = scratch;
  for (long long gc = 0; gc < a.g.ncols_pad; ++gc) {
    if (g_hc_precise) { if (in->qsat.ptr) ad_column<true, true>(gc, &a); else ad_column<false, true>(gc, &a); }
    else { if (in->qsat.ptr) ad_column<true, false>(gc, &a); else ad_column<false, false>(gc, &a); }
  }
  return 0;
}

}  // extern "C"
