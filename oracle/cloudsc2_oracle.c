/*
 * cloudsc2_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the reference's CLOUDSC2 hot path (SATUR, CLOUDSC2, CLOUDSC2TL + CUADJTQSTL,
 * CLOUDSC2AD + CUADJTQS + CUADJTQSAD, KCALL=0 only) used as the parity checker of the HIP kernels.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path never does.
 *
 * PARITY PINNED: this file is checked against the unmodified reference Fortran compiled by oracle/Makefile
 * (oracle/_ref, tests/test_oracle_vs_reference.py) and against golden vectors generated from it
 * (tests/golden/, tests/golden/make_golden.py).  config-files/reference.h5 cannot pin it directly because the
 * matching config-files/input.h5 is not distributed (.MISSING_LARGE_BLOBS).
 *
 * Structure follows the reference, not the GPU design: one column at a time, per-level work arrays, the AD stores
 * its whole trajectory in a forward sweep and unwinds it (cloudsc2ad.F90:366-866 / :877-1740).  Arrays are the
 * reference's (KLON,KLEV) explicit-shape arrays, i.e. a[jk*klon + jl] with 0-based jk, jl.  Line numbers cite
 * /root/reference/src/... as in SURVEY.md.  Compile with -ffp-contract=off: the flang build of the reference uses
 * no FMA either.
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXLEV 200

/* module constants (yomcst.F90, yoethf.F90, yoecldp.F90, yoephli.F90, yoecld.F90, yophnc.F90, yomncl.F90) */
static double RG, RD, RCPD, RETV, RLVTT, RLSTT, RLMLT, RTT;
static double R2ES, R3LES, R3IES, R4LES, R4IES, R5LES, R5IES, R5ALVCP, R5ALSCP, RALVDCP, RALSDCP, RTWAT, RTICE,
    RTWAT_RTICE_R, RVTMP2;
static double RCLCRIT, RKCONV, RLMIN, RPECONS, RLPTRC;
static int LPHYLIN = 1, LEVAPLS2 = 0, LREGCL = 0;
static double CETA[MAXLEV];
static int NLEVP = 0;

void oracle_set_params(const double* r, int lphylin, int levapls2, int lregcl, int nlev, const double* ceta) {
  RG = r[0]; RD = r[1]; RCPD = r[2]; RETV = r[3]; RLVTT = r[4]; RLSTT = r[5]; RLMLT = r[6]; RTT = r[7];
  R2ES = r[8]; R3LES = r[9]; R3IES = r[10]; R4LES = r[11]; R4IES = r[12]; R5LES = r[13]; R5IES = r[14];
  R5ALVCP = r[15]; R5ALSCP = r[16]; RALVDCP = r[17]; RALSDCP = r[18]; RTWAT = r[19]; RTICE = r[20];
  RTWAT_RTICE_R = r[21]; RVTMP2 = r[22];
  RCLCRIT = r[23]; RKCONV = r[24]; RLMIN = r[25]; RPECONS = r[26]; RLPTRC = r[27];
  LPHYLIN = lphylin; LEVAPLS2 = levapls2; LREGCL = lregcl;
  NLEVP = nlev;
  for (int k = 0; k < nlev && k < MAXLEV; ++k) CETA[k] = ceta[k];
}

static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }
static double sq(double a) { return a * a; }
static double cube(double a) { return a * a * a; }

/* FOEALFA, src/common/include/fcttre.func.h:74-75 */
static double foealfa(double ptare) {
  return dmin(1.0, sq((dmax(RTICE, dmin(RTWAT, ptare)) - RTICE) * RTWAT_RTICE_R));
}

/* SATUR, LDPHYLIN branch: src/cloudsc2_nl/satur.F90:106-123 */
void oracle_satur(int kidia, int kfdia, int klon, int klev, const double* paprsf, const double* pt, double* pqsat) {
  const double zqmax = 0.5;
  for (int jk = 0; jk < klev; ++jk)
    for (int jl = kidia - 1; jl < kfdia; ++jl) {
      double ztarg = pt[jk * klon + jl];
      double zalfa = foealfa(ztarg);
      double zfoeewl = R2ES * exp(R3LES * (ztarg - RTT) / (ztarg - R4LES));
      double zfoeewi = R2ES * exp(R3IES * (ztarg - RTT) / (ztarg - R4IES));
      double zfoeew = zalfa * zfoeewl + (1.0 - zalfa) * zfoeewi;
      double zqs = zfoeew / paprsf[jk * klon + jl];
      if (zqs > zqmax) zqs = zqmax;
      double zcor = 1.0 / (1.0 - RETV * zqs);
      pqsat[jk * klon + jl] = zqs * zcor;
    }
}

/* critical relative humidity, cloudsc2.F90:384-399 (identical in cloudsc2tl.F90:515-530, cloudsc2ad.F90:499-514) */
static double crit_rh(double ztrpaus, double ceta) {
  double zeta3 = ztrpaus;
  double zrh1 = 1.0;
  double zrh2 = 0.35 + 0.14 * sq((zeta3 - 0.25) / 0.15) + 0.04 * dmin(zeta3 - 0.25, 0.0) / 0.15;
  double zrh3 = 1.0;
  double zdeta2 = 0.3;
  double zdeta1 = 0.09 + 0.16 * (0.4 - zeta3) / 0.3;
  double zcrh2 = 0.0;
  if (ceta < zeta3) zcrh2 = zrh3;
  else if (ceta >= zeta3 && ceta < (zeta3 + zdeta2)) zcrh2 = zrh3 + (zrh2 - zrh3) * ((ceta - zeta3) / zdeta2);
  else if (ceta >= (zeta3 + zdeta2) && ceta < (1.0 - zdeta1)) zcrh2 = zrh2;
  else if (ceta >= (1.0 - zdeta1)) zcrh2 = zrh1 + (zrh2 - zrh1) * sqrt((1.0 - ceta) / zdeta1);
  return zcrh2;
}

/* Eta value at tropopause, cloudsc2.F90:315-326 */
static double tropopause(int klev, const double* ztp1) {
  double ztrpaus = 0.1;
  for (int jk = 0; jk < klev - 1; ++jk)
    if (CETA[jk] > 0.1 && CETA[jk] < 0.4 && ztp1[jk] > ztp1[jk + 1]) ztrpaus = CETA[jk];
  return ztrpaus;
}

/* one iteration of CUADJTQS KCALL=0 (cuadjtqs.F90:216-229); returns through *pt, *pq */
static void cuadjtqs_iter(double zqp, double z3es, double z4es, double z5alcp, double zaldcp, double* pt, double* pq) {
  const double zqmax = 0.5;
  double ztarg = *pt;
  double zfoeew = R2ES * exp(z3es * (ztarg - RTT) / (ztarg - z4es));
  double zqsat = zqp * zfoeew;
  if (zqsat > zqmax) zqsat = zqmax;
  double zcor = 1.0 / (1.0 - RETV * zqsat);
  zqsat = zqsat * zcor;
  double z2s = z5alcp / sq(ztarg - z4es);
  double zcond1 = (*pq - zqsat) / (1.0 + zqsat * zcor * z2s);
  *pt = *pt + zaldcp * zcond1;
  *pq = *pq - zcond1;
}

#define A(name) name[jk * klon + jl]
#define A1(name) name[(jk + 1) * klon + jl]

/* ================================================================================================================
 * CLOUDSC2, src/cloudsc2_nl/cloudsc2.F90:10-741
 * ================================================================================================================ */
void oracle_cloudsc2(int kidia, int kfdia, int klon, int klev, int ldrain1d, double ptsphy, const double* paphp1,
                     const double* papp1, const double* pqm1, const double* pqs, const double* ptm1, const double* pl,
                     const double* pi, const double* plude, const double* plu, const double* pmfu, const double* pmfd,
                     double* ptent, const double* pgtent, double* ptenq, const double* pgtenq, double* ptenl,
                     const double* pgtenl, double* pteni, const double* pgteni, const double* psupsat, double* pclc,
                     double* pfplsl, double* pfplsn, double* pfhpsl, double* pfhpsn, double* pcovptot) {
  const double zscal = 0.9;
  /* :235-244 */
  const double zckcodtl = 2.0 * RKCONV * ptsphy;
  const double zckcodti = 5.0 * RKCONV * ptsphy;
  const double zcons2 = 1.0 / (ptsphy * RG);
  const double zcons3 = RLVTT / RCPD;
  const double zmeltp2 = RTT + 2.0;
  const double zqtmst = 1.0 / ptsphy;
  const double zqmax = 0.5, zeps1 = 1.e-12, zeps2 = 1.e-10;
  const int evap = LEVAPLS2 || ldrain1d;

  double zscalm[MAXLEV];
  for (int jk = 0; jk < klev; ++jk) zscalm[jk] = zscal * pow(dmax(CETA[jk] - 0.2, zeps1), 0.2); /* :266 */

  for (int jl = kidia - 1; jl < kfdia; ++jl) {
    double ztp1[MAXLEV], zqp1[MAXLEV], zl[MAXLEV], zi[MAXLEV], zdp[MAXLEV], zlfdcp[MAXLEV], zlsdcp[MAXLEV], zlvdcp[MAXLEV];
    /* :253-279 */
    for (int jk = 0; jk < klev; ++jk) {
      ztp1[jk] = A(ptm1) + ptsphy * A(pgtent);
      zqp1[jk] = A(pqm1) + ptsphy * A(pgtenq) + A(psupsat);
      zl[jk] = A(pl) + ptsphy * A(pgtenl);
      zi[jk] = A(pi) + ptsphy * A(pgteni);
    }
    for (int jk = 0; jk < klev; ++jk) {
      zdp[jk] = A1(paphp1) - A(paphp1);
      double zzz = 1.0 / (RCPD + RCPD * RVTMP2 * zqp1[jk]);
      zlfdcp[jk] = RLMLT * zzz;
      zlsdcp[jk] = RLSTT * zzz;
      zlvdcp[jk] = RLVTT * zzz;
    }
    /* :288-312 */
    for (int jk = 0; jk < klev; ++jk) { A(pclc) = 0.0; A(pcovptot) = 0.0; }
    double zrfl = 0.0, zsfl = 0.0, zcovptot = 0.0, zcovpclr = 0.0;
    pfplsl[jl] = 0.0;
    pfplsn[jl] = 0.0;
    double ztrpaus = tropopause(klev, ztp1);

    for (int jk = 0; jk < klev; ++jk) {
      double zqc = 0.0, zrfreeze = 0.0, zevapr = 0.0, zevaps = 0.0;
      /* :349-375 */
      double zfwat, z3es, z4es, zfoeew, zesdp;
      if (LPHYLIN || ldrain1d) {
        double zoealfaw = 0.545 * (tanh(0.17 * (ztp1[jk] - RLPTRC)) + 1.0);
        if (ztp1[jk] < RTT) { zfwat = zoealfaw; z3es = R3IES; z4es = R4IES; }
        else { zfwat = 1.0; z3es = R3LES; z4es = R4LES; }
        zfoeew = R2ES * exp(z3es * (ztp1[jk] - RTT) / (ztp1[jk] - z4es));
        zesdp = zfoeew / A(papp1);
        if (zesdp > zqmax) zesdp = zqmax;
      } else {
        /* FOEALFA/FOEEWM branch (:366-368): unreachable, every reference main forces LPHYLIN=.true. */
        zfwat = foealfa(ztp1[jk]);
        zfoeew = R2ES * (zfwat * exp(R3LES * (ztp1[jk] - RTT) / (ztp1[jk] - R4LES)) +
                         (1.0 - zfwat) * exp(R3IES * (ztp1[jk] - RTT) / (ztp1[jk] - R4IES)));
        zesdp = zfoeew / A(papp1);
      }
      double zfacw = R5LES / sq(ztp1[jk] - R4LES);
      double zfaci = R5IES / sq(ztp1[jk] - R4IES);
      double zfac = zfwat * zfacw + (1.0 - zfwat) * zfaci;
      double zcor = 1.0 / (1.0 - RETV * zesdp);
      double zdqsdtemp = zfac * zcor * A(pqs);
      double zcorqs = 1.0 + zcons3 * zdqsdtemp;
      /* :379-380 */
      double zqlim = zqp1[jk];
      if (zqp1[jk] > A(pqs)) zqlim = A(pqs);
      /* :384-407 */
      double zcrh2 = crit_rh(ztrpaus, CETA[jk]);
      double zsupsat = (ztp1[jk] < RTICE) ? 1.8 - 3.E-03 * ztp1[jk] : 1.0;
      double zqsat = A(pqs) * zsupsat;
      double zqcrit = zcrh2 * zqsat;
      /* :413-426 */
      double zqt = zqp1[jk] + zl[jk] + zi[jk];
      if (zqt <= zqcrit) {
        A(pclc) = 0.0; zqc = 0.0;
      } else if (zqt >= zqsat) {
        A(pclc) = 1.0; zqc = (1.0 - zscalm[jk]) * (zqsat - zqcrit);
      } else {
        double zqpd = zqsat - zqt;
        double zqcd = zqsat - zqcrit;
        A(pclc) = 1.0 - sqrt(zqpd / (zqcd - zscalm[jk] * (zqt - zqcrit)));
        zqc = (zscalm[jk] * zqpd + (1.0 - zscalm[jk]) * zqcd) * sq(A(pclc));
      }
      /* :432-443 */
      double zgdp = RG / (A1(paphp1) - A(paphp1));
      double zlude = A(plude) * ptsphy * zgdp;
      int llo1 = 0;
      if (jk < klev - 1) llo1 = zlude >= RLMIN && A1(plu) >= zeps2;
      if (llo1) {
        A(pclc) = A(pclc) + (1.0 - A(pclc)) * (1.0 - exp(-zlude / A1(plu)));
        zqc = zqc + zlude;
      }
      /* :449-459 */
      {
        double zfac1 = 1.0 / (RD * ztp1[jk]);
        double zrho = A(papp1) * zfac1;
        double zfac2 = 1.0 / (A(papp1) - RETV * zfoeew);
        double zrodqsdp = -zrho * A(pqs) * zfac2;
        double zldcp = zfwat * zlvdcp[jk] + (1.0 - zfwat) * zlsdcp[jk];
        double zfac3 = 1.0 / (1.0 + zldcp * zdqsdtemp);
        double dtdzmo = RG * (1.0 / RCPD - zldcp * zrodqsdp) * zfac3;
        double zdqsdz = zdqsdtemp * dtdzmo - RG * zrodqsdp;
        double zfac4 = 1.0 / zrho;
        double zdqc = dmin(zdqsdz * (A(pmfu) + A(pmfd)) * ptsphy * zfac4, zqc);
        zqc = zqc - zdqc;
      }
      /* :465-468 */
      double zqlwc = zqc * zfwat;
      double zqiwc = zqc * (1.0 - zfwat);
      double zcondl = (zqlwc - zl[jk]) * zqtmst;
      double zcondi = (zqiwc - zi[jk]) * zqtmst;
      /* :476-480 */
      if (A(pclc) > zcovptot) zcovptot = A(pclc);
      zcovpclr = zcovptot - A(pclc);
      zcovpclr = dmax(zcovpclr, 0.0);
      /* :488-497 */
      double zrfln, zsfln;
      if (zsfl != 0.0) {
        double zcons = zcons2 * zdp[jk] / zlfdcp[jk];
        double zsnmlt = dmin(zsfl, zcons * dmax(0.0, (ztp1[jk] - zmeltp2)));
        zrfln = zrfl + zsnmlt;
        zsfln = zsfl - zsnmlt;
        ztp1[jk] = ztp1[jk] - zsnmlt / zcons;
      } else {
        zrfln = zrfl;
        zsfln = zsfl;
      }
      /* :504-534 */
      double zprr, zprs;
      if (A(pclc) > zeps2) {
        double zlcrit = evap ? 1.9 * RCLCRIT : RCLCRIT * 2.;
        double zcldl = zqlwc / A(pclc);
        double zd = zckcodtl * (1.0 - exp(-sq(zcldl / zlcrit)));
        double zlnew = A(pclc) * zcldl * exp(-zd);
        zprr = zqlwc - zlnew;
        zqlwc = zqlwc - zprr;
      } else zprr = 0.0;
      if (A(pclc) > zeps2) {
        double zlcrit = evap ? 1.E-04 : RCLCRIT * 2.;
        double zcldi = zqiwc / A(pclc);
        double zd = zckcodti * exp(0.025 * (ztp1[jk] - RTT)) * (1.0 - exp(-sq(zcldi / zlcrit)));
        double zinew = A(pclc) * zcldi * exp(-zd);
        zprs = zqiwc - zinew;
        zqiwc = zqiwc - zprs;
      } else zprs = 0.0;
      /* :538-552 */
      double zdr = zcons2 * zdp[jk] * (zprr + zprs);
      double zfwatr;
      if (ztp1[jk] < RTT) { zrfreeze = zcons2 * zdp[jk] * zprr; zfwatr = 0.0; }
      else zfwatr = 1.0;
      double zrn = zfwatr * zdr;
      double zsn = (1.0 - zfwatr) * zdr;
      zrfln = zrfln + zrn;
      zsfln = zsfln + zsn;
      /* :556-591 */
      double zprtot = zrfln + zsfln;
      int llo2 = zprtot > zeps2 && zcovpclr > zeps2 && evap;
      if (llo2) {
        double zpreclr = zprtot * zcovpclr / zcovptot;
        double zqe = A(pqs) - (A(pqs) - zqlim) * zcovpclr / sq(1.0 - A(pclc));
        double zbeta = RG * RPECONS * pow(sqrt(A(papp1) / paphp1[klev * klon + jl]) / 5.09E-3 * zpreclr / zcovpclr, 0.5777);
        double zb = ptsphy * zbeta * (A(pqs) - zqe) / (1.0 + zbeta * ptsphy * zcorqs);
        double zdtgdp = ptsphy * RG / (A1(paphp1) - A(paphp1));
        double zdpr = zcovpclr * zb / zdtgdp;
        zdpr = dmin(zdpr, zpreclr);
        zpreclr = zpreclr - zdpr;
        if (zpreclr <= 0.0) zcovptot = A(pclc);
        A(pcovptot) = zcovptot;
        zevapr = zdpr * zrfln / zprtot;
        zrfln = zrfln - zevapr;
        zevaps = zdpr * zsfln / zprtot;
        zsfln = zsfln - zevaps;
      }
      /* :602-617 */
      double zdqdt = -(zcondl + zcondi) + (A(plude) + zevapr + zevaps) * zgdp;
      double zdtdt = zlvdcp[jk] * zcondl + zlsdcp[jk] * zcondi -
                     (zlvdcp[jk] * zevapr + zlsdcp[jk] * zevaps + A(plude) * (zfwat * zlvdcp[jk] + (1.0 - zfwat) * zlsdcp[jk]) -
                      (zlsdcp[jk] - zlvdcp[jk]) * zrfreeze) * zgdp;
      ztp1[jk] = ztp1[jk] + ptsphy * zdtdt;
      zqp1[jk] = zqp1[jk] + ptsphy * zdqdt;
      double zpp = A(papp1);
      double zqold = zqp1[jk];
      /* :630-669 (inlined CUADJTQS) */
      {
        double z3, z4, z5alcp, zaldcp;
        if (ztp1[jk] > RTT) { z3 = R3LES; z4 = R4LES; z5alcp = R5ALVCP; zaldcp = RALVDCP; }
        else { z3 = R3IES; z4 = R4IES; z5alcp = R5ALSCP; zaldcp = RALSDCP; }
        double zqp = 1.0 / zpp;
        cuadjtqs_iter(zqp, z3, z4, z5alcp, zaldcp, &ztp1[jk], &zqp1[jk]);
        cuadjtqs_iter(zqp, z3, z4, z5alcp, zaldcp, &ztp1[jk], &zqp1[jk]);
      }
      /* :673-691 */
      double zdq = dmax(0.0, zqold - zqp1[jk]);
      double zdr2 = zcons2 * zdp[jk] * zdq;
      double zrfreeze2;
      if (ztp1[jk] < RTT) { zrfreeze2 = zfwat * zdr2; zfwatr = 0.0; }
      else { zrfreeze2 = 0.0; zfwatr = 1.0; }
      zrn = zfwatr * zdr2;
      zsn = (1.0 - zfwatr) * zdr2;
      zcondl = zcondl + zfwatr * zdq * zqtmst;
      zcondi = zcondi + (1.0 - zfwatr) * zdq * zqtmst;
      zrfln = zrfln + zrn;
      zsfln = zsfln + zsn;
      zrfreeze = zrfreeze + zrfreeze2;
      /* :695-715 */
      zdqdt = -(zcondl + zcondi) + (A(plude) + zevapr + zevaps) * zgdp;
      zdtdt = zlvdcp[jk] * zcondl + zlsdcp[jk] * zcondi -
              (zlvdcp[jk] * zevapr + zlsdcp[jk] * zevaps + A(plude) * (zfwat * zlvdcp[jk] + (1.0 - zfwat) * zlsdcp[jk]) -
               (zlsdcp[jk] - zlvdcp[jk]) * zrfreeze) * zgdp;
      A(ptenq) = zdqdt;
      A(ptent) = zdtdt;
      A(ptenl) = (zqlwc - zl[jk]) * zqtmst;
      A(pteni) = (zqiwc - zi[jk]) * zqtmst;
      A1(pfplsl) = zrfln;
      A1(pfplsn) = zsfln;
      /* :720-723 */
      zrfl = zrfln;
      zsfl = zsfln;
    }
    /* :730-735 */
    for (int jk = 0; jk <= klev; ++jk) {
      A(pfhpsl) = -A(pfplsl) * RLVTT;
      A(pfhpsn) = -A(pfplsn) * RLSTT;
    }
  }
}

/* ================================================================================================================
 * CLOUDSC2TL, src/cloudsc2_tl/cloudsc2tl.F90:10-1119 (+ CUADJTQSTL KCALL=0, cuadjtqstl.F90:333-405)
 * "5" = trajectory, plain = perturbation.
 * ================================================================================================================ */
static void cuadjtqstl_iter(double zqp5, double zqp, double z3es, double z4es, double z5alcp, double zaldcp, double* pt5,
                            double* pq5, double* pt, double* pq) {
  const double zqmax = 0.5;
  double ztarg = *pt, ztarg5 = *pt5;
  double zfoeew5 = R2ES * exp(z3es * (ztarg5 - RTT) / (ztarg5 - z4es));
  double zfoeew = z3es * (RTT - z4es) * ztarg * zfoeew5 / sq(ztarg5 - z4es);
  double zqsat = zqp5 * zfoeew + zqp * zfoeew5;
  double zqsat5 = zqp5 * zfoeew5;
  if (zqsat5 > zqmax) { zqsat = 0.0; zqsat5 = zqmax; }
  double zcor = (RETV * zqsat) / sq(1.0 - RETV * zqsat5);
  double zcor5 = 1.0 / (1.0 - RETV * zqsat5);
  zqsat = zqsat5 * zcor + zqsat * zcor5;
  zqsat5 = zqsat5 * zcor5;
  double z2s = -2.0 * ztarg * z5alcp / cube(ztarg5 - z4es);
  double z2s5 = z5alcp / sq(ztarg5 - z4es);
  double zcond1 = (*pq - zqsat) / (1.0 + zqsat5 * zcor5 * z2s5) -
                  (*pq5 - zqsat5) * (zqsat * zcor5 * z2s5 + zqsat5 * zcor * z2s5 + zqsat5 * zcor5 * z2s) /
                      sq(1.0 + zqsat5 * zcor5 * z2s5);
  double zcond15 = (*pq5 - zqsat5) / (1.0 + zqsat5 * zcor5 * z2s5);
  *pt = *pt + zaldcp * zcond1;
  *pt5 = *pt5 + zaldcp * zcond15;
  *pq = *pq - zcond1;
  *pq5 = *pq5 - zcond15;
}

void oracle_cloudsc2tl(int kidia, int kfdia, int klon, int klev, int ldrain1d, double ptsphy,
                       /* trajectory */
                       const double* paphp15, const double* papp15, const double* pqm15, const double* pqs5, const double* ptm15,
                       const double* pl5, const double* pi5, const double* plude5, const double* plu5, const double* pmfu5,
                       const double* pmfd5, double* ptent5, const double* pgtent5, double* ptenq5, const double* pgtenq5,
                       double* ptenl5, const double* pgtenl5, double* pteni5, const double* pgteni5, const double* psupsat5,
                       double* pclc5, double* pfplsl5, double* pfplsn5, double* pfhpsl5, double* pfhpsn5, double* pcovptot5,
                       /* perturbation */
                       const double* paphp1, const double* papp1, const double* pqm1, const double* pqs, const double* ptm1,
                       const double* pl, const double* pi, const double* plude, const double* plu, const double* pmfu,
                       const double* pmfd, double* ptent, const double* pgtent, double* ptenq, const double* pgtenq,
                       double* ptenl, const double* pgtenl, double* pteni, const double* pgteni, const double* psupsat,
                       double* pclc, double* pfplsl, double* pfplsn, double* pfhpsl, double* pfhpsn, double* pcovptot) {
  const double zscal = 0.9;
  const double zckcodtl = 2.0 * RKCONV * ptsphy, zckcodti = 5.0 * RKCONV * ptsphy;
  const double zckcodtla = zckcodtl / 100., zckcodtia = zckcodti / 100.;
  const double zcons2 = 1.0 / (ptsphy * RG), zcons3 = RLVTT / RCPD, zmeltp2 = RTT + 2.0, zqtmst = 1.0 / ptsphy;
  const double zqmax = 0.5, zeps1 = 1.e-12, zeps2 = 1.e-10;
  const int evap = LEVAPLS2 || ldrain1d;
  double zscalm[MAXLEV];
  for (int jk = 0; jk < klev; ++jk) zscalm[jk] = zscal * pow(dmax(CETA[jk] - 0.2, zeps1), 0.2);

  for (int jl = kidia - 1; jl < kfdia; ++jl) {
    double ztp1[MAXLEV], ztp15[MAXLEV], zqp1[MAXLEV], zqp15[MAXLEV], zl[MAXLEV], zl5[MAXLEV], zi[MAXLEV], zi5[MAXLEV];
    double zdp[MAXLEV], zdp5[MAXLEV], zlfdcp[MAXLEV], zlfdcp5[MAXLEV], zlsdcp[MAXLEV], zlsdcp5[MAXLEV], zlvdcp[MAXLEV],
        zlvdcp5[MAXLEV];
    /* :341-376 */
    for (int jk = 0; jk < klev; ++jk) {
      ztp1[jk] = A(ptm1) + ptsphy * A(pgtent);
      ztp15[jk] = A(ptm15) + ptsphy * A(pgtent5);
      zqp1[jk] = A(pqm1) + ptsphy * A(pgtenq) + A(psupsat);
      zqp15[jk] = A(pqm15) + ptsphy * A(pgtenq5) + A(psupsat5);
      zl[jk] = A(pl) + ptsphy * A(pgtenl);
      zl5[jk] = A(pl5) + ptsphy * A(pgtenl5);
      zi[jk] = A(pi) + ptsphy * A(pgteni);
      zi5[jk] = A(pi5) + ptsphy * A(pgteni5);
    }
    for (int jk = 0; jk < klev; ++jk) {
      zdp[jk] = A1(paphp1) - A(paphp1);
      zdp5[jk] = A1(paphp15) - A(paphp15);
      double zzz = -RCPD * RVTMP2 * zqp1[jk] / sq(RCPD + RCPD * RVTMP2 * zqp15[jk]);
      double zzz5 = 1.0 / (RCPD + RCPD * RVTMP2 * zqp15[jk]);
      zlfdcp[jk] = RLMLT * zzz;  zlfdcp5[jk] = RLMLT * zzz5;
      zlsdcp[jk] = RLSTT * zzz;  zlsdcp5[jk] = RLSTT * zzz5;
      zlvdcp[jk] = RLVTT * zzz;  zlvdcp5[jk] = RLVTT * zzz5;
    }
    /* :386-426 */
    for (int jk = 0; jk < klev; ++jk) { A(pclc) = 0.0; A(pclc5) = 0.0; A(pcovptot) = 0.0; A(pcovptot5) = 0.0; }
    double zrfl = 0.0, zrfl5 = 0.0, zsfl = 0.0, zsfl5 = 0.0;
    double zcovptot = 0.0, zcovptot5 = 0.0, zcovpclr = 0.0, zcovpclr5 = 0.0;
    pfplsl[jl] = 0.0; pfplsl5[jl] = 0.0; pfplsn[jl] = 0.0; pfplsn5[jl] = 0.0;
    double ztrpaus = tropopause(klev, ztp15); /* :429-440 */

    for (int jk = 0; jk < klev; ++jk) {
      double zqc = 0.0, zqc5 = 0.0, zrfreeze = 0.0, zrfreeze5 = 0.0, zevapr = 0.0, zevapr5 = 0.0, zevaps = 0.0, zevaps5 = 0.0;
      /* :463-501 */
      double zoealfaw = 0.545 * 0.17 * ztp1[jk] / sq(cosh(0.17 * (ztp15[jk] - RLPTRC)));
      double zoealfaw5 = 0.545 * (tanh(0.17 * (ztp15[jk] - RLPTRC)) + 1.0);
      double zfwat, zfwat5, z3es, z4es;
      if (ztp15[jk] < RTT) { zfwat = zoealfaw; zfwat5 = zoealfaw5; z3es = R3IES; z4es = R4IES; }
      else { zfwat = 0.0; zfwat5 = 1.0; z3es = R3LES; z4es = R4LES; }
      double zfoeew5 = R2ES * exp(z3es * (ztp15[jk] - RTT) / (ztp15[jk] - z4es));
      double zfoeew = z3es * (RTT - z4es) * ztp1[jk] * zfoeew5 / sq(ztp15[jk] - z4es);
      double zesdp = zfoeew / A(papp15) - A(papp1) * zfoeew5 / sq(A(papp15));
      double zesdp5 = zfoeew5 / A(papp15);
      if (zesdp5 > zqmax) { zesdp = 0.0; zesdp5 = zqmax; }
      double zfacw = -2.0 * R5LES * ztp1[jk] / cube(ztp15[jk] - R4LES);
      double zfacw5 = R5LES / sq(ztp15[jk] - R4LES);
      double zfaci = -2.0 * R5IES * ztp1[jk] / cube(ztp15[jk] - R4IES);
      double zfaci5 = R5IES / sq(ztp15[jk] - R4IES);
      double zfac = zfwat5 * zfacw + zfacw5 * zfwat + (1.0 - zfwat5) * zfaci - zfaci5 * zfwat;
      double zfac5 = zfwat5 * zfacw5 + (1.0 - zfwat5) * zfaci5;
      double zcor = RETV * zesdp / sq(1.0 - RETV * zesdp5);
      double zcor5 = 1.0 / (1.0 - RETV * zesdp5);
      double zdqsdtemp = zfac5 * zcor5 * A(pqs) + zfac5 * A(pqs5) * zcor + zcor5 * A(pqs5) * zfac;
      double zdqsdtemp5 = zfac5 * zcor5 * A(pqs5);
      double zcorqs = zcons3 * zdqsdtemp;
      double zcorqs5 = 1.0 + zcons3 * zdqsdtemp5;
      /* :505-511 */
      double zqlim, zqlim5;
      if (zqp15[jk] > A(pqs5)) { zqlim = A(pqs); zqlim5 = A(pqs5); }
      else { zqlim = zqp1[jk]; zqlim5 = zqp15[jk]; }
      /* :515-543 */
      double zcrh2 = crit_rh(ztrpaus, CETA[jk]);
      double zsupsat5, zsupsat;
      if (ztp15[jk] < RTICE) { zsupsat5 = 1.8 - 3.E-03 * ztp15[jk]; zsupsat = -3.E-03 * ztp1[jk]; }
      else { zsupsat5 = 1.0; zsupsat = 0.0; }
      double zqsat5 = A(pqs5) * zsupsat5;
      double zqsat = A(pqs) * zsupsat5 + A(pqs5) * zsupsat;
      double zqcrit5 = zcrh2 * zqsat5;
      double zqcrit = zcrh2 * zqsat;
      /* :549-589 */
      double zqt = zqp1[jk] + zl[jk] + zi[jk];
      double zqt5 = zqp15[jk] + zl5[jk] + zi5[jk];
      if (zqt5 <= zqcrit5) {
        A(pclc) = 0.0; A(pclc5) = 0.0; zqc = 0.0; zqc5 = 0.0;
      } else if (zqt5 >= zqsat5) {
        A(pclc) = 0.0; A(pclc5) = 1.0;
        zqc = (1.0 - zscalm[jk]) * (zqsat - zqcrit);
        zqc5 = (1.0 - zscalm[jk]) * (zqsat5 - zqcrit5);
      } else {
        double zqpd = zqsat - zqt, zqpd5 = zqsat5 - zqt5;
        double zqcd = zqsat - zqcrit, zqcd5 = zqsat5 - zqcrit5;
        double zsqrt5 = sqrt(zqpd5 / (zqcd5 - zscalm[jk] * (zqt5 - zqcrit5)));
        A(pclc5) = 1.0 - zsqrt5;
        A(pclc) = -(0.5 / zsqrt5) *
                  (zqpd * (zqcd5 - zscalm[jk] * (zqt5 - zqcrit5)) - zqpd5 * (zqcd - zscalm[jk] * (zqt - zqcrit))) /
                  sq(zqcd5 - zscalm[jk] * (zqt5 - zqcrit5));
        if (LREGCL) {
          double zrat = zqpd5 / zqcd5;
          double zyyy = dmin(0.3, 3.5 * sqrt(zrat * cube(1.0 - zscalm[jk] * (1.0 - zrat))) / (1.0 - zscalm[jk]));
          A(pclc) = zyyy * A(pclc);
        }
        zqc = (zscalm[jk] * zqpd + (1.0 - zscalm[jk]) * zqcd) * sq(A(pclc5)) +
              (zscalm[jk] * zqpd5 + (1.0 - zscalm[jk]) * zqcd5) * 2.0 * A(pclc5) * A(pclc);
        zqc5 = (zscalm[jk] * zqpd5 + (1.0 - zscalm[jk]) * zqcd5) * sq(A(pclc5));
      }
      /* :595-622 */
      double zgdp = -RG * (A1(paphp1) - A(paphp1)) / sq(A1(paphp15) - A(paphp15));
      double zgdp5 = RG / (A1(paphp15) - A(paphp15));
      double zlude = ptsphy * zgdp5 * A(plude) + ptsphy * A(plude5) * zgdp;
      double zlude5 = A(plude5) * ptsphy * zgdp5;
      int llo1 = 0;
      if (jk < klev - 1) llo1 = zlude5 >= RLMIN && A1(plu5) >= zeps2;
      if (llo1) {
        double e = exp(-zlude5 / A1(plu5));
        A(pclc) = A(pclc) - A(pclc) * (1.0 - e) + ((1.0 - A(pclc5)) / A1(plu5)) * e * zlude -
                  ((1.0 - A(pclc5)) * zlude5 / sq(A1(plu5))) * e * A1(plu);
        A(pclc5) = A(pclc5) + (1.0 - A(pclc5)) * (1.0 - e);
        zqc = zqc + zlude;
        zqc5 = zqc5 + zlude5;
      }
      /* :628-664 */
      {
        double zfac1 = 1.0 / (RD * ztp15[jk]);
        double zrho = (A(papp1) - ztp1[jk] * A(papp15) / ztp15[jk]) * zfac1;
        double zrho5 = A(papp15) * zfac1;
        double zfac2 = 1.0 / (A(papp15) - RETV * zfoeew5);
        double zrodqsdp = (-zrho * A(pqs5) - zrho5 * A(pqs) + zrho5 * A(pqs5) * (A(papp1) - RETV * zfoeew) * zfac2) * zfac2;
        double zrodqsdp5 = -zrho5 * A(pqs5) * zfac2;
        double zldcp = zfwat * zlvdcp5[jk] + zfwat5 * zlvdcp[jk] + (1.0 - zfwat5) * zlsdcp[jk] - zfwat * zlsdcp5[jk];
        double zldcp5 = zfwat5 * zlvdcp5[jk] + (1.0 - zfwat5) * zlsdcp5[jk];
        double zfac3 = 1.0 / (1.0 + zldcp5 * zdqsdtemp5);
        double dtdzmo5 = RG * (1.0 / RCPD - zldcp5 * zrodqsdp5) * zfac3;
        double dtdzmo = -(RG * (zldcp * zrodqsdp5 + zldcp5 * zrodqsdp) + dtdzmo5 * (zldcp5 * zdqsdtemp + zldcp * zdqsdtemp5)) * zfac3;
        double zdqsdz = zdqsdtemp5 * dtdzmo + zdqsdtemp * dtdzmo5 - RG * zrodqsdp;
        double zdqsdz5 = zdqsdtemp5 * dtdzmo5 - RG * zrodqsdp5;
        double zfac4 = 1.0 / zrho5;
        int llo3 = (zdqsdz5 * (A(pmfu5) + A(pmfd5)) * ptsphy * zfac4 < zqc5);
        double zdqc, zdqc5;
        if (llo3) {
          zdqc5 = zdqsdz5 * (A(pmfu5) + A(pmfd5)) * ptsphy * zfac4;
          zdqc = (ptsphy * (zdqsdz * (A(pmfu5) + A(pmfd5)) + zdqsdz5 * (A(pmfu) + A(pmfd))) - zdqc5 * zrho) * zfac4;
          if (LREGCL) zdqc = zdqc * 0.1;
        } else { zdqc5 = zqc5; zdqc = zqc; }
        zqc = zqc - zdqc;
        zqc5 = zqc5 - zdqc5;
      }
      /* :670-680 */
      double zqlwc = zqc * zfwat5 + zqc5 * zfwat;
      double zqlwc5 = zqc5 * zfwat5;
      double zqiwc = zqc * (1.0 - zfwat5) - zqc5 * zfwat;
      double zqiwc5 = zqc5 * (1.0 - zfwat5);
      double zcondl = (zqlwc - zl[jk]) * zqtmst, zcondl5 = (zqlwc5 - zl5[jk]) * zqtmst;
      double zcondi = (zqiwc - zi[jk]) * zqtmst, zcondi5 = (zqiwc5 - zi5[jk]) * zqtmst;
      /* :687-696 */
      if (A(pclc5) > zcovptot5) { zcovptot = A(pclc); zcovptot5 = A(pclc5); }
      zcovpclr = zcovptot - A(pclc);
      zcovpclr5 = zcovptot5 - A(pclc5);
      if (zcovpclr5 < 0.0) { zcovpclr = 0.0; zcovpclr5 = 0.0; }
      /* :704-733 */
      double zrfln, zrfln5, zsfln, zsfln5;
      if (zsfl5 != 0.0) {
        double zcons = zcons2 * (zdp[jk] * zlfdcp5[jk] - zdp5[jk] * zlfdcp[jk]) / sq(zlfdcp5[jk]);
        double zcons5 = zcons2 * zdp5[jk] / zlfdcp5[jk];
        double zz2s, zz2s5;
        if ((ztp15[jk] - zmeltp2) > 0.0) { zz2s = zcons5 * ztp1[jk] + zcons * (ztp15[jk] - zmeltp2); zz2s5 = zcons5 * (ztp15[jk] - zmeltp2); }
        else { zz2s = 0.0; zz2s5 = 0.0; }
        double zsnmlt, zsnmlt5;
        if (zsfl5 <= zz2s5) { zsnmlt = zsfl; zsnmlt5 = zsfl5; }
        else { zsnmlt = zz2s; zsnmlt5 = zz2s5; }
        zrfln = zrfl + zsnmlt;  zrfln5 = zrfl5 + zsnmlt5;
        zsfln = zsfl - zsnmlt;  zsfln5 = zsfl5 - zsnmlt5;
        ztp1[jk] = ztp1[jk] - (zsnmlt * zcons5 - zcons * zsnmlt5) / sq(zcons5);
        ztp15[jk] = ztp15[jk] - zsnmlt5 / zcons5;
      } else { zrfln = zrfl; zrfln5 = zrfl5; zsfln = zsfl; zsfln5 = zsfl5; }
      /* :739-814 */
      double zprr, zprr5, zprs, zprs5;
      if (A(pclc5) > zeps2) {
        double zlcrit = evap ? 1.9 * RCLCRIT : RCLCRIT * 2.;
        double zcldl = zqlwc / A(pclc5) - zqlwc5 * A(pclc) / sq(A(pclc5));
        double zcldl5 = zqlwc5 / A(pclc5);
        double zexp35 = exp(-sq(zcldl5 / zlcrit));
        double zd5 = zckcodtl * (1.0 - zexp35);
        double zexpdl5 = exp(-zd5);
        double zd;
        if (LREGCL) zd = (2.0 * zckcodtla / sq(zlcrit)) * exp(-sq(zcldl5 / zlcrit)) * zcldl5 * zcldl;
        else zd = (2.0 * zckcodtl / sq(zlcrit)) * exp(-sq(zcldl5 / zlcrit)) * zcldl5 * zcldl;
        double zlnew = zcldl5 * zexpdl5 * A(pclc) + A(pclc5) * zexpdl5 * zcldl - A(pclc5) * zcldl5 * zexpdl5 * zd;
        double zlnew5 = A(pclc5) * zcldl5 * zexpdl5;
        zprr = zqlwc - zlnew;  zprr5 = zqlwc5 - zlnew5;
        zqlwc = zqlwc - zprr;  zqlwc5 = zqlwc5 - zprr5;
      } else { zprr = 0.0; zprr5 = 0.0; }
      if (A(pclc5) > zeps2) {
        double zlcrit = evap ? 1.E-04 : RCLCRIT * 2.;
        double zcldi = zqiwc / A(pclc5) - zqiwc5 * A(pclc) / sq(A(pclc5));
        double zcldi5 = zqiwc5 / A(pclc5);
        double zexp15 = exp(0.025 * (ztp15[jk] - RTT));
        double zexp25 = exp(-sq(zcldi5 / zlcrit));
        double zd5 = zckcodti * zexp15 * (1.0 - zexp25);
        double zexpdi5 = exp(-zd5);
        double ck = LREGCL ? zckcodtia : zckcodti;
        double zd = ck * zexp15 * (zexp25 * (2.0 * zcldi5 * zcldi / sq(zlcrit) - 0.025 * ztp1[jk]) + 0.025 * ztp1[jk]);
        double zinew = zcldi5 * zexpdi5 * A(pclc) + A(pclc5) * zexpdi5 * zcldi - A(pclc5) * zcldi5 * zexpdi5 * zd;
        double zinew5 = A(pclc5) * zcldi5 * zexpdi5;
        zprs = zqiwc - zinew;  zprs5 = zqiwc5 - zinew5;
        zqiwc = zqiwc - zprs;  zqiwc5 = zqiwc5 - zprs5;
      } else { zprs = 0.0; zprs5 = 0.0; }
      /* :818-840 */
      double zdr = zcons2 * (zdp5[jk] * (zprr + zprs) + zdp[jk] * (zprr5 + zprs5));
      double zdr5 = zcons2 * zdp5[jk] * (zprr5 + zprs5);
      double zfwatr5, zfwatr = 0.0;
      if (ztp15[jk] < RTT) {
        zrfreeze5 = zcons2 * zdp5[jk] * zprr5;
        zrfreeze = zcons2 * (zdp[jk] * zprr5 + zdp5[jk] * zprr);
        zfwatr5 = 0.0;
      } else zfwatr5 = 1.0;
      double zrn = zfwatr5 * zdr + zdr5 * zfwatr, zrn5 = zfwatr5 * zdr5;
      double zsn = -zdr5 * zfwatr + (1.0 - zfwatr5) * zdr, zsn5 = (1.0 - zfwatr5) * zdr5;
      zrfln = zrfln + zrn;  zrfln5 = zrfln5 + zrn5;
      zsfln = zsfln + zsn;  zsfln5 = zsfln5 + zsn5;
      /* :844-936 */
      double zprtot = zrfln + zsfln, zprtot5 = zrfln5 + zsfln5;
      int llo2 = zprtot5 > zeps2 && zcovpclr5 > zeps2 && evap;
      if (llo2) {
        double psurf5 = paphp15[klev * klon + jl], psurf = paphp1[klev * klon + jl];
        double zpreclr = (zprtot5 * zcovpclr + zcovpclr5 * zprtot) / zcovptot5 - zprtot5 * zcovpclr5 * zcovptot / sq(zcovptot5);
        double zpreclr5 = zprtot5 * zcovpclr5 / zcovptot5;
        double zqe = A(pqs) - ((A(pqs5) - zqlim5) * zcovpclr + zcovpclr5 * A(pqs) - zcovpclr5 * zqlim) / sq(1.0 - A(pclc5)) -
                     2.0 * (A(pqs5) - zqlim5) * zcovpclr5 * A(pclc) / cube(1.0 - A(pclc5));
        double zqe5 = A(pqs5) - (A(pqs5) - zqlim5) * zcovpclr5 / sq(1.0 - A(pclc5));
        double zbeta = 0.5777 * (RG * RPECONS / 5.09E-3) *
                       pow(5.09E-3 * zcovpclr5 / (zpreclr5 * sqrt(A(papp15) / psurf5)), 0.4223) *
                       ((sqrt(A(papp15) / psurf5) * zpreclr + 0.5 * zpreclr5 * A(papp1) / sqrt(A(papp15) * psurf5) -
                         0.5 * zpreclr5 * sqrt(A(papp15) / psurf5) * psurf / psurf5) / zcovpclr5 -
                        zpreclr5 * sqrt(A(papp15) / psurf5) * zcovpclr / sq(zcovpclr5));
        double zbeta5 = RG * RPECONS * pow(sqrt(A(papp15) / psurf5) / 5.09E-3 * zpreclr5 / zcovpclr5, 0.5777);
        double zb = ptsphy * ((A(pqs5) - zqe5) * zbeta + zbeta5 * A(pqs) - zbeta5 * zqe) / (1.0 + zbeta5 * ptsphy * zcorqs5) -
                    sq(ptsphy) * zbeta5 * (A(pqs5) - zqe5) * (zbeta5 * zcorqs + zcorqs5 * zbeta) / sq(1.0 + zbeta5 * ptsphy * zcorqs5);
        double zb5 = ptsphy * zbeta5 * (A(pqs5) - zqe5) / (1.0 + zbeta5 * ptsphy * zcorqs5);
        double zdtgdp = -ptsphy * RG * (A1(paphp1) - A(paphp1)) / sq(A1(paphp15) - A(paphp15));
        double zdtgdp5 = ptsphy * RG / (A1(paphp15) - A(paphp15));
        double zdpr = (zcovpclr5 * zb + zb5 * zcovpclr) / zdtgdp5 - zcovpclr5 * zb5 * zdtgdp / sq(zdtgdp5);
        double zdpr5 = zcovpclr5 * zb5 / zdtgdp5;
        if (zdpr5 > zpreclr5) { zdpr = zpreclr; zdpr5 = zpreclr5; }
        zpreclr = zpreclr - zdpr;
        zpreclr5 = zpreclr5 - zdpr5;
        if (zpreclr5 <= 0.0) { zcovptot = A(pclc); zcovptot5 = A(pclc5); }
        A(pcovptot) = zcovptot;
        A(pcovptot5) = zcovptot5;
        zevapr = (zdpr5 * zrfln + zrfln5 * zdpr) / zprtot5 - zdpr5 * zrfln5 * zprtot / sq(zprtot5);
        zevapr5 = zdpr5 * zrfln5 / zprtot5;
        zrfln = zrfln - zevapr;  zrfln5 = zrfln5 - zevapr5;
        zevaps = (zdpr5 * zsfln + zsfln5 * zdpr) / zprtot5 - zdpr5 * zsfln5 * zprtot / sq(zprtot5);
        zevaps5 = zdpr5 * zsfln5 / zprtot5;
        zsfln = zsfln - zevaps;  zsfln5 = zsfln5 - zevaps5;
      }
      /* :943-982 and :1043-1071 share this form */
#define TL_TENDENCIES(ZDQDT, ZDQDT5, ZDTDT, ZDTDT5)                                                                          \
  ZDQDT = -(zcondl + zcondi) + (A(plude) + zevapr + zevaps) * zgdp5 + (A(plude5) + zevapr5 + zevaps5) * zgdp;               \
  ZDQDT5 = -(zcondl5 + zcondi5) + (A(plude5) + zevapr5 + zevaps5) * zgdp5;                                                  \
  ZDTDT = zlvdcp[jk] * zcondl5 + zlsdcp[jk] * zcondi5 + zlvdcp5[jk] * zcondl + zlsdcp5[jk] * zcondi -                        \
          (zlvdcp[jk] * zevapr5 + zlsdcp[jk] * zevaps5 + zlvdcp5[jk] * zevapr + zlsdcp5[jk] * zevaps +                       \
           A(plude) * (zfwat5 * zlvdcp5[jk] + (1.0 - zfwat5) * zlsdcp5[jk]) +                                               \
           A(plude5) * (zfwat * (zlvdcp5[jk] - zlsdcp5[jk]) + (zfwat5 * zlvdcp[jk] + (1.0 - zfwat5) * zlsdcp[jk])) -         \
           (zlsdcp[jk] - zlvdcp[jk]) * zrfreeze5 - (zlsdcp5[jk] - zlvdcp5[jk]) * zrfreeze) * zgdp5 -                         \
          (zlvdcp5[jk] * zevapr5 + zlsdcp5[jk] * zevaps5 + A(plude5) * (zfwat5 * zlvdcp5[jk] + (1.0 - zfwat5) * zlsdcp5[jk]) - \
           (zlsdcp5[jk] - zlvdcp5[jk]) * zrfreeze5) * zgdp;                                                                  \
  ZDTDT5 = zlvdcp5[jk] * zcondl5 + zlsdcp5[jk] * zcondi5 -                                                                   \
           (zlvdcp5[jk] * zevapr5 + zlsdcp5[jk] * zevaps5 + A(plude5) * (zfwat5 * zlvdcp5[jk] + (1.0 - zfwat5) * zlsdcp5[jk]) - \
            (zlsdcp5[jk] - zlvdcp5[jk]) * zrfreeze5) * zgdp5;
      double zdqdt, zdqdt5, zdtdt, zdtdt5;
      TL_TENDENCIES(zdqdt, zdqdt5, zdtdt, zdtdt5)
      ztp1[jk] = ztp1[jk] + ptsphy * zdtdt;    ztp15[jk] = ztp15[jk] + ptsphy * zdtdt5;
      zqp1[jk] = zqp1[jk] + ptsphy * zdqdt;    zqp15[jk] = zqp15[jk] + ptsphy * zdqdt5;
      double zqold = zqp1[jk], zqold5 = zqp15[jk];
      /* :987-991 CUADJTQSTL */
      {
        double z3, z4, z5alcp, zaldcp;
        if (ztp15[jk] > RTT) { z3 = R3LES; z4 = R4LES; z5alcp = R5ALVCP; zaldcp = RALVDCP; }
        else { z3 = R3IES; z4 = R4IES; z5alcp = R5ALSCP; zaldcp = RALSDCP; }
        double zqp = -A(papp1) / sq(A(papp15)), zqp5 = 1.0 / A(papp15);
        cuadjtqstl_iter(zqp5, zqp, z3, z4, z5alcp, zaldcp, &ztp15[jk], &zqp15[jk], &ztp1[jk], &zqp1[jk]);
        cuadjtqstl_iter(zqp5, zqp, z3, z4, z5alcp, zaldcp, &ztp15[jk], &zqp15[jk], &ztp1[jk], &zqp1[jk]);
      }
      /* :994-1039 */
      double zdq, zdq5;
      if ((zqold5 - zqp15[jk]) >= 0.0) {
        zdq5 = zqold5 - zqp15[jk];
        zdq = zqold - zqp1[jk];
        if (LREGCL) zdq = zdq * 0.7;
      } else { zdq = 0.0; zdq5 = 0.0; }
      double zdr2 = zcons2 * (zdp5[jk] * zdq + zdq5 * zdp[jk]);
      double zdr25 = zcons2 * zdp5[jk] * zdq5;
      double zrfreeze25, zrfreeze2;
      if (ztp15[jk] < RTT) { zrfreeze25 = zfwat5 * zdr25; zrfreeze2 = zfwat * zdr25 + zfwat5 * zdr2; zfwatr5 = 0.0; zfwatr = 0.0; }
      else { zrfreeze25 = 0.0; zrfreeze2 = 0.0; zfwatr5 = 1.0; zfwatr = 0.0; }
      zrn = zfwatr5 * zdr2 + zdr25 * zfwatr;  zrn5 = zfwatr5 * zdr25;
      zsn = (1.0 - zfwatr5) * zdr2 - zdr25 * zfwatr;  zsn5 = (1.0 - zfwatr5) * zdr25;
      zcondl = zcondl + (zfwatr5 * zdq + zfwatr * zdq5) * zqtmst;
      zcondl5 = zcondl5 + zfwatr5 * zdq5 * zqtmst;
      zcondi = zcondi + ((1.0 - zfwatr5) * zdq - zfwatr * zdq5) * zqtmst;
      zcondi5 = zcondi5 + (1.0 - zfwatr5) * zdq5 * zqtmst;
      zrfln = zrfln + zrn;  zrfln5 = zrfln5 + zrn5;
      zsfln = zsfln + zsn;  zsfln5 = zsfln5 + zsn5;
      zrfreeze5 = zrfreeze5 + zrfreeze25;
      zrfreeze = zrfreeze + zrfreeze2;
      /* :1043-1091 */
      TL_TENDENCIES(zdqdt, zdqdt5, zdtdt, zdtdt5)
      A(ptenq) = zdqdt;  A(ptenq5) = zdqdt5;
      A(ptent) = zdtdt;  A(ptent5) = zdtdt5;
      A(ptenl) = (zqlwc - zl[jk]) * zqtmst;   A(ptenl5) = (zqlwc5 - zl5[jk]) * zqtmst;
      A(pteni) = (zqiwc - zi[jk]) * zqtmst;   A(pteni5) = (zqiwc5 - zi5[jk]) * zqtmst;
      A1(pfplsl) = zrfln;  A1(pfplsl5) = zrfln5;
      A1(pfplsn) = zsfln;  A1(pfplsn5) = zsfln5;
      zrfl = zrfln;  zrfl5 = zrfln5;  zsfl = zsfln;  zsfl5 = zsfln5;
    }
    /* :1106-1113 */
    for (int jk = 0; jk <= klev; ++jk) {
      A(pfhpsl) = -A(pfplsl) * RLVTT;   A(pfhpsl5) = -A(pfplsl5) * RLVTT;
      A(pfhpsn) = -A(pfplsn) * RLSTT;   A(pfhpsn5) = -A(pfplsn5) * RLSTT;
    }
  }
}

#include "cloudsc2_oracle_ad.inc"
