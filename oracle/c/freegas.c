/*
 * oracle/c/freegas.c -- TEST INFRASTRUCTURE ONLY (see ndpp_oracle.h).
 *
 * CPU restatement of /root/reference/src/freegas.F90 (free-gas Doppler elastic
 * kernel -> Legendre moments), calc_pn of legendre.F90, binary_search of
 * search.F90 and integrate_file4_cm_leg / tolab of scattdata_header.F90.
 * Operation order follows the Fortran expressions exactly (left to right, same
 * parenthesisation) so that with IEEE arithmetic and glibc exp/sqrt the results
 * are bit-identical to the flang -O0 -ffp-contract=off build of the reference.
 */
#include "ndpp_oracle.h"

#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

/* constants.F90:35 -- truncated on purpose, part of the contract */
#define NDPP_PI 3.1415926535898

static _Thread_local unsigned long long tl_nfgk; /* calc_fgk evaluation counter */

void oracle_default_params(oracle_params *p) {
  /* constants.F90:70-100 */
  p->order = 6;
  p->mu_bins = 2001;
  p->sab_threshold = 1.0E-6;
  p->brent_mu_thresh = 1.0E-6;
  p->adaptive_mu_tol = 1.0E-7;
  p->adaptive_eout_tol = 1.0E-8;
  p->adaptive_mu_its = 15;
  p->adaptive_eout_its = 15;
  p->ne_per_grp = 20;
  p->sab_epts_per_bin = 10;
  p->extend_pts = 50;
  p->inel_extend_pts = 30;
}

/* x**n for integer n as flang/LLVM lower it (llvm.powi -> compiler-rt
 * __powidf2: square-and-multiply from the low bit).  Pinned against
 * ref_calc_pn in tests/test_oracle_vs_ref.py. */
static inline double powi(double a, int b) {
  double r = 1.0;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return r;
}

/* legendre.F90:349-432 */
double oracle_calc_pn(int n, double x) {
  switch (n) {
  case 0: return 1.0;
  case 1: return x;
  case 2: return 1.5 * x * x - 0.5;
  case 3: return 2.5 * x * x * x - 1.5 * x;
  case 4: return 4.375 * powi(x, 4) - 3.75 * x * x + 0.375;
  case 5: return 7.875 * powi(x, 5) - 8.75 * x * x * x + 1.875 * x;
  case 6:
    return 14.4375 * powi(x, 6) - 19.6875 * powi(x, 4) + 6.5625 * x * x - 0.3125;
  case 7:
    return 26.8125 * powi(x, 7) - 43.3125 * powi(x, 5) + 19.6875 * x * x * x -
           2.1875 * x;
  case 8:
    return 50.2734375 * powi(x, 8) - 93.84375 * powi(x, 6) +
           54.140625 * powi(x, 4) - 9.84375 * x * x + 0.2734375;
  case 9:
    return 94.9609375 * powi(x, 9) - 201.09375 * powi(x, 7) +
           140.765625 * powi(x, 5) - 36.09375 * x * x * x + 2.4609375 * x;
  case 10:
    return 180.42578125 * powi(x, 10) - 427.32421875 * powi(x, 8) +
           351.9140625 * powi(x, 6) - 117.3046875 * powi(x, 4) +
           13.53515625 * x * x - 0.24609375;
  default: return 1.0; /* legendre.F90:428-429; n > 10 is outside ndpp.xml's
                          scatt_order <= 10 limit (ndpp.F90:290-301) */
  }
}

/* search.F90:21-71 */
int oracle_binary_search(const double *a, int n, double v) {
  int L = 1, R = n, it = 0, idx;
  if (v < a[L - 1] || v > a[R - 1]) return -1; /* reference: fatal_error */
  while (R - L > 1) {
    if (v > a[L - 1] && v < a[L]) return L;
    else if (v > a[R - 2] && v < a[R - 1]) return R - 1;
    idx = L + (R - L) / 2;
    double t = a[idx - 1];
    if (v >= t) L = idx;
    else if (v < t) R = idx;
    if (++it == 64) return -1; /* reference: fatal_error */
  }
  return L;
}

/* scattdata_header.F90:251-257 */
void oracle_mu_grid(int M, double *mu) {
  double dmu = 2.0 / (double)(M - 1);
  for (int i = 1; i <= M - 1; i++) mu[i - 1] = -1.0 + (double)(i - 1) * dmu;
  mu[M - 1] = 1.0;
}

/* freegas.F90:154-181 */
void oracle_calc_fg_eout_bounds(double A, double kT, double Ein, double *lo,
                                double *hi) {
  double alpha = (A - 1.0) / (A + 1.0);
  alpha = alpha * alpha;
  *lo = 0.001 * alpha * Ein;
  if (Ein > 300.0 * kT / A)
    *hi = 12.0 * kT * (A + 1.0) / A + 1.5 * Ein;
  else
    *hi = 12.0 * kT * (A + 1.0) / A + 2.0 * Ein;
}

/* freegas.F90:188-228 */
double oracle_calc_sab(double A, double kT, double Ein, double Eout,
                       double beta, double mu) {
  const double alpha_min = 1.0E-6, sab_min = -225.0, lterm_min = 2.0E-10;
  double r = (A + 1.0) / A;
  double lterm = sqrt(Eout / Ein) / kT * (r * r);
  double alpha = (Ein + Eout - 2.0 * mu * sqrt(Ein * Eout)) / (A * kT);
  if (alpha < alpha_min) alpha = alpha_min;
  double t = alpha + beta;
  double sab = -(t * t) / (4.0 * alpha);
  if (sab < sab_min) {
    sab = 0.0;
  } else {
    sab = lterm * exp(sab) / (sqrt(4.0 * NDPP_PI * alpha));
    if (sab < lterm_min) sab = 0.0;
  }
  return sab;
}

/* freegas.F90:235-345 */
double oracle_brent_mu(const oracle_params *p, double awr, double kT,
                       double Ein, double Eout, double beta, double thresh,
                       double lo, double hi) {
  const double TH = p->brent_mu_thresh;
  double a = lo, b = hi, c = 0.0, d = DBL_MAX; /* INFINITY = huge(0d0) */
  double fa = oracle_calc_sab(awr, kT, Ein, Eout, beta, a) - thresh;
  double fb = oracle_calc_sab(awr, kT, Ein, Eout, beta, b) - thresh;
  double fc = 0.0, s = 0.0, fs = 0.0, tmp;
  int mflag;

  if (fa * fb >= 0.0) return (fa < fb) ? a : b;

  if (fabs(fa) < fabs(fb)) {
    tmp = a; a = b; b = tmp;
    tmp = fa; fa = fb; fb = tmp;
  }
  c = a;
  fc = fa;
  mflag = 1;

  while ((fb != 0.0) && (fabs(a - b) > TH)) {
    if ((fa != fc) && (fb != fc)) {
      s = a * fb * fc / (fa - fb) / (fa - fc) +
          b * fa * fc / (fb - fa) / (fb - fc) +
          c * fa * fb / (fc - fa) / (fc - fb);
    } else {
      s = b - fb * (b - a) / (fb - fa);
    }
    tmp = (3.0 * a + b) * 0.25;
    if ((!(((s > tmp) && (s < b)) || ((s < tmp) && (s > b)))) ||
        (mflag && (fabs(s - b) >= (0.5 * fabs(b - c)))) ||
        (!mflag && (fabs(s - b) >= (fabs(c - d) * 0.5)))) {
      s = 0.5 * (a + b);
      mflag = 1;
    } else {
      if ((mflag && (fabs(b - c) < TH)) || (!mflag && (fabs(c - d) < TH))) {
        s = (a + b) * 0.5;
        mflag = 1;
      } else {
        mflag = 0;
      }
    }
    fs = oracle_calc_sab(awr, kT, Ein, Eout, beta, s) - thresh;
    d = c;
    c = b;
    fc = fb;
    if (fa * fs < 0.0) {
      b = s;
      fb = fs;
    } else {
      a = s;
      fa = fs;
    }
    if (fabs(fa) < fabs(fb)) {
      tmp = a; a = b; b = tmp;
      tmp = fa; fa = fb; fb = tmp;
    }
  }
  return b;
}

/* freegas.F90:356-409 */
void oracle_find_fg_mu(const oracle_params *p, double A, double kT, double Ein,
                       double Eout, double mu2[2]) {
  double beta = (Eout - Ein) / kT;
  double alpha_max = sqrt(beta * beta + 1.0) - 1.0;
  double mu_max = (Ein + Eout - alpha_max * A * kT) / (2.0 * sqrt(Ein * Eout));
  double mu_lo, mu_hi;
  if (fabs(mu_max) > 1.0) {
    mu_lo = -1.0;
    mu_hi = 1.0;
  } else {
    double sab_max = oracle_calc_sab(A, kT, Ein, Eout, beta, mu_max);
    double thr = sab_max * p->sab_threshold;
    if (oracle_calc_sab(A, kT, Ein, Eout, beta, -1.0) > thr)
      mu_lo = -1.0;
    else
      mu_lo = oracle_brent_mu(p, A, kT, Ein, Eout, beta, thr, -1.0, mu_max);
    if (oracle_calc_sab(A, kT, Ein, Eout, beta, 1.0) > thr)
      mu_hi = 1.0;
    else
      mu_hi = oracle_brent_mu(p, A, kT, Ein, Eout, beta, thr, mu_max, 1.0);
  }
  mu2[0] = mu_lo;
  mu2[1] = mu_hi;
}

/* freegas.F90:415-473 */
double oracle_calc_fgk(double awr, double kT, double Ein, double Eout, int l,
                       double mu, const double *fEmu, const double *gmu,
                       int M) {
  tl_nfgk++;
  double dmu = gmu[1] - gmu[0];
  int i;
  if (mu <= gmu[0])
    i = 1;
  else if (mu >= gmu[M - 1])
    i = M - 1;
  else
    i = (int)((mu + 1.0) / dmu) + 1;
  double interp = (mu - gmu[i - 1]) / (gmu[i] - gmu[i - 1]);
  double fval = (1.0 - interp) * fEmu[i - 1] + interp * fEmu[i];
  double r = (awr + 1.0) / awr;
  double lterm = fval * sqrt(Eout / Ein) / kT * (r * r);
  double alpha = (Ein + Eout - 2.0 * mu * sqrt(Ein * Eout)) / (awr * kT);
  double beta = (Eout - Ein) / kT;
  if (alpha < 1.0E-6) alpha = 1.0E-6;
  double t = alpha + beta;
  double fgk = -(t * t) / (4.0 * alpha);
  if (fgk <= -708.0)
    fgk = 0.0;
  else
    fgk = lterm * exp(fgk) / (sqrt(4.0 * NDPP_PI * alpha)) * oracle_calc_pn(l, mu);
  return fgk;
}

typedef struct {
  const oracle_params *p;
  double awr, kT, Ein;
  int l, M;
  const double *fEmu, *gmu;
} fg_ctx;

/* freegas.F90:511-553 */
static double asimp_aux_mu(const fg_ctx *c, double Eout, double a, double b,
                           double eps, double S, double fa, double fb,
                           double fc, int bottom) {
  double cc = 0.5 * (a + b);
  double h = b - a;
  double d = 0.5 * (a + cc);
  double e = 0.5 * (cc + b);
  double fd = oracle_calc_fgk(c->awr, c->kT, c->Ein, Eout, c->l, d, c->fEmu, c->gmu, c->M);
  double fe = oracle_calc_fgk(c->awr, c->kT, c->Ein, Eout, c->l, e, c->fEmu, c->gmu, c->M);
  double Sleft = (h / 12.0) * (fa + 4.0 * fd + fc);
  double Sright = (h / 12.0) * (fc + 4.0 * fe + fb);
  double S2 = Sleft + Sright;
  if ((bottom <= 0) || (fabs(S2 - S) <= 15.0 * eps)) {
    return S2 + (S2 - S) / 15.0;
  } else {
    double left = asimp_aux_mu(c, Eout, a, cc, 0.5 * eps, Sleft, fa, fc, fd, bottom - 1);
    double right = asimp_aux_mu(c, Eout, cc, b, 0.5 * eps, Sright, fc, fb, fe, bottom - 1);
    return left + right;
  }
}

/* freegas.F90:482-509 */
static double asimp_mu(const fg_ctx *c, double Eout, double a, double b) {
  double cc = (a + b) * 0.5;
  double h = (b - a);
  double fa = oracle_calc_fgk(c->awr, c->kT, c->Ein, Eout, c->l, a, c->fEmu, c->gmu, c->M);
  double fb = oracle_calc_fgk(c->awr, c->kT, c->Ein, Eout, c->l, b, c->fEmu, c->gmu, c->M);
  double fc = oracle_calc_fgk(c->awr, c->kT, c->Ein, Eout, c->l, cc, c->fEmu, c->gmu, c->M);
  double S = (h / 6.0) * (fa + 4.0 * fc + fb);
  return asimp_aux_mu(c, Eout, a, b, c->p->adaptive_mu_tol, S, fa, fb, fc,
                      c->p->adaptive_mu_its);
}

static double mu_integral_at(const fg_ctx *c, double Eout) {
  double m[2];
  oracle_find_fg_mu(c->p, c->awr, c->kT, c->Ein, Eout, m);
  return asimp_mu(c, Eout, m[0], m[1]);
}

/* freegas.F90:598-644 */
static double asimp_aux_eout(const fg_ctx *c, double a, double b, double eps,
                             double S, double fa, double fb, double fc,
                             int bottom) {
  double cc = 0.5 * (a + b);
  double d = 0.5 * (a + cc);
  double e = 0.5 * (cc + b);
  double h = b - a;
  double fd = mu_integral_at(c, d);
  double fe = mu_integral_at(c, e);
  double Sleft = (h / 12.0) * (fa + 4.0 * fd + fc);
  double Sright = (h / 12.0) * (fc + 4.0 * fe + fb);
  double S2 = Sleft + Sright;
  if ((bottom <= 0) || (fabs(S2 - S) <= 15.0 * eps)) {
    return S2 + (S2 - S) / 15.0;
  } else {
    double left = asimp_aux_eout(c, a, cc, 0.5 * eps, Sleft, fa, fc, fd, bottom - 1);
    double right = asimp_aux_eout(c, cc, b, 0.5 * eps, Sright, fc, fb, fe, bottom - 1);
    return left + right;
  }
}

/* freegas.F90:563-596 */
static double asimp_eout(const fg_ctx *c, double a, double b) {
  double cc = 0.5 * (a + b);
  double h = b - a;
  double fa = mu_integral_at(c, a);
  double fb = mu_integral_at(c, b);
  double fc = mu_integral_at(c, cc);
  double S = (h / 6.0) * (fa + 4.0 * fc + fb);
  return asimp_aux_eout(c, a, b, c->p->adaptive_eout_tol, S, fa, fb, fc,
                        c->p->adaptive_eout_its);
}

double oracle_adaptive_simpsons_mu(const oracle_params *p, double A, double kT,
                                   double Ein, double Eout, int l,
                                   const double *fEmu, const double *gmu, int M,
                                   double a, double b) {
  fg_ctx c = {p, A, kT, Ein, l, M, fEmu, gmu};
  return asimp_mu(&c, Eout, a, b);
}

double oracle_adaptive_simpsons_eout(const oracle_params *p, double A,
                                     double kT, double Ein, int l,
                                     const double *fEmu, const double *gmu,
                                     int M, double a, double b) {
  fg_ctx c = {p, A, kT, Ein, l, M, fEmu, gmu};
  return asimp_eout(&c, a, b);
}

/* freegas.F90:18-146.  distro[g*L + l] == Fortran distro(l+1, g+1) */
void oracle_integrate_freegas_leg(const oracle_params *p, double Ein, double A,
                                  double kT, const double *fEmu,
                                  const double *gmu, const double *E_bins,
                                  int nbins, double *distro) {
  const int L = p->order, G = nbins - 1;
  fg_ctx c = {p, A, kT, Ein, 0, p->mu_bins, fEmu, gmu};
  double alphaEin = (A - 1.0) / (A + 1.0);
  alphaEin = alphaEin * alphaEin * Ein;
  double p0 = 0.0, Eout_lo, Eout_hi, Elo, Ehi, Ebottom;
  oracle_calc_fg_eout_bounds(A, kT, Ein, &Eout_lo, &Eout_hi);

  for (int g = 0; g < G; g++) {
    double *dg = distro + (size_t)g * L;
    if ((E_bins[g] < Eout_hi) && (E_bins[g + 1] > Eout_lo)) {
      Elo = (Eout_lo > E_bins[g]) ? Eout_lo : E_bins[g];
      Ehi = (Eout_hi < E_bins[g + 1]) ? Eout_hi : E_bins[g + 1];
      Ebottom = (E_bins[g] == 0.0) ? 0.01 * Elo : E_bins[g];
      for (int l = 0; l < L; l++) {
        c.l = l;
        double t1 = asimp_eout(&c, Ebottom, Elo);
        double t2 = asimp_eout(&c, Ehi, E_bins[g + 1]);
        dg[l] = t1 + t2;
      }
      if ((Elo < alphaEin) && (alphaEin < Ehi)) {
        for (int l = 0; l < L; l++) {
          c.l = l;
          dg[l] = dg[l] + asimp_eout(&c, Elo, alphaEin);
        }
        Elo = alphaEin;
      }
      if ((Elo < Ein) && (Ein < Ehi)) {
        for (int l = 0; l < L; l++) {
          c.l = l;
          dg[l] = dg[l] + asimp_eout(&c, Elo, Ein);
        }
        Elo = Ein;
      }
      for (int l = 0; l < L; l++) {
        c.l = l;
        dg[l] = dg[l] + asimp_eout(&c, Elo, Ehi);
      }
    } else {
      /* freegas.F90:118-131 -- Ebottom is computed there but unused */
      for (int l = 0; l < L; l++) {
        c.l = l;
        dg[l] = asimp_eout(&c, E_bins[g], E_bins[g + 1]);
      }
    }
    p0 = p0 + dg[0];
    for (int l = 0; l < L; l++)
      if (fabs(dg[l]) < 1.0E-18) dg[l] = 0.0;
  }
  for (int i = 0; i < L * G; i++) distro[i] = distro[i] / p0;
}

/* scattdata_header.F90:1466-1496 */
double oracle_tolab(double R, double w) {
  double u;
  if (R > 1.0) {
    u = (1.0 + R * w) / sqrt(1.0 + R * R + 2.0 * R * w);
  } else if (R == 1.0) {
    if (w == -1.0)
      u = -1.0;
    else
      u = (1.0 + R * w) / sqrt(1.0 + R * R + 2.0 * R * w);
  } else {
    if (w < -R) {
      u = sqrt(1.0 - R * R);
      double f = (w - (-1.0)) / (-R - 1.0);
      u = (1.0 - f) * (-1.0) + f * u;
    } else {
      u = (1.0 + R * w) / sqrt(1.0 + R * R + 2.0 * R * w);
    }
  }
  return u;
}

/* scattdata_header.F90:956-1078; distro[g*L+l], pre-zeroed by the caller */
void oracle_integrate_file4_cm_leg(const oracle_params *p, const double *fw,
                                   double Ein, double awr, double Q,
                                   const double *E_bins, int nbins,
                                   const double *w, double *distro) {
  const int L = p->order, G = nbins - 1, M = p->mu_bins;
  double dw = w[1] - w[0];
  double R = awr * sqrt((1.0 + Q * (awr + 1.0) / (awr * Ein)));
  double onepawr2 = (1.0 + awr) * (1.0 + awr);
  double onepR2 = 1.0 + R * R;
  double inv2REin = 0.5 / (R * Ein);

  for (int g = 0; g < G; g++) {
    double *dg = distro + (size_t)g * L;
    double wlo = (E_bins[g] * onepawr2 - Ein * onepR2) * inv2REin;
    if (wlo < -1.0) wlo = -1.0;
    else if (wlo > 1.0) wlo = 1.0;
    int ilo = (int)((wlo + 1.0) / dw) + 1;
    double whi = (E_bins[g + 1] * onepawr2 - Ein * onepR2) * inv2REin;
    if (whi < -1.0) whi = -1.0;
    else if (whi > 1.0) whi = 1.0;
    int ihi = (int)((whi + 1.0) / dw) + 1;

    if (wlo == whi) {
      if (wlo == -1.0) continue;
      else if (wlo == 1.0) return;
    }
    double flo, fhi, interp;
    if (ilo == M) {
      flo = fw[M - 1];
    } else {
      interp = (wlo - w[ilo - 1]) / (w[ilo] - w[ilo - 1]);
      flo = (1.0 - interp) * fw[ilo - 1] + interp * fw[ilo];
    }
    if (ihi == M) {
      fhi = fw[M - 1];
    } else {
      interp = (whi - w[ihi - 1]) / (w[ihi] - w[ihi - 1]);
      fhi = (1.0 - interp) * fw[ihi - 1] + interp * fw[ihi];
    }
    double ulo, uhi;
    if (ilo != ihi) {
      ulo = oracle_tolab(R, wlo);
      uhi = oracle_tolab(R, w[ilo]);
      for (int l = 0; l < L; l++)
        dg[l] = (w[ilo] - wlo) * (flo * oracle_calc_pn(l, ulo) +
                                  fw[ilo] * oracle_calc_pn(l, uhi));
      for (int iw = ilo + 1; iw <= ihi - 1; iw++) {
        ulo = uhi;
        uhi = oracle_tolab(R, w[iw]);
        for (int l = 0; l < L; l++)
          dg[l] = dg[l] + (w[iw] - w[iw - 1]) *
                              (fw[iw - 1] * oracle_calc_pn(l, ulo) +
                               fw[iw] * oracle_calc_pn(l, uhi));
      }
      ulo = uhi;
      uhi = oracle_tolab(R, whi);
      for (int l = 0; l < L; l++)
        dg[l] = dg[l] + (whi - w[ihi - 1]) *
                            (fw[ihi - 1] * oracle_calc_pn(l, ulo) +
                             fhi * oracle_calc_pn(l, uhi));
    } else {
      ulo = oracle_tolab(R, wlo);
      uhi = oracle_tolab(R, whi);
      for (int l = 0; l < L; l++)
        dg[l] = (whi - wlo) * (flo * oracle_calc_pn(l, ulo) +
                               fhi * oracle_calc_pn(l, uhi));
    }
    for (int l = 0; l < L; l++) dg[l] = 0.5 * dg[l];
  }
}

/* integrate_distro, adist-only branch: scattdata_header.F90:533-591 */
int oracle_elastic_leg_batch(const oracle_params *p, double A, double kT,
                             double freegas_cutoff, double Q, int n_ein,
                             const double *ein, const int *row_lo,
                             const double *w_hi, int n_rows,
                             const double *f_tab, int G, const double *e_bins,
                             double *out, int nthreads,
                             unsigned long long *counters) {
  const int L = p->order, M = p->mu_bins;
  if (L < 1 || L > 11 || M < 2 || G < 1 || n_ein < 0) return -22;
  for (int i = 0; i < n_ein; i++)
    if (row_lo[i] < 0 || row_lo[i] + 1 >= n_rows) return -22;
  double *gmu = (double *)malloc(sizeof(double) * (size_t)M);
  oracle_mu_grid(M, gmu);
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  unsigned long long total = 0;
#pragma omp parallel num_threads(nthreads) reduction(+ : total)
  {
    double *lo = (double *)malloc(sizeof(double) * (size_t)L * G);
    double *hi = (double *)malloc(sizeof(double) * (size_t)L * G);
    tl_nfgk = 0;
#pragma omp for schedule(dynamic, 1)
    for (int i = 0; i < n_ein; i++) {
      const double *f0 = f_tab + (size_t)row_lo[i] * M;
      const double *f1 = f0 + M;
      double f = w_hi[i];
      double *o = out + (size_t)i * L * G;
      memset(lo, 0, sizeof(double) * (size_t)L * G);
      memset(hi, 0, sizeof(double) * (size_t)L * G);
      if (ein[i] < freegas_cutoff) {
        oracle_integrate_freegas_leg(p, ein[i], A, kT, f0, gmu, e_bins, G + 1, lo);
        oracle_integrate_freegas_leg(p, ein[i], A, kT, f1, gmu, e_bins, G + 1, hi);
      } else {
        oracle_integrate_file4_cm_leg(p, f0, ein[i], A, Q, e_bins, G + 1, gmu, lo);
        oracle_integrate_file4_cm_leg(p, f1, ein[i], A, Q, e_bins, G + 1, gmu, hi);
      }
      for (int k = 0; k < L * G; k++) {
        double r = lo[k] * (1.0 - f);
        o[k] = r + hi[k] * f;
      }
    }
    total += tl_nfgk;
    free(lo);
    free(hi);
  }
  if (counters) counters[0] += total;
  free(gmu);
  return 0;
}
