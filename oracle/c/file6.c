/*
 * oracle/c/file6.c -- TEST INFRASTRUCTURE ONLY (see ndpp_oracle.h).
 *
 * CPU restatement of the correlated energy-angle (ENDF file 6) path of
 * /root/reference/src/scattdata_header.F90: unit-base interpolation
 * (cast_to_unitbase :1554, interp_unitbase :1616, unitbase :1521),
 * integrate_file6_cm_leg :1085, integrate_file6_lab_leg :1334,
 * law9_scatter_lab_leg :1274; plus merge (array_merge.F90:13) and
 * interpolate_tab1_array (interpolation.F90:24).  Same operation order as the
 * Fortran; reference quirks are kept and marked (sic).
 */
#include "ndpp_oracle.h"

#include <float.h>
#include <math.h>
#include <omp.h>
#include <stdlib.h>
#include <string.h>

#define HISTOGRAM 1
#define LINEAR_LINEAR 2
#define LINEAR_LOG 3
#define LOG_LINEAR 4
#define LOG_LOG 5

/* array_merge.F90:13-107.  res must hold na+nb values; returns the length.
 * The count is always `ires - 1` after the loop (:97-99: either the loop ran to
 * completion, or `no_exit` was cleared before the exit).  In the branch where the
 * second operand is exhausted first (:83-88; only reachable when the first
 * operand ends in repeated values equal to the second's last) that drops the
 * value the branch has just stored (sic) -- checked against the flang build. */
int oracle_merge(const double *a, int na, const double *b, int nb, double *res) {
  const double *d1, *d2;
  int n1, n2;
  if (a[na - 1] > b[nb - 1]) { d1 = b; n1 = nb; d2 = a; n2 = na; }
  else { d1 = a; n1 = na; d2 = b; n2 = nb; }
  int nab = n1 + n2, i1 = 0, i2 = 0, n = 0;
  for (int ires = 0; ires < nab; ires++) {
    if (i1 < n1 && i2 < n2) {
      if (d1[i1] < d2[i2]) {
        res[n++] = (d1[i1] == 0.0) ? 1E-14 : d1[i1]; /* MIN_EIN, constants.F90:109 */
        i1++;
      } else if (d1[i1] == d2[i2]) {
        res[n++] = d1[i1];
        i1++; i2++;
      } else {
        res[n++] = (d2[i2] == 0.0) ? 1E-14 : d2[i2];
        i2++;
      }
    } else if (i1 < n1) {
      break;              /* :83-88 stores one value, :97-99 discards it again */
    } else if (i2 < n2) {
      res[n++] = d2[i2];
      i2++;
    } else {
      break;
    }
  }
  return n;
}

/* interpolation.F90:24-123 (loc_start absent) */
double oracle_interpolate_tab1(const double *data, double x) {
  int n_regions = (int)data[0];
  int loc_breakpoints = 1, loc_interp = loc_breakpoints + n_regions;
  int n_points = (int)data[loc_interp + n_regions];
  int loc_x = loc_interp + n_regions + 1, loc_y = loc_x + n_points;
  /* 1-based data(loc_x + k) == data[loc_x + k - 1] */
  if (x < data[loc_x]) return data[loc_y];
  else if (x > data[loc_x + n_points - 1]) return data[loc_y + n_points - 1];
  int i = oracle_binary_search(data + loc_x, n_points, x);
  int interp = LINEAR_LINEAR;
  if (n_regions == 0) interp = LINEAR_LINEAR;
  else if (n_regions == 1) interp = (int)data[loc_interp];
  else
    for (int j = 1; j <= n_regions; j++)
      if (i < data[loc_breakpoints + j - 1]) { interp = (int)data[loc_interp + j - 1]; break; }
  if (interp == HISTOGRAM) return data[loc_y + i - 1];
  double x0 = data[loc_x + i - 1], x1 = data[loc_x + i];
  double y0 = data[loc_y + i - 1], y1 = data[loc_y + i];
  double r;
  switch (interp) {
  case LINEAR_LINEAR: r = (x - x0) / (x1 - x0); return (1 - r) * y0 + r * y1;
  case LINEAR_LOG: r = (log(x) - log(x0)) / (log(x1) - log(x0)); return (1 - r) * y0 + r * y1;
  case LOG_LINEAR: r = (x - x0) / (x1 - x0); return exp((1 - r) * log(y0) + r * log(y1));
  case LOG_LOG: r = (log(x) - log(x0)) / (log(x1) - log(x0)); return exp((1 - r) * log(y0) + r * log(y1));
  default: return NAN; /* reference: fatal_error */
  }
}

/* cast_to_unitbase, scattdata_header.F90:1554-1609 (np >= 2). ub holds np values. */
int oracle_cast_to_unitbase(const double *Eout, int np, double *ub) {
  double inv_dE = Eout[np - 1] - Eout[0];
  if ((inv_dE >= 0.0) && (inv_dE < DBL_MAX)) inv_dE = 1.0 / inv_dE;
  else inv_dE = 0.0;
  for (int i = 0; i < np - 1; i++) ub[i] = (Eout[i] - Eout[0]) * inv_dE;
  ub[np - 1] = 1.0;
  if (ub[np - 2] == 1.0) return np - 1;
  return np;
}

/* unitbase + interp_unitbase, scattdata_header.F90:1521-1717.
 * f1 is [np1][M] (column j = fEmu1(:, j+1)), likewise f2; fEmu out is [nub][M].
 * Returns nub (<= np1+np2), or -1 where the reference would abort in a search. */
int oracle_unitbase(double Ein, int M, int np1, const double *eout1, const double *pdf1,
                    int intt1, const double *f1, double Ei1, int np2, const double *eout2,
                    const double *pdf2, int intt2, const double *f2, double Ei2,
                    double *Eout, double *pdf, int *INTT, double *fEmu) {
  (void)intt2; /* the reference tests INTT1 where INTT2 is meant, :1685-1697 (sic) */
  double *ub1 = (double *)malloc(sizeof(double) * (size_t)(np1 + np2) * 2);
  double *ub2 = ub1 + np1;
  double *ub = ub2 + np2;
  int n1 = oracle_cast_to_unitbase(eout1, np1, ub1);
  int n2 = oracle_cast_to_unitbase(eout2, np2, ub2);
  /* ub needs n1+n2 slots: reuse a fresh buffer */
  double *ubm = (double *)malloc(sizeof(double) * (size_t)(n1 + n2));
  (void)ub;
  int nub = oracle_merge(ub1, n1, ub2, n2, ubm);
  double f = (Ein - Ei1) / (Ei2 - Ei1);
  double dE1 = (eout1[np1 - 1] - eout1[0]);
  double dE2 = (eout2[np2 - 1] - eout2[0]);
  int bad = 0;
  for (int i = 0; i < nub; i++) {
    double u = ubm[i], r = 0.0, p1 = 0.0, p2 = 0.0;
    int j = oracle_binary_search(ub1, n1, u);
    if (j < 0) { bad = 1; break; }
    if (intt1 == HISTOGRAM) r = 0.0;
    else if (intt1 == LINEAR_LINEAR || intt1 == LOG_LINEAR) r = (u - ub1[j - 1]) / (ub1[j] - ub1[j - 1]);
    else if (intt1 == LINEAR_LOG || intt1 == LOG_LOG) r = log(u / ub1[j - 1]) / log(ub1[j] / ub1[j - 1]);
    if (intt1 == HISTOGRAM || intt1 == LINEAR_LINEAR || intt1 == LINEAR_LOG)
      p1 = (1.0 - r) * pdf1[j - 1] + r * pdf1[j];
    else if (intt1 == LOG_LINEAR || intt1 == LOG_LOG)
      p1 = exp((1.0 - r) * log(pdf1[j - 1]) + r * log(pdf1[j]));
    double *fo = fEmu + (size_t)i * M;
    const double *a = f1 + (size_t)(j - 1) * M, *b = f1 + (size_t)j * M;
    for (int k = 0; k < M; k++) fo[k] = (1.0 - f) * ((1.0 - r) * a[k] + r * b[k]);

    j = oracle_binary_search(ub2, n2, u);
    if (j < 0) { bad = 1; break; }
    if (intt1 == HISTOGRAM) r = 0.0;
    else if (intt1 == LINEAR_LINEAR || intt1 == LOG_LINEAR) r = (u - ub2[j - 1]) / (ub2[j] - ub2[j - 1]);
    else if (intt1 == LINEAR_LOG || intt1 == LOG_LOG) r = log(u / ub2[j - 1]) / log(ub2[j] / ub2[j - 1]);
    if (intt1 == HISTOGRAM || intt1 == LINEAR_LINEAR || intt1 == LINEAR_LOG)
      p2 = (1.0 - r) * pdf2[j - 1] + r * pdf2[j];
    else if (intt1 == LOG_LINEAR || intt1 == LOG_LOG)
      p2 = exp((1.0 - r) * log(pdf2[j - 1]) + r * log(pdf2[j]));
    a = f2 + (size_t)(j - 1) * M;
    b = f2 + (size_t)j * M;
    for (int k = 0; k < M; k++) fo[k] = fo[k] + f * ((1.0 - r) * a[k] + r * b[k]);

    pdf[i] = (1.0 - f) * p1 + f * p2;
    Eout[i] = (1.0 - f) * (eout1[0] + dE1 * u) + f * (eout2[0] + dE2 * u);
  }
  *INTT = LINEAR_LINEAR;
  free(ub1);
  free(ubm);
  return bad ? -1 : nub;
}

/* Fortran SUM as flang's runtime does it for REAL(8): Kahan-compensated.
 * (Only used where the reference calls sum(); pinned in test_oracle_vs_ref.) */
static double kahan_sum(const double *x, int n, int stride) {
  double s = 0.0, c = 0.0;
  for (int i = 0; i < n; i++) {
    double y = x[(size_t)i * stride] - c;
    double t = s + y;
    c = (t - s) - y;
    s = t;
  }
  return s;
}

/* integrate_file6_cm_leg, scattdata_header.F90:1085-1266.  fEmu [np][M];
 * distro [G][L] pre-zeroed by the caller (:529). */
void oracle_integrate_file6_cm_leg(const oracle_params *p, const double *fEmu, int np,
                                   const double *mu, double Ein, double awr,
                                   const double *Eout, int INTT, const double *thispdf,
                                   const double *E_bins, int nbins, double *distro) {
  const int L = p->order, M = p->mu_bins, NEG = p->ne_per_grp;
  double deltamu = mu[1] - mu[0];
  double *pdf = (double *)malloc(sizeof(double) * (size_t)np);
  memcpy(pdf, thispdf, sizeof(double) * (size_t)np);
  if (Eout[np - 1] == Eout[np - 2]) pdf[np - 2] = 0.0;
  double *fEl = (double *)malloc(sizeof(double) * (size_t)(L * 2 + 2 * M + nbins + 2));
  double *tmp = fEl + L, *fmu = tmp + L, *mu_l = fmu + M, *E_bnds = mu_l + M;
  double ap1inv = 1.0 / (awr + 1.0);
  double Eo_lo = Eout[0] + (Ein - 2.0 * (awr + 1.0) * sqrt(Ein * Eout[0])) * ap1inv * ap1inv;
  Eo_lo = 1E-12; /* :1141 overwrites it (sic) */
  double Eo_hi = Eout[np - 1] + (Ein + 2.0 * (awr + 1.0) * sqrt(Ein * Eout[np - 1])) * ap1inv * ap1inv;
  int g_lo, g_hi; /* 1-based groups */
  if (Eo_lo <= E_bins[0]) g_lo = 1;
  else if (Eo_lo >= E_bins[nbins - 1]) goto done;
  else g_lo = oracle_binary_search(E_bins, nbins, Eo_lo);
  /* E_bnds(g) stored at E_bnds[g] (1-based use) */
  if (Eo_hi <= E_bins[0]) goto done;
  else if (Eo_hi >= E_bins[nbins - 1]) {
    g_hi = nbins - 1;
    E_bnds[g_lo] = Eo_lo;
    for (int g = g_lo + 1; g <= g_hi; g++) E_bnds[g] = E_bins[g - 1];
    E_bnds[g_hi + 1] = E_bins[g_hi - 1]; /* E_bins(g_hi), :1159 (sic) */
  } else {
    g_hi = oracle_binary_search(E_bins, nbins, Eo_hi);
    E_bnds[g_lo] = Eo_lo;
    for (int g = g_lo + 1; g <= g_hi; g++) E_bnds[g] = E_bins[g - 1];
    E_bnds[g_hi + 1] = Eo_hi;
  }
  for (int g = g_lo; g <= g_hi; g++) {
    double *dg = distro + (size_t)(g - 1) * L;
    double Eo = E_bnds[g];
    double dEo = (E_bnds[g + 1] - E_bnds[g]) / (double)(NEG - 1);
    Eo = Eo - dEo;
    for (int iE = 1; iE <= NEG; iE++) {
      Eo = Eo + dEo;
      for (int l = 0; l < L; l++) fEl[l] = 0.0;
      for (int k = 0; k < M; k++) fmu[k] = 0.0;
      double c = ap1inv * sqrt(Ein / Eo);
      double mu_l_min = (1.0 + c * c - Eout[np - 1] / Eo) / (2.0 * c);
      if (mu_l_min < -1.0) mu_l_min = -1.0;
      else if (fabs(mu_l_min - 1.0) < 1E-10) mu_l_min = 1.0;
      else if (mu_l_min > 1.0) continue;
      double dmu = (1.0 - mu_l_min) / (double)(M - 1);
      for (int imu = 1; imu <= M; imu++) {
        mu_l[imu - 1] = mu_l_min + dmu * (double)(imu - 1);
        double Eo_cm = Eo * (1.0 + c * c - 2.0 * c * mu_l[imu - 1]);
        int iEo;
        if (Eo_cm <= 0.0) continue;
        else if (Eo_cm <= Eout[0]) iEo = 1;
        else if (Eo_cm >= Eout[np - 1]) iEo = np - 1;
        else iEo = oracle_binary_search(Eout, np, Eo_cm);
        double fEo, pEo;
        if (INTT == HISTOGRAM) {
          fEo = 0.0;
          pEo = pdf[iEo - 1];
        } else if (Eout[iEo] == Eout[iEo - 1]) {
          fEo = 0.0;
          pEo = pdf[iEo - 1];
        } else {
          fEo = (Eo_cm - Eout[iEo - 1]) / (Eout[iEo] - Eout[iEo - 1]);
          pEo = (1.0 - fEo) * pdf[iEo - 1] + fEo * pdf[iEo];
        }
        double J = sqrt(Eo / Eo_cm), mu_c;
        if (mu_l[imu - 1] == -1.0) mu_c = -1.0;
        else if (mu_l[imu - 1] == 1.0) mu_c = 1.0;
        else {
          mu_c = (mu_l[imu - 1] - c) * J;
          if (fabs(mu_c) > 1.0) continue;
        }
        int imu_c;
        double f;
        if (fabs(mu_c - 1.0) < 1E-10) {
          imu_c = M - 1;
          f = 1.0;
        } else {
          imu_c = (int)((mu_c + 1.0) / deltamu) + 1;
          f = (mu_c - mu[imu_c - 1]) / (mu[imu_c] - mu[imu_c - 1]);
        }
        const double *c0 = fEmu + (size_t)(iEo - 1) * M, *c1 = fEmu + (size_t)iEo * M;
        double proby = (1.0 - fEo) * ((1.0 - f) * c0[imu_c - 1] + f * c0[imu_c]);
        proby = proby + fEo * ((1.0 - f) * c1[imu_c - 1] + f * c1[imu_c]);
        fmu[imu - 1] = proby * J * pEo;
      }
      for (int imu = 1; imu <= M - 1; imu++) {
        oracle_calc_int_pn_tablelin(L, mu_l[imu - 1], mu_l[imu], fmu[imu - 1], fmu[imu], tmp);
        for (int l = 0; l < L; l++) fEl[l] = fEl[l] + tmp[l];
      }
      if ((iE != 1) && (iE != NEG))
        for (int l = 0; l < L; l++) dg[l] = dg[l] + 2.0 * fEl[l];
      else
        for (int l = 0; l < L; l++) dg[l] = dg[l] + fEl[l];
    }
    for (int l = 0; l < L; l++) dg[l] = dg[l] * dEo * 0.5;
  }
  {
    double fEo = 0.0;
    for (int g = g_lo; g <= g_hi; g++) fEo = fEo + distro[(size_t)(g - 1) * L];
    if (fEo > 0.0) fEo = 1.0 / fEo;
    for (int g = g_lo; g <= g_hi; g++)
      for (int l = 0; l < L; l++) distro[(size_t)(g - 1) * L + l] *= fEo;
  }
done:
  free(pdf);
  free(fEl);
}

/* integrate_file6_lab_leg, scattdata_header.F90:1334-1450 */
void oracle_integrate_file6_lab_leg(const oracle_params *p, const double *fEmu, int np,
                                    const double *mu, const double *Eout, int INTT,
                                    const double *thispdf, const double *E_bins, int nbins,
                                    double *distro) {
  (void)INTT;
  const int L = p->order, M = p->mu_bins, G = nbins - 1;
  double *pdf = (double *)malloc(sizeof(double) * (size_t)(np + M + L));
  double *fint = pdf + np, *tmp = fint + M;
  memcpy(pdf, thispdf, sizeof(double) * (size_t)np);
  for (int i = 0; i < np - 1; i++) pdf[i] = thispdf[i] * (Eout[i + 1] - Eout[i]);
  if (np > 1 && Eout[np - 1] == Eout[np - 2]) pdf[np - 2] = 0.0;
  if (np > 1) {
    for (int g = 0; g < G; g++) {
      double *dg = distro + (size_t)g * L;
      for (int k = 0; k < M; k++) fint[k] = 0.0;
      int iE_lo, iE_hi; /* 1-based */
      if (E_bins[g] < Eout[0]) {
        iE_lo = 1;
      } else if (E_bins[g] >= Eout[np - 1]) {
        for (int l = 0; l < L; l++) dg[l] = 0.0;
        continue;
      } else {
        iE_lo = oracle_binary_search(Eout, np, E_bins[g]);
        double f_lo = (E_bins[g] - Eout[iE_lo - 1]) / (Eout[iE_lo] - Eout[iE_lo - 1]);
        const double *col = fEmu + (size_t)(iE_lo - 1) * M;
        for (int k = 0; k < M; k++) fint[k] = fint[k] + f_lo * pdf[iE_lo - 1] * col[k];
        iE_lo = iE_lo + 1;
      }
      if (E_bins[g + 1] < Eout[0]) {
        for (int l = 0; l < L; l++) dg[l] = 0.0;
        continue;
      } else if (E_bins[g + 1] >= Eout[np - 1]) {
        iE_hi = np - 1;
      } else {
        iE_hi = oracle_binary_search(Eout, np, E_bins[g + 1]);
        double f_hi = (E_bins[g + 1] - Eout[iE_hi - 1]) / (Eout[iE_hi] - Eout[iE_hi - 1]);
        const double *col = fEmu + (size_t)(iE_hi - 1) * M;
        for (int k = 0; k < M; k++) fint[k] = fint[k] + f_hi * pdf[iE_hi - 1] * col[k];
        iE_hi = iE_hi - 1;
      }
      for (int iE = iE_lo; iE <= iE_hi; iE++) {
        const double *col = fEmu + (size_t)(iE - 1) * M;
        for (int k = 0; k < M; k++) fint[k] = fint[k] + pdf[iE - 1] * col[k];
      }
      for (int imu = 1; imu <= M - 1; imu++) {
        oracle_calc_int_pn_tablelin(L, mu[imu - 1], mu[imu], fint[imu - 1], fint[imu], tmp);
        for (int l = 0; l < L; l++) dg[l] = dg[l] + tmp[l];
      }
    }
  } else {
    for (int g = 0; g < G; g++) {
      double *dg = distro + (size_t)g * L;
      if ((Eout[0] > E_bins[g]) && (Eout[0] <= E_bins[g + 1])) {
        for (int imu = 1; imu <= M - 1; imu++) {
          oracle_calc_int_pn_tablelin(L, mu[imu - 1], mu[imu], fEmu[imu - 1], fEmu[imu], tmp);
          for (int l = 0; l < L; l++) dg[l] = dg[l] + tmp[l];
        }
      } else {
        for (int l = 0; l < L; l++) dg[l] = 0.0;
      }
    }
  }
  double f_lo = 1.0 / kahan_sum(distro, G, L); /* ONE / sum(distro(1,:)), :1447 */
  for (int k = 0; k < G * L; k++) distro[k] = distro[k] * f_lo;
  free(pdf);
}

/* law9_scatter_lab_leg, scattdata_header.F90:1274-1326; edata = edist%data */
void oracle_law9_scatter_lab_leg(const oracle_params *p, const double *fmu,
                                 const double *edata, double Ein, const double *E_bins,
                                 int nbins, const double *mu, double *distro) {
  const int L = p->order, M = p->mu_bins;
  double tmp[16];
  int NR = (int)edata[0];
  int NE = (int)edata[1 + 2 * NR];
  double T = oracle_interpolate_tab1(edata, Ein);
  int lc = 2 + 2 * NR + 2 * NE;
  double U = edata[lc];
  double x = (Ein - U) / T;
  double I = T * T * (1.0 - exp(-x) * (1.0 + x));
  if (Ein - U <= 0.0) return;
  for (int g = 0; g < nbins - 1; g++) {
    double *dg = distro + (size_t)g * L;
    double Egp1 = E_bins[g + 1], Eg = E_bins[g];
    if (Egp1 > (Ein - U)) Egp1 = Ein - U;
    if (Eg > (Ein - U)) Eg = Ein - U;
    double pE = (exp(-Egp1 / T) * (T + Egp1)) - (exp(-Eg / T) * (T + Eg));
    pE = -T * pE / I;
    for (int imu = 1; imu <= M - 1; imu++) {
      oracle_calc_int_pn_tablelin(L, mu[imu - 1], mu[imu], fmu[imu - 1], fmu[imu], tmp);
      for (int l = 0; l < L; l++) dg[l] = dg[l] + tmp[l] * pE;
    }
  }
}

/* The edist branches of integrate_distro (scattdata_header.F90:593-656) for
 * n_ein points of one ScattData given as CSR tables:
 *   e_grid[n_rows], row_ptr[n_rows+1] (offsets into eout/pdf and, times M, f),
 *   intt[n_rows], f[(row_ptr[k]+j)*M + imu].
 * frame_cm = 1: unitbase + integrate_file6_cm_leg; 0: unitbase +
 * integrate_file6_lab_leg.  row_lo[i] = iE-1 of scatt_interp_distro.
 * out [n_ein][G][L].  Returns 0, or -1 if a search left its table. */
int oracle_file6_leg_batch(const oracle_params *p, double awr, int frame_cm, int n_ein,
                           const double *ein, const int *row_lo, int n_rows,
                           const double *e_grid, const int *row_ptr, const double *eout,
                           const double *pdf, const int *intt, const double *f, int G,
                           const double *e_bins, double *out, int nthreads) {
  const int L = p->order, M = p->mu_bins;
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  double *gmu = (double *)malloc(sizeof(double) * (size_t)M);
  oracle_mu_grid(M, gmu);
  int bad = 0;
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads) reduction(| : bad)
  for (int i = 0; i < n_ein; i++) {
    int k = row_lo[i];
    if (k < 0 || k + 1 >= n_rows) { bad |= 1; continue; }
    int np1 = row_ptr[k + 1] - row_ptr[k], np2 = row_ptr[k + 2] - row_ptr[k + 1];
    double *Eo = (double *)malloc(sizeof(double) * (size_t)(np1 + np2) * (2 + (size_t)M));
    double *pd = Eo + (np1 + np2), *fE = pd + (np1 + np2);
    int INTT;
    int nub = oracle_unitbase(ein[i], M, np1, eout + row_ptr[k], pdf + row_ptr[k], intt[k],
                              f + (size_t)row_ptr[k] * M, e_grid[k], np2, eout + row_ptr[k + 1],
                              pdf + row_ptr[k + 1], intt[k + 1], f + (size_t)row_ptr[k + 1] * M,
                              e_grid[k + 1], Eo, pd, &INTT, fE);
    double *o = out + (size_t)i * G * L;
    memset(o, 0, sizeof(double) * (size_t)G * L);
    if (nub < 0) bad |= 1;
    else if (frame_cm)
      oracle_integrate_file6_cm_leg(p, fE, nub, gmu, ein[i], awr, Eo, INTT, pd, e_bins, G + 1, o);
    else
      oracle_integrate_file6_lab_leg(p, fE, nub, gmu, Eo, INTT, pd, e_bins, G + 1, o);
    free(Eo);
  }
  free(gmu);
  return bad ? -1 : 0;
}

/* law 9 with adist rows: integrate_distro :605-638 -- both rows + blend */
int oracle_law9_leg_batch(const oracle_params *p, int n_ein, const double *ein,
                          const int *row_lo, const double *w_hi, int n_rows,
                          const double *f_tab, const double *edata, int G,
                          const double *e_bins, double *out, int nthreads) {
  const int L = p->order, M = p->mu_bins;
  if (nthreads <= 0) nthreads = omp_get_max_threads();
  double *gmu = (double *)malloc(sizeof(double) * (size_t)M);
  oracle_mu_grid(M, gmu);
#pragma omp parallel for schedule(dynamic, 1) num_threads(nthreads)
  for (int i = 0; i < n_ein; i++) {
    double lo[16 * 128], hi[16 * 128];
    (void)n_rows;
    memset(lo, 0, sizeof(double) * (size_t)G * L);
    memset(hi, 0, sizeof(double) * (size_t)G * L);
    const double *f0 = f_tab + (size_t)row_lo[i] * M;
    oracle_law9_scatter_lab_leg(p, f0, edata, ein[i], e_bins, G + 1, gmu, lo);
    oracle_law9_scatter_lab_leg(p, f0 + M, edata, ein[i], e_bins, G + 1, gmu, hi);
    double f = w_hi[i];
    double *o = out + (size_t)i * G * L;
    for (int k = 0; k < G * L; k++) {
      double r = (1.0 - f) * lo[k];
      o[k] = r + f * hi[k];
    }
  }
  free(gmu);
  return 0;
}
