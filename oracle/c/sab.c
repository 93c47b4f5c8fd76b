/*
 * oracle/c/sab.c -- TEST INFRASTRUCTURE ONLY (see ndpp_oracle.h).
 * CPU restatement of /root/reference/src/sab.F90: integrate_sab_el :21-109,
 * integrate_sab_inel_disc :142-245, integrate_sab_inel_cont :253-408,
 * combine_sab_grid :415-454, sab_egrid :460-568.  Arrays are flat, in the
 * Fortran element order (first index fastest).
 */
#include "ndpp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define SAB_SECONDARY_EQUAL 0
#define SAB_SECONDARY_SKEWED 1
#define SAB_SECONDARY_CONT 2
#define SAB_ELASTIC_DISCRETE 3
#define SAB_ELASTIC_EXACT 4

/* Fortran SUM as flang's runtime evaluates it (Kahan-compensated) */
static double fsum(const double *x, int n, int stride) {
  double s = 0.0, c = 0.0;
  for (int i = 0; i < n; i++) {
    double y = x[(size_t)i * stride] - c, t = s + y;
    c = (t - s) - y;
    s = t;
  }
  return s;
}

/* integrate_sab_el, sab.F90:21-109.  out [NE][G][L], zeroed here (:40). */
void oracle_sab_el(const oracle_params *p, const oracle_sab_flat *t, int NE,
                   const double *ein, int G, const double *e_bins, double *out) {
  const int L = p->order;
  memset(out, 0, sizeof(double) * (size_t)NE * G * L);
  if (t->threshold_elastic == 0.0) return;
  double wgt = 0.0;
  if (t->elastic_mode == SAB_ELASTIC_DISCRETE) wgt = 1.0 / (double)t->n_elastic_mu;
  for (int i = 0; i < NE; i++) {
    double Ein = ein[i], f, sig = 0.0;
    int isab, g;
    if (Ein < t->elastic_e_in[0]) continue;
    else if (Ein >= t->threshold_elastic) continue;
    isab = oracle_binary_search(t->elastic_e_in, t->n_elastic_e_in, Ein);
    if (isab < 0) continue;
    f = (Ein - t->elastic_e_in[isab - 1]) / (t->elastic_e_in[isab] - t->elastic_e_in[isab - 1]);
    if (Ein < e_bins[0]) continue;
    else if (Ein > e_bins[G]) continue;
    g = oracle_binary_search(e_bins, G + 1, Ein);
    if (t->elastic_mode == SAB_ELASTIC_EXACT) sig = t->elastic_P[isab - 1] / Ein;
    else if (t->elastic_mode == SAB_ELASTIC_DISCRETE)
      sig = (1.0 - f) * t->elastic_P[isab - 1] + f * t->elastic_P[isab];
    double *row = out + (size_t)i * G * L, *dg = row + (size_t)(g - 1) * L;
    if (t->n_elastic_mu == 0) {
      double mu = 1.0 - t->elastic_e_in[isab - 1] / Ein;
      for (int l = 0; l < L; l++) dg[l] = dg[l] + oracle_calc_pn(l, mu);
    } else if (t->elastic_mode == SAB_ELASTIC_DISCRETE) {
      const int NMU = t->n_elastic_mu;
      for (int imu = 0; imu < NMU; imu++) {
        double mu = (1.0 - f) * t->elastic_mu[(size_t)(isab - 1) * NMU + imu] +
                    f * t->elastic_mu[(size_t)isab * NMU + imu];
        for (int l = 0; l < L; l++) dg[l] = dg[l] + wgt * oracle_calc_pn(l, mu);
      }
    }
    for (int k = 0; k < G * L; k++) row[k] = sig * row[k];
  }
}

/* integrate_sab_inel_disc, sab.F90:142-245.  Returns -1 where the reference aborts. */
int oracle_sab_inel_disc(const oracle_params *p, const oracle_sab_flat *t, int NE,
                         const double *ein, int G, const double *e_bins, double *out) {
  const int L = p->order, NEo = t->n_inelastic_e_out, NMU = t->n_inelastic_mu,
            NEi = t->n_inelastic_e_in;
  memset(out, 0, sizeof(double) * (size_t)NE * G * L);
  double *wgt = (double *)malloc(sizeof(double) * (size_t)NEo);
  if (t->secondary_mode == SAB_SECONDARY_EQUAL) {
    for (int k = 0; k < NEo; k++) wgt[k] = 1.0 / ((double)NEo * (double)NMU);
  } else {
    if (NEo <= 4) { free(wgt); return -1; }
    wgt[0] = 0.1; wgt[1] = 0.4;
    for (int k = 2; k < NEo - 2; k++) wgt[k] = 1.0;
    wgt[NEo - 2] = 0.4; wgt[NEo - 1] = 0.1;
    double den = fsum(wgt, NEo, 1) * (double)NMU;
    for (int k = 0; k < NEo; k++) wgt[k] = wgt[k] / den;
  }
  for (int i = 0; i < NE; i++) {
    double Ein = ein[i], f;
    int isab;
    if (Ein < t->inelastic_e_in[0]) { isab = 1; f = 0.0; }
    else if (Ein > t->threshold_inelastic) continue;
    else if (Ein == t->threshold_inelastic) { isab = NEi - 1; f = 1.0; }
    else {
      isab = oracle_binary_search(t->inelastic_e_in, NEi, Ein);
      if (isab < 0) continue;
      f = (Ein - t->inelastic_e_in[isab - 1]) / (t->inelastic_e_in[isab] - t->inelastic_e_in[isab - 1]);
    }
    double sig = (1.0 - f) * t->inelastic_sigma[isab - 1] + f * t->inelastic_sigma[isab];
    double *row = out + (size_t)i * G * L;
    for (int io = 0; io < NEo; io++) {
      double Eout = (1.0 - f) * t->inelastic_e_out[(size_t)(isab - 1) * NEo + io] +
                    f * t->inelastic_e_out[(size_t)isab * NEo + io];
      if (Eout < e_bins[0]) continue;
      else if (Eout >= e_bins[G]) continue;
      int g = oracle_binary_search(e_bins, G + 1, Eout);
      double *dg = row + (size_t)(g - 1) * L;
      for (int imu = 0; imu < NMU; imu++) {
        double mu = (1.0 - f) * t->inelastic_mu[((size_t)(isab - 1) * NEo + io) * NMU + imu] +
                    f * t->inelastic_mu[((size_t)isab * NEo + io) * NMU + imu];
        for (int l = 0; l < L; l++) dg[l] = dg[l] + oracle_calc_pn(l, mu) * wgt[io];
      }
    }
    for (int k = 0; k < G * L; k++) row[k] = sig * row[k];
  }
  free(wgt);
  return 0;
}

/* integrate_sab_inel_cont, sab.F90:253-408 */
void oracle_sab_inel_cont(const oracle_params *p, const oracle_sab_flat *t, int NE,
                          const double *ein, int G, const double *e_bins, double *out) {
  const int L = p->order, NMU = t->n_inelastic_mu, NEi = t->n_inelastic_e_in;
  memset(out, 0, sizeof(double) * (size_t)NE * G * L);
  double *distro = (double *)calloc((size_t)NEi * G * L, sizeof(double));
  for (int k = 0; k < NEi; k++) {
    const int o = t->cont_ptr[k], NEout = t->cont_ptr[k + 1] - o;
    const double *Eo = t->cont_e_out + o, *pd0 = t->cont_pdf + o;
    const double *mu_arr = t->cont_mu + (size_t)o * NMU;
    double *pdf = (double *)malloc(sizeof(double) * (size_t)NEout);
    for (int j = 0; j < NEout - 1; j++) pdf[j] = pd0[j] * (Eo[j + 1] - Eo[j]);
    pdf[NEout - 1] = 0.0;
    for (int g = 0; g < G; g++) {
      double *dg = distro + ((size_t)k * G + g) * L;
      int iE_lo, iE_hi;
      if (e_bins[g] < Eo[0]) iE_lo = 1;
      else if (e_bins[g] >= Eo[NEout - 1]) { for (int l = 0; l < L; l++) dg[l] = 0.0; continue; }
      else {
        iE_lo = oracle_binary_search(Eo, NEout, e_bins[g]);
        double f_lo = (e_bins[g] - Eo[iE_lo - 1]) / (Eo[iE_lo] - Eo[iE_lo - 1]);
        double mult = f_lo * pdf[iE_lo - 1];
        for (int imu = 0; imu < NMU; imu++) {
          double mu = (1.0 - f_lo) * mu_arr[(size_t)(iE_lo - 1) * NMU + imu] +
                      f_lo * mu_arr[(size_t)iE_lo * NMU + imu];
          for (int l = 0; l < L; l++) dg[l] = dg[l] + oracle_calc_pn(l, mu) * mult;
        }
        iE_lo = iE_lo + 1;
      }
      if (e_bins[g + 1] < Eo[0]) { for (int l = 0; l < L; l++) dg[l] = 0.0; continue; }
      else if (e_bins[g + 1] >= Eo[NEout - 1]) iE_hi = NEout - 1;
      else {
        iE_hi = oracle_binary_search(Eo, NEout, e_bins[g + 1]);
        double f_hi = (e_bins[g + 1] - Eo[iE_hi - 1]) / (Eo[iE_hi] - Eo[iE_hi - 1]);
        double mult = f_hi * pdf[iE_hi - 1];
        for (int imu = 0; imu < NMU; imu++) {
          double mu = (1.0 - f_hi) * mu_arr[(size_t)(iE_hi - 1) * NMU + imu] +
                      f_hi * mu_arr[(size_t)iE_hi * NMU + imu];
          for (int l = 0; l < L; l++) dg[l] = dg[l] + oracle_calc_pn(l, mu) * mult;
        }
        iE_hi = iE_hi - 1;
      }
      for (int iE = iE_lo; iE <= iE_hi; iE++)
        for (int imu = 0; imu < NMU; imu++)
          for (int l = 0; l < L; l++)
            dg[l] = dg[l] + oracle_calc_pn(l, mu_arr[(size_t)(iE - 1) * NMU + imu]) * pdf[iE - 1];
      for (int l = 0; l < L; l++) dg[l] = dg[l] / (double)NMU;
    }
    free(pdf);
  }
  for (int i = 0; i < NE; i++) {
    double Ein = ein[i];
    double *row = out + (size_t)i * G * L;
    if (Ein <= t->inelastic_e_in[0]) {
      double sig = t->inelastic_sigma[0];
      for (int k = 0; k < G * L; k++) row[k] = distro[k] * sig;
    } else if (Ein >= t->threshold_inelastic) {
      continue;
    } else {
      int isab = oracle_binary_search(t->inelastic_e_in, NEi, Ein);
      if (isab < 0) continue;
      double f = (Ein - t->inelastic_e_in[isab - 1]) / (t->inelastic_e_in[isab] - t->inelastic_e_in[isab - 1]);
      double sig = (1.0 - f) * t->inelastic_sigma[isab - 1] + f * t->inelastic_sigma[isab];
      const double *d0 = distro + (size_t)(isab - 1) * G * L, *d1 = d0 + (size_t)G * L;
      for (int k = 0; k < G * L; k++) row[k] = ((1.0 - f) * d0[k] + f * d1[k]) * sig;
    }
  }
  free(distro);
}

/* combine_sab_grid, sab.F90:415-454 */
void oracle_sab_combine(int L, int G, int NE, const double *el, const double *inel, double *mat) {
  for (int i = 0; i < NE; i++) {
    double *m = mat + (size_t)i * G * L;
    for (int k = 0; k < G * L; k++) m[k] = el[(size_t)i * G * L + k] + inel[(size_t)i * G * L + k];
    double s = fsum(m, G, L);
    if (s > 0.0) {
      s = 1.0 / s;
      for (int k = 0; k < G * L; k++) m[k] = m[k] * s;
    } else {
      for (int k = 0; k < G * L; k++) m[k] = 0.0;
    }
  }
  if (NE >= 2) memcpy(mat + (size_t)(NE - 1) * G * L, mat + (size_t)(NE - 2) * G * L, sizeof(double) * (size_t)G * L);
}

/* calc_scattsab's Legendre path, scatt.F90:543-596.  el/inel may be NULL. */
int oracle_calc_scattsab(const oracle_params *p, const oracle_sab_flat *t, int NE,
                         const double *ein, int G, const double *e_bins, double *el,
                         double *inel, double *mat) {
  const size_t n = (size_t)NE * G * p->order;
  double *e = el ? el : (double *)malloc(sizeof(double) * n);
  double *q = inel ? inel : (double *)malloc(sizeof(double) * n);
  int rc = 0;
  oracle_sab_el(p, t, NE, ein, G, e_bins, e);
  if (t->secondary_mode == SAB_SECONDARY_CONT) oracle_sab_inel_cont(p, t, NE, ein, G, e_bins, q);
  else rc = oracle_sab_inel_disc(p, t, NE, ein, G, e_bins, q);
  oracle_sab_combine(p->order, G, NE, e, q, mat);
  if (!el) free(e);
  if (!inel) free(q);
  return rc;
}

/* sab_egrid, sab.F90:460-568.  Returns the grid length (written to out if it
 * fits cap), or -1 where a search of the reference would abort. */
int oracle_sab_egrid(const oracle_params *p, const oracle_sab_flat *t, int nb,
                     const double *e_bins, double *out, int cap) {
  const int NEi = t->n_inelastic_e_in, NEo = t->n_inelastic_e_out;
  int n, ncap = NEi + t->n_elastic_e_in + nb + 8;
  double *Ein = (double *)malloc(sizeof(double) * (size_t)ncap);
  double max_ein;
  if (t->n_elastic_e_in > 0) {
    double *tmp = (double *)malloc(sizeof(double) * (size_t)ncap);
    int nt = oracle_merge(t->inelastic_e_in, NEi, t->elastic_e_in, t->n_elastic_e_in, tmp);
    n = oracle_merge(tmp, nt, e_bins, nb, Ein);
    free(tmp);
    max_ein = fmax(t->inelastic_e_in[NEi - 1], t->elastic_e_in[t->n_elastic_e_in - 1]);
  } else {
    n = oracle_merge(t->inelastic_e_in, NEi, e_bins, nb, Ein);
    max_ein = t->inelastic_e_in[NEi - 1];
  }
  if (t->secondary_mode != SAB_SECONDARY_CONT) {
    double *pts = (double *)malloc(sizeof(double) * (size_t)nb);
    for (int i = 0; i < NEi - 1; i++) {
      double Ei1 = t->inelastic_e_in[i], Ei2 = t->inelastic_e_in[i + 1];
      for (int j = 0; j < NEo; j++) {
        int num = 0;
        double Eo1 = t->inelastic_e_out[(size_t)i * NEo + j];
        int g1 = oracle_binary_search(e_bins, nb, Eo1);
        double Eo2 = t->inelastic_e_out[(size_t)(i + 1) * NEo + j];
        int g2 = oracle_binary_search(e_bins, nb, Eo2);
        if (g1 < 0 || g2 < 0) { free(pts); free(Ein); return -1; }
        if (Eo2 < Eo1) { int g = g1; g2 = g1; g1 = g; } /* :504-508: the swap is a no-op (sic) */
        for (int g = g1 + 1; g <= g2; g++)
          pts[num++] = (e_bins[g - 1] - Eo1) / (Eo2 - Eo1) * (Ei2 - Ei1) + Ei1;
        if (num > 0) {
          double *m = (double *)malloc(sizeof(double) * (size_t)(n + num));
          int nm = oracle_merge(pts, num, Ein, n, m);
          free(Ein);
          Ein = m;
          n = nm;
        }
      }
    }
    free(pts);
  }
  int imax = oracle_binary_search(Ein, n, max_ein);
  if (imax < 0) { free(Ein); return -1; }
  int total;
  if (p->sab_epts_per_bin == 0) {
    total = imax;
    if (total <= cap) memcpy(out, Ein, sizeof(double) * (size_t)total);
  } else {
    const int EP = p->extend_pts;
    total = (imax - 1) * EP + imax;
    if (total <= cap) {
      int j = 0;
      for (int iE = 0; iE < imax - 1; iE++) {
        double dE = (log(Ein[iE + 1] / Ein[iE])) / (double)(EP + 1);
        out[j++] = Ein[iE];
        for (int k = 0; k < EP; k++) { out[j] = out[j - 1] * exp(dE); j++; }
      }
      out[total - 1] = Ein[imax - 1];
    }
  }
  free(Ein);
  return total;
}

/* apply_tol_scatt, scatt.F90:786-818: in place on data[n][G][L] */
void oracle_apply_tol_scatt(int L, int G, int n, double *data, double tol) {
  for (int i = 0; i < n; i++) {
    double *d = data + (size_t)i * G * L;
    double orig = fsum(d, G, L), norm;
    for (int g = 0; g < G; g++)
      if ((d[(size_t)g * L] > 0.0) && (d[(size_t)g * L] < tol))
        for (int l = 0; l < L; l++) d[(size_t)g * L + l] = 0.0;
    if (orig > 0.0) norm = orig / fsum(d, G, L);
    else norm = 0.0;
    for (int k = 0; k < G * L; k++) d[k] = d[k] * norm;
  }
}
