/*
 * oracle/c/chi.c -- TEST INFRASTRUCTURE ONLY (see ndpp_oracle.h).
 * CPU restatement of the fission-spectrum (chi) group integration:
 * calc_chi (chi.F90:21-169), ChiData%beta/prob/integrate
 * (chidata_header.F90:139-493), nu_total / nu_delayed (fission.F90:18-103).
 * Reference quirks are kept and marked (sic).
 */
#include "ndpp_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

#define NDPP_PI 3.1415926535898 /* constants.F90:35 */
#define NU_NONE 0
#define NU_POLYNOMIAL 1
#define NU_TABULAR 2

static double ipow(double a, int b) {
  double r = 1.0;
  if (b == 0) return 1.0;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return r;
}

static double fsum1(const double *x, int n) { /* flang SUM: Kahan */
  double s = 0.0, c = 0.0;
  for (int i = 0; i < n; i++) {
    double y = x[i] - c, t = s + y;
    c = (t - s) - y;
    s = t;
  }
  return s;
}

/* interpolate_tab1_object, interpolation.F90:132-208 */
static double tab1_obj(int n_regions, const int *nbt, const int *intp, int n_pairs,
                       const double *x, const double *y, double v) {
  if (v < x[0]) return y[0];
  else if (v > x[n_pairs - 1]) return y[n_pairs - 1];
  int i = oracle_binary_search(x, n_pairs, v), interp = 2;
  if (n_regions == 1) interp = intp[0];
  else if (n_regions > 1)
    for (int j = 0; j < n_regions; j++)
      if (i < nbt[j]) { interp = intp[j]; break; }
  if (interp == 1) return y[i - 1];
  double x0 = x[i - 1], x1 = x[i], y0 = y[i - 1], y1 = y[i], r;
  switch (interp) {
  case 2: r = (v - x0) / (x1 - x0); return (1 - r) * y0 + r * y1;
  case 3: r = (log(v) - log(x0)) / (log(x1) - log(x0)); return (1 - r) * y0 + r * y1;
  case 4: r = (v - x0) / (x1 - x0); return exp((1 - r) * log(y0) + r * log(y1));
  case 5: r = (log(v) - log(x0)) / (log(x1) - log(x0)); return exp((1 - r) * log(y0) + r * log(y1));
  default: return NAN;
  }
}

/* fission.F90:18-45 */
static double nu_total(const oracle_chi_nuclide *n, double E) {
  if (n->nu_t_type == NU_POLYNOMIAL) {
    int NC = (int)n->nu_t_data[0];
    double nu = 0.0;
    for (int i = 0; i <= NC - 1; i++) nu = nu + n->nu_t_data[i + 1] * ipow(E, i);
    return nu;
  } else if (n->nu_t_type == NU_TABULAR) {
    return oracle_interpolate_tab1(n->nu_t_data, E);
  }
  return NAN; /* reference: fatal_error */
}

/* fission.F90:90-103 */
static double nu_delayed(const oracle_chi_nuclide *n, double E) {
  if (n->nu_d_type == NU_TABULAR) return oracle_interpolate_tab1(n->nu_d_data, E);
  return 0.0;
}

/* chi_prob, chidata_header.F90:154-215 */
static double chi_prob(const oracle_chi_nuclide *n, const oracle_chi_spectrum *s, int delayed,
                       int grp, double Ein) {
  if (delayed) {
    const double *pd = n->nu_d_precursor_data; /* 1-based: pd[k-1] */
    int lc = 1, NR = 0, NE = 0;
    for (int j = 1; j <= n->n_precursor; j++) {
      NR = (int)pd[lc];
      NE = (int)pd[lc + 1 + 2 * NR];
      if (j == grp) break;
      lc = lc + 2 + 2 * NR + 2 * NE + 1;
    }
    (void)NE;
    return oracle_interpolate_tab1(pd + lc, Ein);
  }
  int j;
  double f, prob;
  if (Ein < n->energy[0]) { j = 1; f = 0.0; }
  else if (Ein >= n->energy[n->n_grid - 1]) { j = n->n_grid - 1; f = 1.0; }
  else {
    j = oracle_binary_search(n->energy, n->n_grid, Ein);
    f = (Ein - n->energy[j - 1]) / (n->energy[j] - n->energy[j - 1]);
  }
  if (n->energy[j - 1] == n->energy[j]) j = j + 1;
  if (j < s->threshold) prob = 0.0;
  else
    prob = ((1.0 - f) * s->sigma[j - s->threshold] + f * s->sigma[j - s->threshold + 1]) /
           ((1.0 - f) * n->fission[j - 1] + f * n->fission[j]);
  if (s->has_next && s->pv_n_regions > 0) /* only when n_regions > 0 (sic), :210 */
    prob = prob * tab1_obj(s->pv_n_regions, s->pv_nbt, s->pv_int, s->pv_n_pairs, s->pv_x, s->pv_y, Ein);
  return prob;
}

/* chi_integrate, chidata_header.F90:221-493.  chis[G] */
static void chi_integrate(const oracle_chi_spectrum *s, double Ein, int G, const double *E_bins,
                          double *chis) {
  const double *d = s->data; /* data(k) == d[k-1] */
  for (int g = 0; g < G; g++) chis[g] = 0.0;
  int NR = (int)d[0], NE = (int)d[1 + 2 * NR], lc;
  double T, U, I, x, x0;
  switch (s->law) {
  case 4:
  case 61: {
    int hist = 0;
    if (NR == 1 && s->law == 4) hist = (d[2] == 1);
    lc = 2 + 2 * NR;
    int iE;
    if (Ein < d[lc]) { iE = 1; x = 0.0; }
    else if (Ein >= d[lc + NE - 1]) { iE = NE - 1; x = 1.0; }
    else {
      iE = oracle_binary_search(d + lc, NE, Ein);
      x = (Ein - d[lc + iE - 1]) / (d[lc + iE] - d[lc + iE - 1]);
    }
    if (!hist && x > 0.5) iE = iE + 1; /* nearest row, not interpolation, :294-298 */
    lc = (int)d[2 + 2 * NR + NE + iE - 1];
    int NP = (int)d[lc + 1];
    lc = lc + 3;
    int lEout_min = lc;
    double runsum = 0.0;
    for (int g = 1; g <= G; g++) {
      int k;
      for (k = lEout_min; k <= NP + lc - 2; k++)
        if (d[k] > E_bins[g]) break; /* data(iE+1) > E_bins(g+1) */
      if (k == NP + lc - 1) k = k - 1;
      double interp = (E_bins[g] - d[k - 1]) / (d[k] - d[k - 1]);
      double v = (d[k + 2 * NP - 1] + interp * (d[k + 2 * NP] - d[k + 2 * NP - 1]));
      v = v - runsum;
      runsum = runsum + v;
      chis[g - 1] = v;
      lEout_min = k;
    }
    break;
  }
  case 7:
    T = oracle_interpolate_tab1(d, Ein);
    lc = 2 + 2 * NR + 2 * NE;
    U = d[lc];
    if (Ein - U <= 0.0) return;
    x = (Ein - U) / T;
    I = sqrt(T * T * T) * (sqrt(0.25 * NDPP_PI) * erf(x) - x * exp(-x));
    for (int g = 0; g < G; g++) {
      double Egp1 = E_bins[g + 1];
      if (Egp1 > Ein - U) Egp1 = U; /* clamps to U, not Ein-U (sic), :375 */
      double v = 0.5 * (sqrt(NDPP_PI * T) * erf(sqrt(Egp1 / T)) * exp(Egp1 / T) - 2.0 * sqrt(Egp1)) *
                 T * exp(-Egp1 / T);
      double Eg = E_bins[g];
      if (Eg > Ein - U) Eg = U;
      v = v - (0.5 * (sqrt(NDPP_PI * T) * erf(sqrt(Eg / T)) * exp(Eg / T) - 2.0 * sqrt(Eg)) * T *
               exp(-Eg / T));
      chis[g] = v / I;
    }
    break;
  case 9:
    T = oracle_interpolate_tab1(d, Ein);
    lc = 2 + 2 * NR + 2 * NE;
    U = d[lc];
    x = (Ein - U) / T;
    if (Ein - U <= 0.0) return;
    for (int g = 0; g < G; g++) {
      double Egp1 = E_bins[g + 1], Eg = E_bins[g];
      if (Egp1 > (Ein - U)) Egp1 = Ein - U;
      if (Eg > (Ein - U)) Eg = Ein - U;
      double v = (Egp1 * exp(x) + T * exp(x)) * exp(-Egp1 / T);
      v = v - (Eg * exp(x) + T * exp(x)) * exp(-Eg / T);
      chis[g] = v / (T * (x - exp(x) + 1.0));
    }
    break;
  case 11: {
    double Wa = oracle_interpolate_tab1(d, Ein);
    lc = 2 + 2 * (NR + NE);
    double Wb = oracle_interpolate_tab1(d + lc, Ein);
    NR = (int)d[lc];
    NE = (int)d[lc + 1 + 2 * NR];
    lc = lc + 2 + 2 * (NR + NE);
    U = d[lc];
    x = (Ein - U) / Wa;
    if (Ein - U <= 0.0) return;
    x0 = Wa * Wb * 0.25;
    I = 0.25 * sqrt(NDPP_PI * ipow(Wa, 3) * Wb) * exp(x0) *
            (erf(sqrt(x) - sqrt(x0)) + erf(sqrt(x) + sqrt(x0))) -
        Wa * exp(-x * sinh(Wa * Wb * x));
    Wb = sqrt(Wb);
    x = sqrt(NDPP_PI * Wa) * Wb * exp(0.25 * Wa * (Wb * Wb));
    for (int g = 0; g < G; g++) {
      double Egp1 = E_bins[g + 1];
      if (Egp1 > U) Egp1 = U;
      double v = (-x * erf((Wa * Wb - 2.0 * sqrt(Egp1) / (2.0 * Wa))) +
                  x * erf((Wa * Wb + 2.0 * sqrt(Egp1) / (2.0 * Wa))) -
                  2.0 * (exp(2.0 * Wb * sqrt(Egp1)) * exp(-(Wa * Wb * sqrt(Egp1)) / Wa)));
      double Eg = E_bins[g];
      if (Eg > U) Eg = U;
      v = v - (-x * erf((Wa * Wb - 2.0 * sqrt(Eg) / (2.0 * Wa))) +
               x * erf((Wa * Wb + 2.0 * sqrt(Eg) / (2.0 * Wa))) -
               2.0 * (exp(2.0 * Wb * sqrt(Eg)) * exp(-(Wa * Wb * sqrt(Eg)) / Wa)));
      chis[g] = 0.25 * Wa * v / I;
    }
    break;
  }
  default: /* other laws: warning only, chis stays zero, :240-252,:352,:464-481 */
    break;
  }
  I = 0.0;
  for (int g = 0; g < G; g++) I = I + chis[g];
  if (I != 1.0) {
    I = 1.0 / I; /* all-zero spectrum -> 1/0 -> NaN (sic), :482-491 */
    for (int g = 0; g < G; g++) chis[g] = chis[g] * I;
  }
}

/* E grid of one spectrum: chi_init, chidata_header.F90:98-104 */
static int spectrum_grid(const oracle_chi_spectrum *s, const double **e) {
  int NR = (int)s->data[0];
  *e = s->data + 2 + 2 * NR;
  return (int)s->data[1 + 2 * NR];
}

/* union incoming grid, chi.F90:97-113.  Returns the length (out if it fits). */
int oracle_chi_egrid(int n_prompt, const oracle_chi_spectrum *prompt, int n_delay,
                     const oracle_chi_spectrum *delay, double *out, int cap) {
  const double *e;
  int n = spectrum_grid(&prompt[0], &e), tot = n;
  for (int i = 1; i < n_prompt; i++) { const double *q; tot += spectrum_grid(&prompt[i], &q); }
  for (int i = 0; i < n_delay; i++) { const double *q; tot += spectrum_grid(&delay[i], &q); }
  double *a = (double *)malloc(sizeof(double) * (size_t)tot), *b = (double *)malloc(sizeof(double) * (size_t)tot);
  memcpy(a, e, sizeof(double) * (size_t)n);
  for (int i = 1; i < n_prompt + n_delay; i++) {
    const oracle_chi_spectrum *s = (i < n_prompt) ? &prompt[i] : &delay[i - n_prompt];
    const double *q;
    int m = spectrum_grid(s, &q);
    n = oracle_merge(a, n, q, m, b);
    double *t = a; a = b; b = t;
  }
  if (n <= cap) memcpy(out, a, sizeof(double) * (size_t)n);
  free(a);
  free(b);
  return n;
}

/* calc_chi's loop over incoming energies, chi.F90:124-159.
 * chi_t, chi_p [NE][G]; chi_d [n_delay][NE][G] */
void oracle_calc_chi(const oracle_chi_nuclide *n, int n_prompt, const oracle_chi_spectrum *prompt,
                     int n_delay, const oracle_chi_spectrum *delay, int G, const double *E_bins,
                     int NE, const double *E_grid, double *chi_t, double *chi_p, double *chi_d) {
  double *cp = (double *)malloc(sizeof(double) * (size_t)G);
  memset(chi_t, 0, sizeof(double) * (size_t)NE * G);
  memset(chi_p, 0, sizeof(double) * (size_t)NE * G);
  for (int iE = 0; iE < NE; iE++) {
    double Ein = E_grid[iE], prob = 0.0, norm;
    double *ct = chi_t + (size_t)iE * G, *cpm = chi_p + (size_t)iE * G;
    double beta = nu_delayed(n, Ein) / nu_total(n, Ein);
    for (int i = 0; i < n_prompt; i++) {
      chi_integrate(&prompt[i], Ein, G, E_bins, cp);
      prob = chi_prob(n, &prompt[i], 0, 0, Ein);
      for (int g = 0; g < G; g++) ct[g] = ct[g] + prob * (1.0 - beta) * cp[g];
      for (int g = 0; g < G; g++) cpm[g] = cpm[g] + prob * cp[g];
    }
    /* overwrites the (1-beta)-weighted sum using the LAST prob (sic), chi.F90:135 */
    for (int g = 0; g < G; g++) ct[g] = cpm[g] + prob * cpm[g];
    for (int i = 0; i < n_delay; i++) {
      double *cd = chi_d + ((size_t)i * NE + iE) * G;
      chi_integrate(&delay[i], Ein, G, E_bins, cd);
      prob = chi_prob(n, &delay[i], 1, i + 1, Ein);
      for (int g = 0; g < G; g++) ct[g] = ct[g] + prob * beta * cd[g];
    }
    norm = fsum1(ct, G);
    if (norm > 0.0) for (int g = 0; g < G; g++) ct[g] = ct[g] / norm;
    norm = fsum1(cpm, G);
    if (norm > 0.0) for (int g = 0; g < G; g++) cpm[g] = cpm[g] / norm;
    for (int i = 0; i < n_delay; i++) {
      double *cd = chi_d + ((size_t)i * NE + iE) * G;
      norm = fsum1(cd, G);
      if (norm > 0.0) for (int g = 0; g < G; g++) cd[g] = cd[g] / norm;
    }
  }
  free(cp);
}
