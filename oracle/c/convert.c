/*
 * convert.c -- TEST INFRASTRUCTURE ONLY (see ndpp_oracle.h).
 *
 * Restatement of the ACE -> tabular conversion of a ScattData
 * (scattdata_header.F90): convert_file4 (:669-760) and convert_file6 (:769-950),
 * i.e. what scatt_convert_distro (:325-382) fills into distro(iE)%data, Eouts,
 * pdfs, cdfs and INTT.  Pinned by the reference's own known-answer tests
 * (tests/test_scatt/test_scattdata.F90:485-820, :827-1182) in
 * tests/test_convert.py, and against the flang build in the same file.
 *
 * Index convention: `data` is the Fortran array with 1-based indices, so the
 * code below addresses it through D(i) = data[i-1] and keeps the reference's
 * index arithmetic verbatim (including reads one element before a table when
 * mu lies below its first abscissa, as the reference does).
 */
#include <math.h>
#include <string.h>

#include "ndpp_oracle.h"

#define FP_PRECISION 1e-14 /* constants.F90:22 */
#define NUM_EP 32          /* constants.F90:116 */
#define R_NUM_EP (1.0 / 32.0)
#define D(i) data[(i)-1]

/* the tabular branches shared by convert_file4 (:711-751) and convert_file6
 * (:866-944): data(lc .. lc+NP-1) abscissae, pdf NP further on.  The Fortran
 * scans forward from the previous match; because mu increases, that equals a
 * scan from lc for every mu (an index rejected for a smaller mu is rejected
 * for a larger one).  allow_log: convert_file6 knows interpolation 3..5. */
static void tabular_row(const double *data, int lc, int NP, int interp, int allow_log,
                        const double *mu, int M, double *out) {
  for (int imu = 0; imu < M; ++imu) {
    const double m = mu[imu];
    for (int idata = lc; idata <= lc + NP - 1; ++idata) {
      if ((D(idata) - m) > FP_PRECISION) {
        if (interp == 1) {
          out[imu] = D(idata - 1 + NP);
        } else if (interp == 2) {
          const double r = (m - D(idata - 1)) / (D(idata) - D(idata - 1));
          out[imu] = D(idata + NP - 1) + r * (D(idata + NP) - D(idata + NP - 1));
        } else if (allow_log && interp == 3) {
          const double r = (log(m) - log(D(idata - 1))) / (log(D(idata)) - log(D(idata - 1)));
          out[imu] = D(idata + NP - 1) + r * (D(idata + NP) - D(idata - 1 + NP));
        } else if (allow_log && interp == 4) {
          const double r = (m - D(idata - 1)) / (D(idata) - D(idata - 1));
          /* (sic) the weights are swapped relative to a log-lin interpolation, :913 */
          out[imu] = exp((1.0 - r) * log(D(idata + NP)) + r * log(D(idata + NP - 1)));
        } else if (allow_log && interp == 5) {
          const double r = (log(m) - log(D(idata - 1))) / (log(D(idata)) - log(D(idata - 1)));
          out[imu] = exp((1.0 - r) * log(D(idata + NP)) + r * log(D(idata + NP - 1)));
        }
        break;
      } else if (fabs(D(idata) - m) <= FP_PRECISION) {
        out[imu] = D(idata + NP);
        break;
      }
    }
  }
}

/* convert_file4 for one incoming energy: out[M] must be pre-zeroed (:348).
 * type / lc are adist%type(iE) / adist%location(iE).  Unknown types and unknown
 * interpolation codes leave out untouched (the reference's "graceful failure"). */
void oracle_convert_file4_row(int type, int lc, const double *data, const double *mu, int M,
                              double *out) {
  if (type == 1) { /* ANGLE_ISOTROPIC */
    for (int i = 0; i < M; ++i) out[i] = 0.5;
  } else if (type == 2) { /* ANGLE_32_EQUI, :693-710 */
    for (int imu = 0; imu < M; ++imu) {
      for (int idata = lc + 1; idata <= lc + 1 + NUM_EP; ++idata) {
        if (D(idata) >= mu[imu]) {
          if (imu == 0) out[imu] = R_NUM_EP / (D(idata + 1) - D(idata));
          else out[imu] = R_NUM_EP / (D(idata) - D(idata - 1));
          break;
        }
      }
    }
  } else if (type == 3) { /* ANGLE_TABULAR */
    const int interp = (int)D(lc + 1);
    const int NP = (int)D(lc + 2);
    if (interp == 1 || interp == 2) tabular_row(data, lc + 3, NP, interp, 0, mu, M, out);
  }
}

void oracle_convert_file4(int M, int n_rows, const int *type, const int *location,
                          const double *data, double *f_tab) {
  double mu[M];
  oracle_mu_grid(M, mu);
  memset(f_tab, 0, sizeof(double) * (size_t)n_rows * M);
  for (int k = 0; k < n_rows; ++k)
    oracle_convert_file4_row(type[k], location[k], data, mu, M, f_tab + (size_t)k * M);
}

/* number of incoming energies and of outgoing energies of row iE (1-based) */
int oracle_file6_ne(const double *data) {
  const int NR = (int)D(1);
  return (int)D(2 + 2 * NR);
}
int oracle_file6_np(const double *data, int iE) {
  const int NR = (int)D(1);
  const int NE = (int)D(2 + 2 * NR);
  const int lc = (int)D(2 + 2 * NR + NE + iE);
  return (int)D(lc + 2);
}

/* convert_file6 for incoming energy iE (1-based).  Returns 0 on success, 1 for
 * an unsupported law (nothing touched, :796-797), -1 where the reference stops
 * (NR > 0, :803-806; unknown angular interpolation, :945).  distro is [NP][M]
 * (== Fortran distro(M, NP)); for law 4 it is left as the caller filled it. */
int oracle_convert_file6_row(int law, const double *data, int iE, const double *mu, int M,
                             double *eouts, double *pdf, double *cdf, int *INTT, double *distro) {
  if (law != 4 && law != 44 && law != 61) return 1;
  const int NR = (int)D(1);
  if (NR > 0) return -1;
  const int NE = (int)D(2 + 2 * NR);
  int lc = (int)D(2 + 2 * NR + NE + iE);
  int intt = (int)D(lc + 1);
  if (intt > 10) intt = intt % 10;
  *INTT = intt;
  const int NP = (int)D(lc + 2);
  for (int k = 1; k <= NP; ++k) {
    eouts[k - 1] = D(lc + 2 + k);
    pdf[k - 1] = D(lc + 2 + NP + k);
    cdf[k - 1] = D(lc + 2 + 2 * NP + k);
  }
  if (law == 44) {
    lc = lc + 2;
    for (int iEout = 1; iEout <= NP; ++iEout) {
      const double KMR = D(lc + 3 * NP + iEout);
      const double KMA = D(lc + 4 * NP + iEout);
      const double KMconst = 0.5 * KMA / sinh(KMA);
      double *col = distro + (size_t)(iEout - 1) * M;
      for (int i = 0; i < M; ++i) col[i] = KMconst * (cosh(KMA * mu[i]) + KMR * sinh(KMA * mu[i]));
    }
  } else if (law == 61) {
    const int lcin = lc + 2;
    for (int iEout = 1; iEout <= NP; ++iEout) {
      double *col = distro + (size_t)(iEout - 1) * M;
      lc = (int)D(lcin + 3 * NP + iEout);
      if (lc == 0) {
        for (int i = 0; i < M; ++i) col[i] = 0.5;
        continue;
      }
      const int interp = (int)D(lc + 1);
      const int NPang = (int)D(lc + 2);
      if (interp < 1 || interp > 5) return -1;
      tabular_row(data, lc + 3, NPang, interp, 1, mu, M, col);
    }
  }
  return 0;
}

/* scatt_convert_distro for a law 4 / 44 / 61 ScattData (:350-375): all NE incoming
 * energies into the CSR tables ndpp_file6_leg_batch takes.  f is [sum NP][M].
 * Law 4 (:350-370) takes the angular distribution from adist (n_adist energies):
 * the row at the bracketing adist energy is copied into the outgoing-energy
 * columns 1..2 ONLY -- the copy loop runs over size(Eouts), which at that point
 * is still the 2-element placeholder convert_file4 allocated (sic, :364-366) --
 * and the remaining columns stay zero. */
int oracle_convert_file6(int M, int law, const double *data, int n_adist,
                         const double *adist_energy, const int *adist_type,
                         const int *adist_location, const double *adist_data, double *e_grid,
                         int *row_ptr, double *eout, double *pdf, double *cdf, int *intt,
                         double *f) {
  double mu[M];
  oracle_mu_grid(M, mu);
  const int NR = (int)D(1);
  if (NR > 0) return -1;
  const int NE = (int)D(2 + 2 * NR);
  row_ptr[0] = 0;
  for (int iE = 1; iE <= NE; ++iE) {
    e_grid[iE - 1] = D(2 + 2 * NR + iE);
    const int np = oracle_file6_np(data, iE);
    row_ptr[iE] = row_ptr[iE - 1] + np;
    const size_t o = (size_t)row_ptr[iE - 1];
    memset(f + o * M, 0, sizeof(double) * (size_t)np * M);
    if (law == 4) {
      int iEa;
      if (e_grid[iE - 1] <= adist_energy[0]) iEa = 1;
      else if (e_grid[iE - 1] >= adist_energy[n_adist - 1]) iEa = n_adist;
      else iEa = oracle_binary_search(adist_energy, n_adist, e_grid[iE - 1]);
      oracle_convert_file4_row(adist_type[iEa - 1], adist_location[iEa - 1], adist_data, mu, M,
                               f + o * M);
      if (np >= 2) memcpy(f + (o + 1) * M, f + o * M, sizeof(double) * M);
    }
    const int rc = oracle_convert_file6_row(law, data, iE, mu, M, eout + o, pdf + o, cdf + o,
                                            intt + iE - 1, f + o * M);
    if (rc) return rc;
  }
  return 0;
}
