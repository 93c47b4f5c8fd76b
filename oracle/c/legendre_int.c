/*
 * oracle/c/legendre_int.c -- TEST INFRASTRUCTURE ONLY (see ndpp_oracle.h).
 * calc_int_pn_tablelin, legendre.F90:22-336, orders 0..10 (scatt_order <= 10,
 * ndpp.F90:290-301).  The closed forms live in tablelin_forms.inc, beside this file
 * (same operation order as the Fortran).  The product does not use them: it derives the
 * integrals from Legendre identities (ndpp_amd/csrc/legendre_int.h), and this file is what
 * that derivation is checked against.
 */
#include "ndpp_oracle.h"

#include "tablelin_forms.inc"

static inline double ipow(double a, int b) { /* llvm.powi lowering of x**n */
  double r = 1.0;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return r;
}
#define P(x, n) ipow((x), (n))

void oracle_calc_int_pn_tablelin(int n, double xlow, double xhigh, double flow,
                                 double fhigh, double *integrals) {
  for (int l = 0; l < n; l++) integrals[l] = 0.0;
  if (xhigh - xlow < 1e-14) return; /* FP_PRECISION, legendre.F90:44 */
  for (int l = 0; l < n; l++) {
    double v;
    switch (l) {
    case 0: v = NDPP_TABLELIN_0(xlow, xhigh, flow, fhigh, P); break;
    case 1: v = NDPP_TABLELIN_1(xlow, xhigh, flow, fhigh, P); break;
    case 2: v = NDPP_TABLELIN_2(xlow, xhigh, flow, fhigh, P); break;
    case 3: v = NDPP_TABLELIN_3(xlow, xhigh, flow, fhigh, P); break;
    case 4: v = NDPP_TABLELIN_4(xlow, xhigh, flow, fhigh, P); break;
    case 5: v = NDPP_TABLELIN_5(xlow, xhigh, flow, fhigh, P); break;
    case 6: v = NDPP_TABLELIN_6(xlow, xhigh, flow, fhigh, P); break;
    case 7: v = NDPP_TABLELIN_7(xlow, xhigh, flow, fhigh, P); break;
    case 8: v = NDPP_TABLELIN_8(xlow, xhigh, flow, fhigh, P); break;
    case 9: v = NDPP_TABLELIN_9(xlow, xhigh, flow, fhigh, P); break; /* == order 7, sic */
    case 10: v = NDPP_TABLELIN_10(xlow, xhigh, flow, fhigh, P); break;
    default: v = 1.0; break; /* legendre.F90:331-332 */
    }
    integrals[l] = integrals[l] + v;
  }
}
