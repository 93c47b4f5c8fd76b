// legendre_int.h -- integrals of a piecewise-linear function against Legendre polynomials,
//   I_l = int_a^b (f_a + (f_b - f_a) (x - a) / (b - a)) P_l(x) dx,   l = 0 .. L-1,
// the job of the reference's calc_int_pn_tablelin (legendre.F90:22-336), which spells out one
// closed form per order.  Here they come from two identities of the Legendre polynomials
// (Bonnet's recurrence and the derivative relation):
//   Q_l(x) := int P_l       = (P_{l+1}(x) - P_{l-1}(x)) / (2l+1),          Q_0 = x
//   R_l(x) := int x P_l     = ((l+1) Q_{l+1}(x) + l Q_{l-1}(x)) / (2l+1),  R_0 = x^2 / 2
//   I_l = f_a [Q_l]_a^b + s ([R_l]_a^b - a [Q_l]_a^b),   s = (f_b - f_a) / (b - a).
// Q and R are evaluated once per abscissa, so a walk over consecutive panels costs one set of
// evaluations per point (the closed forms cost ~16 divisions and several hundred operations
// per panel and order set).  Agreement with the closed forms: ~1e-15 of the largest moment
// (both lose the same digits to cancellation on narrow panels); tests/test_file6_oracle.py.
//
// Two conventions of the reference are kept because results depend on them:
//   * a panel narrower than 1e-14 contributes nothing (legendre.F90:44);
//   * its order-9 branch is a verbatim copy of the order-7 branch (legendre.F90:117-126 vs
//     :95-104), so the tenth moment it returns is the eighth: reproduced (kLegendreOrder9Is7).
//
// Orders 8, 9 and 10 (walks with more than 8 orders, i.e. scattering orders P8 ... P10): the
// reference's own closed forms carry 1e-10 ... 3e-10 of rounding noise there, which only its own
// operation order reproduces -- those three moments come from legendre_ref_forms.h (the forms
// derived at compile time in the reference's association), the lower ones from the identities.
#pragma once

#include "legendre_ref_forms.h"
#include "ndpp_math.h"

namespace ndpp {
#if NDPP_FAST
inline namespace fast_arith {
#else
inline namespace strict_arith {
#endif

constexpr bool kLegendreOrder9Is7 = true;

// Q_l(x), R_l(x) for l < LMAX (additive constants dropped: only differences are used).
// This is the library's own formulation, not a reference expression: it is written with explicit
// fused multiply-adds (the same bits on the device and in the host build of tests/hostsim,
// whatever -ffp-contract says) -- 3 operations per order for Bonnet's recurrence, 2 for Q, 2 for R.
template <int LMAX>
NDPP_HD void legendre_antiderivatives(double x, double* Q, double* R) {
  double P[LMAX + 2];
  P[0] = 1.0;
  P[1] = x;
#pragma unroll
  for (int n = 1; n <= LMAX; ++n) {     // P_{n+1} = (2n+1)/(n+1) x P_n - n/(n+1) P_{n-1}
    const double an = (double)(2 * n + 1) / (double)(n + 1), bn = (double)n / (double)(n + 1);
    P[n + 1] = fma(an * x, P[n], -(bn * P[n - 1]));
  }
  double Qe[LMAX + 1];
  Qe[0] = x;
#pragma unroll
  for (int l = 1; l <= LMAX; ++l) Qe[l] = (P[l + 1] - P[l - 1]) * (1.0 / (double)(2 * l + 1));
  R[0] = 0.5 * x * x;
#pragma unroll
  for (int l = 1; l < LMAX; ++l) {
    const double dl = (double)(l + 1) / (double)(2 * l + 1), el = (double)l / (double)(2 * l + 1);
    R[l] = fma(dl, Qe[l + 1], el * Qe[l - 1]);
  }
#pragma unroll
  for (int l = 0; l < LMAX; ++l) Q[l] = Qe[l];
}

// Walk over the panels of a piecewise-linear function, left to right.
template <int LMAX>
struct LinearLegendre {
  static constexpr bool kRefHigh = LMAX > 8;     // orders >= 8 in the reference's own forms
  static constexpr int LID = kRefHigh ? 8 : LMAX;  // orders from the identities
  double x, f;               // left end of the next panel
  double Q[LMAX], R[LMAX];   // antiderivatives there
  refform::Powers pw;        // kRefHigh: x ** n there
  NDPP_HD void start(double x0, double f0) {
    x = x0;
    f = f0;
    legendre_antiderivatives<LMAX>(x0, Q, R);
    if constexpr (kRefHigh) pw = refform::powers_of(x0);
  }
  // orders 8 .. LMAX-1 of the panel [xa, xb] in the reference's forms (9 is its copy of 7)
  template <bool kAdd>
  NDPP_HD static void high_orders(double xa, double xb, double fa, double fb, const refform::Powers& pa,
                                  const refform::Powers& pb, double* out) {
    const double v8 = refform::panel<8>(xa, xb, fa, fb, pa, pb);
    out[8] = kAdd ? out[8] + v8 : v8;
    if constexpr (LMAX > 9) {
      const double v9 = kLegendreOrder9Is7 ? refform::panel<7>(xa, xb, fa, fb, pa, pb)
                                           : refform::panel<9>(xa, xb, fa, fb, pa, pb);
      out[9] = kAdd ? out[9] + v9 : v9;
    }
    if constexpr (LMAX > 10) {
      const double v10 = refform::panel<10>(xa, xb, fa, fb, pa, pb);
      out[10] = kAdd ? out[10] + v10 : v10;
    }
  }
  // kAdd = false: out[l] = integral over [x, x1] (f linear from f to f1);
  // kAdd = true:  out[l] += that (the running sum of a walk, one rounding fewer per panel);
  // then (x1, f1) becomes the left end
  template <bool kAdd>
  NDPP_HD void panel_impl(double x1, double f1, double* out) {
    double Q1[LMAX], R1[LMAX];
    legendre_antiderivatives<LMAX>(x1, Q1, R1);
    refform::Powers pw1;
    if constexpr (kRefHigh) pw1 = refform::powers_of(x1);
    const double h = x1 - x;
    if (h < 1e-14) {         // FP_PRECISION, legendre.F90:44
      if constexpr (!kAdd) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) out[l] = 0.0;
      }
    } else {
      const double s = (f1 - f) / h;
#pragma unroll
      for (int l = 0; l < LID; ++l) {
        const double dQ = Q1[l] - Q[l];
        const double t = fma(-x, dQ, R1[l] - R[l]);
        out[l] = kAdd ? fma(s, t, fma(f, dQ, out[l])) : fma(s, t, f * dQ);
      }
      if constexpr (kRefHigh) high_orders<kAdd>(x, x1, f, f1, pw, pw1, out);
    }
    x = x1;
    f = f1;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) { Q[l] = Q1[l]; R[l] = R1[l]; }
    if constexpr (kRefHigh) pw = pw1;
  }
  NDPP_HD void panel(double x1, double f1, double* out) { panel_impl<false>(x1, f1, out); }
  NDPP_HD void panel_add(double x1, double f1, double* acc) { panel_impl<true>(x1, f1, acc); }

  // Two consecutive panels [x, x1], [x1, x2] of a walk over (nearly) equal steps, added to acc.
  // rh ~ 1 / (x1 - x): the caller's reciprocal of the nominal step replaces the division of the
  // slope (the abscissae of a uniform walk are rounded sums, so a panel's width differs from the
  // nominal one by a few 1e-16 -- 1e-13 of the slope, whose term is the small one of the two).
  // The second set of antiderivatives lands in Q, R again: no state is copied.
  NDPP_HD void panel2_add(double x1, double f1, double x2, double f2, double rh, double* acc) {
    double Q1[LMAX], R1[LMAX];
    legendre_antiderivatives<LMAX>(x1, Q1, R1);
    refform::Powers pw1;
    if constexpr (kRefHigh) pw1 = refform::powers_of(x1);
    add_between(x, f, Q, R, pw, x1, f1, Q1, R1, pw1, rh, acc);
    legendre_antiderivatives<LMAX>(x2, Q, R);
    if constexpr (kRefHigh) pw = refform::powers_of(x2);
    add_between(x1, f1, Q1, R1, pw1, x2, f2, Q, R, pw, rh, acc);
    x = x2;
    f = f2;
  }
  NDPP_HD static void add_between(double xa, double fa, const double* Qa, const double* Ra, const refform::Powers& pa,
                                  double xb, double fb, const double* Qb, const double* Rb, const refform::Powers& pb,
                                  double rh, double* acc) {
    if (xb - xa < 1e-14) return;         // FP_PRECISION, legendre.F90:44
    const double s = (fb - fa) * rh;
#pragma unroll
    for (int l = 0; l < LID; ++l) {
      const double dQ = Qb[l] - Qa[l];
      const double t = fma(-xa, dQ, Rb[l] - Ra[l]);
      acc[l] = fma(s, t, fma(fa, dQ, acc[l]));
    }
    if constexpr (kRefHigh) high_orders<true>(xa, xb, fa, fb, pa, pb, acc);
  }
};

// one panel on its own (calc_int_pn_tablelin's signature)
template <int LMAX>
NDPP_HD void tablelin(double xlow, double xhigh, double flow, double fhigh, double* v) {
  LinearLegendre<LMAX> w;
  w.start(xlow, flow);
  w.panel(xhigh, fhigh, v);
}

}  // inline namespace
}  // namespace ndpp
