// legendre_ref_forms.h -- the panel integral of (a linear f) x P_l for l = 7, 8, 10 in the
// REFERENCE'S ROUNDINGS (calc_int_pn_tablelin, legendre.F90:22-336), for scattering orders above
// P7 (the reference admits scatt_order <= 10, ndpp.F90:290-301).
//
// Why.  legendre_int.h evaluates these integrals from two Legendre identities; up to P7 that
// agrees with the reference to its own rounding noise (4e-11 of the largest moment over the
// default grid).  The reference's closed forms lose digits to cancellation -- every term is
// ~ C f x^(l+2) / (x_hi - x_lo) while the result is ~ f (x_hi - x_lo) -- and from l = 8 on their noise
// alone reaches 1e-10 ... 3e-10 of the largest moment: no other evaluation can stay within 1e-10 of
// them.  Most of that noise is a function of the abscissae only (the rounding of x^n), so the same
// operations on the same operands reproduce it.
//
// What.  Nothing here is taken from the reference's text: the forms are DERIVED at compile time.
// With P_l = sum_k c_k x^k and f(x) = (f_lo (x_hi - x) + f_hi (x - x_lo)) / (x_hi - x_lo),
//   int_lo^hi f P_l = [ sum_k  a_k ((k+1) f_hi + f_lo) x_hi^(k+2)  -  x_lo sum_k b_k f_hi x_hi^(k+1) ] / (x_hi - x_lo)
//                   + [ sum_k  a_k (f_hi + (k+1) f_lo) x_lo^(k+2)  -  b_k f_lo x_hi x_lo^(k+1)      ] / (x_hi - x_lo),
//   a_k = c_k / ((k+1)(k+2)),  b_k = c_k / (k+1).
// A computer-algebra system prints this over the common denominator N = lcm of the denominators
// of all a_k, b_k, with integer coefficients A_k = N a_k, B_k = N b_k, the x_lo-group of the first
// bracket with the common factor g = gcd(B_k) pulled out, powers descending, every product left to
// right -- which is the operation order of the reference's expressions (checked: bit-identical to
// the oracle's restatement of them, hence to the Fortran, on random panels; tests/test_file6_oracle.py).
// c_k come from Bonnet's recurrence in exact rational arithmetic (constexpr, 64-bit).
//
// x ** n is what flang makes of an integer power: LLVM's powi, square-and-multiply from the low
// bit (ndpp_math.h powi).  Must be compiled without FMA contraction (file6_kernels.hip is).
#pragma once

#include "ndpp_math.h"

namespace ndpp {
#if NDPP_FAST
inline namespace fast_arith {
#else
inline namespace strict_arith {
#endif

namespace refform {

constexpr long long gcd_ll(long long a, long long b) {
  a = a < 0 ? -a : a;
  b = b < 0 ? -b : b;
  while (b) { const long long t = a % b; a = b; b = t; }
  return a;
}
struct Rat {
  long long n, d;      // d > 0, lowest terms
};
constexpr Rat rat(long long n, long long d) {
  if (d < 0) { n = -n; d = -d; }
  const long long g = gcd_ll(n, d);
  return g ? Rat{n / g, d / g} : Rat{0, 1};
}
constexpr Rat operator*(Rat a, Rat b) { return rat(a.n * b.n, a.d * b.d); }
constexpr Rat operator-(Rat a, Rat b) { return rat(a.n * b.d - b.n * a.d, a.d * b.d); }

constexpr int kMaxOrder = 10;
struct Poly {
  Rat c[kMaxOrder + 1];      // c[k] x^k
};
// (n+1) P_{n+1} = (2n+1) x P_n - n P_{n-1}
constexpr Poly legendre(int l) {
  Poly p0{}, p1{};
  for (int k = 0; k <= kMaxOrder; ++k) { p0.c[k] = Rat{0, 1}; p1.c[k] = Rat{0, 1}; }
  p0.c[0] = Rat{1, 1};
  p1.c[1] = Rat{1, 1};
  if (l == 0) return p0;
  for (int n = 1; n < l; ++n) {
    Poly q{};
    for (int k = 0; k <= kMaxOrder; ++k) {
      const Rat up = k > 0 ? rat(2 * n + 1, n + 1) * p1.c[k - 1] : Rat{0, 1};
      q.c[k] = up - rat(n, n + 1) * p0.c[k];
    }
    p0 = p1;
    p1 = q;
  }
  return p1;
}

// the printed form of order L: terms by descending power k = L, L-2, ...
template <int L>
struct Form {
  static constexpr int kTerms = L / 2 + 1;
  int k[kTerms];
  double A[kTerms], B[kTerms];     // |N a_k|, |N b_k| / g
  bool neg[kTerms];                // c_k < 0
  double g, inv_n;                 // common factor of the x_lo group; RN(1 / N)
};
template <int L>
constexpr Form<L> make_form() {
  const Poly p = legendre(L);
  Form<L> f{};
  long long N = 1;
  Rat a[Form<L>::kTerms]{}, b[Form<L>::kTerms]{};
  for (int j = 0; j < Form<L>::kTerms; ++j) {
    const int k = L - 2 * j;
    f.k[j] = k;
    a[j] = p.c[k] * rat(1, (long long)(k + 1) * (k + 2));
    b[j] = p.c[k] * rat(1, k + 1);
    N = N / gcd_ll(N, a[j].d) * a[j].d;
    N = N / gcd_ll(N, b[j].d) * b[j].d;
  }
  long long g = 0;
  for (int j = 0; j < Form<L>::kTerms; ++j) g = gcd_ll(g, b[j].n * (N / b[j].d));
  for (int j = 0; j < Form<L>::kTerms; ++j) {
    const long long An = a[j].n * (N / a[j].d), Bn = b[j].n * (N / b[j].d);
    f.neg[j] = An < 0;
    f.A[j] = (double)(An < 0 ? -An : An);
    f.B[j] = (double)((Bn < 0 ? -Bn : Bn) / g);
  }
  f.g = (double)g;
  f.inv_n = 1.0 / (double)N;
  return f;
}

// x^0 .. x^(kMaxOrder+2) as the reference's x ** n
struct Powers {
  double p[kMaxOrder + 3];
};
NDPP_HD Powers powers_of(double x) {
  Powers w;
#pragma unroll
  for (int n = 0; n <= kMaxOrder + 2; ++n) w.p[n] = n == 1 ? x : powi(x, n);
  return w;
}

// int_xl^xh (f linear from fl to fh) P_L, L = 7, 8 or 10 (any L <= 10 with at least two terms works)
template <int L>
NDPP_HD double panel(double xl, double xh, double fl, double fh, const Powers& pl, const Powers& ph) {
  constexpr Form<L> F = make_form<L>();
  double hi = 0.0, grp = 0.0, lo = 0.0;
#pragma unroll
  for (int j = 0; j < Form<L>::kTerms; ++j) {
    const int k = F.k[j];
    const double kp = (double)(k + 1);
    const double th = (F.A[j] * (kp * fh + fl)) * ph.p[k + 2];
    const double tg = (F.B[j] * fh) * ph.p[k + 1];
    if (j == 0) { hi = th; grp = tg; }
    else if (F.neg[j]) { hi = hi - th; grp = grp - tg; }
    else { hi = hi + th; grp = grp + tg; }
  }
  hi = hi - (F.g * grp) * xl;
#pragma unroll
  for (int j = 0; j < Form<L>::kTerms; ++j) {
    const int k = F.k[j];
    const double kp = (double)(k + 1);
    const double t1 = (F.A[j] * (fh + kp * fl)) * pl.p[k + 2];
    const double t2 = (((F.g * F.B[j]) * fl) * xh) * pl.p[k + 1];
    if (j == 0) lo = t1 - t2;
    else if (F.neg[j]) lo = (lo - t1) + t2;
    else lo = (lo + t1) - t2;
  }
  const double dx = xh - xl;
  return F.inv_n * hi / dx + F.inv_n * lo / dx;
}

}  // namespace refform
}  // inline namespace
}  // namespace ndpp
