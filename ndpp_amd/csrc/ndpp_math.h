// ndpp_math.h -- scalar building blocks of the scattering-moment kernels.
//
// Every function is NDPP_HD (host+device) so that the exact code the gfx950
// kernels run can also be driven lane-by-lane on the CPU by the test-only
// host simulator (tests/hostsim) in the GPU-less build container.  The product
// path (libndpp_hip.so) only ever calls them from device code.
//
// Arithmetic follows the reference expressions operation for operation
// (file:line citations are under the reference's src/); the library is built
// with -ffp-contract=off so no mul+add is fused that the reference does not
// fuse.
#pragma once

#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define NDPP_HD __host__ __device__ __forceinline__
#else
#define NDPP_HD inline
#endif

// NDPP_FAST selects how the free-gas kernel value K(mu) and the Simpson leaf
// correction are evaluated:
//   0  "strict": every +,-,*,/,sqrt of the reference expression, in its order
//      (build with -ffp-contract=off) -- results differ from the reference only
//      through exp()'s last bit;
//   1  "fast" (product default): divides by (E_in,E_out)-invariants become
//      multiplications by their reciprocals, 1/alpha and 1/sqrt(alpha) come from
//      one rsqrt, x/15 becomes x*(1/15), and FMA contraction is allowed.  Each
//      K then carries a few ulp (~5e-16) instead of ~1; the adaptive trees are
//      unchanged except where an accept/refine test sits within that distance
//      of its threshold.  Parity vs the reference stays ~1e-14 (bar: 1e-10).
#ifndef NDPP_FAST
#define NDPP_FAST 1
#endif

namespace ndpp {
// translation units of one library may be built with either arithmetic; the
// inline namespace keeps their (inline) functions apart for the linker
#if NDPP_FAST
inline namespace fast_arith {
#else
inline namespace strict_arith {
#endif

// constants.F90:35 -- the reference truncates pi; results depend on it.
constexpr double kPi = 3.1415926535898;
constexpr double kFourPi = 4.0 * kPi;  // `4.0_8 * PI` folds to this exact product

constexpr int kMaxL = 11;  // scatt_order <= 10 (ndpp.F90:290-301)

// x**n, integer n, as LLVM lowers it for flang (llvm.powi: square-and-multiply
// from the low bit).  With a constant n the loop folds to the multiply chain.
NDPP_HD double powi(double a, int b) {
  double r = 1.0;
  for (;;) {
    if (b & 1) r *= a;
    b /= 2;
    if (b == 0) break;
    a *= a;
  }
  return r;
}

// calc_pn, legendre.F90:349-432: closed polynomial forms, NOT the recurrence.
template <int N>
NDPP_HD double pn(double x) {
  if constexpr (N == 0) return 1.0;
  else if constexpr (N == 1) return x;
  else if constexpr (N == 2) return 1.5 * x * x - 0.5;
  else if constexpr (N == 3) return 2.5 * x * x * x - 1.5 * x;
  else if constexpr (N == 4) return 4.375 * powi(x, 4) - 3.75 * x * x + 0.375;
  else if constexpr (N == 5)
    return 7.875 * powi(x, 5) - 8.75 * x * x * x + 1.875 * x;
  else if constexpr (N == 6)
    return 14.4375 * powi(x, 6) - 19.6875 * powi(x, 4) + 6.5625 * x * x - 0.3125;
  else if constexpr (N == 7)
    return 26.8125 * powi(x, 7) - 43.3125 * powi(x, 5) + 19.6875 * x * x * x -
           2.1875 * x;
  else if constexpr (N == 8)
    return 50.2734375 * powi(x, 8) - 93.84375 * powi(x, 6) +
           54.140625 * powi(x, 4) - 9.84375 * x * x + 0.2734375;
  else if constexpr (N == 9)
    return 94.9609375 * powi(x, 9) - 201.09375 * powi(x, 7) +
           140.765625 * powi(x, 5) - 36.09375 * x * x * x + 2.4609375 * x;
  else if constexpr (N == 10)
    return 180.42578125 * powi(x, 10) - 427.32421875 * powi(x, 8) +
           351.9140625 * powi(x, 6) - 117.3046875 * powi(x, 4) +
           13.53515625 * x * x - 0.24609375;
  else return 1.0;
}

// Value that the optimiser must take as given (no instruction is emitted for it).  Two uses:
// a product is rounded before the subtraction that consumes it instead of being contracted
// into it, and a constant stays in a register across a loop instead of being rebuilt from
// literals at every use (gfx950's three-operand instructions take one scalar or literal
// operand only, so c1*y + c2 with two literal constants costs two extra moves each time).
NDPP_HD double opaque(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  asm("" : "+v"(x));
#elif defined(__x86_64__)
  asm("" : "+x"(x));
#endif
  return x;
}

// x / C for C = 6, 12, 15 with the bits of the IEEE quotient, without the division sequence (13
// issue slots on gfx950): q = RN(x RN(1/C)), then one correction by the exact residual
// x - q C (an FMA), q' = RN(q + residual RN(1/C)) -- the correctly rounded quotient (Markstein's
// theorem; for these three divisors checked against the hardware division on 9.2e8 operands incl.
// every multiple of 15 below 3e8 and its two neighbours: no mismatch).  Not for quotients below
// ~1e-290, where the residual is subnormal: nothing of that size reaches a result.
template <int C>
NDPP_HD double div_by(double x) {
  constexpr double r = 1.0 / (double)C;
  const double q = x * r;
  return fma(fma(-q, (double)C, x), r, q);
}

// x / y for a divisor y that is constant over many quotients, given ry = RN(1 / y): the same
// two-step scheme as div_by (q = RN(x ry) corrected once by the exact residual).  With a correctly
// rounded reciprocal this is the IEEE quotient for all but a measure-zero set of (x, y) (checked
// against the hardware division on 8e8 random pairs over 4e5 divisors: no mismatch); used where
// the reference divides by A kT, kT and the mu-grid spacing, three of the eight divisions of
// every kernel value.
NDPP_HD double quot_by(double x, double y, double ry) {
  const double q = x * ry;
  return fma(fma(-q, y, x), ry, q);
}

// The two constants of the innermost Horner step a*y + c of P_3..P_10 (pn_all below), held
// in registers for the lifetime of a walk.  kPinned = false leaves them ordinary constants
// (no register is reserved).
struct PnConsts {
  double a[8], c[8];
};
template <bool kPinned = false>
NDPP_HD PnConsts make_pn_consts() {
  PnConsts k;
  const double a[8] = {2.5, 4.375, 7.875, 14.4375, 26.8125, 50.2734375, 94.9609375, 180.42578125};
  const double c[8] = {-1.5, -3.75, -8.75, -19.6875, -43.3125, -93.84375, -201.09375, -427.32421875};
  for (int i = 0; i < 8; ++i) {
    k.a[i] = kPinned ? opaque(a[i]) : a[i];
    k.c[i] = kPinned ? opaque(c[i]) : c[i];
  }
  return k;
}

#if NDPP_FAST
// P_{L0}..P_{L0+N-1} at x, even/odd Horner in x^2 (same polynomials as calc_pn, other
// association: differs from the reference forms by rounding only).  Every order has its own
// short chain, so a range costs only what its orders cost.
template <int L0, int N>
NDPP_HD void pn_range(double x, double* out, const PnConsts& k) {
  const double y = x * x;
#define NDPP_PN(l, expr) if constexpr (L0 <= (l) && (l) < L0 + N) out[(l) - L0] = (expr)
  NDPP_PN(0, 1.0);
  NDPP_PN(1, x);
  NDPP_PN(2, 1.5 * y - 0.5);
  NDPP_PN(3, x * (k.a[0] * y + k.c[0]));
  NDPP_PN(4, (k.a[1] * y + k.c[1]) * y + 0.375);
  NDPP_PN(5, x * ((k.a[2] * y + k.c[2]) * y + 1.875));
  NDPP_PN(6, ((k.a[3] * y + k.c[3]) * y + 6.5625) * y - 0.3125);
  NDPP_PN(7, x * (((k.a[4] * y + k.c[4]) * y + 19.6875) * y - 2.1875));
  NDPP_PN(8, (((k.a[5] * y + k.c[5]) * y + 54.140625) * y - 9.84375) * y + 0.2734375);
  NDPP_PN(9, x * ((((k.a[6] * y + k.c[6]) * y + 140.765625) * y - 36.09375) * y + 2.4609375));
  NDPP_PN(10, ((((k.a[7] * y + k.c[7]) * y + 351.9140625) * y - 117.3046875) * y + 13.53515625) * y - 0.24609375);
#undef NDPP_PN
  (void)y;
}
#else
template <int L0, int N, int I = 0>
NDPP_HD void pn_range(double x, double* out, const PnConsts& k) {
  if constexpr (I < N) {
    out[I] = pn<L0 + I>(x);
    pn_range<L0, N, I + 1>(x, out, k);
  }
}
#endif
template <int L>
NDPP_HD void pn_all(double x, double* out, const PnConsts& k) {
  pn_range<0, L>(x, out, k);
}

template <int L>
NDPP_HD void pn_all(double x, double* out) {
  pn_all<L>(x, out, make_pn_consts());
}

NDPP_HD double pn_rt(int n, double x) {
  switch (n) {
    case 0: return pn<0>(x);
    case 1: return pn<1>(x);
    case 2: return pn<2>(x);
    case 3: return pn<3>(x);
    case 4: return pn<4>(x);
    case 5: return pn<5>(x);
    case 6: return pn<6>(x);
    case 7: return pn<7>(x);
    case 8: return pn<8>(x);
    case 9: return pn<9>(x);
    case 10: return pn<10>(x);
    default: return 1.0;
  }
}

// Uniform mu grid of scatt_init, scattdata_header.F90:251-257 (0-based i).
struct MuGrid {
  int M;
  double dmu;      // TWO / real(mu_bins - 1, 8)
  double dmu_fgk;  // global_mu(2) - global_mu(1), freegas.F90:437
  double inv_dmu;  // 1 / dmu_fgk (fast path)
  NDPP_HD double at(int i) const {
    return (i == M - 1) ? 1.0 : -1.0 + (double)i * dmu;
  }
};

NDPP_HD MuGrid make_mu_grid(int M) {
  MuGrid g;
  g.M = M;
  g.dmu = 2.0 / (double)(M - 1);
  g.dmu_fgk = (-1.0 + 1.0 * g.dmu) - (-1.0);
  if (M == 2) g.dmu_fgk = 1.0 - (-1.0);
  g.inv_dmu = 1.0 / g.dmu_fgk;
  return g;
}

// calc_FG_Eout_bounds, freegas.F90:154-181
NDPP_HD void fg_eout_bounds(double A, double kT, double Ein, double& lo,
                            double& hi) {
  double alpha = (A - 1.0) / (A + 1.0);
  alpha = alpha * alpha;
  lo = 0.001 * alpha * Ein;
  if (Ein > 300.0 * kT / A)
    hi = 12.0 * kT * (A + 1.0) / A + 1.5 * Ein;
  else
    hi = 12.0 * kT * (A + 1.0) / A + 2.0 * Ein;
}

// Quantities of one (E_in, E_out) pair that calc_sab / calc_fgk recompute on
// every call (freegas.F90:207-210, :451-457).  Each member is the value of
// exactly the sub-expression the reference evaluates, so hoisting changes no
// bits.
struct FgPair {
  double s1;    // sqrt(Eout / Ein)
  double c2;    // ((A + ONE) / A) ** 2
  double kT;
  double EpE;   // Ein + Eout
  double s2;    // sqrt(Ein * Eout)
  double AkT;   // A * kT
  double beta;  // (Eout - Ein) / kT
#if NDPP_FAST
  double C1;    // s1 / kT * c2 / sqrt(4 pi): everything of lterm but f(mu)
  double p, q;  // alpha(mu) ~ p - q*mu (used for estimates only: see fg_alpha)
#endif
  double inv_AkT;   // RN(1 / AkT), RN(1 / kT): divisions by a per-pair constant (quot_by)
  double inv_kT;
};

NDPP_HD FgPair make_pair(double A, double kT, double Ein, double Eout) {
  FgPair q;
  q.s1 = sqrt(Eout / Ein);
  double r = (A + 1.0) / A;
  q.c2 = r * r;
  q.kT = kT;
  q.EpE = Ein + Eout;
  q.s2 = sqrt(Ein * Eout);
  q.AkT = A * kT;
  q.beta = (Eout - Ein) / kT;
#if NDPP_FAST
  q.C1 = q.s1 / kT * q.c2 / sqrt(kFourPi);
  q.p = q.EpE / q.AkT;
  q.q = 2.0 * q.s2 / q.AkT;
#endif
  q.inv_AkT = 1.0 / q.AkT;
  q.inv_kT = 1.0 / kT;
  return q;
}

// 1/sqrt(x) for x >= 1e-6: hardware seed (v_rsq_f64, ~2^-23) and one third-order step
// r (1 + e/2 + 3 e^2/8), e = 1 - x r^2 (residual error (5/16) e^3 < 2^-68) on the device,
// plain 1/sqrt on the host.
NDPP_HD double fast_rsqrt(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double r = __builtin_amdgcn_rsq(x);
  const double e = fma(-(x * r), r, 1.0);
  return fma(r * e, fma(0.375, e, 0.5), r);
#else
  return 1.0 / sqrt(x);
#endif
}

#if !NDPP_FAST
// exp for the strict arithmetic on the device: the exp the reference itself calls.
//
// Where the incoming energy is far below kT the reference's result follows the last bit of
// every kernel value, and with it the last bit of its exp, which is not
// an in-tree routine but the libm the Fortran is linked against: glibc 2.35's exp (Ubuntu
// 2.35-0ubuntu3.11 in this image; sysdeps/ieee754/dbl-64/e_exp.c, the algorithm glibc has used
// since 2.28), in the build that glibc's ifunc selects on every x86-64 CPU with FMA + AVX2.
// That routine is restated here operation by operation (it is deterministic: the same fused
// multiply-adds in the same order give the same bits on any IEEE machine):
//   k = round(x * 128/ln2),  r = x - k ln2/128 (ln2/128 in two parts),
//   2^(k/128) = 2^(k div 128) * H_j (1 + T_j), j = k mod 128 (exp_tab.inc, tools/gen_exp_table.py),
//   exp(x) = s + s * (T_j + r + r^2 (C2 + r C3) + r^4 (C4 + r C5)),  s = 2^(k div 128) H_j.
// Bit-identical to the host's exp on 4e7 arguments incl. the subnormal range (tests/test_hostsim.py).
// Needs -ffp-contract=off: only the fma() calls below may be fused.
// (Algorithm and constants: glibc / ARM Optimized Routines, see NOTICE.md; no source text of either.)
#if defined(__HIPCC__)
#define NDPP_TABLE static __device__ const
#else
#define NDPP_TABLE static const
#endif
NDPP_TABLE uint64_t kExpTab[256] = {
#include "exp_tab.inc"
};
NDPP_HD uint64_t f64_bits(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return (uint64_t)__double_as_longlong(x);
#else
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
#endif
}
NDPP_HD double bits_f64(uint64_t u) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __longlong_as_double((long long)u);
#else
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
#endif
}
NDPP_HD double exp_glibc(double x) {
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
               C5 = 0x1.1111167a4d017p-7;
  const uint64_t ix = f64_bits(x);
  uint32_t abstop = (uint32_t)(ix >> 52) & 0x7ffu;
  if (abstop - 0x3c9u > 0x3eu) {
    if ((int32_t)(abstop - 0x3c9u) < 0) return 1.0 + x;           // |x| < 2^-54
    if (abstop >= 0x409u) {                                        // |x| >= 1024, inf, nan
      if (ix == 0xfff0000000000000ull) return 0.0;
      if (abstop >= 0x7ffu) return 1.0 + x;
      return (ix >> 63) ? 0.0 : 1.0 / 0.0;
    }
    abstop = 0;                                                    // 512 <= |x| < 1024: see below
  }
  double kd = fma(x, InvLn2N, Shift);
  const uint64_t ki = f64_bits(kd);
  kd = kd - Shift;
  double r = fma(kd, NegLn2hiN, x);
  r = fma(kd, NegLn2loN, r);
  const uint64_t idx = 2 * (ki & 127u), top = ki << 45;
  const double tail = bits_f64(kExpTab[idx]);
  uint64_t sbits = kExpTab[idx + 1] + top;
  const double p23 = fma(C3, r, C2);
  const double tr = r + tail;
  const double r2 = r * r;
  const double p45 = fma(r, C5, C4);
  const double t1 = fma(p23, r2, tr);
  const double r4 = r2 * r2;
  const double tmp = fma(r4, p45, t1);
  if (abstop == 0) {
    // the result may overflow or be subnormal: scaled evaluation with one final rounding
    if ((ki & 0x80000000u) == 0) {
      sbits -= 1009ull << 52;
      const double sc = bits_f64(sbits);
      return 0x1p1009 * fma(sc, tmp, sc);
    }
    sbits += 1022ull << 52;
    const double sc = bits_f64(sbits);
    const double st = sc * tmp;
    double y = sc + st;
    if (y < 1.0) {
      const double hi = 1.0 + y;
      double lo = (sc - y) + st;
      lo = ((1.0 - hi) + y) + lo;
      y = (hi + lo) - 1.0;
      if (y == 0.0) y = 0.0;
    }
    return 0x1p-1022 * y;
  }
  const double sc = bits_f64(sbits);
  return fma(sc, tmp, sc);
}
// The main path of exp_glibc alone: 2^-54 <= |x| < 512 (no result near the subnormal range, no
// tiny argument).  exp_glibc_plain(x) says whether x is in that range.
NDPP_HD bool exp_glibc_plain(double x) {
  const double ax = fabs(x);
  return ax >= 0x1p-54 && ax < 512.0;
}
NDPP_HD double exp_glibc_core(double x) {
  const double InvLn2N = 0x1.71547652b82fep+7, Shift = 0x1.8p52;
  const double NegLn2hiN = -0x1.62e42fefa0000p-8, NegLn2loN = -0x1.cf79abc9e3b3ap-47;
  const double C2 = 0x1.ffffffffffdbdp-2, C3 = 0x1.555555555543cp-3, C4 = 0x1.55555cf172b91p-5,
               C5 = 0x1.1111167a4d017p-7;
  double kd = fma(x, InvLn2N, Shift);
  const uint64_t ki = f64_bits(kd);
  kd = kd - Shift;
  double r = fma(kd, NegLn2hiN, x);
  r = fma(kd, NegLn2loN, r);
  const uint64_t idx = 2 * (ki & 127u), top = ki << 45;
  const double tail = bits_f64(kExpTab[idx]);
  const uint64_t sbits = kExpTab[idx + 1] + top;
  const double p23 = fma(C3, r, C2);
  const double tr = r + tail;
  const double r2 = r * r;
  const double p45 = fma(r, C5, C4);
  const double t1 = fma(p23, r2, tr);
  const double r4 = r2 * r2;
  const double tmp = fma(r4, p45, t1);
  const double sc = bits_f64(sbits);
  return fma(sc, tmp, sc);
}
// the reference's exp: the host's libm on the host (it IS the reference's there), its
// restatement on the device
NDPP_HD double exp_ref(double x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return exp_glibc(x);
#else
  return exp(x);
#endif
}
#else
NDPP_HD double exp_ref(double x) { return exp(x); }
#endif

// calc_sab, freegas.F90:188-228
NDPP_HD double fg_sab(const FgPair& q, double mu) {
  double lterm = q.s1 / q.kT * q.c2;
  double alpha = (q.EpE - 2.0 * mu * q.s2) / q.AkT;
  if (alpha < 1.0E-6) alpha = 1.0E-6;
  double t = alpha + q.beta;
  double sab = -(t * t) / (4.0 * alpha);
  if (sab < -225.0) {
    sab = 0.0;
  } else {
    sab = lterm * exp_ref(sab) / (sqrt(kFourPi * alpha));
    if (sab < 2.0E-10) sab = 0.0;
  }
  return sab;
}

// brent_mu, freegas.F90:235-345
NDPP_HD double fg_brent_mu(const FgPair& q, double TH, double thresh, double lo,
                           double hi) {
  double a = lo, b = hi, c = 0.0, d = 1.7976931348623157e308;  // huge(0d0)
  double fa = fg_sab(q, a) - thresh;
  double fb = fg_sab(q, b) - thresh;
  double fc = 0.0, s = 0.0, fs = 0.0, tmp;
  bool mflag;
  if (fa * fb >= 0.0) return (fa < fb) ? a : b;
  if (fabs(fa) < fabs(fb)) {
    tmp = a; a = b; b = tmp;
    tmp = fa; fa = fb; fb = tmp;
  }
  c = a;
  fc = fa;
  mflag = true;
  // The reference loop has no iteration cap; |a-b| shrinks by bisection at
  // least every other step, so 200 is never reached in exact arithmetic -- it
  // only guarantees that a NaN-poisoned lane leaves the loop on the GPU.
  for (int it = 0; it < 200 && (fb != 0.0) && (fabs(a - b) > TH); ++it) {
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
      mflag = true;
    } else {
      if ((mflag && (fabs(b - c) < TH)) || (!mflag && (fabs(c - d) < TH))) {
        s = (a + b) * 0.5;
        mflag = true;
      } else {
        mflag = false;
      }
    }
    fs = fg_sab(q, s) - thresh;
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

// find_FG_mu, freegas.F90:356-409
NDPP_HD void fg_find_mu(const FgPair& q, double A, double Ein, double Eout,
                        double sab_threshold, double brent_thresh, double& mu_lo,
                        double& mu_hi) {
  double alpha_max = sqrt(q.beta * q.beta + 1.0) - 1.0;
  double mu_max = (Ein + Eout - alpha_max * A * q.kT) / (2.0 * sqrt(Ein * Eout));
  if (fabs(mu_max) > 1.0) {
    mu_lo = -1.0;
    mu_hi = 1.0;
  } else {
    double sab_max = fg_sab(q, mu_max);
    double thr = sab_max * sab_threshold;
    if (fg_sab(q, -1.0) > thr)
      mu_lo = -1.0;
    else
      mu_lo = fg_brent_mu(q, brent_thresh, thr, -1.0, mu_max);
    if (fg_sab(q, 1.0) > thr)
      mu_hi = 1.0;
    else
      mu_hi = fg_brent_mu(q, brent_thresh, thr, mu_max, 1.0);
  }
}

// The tabulated rows one inner integral reads: one row of f_tab, or the two bracketing rows of a
// job (always adjacent: row_lo, row_lo + 1).  A lane keeps the BYTE OFFSET of row_lo from the
// start of the batch's table (32 bits: the table of one batch stays below 4 GiB, checked on the
// host); a lookup is then the uniform table pointer plus a 32-bit per-lane offset -- the
// addressing mode global loads take from a scalar base -- instead of 64-bit pointer arithmetic
// per lane and visit.  (A pair table -- the four values of a lookup in 32 contiguous bytes -- cut
// the walk's L2 reads by 12 % and bought nothing, round 3: experiments/README.md.)
struct FRows {
  unsigned off;         // bytes from f_tab to row_lo
};
template <int R>
struct FView {          // built at the point of use: `base` and `stride` are wave-uniform
  const char* base;     // f_tab
  unsigned off;         // the lane's row_lo
  unsigned stride;      // bytes per row
  // &f[row_lo + r][i]: f[i] and f[i + 1] are read through this one address (one 16-byte load)
  NDPP_HD const double* at(int r, int i) const {
    return reinterpret_cast<const double*>(base + (size_t)(unsigned)(off + (unsigned)r * stride + 8u * (unsigned)i));
  }
};
template <int R>
NDPP_HD FView<R> f_view(const double* f_tab, const FRows& f, int M) {
  return FView<R>{reinterpret_cast<const char*>(f_tab), f.off, 8u * (unsigned)M};
}
// every row's values at grid points i and i + 1
template <int R>
NDPP_HD void rows_at(const double* const* f, int i, double* f0, double* f1) {
#pragma unroll
  for (int r = 0; r < R; ++r) { f0[r] = f[r][i]; f1[r] = f[r][i + 1]; }
}
template <int R>
NDPP_HD void rows_at(const FView<R>& f, int i, double* f0, double* f1) {
#pragma unroll
  for (int r = 0; r < R; ++r) {
    const double* p = f.at(r, i);
    f0[r] = p[0];
    f1[r] = p[1];
  }
}

// The l-independent factor of calc_fgk (freegas.F90:437-470):
//   fgk(l,mu) = lterm*exp(arg)/sqrt(4 pi alpha) * calc_pn(l,mu) = K(mu)*P_l(mu)
// The reference multiplies by P_l last, so K*P_l reproduces fgk bit for bit
// and one K serves every Legendre order.
#if NDPP_FAST
// K(mu) = [C1 * f(mu)] * E(mu): f(mu) is the only factor that depends on which
// tabulated row is integrated; E(mu) = exp(-(alpha+beta)^2/4alpha)/sqrt(alpha) is
// shared by the two bracketing rows of one incoming energy.
// position on the uniform grid: t = (mu + 1) / dmu = i + interp
NDPP_HD int fg_grid_pos(const MuGrid& g, double mu, double& interp) {
  const double t = fma(mu, g.inv_dmu, g.inv_dmu);
  int i = (int)t;
  const int top = g.M - 2;
  i = i > top ? top : i;
  i = i < 0 ? 0 : i;
  interp = t - (double)i;
  return i;
}

NDPP_HD double fg_fval(const MuGrid& g, const double* f, double mu) {
  double interp;
  const int i = fg_grid_pos(g, mu, interp);
  const double f0 = f[i], f1 = f[i + 1];
  return f0 + interp * (f1 - f0);
}

// The same lookup split in two so that the table reads can be issued long before
// their values are needed.
struct FvLoad { double f0, f1, interp; };
NDPP_HD FvLoad fg_fval_load(const MuGrid& g, const double* f, double mu) {
  FvLoad v;
  const int i = fg_grid_pos(g, mu, v.interp);
  v.f0 = f[i];
  v.f1 = f[i + 1];
  return v;
}
template <int R>
NDPP_HD void fg_fval_load_rows(const MuGrid& g, const FView<R>& f, double mu, FvLoad* v) {
  double interp;
  const int i = fg_grid_pos(g, mu, interp);
  double f0[R], f1[R];
  rows_at<R>(f, i, f0, f1);
#pragma unroll
  for (int r = 0; r < R; ++r) { v[r].interp = interp; v[r].f0 = f0[r]; v[r].f1 = f1[r]; }
}
NDPP_HD double fg_fval_use(const FvLoad& v) { return v.f0 + v.interp * (v.f1 - v.f0); }

// exp(x) for x <= 0, <= 2 ulp: Cody-Waite reduction x = n ln2 + r, |r| <= ln2/2,
// exp(r) = 1 + r + r^2 q(r) (degree-9 near-minimax q), scaled by 2^n.  Leaner than
// the library exp because it needs no overflow handling; underflow is ldexp's.
NDPP_HD double exp_neg(double x) {
  x = fmax(x, -750.0);  // exp(-750) already underflows to 0
  const double n = rint(x * 1.4426950408889634074);
  double r = fma(n, -6.93147180369123816490e-01, x);
  r = fma(n, -1.90821492927058770002e-10, r);
  double q = 2.52479970560794156e-08;
  q = fma(q, r, 2.76229515090678550e-07);
  q = fma(q, r, 2.75568902848521959e-06);
  q = fma(q, r, 2.48015150230775488e-05);
  q = fma(q, r, 1.98412701876778889e-04);
  q = fma(q, r, 1.38888889215492179e-03);
  q = fma(q, r, 8.33333333322663732e-03);
  q = fma(q, r, 4.16666666666140328e-02);
  q = fma(q, r, 1.66666666666667518e-01);
  q = fma(q, r, 5.00000000000000555e-01);
  const double p = fma(r * r, q, r) + 1.0;
  return ldexp(p, (int)n);
}

// alpha(mu) ~ p - q mu with the clamp of freegas.F90:459.  (The reference's own roundings of
// alpha -- product 2 mu s2, difference, quotient by A kT -- differ from this by up to 1e-9 towards
// forward scattering, where the numerator cancels; reproducing them moved no result of the
// 266-case two-group fixture and cost 1.1 %: the deviations of the product arithmetic are
// accept/refine decisions)
NDPP_HD double fg_alpha(const FgPair& q, double mu) {
  return fmax(q.p - q.q * mu, 1.0E-6);
}

NDPP_HD double fg_E(const FgPair& q, double mu) {
  const double alpha = fg_alpha(q, mu);
  const double r = fast_rsqrt(alpha);
  const double tr = (alpha + q.beta) * r;
  // The reference zeroes the kernel where the exponent is <= -708 (freegas.F90:464);
  // here exp() simply underflows (values < 1e-307 either way, no effect on any sum
  // or accept/refine test).
  return exp_neg(-0.25 * (tr * tr)) * r;
}

// Two E evaluations advanced in lockstep: each is one long dependent chain
// (rsqrt Newton steps, 10 Horner FMAs), and a wave only has one partner on its
// SIMD to hide FP64 latency behind, so the two chains are interleaved by hand.
NDPP_HD void fg_E2(const FgPair& q, double mu0, double mu1, double& E0, double& E1) {
  const double a0 = fg_alpha(q, mu0);
  const double a1 = fg_alpha(q, mu1);
#if defined(__HIP_DEVICE_COMPILE__)
  double r0 = __builtin_amdgcn_rsq(a0), r1 = __builtin_amdgcn_rsq(a1);   // see fast_rsqrt
  const double e0 = fma(-(a0 * r0), r0, 1.0), e1 = fma(-(a1 * r1), r1, 1.0);
  r0 = fma(r0 * e0, fma(0.375, e0, 0.5), r0); r1 = fma(r1 * e1, fma(0.375, e1, 0.5), r1);
#else
  const double r0 = 1.0 / sqrt(a0), r1 = 1.0 / sqrt(a1);
#endif
  const double t0 = (a0 + q.beta) * r0, t1 = (a1 + q.beta) * r1;
  const double x0 = fmax(-0.25 * (t0 * t0), -750.0), x1 = fmax(-0.25 * (t1 * t1), -750.0);
  const double n0 = rint(x0 * 1.4426950408889634074), n1 = rint(x1 * 1.4426950408889634074);
  double s0 = fma(n0, -6.93147180369123816490e-01, x0), s1 = fma(n1, -6.93147180369123816490e-01, x1);
  s0 = fma(n0, -1.90821492927058770002e-10, s0); s1 = fma(n1, -1.90821492927058770002e-10, s1);
  double q0 = 2.52479970560794156e-08, q1 = 2.52479970560794156e-08;
  q0 = fma(q0, s0, 2.76229515090678550e-07); q1 = fma(q1, s1, 2.76229515090678550e-07);
  q0 = fma(q0, s0, 2.75568902848521959e-06); q1 = fma(q1, s1, 2.75568902848521959e-06);
  q0 = fma(q0, s0, 2.48015150230775488e-05); q1 = fma(q1, s1, 2.48015150230775488e-05);
  q0 = fma(q0, s0, 1.98412701876778889e-04); q1 = fma(q1, s1, 1.98412701876778889e-04);
  q0 = fma(q0, s0, 1.38888889215492179e-03); q1 = fma(q1, s1, 1.38888889215492179e-03);
  q0 = fma(q0, s0, 8.33333333322663732e-03); q1 = fma(q1, s1, 8.33333333322663732e-03);
  q0 = fma(q0, s0, 4.16666666666140328e-02); q1 = fma(q1, s1, 4.16666666666140328e-02);
  q0 = fma(q0, s0, 1.66666666666667518e-01); q1 = fma(q1, s1, 1.66666666666667518e-01);
  q0 = fma(q0, s0, 5.00000000000000555e-01); q1 = fma(q1, s1, 5.00000000000000555e-01);
  const double p0 = fma(s0 * s0, q0, s0) + 1.0, p1 = fma(s1 * s1, q1, s1) + 1.0;
  E0 = ldexp(p0, (int)n0) * r0;
  E1 = ldexp(p1, (int)n1) * r1;
}

NDPP_HD double fg_K(const FgPair& q, const MuGrid& g, const double* f, double mu) {
  return (q.C1 * fg_fval(g, f, mu)) * fg_E(q, mu);
}
#else
// calc_fgk (freegas.F90:437-470) for R tabulated rows at one point, every operation of the
// reference expression in its order.  Only f(mu) depends on the row: the grid position, alpha,
// the exponent, exp and the square root are evaluated once and used for all rows.
// S = sqrt(x) and rS ~ 1 / S for x in the normal range (here x = 4 pi alpha >= 1e-5).
// Device: the compiler's own double-precision square root (v_rsq_f64 seed, one coupled
// Goldschmidt step for g ~ sqrt(x) and h ~ 1 / (2 sqrt(x)), two residual corrections of g) without
// the scaling it wraps around it for arguments below 2^-767 -- the same instructions on the same
// operands, so the same bits -- and its h, corrected once against S, as the reciprocal the two-step
// quotients by S need (within an ulp of 1 / S; quot_by).  One sequence of 13 instructions instead
// of a square root (18) and a division (11).  Host: sqrt and a division.
NDPP_HD void sqrt_and_reciprocal(double x, double& S, double& rS) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  const double r = fma(-h, g, 0.5);
  g = fma(g, r, g);
  h = fma(h, r, h);
  double d = fma(-g, g, x);
  g = fma(d, h, g);
  d = fma(-g, g, x);
  g = fma(d, h, g);
  S = g;
  const double t = h + h;
  rS = fma(t, fma(-g, t, 1.0), t);
#else
  S = sqrt(x);
  rS = 1.0 / S;
#endif
}
template <int R, class F>
NDPP_HD void fg_K_rows(const FgPair& q, const MuGrid& g, const F& f, double mu, double* K) {
  int i;  // 0-based lower grid index
  if (mu <= -1.0)
    i = 0;
  else if (mu >= 1.0)
    i = g.M - 2;
  else
    i = (int)quot_by(mu + 1.0, g.dmu_fgk, g.inv_dmu);
  if (i > g.M - 2) i = g.M - 2;  // the reference would index past the table here
  double m0 = g.at(i), m1 = g.at(i + 1);
  // (mu - m0) / (m1 - m0), correctly rounded without the division sequence: the cell widths of the
  // uniform grid differ from the first one by a few 1e-16, so one Newton step from RN(1 / width_0)
  // gives this cell's reciprocal to the last bit or the one before, which is all the two-step
  // quotient needs (quot_by; the exact quotient of two doubles is never closer to a rounding
  // boundary than the error that leaves)
  const double den = m1 - m0;
  const double rden = fma(g.inv_dmu, fma(-den, g.inv_dmu, 1.0), g.inv_dmu);
  double interp = quot_by(mu - m0, den, rden);
  double alpha = quot_by(q.EpE - 2.0 * mu * q.s2, q.AkT, q.inv_AkT);
  if (alpha < 1.0E-6) alpha = 1.0E-6;
  double t = alpha + q.beta;
  double arg = -(t * t) / (4.0 * alpha);
  if (arg <= -708.0) {
    for (int r = 0; r < R; ++r) K[r] = 0.0;
    return;
  }
  const double E = exp_ref(arg);
  double f0[R], f1[R];
  rows_at<R>(f, i, f0, f1);
  // two rows divide by the same S: its reciprocal comes with the square root, two two-step quotients
  double S, rS;
  if constexpr (R > 1) {
    sqrt_and_reciprocal(kFourPi * alpha, S, rS);
  } else {
    S = sqrt(kFourPi * alpha);
    rS = 0.0;
  }
  for (int r = 0; r < R; ++r) {
    double fval = (1.0 - interp) * f0[r] + interp * f1[r];
    double lterm = quot_by(fval * q.s1, q.kT, q.inv_kT) * q.c2;
    K[r] = (R > 1) ? quot_by(lterm * E, S, rS) : lterm * E / S;
  }
}
NDPP_HD double fg_K(const FgPair& q, const MuGrid& g, const double* f, double mu) {
  const double* fr[1] = {f};
  double K;
  fg_K_rows<1>(q, g, (const double* const*)fr, mu, &K);
  return K;
}

// The kernel values of a visit's two new points in ONE straight-line block.  fg_K_rows leaves
// early where the exponent is below -708, and the restated exp branches on its argument's range:
// called twice, the two evaluations sit in separate basic blocks and their long dependent chains
// (quotient, exp, square root, quotients) run one after the other.  Here both points go through
// the branch-free main path of exp side by side; a point outside that path's range (an exponent
// beyond -512, or below 2^-54 in size) is redone by the full routine afterwards, a dead point
// (exponent <= -708) gets its zero by selection.  Same operations on the same operands: same bits.
template <int R, class F>
NDPP_HD void fg_K_rows_pair(const FgPair& q, const MuGrid& g, const F& f, double muA, double muB, double* KA,
                            double* KB) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double mu[2] = {muA, muB};
  int idx[2];
  double interp[2], alpha[2], arg[2], argc[2], S[2], rS[2], E[2];
  bool dead[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    int i;
    if (mu[k] <= -1.0) i = 0;
    else if (mu[k] >= 1.0) i = g.M - 2;
    else i = (int)quot_by(mu[k] + 1.0, g.dmu_fgk, g.inv_dmu);
    if (i > g.M - 2) i = g.M - 2;
    idx[k] = i;
    const double m0 = g.at(i), m1 = g.at(i + 1);
    const double den = m1 - m0;
    const double rden = fma(g.inv_dmu, fma(-den, g.inv_dmu, 1.0), g.inv_dmu);
    interp[k] = quot_by(mu[k] - m0, den, rden);
    double a = quot_by(q.EpE - 2.0 * mu[k] * q.s2, q.AkT, q.inv_AkT);
    if (a < 1.0E-6) a = 1.0E-6;
    alpha[k] = a;
    const double t = a + q.beta;
    arg[k] = -(t * t) / (4.0 * a);
    dead[k] = arg[k] <= -708.0;
    argc[k] = dead[k] ? -1.0 : arg[k];
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    E[k] = exp_glibc_core(argc[k]);
    if constexpr (R > 1) {
      sqrt_and_reciprocal(kFourPi * alpha[k], S[k], rS[k]);
    } else {
      S[k] = sqrt(kFourPi * alpha[k]);
      rS[k] = 0.0;
    }
  }
  if (!(exp_glibc_plain(argc[0]) && exp_glibc_plain(argc[1]))) {
    E[0] = exp_glibc(argc[0]);
    E[1] = exp_glibc(argc[1]);
  }
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    double f0[R], f1[R];
    rows_at<R>(f, idx[k], f0, f1);
    double* K = k == 0 ? KA : KB;
    for (int r = 0; r < R; ++r) {
      const double fval = (1.0 - interp[k]) * f0[r] + interp[k] * f1[r];
      const double lterm = quot_by(fval * q.s1, q.kT, q.inv_kT) * q.c2;
      const double v = (R > 1) ? quot_by(lterm * E[k], S[k], rS[k]) : lterm * E[k] / S[k];
      K[r] = dead[k] ? 0.0 : v;
    }
  }
#else
  fg_K_rows<R>(q, g, f, muA, KA);
  fg_K_rows<R>(q, g, f, muB, KB);
#endif
}
#endif

// tolab, scattdata_header.F90:1466-1496
NDPP_HD double tolab(double R, double w) {
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

}  // inline namespace
}  // namespace ndpp
