// file6_kernels.hip -- correlated energy-angle (ENDF file 6) scattering moments:
// unit-base interpolation between two tabulated incoming energies fused into the
// CM->lab (integrate_file6_cm_leg) and lab (integrate_file6_lab_leg) integrators,
// and the law-9 evaporation kernel.  Reference: scattdata_header.F90:1085-1450,
// :1521-1717.
//
// Always built with -DNDPP_FAST=0 -ffp-contract=off and written so that every
// output element is accumulated by ONE thread in the reference's order.  Everything but the
// panel integrals of (piecewise-linear f) x P_l follows the Fortran operation by operation;
// those come from Legendre identities (legendre_int.h) instead of the reference's per-order
// closed forms, so results agree with the Fortran to rounding (~1e-15 of a row's largest
// moment), not bit for bit.
// The reference materialises fEmu(M, |ub|) per incoming energy (1.6 MB at
// M=2001, |ub|~100, scattdata_header.F90:1651); here only the per-column
// interpolation coefficients are stored and the column values are recombined
// on the fly from the two tabulated rows.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"
#include "ndpp_math.h"
#include "legendre_int.h"

#if NDPP_FAST
#error "file6_kernels.hip must be compiled with -DNDPP_FAST=0 -ffp-contract=off"
#endif

namespace ndpp {
namespace {

constexpr int HISTOGRAM = 1, LINEAR_LINEAR = 2, LINEAR_LOG = 3, LOG_LINEAR = 4, LOG_LOG = 5;

// interpolate_tab1_array, interpolation.F90:24-123
__device__ double tab1(const double* data, double x) {
  const int n_regions = (int)data[0];
  const int loc_interp = 1 + n_regions;
  const int n_points = (int)data[loc_interp + n_regions];
  const int loc_x = loc_interp + n_regions + 1, loc_y = loc_x + n_points;
  if (x < data[loc_x]) return data[loc_y];
  else if (x > data[loc_x + n_points - 1]) return data[loc_y + n_points - 1];
  int i = bsearch1(data + loc_x, n_points, x);
  if (i < 1) i = 1;  // NaN argument: stay inside the table
  int interp = LINEAR_LINEAR;
  if (n_regions == 1) interp = (int)data[loc_interp];
  else if (n_regions > 1)
    for (int j = 1; j <= n_regions; ++j)
      if (i < data[j]) { interp = (int)data[loc_interp + j - 1]; break; }
  if (interp == HISTOGRAM) return data[loc_y + i - 1];
  const double x0 = data[loc_x + i - 1], x1 = data[loc_x + i];
  const double y0 = data[loc_y + i - 1], y1 = data[loc_y + i];
  double r;
  switch (interp) {
    case LINEAR_LINEAR: r = (x - x0) / (x1 - x0); return (1 - r) * y0 + r * y1;
    case LINEAR_LOG: r = (log(x) - log(x0)) / (log(x1) - log(x0)); return (1 - r) * y0 + r * y1;
    case LOG_LINEAR: r = (x - x0) / (x1 - x0); return exp((1 - r) * log(y0) + r * log(y1));
    case LOG_LOG: r = (log(x) - log(x0)) / (log(x1) - log(x0)); return exp((1 - r) * log(y0) + r * log(y1));
    default: return NAN;
  }
}

// Per incoming energy: the unit-base description of its interpolated table.
struct UbView {
  int nub;            // columns
  double f;           // (Ein - Ei1) / (Ei2 - Ei1), :1655
  const double* Eo;   // [nub]  Eout(i), :1709
  const double* pd;   // [nub]  pdf(i), :1706
  const int* j1;      // [nub]  1-based lower column in row 1
  const double* r1;
  const int* j2;
  const double* r2;
  const double* f1;   // row 1 columns [np1][M]
  const double* f2;
  int M;
  // fEmu(k+1, i+1) of interp_unitbase (:1680,:1701), recombined on the fly
  __device__ __forceinline__ double at(int k, int i) const {
    const double a = (1.0 - f) * ((1.0 - r1[i]) * f1[(size_t)(j1[i] - 1) * M + k] +
                                  r1[i] * f1[(size_t)j1[i] * M + k]);
    return a + f * ((1.0 - r2[i]) * f2[(size_t)(j2[i] - 1) * M + k] +
                    r2[i] * f2[(size_t)j2[i] * M + k]);
  }
};

struct F6Batch {
  int n_ein, G, L, M, NEG, frame_cm, ubcap, npmax;
  double awr;
  const double* ein;
  const int* row_lo;
  const double* e_grid;
  const int* row_ptr;
  const double* eout;
  const double* pdf;
  const int* intt;
  const double* f;
  const double* e_bins;
  // workspace
  double* ub_a;   // [n_ein][npmax] scratch for cast_to_unitbase, row 1
  double* ub_b;   // [n_ein][npmax] row 2
  double* ub;     // [n_ein][ubcap] merged
  int* nub;       // [n_ein]
  double* wf;     // [n_ein]
  double* Eo;     // [n_ein][ubcap]
  double* pd;     // [n_ein][ubcap]
  int* j1; int* j2;       // [n_ein][ubcap]
  double* r1; double* r2; // [n_ein][ubcap]
  double* fEl;    // CM: [n_ein][G][NEG][L]; lab: fint [n_ein][G][M]
  int* glohi;     // CM: [n_ein][2]
  double* ebnds;  // CM: [n_ein][G+2]
  double* out;    // [n_ein][G][L]
  int* status;    // [n_ein]
  unsigned* cm_list;   // CM: the (incoming energy, group, lab energy) items that integrate anything
  unsigned* cm_live;   // CM: [1] how many
  MuGrid grid;
  __device__ UbView view(int e) const {
    UbView v;
    const int k = row_lo[e];
    v.nub = nub[e]; v.f = wf[e];
    v.Eo = Eo + (size_t)e * ubcap; v.pd = pd + (size_t)e * ubcap;
    v.j1 = j1 + (size_t)e * ubcap; v.r1 = r1 + (size_t)e * ubcap;
    v.j2 = j2 + (size_t)e * ubcap; v.r2 = r2 + (size_t)e * ubcap;
    v.f1 = f + (size_t)row_ptr[k] * M; v.f2 = f + (size_t)row_ptr[k + 1] * M;
    v.M = M;
    return v;
  }
};

// cast_to_unitbase, :1554-1609 (np >= 2)
__device__ int cast_ub(const double* Eout, int np, double* ub) {
  double inv_dE = Eout[np - 1] - Eout[0];
  if ((inv_dE >= 0.0) && (inv_dE < DBL_MAX)) inv_dE = 1.0 / inv_dE;
  else inv_dE = 0.0;
  for (int i = 0; i < np - 1; ++i) ub[i] = (Eout[i] - Eout[0]) * inv_dE;
  ub[np - 1] = 1.0;
  return (ub[np - 2] == 1.0) ? np - 1 : np;
}

// merge, array_merge.F90:13-107
__device__ int merge_ub(const double* a, int na, const double* b, int nb, double* res) {
  const double *d1, *d2;
  int n1, n2;
  if (a[na - 1] > b[nb - 1]) { d1 = b; n1 = nb; d2 = a; n2 = na; }
  else { d1 = a; n1 = na; d2 = b; n2 = nb; }
  int i1 = 0, i2 = 0, n = 0;
  const int nab = n1 + n2;
  for (int ires = 0; ires < nab; ++ires) {
    if (i1 < n1 && i2 < n2) {
      if (d1[i1] < d2[i2]) { res[n++] = (d1[i1] == 0.0) ? 1E-14 : d1[i1]; ++i1; }
      else if (d1[i1] == d2[i2]) { res[n++] = d1[i1]; ++i1; ++i2; }
      else { res[n++] = (d2[i2] == 0.0) ? 1E-14 : d2[i2]; ++i2; }
    } else if (i1 < n1) { res[n++] = d1[i1]; ++i1; break; }
    else if (i2 < n2) { res[n++] = d2[i2]; ++i2; }
    else break;
  }
  return n;
}

// Stage U: thread per incoming energy -- unitbase + interp_unitbase minus the
// fEmu table (scattdata_header.F90:1521-1717).
__global__ void f6_unitbase_kernel(F6Batch B) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B.n_ein; e += gridDim.x * blockDim.x) {
    const int k = B.row_lo[e];
    const int o1 = B.row_ptr[k], o2 = B.row_ptr[k + 1];
    const int np1 = o2 - o1, np2 = B.row_ptr[k + 2] - o2;
    const double *eo1 = B.eout + o1, *eo2 = B.eout + o2, *p1 = B.pdf + o1, *p2 = B.pdf + o2;
    double* ub1 = B.ub_a + (size_t)e * B.npmax;
    double* ub2 = B.ub_b + (size_t)e * B.npmax;
    double* ub = B.ub + (size_t)e * B.ubcap;
    const int n1 = cast_ub(eo1, np1, ub1), n2 = cast_ub(eo2, np2, ub2);
    const int nub = merge_ub(ub1, n1, ub2, n2, ub);
    const int intt1 = B.intt[k];  // INTT1 is used for both rows, :1685-1697 (sic)
    const double f = (B.ein[e] - B.e_grid[k]) / (B.e_grid[k + 1] - B.e_grid[k]);
    const double dE1 = eo1[np1 - 1] - eo1[0], dE2 = eo2[np2 - 1] - eo2[0];
    int st = 0;
    for (int i = 0; i < nub; ++i) {
      const double u = ub[i];
      double r = 0.0, pa = 0.0, pb = 0.0;
      int j = bsearch1(ub1, n1, u);
      if (j < 0) { st = NDPP_ST_RANGE; j = 1; }
      if (intt1 == HISTOGRAM) r = 0.0;
      else if (intt1 == LINEAR_LINEAR || intt1 == LOG_LINEAR) r = (u - ub1[j - 1]) / (ub1[j] - ub1[j - 1]);
      else if (intt1 == LINEAR_LOG || intt1 == LOG_LOG) r = log(u / ub1[j - 1]) / log(ub1[j] / ub1[j - 1]);
      if (intt1 == HISTOGRAM || intt1 == LINEAR_LINEAR || intt1 == LINEAR_LOG)
        pa = (1.0 - r) * p1[j - 1] + r * p1[j];
      else if (intt1 == LOG_LINEAR || intt1 == LOG_LOG)
        pa = exp((1.0 - r) * log(p1[j - 1]) + r * log(p1[j]));
      B.j1[(size_t)e * B.ubcap + i] = j;
      B.r1[(size_t)e * B.ubcap + i] = r;
      j = bsearch1(ub2, n2, u);
      if (j < 0) { st = NDPP_ST_RANGE; j = 1; }
      if (intt1 == HISTOGRAM) r = 0.0;
      else if (intt1 == LINEAR_LINEAR || intt1 == LOG_LINEAR) r = (u - ub2[j - 1]) / (ub2[j] - ub2[j - 1]);
      else if (intt1 == LINEAR_LOG || intt1 == LOG_LOG) r = log(u / ub2[j - 1]) / log(ub2[j] / ub2[j - 1]);
      if (intt1 == HISTOGRAM || intt1 == LINEAR_LINEAR || intt1 == LINEAR_LOG)
        pb = (1.0 - r) * p2[j - 1] + r * p2[j];
      else if (intt1 == LOG_LINEAR || intt1 == LOG_LOG)
        pb = exp((1.0 - r) * log(p2[j - 1]) + r * log(p2[j]));
      B.j2[(size_t)e * B.ubcap + i] = j;
      B.r2[(size_t)e * B.ubcap + i] = r;
      B.pd[(size_t)e * B.ubcap + i] = (1.0 - f) * pa + f * pb;
      B.Eo[(size_t)e * B.ubcap + i] = (1.0 - f) * (eo1[0] + dE1 * u) + f * (eo2[0] + dE2 * u);
    }
    B.nub[e] = nub;
    B.wf[e] = f;
    B.status[e] = st;
  }
}

// ---- CM frame --------------------------------------------------------------
// Stage C0: thread per incoming energy -- lab energy window and group range of
// integrate_file6_cm_leg (:1136-1166).  glohi = {g_lo, g_hi} 1-based, or {0,-1}.
__global__ void f6_cm_bounds_kernel(F6Batch B) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B.n_ein; e += gridDim.x * blockDim.x) {
    const UbView v = B.view(e);
    const int nb = B.G + 1, np = v.nub;
    const double Ein = B.ein[e], awr = B.awr;
    const double ap1inv = 1.0 / (awr + 1.0);
    double* Eb = B.ebnds + (size_t)e * (B.G + 2);
    const double Eo_lo = 1E-12;  // :1141 (sic)
    const double Eo_hi = v.Eo[np - 1] +
        (Ein + 2.0 * (awr + 1.0) * sqrt(Ein * v.Eo[np - 1])) * ap1inv * ap1inv;
    int g_lo = 0, g_hi = -1;
    bool live = true;
    if (Eo_lo <= B.e_bins[0]) g_lo = 1;
    else if (Eo_lo >= B.e_bins[nb - 1]) live = false;
    else g_lo = bsearch1(B.e_bins, nb, Eo_lo);
    if (g_lo < 1) { live = false; g_lo = 1; }  // NaN bounds (bad kinematics): no group
    if (live) {
      if (Eo_hi <= B.e_bins[0]) live = false;
      else if (Eo_hi >= B.e_bins[nb - 1]) {
        g_hi = nb - 1;
        Eb[g_lo] = Eo_lo;
        for (int g = g_lo + 1; g <= g_hi; ++g) Eb[g] = B.e_bins[g - 1];
        Eb[g_hi + 1] = B.e_bins[g_hi - 1];  // E_bins(g_hi), :1159 (sic)
      } else {
        g_hi = bsearch1(B.e_bins, nb, Eo_hi);
        if (g_hi < 1) g_hi = g_lo;
        Eb[g_lo] = Eo_lo;
        for (int g = g_lo + 1; g <= g_hi; ++g) Eb[g] = B.e_bins[g - 1];
        Eb[g_hi + 1] = Eo_hi;
      }
    }
    B.glohi[2 * e] = live ? g_lo : 0;
    B.glohi[2 * e + 1] = live ? g_hi : -1;
  }
}

// The integrand of the mu loop of integrate_file6_cm_leg at one lab cosine (:1186-1238):
// f(E_out(CM), mu(CM)) of the unit-base table times the Jacobian and the outgoing-energy pdf.
// The table interval of the previous call is kept (CmCols) with its column data: along the mu
// loop E_out(CM) falls monotonically, so the interval is walked down from there instead of
// searched (same index: the largest i < np with Eo(i) <= E, search.F90:21-71).
//
// Two kinds of arithmetic.  What DECIDES something is evaluated as the reference writes it,
// operation for operation: E_out(CM), the interval it falls in, the Jacobian sqrt(Eo / Eo_cm) and
// the CM cosine with its |mu_c| > 1 cut -- at the last lab cosine that cut is decided by the last
// bit, and the integrand jumps there.  What is CONTINUOUS in those -- the position inside the
// energy interval and on the cosine grid, the interpolation of the four columns, the pdf -- is
// this library's own formulation: reciprocals cached per interval instead of a division per
// cosine, the grid position by one multiplication, interpolations as fused multiply-adds.  It
// differs from the reference expression by ~1e-13 of the value (the reference's own panel
// integrals carry 1e-11, legendre_int.h).
struct CmCols {
  int cur = 0;                       // interval the cached column data belong to (0: none)
  double Eo_lo = 0.0, Eo_hi = 0.0, pd_lo = 0.0, pd_hi = 0.0, dpd = 0.0, rden = 0.0;
  double r1_lo = 0.0, r2_lo = 0.0, r1_hi = 0.0, r2_hi = 0.0;
  double s1_lo = 0.0, s2_lo = 0.0, s1_hi = 0.0, s2_hi = 0.0;     // 1 - r
  const double *c1_lo = nullptr, *c2_lo = nullptr, *c1_hi = nullptr, *c2_hi = nullptr;
};
// kRef: the continuous part too in the reference's own operations (:1199-1236).  Used by the walks
// with more than 8 orders: the closed forms that give their moments of orders 8 ... 10
// (legendre_ref_forms.h) reproduce the reference's rounding noise only on the reference's own
// integrand values.
template <bool kRef = false>
__device__ __forceinline__ double f6_cm_fval(const MuGrid& grid, const UbView& v, CmCols& cc,
                                             double Eo, double c, double mu_l, bool dup_end) {
  const int np = v.nub, M = v.M;
  const double wf = v.f, om_wf = 1.0 - v.f;
  // ---- as the reference writes it
  const double Eo_cm = Eo * (1.0 + c * c - 2.0 * c * mu_l);
  int iEo;
  if (Eo_cm <= 0.0) return 0.0;
  else if (Eo_cm <= v.Eo[0]) iEo = 1;
  else if (Eo_cm >= v.Eo[np - 1]) iEo = np - 1;
  else if (cc.cur >= 1 && Eo_cm >= cc.Eo_lo) {
    iEo = cc.cur;
    if (!(Eo_cm < cc.Eo_hi)) iEo = bsearch1(v.Eo, np, Eo_cm);  // not expected: E rose
  } else if (cc.cur >= 2) {
    iEo = cc.cur - 1;
    while (iEo > 1 && !(v.Eo[iEo - 1] <= Eo_cm)) --iEo;
  } else {
    iEo = bsearch1(v.Eo, np, Eo_cm);
  }
  if (iEo < 1) iEo = 1;  // NaN energy: stay inside the table (the value is NaN anyway)
  if (iEo != cc.cur) {
    cc.cur = iEo;
    cc.Eo_lo = v.Eo[iEo - 1];
    cc.Eo_hi = v.Eo[iEo];
    cc.pd_lo = (dup_end && iEo - 1 == np - 2) ? 0.0 : v.pd[iEo - 1];
    const double pd_hi = (dup_end && iEo == np - 2) ? 0.0 : v.pd[iEo];
    cc.pd_hi = pd_hi;
    // (INTT is always lin-lin after unitbase, :1716; an interval of no width takes its lower end)
    const bool flat = (cc.Eo_hi == cc.Eo_lo);
    cc.rden = flat ? 0.0 : 1.0 / (cc.Eo_hi - cc.Eo_lo);
    cc.dpd = flat ? 0.0 : pd_hi - cc.pd_lo;
    cc.r1_lo = v.r1[iEo - 1]; cc.r2_lo = v.r2[iEo - 1];
    cc.r1_hi = v.r1[iEo]; cc.r2_hi = v.r2[iEo];
    cc.s1_lo = 1.0 - cc.r1_lo; cc.s2_lo = 1.0 - cc.r2_lo;
    cc.s1_hi = 1.0 - cc.r1_hi; cc.s2_hi = 1.0 - cc.r2_hi;
    cc.c1_lo = v.f1 + (size_t)(v.j1[iEo - 1] - 1) * M; cc.c2_lo = v.f2 + (size_t)(v.j2[iEo - 1] - 1) * M;
    cc.c1_hi = v.f1 + (size_t)(v.j1[iEo] - 1) * M; cc.c2_hi = v.f2 + (size_t)(v.j2[iEo] - 1) * M;
  }
  const double J = sqrt(Eo / Eo_cm);
  double mu_c;
  if (mu_l == -1.0) mu_c = -1.0;
  else if (mu_l == 1.0) mu_c = 1.0;
  else {
    mu_c = (mu_l - c) * J;
    if (fabs(mu_c) > 1.0) return 0.0;
  }
  if constexpr (kRef) {
    // ---- the rest as the reference writes it too
    double fEo, pEo;
    if (cc.Eo_hi == cc.Eo_lo) {
      fEo = 0.0;
      pEo = cc.pd_lo;
    } else {
      fEo = (Eo_cm - cc.Eo_lo) / (cc.Eo_hi - cc.Eo_lo);
      pEo = (1.0 - fEo) * cc.pd_lo + fEo * cc.pd_hi;
    }
    int imu_c;
    double f;
    if (fabs(mu_c - 1.0) < 1E-10) {
      imu_c = M - 1;
      f = 1.0;
    } else {
      imu_c = (int)((mu_c + 1.0) / grid.dmu_fgk) + 1;      // deltamu = mu(2) - mu(1), :1122
      if (imu_c > M - 1) imu_c = M - 1;  // the reference would index past the grid
      f = (mu_c - grid.at(imu_c - 1)) / (grid.at(imu_c) - grid.at(imu_c - 1));
    }
    auto colr = [&](const double* c1, double r1, const double* c2, double r2, int k) {
      const double a = (1.0 - wf) * ((1.0 - r1) * c1[k] + r1 * c1[(size_t)M + k]);
      return a + wf * ((1.0 - r2) * c2[k] + r2 * c2[(size_t)M + k]);
    };
    double proby = (1.0 - fEo) * ((1.0 - f) * colr(cc.c1_lo, cc.r1_lo, cc.c2_lo, cc.r2_lo, imu_c - 1) +
                                  f * colr(cc.c1_lo, cc.r1_lo, cc.c2_lo, cc.r2_lo, imu_c));
    proby = proby + fEo * ((1.0 - f) * colr(cc.c1_hi, cc.r1_hi, cc.c2_hi, cc.r2_hi, imu_c - 1) +
                           f * colr(cc.c1_hi, cc.r1_hi, cc.c2_hi, cc.r2_hi, imu_c));
    return proby * J * pEo;
  }
  // ---- continuous in the above: own formulation
  const double fEo = (Eo_cm - cc.Eo_lo) * cc.rden;          // 0 on an interval of no width
  const double pEo = fma(fEo, cc.dpd, cc.pd_lo);
  int k;         // 0-based lower index on the cosine grid, f the position above it
  double f;
  if (fabs(mu_c - 1.0) < 1E-10) {
    k = M - 2;
    f = 1.0;
  } else {
    const double t = fma(mu_c, grid.inv_dmu, grid.inv_dmu);     // (mu_c + 1) / deltamu
    k = (int)t;
    k = k > M - 2 ? M - 2 : k;   // the reference would index past the grid
    f = t - (double)k;
  }
  // fEmu(k+1, i+1) of interp_unitbase (:1680,:1701) = UbView::at, on the cached columns
  auto col = [&](const double* c1, double r1, double s1, const double* c2, double r2, double s2, int kk) {
    const double a = fma(r1, c1[(size_t)M + kk], s1 * c1[kk]);
    const double b = fma(r2, c2[(size_t)M + kk], s2 * c2[kk]);
    return fma(wf, b, om_wf * a);
  };
  const double lo0 = col(cc.c1_lo, cc.r1_lo, cc.s1_lo, cc.c2_lo, cc.r2_lo, cc.s2_lo, k);
  const double lo1 = col(cc.c1_lo, cc.r1_lo, cc.s1_lo, cc.c2_lo, cc.r2_lo, cc.s2_lo, k + 1);
  const double hi0 = col(cc.c1_hi, cc.r1_hi, cc.s1_hi, cc.c2_hi, cc.r2_hi, cc.s2_hi, k);
  const double hi1 = col(cc.c1_hi, cc.r1_hi, cc.s1_hi, cc.c2_hi, cc.r2_hi, cc.s2_hi, k + 1);
  const double lo = fma(f, lo1 - lo0, lo0), hi = fma(f, hi1 - hi0, hi0);
  const double proby = fma(fEo, hi - lo, lo);
  return proby * J * pEo;
}

// What one (incoming energy, group, lab energy point) integrates over (:1168-1185); false:
// nothing (the group is outside the lab energy window, or the `cycle` of :1183).
struct CmItem {
  double Eo, c, mu_l_min, dmu;
  bool dup_end;
};
__device__ __forceinline__ bool f6_cm_item(const F6Batch& B, const UbView& v, int e, int g, int iE,
                                           CmItem& it) {
  if (g < B.glohi[2 * e] || g > B.glohi[2 * e + 1]) return false;
  const int np = v.nub, M = B.M;
  const double* Eb = B.ebnds + (size_t)e * (B.G + 2);
  const double Ein = B.ein[e];
  const double ap1inv = 1.0 / (B.awr + 1.0);
  const double dEo = (Eb[g + 1] - Eb[g]) / (double)(B.NEG - 1);
  double Eo = Eb[g] - dEo;
  for (int k = 1; k <= iE; ++k) Eo = Eo + dEo;  // the reference's running sum, :1171-1173
  const double c = ap1inv * sqrt(Ein / Eo);
  double mu_l_min = (1.0 + c * c - v.Eo[np - 1] / Eo) / (2.0 * c);
  if (mu_l_min < -1.0) mu_l_min = -1.0;
  else if (fabs(mu_l_min - 1.0) < 1E-10) mu_l_min = 1.0;
  else if (mu_l_min > 1.0) return false;  // `cycle`, :1183
  it.Eo = Eo;
  it.c = c;
  it.mu_l_min = mu_l_min;
  it.dmu = (1.0 - mu_l_min) / (double)(M - 1);
  it.dup_end = (v.Eo[np - 1] == v.Eo[np - 2]);  // pdf(np-1) := 0, :1127-1130
  return true;
}

// Stage C1, thread per (incoming energy, group, lab energy point): the mu loop (:1186-1238)
// streamed straight into the panel integrals (:1240-1244).  (A variant with one wave per item and
// the lanes over the lab cosines -- coalesced column reads, ordered sum through LDS, bit-identical
// -- was measured 2.0x (G = 2) to 2.9x (G = 70) slower: every lane then pays the interval search
// and the column set-up that this loop amortises over a run of cosines)
//
// Which items integrate anything is decided first (f6_cm_list_kernel): with many groups most
// (group, lab energy) pairs lie outside an incoming energy's lab window, and a launch over all of
// them leaves the long-running waves of the live ones scattered among empty ones -- less than one
// resident wave per SIMD on average at G = 70 (SQ counters, profiles/r03).  The point kernel runs
// over the compacted list: full waves of equal items.  (The list's order depends on the atomics;
// no result does: an item owns its L outputs.)
__global__ void f6_cm_list_kernel(F6Batch B) {
  const long tot = (long)B.n_ein * B.G * B.NEG;
  for (long t0 = blockIdx.x * (long)blockDim.x; t0 < tot; t0 += (long)gridDim.x * blockDim.x) {
    const long t = t0 + threadIdx.x;
    bool live = false;
    if (t < tot) {
      const int iE = (int)(t % B.NEG) + 1;
      const int g = (int)((t / B.NEG) % B.G) + 1;
      const int e = (int)(t / ((long)B.NEG * B.G));
      double* dst = B.fEl + (size_t)t * B.L;
      for (int l = 0; l < B.L; ++l) dst[l] = 0.0;
      const UbView v = B.view(e);
      CmItem it;
      live = f6_cm_item(B, v, e, g, iE, it);
    }
    // one atomic per wave
    const unsigned long long m = __ballot(live);
    if (m) {
      const int lane = threadIdx.x & 63, leader = __ffsll((long long)m) - 1;
      unsigned base = 0;
      if (lane == leader) base = atomicAdd(B.cm_live, (unsigned)__popcll(m));
      base = __shfl(base, leader);
      if (live) B.cm_list[base + __popcll(m & ((1ull << lane) - 1ull))] = (unsigned)t;
    }
  }
}

template <int LMAX>
__global__ __launch_bounds__(64) void f6_cm_point_kernel(F6Batch B) {
  const long n_live = (long)*B.cm_live;
  for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < n_live;
       k += (long)gridDim.x * blockDim.x) {
    const long t = (long)B.cm_list[k];
    const int iE = (int)(t % B.NEG) + 1;
    const int g = (int)((t / B.NEG) % B.G) + 1;
    const int e = (int)(t / ((long)B.NEG * B.G));
    double* dst = B.fEl + (size_t)t * B.L;
    const UbView v = B.view(e);
    CmItem it;
    if (!f6_cm_item(B, v, e, g, iE, it)) continue;     // (never: the list holds live items)
    const int M = B.M;
    double acc[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) acc[l] = 0.0;
    LinearLegendre<LMAX> walk;       // the M-1 panel integrals, :1240-1244
    CmCols cc;
    auto mu_at = [&](int imu) { return it.mu_l_min + it.dmu * (double)(imu - 1); };
    constexpr bool kRef = LMAX > 8;
    walk.start(mu_at(1), f6_cm_fval<kRef>(B.grid, v, cc, it.Eo, it.c, mu_at(1), it.dup_end));
    const double rh = 1.0 / it.dmu;     // (unused where dmu < 1e-14: those panels contribute nothing)
    int imu = 2;
    for (; imu + 1 <= M; imu += 2) {
      const double x1 = mu_at(imu), x2 = mu_at(imu + 1);
      const double f1 = f6_cm_fval<kRef>(B.grid, v, cc, it.Eo, it.c, x1, it.dup_end);
      const double f2 = f6_cm_fval<kRef>(B.grid, v, cc, it.Eo, it.c, x2, it.dup_end);
      walk.panel2_add(x1, f1, x2, f2, rh, acc);
    }
    if (imu <= M) {
      const double x1 = mu_at(imu);
      walk.panel_add(x1, f6_cm_fval<kRef>(B.grid, v, cc, it.Eo, it.c, x1, it.dup_end), acc);
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
      if (l < B.L) dst[l] = acc[l];
  }
}

// The reaction sum of calc_inelastic_grid on the device (kernels.h launch_reaction_sum): thread
// per (row of the batch, entry); a reaction's incoming energies own distinct rows of the matrices.
__global__ void reaction_sum_kernel(int nb, size_t GL, const double* src, const int* where,
                                    const double* scale, const double* pv, const double* yield,
                                    double* dst, double* nudst) {
  const size_t tot = (size_t)nb * GL;
  for (size_t t = blockIdx.x * (size_t)blockDim.x + threadIdx.x; t < tot; t += (size_t)gridDim.x * blockDim.x) {
    const int k = (int)(t / GL);
    const size_t j = t - (size_t)k * GL;
    const size_t o = (size_t)where[k] * GL + j;
    const double v = src[t] * scale[k] * pv[k];        // scattdata_header.F90:496
    dst[o] = dst[o] + v;                               // scatt.F90:753
    if (nudst) nudst[o] = nudst[o] + yield[k] * v;     // :762
  }
}

// Orders above P7 (L > 8): the panel integrals of orders 8, 9 and 10 are evaluated in the
// reference's own operation order (legendre_ref_forms.h) -- its closed forms carry 1e-10 ... 3e-10
// of cancellation noise of their own there, which nothing else reproduces; the lower orders come
// from Legendre identities (legendre_int.h).  NDPP_ST_ORDER_NOISE, which rounds 2-3 raised on such
// calls, is no longer set.
static inline int order_noise_bits(int) { return 0; }

// Last stage of every batch here: NDPP_ST_NONFINITE for incoming energies whose row holds a NaN
// or an infinity (the reference would have printed it; e.g. a log-interpolated table evaluated
// at the unit-base origin), on top of what the earlier stages flagged.
// `extra`: bits every row of the call carries (none at present).
__global__ void nonfinite_status_kernel(int n_ein, int GL, const double* out, int* status, int extra) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < n_ein; e += gridDim.x * blockDim.x) {
    int st = status[e] | extra;
    for (int k = 0; k < GL; ++k)
      if (!(fabs(out[(size_t)e * GL + k]) <= DBL_MAX)) st |= NDPP_ST_NONFINITE;
    status[e] = st;
  }
}

// Stage C2: thread per incoming energy -- trapezoid over the lab energy points
// and the P0 normalisation (:1246-1264), in the reference's order.
__global__ void f6_cm_finish_kernel(F6Batch B) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B.n_ein; e += gridDim.x * blockDim.x) {
    double* o = B.out + (size_t)e * B.G * B.L;
    for (int k = 0; k < B.G * B.L; ++k) o[k] = 0.0;
    const int g_lo = B.glohi[2 * e], g_hi = B.glohi[2 * e + 1];
    const double* Eb = B.ebnds + (size_t)e * (B.G + 2);
    for (int g = g_lo; g <= g_hi; ++g) {
      const double dEo = (Eb[g + 1] - Eb[g]) / (double)(B.NEG - 1);
      double* dg = o + (size_t)(g - 1) * B.L;
      for (int iE = 1; iE <= B.NEG; ++iE) {
        const double* fEl = B.fEl + (((size_t)e * B.G + (g - 1)) * B.NEG + (iE - 1)) * B.L;
        if ((iE != 1) && (iE != B.NEG))
          for (int l = 0; l < B.L; ++l) dg[l] = dg[l] + 2.0 * fEl[l];
        else
          for (int l = 0; l < B.L; ++l) dg[l] = dg[l] + fEl[l];
      }
      for (int l = 0; l < B.L; ++l) dg[l] = dg[l] * dEo * 0.5;
    }
    double s = 0.0;
    for (int g = g_lo; g <= g_hi; ++g) s = s + o[(size_t)(g - 1) * B.L];
    if (s > 0.0) s = 1.0 / s;
    for (int g = g_lo; g <= g_hi; ++g)
      for (int l = 0; l < B.L; ++l) o[(size_t)(g - 1) * B.L + l] *= s;
  }
}

// ---- lab frame ---------------------------------------------------------------
// Stage L1: thread per (incoming energy, group, mu point) -- the pdf*dE weighted
// sum of fEmu columns between the group edges (:1374-1418), in that order.
// fint[e][g][imu]; glohi[2e+..] unused; a group the reference zeroes gets NaN-free 0
// and is flagged through ebnds[e*(G+2)+g] = 0/1 (1 = integrate).
__global__ void f6_lab_int_kernel(F6Batch B) {
  const long tot = (long)B.n_ein * B.G * B.M;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < tot;
       t += (long)gridDim.x * blockDim.x) {
    const int k = (int)(t % B.M);
    const int g = (int)((t / B.M) % B.G);
    const int e = (int)(t / ((long)B.M * B.G));
    const UbView v = B.view(e);
    const int np = v.nub;
    const double eg = B.e_bins[g], eg1 = B.e_bins[g + 1];
    const bool dup_end = (v.Eo[np - 1] == v.Eo[np - 2]);
    // pdf(iE) = thispdf(iE) * (Eout(iE+1) - Eout(iE)); pdf(np) = thispdf(np); dup -> 0
    auto w = [&](int iE1) -> double {  // 1-based
      if (dup_end && iE1 == np - 1) return 0.0;
      if (iE1 == np) return v.pd[np - 1];
      return v.pd[iE1 - 1] * (v.Eo[iE1] - v.Eo[iE1 - 1]);
    };
    double acc = 0.0;
    int iE_lo, iE_hi;
    bool live = true;
    if (eg < v.Eo[0]) iE_lo = 1;
    else if (eg >= v.Eo[np - 1]) { live = false; iE_lo = 1; }
    else {
      iE_lo = bsearch1(v.Eo, np, eg);
      if (iE_lo < 1) iE_lo = 1;
      const double f_lo = (eg - v.Eo[iE_lo - 1]) / (v.Eo[iE_lo] - v.Eo[iE_lo - 1]);
      acc = acc + f_lo * w(iE_lo) * v.at(k, iE_lo - 1);
      iE_lo = iE_lo + 1;
    }
    if (live) {
      if (eg1 < v.Eo[0]) { live = false; iE_hi = 0; }
      else if (eg1 >= v.Eo[np - 1]) iE_hi = np - 1;
      else {
        iE_hi = bsearch1(v.Eo, np, eg1);
        if (iE_hi < 1) iE_hi = 1;
        const double f_hi = (eg1 - v.Eo[iE_hi - 1]) / (v.Eo[iE_hi] - v.Eo[iE_hi - 1]);
        acc = acc + f_hi * w(iE_hi) * v.at(k, iE_hi - 1);
        iE_hi = iE_hi - 1;
      }
    }
    if (live)
      for (int iE = iE_lo; iE <= iE_hi; ++iE) acc = acc + w(iE) * v.at(k, iE - 1);
    B.fEl[t] = live ? acc : 0.0;
    if (k == 0) B.ebnds[(size_t)e * (B.G + 2) + g] = live ? 1.0 : 0.0;
  }
}

// Stage L2: thread per (incoming energy, group): the M-1 panel integrals (:1421-1425)
template <int LMAX>
__global__ __launch_bounds__(64) void f6_lab_panel_kernel(F6Batch B) {
  const long tot = (long)B.n_ein * B.G;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < tot;
       t += (long)gridDim.x * blockDim.x) {
    const int g = (int)(t % B.G), e = (int)(t / B.G);
    double* dg = B.out + (size_t)t * B.L;
    double acc[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) acc[l] = 0.0;
    if (B.ebnds[(size_t)e * (B.G + 2) + g] != 0.0) {
      const double* fint = B.fEl + (size_t)t * B.M;
      LinearLegendre<LMAX> walk;
      walk.start(B.grid.at(0), fint[0]);
      for (int imu = 1; imu <= B.M - 1; ++imu) {
        walk.panel_add(B.grid.at(imu), fint[imu], acc);
      }
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
      if (l < B.L) dg[l] = acc[l];
  }
}

// Stage L3: thread per incoming energy: f_lo = ONE / sum(distro(1,:)) (:1447-1448).
// flang's SUM intrinsic is Kahan-compensated; reproduced here.
__global__ void f6_lab_norm_kernel(F6Batch B) {
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < B.n_ein; e += gridDim.x * blockDim.x) {
    double* o = B.out + (size_t)e * B.G * B.L;
    double s = 0.0, c = 0.0;
    for (int g = 0; g < B.G; ++g) {
      const double y = o[(size_t)g * B.L] - c;
      const double t = s + y;
      c = (t - s) - y;
      s = t;
    }
    const double f_lo = 1.0 / s;
    for (int k = 0; k < B.G * B.L; ++k) o[k] = o[k] * f_lo;
  }
}

// ---- law 9 ---------------------------------------------------------------------
// thread per (incoming energy, row in {lo,hi}, group): law9_scatter_lab_leg (:1274-1326)
template <int LMAX>
__global__ __launch_bounds__(64) void law9_kernel(int n_ein, const double* ein, const int* row_lo, MuGrid grid,
                            const double* f_tab, const double* edata, int G, int L,
                            const double* e_bins, double* raw /*[n_ein][2][G][L]*/) {
  const long tot = (long)n_ein * 2 * G;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < tot;
       t += (long)gridDim.x * blockDim.x) {
    const int g = (int)(t % G), r = (int)((t / G) % 2), e = (int)(t / (2L * G));
    const double Ein = ein[e];
    const double* fmu = f_tab + (size_t)(row_lo[e] + r) * grid.M;
    double acc[LMAX], pan[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) acc[l] = 0.0;
    const int NR = (int)edata[0];
    const int NE = (int)edata[1 + 2 * NR];
    const double T = tab1(edata, Ein);
    const double U = edata[2 + 2 * NR + 2 * NE];
    const double x = (Ein - U) / T;
    // exp_glibc (ndpp_math.h), the reference's own exp: the differences of exponentials below
    // cancel, a last-bit error of exp shows up at 1e-11 in the group fractions
    const double I = T * T * (1.0 - exp_glibc(-x) * (1.0 + x));
    if (!(Ein - U <= 0.0)) {
      double Egp1 = e_bins[g + 1], Eg = e_bins[g];
      if (Egp1 > (Ein - U)) Egp1 = Ein - U;
      if (Eg > (Ein - U)) Eg = Ein - U;
      double pE = (exp_glibc(-Egp1 / T) * (T + Egp1)) - (exp_glibc(-Eg / T) * (T + Eg));
      pE = -T * pE / I;
      LinearLegendre<LMAX> walk;
      walk.start(grid.at(0), fmu[0]);
      for (int imu = 1; imu <= grid.M - 1; ++imu) {
        walk.panel(grid.at(imu), fmu[imu], pan);
#pragma unroll
        for (int l = 0; l < LMAX; ++l) acc[l] = acc[l] + pan[l] * pE;
      }
    }
    double* o = raw + (size_t)t * L;
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
      if (l < L) o[l] = acc[l];
  }
}

// result = (1-f)*lo + f*hi, scattdata_header.F90:628,:636
__global__ void law9_blend_kernel(int n_ein, const double* w_hi, const double* raw, int GL,
                                  double* out, int* status) {
  const long tot = (long)n_ein * GL;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < tot;
       t += (long)gridDim.x * blockDim.x) {
    const int e = (int)(t / GL), k = (int)(t % GL);
    const double f = w_hi[e];
    const double r = (1.0 - f) * raw[((size_t)2 * e) * GL + k];
    out[t] = r + f * raw[((size_t)2 * e + 1) * GL + k];
    if (k == 0 && status) status[e] = 0;
  }
}



#define F6_TRY(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(NDPP_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

template <int LMAX>
void launch_cm_point(const F6Batch& B) {
  const long tot = (long)B.n_ein * B.G * B.NEG;
  hipLaunchKernelGGL(f6_cm_list_kernel, dim3(nblk(tot, 256)), dim3(256), 0, 0, B);
  hipLaunchKernelGGL((f6_cm_point_kernel<LMAX>), dim3(nblk(tot, 64)), dim3(64), 0, 0, B);
}
template <int LMAX>
void launch_lab_panel(const F6Batch& B) {
  const long tot = (long)B.n_ein * B.G;
  hipLaunchKernelGGL((f6_lab_panel_kernel<LMAX>), dim3(nblk(tot, 64)), dim3(64), 0, 0, B);
}
template <int LMAX>
void launch_law9(int n_ein, const double* ein, const int* row_lo, const MuGrid& grid,
                 const double* f_tab, const double* edata, int G, int L, const double* e_bins,
                 double* raw) {
  const long tot = (long)n_ein * 2 * G;
  hipLaunchKernelGGL((law9_kernel<LMAX>), dim3(nblk(tot, 64)), dim3(64), 0, 0, n_ein, ein,
                     row_lo, grid, f_tab, edata, G, L, e_bins, raw);
}

int check_common(const ndpp_params* p, int G) {
  if (!p) return fail(NDPP_EINVAL, "params is NULL");
  if (p->order < 1 || p->order > NDPP_MAX_ORDER)
    return fail(NDPP_EINVAL, "order=%d outside 1..%d", p->order, NDPP_MAX_ORDER);
  if (p->mu_bins < 2) return fail(NDPP_EINVAL, "mu_bins=%d < 2", p->mu_bins);
  if (G < 1) return fail(NDPP_EINVAL, "need at least one group");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  return NDPP_OK;
}

}  // namespace
}  // namespace ndpp

using namespace ndpp;

void ndpp::launch_reaction_sum(int nb, size_t GL, const double* src, const int* where, const double* scale,
                               const double* pv, const double* yield, double* dst, double* nudst) {
  if (nb <= 0) return;
  hipLaunchKernelGGL(reaction_sum_kernel, dim3(nblk((long)nb * (long)GL, 256)), dim3(256), 0, 0, nb, GL, src,
                     where, scale, pv, yield, dst, nudst);
}

extern "C" int ndpp_file6_leg_batch(const ndpp_params* p, double awr, int frame_cm, int n_ein,
                                    const double* ein, const int* row_lo, int n_rows,
                                    const double* e_grid, const int* row_ptr,
                                    const double* eout, const double* pdf, const int* intt,
                                    const double* f, int G, const double* e_bins, double* out,
                                    int* status) {
  return file6_leg_batch_sink(p, awr, frame_cm, n_ein, ein, row_lo, n_rows, e_grid, row_ptr, eout, pdf,
                              intt, f, G, e_bins, out, status, nullptr);
}

int ndpp::file6_leg_batch_sink(const ndpp_params* p, double awr, int frame_cm, int n_ein,
                               const double* ein, const int* row_lo, int n_rows,
                               const double* e_grid, const int* row_ptr,
                               const double* eout, const double* pdf, const int* intt,
                               const double* f, int G, const double* e_bins, double* out,
                               int* status, DeviceSink* sink, const double* f_dev) {
  if (!p || !ein || !row_lo || !e_grid || !row_ptr || !eout || !pdf || !intt || (!f && !f_dev) || !e_bins || (!out && !sink))
    if (n_ein != 0) return fail(NDPP_EINVAL, "NULL argument");
  if (n_ein < 0 || n_rows < 2) return fail(NDPP_EINVAL, "n_ein=%d n_rows=%d", n_ein, n_rows);
  if (n_ein == 0) return NDPP_OK;
  if (p && p->ne_per_grp < 2) return fail(NDPP_EINVAL, "ne_per_grp=%d < 2", p->ne_per_grp);
  int npmax = 0, ubcap = 0;
  for (int k = 0; k < n_rows; ++k) {
    const int np = row_ptr[k + 1] - row_ptr[k];
    if (np < 2) return fail(NDPP_EINVAL, "row %d has %d outgoing energies (need >= 2)", k, np);
    npmax = std::max(npmax, np);
    if (k + 1 < n_rows) ubcap = std::max(ubcap, np + row_ptr[k + 2] - row_ptr[k + 1]);
  }
  for (int i = 0; i < n_ein; ++i)
    if (row_lo[i] < 0 || row_lo[i] + 1 >= n_rows)
      return fail(NDPP_EINVAL, "row_lo[%d]=%d outside [0, n_rows-2]", i, row_lo[i]);
  int rc = check_common(p, G);
  if (rc) return rc;

  const int L = p->order, M = p->mu_bins, NEG = p->ne_per_grp;
  const size_t ntot = (size_t)row_ptr[n_rows];
  F6Batch B;
  B.n_ein = n_ein; B.G = G; B.L = L; B.M = M; B.NEG = NEG; B.frame_cm = frame_cm;
  B.ubcap = ubcap; B.npmax = npmax; B.awr = awr; B.grid = make_mu_grid(M);
  DevBuf<double> d_ein, d_eg, d_eout, d_pdf, d_f, d_bins, d_uba, d_ubb, d_ub, d_wf, d_Eo, d_pd,
      d_r1, d_r2, d_fEl, d_ebnds, d_out;
  DevBuf<int> d_row, d_rp, d_intt, d_nub, d_j1, d_j2, d_glohi, d_st;
  F6_TRY(d_ein.upload(ein, n_ein));
  F6_TRY(d_row.upload(row_lo, n_ein));
  F6_TRY(d_eg.upload(e_grid, n_rows));
  F6_TRY(d_rp.upload(row_ptr, n_rows + 1));
  F6_TRY(d_eout.upload(eout, ntot));
  F6_TRY(d_pdf.upload(pdf, ntot));
  F6_TRY(d_intt.upload(intt, n_rows));
  if (!f_dev) F6_TRY(d_f.upload(f, ntot * M));
  F6_TRY(d_bins.upload(e_bins, G + 1));
  F6_TRY(d_uba.alloc((size_t)n_ein * npmax));
  F6_TRY(d_ubb.alloc((size_t)n_ein * npmax));
  F6_TRY(d_ub.alloc((size_t)n_ein * ubcap));
  F6_TRY(d_nub.alloc(n_ein));
  F6_TRY(d_wf.alloc(n_ein));
  F6_TRY(d_Eo.alloc((size_t)n_ein * ubcap));
  F6_TRY(d_pd.alloc((size_t)n_ein * ubcap));
  F6_TRY(d_j1.alloc((size_t)n_ein * ubcap));
  F6_TRY(d_j2.alloc((size_t)n_ein * ubcap));
  F6_TRY(d_r1.alloc((size_t)n_ein * ubcap));
  F6_TRY(d_r2.alloc((size_t)n_ein * ubcap));
  const size_t nwork = frame_cm ? (size_t)n_ein * G * NEG * L : (size_t)n_ein * G * M;
  F6_TRY(d_fEl.alloc(nwork));
  F6_TRY(d_glohi.alloc((size_t)2 * n_ein));
  F6_TRY(d_ebnds.alloc((size_t)n_ein * (G + 2)));
  F6_TRY(d_out.alloc((size_t)n_ein * G * L));
  F6_TRY(d_st.alloc(n_ein));
  DevBuf<unsigned> d_list;                       // [items] + the count behind them
  const size_t n_items = frame_cm ? (size_t)n_ein * G * NEG : 0;
  if (n_items >= 0xffffffffull) return fail(NDPP_EINVAL, "file 6 CM batch of %zu items: split the call", n_items);
  F6_TRY(d_list.alloc(n_items + 1));
  F6_TRY(hipMemsetAsync(d_list.p + n_items, 0, sizeof(unsigned), 0));
  B.cm_list = d_list.p; B.cm_live = d_list.p + n_items;
  B.ein = d_ein.p; B.row_lo = d_row.p; B.e_grid = d_eg.p; B.row_ptr = d_rp.p;
  B.eout = d_eout.p; B.pdf = d_pdf.p; B.intt = d_intt.p; B.f = f_dev ? f_dev : d_f.p; B.e_bins = d_bins.p;
  B.ub_a = d_uba.p; B.ub_b = d_ubb.p; B.ub = d_ub.p; B.nub = d_nub.p; B.wf = d_wf.p;
  B.Eo = d_Eo.p; B.pd = d_pd.p; B.j1 = d_j1.p; B.j2 = d_j2.p; B.r1 = d_r1.p; B.r2 = d_r2.p;
  B.fEl = d_fEl.p; B.glohi = d_glohi.p; B.ebnds = d_ebnds.p; B.out = d_out.p; B.status = d_st.p;

  GpuSpan span(nullptr, frame_cm ? kProfFile6Cm : kProfFile6Lab);
  hipLaunchKernelGGL(f6_unitbase_kernel, dim3(nblk(n_ein, 64)), dim3(64), 0, 0, B);
  if (frame_cm) {
    hipLaunchKernelGGL(f6_cm_bounds_kernel, dim3(nblk(n_ein, 64)), dim3(64), 0, 0, B);
    if (L <= 4) launch_cm_point<4>(B);
    else if (L <= 6) launch_cm_point<6>(B);
    else if (L <= 8) launch_cm_point<8>(B);
    else launch_cm_point<11>(B);
    hipLaunchKernelGGL(f6_cm_finish_kernel, dim3(nblk(n_ein, 64)), dim3(64), 0, 0, B);
  } else {
    hipLaunchKernelGGL(f6_lab_int_kernel, dim3(nblk((long)n_ein * G * M, 256)), dim3(256), 0, 0, B);
    if (L <= 4) launch_lab_panel<4>(B);
    else if (L <= 6) launch_lab_panel<6>(B);
    else if (L <= 8) launch_lab_panel<8>(B);
    else launch_lab_panel<11>(B);
    hipLaunchKernelGGL(f6_lab_norm_kernel, dim3(nblk(n_ein, 64)), dim3(64), 0, 0, B);
  }
  hipLaunchKernelGGL(nonfinite_status_kernel, dim3(nblk(n_ein, 64)), dim3(64), 0, 0, n_ein, G * L, d_out.p, d_st.p,
                     order_noise_bits(L));
  span.end();
  F6_TRY(hipGetLastError());
  if (sink) {
    rc = sink->consume(d_out.p, n_ein, (size_t)G * L);
    if (rc) return rc;
  }
  F6_TRY(hipDeviceSynchronize());
  if (!sink) F6_TRY(hipMemcpy(out, d_out.p, sizeof(double) * (size_t)n_ein * G * L, hipMemcpyDeviceToHost));
  if (status) F6_TRY(hipMemcpy(status, d_st.p, sizeof(int) * n_ein, hipMemcpyDeviceToHost));
  return NDPP_OK;
}

extern "C" int ndpp_law9_leg_batch(const ndpp_params* p, int n_ein, const double* ein,
                                   const int* row_lo, const double* w_hi, int n_rows,
                                   const double* f_tab, int n_edata, const double* edata, int G,
                                   const double* e_bins, double* out, int* status) {
  return law9_leg_batch_sink(p, n_ein, ein, row_lo, w_hi, n_rows, f_tab, n_edata, edata, G, e_bins, out,
                             status, nullptr);
}

int ndpp::law9_leg_batch_sink(const ndpp_params* p, int n_ein, const double* ein,
                              const int* row_lo, const double* w_hi, int n_rows,
                              const double* f_tab, int n_edata, const double* edata, int G,
                              const double* e_bins, double* out, int* status, DeviceSink* sink) {
  if (n_ein < 0 || n_rows < 2 || n_edata < 5) return fail(NDPP_EINVAL, "bad sizes");
  if (n_ein == 0) return NDPP_OK;
  if (!p || !ein || !row_lo || !w_hi || !f_tab || !edata || !e_bins || (!out && !sink))
    return fail(NDPP_EINVAL, "NULL argument");
  for (int i = 0; i < n_ein; ++i)
    if (row_lo[i] < 0 || row_lo[i] + 1 >= n_rows)
      return fail(NDPP_EINVAL, "row_lo[%d]=%d outside [0, n_rows-2]", i, row_lo[i]);
  {
    const int NR = (int)edata[0];
    if (NR < 0 || 2 + 2 * NR > n_edata) return fail(NDPP_EINVAL, "edata: bad NR");
    const int NE = (int)edata[1 + 2 * NR];
    if (NE < 1 || 2 + 2 * NR + 2 * NE + 1 > n_edata) return fail(NDPP_EINVAL, "edata: bad NE");
  }
  int rc = check_common(p, G);
  if (rc) return rc;
  const int L = p->order, M = p->mu_bins, GL = G * L;
  DevBuf<double> d_ein, d_w, d_f, d_ed, d_bins, d_raw, d_out;
  DevBuf<int> d_row, d_st;
  F6_TRY(d_ein.upload(ein, n_ein));
  F6_TRY(d_w.upload(w_hi, n_ein));
  F6_TRY(d_row.upload(row_lo, n_ein));
  F6_TRY(d_f.upload(f_tab, (size_t)n_rows * M));
  F6_TRY(d_ed.upload(edata, n_edata));
  F6_TRY(d_bins.upload(e_bins, G + 1));
  F6_TRY(d_raw.alloc((size_t)n_ein * 2 * GL));
  F6_TRY(d_out.alloc((size_t)n_ein * GL));
  F6_TRY(d_st.alloc(n_ein));
  const MuGrid grid = make_mu_grid(M);
  GpuSpan span(nullptr, kProfLaw9);
  if (L <= 4) launch_law9<4>(n_ein, d_ein.p, d_row.p, grid, d_f.p, d_ed.p, G, L, d_bins.p, d_raw.p);
  else if (L <= 6) launch_law9<6>(n_ein, d_ein.p, d_row.p, grid, d_f.p, d_ed.p, G, L, d_bins.p, d_raw.p);
  else if (L <= 8) launch_law9<8>(n_ein, d_ein.p, d_row.p, grid, d_f.p, d_ed.p, G, L, d_bins.p, d_raw.p);
  else launch_law9<11>(n_ein, d_ein.p, d_row.p, grid, d_f.p, d_ed.p, G, L, d_bins.p, d_raw.p);
  hipLaunchKernelGGL(law9_blend_kernel, dim3(nblk((long)n_ein * GL, 256)), dim3(256), 0, 0, n_ein,
                     d_w.p, d_raw.p, GL, d_out.p, d_st.p);
  hipLaunchKernelGGL(nonfinite_status_kernel, dim3(nblk(n_ein, 64)), dim3(64), 0, 0, n_ein, GL, d_out.p, d_st.p,
                     order_noise_bits(L));
  span.end();
  F6_TRY(hipGetLastError());
  if (sink) {
    rc = sink->consume(d_out.p, n_ein, (size_t)GL);
    if (rc) return rc;
  }
  F6_TRY(hipDeviceSynchronize());
  if (!sink) F6_TRY(hipMemcpy(out, d_out.p, sizeof(double) * (size_t)n_ein * GL, hipMemcpyDeviceToHost));
  if (status) F6_TRY(hipMemcpy(status, d_st.p, sizeof(int) * n_ein, hipMemcpyDeviceToHost));
  return NDPP_OK;
}
