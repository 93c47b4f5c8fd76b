// file4_kernels.hip -- two-body CM kinematics moments (integrate_file4_cm_leg).
// Built with -DNDPP_FAST=0 -ffp-contract=off: only + - * / sqrt are involved,
// all IEEE-exact on gfx950, so the results are bit-identical to the reference.
#include <hip/hip_runtime.h>

#include <algorithm>

#include "kernels.h"
#include "ndpp_math.h"

#if NDPP_FAST
#error "file4_kernels.hip must be compiled with -DNDPP_FAST=0 -ffp-contract=off"
#endif

namespace ndpp {
namespace {

// integrate_file4_cm_leg, scattdata_header.F90:956-1078, one (call, group) per
// thread, all orders jointly.  Groups are independent: the reference's early
// `return` (:1015) only skips groups whose own bounds are both clamped to +1,
// which this thread detects itself.
template <int LMAX>
__device__ void file4_group(const MuGrid& grid, const double* fw, double Ein,
                            double awr, double Q, double eg, double eg1, int L,
                            double* dg /*[L]*/) {
  const int M = grid.M;
  const double dw = grid.dmu_fgk;  // w(2) - w(1), :980
  const double R = awr * sqrt((1.0 + Q * (awr + 1.0) / (awr * Ein)));
  const double onepawr2 = (1.0 + awr) * (1.0 + awr);
  const double onepR2 = 1.0 + R * R;
  const double inv2REin = 0.5 / (R * Ein);
  double acc[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) acc[l] = 0.0;

  double wlo = (eg * onepawr2 - Ein * onepR2) * inv2REin;
  if (wlo < -1.0) wlo = -1.0; else if (wlo > 1.0) wlo = 1.0;
  const int ilo = (int)((wlo + 1.0) / dw) + 1;  // 1-based like the reference
  double whi = (eg1 * onepawr2 - Ein * onepR2) * inv2REin;
  if (whi < -1.0) whi = -1.0; else if (whi > 1.0) whi = 1.0;
  const int ihi = (int)((whi + 1.0) / dw) + 1;

  const bool skip = (wlo == whi) && (wlo == -1.0 || wlo == 1.0);
  if (!skip) {
    double flo, fhi, interp;
    if (ilo >= M) {
      flo = fw[M - 1];
    } else {
      interp = (wlo - grid.at(ilo - 1)) / (grid.at(ilo) - grid.at(ilo - 1));
      flo = (1.0 - interp) * fw[ilo - 1] + interp * fw[ilo];
    }
    if (ihi >= M) {
      fhi = fw[M - 1];
    } else {
      interp = (whi - grid.at(ihi - 1)) / (grid.at(ihi) - grid.at(ihi - 1));
      fhi = (1.0 - interp) * fw[ihi - 1] + interp * fw[ihi];
    }
    double Plo[LMAX], Phi[LMAX];
    if (ilo != ihi) {
      double ulo = tolab(R, wlo);
      double uhi = tolab(R, grid.at(ilo));
      pn_all<LMAX>(ulo, Plo);
      pn_all<LMAX>(uhi, Phi);
      {
        const double dx = grid.at(ilo) - wlo;
        const double f1 = fw[ilo];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) acc[l] = dx * (flo * Plo[l] + f1 * Phi[l]);
      }
      for (int iw = ilo + 1; iw <= ihi - 1; ++iw) {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) Plo[l] = Phi[l];
        uhi = tolab(R, grid.at(iw));
        pn_all<LMAX>(uhi, Phi);
        const double dx = grid.at(iw) - grid.at(iw - 1);
        const double f0 = fw[iw - 1], f1 = fw[iw];
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
          acc[l] = acc[l] + dx * (f0 * Plo[l] + f1 * Phi[l]);
      }
#pragma unroll
      for (int l = 0; l < LMAX; ++l) Plo[l] = Phi[l];
      uhi = tolab(R, whi);
      pn_all<LMAX>(uhi, Phi);
      {
        const double dx = whi - grid.at(ihi - 1);
        const double f0 = fw[ihi - 1];
#pragma unroll
        for (int l = 0; l < LMAX; ++l)
          acc[l] = acc[l] + dx * (f0 * Plo[l] + fhi * Phi[l]);
      }
    } else {
      pn_all<LMAX>(tolab(R, wlo), Plo);
      pn_all<LMAX>(tolab(R, whi), Phi);
      const double dx = whi - wlo;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) acc[l] = dx * (flo * Plo[l] + fhi * Phi[l]);
    }
#pragma unroll
    for (int l = 0; l < LMAX; ++l) acc[l] = 0.5 * acc[l];
  }
#pragma unroll
  for (int l = 0; l < LMAX; ++l)
    if (l < L) dg[l] = acc[l];
}

// file4 batch: thread per (E_in of the list, group); both rows + blend.
template <int LMAX>
__global__ __launch_bounds__(64) void file4_blend_kernel(int n, const int* list, MuGrid grid,
                                   const double* ein, const int* row_lo,
                                   const double* w_hi, const double* f_tab,
                                   double awr, double Q, int G, int L,
                                   const double* e_bins, int rows_per_ein,
                                   double* out, const int* nuc_of_ein,
                                   const double* nuc_awr, const double* nuc_Q) {
  const long tot = (long)n * G;
  for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < tot;
       k += (long)gridDim.x * blockDim.x) {
    const int j = (int)(k / G), g = (int)(k % G);
    const int i = list ? list[j] : j;
    if (nuc_of_ein) {
      awr = nuc_awr[nuc_of_ein[i]];
      Q = nuc_Q[nuc_of_ein[i]];
    }
    const double* f0 = f_tab + (size_t)row_lo[i] * grid.M;
    double lo[LMAX], hi[LMAX];
    file4_group<LMAX>(grid, f0, ein[i], awr, Q, e_bins[g], e_bins[g + 1], L, lo);
    double* o = out + ((size_t)i * G + g) * L;
    if (rows_per_ein == 2) {
      file4_group<LMAX>(grid, f0 + grid.M, ein[i], awr, Q, e_bins[g],
                        e_bins[g + 1], L, hi);
      const double f = w_hi[i];
      for (int l = 0; l < L; ++l) {
        const double r = lo[l] * (1.0 - f);
        o[l] = r + hi[l] * f;
      }
    } else {
      for (int l = 0; l < L; ++l) o[l] = lo[l];
    }
  }
}

}  // namespace

template <int LMAX>
void launch_file4(int n, const int* list, const MuGrid& grid, const double* ein,
                  const int* row_lo, const double* w_hi, const double* f_tab,
                  double awr, double Q, int G, int L, const double* e_bins,
                  int rows_per_ein, double* out, hipStream_t s, const int* nuc_of_ein,
                  const double* nuc_awr, const double* nuc_Q) {
  const long tot = (long)n * G;
  const int blocks = (int)std::min<long>((tot + 63) / 64, 1 << 16);
  hipLaunchKernelGGL((file4_blend_kernel<LMAX>), dim3(blocks), dim3(64), 0, s, n,
                     list, grid, ein, row_lo, w_hi, f_tab, awr, Q, G, L, e_bins,
                     rows_per_ein, out, nuc_of_ein, nuc_awr, nuc_Q);
}

void launch_file4_any(int n, const int* list, int mu_bins, const double* ein,
                      const int* row_lo, const double* w_hi, const double* f_tab,
                      double awr, double Q, int G, int L, const double* e_bins,
                      int rows_per_ein, double* out, hipStream_t s, const int* nuc_of_ein,
                      const double* nuc_awr, const double* nuc_Q) {
  if (n <= 0) return;
  const MuGrid grid = make_mu_grid(mu_bins);
  if (L <= 4) launch_file4<4>(n, list, grid, ein, row_lo, w_hi, f_tab, awr, Q, G, L, e_bins, rows_per_ein, out, s, nuc_of_ein, nuc_awr, nuc_Q);
  else if (L <= 6) launch_file4<6>(n, list, grid, ein, row_lo, w_hi, f_tab, awr, Q, G, L, e_bins, rows_per_ein, out, s, nuc_of_ein, nuc_awr, nuc_Q);
  else if (L <= 8) launch_file4<8>(n, list, grid, ein, row_lo, w_hi, f_tab, awr, Q, G, L, e_bins, rows_per_ein, out, s, nuc_of_ein, nuc_awr, nuc_Q);
  else launch_file4<11>(n, list, grid, ein, row_lo, w_hi, f_tab, awr, Q, G, L, e_bins, rows_per_ein, out, s, nuc_of_ein, nuc_awr, nuc_Q);
}


}  // namespace ndpp
