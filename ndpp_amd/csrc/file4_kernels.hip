// file4_kernels.hip -- two-body CM kinematics moments (integrate_file4_cm_leg).
// Built with -DNDPP_FAST=0 -ffp-contract=off: only + - * / sqrt are involved,
// all IEEE-exact on gfx950, so the results are bit-identical to the reference.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>

#include "kernels.h"
#include "ndpp_math.h"

#if NDPP_FAST
#error "file4_kernels.hip must be compiled with -DNDPP_FAST=0 -ffp-contract=off"
#endif

namespace ndpp {
namespace {

// One group of one incoming energy (scattdata_header.F90:986-1015): the CM cosines of the
// group's bounds clamped to [-1, 1], the 1-based cells of the cosine grid they fall in, and
// whether the reference skips the group (both bounds clamped to the same end).
struct F4Bounds {
  double wlo, whi;
  int ilo, ihi;
  bool skip;
};
struct F4Kin {
  double R, onepawr2, onepR2, inv2REin;
};
__device__ inline F4Kin file4_kin(double Ein, double awr, double Q) {
  F4Kin k;
  k.R = awr * sqrt((1.0 + Q * (awr + 1.0) / (awr * Ein)));
  k.onepawr2 = (1.0 + awr) * (1.0 + awr);
  k.onepR2 = 1.0 + k.R * k.R;
  k.inv2REin = 0.5 / (k.R * Ein);
  return k;
}
__device__ inline F4Bounds file4_bounds(const F4Kin& k, double Ein, double dw, double eg, double eg1) {
  F4Bounds b;
  double wlo = (eg * k.onepawr2 - Ein * k.onepR2) * k.inv2REin;
  if (wlo < -1.0) wlo = -1.0; else if (wlo > 1.0) wlo = 1.0;
  b.ilo = (int)((wlo + 1.0) / dw) + 1;  // 1-based like the reference
  double whi = (eg1 * k.onepawr2 - Ein * k.onepR2) * k.inv2REin;
  if (whi < -1.0) whi = -1.0; else if (whi > 1.0) whi = 1.0;
  b.ihi = (int)((whi + 1.0) / dw) + 1;
  b.wlo = wlo;
  b.whi = whi;
  b.skip = (wlo == whi) && (wlo == -1.0 || wlo == 1.0);
  return b;
}
// the tabulated distribution at a bound (:1019-1034)
__device__ inline double file4_f_at(const MuGrid& grid, const double* fw, double w, int iw) {
  if (iw >= grid.M) return fw[grid.M - 1];
  const double interp = (w - grid.at(iw - 1)) / (grid.at(iw) - grid.at(iw - 1));
  return (1.0 - interp) * fw[iw - 1] + interp * fw[iw];
}

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
  const F4Kin kin = file4_kin(Ein, awr, Q);
  const double R = kin.R;
  double acc[LMAX];
#pragma unroll
  for (int l = 0; l < LMAX; ++l) acc[l] = 0.0;

  const F4Bounds bd = file4_bounds(kin, Ein, dw, eg, eg1);
  const double wlo = bd.wlo, whi = bd.whi;
  const int ilo = bd.ilo, ihi = bd.ihi;
  (void)M;

  const bool skip = bd.skip;
  if (!skip) {
    const double flo = file4_f_at(grid, fw, wlo, ilo);
    const double fhi = file4_f_at(grid, fw, whi, ihi);
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

// The same integral with one wave per incoming energy.  Across the groups of one E_in the
// pieces of the trapezoid sum -- the cosine-grid panels, cut at the groups' bounds -- form one
// contiguous run of about mu_bins pieces (a group's upper bound is the next group's lower one).
// A batch of 64 consecutive pieces: every lane evaluates tolab and the Legendre polynomials at
// the RIGHT end of its piece and takes the left end from the lane below (the reference carries
// P at the previous grid point along in the same way), then forms the piece's term for both
// bracketing rows, which share the cosines.  The reference adds a group's terms in panel order;
// so do the lanes (row, order) here, walking the batch's terms through LDS, starting afresh at
// a group's first piece and blending and storing the two rows at its last: bit for bit the
// thread-per-group kernel above, ~mu_bins / 64 steps instead of mu_bins, independent of how
// the groups divide the range (two groups: one of them owns nearly every panel).
template <int LMAX>
__global__ __launch_bounds__(64) void file4_wave_kernel(int n, const int* list, MuGrid grid,
                                  const double* ein, const int* row_lo,
                                  const double* w_hi, const double* f_tab,
                                  double awr, double Q, int G, int L,
                                  const double* e_bins, int rows_per_ein,
                                  double* out, const int* nuc_of_ein,
                                  const double* nuc_awr, const double* nuc_Q) {
  extern __shared__ double f4_smem[];
  constexpr int NC = 2 * LMAX;   // channels: (row, order)
  constexpr int TS = NC + 1;     // odd stride in doubles: the lanes' term rows spread over the banks
  double* term = f4_smem;                                   // [64][TS]
  int* flag = reinterpret_cast<int*>(term + 64 * TS);       // [64]  group << 2 | last << 1 | first
  int* start = flag + 64;                                   // [G + 1] first piece of each group
  const int t = threadIdx.x;
  const int M = grid.M;
  const double dw = grid.dmu_fgk;
  const int c = t < NC ? t : 0;             // this lane's channel in the ordered sum
  const int cl = c < LMAX ? c : c - LMAX;   // its order
  for (int j = blockIdx.x; j < n; j += gridDim.x) {
    const int i = list ? list[j] : j;
    if (nuc_of_ein) {
      awr = nuc_awr[nuc_of_ein[i]];
      Q = nuc_Q[nuc_of_ein[i]];
    }
    const double Ein = ein[i];
    const F4Kin kin = file4_kin(Ein, awr, Q);
    const double* f0 = f_tab + (size_t)row_lo[i] * M;
    const double* f1 = rows_per_ein == 2 ? f0 + M : f0;
    const double fb = rows_per_ein == 2 ? w_hi[i] : 0.0;

    // pieces per group and their exclusive prefix; a skipped group's moments are zero
    int running = 0;
    for (int base = 0; base < G; base += 64) {
      const int g = base + t;
      int cnt = 0;
      if (g < G) {
        const F4Bounds b = file4_bounds(kin, Ein, dw, e_bins[g], e_bins[g + 1]);
        if (b.skip) {
          double* o = out + ((size_t)i * G + g) * L;
          double z = 0.0;
          if (rows_per_ein == 2) { const double r = 0.0 * (1.0 - fb); z = r + 0.0 * fb; }
          for (int l = 0; l < L; ++l) o[l] = z;
        } else {
          // first partial panel, the whole panels between, last partial panel (:1040-1068);
          // one piece when both bounds share a cell (:1070-1076)
          cnt = (b.ilo == b.ihi) ? 1 : max(2, b.ihi - b.ilo + 1);
        }
      }
      int incl = cnt;
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (t >= d) incl += v;
      }
      if (g < G) start[g] = running + incl - cnt;
      running += __shfl(incl, 63);
    }
    if (t == 0) start[G] = running;
    __syncthreads();
    const int N = running;

    double carry_x = __builtin_nan("");   // right end of the previous batch's last piece, and P there
    double carry[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) carry[l] = 0.0;
    double acc = 0.0;
    for (int p0 = 0; p0 < N; p0 += 64) {
      const int pidx = p0 + t;
      const bool live = pidx < N;
      double xl = 0.0, xr = 0.0, dx = 0.0, fL0 = 0.0, fR0 = 0.0, fL1 = 0.0, fR1 = 0.0;
      double Pr[LMAX], Pl[LMAX];
      int fl = 0;
      if (live) {
        int lo = 0, hi = G;   // start[lo] <= pidx < start[hi]
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (start[mid] <= pidx) lo = mid; else hi = mid;
        }
        const int g = lo, jj = pidx - start[g];
        const F4Bounds b = file4_bounds(kin, Ein, dw, e_bins[g], e_bins[g + 1]);
        const int cnt = (b.ilo == b.ihi) ? 1 : max(2, b.ihi - b.ilo + 1);
        const bool first = (jj == 0), last = (jj == cnt - 1);
        fl = (g << 2) | (last ? 2 : 0) | (first ? 1 : 0);
        if (first) {
          xl = b.wlo;
          fL0 = file4_f_at(grid, f0, b.wlo, b.ilo);
          if (rows_per_ein == 2) fL1 = file4_f_at(grid, f1, b.wlo, b.ilo);
        } else {
          const int iw = last ? b.ihi - 1 : b.ilo + jj - 1;   // 0-based grid point at the left end
          xl = grid.at(iw);
          fL0 = f0[iw];
          fL1 = f1[iw];
        }
        if (last) {
          xr = b.whi;
          fR0 = file4_f_at(grid, f0, b.whi, b.ihi);
          if (rows_per_ein == 2) fR1 = file4_f_at(grid, f1, b.whi, b.ihi);
        } else {
          const int iw = b.ilo + jj;
          xr = grid.at(iw);
          fR0 = f0[iw];
          fR1 = f1[iw];
        }
        dx = xr - xl;
        pn_all<LMAX>(tolab(kin.R, xr), Pr);
      }
      // the left end: P of the lane below (of the previous batch for lane 0), if that is the
      // same cosine -- it always is inside a run of pieces; else evaluated here
      double xprev = __shfl_up(xr, 1);
      if (t == 0) xprev = carry_x;
#pragma unroll
      for (int l = 0; l < LMAX; ++l) {
        Pl[l] = __shfl_up(Pr[l], 1);
        if (t == 0) Pl[l] = carry[l];
      }
      if (live && f64_bits(xprev) != f64_bits(xl)) pn_all<LMAX>(tolab(kin.R, xl), Pl);
      carry_x = __shfl(xr, 63);
#pragma unroll
      for (int l = 0; l < LMAX; ++l) carry[l] = __shfl(Pr[l], 63);
      if (live) {
        double* tr = term + t * TS;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) tr[l] = dx * (fL0 * Pl[l] + fR0 * Pr[l]);
        if (rows_per_ein == 2) {
#pragma unroll
          for (int l = 0; l < LMAX; ++l) tr[LMAX + l] = dx * (fL1 * Pl[l] + fR1 * Pr[l]);
        }
        flag[t] = fl;
      }
      const bool edges = __ballot(live && (fl & 3)) != 0;
      __syncthreads();
      // the ordered sum, lanes (row, order)
      const int nb = min(64, N - p0);
      if (!edges) {
        for (int k = 0; k < nb; ++k) acc = acc + term[k * TS + c];
      } else {
        for (int k = 0; k < nb; ++k) {
          const int fk = flag[k];
          const double v = term[k * TS + c];
          acc = (fk & 1) ? v : acc + v;
          if (fk & 2) {
            const double mine = 0.5 * acc;
            const double other = __shfl(mine, (c + LMAX) & 63);
            if (t < LMAX && t < L) {
              double* o = out + ((size_t)i * G + (fk >> 2)) * L;
              if (rows_per_ein == 2) {
                const double r = mine * (1.0 - fb);
                o[cl] = r + other * fb;
              } else {
                o[cl] = mine;
              }
            }
          }
        }
      }
      __syncthreads();
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
  // one wave per incoming energy; the thread-per-(E_in, group) kernel remains for group
  // structures whose prefix array does not fit the LDS (and as the cross-check of the tests:
  // NDPP_HIP_FILE4_PER_GROUP=1)
  const size_t lds = sizeof(double) * 64 * (2 * LMAX + 1) + sizeof(int) * (64 + (size_t)G + 1);
  const char* pg = getenv("NDPP_HIP_FILE4_PER_GROUP");
  if (lds <= 48 * 1024 && !(pg && pg[0] == '1')) {
    hipLaunchKernelGGL((file4_wave_kernel<LMAX>), dim3(std::min(n, 1 << 16)), dim3(64), lds, s, n,
                       list, grid, ein, row_lo, w_hi, f_tab, awr, Q, G, L, e_bins,
                       rows_per_ein, out, nuc_of_ein, nuc_awr, nuc_Q);
    return;
  }
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
