// thin.hip -- host-only: thin_grid (thin.F90:17-501), the post-processing pass that drops
// incoming energies whose matrices are reproduced by log-interpolation between their
// neighbours to a relative tolerance.  A sequential greedy scan (each decision depends on
// the last kept point), a few thousand points per nuclide: no kernel.  One routine covers
// thin_grid_one / _two / _three (elastic; inelastic + nu-inelastic; + a 1-D array).
//
// Kept as in the reference: the signed relative error `|test - y| / y` (a negative y always
// passes, :116-118) and `maxerr`, which is compared with relative errors but stores the
// absolute one (:120-122) (sic).
#include <cmath>
#include <cstring>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "kernels.h"

using namespace ndpp;

extern "C" int ndpp_thin_grid(int n, double* x, int L, int G, double* y, double* y2, double* y3,
                              int n_keep, const double* tokeep, double tol, int* n_out,
                              double* compression, double* maxerr_out) {
  if (n < 2 || !x || L < 1 || G < 1 || !y || !n_out || (y3 && !y2) || (n_keep > 0 && !tokeep))
    return fail(NDPP_EINVAL, "thin_grid: bad argument");
  const int rows = L * G, rows2 = rows;
  const std::vector<double> xin(x, x + n), yin(y, y + (size_t)n * rows);
  std::vector<double> yin2, yin3;
  if (y2) yin2.assign(y2, y2 + (size_t)n * rows2);
  if (y3) yin3.assign(y3, y3 + n);
  const int all_ok = rows + (y2 ? rows2 : 0) + (y3 ? 1 : 0);
  double maxerr = 0.0;
  auto keep = [&](int dst, int src) {
    x[dst] = xin[src];
    memcpy(y + (size_t)dst * rows, &yin[(size_t)src * rows], sizeof(double) * rows);
    if (y2) memcpy(y2 + (size_t)dst * rows2, &yin2[(size_t)src * rows2], sizeof(double) * rows2);
    if (y3) y3[dst] = yin3[src];
  };
  // one comparison of the scan (:108-124)
  auto test = [&](double y1, double y2v, double yv, double x_frac, int& remove_it) {
    const double testval = y1 + (y2v - y1) * x_frac;
    double error = std::fabs(testval - yv);
    if (yv != 0.0) error = error / yv;
    if (error <= tol) {
      remove_it += 1;
      if (error > maxerr) maxerr = std::fabs(testval - yv);
    }
  };
  keep(0, 0);
  int num_keep = 1, klo = 0, khi = 2, k = 1;   // 0-based klo, k, khi
  while (khi <= n - 1) {
    int remove_it = 0;
    const double x1 = xin[klo], x2 = xin[khi], xv = xin[k];
    const double x_frac = 1.0 / std::log(x2 / x1) * std::log(xv / x1);
    bool in_keep = false;
    for (int t = 0; t < n_keep; ++t) in_keep = in_keep || (tokeep[t] == xv);
    if (!in_keep) {
      // the Fortran loops i over dim 1 (orders) outside j over dim 2 (groups), y then y2 per
      // element; the order only matters for the sequence of maxerr updates
      for (int i = 0; i < L; ++i)
        for (int j = 0; j < G; ++j) {
          const size_t e = (size_t)j * L + i;
          test(yin[(size_t)klo * rows + e], yin[(size_t)khi * rows + e], yin[(size_t)k * rows + e], x_frac, remove_it);
          if (y2)
            test(yin2[(size_t)klo * rows + e], yin2[(size_t)khi * rows + e], yin2[(size_t)k * rows + e], x_frac, remove_it);
        }
      if (y3) test(yin3[klo], yin3[khi], yin3[k], x_frac, remove_it);
    }
    if (remove_it != all_ok) {
      keep(num_keep, k);
      num_keep += 1;
      klo = k;
    }
    k += 1;
    khi += 1;
  }
  keep(num_keep, n - 1);
  num_keep += 1;
  *n_out = num_keep;
  if (compression) *compression = ((double)n - (double)num_keep) / (double)n;
  if (maxerr_out) *maxerr_out = maxerr;
  return NDPP_OK;
}
