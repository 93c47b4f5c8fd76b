// ein_grid.hip -- host-only: the incoming-energy grids of one nuclide,
// create_Ein_grid (scatt.F90:166-243) = merge (array_merge.F90:13-107) +
// combine_Eins (:252-302) + add_elastic_Eins (:313-419) + add_one_more_point
// (:426-445) + add_inelastic_Eins (:456-535).
//
// No kernel: a few thousand points per nuclide, built once before the batch
// calls.  It lives in the library so that non-Fortran hosts get the
// reference's grids too (the Fortran host layer keeps calling the reference's
// own builders).  Compiled with -ffp-contract=off; exp/log are the host libm's,
// the same functions the flang-built reference calls.
#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"

namespace ndpp {
namespace {

constexpr double kMinEin = 1e-14;  // MIN_EIN, constants.F90:109
typedef std::vector<double> Grid;

// array_merge.F90:13-107: union of two ascending grids; the array whose last
// element is larger is the second operand; a zero taken from either side
// becomes MIN_EIN unless it meets an equal value (then the first operand's
// element is kept as is); duplicates inside one array are kept.
Grid merge(const Grid& a, const Grid& b) {
  const Grid& d1 = (a.back() > b.back()) ? b : a;
  const Grid& d2 = (a.back() > b.back()) ? a : b;
  const size_t n1 = d1.size(), n2 = d2.size();
  Grid out;
  out.reserve(n1 + n2);
  size_t i1 = 0, i2 = 0;
  for (size_t k = 0; k < n1 + n2; ++k) {
    if (i1 < n1 && i2 < n2) {
      if (d1[i1] < d2[i2]) {
        out.push_back(d1[i1] == 0.0 ? kMinEin : d1[i1]);
        ++i1;
      } else if (d1[i1] == d2[i2]) {
        out.push_back(d1[i1]);
        ++i1; ++i2;
      } else {
        out.push_back(d2[i2] == 0.0 ? kMinEin : d2[i2]);
        ++i2;
      }
    } else if (i1 < n1) {
      break;  // :83-88 takes one element and then drops it again (:97-99)
    } else if (i2 < n2) {
      out.push_back(d2[i2]);
      ++i2;
    } else {
      break;
    }
  }
  return out;
}


void add_one_more_point(Grid& g) {
  // `ONE + 1.0E-3`: a default-real literal promoted to double (scatt.F90:438)
  g.push_back(g.back() * (1.0 + (double)1.0e-3f));
}

}  // namespace
}  // namespace ndpp

using namespace ndpp;

extern "C" int ndpp_merge_grids(int na, const double* a, int nb, const double* b, int cap,
                                double* out, int* n_out) {
  if (na < 1 || nb < 1 || !a || !b || !n_out) return fail(NDPP_EINVAL, "merge: empty operand");
  const Grid r = merge(Grid(a, a + na), Grid(b, b + nb));
  *n_out = (int)r.size();
  if (out && cap >= (int)r.size()) std::copy(r.begin(), r.end(), out);
  return NDPP_OK;
}

extern "C" int ndpp_create_ein_grid(const ndpp_params* p, int n_sd, const ndpp_sd_grid* sds,
                                    int n_bins, const double* e_bins, int n_nuc,
                                    const double* nuc_grid, double awr, double kT, double cutoff,
                                    double thresh, int cap_el, double* ein_el, int* n_el,
                                    int cap_inel, double* ein_inel, int* n_inel) {
  if (!p || !e_bins || !nuc_grid || !n_el || !n_inel || (n_sd > 0 && !sds))
    return fail(NDPP_EINVAL, "NULL argument");
  if (n_bins < 2 || n_nuc < 2) return fail(NDPP_EINVAL, "n_bins=%d n_nuc=%d", n_bins, n_nuc);
  const int EXT = p->extend_pts, IEXT = p->inel_extend_pts;
  if (EXT < 1 || IEXT < 1) return fail(NDPP_EINVAL, "extend_pts=%d inel_extend_pts=%d", EXT, IEXT);
  const Grid bins(e_bins, e_bins + n_bins);
  const double Etop = bins.back();

  // ---- create_Ein_grid :184-193: nuclide grid up to the top group edge, + edges
  int iEmax;
  if (Etop >= nuc_grid[n_nuc - 1]) {
    iEmax = n_nuc;
  } else {
    if (Etop < nuc_grid[0])  // the reference's binary_search aborts here
      return fail(NDPP_EINVAL, "top group edge below the nuclide grid");
    iEmax = bsearch1_clamped(nuc_grid, n_nuc, Etop);
  }
  Grid el = merge(Grid(nuc_grid, nuc_grid + iEmax), bins);

  // ---- combine_Eins :252-302
  bool only_el = true;
  {
    Grid acc(1, el[0]);
    for (int k = 0; k < n_sd; ++k) {
      const ndpp_sd_grid& sd = sds[k];
      if (!sd.is_init) continue;
      if (sd.MT != 2) only_el = false;
      if (sd.n < 1 || !sd.e_grid) return fail(NDPP_EINVAL, "ScattData %d has no energy grid", k);
      const double lo = bins.front(), hi = Etop;
      if (lo >= sd.e_grid[sd.n - 1]) continue;
      if (hi <= sd.e_grid[0]) continue;
      const int imax = (hi >= sd.e_grid[sd.n - 1]) ? sd.n : bsearch1_clamped(sd.e_grid, sd.n, hi);
      acc = merge(Grid(sd.e_grid, sd.e_grid + imax), acc);
    }
    el = merge(acc, el);
  }

  // ---- add_elastic_Eins :313-419
  {
    const double a1 = (awr - 1.0) / (awr + 1.0);
    const double alpha = a1 * a1;
    const double lo_shift = 2.0 * kT * (awr + 1.0) / awr;
    Grid pts;
    if (cutoff != 0.0) {
      for (int g = 0; g + 1 < n_bins; ++g) {
        double Ehi = bins[g + 1];
        const double Elo = bins[g];
        if (Ehi <= cutoff) {
          double dElo;
          if (Ehi - lo_shift > Elo) dElo = std::log(Ehi / (Ehi - lo_shift)) / (double)EXT;
          else dElo = std::log(Ehi / 1e-11) / (double)EXT;
          for (int i = -EXT; i <= -1; ++i) {
            const double newE = Ehi * std::exp((double)i * dElo);
            if (newE >= Elo) pts.push_back(newE);
          }
        } else if (Elo < cutoff) {
          Ehi = cutoff;
          const double dElo = std::log(Ehi / (Ehi - lo_shift)) / (double)EXT;
          for (int i = -EXT; i <= -1; ++i) {
            const double newE = Ehi * std::exp((double)i * dElo);
            if (newE > Elo) pts.push_back(newE);
          }
        }
      }
      // merge(new_pts(1:0), ...) of an empty section: the reference would index
      // a(size(a)) = a(0); no group below the cutoff means nothing to add
      if (!pts.empty()) el = merge(pts, el);
      else return fail(NDPP_EINVAL, "free-gas cutoff %g below every group: the reference reads out of bounds", cutoff);
      pts.clear();
    }
    const double dEhi = 7.0 * std::log(1.0 / alpha) / (double)EXT;
    for (int g = 0; g + 1 < n_bins; ++g) {
      if (bins[g] == 0.0) continue;
      const double Ehi = bins[g + 1];
      for (int i = 1; i <= EXT - 1; ++i) {
        const double newE = bins[g] * std::exp((double)i * dEhi);
        if (newE < Ehi) pts.push_back(newE);
        else break;
      }
    }
    if (pts.empty())
      return fail(NDPP_EINVAL, "no down-scatter points to add: the reference reads out of bounds");
    el = merge(pts, el);
  }
  add_one_more_point(el);

  // ---- inelastic grid :204-240
  Grid inel;
  if (!only_el) {
    if (thresh < el.front() || thresh > el.back())
      return fail(NDPP_EINVAL, "inelastic threshold %g outside the elastic grid", thresh);
    const int iT = bsearch1_clamped(el.data(), (int)el.size(), thresh);
    inel.assign(el.begin() + (iT - 1), el.end());
    // add_inelastic_Eins :456-535 (NJOY manual eqs. 239-242)
    for (int k = 0; k < n_sd; ++k) {
      const ndpp_sd_grid& sd = sds[k];
      if (!sd.is_init) continue;
      const double Q = -sd.Q_value;
      if (Q == 0.0) continue;
      for (int g = 2; g <= n_bins - 1; ++g) {
        const double Eg = bins[g - 1];
        const double Ef = (1.0 + awr) / (awr) * Eg;
        const double D = ((awr * awr) * (1.0 + Ef / Q) - 1.0) * (Ef / Q);
        const double Fp = (1.0 + std::sqrt(D)) / (1.0 + Ef / Q);
        const double Fm = (1.0 - std::sqrt(D)) / (1.0 + Ef / Q);
        const double Ecp = ((1.0 + awr) / (awr) * Q) / (1.0 - Fp * Fp / (awr * awr));
        const double Ecm = ((1.0 + awr) / (awr) * Q) / (1.0 - Fm * Fm / (awr * awr));
        double Elo, Ehi;
        if (Ecp > Ecm) { Elo = Ecm; Ehi = Ecp; } else { Elo = Ecp; Ehi = Ecm; }
        if (Elo < thresh) Elo = thresh;
        if (Ehi < thresh) Ehi = thresh;
        if (Elo != Ehi) {
          const double dE = std::log(Ehi / Elo) / (double)IEXT;
          Grid pts;
          for (int i = 1; i <= IEXT - 1; ++i) pts.push_back(Elo * std::exp((double)i * dE));
          if (!pts.empty()) inel = merge(pts, inel);
        }
      }
    }
    if (Etop < inel.front() || Etop > inel.back())
      return fail(NDPP_EINVAL, "top group edge outside the inelastic grid");
    const int iTop = bsearch1_clamped(inel.data(), (int)inel.size(), Etop);
    inel.resize(iTop);
    add_one_more_point(inel);
  }

  *n_el = (int)el.size();
  *n_inel = (int)inel.size();
  if (ein_el && cap_el >= *n_el) std::copy(el.begin(), el.end(), ein_el);
  if (ein_inel && cap_inel >= *n_inel) std::copy(inel.begin(), inel.end(), ein_inel);
  return NDPP_OK;
}
