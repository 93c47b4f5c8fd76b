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

// sab_egrid, sab.F90:460-568: the incoming grid of a thermal table = its inelastic (and
// elastic) energies + the group edges + an energy wherever a discrete outgoing energy
// crosses a group edge between two table energies (:493-537, incl. the exchange at
// :504-508 that is not a swap), cut at the table's top energy, then EXTEND_PTS log-spaced
// points inside every interval (unless SAB_EPTS_PER_BIN == 0).
extern "C" int ndpp_sab_egrid(const ndpp_params* p, const ndpp_sab_flat* t, int n_bins,
                              const double* e_bins, int cap, double* ein, int* n_out) {
  if (!p || !t || !e_bins || !n_out || n_bins < 2) return fail(NDPP_EINVAL, "sab_egrid: bad argument");
  const int NEi = t->n_inelastic_e_in, NEo = t->n_inelastic_e_out;
  if (NEi < 2 || !t->inelastic_e_in) return fail(NDPP_EINVAL, "sab_egrid: need >= 2 inelastic E_in");
  const Grid bins(e_bins, e_bins + n_bins);
  const Grid ei(t->inelastic_e_in, t->inelastic_e_in + NEi);
  Grid g;
  double max_ein;
  if (t->n_elastic_e_in > 0 && t->elastic_e_in) {       // allocated(sab%elastic_e_in)
    const Grid ee(t->elastic_e_in, t->elastic_e_in + t->n_elastic_e_in);
    g = merge(merge(ei, ee), bins);
    max_ein = std::max(ei.back(), ee.back());
  } else {
    g = merge(ei, bins);
    max_ein = ei.back();
  }
  if (t->secondary_mode != 2) {
    if (!t->inelastic_e_out || NEo < 1) return fail(NDPP_EINVAL, "sab_egrid: discrete E_out missing");
    for (int i = 0; i + 1 < NEi; ++i) {
      const double Ei1 = ei[i], Ei2 = ei[i + 1];
      for (int j = 0; j < NEo; ++j) {
        const double Eo1 = t->inelastic_e_out[(size_t)i * NEo + j];
        const double Eo2 = t->inelastic_e_out[(size_t)(i + 1) * NEo + j];
        if (Eo1 < bins.front() || Eo1 > bins.back() || Eo2 < bins.front() || Eo2 > bins.back())
          return fail(NDPP_EINVAL, "sab_egrid: outgoing energy outside the group structure");
        int g1 = bsearch1_clamped(bins.data(), n_bins, Eo1);
        int g2 = bsearch1_clamped(bins.data(), n_bins, Eo2);
        if (Eo2 < Eo1) g2 = g1;                           // (sic) :504-508
        Grid pts;
        for (int gg = g1 + 1; gg <= g2; ++gg)
          pts.push_back((bins[gg - 1] - Eo1) / (Eo2 - Eo1) * (Ei2 - Ei1) + Ei1);
        if (!pts.empty()) g = merge(pts, g);
      }
    }
  }
  if (max_ein < g.front() || max_ein > g.back()) return fail(NDPP_EINVAL, "sab_egrid: top energy outside the grid");
  const int i_max = bsearch1_clamped(g.data(), (int)g.size(), max_ein);
  Grid out;
  if (p->sab_epts_per_bin == 0) {
    out.assign(g.begin(), g.begin() + i_max);
  } else {
    const int EXT = p->extend_pts;
    out.resize((size_t)(i_max - 1) * EXT + i_max);
    size_t j = 0;
    for (int iE = 0; iE < i_max - 1; ++iE) {
      const double dE = std::log(g[iE + 1] / g[iE]) / (double)(EXT + 1);
      out[j++] = g[iE];
      for (int k = 0; k < EXT; ++k) { out[j] = out[j - 1] * std::exp(dE); ++j; }
    }
    out.back() = g[i_max - 1];
  }
  *n_out = (int)out.size();
  if (ein && cap >= *n_out) std::copy(out.begin(), out.end(), ein);
  return NDPP_OK;
}

// union incoming grid of calc_chi (chi.F90:97-113): the spectra's own incoming energies
// (edist%data(2+2NR+1 : 2+2NR+NE), chidata_header.F90:98-104) merged prompt first, then delayed
extern "C" int ndpp_chi_egrid(int n_prompt, const ndpp_chi_spectrum* prompt, int n_delay,
                              const ndpp_chi_spectrum* delay, int cap, double* e_grid, int* n_out) {
  if (n_prompt < 1 || !prompt || n_delay < 0 || (n_delay > 0 && !delay) || !n_out)
    return fail(NDPP_EINVAL, "chi_egrid: bad argument");
  Grid g;
  for (int k = 0; k < n_prompt + n_delay; ++k) {
    const ndpp_chi_spectrum& s = (k < n_prompt) ? prompt[k] : delay[k - n_prompt];
    if (!s.data || s.n_data < 3) return fail(NDPP_EINVAL, "chi_egrid: spectrum %d has no data", k);
    const int NR = (int)s.data[0];
    if (NR < 0 || 2 + 2 * NR > s.n_data) return fail(NDPP_EINVAL, "chi_egrid: spectrum %d: bad NR", k);
    const int NE = (int)s.data[1 + 2 * NR];
    if (NE < 1 || 2 + 2 * NR + NE > s.n_data) return fail(NDPP_EINVAL, "chi_egrid: spectrum %d: bad NE", k);
    const Grid e(s.data + 2 + 2 * NR, s.data + 2 + 2 * NR + NE);
    g = g.empty() ? e : merge(g, e);
  }
  *n_out = (int)g.size();
  if (e_grid && cap >= *n_out) std::copy(g.begin(), g.end(), e_grid);
  return NDPP_OK;
}
