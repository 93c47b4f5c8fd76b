// wire.hip -- host-only: the reference's binary (stream-access) record layout of one
// nuclide's results, so that output written from libndpp_hip results is byte-identical
// to what the reference's writers produce.
//   group_index      ndpp.F90:648-679   (restated; the driver module is not buildable here)
//   print_scatt_bin  scatt.F90:1139-1258
//   print_chi_bin    chi.F90:319-353
//   file header      ndpp.F90:1305-1329
// Stream access = raw little-endian 4-byte integers and 8-byte reals, no record marks.
#include <cstring>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"

namespace ndpp {
namespace {

struct Writer {
  unsigned char* buf;
  long cap, n = 0;
  void put(const void* p, size_t bytes) {
    if (buf && n + (long)bytes <= cap) memcpy(buf + n, p, bytes);
    n += (long)bytes;
  }
  void i32(int v) { put(&v, 4); }
  void f64s(const double* v, size_t k) { put(v, 8 * k); }
};


// one matrix section of print_scatt_bin: per E_in "gmin, gmax, moments of gmin..gmax",
// the range found on the P0 moment (:1181-1198); mat is (L, G, n) in Fortran order
void put_matrix(Writer& w, const double* mat, int n, int G, int L) {
  for (int iE = 0; iE < n; ++iE) {
    const double* m = mat + (size_t)iE * G * L;
    int gmin = 1, gmax = G;
    while (gmin <= G && !(m[(size_t)(gmin - 1) * L] > 0.0)) ++gmin;
    while (gmax >= 1 && !(m[(size_t)(gmax - 1) * L] > 0.0)) --gmax;
    if (gmin > gmax) {
      w.i32(0); w.i32(0);
    } else {
      w.i32(gmin); w.i32(gmax);
      for (int g = gmin; g <= gmax; ++g) w.f64s(m + (size_t)(g - 1) * L, L);
    }
  }
}

}  // namespace
}  // namespace ndpp

using namespace ndpp;

extern "C" int ndpp_group_index(int n_bins, const double* e_bins, int n_ein, const double* ein,
                                int* index) {
  if (n_bins < 2 || n_ein < 1 || !e_bins || !ein || !index) return fail(NDPP_EINVAL, "group_index: bad argument");
  for (int g = 0; g < n_bins; ++g) {
    if (e_bins[g] < ein[0]) index[g] = 1;
    else if (e_bins[g] >= ein[n_ein - 1]) index[g] = n_ein;
    else index[g] = bsearch1_clamped(ein, n_ein, e_bins[g]);
  }
  index[n_bins - 1] = n_ein;  // ndpp.F90:661
  return NDPP_OK;
}

extern "C" long ndpp_scatt_wire(const ndpp_scatt_result* r, int n_bins, const double* e_bins,
                                long cap, unsigned char* buf) {
  if (!r || !e_bins || n_bins != r->G + 1 || r->n_el < 1 || !r->ein_el || !r->el_mat) {
    fail(NDPP_EINVAL, "scatt_wire: incomplete result");
    return -1;
  }
  Writer w{buf, cap};
  std::vector<int> gi(n_bins);
  w.i32(r->n_el);
  w.f64s(r->ein_el, r->n_el);
  ndpp_group_index(n_bins, e_bins, r->n_el, r->ein_el, gi.data());
  w.put(gi.data(), 4 * (size_t)n_bins);
  put_matrix(w, r->el_mat, r->n_el, r->G, r->L);
  if (r->n_inel > 0) {
    w.i32(r->n_inel);
    w.f64s(r->ein_inel, r->n_inel);
    ndpp_group_index(n_bins, e_bins, r->n_inel, r->ein_inel, gi.data());
    w.put(gi.data(), 4 * (size_t)n_bins);
    put_matrix(w, r->inel_mat, r->n_inel, r->G, r->L);
    if (r->nuinel_mat) put_matrix(w, r->nuinel_mat, r->n_inel, r->G, r->L);
  } else {
    w.i32(0);
  }
  return w.n;
}

extern "C" long ndpp_chi_wire(int G, int n_ein, int n_prec, const double* e_grid, const double* chi_t,
                              const double* chi_p, const double* chi_d, long cap, unsigned char* buf) {
  if (G < 1 || n_ein < 1 || n_prec < 0 || !e_grid || !chi_t || !chi_p || (n_prec > 0 && !chi_d)) {
    fail(NDPP_EINVAL, "chi_wire: bad argument");
    return -1;
  }
  Writer w{buf, cap};
  w.i32(n_ein); w.i32(n_prec);
  w.f64s(e_grid, n_ein);
  w.f64s(chi_t, (size_t)G * n_ein);
  w.f64s(chi_p, (size_t)G * n_ein);
  if (n_prec > 0) w.f64s(chi_d, (size_t)G * n_ein * n_prec);
  return w.n;
}

extern "C" long ndpp_header_wire(const char* name, int name_len, double kT, int G, const double* e_bins,
                                 int scatt_type, int scatt_order, int nuscatter, int chi_present,
                                 int mu_bins, double thin_tol, long cap, unsigned char* buf) {
  if (!name || name_len < 0 || G < 1 || !e_bins) { fail(NDPP_EINVAL, "header_wire: bad argument"); return -1; }
  Writer w{buf, cap};
  w.put(name, (size_t)name_len);      // character(len=*) written as passed
  w.put(&kT, 8);
  w.i32(G);
  w.f64s(e_bins, (size_t)G + 1);
  w.i32(scatt_type); w.i32(scatt_order); w.i32(nuscatter); w.i32(chi_present);
  w.i32(mu_bins);
  w.put(&thin_tol, 8);
  return w.n;
}
