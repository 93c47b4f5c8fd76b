// wire_text.hip -- host-only: the reference's formatted outputs.
//   ASCII library sections   print_scatt_ascii scatt.F90:881-997, print_chi_ascii chi.F90:203-244,
//                            ASCII header ndpp.F90:1283-1304
//   ndpp_lib.xml             ndpp.F90:958-1110 (restated; the driver module is not buildable here)
//   to_str(real)             string.F90:408-455, print_ascii_array output.F90:221-253
//   one nuclide's file       ndpp.F90:594-717 (after calc_scatt): tolerance, thinning, group
//                            indices, header + scatter + chi sections
// Formatted Fortran output is reproduced edit descriptor by edit descriptor: I20, 1PE20.12
// (three-digit exponents drop the 'E'), A20, Fw.d / ESw.d of to_str; lines end with '\n'.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"

namespace ndpp {
namespace {

// Fortran E / ES editing writes three-digit exponents without the letter: 1.000000000000+100
void drop_e_for_3_digit_exponent(char* b) {
  char* e = strchr(b, 'E');
  if (!e) return;
  const int ex = atoi(e + 1);
  if (ex > 99 || ex < -99) snprintf(e, 8, "%+04d", ex);
}

// 1PE20.12: d.dddddddddddd E+xx right-justified in 20 columns
void e20_12(std::string& s, double x) {
  char b[64];
  if (std::isnan(x)) { snprintf(b, sizeof b, "%20s", "NaN"); s += b; return; }
  if (std::isinf(x)) { snprintf(b, sizeof b, "%20s", x > 0 ? "Inf" : "-Inf"); s += b; return; }
  snprintf(b, sizeof b, "%.12E", x);
  drop_e_for_3_digit_exponent(b);
  char out[64];
  snprintf(out, sizeof out, "%20s", b);
  s += out;
}
void i20(std::string& s, long v) {
  char b[32];
  snprintf(b, sizeof b, "%20ld", v);
  s += b;
}
void rtrim_newline(std::string& s, size_t line_start) {
  size_t e = s.size();
  while (e > line_start && s[e - 1] == ' ') --e;
  s.resize(e);
  s += '\n';
}

// print_ascii_array (output.F90:221): up to four 1PE20.12 fields per line, lines trimmed
void ascii_array(std::string& s, const double* a, size_t n) {
  for (size_t i = 0; i < n; i += 4) {
    const size_t at = s.size();
    for (size_t k = i; k < n && k < i + 4; ++k) e20_12(s, a[k]);
    rtrim_newline(s, at);
  }
}
void ascii_int_array(std::string& s, const int* a, size_t n) {   // output.F90:255
  for (size_t i = 0; i < n; i += 4) {
    const size_t at = s.size();
    for (size_t k = i; k < n && k < i + 4; ++k) i20(s, a[k]);
    rtrim_newline(s, at);
  }
}

// real_to_str with the default 6 significant digits (string.F90:408): the format is chosen
// by magnitude (incl. the reference's ">= 100 .and. < 10000" branch), then left-adjusted
std::string to_str(double x) {
  const double a = std::fabs(x);
  char b[64];
  const int dec = 6;
  if (a == 0.0) snprintf(b, sizeof b, "%15.1f", x);
  else if (a < 1.0e-1) snprintf(b, sizeof b, "%15.*E", dec - 1, x);
  else if (a < 1.0) snprintf(b, sizeof b, "%15.*f", dec, x);
  else if (a < 10.0) snprintf(b, sizeof b, "%15.*f", dec - 1, x);
  else if (a < 100.0) snprintf(b, sizeof b, "%15.*f", dec - 2, x);
  else if (a < 1000.0) snprintf(b, sizeof b, "%15.*f", dec - 3, x);
  else if (a < 10000.0) snprintf(b, sizeof b, "%15.*f", dec - 4, x);
  else if (a < 100000.0) snprintf(b, sizeof b, "%15.*f", dec - 5, x);
  else snprintf(b, sizeof b, "%15.*E", dec - 1, x);
  drop_e_for_3_digit_exponent(b);
  std::string s(b);
  if (s.size() > 15) s = std::string(15, '*');          // field overflow
  const size_t p = s.find_first_not_of(' ');
  s = p == std::string::npos ? std::string() : s.substr(p);
  while (!s.empty() && s.back() == ' ') s.pop_back();
  return s;
}
std::string to_str(int v) { return std::to_string(v); }

std::string trimmed(const char* c) {
  std::string s(c ? c : "");
  while (!s.empty() && s.back() == ' ') s.pop_back();
  return s;
}

// one matrix section of print_scatt_ascii (:923-939)
void ascii_matrix(std::string& s, const double* mat, int n, int G, int L) {
  for (int iE = 0; iE < n; ++iE) {
    const double* m = mat + (size_t)iE * G * L;
    int gmin = 1, gmax = G;
    while (gmin <= G && !(m[(size_t)(gmin - 1) * L] > 0.0)) ++gmin;
    while (gmax >= 1 && !(m[(size_t)(gmax - 1) * L] > 0.0)) --gmax;
    if (gmin > gmax) {
      i20(s, 0); i20(s, 0); s += '\n';
    } else {
      i20(s, gmin); i20(s, gmax); s += '\n';
      ascii_array(s, m + (size_t)(gmin - 1) * L, (size_t)L * (gmax - gmin + 1));
    }
  }
}

long emit(const std::string& s, long cap, void* buf) {
  if (buf && (long)s.size() <= cap) memcpy(buf, s.data(), s.size());
  return (long)s.size();
}

bool scatt_ok(const ndpp_scatt_result* r, int n_bins, const double* e_bins) {
  return r && e_bins && n_bins == r->G + 1 && r->n_el >= 1 && r->ein_el && r->el_mat;
}

std::string scatt_ascii(const ndpp_scatt_result* r, int n_bins, const double* e_bins) {
  std::string s;
  std::vector<int> gi(n_bins);
  i20(s, r->n_el); s += '\n';
  ascii_array(s, r->ein_el, r->n_el);
  ndpp_group_index(n_bins, e_bins, r->n_el, r->ein_el, gi.data());
  ascii_int_array(s, gi.data(), n_bins);
  ascii_matrix(s, r->el_mat, r->n_el, r->G, r->L);
  if (r->n_inel > 0) {
    i20(s, r->n_inel); s += '\n';
    ascii_array(s, r->ein_inel, r->n_inel);
    ndpp_group_index(n_bins, e_bins, r->n_inel, r->ein_inel, gi.data());
    ascii_int_array(s, gi.data(), n_bins);
    ascii_matrix(s, r->inel_mat, r->n_inel, r->G, r->L);
    if (r->nuinel_mat) ascii_matrix(s, r->nuinel_mat, r->n_inel, r->G, r->L);
  } else {
    i20(s, 0); s += '\n';
  }
  return s;
}

std::string chi_ascii(int G, int n_ein, int n_prec, const double* e_grid, const double* chi_t,
                      const double* chi_p, const double* chi_d) {
  std::string s;
  i20(s, n_ein); i20(s, n_prec); s += '\n';
  ascii_array(s, e_grid, n_ein);
  ascii_array(s, chi_t, (size_t)G * n_ein);
  ascii_array(s, chi_p, (size_t)G * n_ein);
  for (int c = 0; c < n_prec; ++c) ascii_array(s, chi_d + (size_t)c * G * n_ein, (size_t)G * n_ein);
  return s;
}

std::string header_ascii(const char* name, int name_len, double kT, int G, const double* e_bins,
                         int scatt_type, int scatt_order, int nuscatter, int chi_present,
                         int mu_bins, double thin_tol) {
  std::string s;
  // '(A20,1PE20.12,I20,A20)' name, kT, groups: A20 right-justifies a shorter name and keeps the
  // first 20 characters of a longer one; the format stops at the item-less A20
  size_t at = s.size();
  if (name_len >= 20) s.append(name, 20);
  else { s.append(20 - name_len, ' '); s.append(name, name_len); }
  e20_12(s, kT);
  i20(s, G);
  rtrim_newline(s, at);
  ascii_array(s, e_bins, (size_t)G + 1);
  at = s.size();
  i20(s, scatt_type); i20(s, scatt_order); i20(s, nuscatter); i20(s, chi_present);
  rtrim_newline(s, at);
  at = s.size();
  i20(s, mu_bins); e20_12(s, thin_tol);
  rtrim_newline(s, at);
  return s;
}

}  // namespace
}  // namespace ndpp

using namespace ndpp;

extern "C" int ndpp_real_to_str(double x, char* out16) {
  if (!out16) return fail(NDPP_EINVAL, "real_to_str: null buffer");
  const std::string s = to_str(x);
  memcpy(out16, s.c_str(), s.size() + 1);
  return (int)s.size();
}

extern "C" long ndpp_ascii_array(int n, const double* a, long cap, char* buf) {
  if (n < 0 || (n > 0 && !a)) { fail(NDPP_EINVAL, "ascii_array: bad argument"); return -1; }
  std::string s;
  ascii_array(s, a, (size_t)n);
  return emit(s, cap, buf);
}

extern "C" long ndpp_scatt_ascii(const ndpp_scatt_result* r, int n_bins, const double* e_bins, long cap,
                                 char* buf) {
  if (!scatt_ok(r, n_bins, e_bins)) { fail(NDPP_EINVAL, "scatt_ascii: incomplete result"); return -1; }
  return emit(scatt_ascii(r, n_bins, e_bins), cap, buf);
}

extern "C" long ndpp_chi_ascii(int G, int n_ein, int n_prec, const double* e_grid, const double* chi_t,
                               const double* chi_p, const double* chi_d, long cap, char* buf) {
  if (G < 1 || n_ein < 1 || n_prec < 0 || !e_grid || !chi_t || !chi_p || (n_prec > 0 && !chi_d)) {
    fail(NDPP_EINVAL, "chi_ascii: bad argument");
    return -1;
  }
  return emit(chi_ascii(G, n_ein, n_prec, e_grid, chi_t, chi_p, chi_d), cap, buf);
}

extern "C" long ndpp_header_ascii(const char* name, int name_len, double kT, int G, const double* e_bins,
                                  int scatt_type, int scatt_order, int nuscatter, int chi_present,
                                  int mu_bins, double thin_tol, long cap, char* buf) {
  if (!name || name_len < 0 || G < 1 || !e_bins) { fail(NDPP_EINVAL, "header_ascii: bad argument"); return -1; }
  return emit(header_ascii(name, name_len, kT, G, e_bins, scatt_type, scatt_order, nuscatter,
                           chi_present, mu_bins, thin_tol), cap, buf);
}

extern "C" long ndpp_lib_xml_header(const char* directory, int lib_format, int n_listings, int nuscatter,
                                    int chi_present, int scatt_type, int scatt_order, double print_tol,
                                    double thin_tol, int mu_bins, int n_bins, const double* e_bins,
                                    long cap, char* buf) {
  if (n_bins < 2 || !e_bins) { fail(NDPP_EINVAL, "lib_xml_header: bad argument"); return -1; }
  if (lib_format == NDPP_FMT_NONE) return 0;                      // ndpp.F90:977
  const std::string in = "  ";
  std::string s = "<?xml version=\"1.0\"?>\n<ndpp_lib>\n";
  s += in + "<directory> " + trimmed(directory) + "  </directory>\n";
  if (lib_format == NDPP_FMT_ASCII) s += in + "<filetype> ascii </filetype>\n";
  else if (lib_format == NDPP_FMT_BINARY) s += in + "<filetype> binary </filetype>\n";
  else if (lib_format == NDPP_FMT_HDF5) s += in + "<filetype> hdf5 </filetype>\n";
  else if (lib_format == NDPP_FMT_HUMAN) s += in + "<filetype> human </filetype>\n";
  s += in + "<entries> " + to_str(n_listings) + "  </entries>\n";
  s += in + (nuscatter ? "<nuscatter> true </nuscatter>\n" : "<nuscatter> false </nuscatter>\n");
  s += in + (chi_present ? "<chi_present> true </chi_present>\n" : "<chi_present> false </chi_present>\n");
  s += in + "<scatt_type> " + to_str(scatt_type) + " </scatt_type>\n";
  s += in + "<scatt_order> " + to_str(scatt_order) + " </scatt_order>\n";
  s += in + "<print_tol> " + to_str(print_tol) + " </print_tol>\n";
  s += in + "<thin_tol> " + to_str(thin_tol) + " </thin_tol>\n";
  s += in + "<mu_bins> " + to_str(mu_bins) + " </mu_bins>\n";
  s += in + "<energy_bins>\n";
  ascii_array(s, e_bins, (size_t)n_bins);
  s += in + "</energy_bins>\n";
  s += "\n";
  return emit(s, cap, buf);
}

extern "C" long ndpp_lib_xml_nuclide(const char* alias, double awr, const char* name, const char* path,
                                     double kT, int zaid, int metastable, double freegas_cutoff,
                                     int lib_format, long cap, char* buf) {
  if (lib_format == NDPP_FMT_NONE) return 0;
  std::string s = "  <ndpp_table alias=\"" + trimmed(alias) + "\" awr=\"" + to_str(awr) +
                  "\" location=\"1\" name=\"" + trimmed(name) + "\" path=\"" + trimmed(path) +
                  "\" temperature=\"" + to_str(kT) + "\" zaid=\"" + to_str(zaid) + "\" ";
  if (metastable) s += "metastable= \"1\" ";                        // ndpp.F90:1048 (spacing as written)
  s += "freegas_cutoff=\"" + to_str(freegas_cutoff) + "\"/>\n";
  return emit(s, cap, buf);
}

extern "C" long ndpp_lib_xml_closer(int lib_format, long cap, char* buf) {
  if (lib_format == NDPP_FMT_NONE) return 0;
  return emit("</ndpp_lib>\n", cap, buf);
}

// ndpp.F90:611-646: tolerance first (so that thinning sees the zeros), then thinning
extern "C" int ndpp_finish_scatt(const ndpp_output_options* o, ndpp_scatt_result* r, int n_bins,
                                 const double* e_bins, double* thin_report) {
  if (!o || !scatt_ok(r, n_bins, e_bins)) return fail(NDPP_EINVAL, "finish_scatt: bad argument");
  const int L = r->L, G = r->G;
  int rc = ndpp_apply_tol_scatt(L, G, r->n_el, r->el_mat, o->print_tol);
  if (rc == NDPP_OK && r->n_inel > 0) {
    rc = ndpp_apply_tol_scatt(L, G, r->n_inel, r->inel_mat, o->print_tol);
    if (rc == NDPP_OK && o->nuscatter && r->nuinel_mat)
      rc = ndpp_apply_tol_scatt(L, G, r->n_inel, r->nuinel_mat, o->print_tol);
  }
  if (rc != NDPP_OK) return rc;
  double rep[4] = {0, 0, 0, 0};
  if (o->thin_tol > 0.0) {
    std::vector<double> keep(e_bins, e_bins + n_bins);
    rc = ndpp_thin_grid(r->n_el, r->ein_el, L, G, r->el_mat, nullptr, nullptr, n_bins, keep.data(),
                        o->thin_tol, &r->n_el, &rep[0], &rep[1]);
    if (rc == NDPP_OK && r->n_inel > 0)
      rc = ndpp_thin_grid(r->n_inel, r->ein_inel, L, G, r->inel_mat, r->nuinel_mat, nullptr, n_bins,
                          keep.data(), o->thin_tol, &r->n_inel, &rep[2], &rep[3]);
    if (rc != NDPP_OK) return rc;
  }
  if (thin_report) memcpy(thin_report, rep, sizeof rep);
  return NDPP_OK;
}

// ndpp.F90:594 (init_library) + :682-690 (print_scatt) + :717 (print_chi): one table's file
extern "C" long ndpp_nuclide_file(const ndpp_output_options* o, const char* name, int name_len, double kT,
                                  int is_sab, const ndpp_scatt_result* r, int n_bins, const double* e_bins,
                                  int n_chi, int n_prec, const double* e_chi, const double* chi_t,
                                  const double* chi_p, const double* chi_d, long cap, unsigned char* buf) {
  if (!o || !name || name_len < 0 || !scatt_ok(r, n_bins, e_bins)) {
    fail(NDPP_EINVAL, "nuclide_file: bad argument");
    return -1;
  }
  if (o->lib_format != NDPP_FMT_ASCII && o->lib_format != NDPP_FMT_BINARY) {
    fail(NDPP_EINVAL, "nuclide_file: lib_format must be NDPP_FMT_ASCII or NDPP_FMT_BINARY");
    return -1;
  }
  const bool with_chi = o->integrate_chi && !is_sab && n_chi > 0;
  if (with_chi && (!e_chi || !chi_t || !chi_p || n_prec < 0 || (n_prec > 0 && !chi_d))) {
    fail(NDPP_EINVAL, "nuclide_file: chi arrays missing");
    return -1;
  }
  const int G = r->G;
  // the writers print nu-inelastic only when the run asked for it (:682-690)
  ndpp_scatt_result pr = *r;
  if (!o->nuscatter) pr.nuinel_mat = nullptr;
  const int nu_int = (o->nuscatter && !is_sab) ? 1 : 0;             // :1268-1277
  const int chi_int = with_chi ? 1 : 0;
  if (o->lib_format == NDPP_FMT_ASCII) {
    std::string s = header_ascii(name, name_len, kT, G, e_bins, o->scatt_type, o->scatt_order, nu_int,
                                 chi_int, o->mu_bins, o->thin_tol);
    s += scatt_ascii(&pr, n_bins, e_bins);
    if (with_chi) s += chi_ascii(G, n_chi, n_prec, e_chi, chi_t, chi_p, chi_d);
    return emit(s, cap, buf);
  }
  std::vector<unsigned char> out;
  auto section = [&](auto&& call) {                                 // size, then fill
    const long k = call(0L, (unsigned char*)nullptr);
    if (k < 0) return false;
    const size_t at = out.size();
    out.resize(at + (size_t)k);
    call(k, out.data() + at);
    return true;
  };
  if (!section([&](long c, unsigned char* b) {
        return ndpp_header_wire(name, name_len, kT, G, e_bins, o->scatt_type, o->scatt_order, nu_int,
                                chi_int, o->mu_bins, o->thin_tol, c, b); }))
    return -1;
  if (!section([&](long c, unsigned char* b) { return ndpp_scatt_wire(&pr, n_bins, e_bins, c, b); }))
    return -1;
  if (with_chi && !section([&](long c, unsigned char* b) {
        return ndpp_chi_wire(G, n_chi, n_prec, e_chi, chi_t, chi_p, chi_d, c, b); }))
    return -1;
  if (buf && (long)out.size() <= cap) memcpy(buf, out.data(), out.size());
  return (long)out.size();
}
