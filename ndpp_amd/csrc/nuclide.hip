// nuclide.hip -- host-only orchestration of one nuclide: calc_scatt (scatt.F90:33-157)
// behind the C ABI, for hosts that are not the reference's Fortran.
//
//   ScattData%init + convert_distro      -> ndpp_scattdata_shape / ndpp_convert_distro
//   cutoff / inelastic threshold         -> scatt.F90:103-123
//   create_Ein_grid                      -> ndpp_create_ein_grid
//   calc_elastic_grid   (:603-675)       -> bookkeeping here + ndpp_elastic_leg_batch
//   calc_inelastic_grid (:682-778)       -> bookkeeping here + ndpp_elastic_leg_batch /
//                                           ndpp_file6_leg_batch / ndpp_law9_leg_batch
// "Bookkeeping" is scatt_interp_distro (scattdata_header.F90:391-499): threshold and
// top-of-grid tests, the cross-section interpolation, the row search with the
// duplicate-row skip, p_valid, and after the batch call the sigma * p_valid scaling,
// the reaction sum and the nu-scatter yield (scatt.F90:745-762) in the reference's
// order of operations.  fortran/ndpp_hip_mod.f90 holds the same logic for the
// Fortran host; both are checked against the reference's calc_scatt.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <memory>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"

namespace ndpp {
namespace {


// NDPP_HIP_HOST_TIMING=1: where the host side of one nuclide spends its time (stderr)
struct HostClock {
  using clk = std::chrono::steady_clock;
  clk::time_point t0 = clk::now();
  double acc[6] = {0, 0, 0, 0, 0, 0};   // convert, grids + matrices, bookkeeping, batch calls, reaction sum, top rows
  void lap(int k) {
    const clk::time_point t = clk::now();
    acc[k] += std::chrono::duration<double, std::milli>(t - t0).count();
    t0 = t;
  }
  ~HostClock() {
    const char* e = getenv("NDPP_HIP_HOST_TIMING");
    if (e && e[0] == '1')
      fprintf(stderr, "ndpp_scatt_nuclide host ms: convert %.1f  grids+matrices %.1f  bookkeeping %.1f  "
                      "batch calls %.1f  reaction sum / download %.1f  top rows %.1f\n",
              acc[0], acc[1], acc[2], acc[3], acc[4], acc[5]);
  }
};

// The (L, G) blocks a batch call returns for one reaction: page-locked when the driver grants
// it (the device-to-host copy of a many-group reaction -- 54 MB at G = 70 -- then runs at the
// link's rate instead of through the runtime's bounce buffers), ordinary memory otherwise.
// Not zero-filled: every batch call writes all of what it is given.
struct ResultStage {
  double* p = nullptr;
  size_t cap = 0;
  bool pinned = false;
  double* get(size_t n) {
    if (n <= cap) return p;
    release();
    if (n * sizeof(double) >= ((size_t)1 << 20) &&
        hipHostMalloc((void**)&p, n * sizeof(double), hipHostMallocDefault) == hipSuccess) {
      pinned = true;
    } else {
      (void)hipGetLastError();
      p = static_cast<double*>(malloc(n * sizeof(double)));
      pinned = false;
    }
    cap = p ? n : 0;
    return p;
  }
  void release() {
    if (p) { if (pinned) (void)hipHostFree(p); else free(p); }
    p = nullptr;
    cap = 0;
  }
  ~ResultStage() { release(); }
};

// calc_inelastic_grid's reaction sum kept on the device: the matrices of the inelastic grid live
// in HBM while the reactions are integrated, every batch call hands its moments over on the
// device (kernels.h DeviceSink) and one kernel scales and adds them; one copy to the host at the
// end.  (At G = 70 a U-238-like nuclide has 44 reactions x 54 MB: the host-side sum and the
// copies were 0.7 s of its 12 s.)
struct ReactionSum : DeviceSink {
  DevBuf<double> mat, numat;
  DevBuf<char> args;            // one upload per reaction: scale[cap], pv[cap], yield[cap], where[cap]
  std::vector<char> args_h;
  int cap = 0, nb = 0;
  size_t GL = 0, rows = 0;
  bool with_nu = false;
  const double* scale_d() const { return reinterpret_cast<const double*>(args.p); }
  const double* pv_d() const { return scale_d() + cap; }
  const double* yield_d() const { return scale_d() + 2 * (size_t)cap; }
  const int* where_d() const { return reinterpret_cast<const int*>(scale_d() + 3 * (size_t)cap); }
  int init(size_t n_rows, size_t gl, bool nu, int max_nb) {
    rows = n_rows; GL = gl; with_nu = nu; cap = max_nb;
    args_h.resize((size_t)cap * (3 * sizeof(double) + sizeof(int)));
    if (mat.alloc(rows * GL) != hipSuccess || (nu && numat.alloc(rows * GL) != hipSuccess) ||
        args.alloc(args_h.size()) != hipSuccess)
      return fail(NDPP_ENOMEM, "out of device memory for the inelastic matrices");
    if (hipMemset(mat.p, 0, rows * GL * sizeof(double)) != hipSuccess ||
        (nu && hipMemset(numat.p, 0, rows * GL * sizeof(double)) != hipSuccess))
      return fail(NDPP_EDEVICE, "hipMemset failed");
    return NDPP_OK;
  }
  // the rows, cross sections, p_valid and yields of the next batch call
  int stage(int n, const int* w, const double* s, const double* v, const double* y) {
    nb = n;
    double* d = reinterpret_cast<double*>(args_h.data());
    std::copy(s, s + n, d);
    std::copy(v, v + n, d + cap);
    std::copy(y, y + n, d + 2 * (size_t)cap);
    std::copy(w, w + n, reinterpret_cast<int*>(d + 3 * (size_t)cap));
    if (hipMemcpy(args.p, args_h.data(), args_h.size(), hipMemcpyHostToDevice) != hipSuccess)
      return fail(NDPP_EDEVICE, "upload of a reaction's scaling failed");
    return NDPP_OK;
  }
  int consume(const double* out_d, int n, size_t gl) override {
    if (n != nb || gl != GL) return fail(NDPP_EINVAL, "reaction sum: batch of %d x %zu, staged %d x %zu", n, gl, nb, GL);
    launch_reaction_sum(n, GL, out_d, where_d(), scale_d(), pv_d(), yield_d(), mat.p, with_nu ? numat.p : nullptr);
    return hipGetLastError() == hipSuccess ? NDPP_OK : fail(NDPP_EDEVICE, "reaction sum kernel failed to launch");
  }
  int download(double* m, double* nm) {
    if (hipDeviceSynchronize() != hipSuccess ||
        hipMemcpy(m, mat.p, rows * GL * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        (with_nu && nm && hipMemcpy(nm, numat.p, rows * GL * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess))
      return fail(NDPP_EDEVICE, "download of the inelastic matrices failed");
    return NDPP_OK;
  }
};

// body(k0, k1) over [0, n) on a few host threads when the range carries enough work (the
// reaction sum of a many-group structure moves ~250 MB per reaction; the items are independent:
// every incoming energy of a reaction owns its row of the matrices)
template <class F>
void parallel_rows(int n, size_t work_per_row, F body) {
  const size_t work = (size_t)n * work_per_row;
  unsigned nt = std::thread::hardware_concurrency();
  nt = std::min<unsigned>(nt ? nt : 1, 16);
  if (work < ((size_t)1 << 21) || nt < 2 || n < 2 * (int)nt) { body(0, n); return; }
  std::vector<std::thread> th;
  const int per = (n + (int)nt - 1) / (int)nt;
  for (unsigned t = 1; t < nt; ++t) {
    const int k0 = std::min(n, (int)t * per), k1 = std::min(n, k0 + per);
    if (k0 < k1) th.emplace_back([=] { body(k0, k1); });
  }
  body(0, std::min(n, per));
  for (auto& x : th) x.join();
}

// interpolate_tab1_object, interpolation.F90:132-206; rc != 0 where it aborts
int tab1(int n_regions, const int* nbt, const int* intc, int n_pairs, const double* x,
         const double* y, double xv, double* out) {
  if (n_pairs < 1 || !x || !y) return fail(NDPP_EINVAL, "empty TAB1 function");
  if (xv < x[0]) { *out = y[0]; return NDPP_OK; }
  if (xv > x[n_pairs - 1]) { *out = y[n_pairs - 1]; return NDPP_OK; }
  if (n_pairs == 1) { *out = y[0]; return NDPP_OK; }
  const int i = bsearch1_clamped(x, n_pairs, xv);
  int interp = 2;
  if (n_regions == 1) interp = intc[0];
  else if (n_regions > 1)
    for (int j = 0; j < n_regions; ++j)
      if (i < nbt[j]) { interp = intc[j]; break; }
  if (interp == 1) { *out = y[i - 1]; return NDPP_OK; }
  const double x0 = x[i - 1], x1 = x[i], y0 = y[i - 1], y1 = y[i];
  double r;
  switch (interp) {
    case 2: r = (xv - x0) / (x1 - x0); *out = (1 - r) * y0 + r * y1; break;
    case 3: r = (std::log(xv) - std::log(x0)) / (std::log(x1) - std::log(x0));
            *out = (1 - r) * y0 + r * y1; break;
    case 4: r = (xv - x0) / (x1 - x0);
            *out = std::exp((1 - r) * std::log(y0) + r * std::log(y1)); break;
    case 5: r = (std::log(xv) - std::log(x0)) / (std::log(x1) - std::log(x0));
            *out = std::exp((1 - r) * std::log(y0) + r * std::log(y1)); break;
    default: return fail(NDPP_EINVAL, "Unsupported interpolation scheme: %d", interp);
  }
  return NDPP_OK;
}

// one ScattData after init + convert_distro
struct SD {
  bool is_init = false;
  int law = 0;
  bool has_adist = false, has_edist = false;   // associated(this%adist / this%edist)
  const ndpp_ace_rxn* rxn = nullptr;
  const ndpp_ace_edist* edist = nullptr;
  bool in_cm = false;
  int NE = 0;
  std::vector<double> e_grid, eout, pdf, cdf, f;
  std::vector<int> row_ptr, intt;
  // a table only the file-6 integrators read stays on the device, where convert_kernel wrote it and
  // they read it (f is then empty): a continuum's M x sum NP doubles -- 10 to 200 MB for a U-238-class
  // nuclide -- cross to the host and back otherwise
  std::shared_ptr<double> f_dev;
};

// Elastic batches of several nuclides, collected instead of run, so that a library is
// integrated by ONE ndpp_elastic_leg_multi call (small per-nuclide batches leave the GPU
// mostly idle).
struct ElasticDefer {
  std::vector<double> A, kT, cut, Q;        // per collected batch ("nuclide" of the multi call)
  std::vector<double> ein, w, f_tab;
  std::vector<int> nuc, row;
  std::vector<double*> dst;                 // where each incoming energy's (L,G) block goes
  struct Top { double* mat; const double* Ein; int n; };
  std::vector<Top> tops;                    // elastic matrices whose top point is copied last
  int n_rows = 0;
};

// the parts of a Reaction that ScattData%init may rewrite (:160-223)
struct RxnState {
  bool has_angle_dist;
  bool in_cm;
  bool fabricated = false;
  double fab_energy[2];
};

}  // namespace
}  // namespace ndpp

using namespace ndpp;

extern "C" void ndpp_free_scatt_result(ndpp_scatt_result* r) {
  if (!r) return;
  free(r->ein_el); free(r->ein_inel); free(r->el_mat); free(r->inel_mat); free(r->nuinel_mat);
  memset(r, 0, sizeof(*r));
}

static int scatt_nuclide_impl(const ndpp_params* p, const ndpp_ace_nuclide* nuc, int n_bins,
                              const double* e_bins, int nuscatt, ndpp_scatt_result* out,
                              ElasticDefer* defer) {
  if (!p || !nuc || !e_bins || !out) return fail(NDPP_EINVAL, "NULL argument");
  memset(out, 0, sizeof(*out));
  if (n_bins < 2) return fail(NDPP_EINVAL, "need at least one group");
  if (nuc->n_grid < 2 || !nuc->energy || !nuc->elastic)
    return fail(NDPP_EINVAL, "nuclide energy grid / elastic cross section missing");
  if (nuc->n_reaction < 1 || !nuc->reactions) return fail(NDPP_EINVAL, "nuclide has no reactions");
  const int G = n_bins - 1, L = p->order, M = p->mu_bins;
  const double Etop = e_bins[G];
  int rc;
  HostClock hc;

  // ---- init + convert_distro for every reaction and nested distribution (:59-106)
  std::vector<SD> sds;
  std::vector<RxnState> st((size_t)nuc->n_reaction);
  const static int kIso[2] = {1, 1}, kZero[2] = {0, 0};
  const static double kZeroD[2] = {0.0, 0.0};
  for (int ir = 0; ir < nuc->n_reaction; ++ir) {
    const ndpp_ace_rxn& rx = nuc->reactions[ir];
    if (rx.threshold < 1 || rx.threshold > nuc->n_grid)
      return fail(NDPP_EINVAL, "reaction %d: threshold index %d outside the grid", ir, rx.threshold);
    st[ir].has_angle_dist = rx.has_angle_dist != 0;
    st[ir].in_cm = rx.scatter_in_cm != 0;
    const int nsd = std::max(rx.n_edist, 1);
    for (int k = 0; k < nsd; ++k) {
      const ndpp_ace_edist* ed = (rx.n_edist > 0) ? &rx.edist[k] : nullptr;
      ndpp_ace_reaction a;
      memset(&a, 0, sizeof(a));
      a.MT = rx.MT;
      a.law = ed ? ed->law : 0;
      a.threshold_energy = nuc->energy[rx.threshold - 1];
      if (st[ir].has_angle_dist) {
        a.has_angle_dist = 1;
        if (st[ir].fabricated) {   // the isotropic adist an earlier init wrote into rxn%adist
          a.n_adist = 2; a.adist_energy = st[ir].fab_energy; a.adist_type = kIso;
          a.adist_location = kZero; a.n_adist_data = 2; a.adist_data = kZeroD;
        } else {
          a.n_adist = rx.n_adist; a.adist_energy = rx.adist_energy; a.adist_type = rx.adist_type;
          a.adist_location = rx.adist_location; a.n_adist_data = rx.n_adist_data;
          a.adist_data = rx.adist_data;
        }
      }
      if (ed) { a.n_edata = ed->n_data; a.edata = ed->data; }
      int is_init = 0, law = 0, NE = 0, tot = 0;
      rc = ndpp_scattdata_shape(&a, &is_init, &law, &NE, &tot);
      if (rc) return rc;
      SD sd;
      sd.rxn = &rx;
      sd.is_init = is_init != 0;
      if (sd.is_init) {
        // what init leaves behind (:135-223)
        const bool had_adist = st[ir].has_angle_dist;
        if (had_adist) {
          sd.has_adist = true;
          sd.has_edist = ed && ed->law != 3;
        } else if (ed) {
          if (ed->law == 4 || ed->law == 3 || ed->law == 9) {
            sd.has_adist = true;
            sd.has_edist = (ed->law == 9 || ed->law == 4);
          } else {
            sd.has_edist = true;
          }
        } else {
          sd.has_adist = true;
          st[ir].in_cm = true;   // :218
        }
        if (!had_adist && sd.has_adist) {      // rxn%adist now holds the isotropic table
          st[ir].has_angle_dist = true;
          st[ir].fabricated = true;
          st[ir].fab_energy[1] = Etop;
          st[ir].fab_energy[0] = (a.threshold_energy > e_bins[0]) ? a.threshold_energy : e_bins[0];
        }
        sd.law = law;
        sd.edist = sd.has_edist ? ed : nullptr;
        sd.NE = NE;
        sd.e_grid.resize(NE); sd.row_ptr.resize(NE + 1); sd.intt.resize(NE);
        sd.eout.resize(tot); sd.pdf.resize(tot); sd.cdf.resize(tot);
        // integrate_distro's dispatch below: angular-only tables (kind 1) go into host-side batches,
        // law 9 reads column 1 of every row on the host; the rest is file 6
        const bool file6_only = !(sd.has_adist && !sd.has_edist) && !(sd.has_adist && sd.law == 9);
        const char* nk = getenv("NDPP_HIP_NO_DEVICE_TABLES");        // test hook: every table through the host
        if (file6_only && !(nk && nk[0] == '1')) {
          double* fd = nullptr;
          rc = convert_distro_keep(M, &a, G, e_bins, NE, tot, sd.e_grid.data(), sd.row_ptr.data(),
                                   sd.eout.data(), sd.pdf.data(), sd.cdf.data(), sd.intt.data(), &fd);
          sd.f_dev = std::shared_ptr<double>(fd, [](double* q) { free_converted(q); });
        } else {
          sd.f.resize((size_t)tot * M);
          rc = ndpp_convert_distro(M, &a, G, e_bins, NE, tot, sd.e_grid.data(), sd.row_ptr.data(),
                                   sd.eout.data(), sd.pdf.data(), sd.cdf.data(), sd.intt.data(),
                                   sd.f.data());
        }
        if (rc) return rc;
      }
      sds.push_back(std::move(sd));
    }
  }
  // scatter_in_cm as the integrators will see it (the flag lives on the reaction)
  {
    size_t k = 0;
    for (int ir = 0; ir < nuc->n_reaction; ++ir)
      for (int j = 0; j < std::max(nuc->reactions[ir].n_edist, 1); ++j) sds[k++].in_cm = st[ir].in_cm;
  }

  // ---- free-gas cutoff and inelastic threshold (:103-123)
  double cutoff = 0.0, inel_thresh = Etop;
  bool any = false;
  for (const SD& sd : sds) {
    if (!sd.is_init) continue;
    any = true;
    if (sd.rxn->MT == 2) cutoff = nuc->freegas_cutoff;
    else if (nuc->energy[sd.rxn->threshold - 1] < inel_thresh)
      inel_thresh = nuc->energy[sd.rxn->threshold - 1];
  }
  if (!any) return fail(NDPP_EINVAL, "no scattering reaction in this nuclide");

  // ---- incoming grids (:134-135)
  std::vector<ndpp_sd_grid> gs(sds.size());
  for (size_t k = 0; k < sds.size(); ++k) {
    gs[k].is_init = sds[k].is_init;
    gs[k].MT = sds[k].rxn->MT;
    gs[k].Q_value = sds[k].rxn->Q_value;
    gs[k].n = sds[k].NE;
    gs[k].e_grid = sds[k].e_grid.data();
  }
  int n_el = 0, n_in = 0;
  rc = ndpp_create_ein_grid(p, (int)gs.size(), gs.data(), n_bins, e_bins, nuc->n_grid, nuc->energy,
                            nuc->awr, nuc->kT, cutoff, inel_thresh, 0, nullptr, &n_el, 0, nullptr,
                            &n_in);
  if (rc) return rc;
  out->L = L; out->G = G; out->n_el = n_el; out->n_inel = n_in;
  const size_t GL = (size_t)G * L;
  hc.lap(0);
  out->ein_el = (double*)calloc((size_t)n_el, sizeof(double));
  out->el_mat = (double*)calloc((size_t)n_el * GL, sizeof(double));
  if (n_in) {
    out->ein_inel = (double*)calloc((size_t)n_in, sizeof(double));
    // (overwritten as a whole by the download of the device-side reaction sum)
    out->inel_mat = (double*)malloc(std::max<size_t>((size_t)n_in * GL, 1) * sizeof(double));
    if (nuscatt) out->nuinel_mat = (double*)malloc(std::max<size_t>((size_t)n_in * GL, 1) * sizeof(double));
  }
  if (!out->ein_el || !out->el_mat || (n_in && (!out->ein_inel || !out->inel_mat)) ||
      (n_in && nuscatt && !out->nuinel_mat)) {
    ndpp_free_scatt_result(out);
    return fail(NDPP_ENOMEM, "out of host memory for the result matrices");
  }
  rc = ndpp_create_ein_grid(p, (int)gs.size(), gs.data(), n_bins, e_bins, nuc->n_grid, nuc->energy,
                            nuc->awr, nuc->kT, cutoff, inel_thresh, n_el, out->ein_el, &n_el,
                            std::max(n_in, 0), out->ein_inel, &n_in);
  if (rc) { ndpp_free_scatt_result(out); return rc; }

  hc.lap(1);
  // ---- the two grids
  for (int pass = 0; pass < 2; ++pass) {
    const bool elastic = pass == 0;
    const int NEin = elastic ? n_el : n_in;
    const double* Ein = elastic ? out->ein_el : out->ein_inel;
    double* mat = elastic ? out->el_mat : out->inel_mat;
    double* numat = elastic ? nullptr : out->nuinel_mat;
    if (NEin == 0) continue;
    std::vector<double> ein_b(NEin), w_hi(NEin), scale(NEin), pv(NEin), yield_(NEin);
    ResultStage stage;
    std::vector<int> row_lo(NEin), where_(NEin), status(NEin);
    // The reaction sum of a large inelastic grid stays on the device; a small one (the shipped
    // two-group structure: a few hundred KB) is summed here, which costs less than the
    // allocations of the device matrices (0.4 s over the 423 nuclides of the library workload).
    size_t dev_sum_min = (size_t)1 << 20;
    if (const char* e = getenv("NDPP_HIP_DEV_SUM_MIN")) dev_sum_min = (size_t)atoll(e);   // test hook
    const bool dev_sum = !elastic && (size_t)NEin * GL >= dev_sum_min;
    ReactionSum rsum;
    if (dev_sum) {
      rc = rsum.init((size_t)NEin, GL, numat != nullptr, NEin);
      if (rc) { ndpp_free_scatt_result(out); return rc; }
    } else if (!elastic) {
      std::fill(mat, mat + (size_t)NEin * GL, 0.0);
      if (numat) std::fill(numat, numat + (size_t)NEin * GL, 0.0);
    }
    // Host-summed inelastic grid (few groups): the level reactions -- angular distribution only,
    // dozens per nuclide, a few hundred incoming energies each -- are collected and integrated by
    // ONE ndpp_elastic_leg_multi call (a "nuclide" of that call = one level: its Q and its rows)
    // instead of one batch call each; every reaction's moments are kept until all are there and
    // then summed in the reaction order of the loop, as before: same bits.
    // NDPP_HIP_NO_LEVEL_BATCH=1 (test hook): one call per level.
    struct Pending {
      int nb = 0;
      long off = -1;                          // >= 0: first row of this reaction in the level batch
      std::vector<int> where;
      std::vector<double> scale, pv, yld, res;
    };
    std::vector<Pending> pend;
    ElasticDefer lvl;
    const char* nlb = getenv("NDPP_HIP_NO_LEVEL_BATCH");
    const bool host_sum = !elastic && !dev_sum;
    const bool level_batch = host_sum && !(nlb && nlb[0] == '1');
    for (const SD& sd : sds) {
      if (!sd.is_init) continue;
      if ((sd.rxn->MT == 2) != elastic) continue;
      const ndpp_ace_rxn& rx = *sd.rxn;
      const double* sig = elastic ? nuc->elastic : rx.sigma;
      const int nsig = elastic ? nuc->n_grid : rx.n_sigma;
      if (!sig || nsig < 1) { ndpp_free_scatt_result(out); return fail(NDPP_EINVAL, "MT %d has no cross section", rx.MT); }
      if (sd.NE < 2) { ndpp_free_scatt_result(out); return fail(NDPP_EINVAL, "MT %d: fewer than 2 tabulated energies", rx.MT); }
      int nb = 0;
      for (int iE = 0; iE < NEin; ++iE) {
        const double E = Ein[iE];
        if (E > Etop) continue;                                              // top point, copied below
        if ((E <= nuc->energy[rx.threshold - 1] && rx.threshold > 1) || E > Etop) continue;  // :423-431
        double sigS;
        int iEg;
        if (E >= nuc->energy[nuc->n_grid - 1]) {                             // :432-442
          sigS = sig[nsig - 1];
          iEg = sd.NE - 1;
        } else {
          int ni = (E <= nuc->energy[0]) ? 1 : bsearch1_clamped(nuc->energy, nuc->n_grid, E);
          if (nuc->energy[ni - 1] == nuc->energy[ni]) ni = ni + 1;
          const double fr = (E - nuc->energy[ni - 1]) / (nuc->energy[ni] - nuc->energy[ni - 1]);
          ni = ni - rx.threshold + 1;
          if (ni < 1 || ni + 1 > nsig) { ndpp_free_scatt_result(out); return fail(NDPP_EINVAL, "MT %d: cross section shorter than the grid", rx.MT); }
          sigS = (1.0 - fr) * sig[ni - 1] + fr * sig[ni];
          if (sigS <= 0.0) continue;                                         // :466-468
          iEg = (E < sd.e_grid[0]) ? 1 : ((E > sd.e_grid[sd.NE - 1]) ? -1 : bsearch1_clamped(sd.e_grid.data(), sd.NE, E));
          if (iEg < 0) { ndpp_free_scatt_result(out); return fail(NDPP_EINVAL, "MT %d: E_in %g above its tabulated energies", rx.MT, E); }
          if (iEg + 1 <= sd.NE - 1 && sd.e_grid[iEg - 1] >= sd.e_grid[iEg]) iEg = iEg + 1;   // :480-482
        }
        double pval = 1.0;
        if (sd.has_edist && sd.edist) {
          rc = tab1(sd.edist->pv_n_regions, sd.edist->pv_nbt, sd.edist->pv_int, sd.edist->pv_n_pairs,
                    sd.edist->pv_x, sd.edist->pv_y, E, &pval);
          if (rc) { ndpp_free_scatt_result(out); return rc; }
        }
        where_[nb] = iE;
        ein_b[nb] = E;
        row_lo[nb] = iEg - 1;
        w_hi[nb] = (E - sd.e_grid[iEg - 1]) / (sd.e_grid[iEg] - sd.e_grid[iEg - 1]);   // :542
        scale[nb] = sigS;
        pv[nb] = pval;
        ++nb;
      }
      hc.lap(2);
      if (nb == 0) continue;
      // integrate_distro's dispatch (:533-656)
      int kind;
      if (sd.has_adist && !sd.has_edist) kind = 1;
      else if (sd.in_cm) kind = 2;
      else if (sd.has_adist && sd.law == 9) kind = 3;
      else kind = 4;
      if (kind == 1 && elastic && defer) {
        const int k = (int)defer->A.size();
        defer->A.push_back(nuc->awr); defer->kT.push_back(nuc->kT);
        defer->cut.push_back(nuc->freegas_cutoff); defer->Q.push_back(rx.Q_value);
        for (int j = 0; j < nb; ++j) {
          defer->ein.push_back(ein_b[j]); defer->w.push_back(w_hi[j]);
          defer->nuc.push_back(k); defer->row.push_back(defer->n_rows + row_lo[j]);
          defer->dst.push_back(mat + (size_t)where_[j] * GL);
        }
        defer->f_tab.insert(defer->f_tab.end(), sd.f.begin(), sd.f.end());
        defer->n_rows += sd.NE;
        continue;
      }
      double* res = nullptr;
      DeviceSink* sink = nullptr;
      const bool deferred = level_batch && kind == 1;
      if (host_sum) {
        pend.emplace_back();
        Pending& pd = pend.back();
        pd.nb = nb;
        pd.where.assign(where_.begin(), where_.begin() + nb);
        pd.scale.assign(scale.begin(), scale.begin() + nb);
        pd.pv.assign(pv.begin(), pv.begin() + nb);
        if (!deferred) { pd.res.resize((size_t)nb * GL); res = pd.res.data(); }
      } else if (!dev_sum) {
        res = stage.get((size_t)nb * GL);
        if (!res) { ndpp_free_scatt_result(out); return fail(NDPP_ENOMEM, "out of host memory for a reaction's moments"); }
      }
      if (!elastic) {
        for (int k = 0; k < nb; ++k) {
          yield_[k] = (double)rx.multiplicity;
          if (numat && rx.has_mult_E) {
            rc = tab1(rx.mE_n_regions, rx.mE_nbt, rx.mE_int, rx.mE_n_pairs, rx.mE_x, rx.mE_y, ein_b[k], &yield_[k]);
            if (rc) { ndpp_free_scatt_result(out); return rc; }
          }
        }
      }
      if (dev_sum) {
        rc = rsum.stage(nb, where_.data(), scale.data(), pv.data(), yield_.data());
        if (rc) { ndpp_free_scatt_result(out); return rc; }
        sink = &rsum;
      }
      if (host_sum) pend.back().yld.assign(yield_.begin(), yield_.begin() + nb);
      if (deferred) {
        const int k = (int)lvl.A.size();
        pend.back().off = (long)lvl.ein.size();
        lvl.A.push_back(nuc->awr); lvl.kT.push_back(nuc->kT); lvl.cut.push_back(0.0); lvl.Q.push_back(rx.Q_value);
        for (int j = 0; j < nb; ++j) {
          lvl.ein.push_back(ein_b[j]); lvl.w.push_back(w_hi[j]);
          lvl.nuc.push_back(k); lvl.row.push_back(lvl.n_rows + row_lo[j]);
        }
        lvl.f_tab.insert(lvl.f_tab.end(), sd.f.begin(), sd.f.end());
        lvl.n_rows += sd.NE;
        hc.lap(2);
        continue;
      }
      if (kind == 1) {
        rc = elastic_leg_batch_sink(p, nuc->awr, nuc->kT, elastic ? nuc->freegas_cutoff : 0.0,
                                    rx.Q_value, nb, ein_b.data(), row_lo.data(), w_hi.data(), sd.NE,
                                    sd.f.data(), G, e_bins, res, status.data(), sink);
      } else if (kind == 3) {
        std::vector<double> ftab((size_t)sd.NE * M);   // column 1 of every row
        for (int k = 0; k < sd.NE; ++k)
          std::copy(sd.f.begin() + (size_t)sd.row_ptr[k] * M, sd.f.begin() + (size_t)(sd.row_ptr[k] + 1) * M,
                    ftab.begin() + (size_t)k * M);
        rc = law9_leg_batch_sink(p, nb, ein_b.data(), row_lo.data(), w_hi.data(), sd.NE, ftab.data(),
                                 sd.edist->n_data, sd.edist->data, G, e_bins, res, status.data(), sink);
      } else {
        rc = file6_leg_batch_sink(p, nuc->awr, kind == 2 ? 1 : 0, nb, ein_b.data(), row_lo.data(), sd.NE,
                                  sd.e_grid.data(), sd.row_ptr.data(), sd.eout.data(), sd.pdf.data(),
                                  sd.intt.data(), sd.f.data(), G, e_bins, res, status.data(), sink,
                                  sd.f_dev.get());
      }
      hc.lap(3);
      if (rc) { ndpp_free_scatt_result(out); return rc; }
      if (elastic)                                            // assigned, not scaled (:494-497, scatt.F90:660)
        parallel_rows(nb, GL * 2 * sizeof(double), [&](int k0, int k1) {
          for (int k = k0; k < k1; ++k)
            std::copy(res + (size_t)k * GL, res + (size_t)(k + 1) * GL, mat + (size_t)where_[k] * GL);
        });
      hc.lap(4);
    }
    if (host_sum) {
      std::vector<double> lvl_res;
      if (!lvl.ein.empty()) {
        const int n = (int)lvl.ein.size();
        lvl_res.resize((size_t)n * GL);
        std::vector<int> lst(n);
        rc = ndpp_elastic_leg_multi(p, (int)lvl.A.size(), lvl.A.data(), lvl.kT.data(), lvl.cut.data(), lvl.Q.data(),
                                    n, lvl.ein.data(), lvl.nuc.data(), lvl.row.data(), lvl.w.data(), lvl.n_rows,
                                    lvl.f_tab.data(), G, e_bins, lvl_res.data(), lst.data(), nullptr);
        hc.lap(3);
        if (rc) { ndpp_free_scatt_result(out); return rc; }
      }
      for (const Pending& pd : pend)             // the reaction sum, in the order of the loop above
        for (int k = 0; k < pd.nb; ++k) {
          double* dst = mat + (size_t)pd.where[k] * GL;
          double* nudst = numat ? numat + (size_t)pd.where[k] * GL : nullptr;
          const double* src = pd.off >= 0 ? lvl_res.data() + (size_t)(pd.off + k) * GL : pd.res.data() + (size_t)k * GL;
          for (size_t j = 0; j < GL; ++j) {
            const double t = src[j] * pd.scale[k] * pd.pv[k];     // :496
            dst[j] = dst[j] + t;                                  // scatt.F90:753
            if (nudst) nudst[j] = nudst[j] + pd.yld[k] * t;       // :762
          }
        }
      hc.lap(4);
    }
    if (dev_sum) {
      rc = rsum.download(mat, numat);
      if (rc) { ndpp_free_scatt_result(out); return rc; }
      hc.lap(4);
    }
    if (elastic && defer) {                                  // filled and copied by the caller
      defer->tops.push_back({mat, Ein, NEin});
      continue;
    }
    for (int iE = 1; iE < NEin; ++iE)                        // scatt.F90:664-670, :766-774
      if (Ein[iE] > Etop) {
        std::copy(mat + (size_t)(iE - 1) * GL, mat + (size_t)iE * GL, mat + (size_t)iE * GL);
        if (numat) std::copy(numat + (size_t)(iE - 1) * GL, numat + (size_t)iE * GL, numat + (size_t)iE * GL);
      }
    hc.lap(5);
  }
  return NDPP_OK;
}

extern "C" int ndpp_scatt_nuclide(const ndpp_params* p, const ndpp_ace_nuclide* nuc, int n_bins,
                                  const double* e_bins, int nuscatt, ndpp_scatt_result* out) {
  return scatt_nuclide_impl(p, nuc, n_bins, e_bins, nuscatt, out, nullptr);
}

extern "C" int ndpp_scatt_library(const ndpp_params* p, int n_nuclides,
                                  const ndpp_ace_nuclide* nuclides, int n_bins,
                                  const double* e_bins, int nuscatt, ndpp_scatt_result* out) {
  if (n_nuclides < 0 || (n_nuclides > 0 && (!nuclides || !out)))
    return fail(NDPP_EINVAL, "n_nuclides=%d or NULL array", n_nuclides);
  for (int k = 0; k < n_nuclides; ++k) memset(&out[k], 0, sizeof(out[k]));
  ElasticDefer d;
  int rc = NDPP_OK;
  // the deferred elastic grids as ONE mixed batch -- or several, when the tables collected so far
  // approach what one batch call addresses (32-bit byte offsets into f_tab: run_batch_d)
  auto flush = [&]() -> int {
    if (d.ein.empty()) return NDPP_OK;
    const int G = n_bins - 1, n = (int)d.ein.size();
    const size_t GL = (size_t)G * p->order;
    std::vector<double> res((size_t)n * GL);
    std::vector<int> status(n);
    int rc = ndpp_elastic_leg_multi(p, (int)d.A.size(), d.A.data(), d.kT.data(), d.cut.data(),
                                    d.Q.data(), n, d.ein.data(), d.nuc.data(), d.row.data(),
                                    d.w.data(), d.n_rows, d.f_tab.data(), G, e_bins, res.data(),
                                    status.data(), nullptr);
    if (rc == NDPP_OK) {
      for (int i = 0; i < n; ++i) std::copy(res.begin() + (size_t)i * GL, res.begin() + (size_t)(i + 1) * GL, d.dst[i]);
      const double Etop = e_bins[G];
      for (const ElasticDefer::Top& t : d.tops)                // scatt.F90:664-670
        for (int iE = 1; iE < t.n; ++iE)
          if (t.Ein[iE] > Etop)
            std::copy(t.mat + (size_t)(iE - 1) * GL, t.mat + (size_t)iE * GL, t.mat + (size_t)iE * GL);
    }
    d = ElasticDefer();
    return rc;
  };
  constexpr size_t kFlushBytes = (size_t)3 << 30;
  for (int k = 0; k < n_nuclides && rc == NDPP_OK; ++k) {
    rc = scatt_nuclide_impl(p, &nuclides[k], n_bins, e_bins, nuscatt, &out[k], &d);
    if (rc == NDPP_OK && d.f_tab.size() * sizeof(double) > kFlushBytes) rc = flush();
  }
  if (rc == NDPP_OK) rc = flush();
  if (rc != NDPP_OK)
    for (int k = 0; k < n_nuclides; ++k) ndpp_free_scatt_result(&out[k]);
  return rc;
}
