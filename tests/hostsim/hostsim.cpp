// tests/hostsim/hostsim.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Drives the product's NDPP_HD stage functions (ndpp_amd/csrc/fg_pipeline.h)
// sequentially on the CPU, level by level, exactly in the order the gfx950
// kernels of fg_kernels.hip launch them.  It exists because the build
// container has no GPU: it lets the CPU test-suite check the *algorithm* the
// kernels implement (joint-order union trees, direct-mapped sibling stack,
// breadth-first outer levels, bottom-up reduction) against the oracle.
// It is never built into, nor loaded by, the product library.
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../../ndpp_amd/csrc/fg_pipeline.h"
#include "../../ndpp_amd/csrc/legendre_int.h"
#include "../../include/ndpp_hip.h"

using namespace ndpp;

template <int R, int LMAX>
static void run_mu_level(const FgBatch& B, int level, int base) {
  const int nt = B.n_mu_tasks(level);
  unsigned long long nk = 0, nv = 0, ni = 0;
  const bool split = B.split_level(level);
  const int nwork = split ? nt * kSplit : nt;
#pragma omp parallel for schedule(dynamic, 4) reduction(+ : nk, nv, ni)
  for (int t = 0; t < nwork; ++t) {
    MuLane<R, LMAX> s;
    HostMuStack<R> st{};
    if (split) mu_init_split<R, LMAX>(B, level, base, t, s);
    else mu_init<R, LMAX>(B, level, base, t, s);
    if (s.mask == 0) continue;
    const bool counts = !split || s.path_bits == 0;   // (one item per integral in the statistics)
    mu_tot_zero(s);
    const PnConsts pk = make_pn_consts();
    // (kPath = the split walk, as the device instantiates it: its segments go to the slots of B.seg)
    if (split) while (mu_step<R, LMAX, HostMuStack<R>, true>(B, s, st, pk)) {}
    else while (mu_step<R, LMAX, HostMuStack<R>, false>(B, s, st, pk)) {}
    mu_finish(B, s, split);
    nk += 2ull * s.visits + 3;
    nv += s.visits;
    ni += counts ? 1 : 0;
  }
  if (split)
    for (int t = 0; t < nt; ++t) fg_mu_combine_task(B, level, base, t);
  B.stats[kStatKEvals] += nk;
  B.stats[kStatMuVisits] += nv;
  B.stats[kStatMuIntegrals] += ni;
}

#if NDPP_FAST
template <int R, int LMAX>
static void run_gauss_level(const FgBatch& B, int level, int base) {
  const int nt = B.n_tasks(level);
  unsigned long long nk = 0;
#pragma omp parallel for schedule(dynamic, 16) reduction(+ : nk)
  for (int t = 0; t < nt; ++t) nk += (unsigned long long)mu_gauss_task<R, LMAX>(B, level, base, t);
  B.stats[kStatKEvals] += nk;
}
#endif

// The inner walk of one level with the lane type the device pipeline launches for the batch's
// shape (fg_device.h launch_mu_any), after the Gauss-rule stage where that is on (B.t_gl).
static void run_mu(FgBatch& B, int level, int base) {
  const int R = B.R, L = B.L;
#if NDPP_FAST
  if (B.t_gl) {
    if (R == 2) {
      if (L <= 4) run_gauss_level<2, 4>(B, level, base);
      else if (L <= 6) run_gauss_level<2, 6>(B, level, base);
      else run_gauss_level<2, 8>(B, level, base);
    } else {
      if (L <= 4) run_gauss_level<1, 4>(B, level, base);
      else if (L <= 6) run_gauss_level<1, 6>(B, level, base);
      else if (L <= 8) run_gauss_level<1, 8>(B, level, base);
      else run_gauss_level<1, 11>(B, level, base);
    }
  }
#endif
  // HOSTSIM_SPLIT=1: every level in split mode; the segment slots of the level's integrals start at zero
  std::vector<double> segbuf;
  if (B.split_below > 0) {
    segbuf.assign((size_t)B.n_tasks(level) * kSplit * R * L, 0.0);
    B.seg = segbuf.data();
  }
  if (R == 2) {
    if (L <= 4) run_mu_level<2, 4>(B, level, base);
    else if (L <= 6) run_mu_level<2, 6>(B, level, base);
    else run_mu_level<2, 8>(B, level, base);
  } else {
    if (L <= 4) run_mu_level<1, 4>(B, level, base);
    else if (L <= 6) run_mu_level<1, 6>(B, level, base);
    else if (L <= 8) run_mu_level<1, 8>(B, level, base);
    else run_mu_level<1, 11>(B, level, base);
  }
  B.seg = nullptr;
}

// n_jobs incoming energies with R rows each;
// row[n_jobs*R]; raw [n_jobs*R][G][L].
extern "C" int hostsim_freegas_jobs(const ndpp_params* p, double A, double kT,
                                     int n_jobs, int R, const double* ein,
                                     const int* row, int n_rows,
                                     const double* f_tab, int G,
                                     const double* e_bins, int ncap,
                                     double* raw, unsigned long long* stats_out,
                                     int* lvl_cnt_out) {
  FgBatch B;
  if (R < 1 || R > 2 || (R == 2 && p->order > 8)) return NDPP_EINVAL;
  B.n_jobs = n_jobs; B.R = R; B.G = G; B.L = p->order; B.M = p->mu_bins;
  B.A = A; B.kT = kT;
  B.job_ein = ein; B.job_row = row; B.f_tab = f_tab; B.e_bins = e_bins;
  (void)n_rows;
  B.sab_threshold = p->sab_threshold; B.brent_thresh = p->brent_mu_thresh;
  B.mu_tol = p->adaptive_mu_tol; B.eout_tol = p->adaptive_eout_tol;
  B.mu_its = p->adaptive_mu_its; B.eout_its = p->adaptive_eout_its;
  B.grid = make_mu_grid(B.M);
  B.ncap = ncap;
  const int L = B.L;
  std::vector<double> na(ncap), nb(ncap), nF((size_t)5 * R * L * ncap), nS((size_t)R * L * ncap);
  std::vector<int> info((size_t)4 * ncap);
  B.node_a = na.data(); B.node_b = nb.data(); B.node_F = nF.data();
  B.node_S = nS.data(); B.node_info = info.data();
  B.tcap = 5 * B.n_trees() > 2 * ncap ? 5 * B.n_trees() : 2 * ncap;
  std::vector<double> t1(B.tcap), t2(B.tcap), t3((size_t)3 * R * B.tcap);
  B.t_mulo = t1.data(); B.t_muhi = t2.data(); B.t_X = t3.data();
  // HOSTSIM_GAUSS=1 (product arithmetic only): the Gauss-rule stage as the device pipeline runs it
  // on tables that are linear in mu
  std::vector<unsigned char> tgl(B.tcap, 0);
  if (NDPP_FAST && getenv("HOSTSIM_GAUSS") && getenv("HOSTSIM_GAUSS")[0] == '1') {
    B.t_gl = tgl.data();
    // (the knobs of ndpp_hip.hip's NDPP_HIP_GAUSS_*)
    if (const char* e = getenv("HOSTSIM_GAUSS_RATIO")) B.gl_ratio = atof(e);
    if (const char* e = getenv("HOSTSIM_GAUSS_NEAR")) B.gl_near = atoi(e) != 0;
    if (const char* e = getenv("HOSTSIM_GAUSS_AMIN")) B.gl_amin = atof(e);
    if (const char* e = getenv("HOSTSIM_GAUSS_PANELS")) B.gl_panels = atoi(e);
    if (const char* e = getenv("HOSTSIM_GAUSS_GRADED")) B.gl_graded = atoi(e);
    if (const char* e = getenv("HOSTSIM_GAUSS_DEPTH")) B.gl_cert_depth = atoi(e);
    if (const char* e = getenv("HOSTSIM_GAUSS_DEPTH_NEAR")) B.gl_cert_depth_near = atoi(e);
  }
  std::vector<int> cnt(kMaxLevels + 2, 0);
  int next = 0, ovf = 0;
  unsigned long long stats[kNumStats] = {0};
  B.lvl_cnt = cnt.data(); B.next_task = &next; B.overflow = &ovf; B.stats = stats;
  B.raw = raw;
  if (B.n_trees() > ncap) return NDPP_EOVERFLOW;
  // HOSTSIM_SPLIT=1: every level in split mode (kSplit lanes per inner integral)
  if (getenv("HOSTSIM_SPLIT") && getenv("HOSTSIM_SPLIT")[0] == '1') B.split_below = B.tcap;   // (run_mu)

  cnt[0] = B.n_trees();
  for (int c = 0; c < n_jobs; ++c)
    for (int g = 0; g < G; ++g) fg_setup_group(B, c, g);

  int nlev = 0;
  for (int level = 0; level <= B.eout_its; ++level) {
    if (cnt[level] == 0) break;
    nlev = level + 1;
    const int base = B.lvl_off(level);
    const int nt = B.n_tasks(level);
#pragma omp parallel for schedule(dynamic, 16)
    for (int t = 0; t < nt; ++t) fg_prep_task(B, level, base, t);
    run_mu(B, level, base);
    for (int i = 0; i < cnt[level]; ++i)
      fg_node_process<HostAtomics>(B, level, base, i);
    stats[kStatEoutNodes] += cnt[level];
    if (ovf) return NDPP_EOVERFLOW;
  }
  for (int level = nlev - 1; level >= 0; --level) {
    const int base = B.lvl_off(level);
    for (int i = 0; i < cnt[level]; ++i) fg_reduce_node(B, base, i);
  }
  for (int c = 0; c < n_jobs * R; ++c) fg_assemble_call(B, c);
  if (stats_out) memcpy(stats_out, stats, 4 * sizeof(unsigned long long));
  if (lvl_cnt_out) memcpy(lvl_cnt_out, cnt.data(), sizeof(int) * (kMaxLevels + 1));
  return 0;
}

// exp_glibc (ndpp_math.h): the exp of the strict arithmetic on the device, compiled for the
// host so that it can be checked against the host's libm without a GPU (strict variant only).
// Returns the number of arguments (of n) on which the two differ in any bit.
#if !NDPP_FAST
extern "C" double hostsim_exp_glibc(double x) { return exp_glibc(x); }
extern "C" long hostsim_exp_glibc_mismatches(const double* x, long n) {
  long bad = 0;
  for (long i = 0; i < n; ++i) {
    const double a = exp_glibc(x[i]), b = exp(x[i]);
    if (__builtin_memcmp(&a, &b, 8) != 0) ++bad;
  }
  return bad;
}
#endif

// legendre_int.h: the product's integrals of (linear f) x P_l over one panel, compiled for the
// host (both arithmetic variants) so that they can be checked against the oracle's restatement
// of the reference's closed forms without a GPU.  n <= 11.
extern "C" void hostsim_tablelin(int n, double xlo, double xhi, double flo, double fhi, double* out) {
  double v[11];
  tablelin<11>(xlo, xhi, flo, fhi, v);
  for (int l = 0; l < n && l < 11; ++l) out[l] = v[l];
}
// a walk over consecutive panels x[0..np-1] (the way the kernels use it): sum of the panel integrals
extern "C" void hostsim_linear_legendre_walk(int n, int np, const double* x, const double* f, double* out) {
  LinearLegendre<11> w;
  double pan[11], acc[11] = {0};
  w.start(x[0], f[0]);
  for (int k = 1; k < np; ++k) {
    w.panel(x[k], f[k], pan);
    for (int l = 0; l < 11; ++l) acc[l] += pan[l];
  }
  for (int l = 0; l < n && l < 11; ++l) out[l] = acc[l];
}

// the same walk in the running-sum forms the kernels use: mode 1 = panel_add, mode 2 = pairs of
// panels with the caller's reciprocal step (panel2_add, file 6 CM), an odd last panel by panel_add
extern "C" void hostsim_linear_legendre_walk_add(int n, int np, const double* x, const double* f, int mode,
                                                 double* out) {
  LinearLegendre<11> w;
  double acc[11] = {0};
  w.start(x[0], f[0]);
  int k = 1;
  if (mode == 2) {
    const double rh = 1.0 / ((x[np - 1] - x[0]) / (double)(np - 1));
    for (; k + 1 < np; k += 2) w.panel2_add(x[k], f[k], x[k + 1], f[k + 1], rh, acc);
  }
  for (; k < np; ++k) w.panel_add(x[k], f[k], acc);
  for (int l = 0; l < n && l < 11; ++l) out[l] = acc[l];
}

// ---- calibration of the product arithmetic's decision guard (tools/guard_calibrate.py) --------
// One node [a, b] of an inner integral in THIS library's arithmetic: every channel's S2 - S
// (the quantity adaptiveSimpsonsAux_mu tests, freegas.F90:544), the row scales
// w (|K(a)| + 4 |K(d)| + 2 |K(c)| + 4 |K(e)| + |K(b)|) the guard measures noise in, and the
// largest exponent -arg of the five points.  Pointwise: f = K_r(x) P_l(x) as in mu_step.
template <int R, int LMAX>
static void node_eval(const FgPair& q, const MuGrid& grid, const double* const* f, double a, double b,
                      double wp, double* diff, double* S2out, double* scale, double* xmax) {
  const double c = 0.5 * (a + b), h = b - a, d = 0.5 * (a + c), e = 0.5 * (c + b);
#if NDPP_FAST
  const double w = h * (1.0 / 12.0);
#else
  const double w = h / 12.0;
#endif
  const double x5[5] = {a, d, c, e, b};
  double K[5][R], P[5][LMAX];
  *xmax = 0.0;
  for (int k = 0; k < 5; ++k) {
    fg_Krows<R>(q, grid, f, x5[k], K[k]);
    pn_all<LMAX>(x5[k], P[k]);
    double alpha = (q.EpE - 2.0 * x5[k] * q.s2) / q.AkT;
    if (alpha < 1e-6) alpha = 1e-6;
    const double t = alpha + q.beta, x = (t * t) / (4.0 * alpha);
    if (x > *xmax) *xmax = x;
  }
  for (int r = 0; r < R; ++r) {
    scale[r] = w * (fabs(K[0][r]) + 4.0 * fabs(K[1][r]) + 2.0 * fabs(K[2][r]) + 4.0 * fabs(K[3][r]) + fabs(K[4][r]));
    for (int l = 0; l < LMAX; ++l) {
      const double fa = K[0][r] * P[0][l], fd = K[1][r] * P[1][l], fc = K[2][r] * P[2][l],
                   fe = K[3][r] * P[3][l], fb = K[4][r] * P[4][l];
      const double S = opaque(simpson(wp, fa, fc, fb));
      const double S2 = simpson(w, fa, fd, fc) + simpson(w, fc, fe, fb);
      diff[r * LMAX + l] = S2 - S;
      S2out[r * LMAX + l] = S2;
    }
  }
}

struct WalkLog { long n, cap; double *a, *b, *wp; int* depth; };

template <int R, int LMAX>
static void union_walk(const FgPair& q, const MuGrid& grid, const double* const* f, double a, double b, double wp,
                       int depth, double tol, int its, WalkLog& log) {
  if (log.n >= log.cap) return;
  double diff[R * LMAX], S2[R * LMAX], sc[R], xm;
  node_eval<R, LMAX>(q, grid, f, a, b, wp, diff, S2, sc, &xm);
  log.a[log.n] = a; log.b[log.n] = b; log.wp[log.n] = wp; log.depth[log.n] = depth; log.n++;
  const double eps15 = 15.0 * ldexp(tol, -depth);
  bool refine = false;
  for (int ch = 0; ch < R * LMAX; ++ch) refine |= !(fabs(diff[ch]) <= eps15);
  if (!refine || its - depth <= 0) return;
  const double c = 0.5 * (a + b), h = b - a;
#if NDPP_FAST
  const double w = h * (1.0 / 12.0);
#else
  const double w = h / 12.0;
#endif
  union_walk<R, LMAX>(q, grid, f, a, c, w, depth + 1, tol, its, log);
  union_walk<R, LMAX>(q, grid, f, c, b, w, depth + 1, tol, its, log);
}

// mode 0: walk the union tree of one inner integral (all channels active at the root, a node is
// refined while ANY channel fails its test) in this library's arithmetic and log the nodes;
// mode 1: evaluate the n given nodes.  R = 2 rows, L = 6 orders (the headline's shape).
extern "C" long hostsim_guard_nodes(const ndpp_params* p, double A, double kT, double Ein, double Eout,
                                    const double* f_rows, int mode, long n, long cap, double* a, double* b,
                                    double* wp, int* depth, double* diff, double* S2, double* scale,
                                    double* xmax, double* mu_lim) {
  constexpr int R = 2, LMAX = 6;
  const MuGrid grid = make_mu_grid(p->mu_bins);
  const double* f[R] = {f_rows, f_rows + p->mu_bins};
  const FgPair q = make_pair(A, kT, Ein, Eout);
  if (mode == 0) {
    double mlo, mhi;
    fg_find_mu(q, A, Ein, Eout, p->sab_threshold, p->brent_mu_thresh, mlo, mhi);
    mu_lim[0] = mlo; mu_lim[1] = mhi;
    WalkLog log{0, cap, a, b, wp, depth};
    if (mhi > mlo) union_walk<R, LMAX>(q, grid, f, mlo, mhi, (mhi - mlo) / 6.0, 0, p->adaptive_mu_tol,
                                       p->adaptive_mu_its, log);
    n = log.n;
  }
  for (long i = 0; i < n; ++i)
    node_eval<R, LMAX>(q, grid, f, a[i], b[i], wp[i], diff + i * R * LMAX, S2 + i * R * LMAX, scale + i * R, xmax + i);
  return n;
}

// One inner integral (one E_out point of one incoming energy, both rows, L = 6) through the very
// stage functions the kernels run (fg_prep_task, mu_init, mu_step, mu_finish) in this library's
// arithmetic: out[12] = F of every channel, returns the node visits.
// prep[8] = {mu_lo, mu_hi, K of row 0/1 at mu_lo, at mu_hi, at the midpoint}: written when use_prep = 0,
// taken as given when use_prep = 1 (the product always takes them from the strict prep stage).
extern "C" long hostsim_inner_integral(const ndpp_params* p, double A, double kT, double Ein, double Eout,
                                       const double* f_rows, double* out, double* prep, int use_prep) {
  constexpr int R = 2, LMAX = 6;
  FgBatch B;
  B.n_jobs = 1; B.R = R; B.G = 1; B.L = LMAX; B.M = p->mu_bins;
  B.A = A; B.kT = kT;
  const int row[2] = {0, 1};
  const double ebins[2] = {0.0, 20.0};
  B.job_ein = &Ein; B.job_row = row; B.f_tab = f_rows; B.e_bins = ebins;
  B.sab_threshold = p->sab_threshold; B.brent_thresh = p->brent_mu_thresh;
  B.mu_tol = p->adaptive_mu_tol; B.eout_tol = p->adaptive_eout_tol;
  B.mu_its = p->adaptive_mu_its; B.eout_its = p->adaptive_eout_its;
  B.grid = make_mu_grid(B.M);
  B.ncap = 8;
  std::vector<double> na(8, Eout), nb(8, Eout), nF((size_t)5 * R * LMAX * 8), nS((size_t)R * LMAX * 8);
  std::vector<int> info(32, 0);
  B.node_a = na.data(); B.node_b = nb.data(); B.node_F = nF.data(); B.node_S = nS.data(); B.node_info = info.data();
  info[0] = (int)B.full_mask(); info[1] = -1;
  B.tcap = 16;
  std::vector<double> t1(16), t2(16), t3((size_t)3 * R * 16);
  B.t_mulo = t1.data(); B.t_muhi = t2.data(); B.t_X = t3.data();
  std::vector<int> cnt(kMaxLevels + 2, 0);
  int next = 0, ovf = 0;
  unsigned long long stats[kNumStats] = {0};
  B.lvl_cnt = cnt.data(); B.next_task = &next; B.overflow = &ovf; B.stats = stats; B.raw = nullptr;
  cnt[0] = 1;
  if (use_prep) {
    B.t_mulo[0] = prep[0]; B.t_muhi[0] = prep[1];
    for (int k = 0; k < 3; ++k) for (int r = 0; r < R; ++r) B.tX(k, r, 0) = prep[2 + k * R + r];
  } else {
    fg_prep_task(B, 0, 0, 0);
    prep[0] = B.t_mulo[0]; prep[1] = B.t_muhi[0];
    for (int k = 0; k < 3; ++k) for (int r = 0; r < R; ++r) prep[2 + k * R + r] = B.tX(k, r, 0);
  }
  MuLane<R, LMAX> s;
  HostMuStack<R> st{};
  mu_init<R, LMAX>(B, 0, 0, 0, s);
  if (s.mask == 0) return 0;
  mu_tot_zero(s);
  const PnConsts pk = make_pn_consts();
  while (mu_step<R, LMAX, HostMuStack<R>, false>(B, s, st, pk)) {}
  mu_finish(B, s, false);
  for (int ch = 0; ch < R * LMAX; ++ch) out[ch] = B.F(0, ch, 0);
  return (long)s.visits;
}

// Statistics for the design of the inner walk: the union tree of one inner integral walked with
// every channel's own activity (a channel is tested at a node only while it still refines), and a
// histogram of the visits over the set of orders active in any row (6 bits).
template <int R, int LMAX>
static void mask_walk(const FgPair& q, const MuGrid& grid, const double* const* f, double a, double b, double wp,
                      int depth, unsigned mask, double tol, int its, unsigned long long* hist) {
  double diff[R * LMAX], S2[R * LMAX], sc[R], xm;
  node_eval<R, LMAX>(q, grid, f, a, b, wp, diff, S2, sc, &xm);
  unsigned any = 0;
  for (int r = 0; r < R; ++r) any |= (mask >> (r * kRowBits)) & ((1u << LMAX) - 1u);
  hist[any] += 1;
  const double eps15 = 15.0 * ldexp(tol, -depth);
  unsigned refine = 0;
  if (its - depth > 0)
    for (int r = 0; r < R; ++r)
      for (int l = 0; l < LMAX; ++l)
        if ((mask & chan_bit(r, l)) && !(fabs(diff[r * LMAX + l]) <= eps15)) refine |= chan_bit(r, l);
  if (!refine) return;
  const double c = 0.5 * (a + b), h = b - a;
#if NDPP_FAST
  const double w = h * (1.0 / 12.0);
#else
  const double w = h / 12.0;
#endif
  mask_walk<R, LMAX>(q, grid, f, a, c, w, depth + 1, refine, tol, its, hist);
  mask_walk<R, LMAX>(q, grid, f, c, b, w, depth + 1, refine, tol, its, hist);
}

extern "C" void hostsim_mask_histogram(const ndpp_params* p, double A, double kT, double Ein, double Eout,
                                       const double* f_rows, unsigned long long* hist64) {
  constexpr int R = 2, LMAX = 6;
  const MuGrid grid = make_mu_grid(p->mu_bins);
  const double* f[R] = {f_rows, f_rows + p->mu_bins};
  const FgPair q = make_pair(A, kT, Ein, Eout);
  double mlo, mhi;
  fg_find_mu(q, A, Ein, Eout, p->sab_threshold, p->brent_mu_thresh, mlo, mhi);
  if (!(mhi > mlo)) return;
  unsigned full = 0;
  for (int r = 0; r < R; ++r) full |= ((1u << LMAX) - 1u) << (r * kRowBits);
  mask_walk<R, LMAX>(q, grid, f, mlo, mhi, (mhi - mlo) / 6.0, 0, full, p->adaptive_mu_tol, p->adaptive_mu_its, hist64);
}
