// fg_pipeline.h -- the free-gas (Doppler) elastic scattering-moment pipeline,
// expressed as per-work-item stage functions (NDPP_HD).  fg_device.h wraps
// each stage in a gfx950 kernel; tests/hostsim drives the very same functions
// sequentially on the CPU to check the algorithm where no GPU exists.
//
// What is computed (reference: integrate_freegas_leg, freegas.F90:18-146):
// for every "call" (one incoming energy E_in x one tabulated f(mu) row) and
// every outgoing group g, up to five E_out segments are integrated by nested
// adaptive Simpson quadrature (outer: E_out, freegas.F90:563-644; inner: mu,
// :482-553) for each Legendre order l.
//
// How it is organised here (MI355X-first, not the reference's recursion):
//   * The reference re-runs the whole nested quadrature once per Legendre
//     order.  All orders evaluate fgk(l,mu) = K(mu)*P_l(mu) at dyadic points
//     of the SAME intervals, so we walk the UNION of the per-order refinement
//     trees once: K (exp, sqrt, 5 divides) is evaluated once per point, while
//     every order keeps its own Simpson estimates, its own accept/refine
//     decision and its own sum -- i.e. exactly its own reference tree (a bit
//     mask says which orders are still refining below a node).
//   * Outer (E_out) trees are expanded breadth-first, one level per launch
//     sequence (prep -> mu -> node), over ALL calls of the batch at once, so a
//     level exposes 10^5..10^7 independent inner integrals.
//   * Inner (mu) integrals are the hot part (>99% of the arithmetic): one lane
//     owns one integral and walks its tree depth-first with an explicit,
//     direct-mapped (slot = depth) stack of right siblings; lanes of a
//     wavefront run the same branch-free step and pull the next integral from
//     a global counter when theirs is finished.
//   * The two bracketing f(mu) rows that integrate_distro evaluates at the same
//     incoming energy (scattdata_header.F90:550,573) share E_in, E_out, mu and
//     therefore the exp/rsqrt factor of K: they are walked as one union tree
//     with 2*L channels (both arithmetics; the reference arithmetic evaluates each
//     row's kernel value as the reference writes it and shares what does not depend
//     on the row).
//   * Sums: outer trees are reduced bottom-up in the reference's own order
//     (val = left + right, freegas.F90:639-642) -> schedule independent and
//     identical to the reference.  Inner leaves are accumulated left-to-right
//     per order and segment (the reference adds them pairwise up the tree; both
//     are within rounding of the exact sum and decide nothing inside the walk).
#pragma once

#include "ndpp_math.h"

namespace ndpp {
#if NDPP_FAST
inline namespace fast_arith {
#else
inline namespace strict_arith {
#endif

// Legendre orders per block of the inner walk's step (mu_step): a block is skipped by a wave
// none of whose lanes has any of its orders active; inside a block the orders' (short,
// dependent) chains overlap.  Two in the product arithmetic (+0.9 %); one in the reference
// arithmetic, whose blocks are long enough -- divisions, the parent's estimate rebuilt -- that
// pairing them only costs registers (-14 %).
#if NDPP_FAST
constexpr int kMuBlock = 2;
#else
constexpr int kMuBlock = 1;
#endif
constexpr int kSegPerGroup = 5;  // 2 tails + up to 3 pieces (freegas.F90:80-116)
constexpr int kMaxLevels = 32;   // supported adaptive_*_its < kMaxLevels
constexpr int kStackLdsLevels = 8;
constexpr int kMaxRows = 2;      // tabulated rows integrated jointly per incoming energy
// Inner integrals are summed per "segment" = per depth-kSplitLog2 node of their tree
// (leaves accepted higher up count for their left-most segment) and the segment sums
// are added left to right.  One lane walking the whole tree and many lanes walking a few
// segments each therefore produce the same bits (mu_step / fg_mu_combine).  The leaves of a
// segment are added left to right in plain double in both arithmetics: the reference adds them
// pairwise up its recursion, without compensation either, and the sums decide nothing inside
// the walk (measured in round 3: every parity statistic unchanged against Kahan's sums).
//
// Split walk (levels with few inner integrals): an integral is handed out as kSplit work
// items, each a node of depth kSplitLog2 whose sum goes to its own slot.  (A finer hand-out --
// 64 slots, 25 items -- was measured in round 3: no gain on a 12 500-energy shard, -25 % on the
// 16-channel walk; experiments/README.md.)
constexpr int kSplitLog2 = 4;
constexpr int kSplit = 1 << kSplitLog2;          // segments (= summation slots = work items) per inner integral
constexpr int kRowBits = 16;     // channel (row r, order l) <-> mask bit r*kRowBits + l

enum { kStatKEvals = 0, kStatMuVisits, kStatMuIntegrals, kStatEoutNodes,
       kStatWaveIters, kStatLaneIters, kStatOrderVisits, kStatGaussIntegrals, kNumStats };

NDPP_HD unsigned chan_bit(int r, int l) { return 1u << (r * kRowBits + l); }

// One batch of jobs, everything the stages need.  Plain pointers: device
// pointers inside kernels, host pointers inside the host simulator.
//
// A "job" is one incoming energy with R tabulated f(mu) rows (R = 2: the two
// bracketing rows integrate_distro needs, scattdata_header.F90:550,573; R = 1:
// a single integrate_freegas_leg call).  The rows of a job share E_in, the
// E_out segments, find_FG_mu and -- in the product's arithmetic -- the
// exp/rsqrt factor of the kernel, so they are walked as ONE union tree with
// R*L "channels" (row, order), each channel keeping its own reference tree.
// "call" c = job*R + r indexes the per-row results (raw).
struct FgBatch {
  // ---- problem
  int n_jobs, R, G, L, M;
  double A, kT;           // target mass ratio and temperature (MeV) of the whole batch ...
  const double* job_A = nullptr;   // ... or per job [n_jobs] when nuclides are mixed in
  const double* job_kT = nullptr;  //     one batch (ndpp_elastic_leg_multi)
  const double* job_ein;  // [n_jobs]
  const int* job_row;     // [n_jobs*R] rows of f_tab
  const double* f_tab;    // [n_rows][M]
  const double* e_bins;   // [G+1]
  // ---- numerics (module global, global.F90:32-48)
  double sab_threshold, brent_thresh, mu_tol, eout_tol;
  int mu_its, eout_its;
  MuGrid grid;
  // ---- outer-tree node arena (structure of arrays, capacity ncap)
  int ncap;
  double* node_a;   // [ncap]
  double* node_b;   // [ncap]
  double* node_F;   // [(slot*NCH + ch)*ncap + n]; slot 0..4 = points a,d,c,e,b
  double* node_S;   // [ch*ncap + n]; coarse estimate S, then the node's value
  int* node_info;   // [4*n + {0: channel mask, 1: left child or -1,
                    //         2: job | depth<<26, 3: refine mask}]
  // ---- inner-integral task records of the current level, [5*n_trees] or [2*n]
  int tcap;
  double* t_mulo;
  double* t_muhi;
  double* t_X;      // [(k*R + r)*tcap + t]: K of row r at mu_lo (k = 0), mu_hi (1), the midpoint (2)
  // per task record: bit r = row r's inner integrals are done by the Gauss rule (mu_gauss_task) and
  // the adaptive walk leaves that row's channels out (decided per row: a row's result does not
  // depend on which other row shares its job); null = the Gauss rule is off (reference arithmetic,
  // non-linear tables)
  unsigned char* t_gl = nullptr;
  // the Gauss stage's zone (fg_gauss_zone) and effort (mu_gauss_certify, mu_gauss_task)
  double gl_ratio = 2.0;       // "far" inner integrals: E_out / E_in or E_in / E_out at least this
  int gl_near = 1;             // 1: the others ("near": next to the peak E_out = E_in) are candidates too
  double gl_amin = 2.0E-6;     // alpha(mu = 1) at least this (the reference clamps alpha at 1e-6: a kink)
  int gl_cert_depth = 7;       // levels of the reference's inner tree certified, far ...
  int gl_cert_depth_near = 8;  // ... and near
  int gl_panels = 16;          // finest uniform composite rule tried (panels of 16 points; doubled from 8 up to this)
  int gl_graded = 8;           // near candidates: levels of the graded rule tried first (0: off)
  // ---- counters
  int* lvl_cnt;   // [kMaxLevels+1] nodes per outer level
  int* next_task; // [kMaxLevels+1] dynamic task counters of the mu kernel
  int* overflow;  // [1] set when ncap was too small
  unsigned long long* stats;  // [kNumStats]
  // ---- split mode: a level with at most split_below inner integrals is walked by
  // kSplit lanes per integral, each writing its segment sum to seg[(t*kSplit+j)*nch+ch]
  double* seg = nullptr;
  int split_below = 0;
  // ---- task order of the current level (nodes sorted by mask); null = node order
  const int* order = nullptr;
  const int* mask_rank = nullptr;   // [2^L] bucket of a sort key (fg_sort_key): many orders first
  // mu_nodes (device pipeline): number of nodes of the level with an order still active; they
  // come first in `order`
  const int* mu_nodes = nullptr;
  // ---- results
  double* raw;    // [n_jobs*R][G][L] per-call normalised moments

  NDPP_HD double A_of(int job) const { return job_A ? job_A[job] : A; }
  NDPP_HD double kT_of(int job) const { return job_kT ? job_kT[job] : kT; }
  NDPP_HD int nch() const { return R * L; }
  NDPP_HD int n_trees() const { return n_jobs * G * kSegPerGroup; }
  NDPP_HD int lvl_off(int level) const {
    int o = 0;
    for (int k = 0; k < level; ++k) o += lvl_cnt[k];
    return o;
  }
  NDPP_HD bool split_level(int level) const {
    const int nt = n_mu_tasks(level);
    return seg != nullptr && nt > 0 && nt <= split_below;
  }
  NDPP_HD int tasks_per_node(int level) const { return level == 0 ? 5 : 2; }
  NDPP_HD int n_tasks(int level) const { return lvl_cnt[level] * tasks_per_node(level); }
  // inner integrals the walk has to do on this level
  NDPP_HD int n_mu_tasks(int level) const {
    return (mu_nodes ? *mu_nodes : lvl_cnt[level]) * tasks_per_node(level);
  }
  // The task records (mu limits, kernel values of the root estimate) are indexed by node and
  // point, not by position in the task order.
  NDPP_HD int rec_index(int level, int base, int n, int slot) const {
    return level == 0 ? 5 * n + slot : 2 * (n - base) + (slot == 3 ? 1 : 0);
  }
  NDPP_HD double& F(int slot, int ch, int n) const {
    return node_F[((size_t)(slot * R * L + ch)) * ncap + n];
  }
  NDPP_HD double& S(int ch, int n) const { return node_S[(size_t)ch * ncap + n]; }
  NDPP_HD double& tX(int k, int r, int t) const { return t_X[((size_t)(k * R + r)) * tcap + t]; }
  NDPP_HD unsigned full_mask() const {
    unsigned m = 0;
    for (int r = 0; r < R; ++r) m |= ((1u << L) - 1u) << (r * kRowBits);
    return m;
  }
  NDPP_HD int node_job(int n) const { return node_info[4 * n + 2] & 0x3ffffff; }
  NDPP_HD int node_depth(int n) const { return node_info[4 * n + 2] >> 26; }
};

// -----------------------------------------------------------------------------
// Stage 0: per (job, group) -- lay out the E_out segments of
// integrate_freegas_leg (freegas.F90:52-131) as root nodes of level 0.
// Root slot s of (job,g): 0 = low tail, 1 = high tail, 2 = [Elo,alphaEin],
// 3 = [Elo,Ein], 4 = remainder (or the whole group in the `else` branch).
// A slot that the reference does not integrate, or integrates over a
// zero-width interval (value exactly 0), gets mask 0.
// -----------------------------------------------------------------------------
NDPP_HD void fg_setup_group(const FgBatch& B, int job, int g) {
  const double Ein = B.job_ein[job];
  const double A = B.A_of(job), kT = B.kT_of(job);
  double alphaEin = (A - 1.0) / (A + 1.0);
  alphaEin = alphaEin * alphaEin * Ein;
  double Eout_lo, Eout_hi;
  fg_eout_bounds(A, kT, Ein, Eout_lo, Eout_hi);
  const double eg = B.e_bins[g], eg1 = B.e_bins[g + 1];

  double sa[kSegPerGroup], sb[kSegPerGroup];
  bool on[kSegPerGroup] = {false, false, false, false, false};
  if ((eg < Eout_hi) && (eg1 > Eout_lo)) {
    double Elo = (Eout_lo > eg) ? Eout_lo : eg;
    double Ehi = (Eout_hi < eg1) ? Eout_hi : eg1;
    double Ebottom = (eg == 0.0) ? 0.01 * Elo : eg;
    sa[0] = Ebottom; sb[0] = Elo; on[0] = true;
    sa[1] = Ehi;     sb[1] = eg1; on[1] = true;
    if ((Elo < alphaEin) && (alphaEin < Ehi)) {
      sa[2] = Elo; sb[2] = alphaEin; on[2] = true;
      Elo = alphaEin;
    }
    if ((Elo < Ein) && (Ein < Ehi)) {
      sa[3] = Elo; sb[3] = Ein; on[3] = true;
      Elo = Ein;
    }
    sa[4] = Elo; sb[4] = Ehi; on[4] = true;
  } else {
    sa[4] = eg; sb[4] = eg1; on[4] = true;  // freegas.F90:126-130
  }
  const unsigned full = B.full_mask();
  const int nch = B.nch();
  for (int s = 0; s < kSegPerGroup; ++s) {
    int n = (job * B.G + g) * kSegPerGroup + s;
    bool live = on[s] && (sa[s] != sb[s]);
    B.node_a[n] = live ? sa[s] : 0.0;
    B.node_b[n] = live ? sb[s] : 0.0;
    B.node_info[4 * n + 0] = live ? (int)full : 0;
    B.node_info[4 * n + 1] = -1;
    B.node_info[4 * n + 2] = job;  // depth 0
    B.node_info[4 * n + 3] = 0;
    for (int ch = 0; ch < nch; ++ch) B.S(ch, n) = 0.0;
  }
}

// task t of `level` -> (node, point slot)
NDPP_HD void fg_task_decode(const FgBatch& B, int level, int base, int t, int& n,
                            int& slot) {
  (void)B;
  int idx;
  if (level == 0) {
    idx = t / 5;
    slot = t - 5 * idx;
  } else {
    idx = t >> 1;
    slot = 1 + 2 * (t & 1);
  }
  // B.order (device pipeline): the level's nodes sorted by order mask, so that the lanes of a
  // wave walk integrals with the same set of active orders and the per-order blocks of the
  // others are skipped for the whole wave
  n = B.order ? B.order[idx] : (level == 0 ? idx : base + idx);
}

// Sort key of a node for the task order of a level: the orders active in ANY row (L bits).  The
// per-order blocks of the inner walk are shared by the rows of a job and skipped by a wave when
// no lane has the order active in any row, so this is exactly what the lanes of a wave should
// have in common.
NDPP_HD unsigned fg_sort_key(const FgBatch& B, unsigned mask) {
  unsigned m = 0;
  for (int r = 0; r < B.R; ++r) m |= (mask >> (r * kRowBits)) & ((1u << B.L) - 1u);
  return m;
}

// Bucket of a node in the task order of its level (after the prep and Gauss stages): weight class
// first -- the long inner integrals start first, the short ones fill the end of the launch --,
// then the orders still active (mask_rank: many orders first).  The weight of an inner integral
// is window x largest root kernel value (row 0): the reference's tolerance is absolute, so that
// product says how deep its tree goes (measured: >= 2^30 -> 2e4 visits, 2^20 ... 2^30 -> ~1e4,
// below 2^8 -> hundreds).  A node with nothing left for the walk -- no order active, or every
// inner integral taken by the Gauss stage -- goes to the last bucket, which the walk does not visit.
// The order decides nothing but the schedule: every result has its own slot.
constexpr int kSortClasses = 4;
NDPP_HD int fg_node_bucket(const FgBatch& B, int level, int base, int n, int nb) {
  const unsigned mask = (unsigned)B.node_info[4 * n + 0];
  const int last = kSortClasses * nb - 1;
  if (mask == 0) return last;
  double wmax = -1.0;
  const int nslots = B.tasks_per_node(level);
  for (int k = 0; k < nslots; ++k) {
    const int slot = level == 0 ? k : 1 + 2 * k;
    const int rec = B.rec_index(level, base, n, slot);
    unsigned m = mask;
    if (B.t_gl) {
      const unsigned rows = B.t_gl[rec];
      for (int r = 0; r < B.R; ++r)
        if (rows >> r & 1u) m &= ~(((1u << kRowBits) - 1u) << (r * kRowBits));
    }
    if (m == 0) continue;                     // this inner integral is done
    const double km = fmax(fmax(fabs(B.tX(0, 0, rec)), fabs(B.tX(1, 0, rec))), fabs(B.tX(2, 0, rec)));
    const double w = (B.t_muhi[rec] - B.t_mulo[rec]) * km;
    wmax = fmax(wmax, w >= 0.0 ? w : 0.0);    // (NaN: weight 0)
  }
  if (wmax < 0.0) return last;
  const int cls = wmax >= 1073741824.0 ? 0 : wmax >= 1048576.0 ? 1 : wmax >= 256.0 ? 2 : 3;
  int b = cls * nb + B.mask_rank[fg_sort_key(B, mask)];
  return b < last ? b : last - 1;             // (an active node never lands in the last bucket)
}

NDPP_HD double fg_slot_point(double a, double b, int slot) {
  double c = 0.5 * (a + b);
  switch (slot) {
    case 0: return a;
    case 1: return 0.5 * (a + c);
    case 2: return c;
    case 3: return 0.5 * (c + b);
    default: return b;
  }
}

// The kernel value K_r(mu) of tabulated row r at one point.  In the product arithmetic
// K_r = (C1 * f_r(mu)) * E(mu) with the row-independent factor E (exp and rsqrt) shared by
// the rows of a job; the reference arithmetic evaluates calc_fgk as written for every row
// and shares the sub-expressions that do not involve the row (same operands, same bits).
template <int R>
NDPP_HD void fg_Krows(const FgPair& q, const MuGrid& g, const double* const* f, double mu,
                      double* K) {
#if NDPP_FAST
  const double E = fg_E(q, mu);
#pragma unroll
  for (int r = 0; r < R; ++r) K[r] = (q.C1 * fg_fval(g, f[r], mu)) * E;
#else
  fg_K_rows<R>(q, g, f, mu, K);
#endif
}
// w * (f0 + 4 f1 + f2): Simpson's rule on one interval (freegas.F90:505, :539-541)
// 4 f1 is exact, so RN(f0 + RN(4 f1)) is one fused multiply-add: the reference's bits, one
// instruction fewer (three per channel and visit in the reference arithmetic).
NDPP_HD double simpson(double w, double f0, double f1, double f2) {
  return w * (fma(4.0, f1, f0) + f2);
}

// -----------------------------------------------------------------------------
// Stage 1 (prep): one inner integral = one E_out point.  find_FG_mu
// (freegas.F90:356-409, incl. the Brent searches) and the three kernel values
// the root Simpson estimate needs (adaptiveSimpsons_mu, :498-503).
// -----------------------------------------------------------------------------
// Is the inner integral of this pair in the zone the Gauss rule may take (mu_gauss_task below)?
constexpr unsigned kGaussNear = 0x80u;     // t_gl flag (prep -> Gauss stage): a "near" candidate
NDPP_HD unsigned fg_gauss_zone(const FgBatch& B, const FgPair& q, double Ein, double Eout) {
  const double amin = (q.EpE - 2.0 * q.s2) / q.AkT;
  if (!(amin >= B.gl_amin)) return 0u;
  const unsigned rows = (1u << B.R) - 1u;
  if (Eout >= B.gl_ratio * Ein || Ein >= B.gl_ratio * Eout) return rows;
  return B.gl_near ? (rows | kGaussNear) : 0u;
}

NDPP_HD void fg_prep_task(const FgBatch& B, int level, int base, int t) {
  // tasks in node order whatever the walk's order is: t == rec_index(level, base, n, slot)
  int n, slot;
  if (level == 0) { n = t / 5; slot = t - 5 * n; }
  else { n = base + (t >> 1); slot = 1 + 2 * (t & 1); }
  if (B.node_info[4 * n + 0] == 0) {
    if (B.t_gl) B.t_gl[t] = 0;
    return;
  }
  const int job = B.node_job(n);
  const double Ein = B.job_ein[job];
  const double Eout = fg_slot_point(B.node_a[n], B.node_b[n], slot);
  const double A = B.A_of(job);
  const FgPair q = make_pair(A, B.kT_of(job), Ein, Eout);
  double mlo, mhi;
  fg_find_mu(q, A, Ein, Eout, B.sab_threshold, B.brent_thresh, mlo, mhi);
  const double mc = (mlo + mhi) * 0.5;
  B.t_mulo[t] = mlo;
  B.t_muhi[t] = mhi;
  if (B.t_gl) B.t_gl[t] = (unsigned char)fg_gauss_zone(B, q, Ein, Eout);
  // the three kernel values of every row's root estimate (adaptiveSimpsons_mu, :498-503)
  for (int r = 0; r < B.R; ++r) {
    const double* fr[1] = {B.f_tab + (size_t)B.job_row[(size_t)job * B.R + r] * B.M};
    fg_Krows<1>(q, B.grid, fr, mlo, &B.tX(0, r, t));
    fg_Krows<1>(q, B.grid, fr, mhi, &B.tX(1, r, t));
    fg_Krows<1>(q, B.grid, fr, mc, &B.tX(2, r, t));
  }
}

// -----------------------------------------------------------------------------
// Stage 1b (Gauss): the inner integrals the reference has CONVERGED, by a fixed high-order rule.
//
// The reference's inner integration (adaptiveSimpsons_mu, freegas.F90:482-553) asks for an absolute
// 1e-7 on an integrand of 1e4 ... 1e9: it refines until the Simpson estimates agree to rounding (or
// the depth limit stops it), 3000 ... 20 000 node visits per integral.  Where every node it accepts
// is accepted because the estimate HAS converged, what it returns is the integral itself, to
// ~1e-13 of the integral's scale -- and any converged quadrature returns the same number.  What a
// fixed rule cannot reproduce is an acceptance that is NOT convergence:
//   (i)  in the far tails the absolute tolerance accepts the root estimate of an oscillating
//        K P_l from five points (off by 10x at P5), and the outer quadrature multiplies the end
//        point of the high-tail segment by a node 2^-15 x 20 MeV wide: the reference's up-scatter
//        moments ARE that artefact;
//   (ii) an accidental agreement of S and S2 on a node whose true error is large (the fourth
//        derivative of K P_l changes sign inside it): measured 2.4e-10 of a row from ONE such node
//        at depth 5 of one inner integral, 4e-11 at depth 6, 2.6e-11 at depth 7;
//   (iii) the clamp alpha >= 1e-6 (a kink inside the window) and the depth limit.
// mu_gauss_certify looks for (i) and (ii) in the top levels of the reference's own tree, per inner
// integral and row; (iii) is excluded by the zone (alpha(mu = 1) >= 2e-6) and by the agreement test
// of the rule itself.  Whatever is not certified is walked.  The parity of the whole is measured by
// tools/parity_tail.py on every energy of the production workloads (profiles/r04/).
//
// Zone (decided in the prep stage, from the pair alone): alpha(mu = 1) = (sqrt E - sqrt E')^2 / (A kT)
// >= 2e-6; "far" = E_out / E_in outside (1/2, 2), "near" = the rest (next to the peak, where the
// integrals weigh most in their rows: certified one level deeper).  Only for tables certified
// linear in mu (the product arithmetic's domain: the interpolant of any other table has kinks
// inside the window).
// -----------------------------------------------------------------------------
#if NDPP_FAST
constexpr int kGaussN = 16;
// nodes and weights of the 16-point Gauss-Legendre rule on [-1, 1]
NDPP_HD double gauss_node(int j) {
  constexpr double x[kGaussN] = {
      -0.98940093499164994, -0.9445750230732326, -0.86563120238783176, -0.755404408355003,
      -0.61787624440264377, -0.45801677765722737, -0.28160355077925892, -0.095012509837637454,
      0.095012509837637454, 0.28160355077925892, 0.45801677765722737, 0.61787624440264377,
      0.755404408355003, 0.86563120238783176, 0.9445750230732326, 0.98940093499164994};
  return x[j];
}
NDPP_HD double gauss_weight(int j) {
  constexpr double w[kGaussN] = {
      0.027152459411754037, 0.062253523938647706, 0.095158511682492591, 0.12462897125553403,
      0.14959598881657676, 0.16915651939500262, 0.18260341504492361, 0.18945061045506859,
      0.18945061045506859, 0.18260341504492361, 0.16915651939500262, 0.14959598881657676,
      0.12462897125553403, 0.095158511682492591, 0.062253523938647706, 0.027152459411754037};
  return w[j];
}

// composite rule with `panels` equal panels on [a, b]: acc[r*LMAX + l] (+)= sum w K_r(mu) P_l(mu)
template <int R, int LMAX, bool kAdd = false>
NDPP_HD void gauss_composite(const FgBatch& B, const FgPair& q, const FView<R>& fv, double a, double b,
                             int panels, const PnConsts& pk, double* acc) {
  if (!kAdd) {
#pragma unroll
    for (int ch = 0; ch < R * LMAX; ++ch) acc[ch] = 0.0;
  }
  const double h = (b - a) / (double)(2 * panels);          // half width of a panel
  for (int p = 0; p < panels; ++p) {
    const double c = a + h * (double)(2 * p + 1);
    // two points per step: both table lookups are requested before the two exp/rsqrt chains run
    // side by side (fg_E2), as in the walk's step
    for (int j = 0; j < kGaussN; j += 2) {
      const double mu0 = fma(h, gauss_node(j), c), mu1 = fma(h, gauss_node(j + 1), c);
      FvLoad v0[R], v1[R];
      fg_fval_load_rows<R>(B.grid, fv, mu0, v0);
      fg_fval_load_rows<R>(B.grid, fv, mu1, v1);
      double E0, E1;
      fg_E2(q, mu0, mu1, E0, E1);
      E0 *= h * gauss_weight(j);
      E1 *= h * gauss_weight(j + 1);
      double P0[LMAX], P1[LMAX];
      pn_all<LMAX>(mu0, P0, pk);
      pn_all<LMAX>(mu1, P1, pk);
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double K0 = (q.C1 * fg_fval_use(v0[r])) * E0;
        const double K1 = (q.C1 * fg_fval_use(v1[r])) * E1;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) acc[r * LMAX + l] = fma(K1, P1[l], fma(K0, P0[l], acc[r * LMAX + l]));
      }
    }
  }
}

// Graded rule for an integrand with its sharp end at b (next to E_out = E_in the kernel peaks within
// alpha_min / 2 of mu = 1, the upper end of every window): panels [a, b - W/2], [b - W/2, b - W/4],
// ... [b - W/2^m, b], each with `split` equal sub-panels of 16 points.
template <int R, int LMAX>
NDPP_HD void gauss_graded(const FgBatch& B, const FgPair& q, const FView<R>& fv, double a, double b,
                          int m, int split, const PnConsts& pk, double* acc) {
#pragma unroll
  for (int ch = 0; ch < R * LMAX; ++ch) acc[ch] = 0.0;
  const double W = b - a;
  double lo = a;
  for (int k = 1; k <= m + 1; ++k) {
    const double hi = k <= m ? b - ldexp(W, -k) : b;
    gauss_composite<R, LMAX, true>(B, q, fv, lo, hi, split, pk, acc);
    lo = hi;
  }
}

// One flagged inner integral (task t in node order, as in fg_prep_task): composite 16-point
// Gauss-Legendre rules with 4, 8, 16 ... B.gl_panels panels; a row takes the first rule that agrees
// with the one before it to 1e-13 of the integral of K (channel (r, 0): K > 0) in every channel
// (far from the peak the 8-panel rule, next to it up to 128 panels); a row for which none does is
// walked.  Returns the kernel evaluations spent.
constexpr double kGaussAgree = 1.0E-13;
// Certification (mu_gauss_certify): the top levels of the integral's tree are evaluated as the
// walk evaluates them (the same dyadic points, the same kernel values -- the root's three from the
// prep stage --, the same channel arithmetic as mu_step).
//   * Levels 0 ... kCertDepth - 1 (31 nodes): every channel must REFINE at every node -- (i) above:
//     an acceptance this high up, converged or not, is the reference's own artefact.
//   * Levels kCertDepth ... B.gl_cert_depth[_near] - 1: an acceptance is fine when the estimate has
//     converged there and an ACCIDENT when it has not; the two are told apart by the neighbours of
//     the same level: a converged region has small test values all around, an accident sits next
//     to a node whose test value is kCertBig times above the threshold.  A row with an accident
//     goes to the walk.
// Below the certified depth a node is narrower than 1/128 (far) or 1/256 (near) of the window and
// an accident there is worth < 1e-11 of a row (measured).
constexpr int kCertDepth = 5;
constexpr double kCertBig = 64.0;

template <int R, int LMAX>
NDPP_HD unsigned mu_gauss_certify(const FgBatch& B, const FgPair& q, const FView<R>& fv, int t, unsigned mask,
                                  unsigned rows, int cert_depth, const PnConsts& pk) {
  // -> the rows of `rows` that pass (see above).
  // Level by level, left to right; a node takes its left end from its left neighbour and evaluates
  // its other four points (no per-lane arrays: four kernel values per node instead of the tree's
  // two, all in registers).  The dyadic points are a + i (b - a) / 2^k here, the walk's nested midpoints there:
  // an ulp apart at most, which cannot turn an acceptance into a refinement that matters -- the
  // nodes this looks for accept by a wide margin.
  const double a = B.t_mulo[t], b = B.t_muhi[t];
  auto Kat = [&](double mu0, double mu1, double* K0, double* K1) {
    FvLoad v0[R], v1[R];
    fg_fval_load_rows<R>(B.grid, fv, mu0, v0);
    fg_fval_load_rows<R>(B.grid, fv, mu1, v1);
    double E0, E1;
    fg_E2(q, mu0, mu1, E0, E1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      K0[r] = (q.C1 * fg_fval_use(v0[r])) * E0;
      K1[r] = (q.C1 * fg_fval_use(v1[r])) * E1;
    }
  };
  for (int dep = 0; dep < cert_depth; ++dep) {
    const int nn = 1 << dep;
    const double hd = (b - a) / (double)nn;
    const double w = hd * (1.0 / 12.0);
    const double wp = dep == 0 ? hd / 6.0 : (2.0 * hd) * (1.0 / 12.0);
    const double eps15 = 15.0 * ldexp(B.mu_tol, -dep);
    // below kCertDepth an acceptance is an ACCIDENT -- and the row goes to the walk -- only next to
    // a node of the same level whose own test is far from passing (see kCertBig)
    const bool strict = dep < kCertDepth;
    const double big15 = kCertBig * eps15;
    unsigned prev_leaf = 0, prev_big = 0;
    double Ka[R], Pa[LMAX];
#pragma unroll
    for (int r = 0; r < R; ++r) Ka[r] = B.tX(0, r, t);          // (the window's left end: prep stage)
    pn_all<LMAX>(a, Pa, pk);
    for (int j = 0; j < nn; ++j) {
      const double xa = a + hd * (double)j;
      const double xb = (j == nn - 1) ? b : a + hd * (double)(j + 1);
      const double xc = 0.5 * (xa + xb), xd = 0.5 * (xa + xc), xe = 0.5 * (xc + xb);
      double Kd[R], Ke[R], Kc[R], Kb[R];
      Kat(xd, xe, Kd, Ke);
      Kat(xc, xb, Kc, Kb);
      if (dep == 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) { Kc[r] = B.tX(2, r, t); Kb[r] = B.tX(1, r, t); }
      } else if (j == nn - 1) {
#pragma unroll
        for (int r = 0; r < R; ++r) Kb[r] = B.tX(1, r, t);
      }
      double Pd[LMAX], Pc[LMAX], Pe[LMAX], Pb[LMAX];
      pn_all<LMAX>(xd, Pd, pk); pn_all<LMAX>(xc, Pc, pk); pn_all<LMAX>(xe, Pe, pk); pn_all<LMAX>(xb, Pb, pk);
      unsigned leaf = 0, big = 0;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const double Xc2 = 2.0 * Kc[r], Xc4 = 4.0 * Kc[r], Kd4 = 4.0 * Kd[r], Ke4 = 4.0 * Ke[r];
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
          const double T = fma(Kb[r], Pb[l], Ka[r] * Pa[l]);
          const double s1 = fma(Xc4, Pc[l], T);
          const double s2 = fma(Ke4, Pe[l], fma(Kd4, Pd[l], fma(Xc2, Pc[l], T)));
          const double dS = fabs(fma(w, s2, -(wp * s1)));
          if (mask & chan_bit(r, l)) {
            if (!(dS > eps15)) leaf |= chan_bit(r, l);
            if (dS > big15) big |= chan_bit(r, l);
          }
        }
      }
      const unsigned bad = strict ? leaf : ((leaf & prev_big) | (prev_leaf & big));
#pragma unroll
      for (int r = 0; r < R; ++r)
        if (bad & (((1u << kRowBits) - 1u) << (r * kRowBits))) rows &= ~(1u << r);
      if (rows == 0) return 0;
      prev_leaf = leaf;
      prev_big = big;
#pragma unroll
      for (int r = 0; r < R; ++r) Ka[r] = Kb[r];
#pragma unroll
      for (int l = 0; l < LMAX; ++l) Pa[l] = Pb[l];
    }
  }
  return rows;
}

template <int R, int LMAX>
NDPP_HD int mu_gauss_task(const FgBatch& B, int level, int base, int t) {
  unsigned rows = B.t_gl[t];
  if (!rows) return 0;
  const bool near = (rows & kGaussNear) != 0;
  const int cert_depth = near ? B.gl_cert_depth_near : B.gl_cert_depth;
  rows &= ~kGaussNear;
  int n_node, slot;
  if (level == 0) { n_node = t / 5; slot = t - 5 * n_node; }
  else { n_node = base + (t >> 1); slot = 1 + 2 * (t & 1); }
  const unsigned mask = (unsigned)B.node_info[4 * n_node + 0];
#pragma unroll
  for (int r = 0; r < R; ++r)
    if (!(mask & (((1u << kRowBits) - 1u) << (r * kRowBits)))) rows &= ~(1u << r);   // nothing to do for the row
  if (rows == 0) { B.t_gl[t] = 0; return 0; }
  const int job = B.node_job(n_node);
  const double Ein = B.job_ein[job];
  const double Eout = fg_slot_point(B.node_a[n_node], B.node_b[n_node], slot);
  const FgPair q = make_pair(B.A_of(job), B.kT_of(job), Ein, Eout);
  FRows f;
  f.off = 8u * (unsigned)B.M * (unsigned)B.job_row[(size_t)job * R];
  const FView<R> fv = f_view<R>(B.f_tab, f, B.M);
  const double a = B.t_mulo[t], b = B.t_muhi[t];
  const PnConsts pk = make_pn_consts();
  rows = mu_gauss_certify<R, LMAX>(B, q, fv, t, mask, rows, cert_depth, pk);
  if (rows == 0) {
    B.t_gl[t] = 0;
    return 4 * ((1 << cert_depth) - 1);
  }
  // the rule with n panels against the one with n / 2, n doubled until every row left agrees (or
  // B.gl_panels is reached: the rows that still disagree are walked).  A row takes the value of the
  // first rule that agrees for IT, whatever the job's other row needs (joint == single-row bits).
  double Ic[R * LMAX], If[R * LMAX];
  int evals = 4 * ((1 << cert_depth) - 1);
  unsigned done = 0;
  auto take_agreeing_rows = [&]() {
#pragma unroll
    for (int r = 0; r < R; ++r) {
      if (!((rows & ~done) >> r & 1u)) continue;
      bool ok = true;
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
        ok = ok && (fabs(If[r * LMAX + l] - Ic[r * LMAX + l]) <= kGaussAgree * fabs(If[r * LMAX]));
      if (!ok) continue;
      done |= 1u << r;
#pragma unroll
      for (int l = 0; l < LMAX; ++l)
        if (l < B.L && (mask & chan_bit(r, l))) B.F(slot, r * B.L + l, n_node) = If[r * LMAX + l];
    }
  };
  if (near && B.gl_graded > 0) {
    // next to the peak: the graded rule against itself with every panel halved
    gauss_graded<R, LMAX>(B, q, fv, a, b, B.gl_graded, 1, pk, Ic);
    gauss_graded<R, LMAX>(B, q, fv, a, b, B.gl_graded, 2, pk, If);
    evals += 3 * (B.gl_graded + 1) * kGaussN;
    take_agreeing_rows();
    if (done == rows) {
      B.t_gl[t] = (unsigned char)rows;
      return evals;
    }
  }
  gauss_composite<R, LMAX>(B, q, fv, a, b, 4, pk, Ic);
  evals += 4 * kGaussN;
  for (int n = 8;; n *= 2) {
    gauss_composite<R, LMAX>(B, q, fv, a, b, n, pk, If);
    evals += n * kGaussN;
    take_agreeing_rows();
    if (done == rows || 2 * n > B.gl_panels) break;
#pragma unroll
    for (int k = 0; k < R * LMAX; ++k) Ic[k] = If[k];
  }
  rows = done;
  B.t_gl[t] = (unsigned char)rows;
  return evals;
}
#endif

// -----------------------------------------------------------------------------
// Stage 2 (mu): the inner adaptive Simpson integral, all channels jointly.
// -----------------------------------------------------------------------------
// What a lane keeps of the integral it walks.  Per channel (row r, order l) only the value
// at the left end of the current node (fa) and the running sums: the values at the midpoint
// and the right end and the coarse estimate S are rebuilt from the carried kernel values Xc,
// Xb and the parent's weight wp at every visit, and fa of a resumed right sibling from the
// kernel value at the right end of the node just left -- the same products of the same
// operands as when they were first formed, hence the same bits.  That keeps the push path
// and the pop path of a visit short (the lanes of a wave take both in every iteration) and
// the state small enough for two rows at L = 6.
// Channel (r, l) is order l of row r, mask bit chan_bit(r, l).
template <int R, int LMAX>
struct MuLane {
  static constexpr int NCH = R * LMAX;
  static constexpr unsigned kChanMask = []() {
    unsigned m = 0;
    for (int r = 0; r < R; ++r) m |= (LMAX >= 32 ? ~0u : ((1u << LMAX) - 1u)) << (r * kRowBits);
    return m;
  }();
  FgPair q;
  FRows f;                 // one row of f_tab, or the job's two (adjacent) rows
  double a, b;             // the current node
  double wp;               // weight of its coarse estimate: h/6 at the root (freegas.F90:505),
                           // the parent's h/12 below (:541)
  double Xc[R], Xb[R];     // kernel values of each row at the midpoint and at b
  double fa[NCH];          // f at the left end, per channel
  double acc[NCH];         // sum of the current segment's leaves ...
  double tot[NCH];         // ... and of the finished segments, left to right
  // split mode: this lane walks only the subtree of depth-kSplitLog2 node `path_bits`;
  // path_left = ancestors still to pass; own_from = depth from which accepted leaves
  // on the way down are this lane's (it is the left-most lane below them)
  int path_left, own_from;
  unsigned path_bits;
  bool own_pending;
  unsigned slot_path;  // split mode: the turns (0 left, 1 right) taken down to depth kSplitLog2, most
                       // significant first = the segment slot the running sum belongs to
  int task;          // index of the integral
  unsigned chans;    // the channels this walk integrates (the node's, less the rows the Gauss stage took)
  unsigned mask;     // channels still refining at the current node
  unsigned pending;  // depths that hold a stacked right sibling
  int depth;
  int node, slot;    // where the result goes
  unsigned visits, ovisits;
};

template <int R, int LMAX>
NDPP_HD void mu_tot_zero(MuLane<R, LMAX>& s) {
#pragma unroll
  for (int ch = 0; ch < R * LMAX; ++ch) s.tot[ch] = 0.0;
}

// Per-lane stack of right siblings, direct-mapped by depth.  An entry is what
// cannot be recomputed bit-exactly when the sibling is resumed: its right end
// b, the parent's h/12, every row's K(b) and K(e) [e = the sibling's midpoint], and the
// channels that refine into it.
template <int R>
struct HostMuStack {
  double b[kMaxLevels], w[kMaxLevels], Xb[kMaxLevels][R], Xe[kMaxLevels][R];
  unsigned m[kMaxLevels];
  NDPP_HD void push(int d, double b_, double w_, const double* Xb_, const double* Xe_, unsigned m_) {
    b[d] = b_; w[d] = w_; m[d] = m_;
    for (int r = 0; r < R; ++r) { Xb[d][r] = Xb_[r]; Xe[d][r] = Xe_[r]; }
  }
  NDPP_HD void pop(int d, double& b_, double& w_, double* Xb_, double* Xe_, unsigned& m_) const {
    b_ = b[d]; w_ = w[d]; m_ = m[d];
    for (int r = 0; r < R; ++r) { Xb_[r] = Xb[d][r]; Xe_[r] = Xe[d][r]; }
  }
};

NDPP_HD int highest_bit(unsigned x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return 31 - __clz((int)x);
#else
  return 31 - __builtin_clz(x);
#endif
}

NDPP_HD int popcount32(unsigned x) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __popc(x);
#else
  return __builtin_popcount(x);
#endif
}

template <int R, int LMAX>
NDPP_HD void mu_init(const FgBatch& B, int level, int base, int t, MuLane<R, LMAX>& s) {
  int n, slot;
  fg_task_decode(B, level, base, t, n, slot);
  s.node = n;
  s.slot = slot;
  s.mask = (unsigned)B.node_info[4 * n + 0] & MuLane<R, LMAX>::kChanMask;
  s.pending = 0;
  s.depth = 0;
  s.visits = 0;
  s.ovisits = 0;
#pragma unroll
  for (int ch = 0; ch < R * LMAX; ++ch) s.acc[ch] = 0.0;   // (the caller zeroes the segment totals)
  s.path_left = 0; s.own_from = 0; s.path_bits = 0; s.own_pending = false; s.slot_path = 0;
  s.task = t;
  s.chans = s.mask;
  if (s.mask == 0) return;
  const int rec = B.rec_index(level, base, n, slot);
  if (B.t_gl) {                                             // rows done by the Gauss rule (mu_gauss_task)
    const unsigned rows = B.t_gl[rec];
#pragma unroll
    for (int r = 0; r < R; ++r)
      if (rows >> r & 1u) s.mask &= ~(((1u << kRowBits) - 1u) << (r * kRowBits));
    s.chans = s.mask;
    if (s.mask == 0) return;
  }
  const int job = B.node_job(n);
  const double Ein = B.job_ein[job];
  const double Eout = fg_slot_point(B.node_a[n], B.node_b[n], slot);
  s.q = make_pair(B.A_of(job), B.kT_of(job), Ein, Eout);
  // (a two-row job's rows are row_lo and row_lo + 1: make_jobs_kernel)
  s.f.off = 8u * (unsigned)B.M * (unsigned)B.job_row[(size_t)job * R];
  s.a = B.t_mulo[rec];
  s.b = B.t_muhi[rec];
  double Xa[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    Xa[r] = B.tX(0, r, rec);
    s.Xb[r] = B.tX(1, r, rec);
    s.Xc[r] = B.tX(2, r, rec);
  }
  const double h = s.b - s.a;
  s.wp = h / 6.0;
  double Pa[LMAX];
  pn_all<LMAX>(s.a, Pa, make_pn_consts());
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int l = 0; l < LMAX; ++l) s.fa[r * LMAX + l] = Xa[r] * Pa[l];
}

// One node of the joint inner tree (adaptiveSimpsonsAux_mu, freegas.F90:
// 533-551).  Returns false when the integral is finished.
//
// Instruction order matters here (2 waves per SIMD, FP64 latency, L1/LDS
// latency): the table values of both new points and the top stack entry are
// requested first, the two long exp/rsqrt chains run interleaved, and the
// loaded values are consumed last.
// kPath = false compiles the split-mode path following out (the hot instantiation of
// the device kernel: a level in single-lane mode never has path_left / own_pending set).
template <int R, int LMAX, class Stack, bool kPath = true>
NDPP_HD bool mu_step(const FgBatch& B, MuLane<R, LMAX>& s, Stack& st, const PnConsts& pk) {
  if (kPath && s.own_pending && s.depth == s.own_from) {
    // split mode: from here on accepted leaves belong to this lane's segment
    s.own_pending = false;
#pragma unroll
    for (int ch = 0; ch < R * LMAX; ++ch) s.acc[ch] = 0.0;
    mu_tot_zero(s);
  }
  const double c = 0.5 * (s.a + s.b);
  const double h = s.b - s.a;
  const double d = 0.5 * (s.a + c);
  const double e = 0.5 * (c + s.b);
  // (1) the sibling that would be resumed if this node turns out all-leaf.  Read
  // unconditionally: without a pending sibling the deepest level's slot (always a valid,
  // LDS-resident slot of this lane) is read and its content ignored
  const int dj_top = s.pending ? highest_bit(s.pending) : (B.mu_its > 0 ? B.mu_its - 1 : 0);
  int dj = dj_top;
  double bj, wj, Xbj[R], Xej[R];
  unsigned mj;
  st.pop(dj, bj, wj, Xbj, Xej, mj);
  // (2) every row's kernel value at the two new points
  double Kd[R], Ke[R];
#if NDPP_FAST
  {
    FvLoad fvd[R], fve[R];
    const FView<R> fv = f_view<R>(B.f_tab, s.f, B.M);
    fg_fval_load_rows<R>(B.grid, fv, d, fvd);
    fg_fval_load_rows<R>(B.grid, fv, e, fve);
#if defined(__HIP_DEVICE_COMPILE__)
    // the table reads are in flight while the two exp / rsqrt chains run: left to itself the
    // scheduler sinks them to their first use and the wave waits out the cache latency there,
    // in every visit
    __builtin_amdgcn_sched_barrier(0);
#endif
    double Ed, Ee;
    fg_E2(s.q, d, e, Ed, Ee);
#pragma unroll
    for (int r = 0; r < R; ++r) {
      Kd[r] = (s.q.C1 * fg_fval_use(fvd[r])) * Ed;
      Ke[r] = (s.q.C1 * fg_fval_use(fve[r])) * Ee;
    }
  }
  const double w = h * (1.0 / 12.0);
  // Simpson's weights folded into the kernel values (exact scalings), so that every term of a
  // channel's two sums is ONE fused multiply-add of a kernel value and a Legendre polynomial
  double Xc2[R], Xc4[R], Kd4[R], Ke4[R];
#pragma unroll
  for (int r = 0; r < R; ++r) {
    Xc2[r] = 2.0 * s.Xc[r]; Xc4[r] = 4.0 * s.Xc[r];
    Kd4[r] = 4.0 * Kd[r]; Ke4[r] = 4.0 * Ke[r];
  }
#else
  fg_K_rows_pair<R>(s.q, B.grid, f_view<R>(B.f_tab, s.f, B.M), d, e, Kd, Ke);      // (ndpp_math.h: both points in one block)
  const double w = div_by<12>(h);      // == h / 12.0 (ndpp_math.h)
#endif
  // eps halves per level (:548); 15*eps as in :544.  At the depth limit every channel accepts
  // (:544 `its <= 0`): the threshold is infinite there, and the test is written "not greater" so
  // that a NaN accepts too instead of refining past the limit (it reaches the sum and the row's
  // NDPP_ST_NONFINITE either way).
  const bool bottom = (B.mu_its - s.depth) <= 0;
  const double eps15 = bottom ? 1.7976931348623157e308 * 2.0 : 15.0 * ldexp(B.mu_tol, -s.depth);
  double Pd[LMAX], Pc[LMAX], Pe[LMAX], Pb[LMAX];
  pn_all<LMAX>(d, Pd, pk);
  pn_all<LMAX>(c, Pc, pk);
  pn_all<LMAX>(e, Pe, pk);
  pn_all<LMAX>(s.b, Pb, pk);
  unsigned refine = 0;
  // Blocks of kMuBlock Legendre orders, skipped by the whole wave when no lane has one of them active
  // in any row (the tasks of a level are sorted by mask, fg_task_decode).  The rows of a job
  // share the block: their trees nearly coincide, P_l at the four points is formed once, and
  // the two independent chains overlap; a row that is not active discards its results.
#pragma unroll
  for (int l0 = 0; l0 < LMAX; l0 += kMuBlock) {
    unsigned any = 0;
#pragma unroll
    for (int l = l0; l < l0 + kMuBlock && l < LMAX; ++l)
#pragma unroll
      for (int r = 0; r < R; ++r) any |= s.mask & chan_bit(r, l);
    if (!any) continue;
#pragma unroll
    for (int l = l0; l < l0 + kMuBlock && l < LMAX; ++l) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int ch = r * LMAX + l;
        constexpr bool kAlone = (R == 1 && kMuBlock == 1);   // the block is this channel's own
        const bool active = kAlone || (s.mask & chan_bit(r, l)) != 0;
        const double fa = s.fa[ch];
#if NDPP_FAST
        // The two estimates from shared partial sums, their difference with one rounding:
        // s1 = fa + 4 fc + fb, s2 = fa + 4 fd + 2 fc + 4 fe + fb with f_x = K(x) P_l(x), five
        // fused multiply-adds; S = wp s1 (the parent's estimate of this half), S2 = w s2.
        // (Spelling S2 - S out as w (4 (fd + fe) - (fa + fb) - 6 fc) was measured in round 2: it
        // accepts 1.3 % more nodes than the reference's cancelling difference -- a bias, not noise.)
        const double T = fma(s.Xb[r], Pb[l], fa);
        const double s1 = fma(Xc4[r], Pc[l], T);
        const double s2 = fma(Ke4[r], Pe[l], fma(Kd4[r], Pd[l], fma(Xc2[r], Pc[l], T)));
        const double S = s.wp * s1;
        const double dS = fma(w, s2, -S);
        // the accepted value S2 + (S2 - S) / 15 (freegas.F90:545) as S + (16/15) (S2 - S): the same
        // number, one operation (it only enters the sums; the difference above is what decides)
        const double v = fma(dS, 16.0 / 15.0, S);
#else
        const double fd = Kd[r] * Pd[l];
        const double fc = s.Xc[r] * Pc[l];
        const double fe = Ke[r] * Pe[l];
        const double fb = s.Xb[r] * Pb[l];
        // (the quotient by 15 and h / 12 as x RN(1/C) plus one exact-residual correction:
        // bit-identical to the division, ndpp_math.h div_by)
        const double S = opaque(simpson(s.wp, fa, fc, fb));   // the parent's estimate of this half
        const double S2 = simpson(w, fa, fd, fc) + simpson(w, fc, fe, fb);
        const double dS = S2 - S;
        // The accepted value only enters sums (which are not the Fortran's tree-shaped ones anyway);
        // the quantities that DECIDE -- S, S2 and their difference -- are the reference's
        // operations above.  One fused multiply-add instead of the exact quotient by 15 and an
        // addition: all-strict sweeps unchanged at 3.7e-16 / 5.5e-16.
        const double v = fma(dS, 1.0 / 15.0, S2);
#endif
        const bool leaf = !(fabs(dS) > eps15);
        if (kAlone) {
          if (leaf) s.acc[ch] = s.acc[ch] + v;
          else refine |= chan_bit(r, l);
        } else {
#if defined(__HIP_DEVICE_COMPILE__)
          // Lane masks as scalars: am = the channel is active, lm = its test accepts (or the depth
          // limit is reached).  The sum of a taken leaf (am & lm) is added in place under that mask;
          // the channel's bit of `refine` is selected by am & ~lm.  One bit test, one comparison,
          // one select per channel -- spelled with booleans the compiler tests the bit twice (once
          // for the ballot, once for the select) and rebuilds the masks through three scalar ORs.
          {
            const unsigned long long am = __builtin_amdgcn_ballot_w64((s.mask & chan_bit(r, l)) != 0);
            const unsigned long long lm = __builtin_amdgcn_ballot_w64(!(fabs(dS) > eps15));
            const unsigned long long tm = am & lm, rm = am & ~lm;
            unsigned long long sv;
            unsigned rb;
            asm("s_and_saveexec_b64 %[sv], %[tm]\n\t"
                "v_add_f64 %[a], %[a], %[v]\n\t"
                "s_mov_b64 exec, %[sv]"
                : [sv] "=&s"(sv), [a] "+v"(s.acc[ch])
                : [tm] "s"(tm), [v] "v"(v)
                : "scc");
            asm("v_cndmask_b32_e64 %[rb], 0, %[bit], %[rm]" : [rb] "=v"(rb) : [bit] "v"(chan_bit(r, l)), [rm] "s"(rm));
            refine |= rb;
          }
#else
          s.acc[ch] = (active && leaf) ? s.acc[ch] + v : s.acc[ch];
          refine |= (active && !leaf) ? chan_bit(r, l) : 0u;
#endif
        }
      }
    }
  }
  s.visits += 1;
  s.ovisits += (unsigned)popcount32(s.mask);
  bool resume = false;
  if (refine) {
    bool go_right = false;
    if (kPath && s.path_left > 0) {
      // split mode, still above the lane's own subtree: follow the path instead of
      // walking both children (the other child belongs to other lanes)
      s.path_left -= 1;
      go_right = ((s.path_bits >> s.path_left) & 1u) != 0;
      if (!go_right) {
        s.b = c;
#pragma unroll
        for (int r = 0; r < R; ++r) { s.Xb[r] = s.Xc[r]; s.Xc[r] = Kd[r]; }
        s.wp = w;
        s.mask = refine;
        s.depth += 1;
        return true;
      }
    }
    if (go_right) {
      // as if the left child had been walked and the right sibling popped right away
      dj = s.depth; bj = s.b; wj = w; mj = refine;
#pragma unroll
      for (int r = 0; r < R; ++r) { Xbj[r] = s.Xb[r]; Xej[r] = Ke[r]; s.Xb[r] = s.Xc[r]; }
      s.b = c;          // the resume below takes its left end (and the kernel value there) from here
      resume = true;
    } else {
      // left child (a, c); the right one waits on the stack
      st.push(s.depth, s.b, w, s.Xb, Ke, refine);
      s.pending |= 1u << s.depth;
      s.b = c;
#pragma unroll
      for (int r = 0; r < R; ++r) { s.Xb[r] = s.Xc[r]; s.Xc[r] = Kd[r]; }
      s.wp = w;
      s.mask = refine;
      s.depth += 1;
      return true;
    }
  }
  if (resume || s.pending) {
    s.pending &= ~(1u << dj);
    if (dj + 1 <= kSplitLog2) {
      // a new segment starts: close the running one (see kSplitLog2)
      if constexpr (kPath) {
        // split walk: every segment of the item goes to its own slot of the integral (a lane
        // still above the first node it owns has nothing of its own yet)
        if (!s.own_pending) {
          const unsigned nmask = (unsigned)B.node_info[4 * s.node + 0];
#pragma unroll
          for (int r = 0; r < R; ++r)
#pragma unroll
            for (int l = 0; l < LMAX; ++l)
              if (nmask & chan_bit(r, l))
                B.seg[((size_t)s.task * kSplit + s.slot_path) * B.nch() + r * B.L + l] = s.acc[r * LMAX + l];
        }
#pragma unroll
        for (int ch = 0; ch < R * LMAX; ++ch) s.acc[ch] = 0.0;
        s.slot_path = (s.slot_path & ~((1u << (kSplitLog2 - dj)) - 1u)) | (1u << (kSplitLog2 - dj - 1));
      } else {
#pragma unroll
        for (int ch = 0; ch < R * LMAX; ++ch) {
          s.tot[ch] = s.tot[ch] + s.acc[ch];
          s.acc[ch] = 0.0;
        }
      }
    }
    // the node just finished is the right-most leaf of sibling j's left neighbour, so its b
    // IS c_j and its Xb the kernel value there: f(c_j) = Xb * P_l(c_j) is the product that
    // was formed when c_j was first evaluated
    s.a = s.b;
    {
      // P_l at the new left end: it is this node's right end (the values formed above, same
      // argument, same bits) -- unless the split walk just turned right at this node (`resume`),
      // where s.b has become its midpoint
      double Pa[LMAX];
      if (kPath && resume) {
        pn_all<LMAX>(s.a, Pa, pk);
      } else {
#pragma unroll
        for (int l = 0; l < LMAX; ++l) Pa[l] = Pb[l];
      }
#pragma unroll
      for (int r = 0; r < R; ++r)
#pragma unroll
        for (int l = 0; l < LMAX; ++l) s.fa[r * LMAX + l] = s.Xb[r] * Pa[l];
    }
    s.b = bj;
#pragma unroll
    for (int r = 0; r < R; ++r) { s.Xb[r] = Xbj[r]; s.Xc[r] = Xej[r]; }
    s.wp = wj;
    s.mask = mj;
    s.depth = dj + 1;
    return true;
  }
  return false;
}

template <int R, int LMAX>
NDPP_HD void mu_finish(const FgBatch& B, const MuLane<R, LMAX>& s, bool split = false) {
  const unsigned mask = s.chans;
#pragma unroll
  for (int r = 0; r < R; ++r)
#pragma unroll
    for (int l = 0; l < LMAX; ++l)
      if (mask & chan_bit(r, l)) {
        if (split) {
          // the item's last segment (earlier ones went out as they were finished, mu_step); a lane
          // that never reached a node of its own (everything above it was accepted) leaves its
          // slots at the zero they were set to
          if (!s.own_pending)
            B.seg[((size_t)s.task * kSplit + s.slot_path) * B.nch() + r * B.L + l] = s.acc[r * LMAX + l];
          continue;
        }
        B.F(s.slot, r * B.L + l, s.node) = s.tot[r * LMAX + l] + s.acc[r * LMAX + l];
      }
}

// split mode: the nt * kSplit work items of a level are handed out heaviest first.  Item t is
// item t / nt of integral t % nt; the items of an integral -- its depth-kSplitLog2 nodes -- are
// ordered from the peak of the kernel outwards, alternately above and below it: the exponent
// -(alpha + beta)^2 / (4 alpha) is largest at alpha = |beta|, i.e. at mu* = (p - |beta|) / q with
// alpha = p - q mu, and the adaptive refinement concentrates there.  The peak's pieces of all
// integrals start together at the beginning of the level and what is left for its end are the
// cheap far nodes.  (Results do not depend on the order: every segment of every integral has its
// own slot.)
template <int R, int LMAX>
NDPP_HD void mu_init_split(const FgBatch& B, int level, int base, int t, MuLane<R, LMAX>& s) {
  const int nt = B.n_mu_tasks(level);
  const int i = t % nt, rank = t / nt;
  mu_init<R, LMAX>(B, level, base, i, s);
  int jp = 0;
  if (s.mask != 0) {
#if NDPP_FAST
    const double pa = s.q.p, qa = s.q.q;
#else
    const double pa = s.q.EpE / s.q.AkT, qa = 2.0 * s.q.s2 / s.q.AkT;
#endif
    const double x = ((pa - fabs(s.q.beta)) / qa - s.a) / (s.b - s.a) * (double)kSplit;
    jp = (x > 0.0) ? (x < (double)(kSplit - 1) ? (int)x : kSplit - 1) : 0;   // (NaN: 0)
  }
  // the rank-th node counted from the peak outwards
  int lo = jp - 1, hi = jp, cur = hi;
  for (int k = 0; k <= rank; ++k) {
    if (((k & 1) == 0 && hi < kSplit) || lo < 0) cur = hi++;
    else cur = lo--;
  }
  s.path_left = kSplitLog2;
  s.path_bits = (unsigned)cur;
  s.slot_path = 0;
  // the item is the left-most one below an ancestor at depth a iff its low kSplitLog2 - a index
  // bits are zero
  int tz = 0;
  while (tz < kSplitLog2 && !(((unsigned)cur >> tz) & 1u)) ++tz;
  s.own_from = kSplitLog2 - tz;
  s.own_pending = (s.own_from != 0);
}

// split mode: F = ((0 + seg_0) + seg_1) + ... of every integral of the level
NDPP_HD void fg_mu_combine_task(const FgBatch& B, int level, int base, int t) {
  int n, slot;
  fg_task_decode(B, level, base, t, n, slot);
  const unsigned mask = (unsigned)B.node_info[4 * n + 0];
  if (mask == 0) return;
  const unsigned gl_rows = B.t_gl ? B.t_gl[B.rec_index(level, base, n, slot)] : 0u;   // done by the Gauss rule
  const int nch = B.nch();
  for (int r = 0; r < B.R; ++r)
    for (int l = 0; l < B.L; ++l)
      if ((mask & chan_bit(r, l)) && !(gl_rows >> r & 1u)) {
        const int ch = r * B.L + l;
        double tot = 0.0;
        for (int j = 0; j < kSplit; ++j) tot = tot + B.seg[((size_t)t * kSplit + j) * nch + ch];
        B.F(slot, ch, n) = tot;    // (slots nobody wrote are zero: x + 0.0 == x)
      }
}

// -----------------------------------------------------------------------------
// Stage 3 (node): one outer-tree node, all channels jointly
// (adaptiveSimpsons_Eout / _Aux_Eout, freegas.F90:593-643).
// Children are appended to the next level with one atomic per node.
// -----------------------------------------------------------------------------
struct HostAtomics {
  static int add(int* p, int v) { int o = *p; *p += v; return o; }
};

template <class Atomics>
NDPP_HD void fg_node_process(const FgBatch& B, int level, int base, int i) {
  const int n = base + i;
  const unsigned mask = (unsigned)B.node_info[4 * n + 0];
  if (mask == 0) return;
  const int depth = B.node_depth(n);
  const double a = B.node_a[n], b = B.node_b[n];
  const double c = 0.5 * (a + b);
  const double h = b - a;
  const double w = h / 12.0;
  const double eps15 = 15.0 * ldexp(B.eout_tol, -depth);
  const bool bottom = (B.eout_its - depth) <= 0;
  unsigned refine = 0;
  // Simpson estimate of one half: the children inherit it as their coarse estimate (:541), so
  // it is evaluated by this one expression both times (no per-channel arrays: they would
  // live in scratch memory)
  auto half = [w](double f0, double f1, double f2) { return w * (f0 + 4.0 * f1 + f2); };
  for (int r = 0; r < B.R; ++r)
    for (int l = 0; l < B.L; ++l) {
      if (!(mask & chan_bit(r, l))) continue;
      const int ch = r * B.L + l;
      const double Fa = B.F(0, ch, n), Fd = B.F(1, ch, n), Fc = B.F(2, ch, n),
                   Fe = B.F(3, ch, n), Fb = B.F(4, ch, n);
      double S = B.S(ch, n);
      if (depth == 0) S = (h / 6.0) * (Fa + 4.0 * Fc + Fb);  // :593
      const double S2 = half(Fa, Fd, Fc) + half(Fc, Fe, Fb);
      if (bottom || (fabs(S2 - S) <= eps15)) {
        B.S(ch, n) = S2 + (S2 - S) / 15.0;  // the node's value for this channel
      } else {
        refine |= chan_bit(r, l);
      }
    }
  B.node_info[4 * n + 3] = (int)refine;
  if (!refine) return;
  const int pos = Atomics::add(&B.lvl_cnt[level + 1], 2);
  const int left = base + B.lvl_cnt[level] + pos;  // next level starts there
  if (left + 2 > B.ncap) {
    *B.overflow = 1;
    B.node_info[4 * n + 3] = 0;
    return;
  }
  B.node_info[4 * n + 1] = left;
  const int job = B.node_job(n);
  for (int k = 0; k < 2; ++k) {
    const int m = left + k;
    B.node_a[m] = k ? c : a;
    B.node_b[m] = k ? b : c;
    B.node_info[4 * m + 0] = (int)refine;
    B.node_info[4 * m + 1] = -1;
    B.node_info[4 * m + 2] = job | ((depth + 1) << 26);
    B.node_info[4 * m + 3] = 0;
    for (int r = 0; r < B.R; ++r)
      for (int l = 0; l < B.L; ++l) {
        if (!(refine & chan_bit(r, l))) continue;
        const int ch = r * B.L + l;
        const double f0 = B.F(k ? 2 : 0, ch, n), f1 = B.F(k ? 3 : 1, ch, n), f2 = B.F(k ? 4 : 2, ch, n);
        B.F(0, ch, m) = f0;
        B.F(2, ch, m) = f1;
        B.F(4, ch, m) = f2;
        B.S(ch, m) = half(f0, f1, f2);
      }
  }
}

// Bottom-up: value of an internal node = left + right (freegas.F90:639-642)
NDPP_HD void fg_reduce_node(const FgBatch& B, int base, int i) {
  const int n = base + i;
  const unsigned refine = (unsigned)B.node_info[4 * n + 3];
  if (!refine) return;
  const int left = B.node_info[4 * n + 1];
  for (int r = 0; r < B.R; ++r)
    for (int l = 0; l < B.L; ++l)
      if (refine & chan_bit(r, l)) {
        const int ch = r * B.L + l;
        B.S(ch, n) = B.S(ch, left) + B.S(ch, left + 1);
      }
}

// -----------------------------------------------------------------------------
// Stage 4 (assemble): integrate_freegas_leg's per-group sums, the |x|<1e-18
// flush and the P0 normalisation (freegas.F90:80-145), in its operation order;
// one (job, row) = one call of the reference routine.
// -----------------------------------------------------------------------------
NDPP_HD void fg_assemble_call(const FgBatch& B, int call) {
  const int L = B.L, G = B.G;
  const int job = call / B.R, r = call - job * B.R;
  double* out = B.raw + (size_t)call * G * L;
  double p0 = 0.0;
  for (int g = 0; g < G; ++g) {
    const int root = (job * G + g) * kSegPerGroup;
    for (int l = 0; l < L; ++l) {
      const int ch = r * L + l;
      // slots the reference does not integrate hold 0.0, and x + 0.0 == x
      double v = B.S(ch, root + 0) + B.S(ch, root + 1);
      v = v + B.S(ch, root + 2);
      v = v + B.S(ch, root + 3);
      v = v + B.S(ch, root + 4);
      out[g * L + l] = v;
    }
    p0 = p0 + out[g * L + 0];
    for (int l = 0; l < L; ++l)
      if (fabs(out[g * L + l]) < 1.0E-18) out[g * L + l] = 0.0;
  }
  for (int k = 0; k < G * L; ++k) out[k] = out[k] / p0;
}

}  // inline namespace
}  // namespace ndpp
