// tests/hostsim/sanitize_main.cpp -- TEST INFRASTRUCTURE ONLY.
// The product's stage functions (fg_pipeline.h: per-lane explicit stacks, the lane types the
// device launches, split walk, arena overflow) driven on the CPU under AddressSanitizer and
// UndefinedBehaviorSanitizer (GPU sanitizers are not available on the pool, so
// the indexing of exactly this code is checked here).  Built by `make -C tests/hostsim sanitize`
// in both arithmetic variants; exits non-zero on the first finding.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/ndpp_hip.h"

extern "C" int hostsim_freegas_jobs(const ndpp_params* p, double A, double kT, int n_jobs, int R, const double* ein,
                                     const int* row, int n_rows, const double* f_tab, int G, const double* e_bins,
                                     int ncap, double* raw, unsigned long long* stats_out, int* lvl_cnt_out);

static ndpp_params params(int L, int M) {
  ndpp_params p;
  p.order = L; p.mu_bins = M;
  p.sab_threshold = 1e-6; p.brent_mu_thresh = 1e-6;      // constants.F90:70-100 (ndpp_default_params)
  p.adaptive_mu_tol = 1e-7; p.adaptive_eout_tol = 1e-8;
  p.adaptive_mu_its = 15; p.adaptive_eout_its = 15;
  p.ne_per_grp = 20; p.sab_epts_per_bin = 10; p.extend_pts = 50; p.inel_extend_pts = 30;
  return p;
}

static int run(const char* what, int L, int R, int G, int ncap, int expect_rc, double A, const std::vector<double>& ein) {
  const int M = 257, n = (int)ein.size();
  ndpp_params p = params(L, M);
  std::vector<double> f((size_t)3 * M);
  for (int r = 0; r < 3; ++r)
    for (int i = 0; i < M; ++i) {
      const double mu = (i == M - 1) ? 1.0 : -1.0 + i * (2.0 / (M - 1));
      f[(size_t)r * M + i] = 0.5 * (1.0 + 0.15 * r * mu);
    }
  std::vector<double> bins(G + 1);
  bins[0] = 0.0;
  for (int g = 1; g <= G; ++g) bins[g] = 1e-9 * std::pow(2e10, (double)g / G);
  std::vector<int> row((size_t)n * R);
  std::vector<double> e((size_t)n * (R == 2 ? 1 : 1));
  for (int k = 0; k < n; ++k) {
    e[k] = ein[k];
    for (int r = 0; r < R; ++r) row[(size_t)k * R + r] = (k + r) % 2 + (R == 2 ? 0 : 0);
  }
  if (R == 2) for (int k = 0; k < n; ++k) { row[2 * k] = k % 2; row[2 * k + 1] = k % 2 + 1; }
  std::vector<double> raw((size_t)n * R * G * L, 0.0);
  unsigned long long st[4] = {0, 0, 0, 0};
  int cnt[40] = {0};
  const int rc = hostsim_freegas_jobs(&p, A, 2.5301e-8, n, R, e.data(), row.data(), 3, f.data(), G, bins.data(), ncap,
                                      raw.data(), st, cnt);
  double p0 = 0.0;
  if (rc == 0) for (int g = 0; g < G; ++g) p0 += raw[(size_t)g * L];
  std::printf("%-44s rc %d  K evaluations %llu  sum_g P0 of the first row %.15f\n", what, rc, st[0], p0);
  if (rc != expect_rc) return 1;
  if (rc == 0 && !(std::fabs(p0 - 1.0) < 1e-12)) return 1;
  return 0;
}

int main() {
  int bad = 0;
  const std::vector<double> two = {2.53e-8, 4e-7};
  bad += run("single row, P3", 4, 1, 2, 200000, 0, 0.999167, two);
  bad += run("joint rows, P5 (12 channels)", 6, 2, 2, 200000, 0, 0.999167, two);
  bad += run("joint rows, P7 (16 channels)", 8, 2, 3, 200000, 0, 15.86, {1e-7});
  bad += run("single row, P10", 11, 1, 2, 200000, 0, 0.999167, {2.53e-8});
  setenv("HOSTSIM_SPLIT", "1", 1);
  bad += run("split walk, joint rows, P5", 6, 2, 2, 200000, 0, 0.999167, {2.53e-8});
  bad += run("split walk, joint rows, P7", 8, 2, 2, 200000, 0, 15.86, {1e-7});
  unsetenv("HOSTSIM_SPLIT");
  bad += run("cold heavy corner (depth limit of the inner walk)", 6, 2, 2, 400000, 0, 236.0058, {1e-10});
  bad += run("arena overflow is reported, not overrun", 6, 2, 2, 64, NDPP_EOVERFLOW, 0.999167, two);
  std::printf(bad ? "SANITIZE_FAILED\n" : "SANITIZE_OK\n");
  return bad ? 1 : 0;
}
