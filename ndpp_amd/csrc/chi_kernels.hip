// chi_kernels.hip -- group integrals of the fission energy spectra (chi):
// calc_chi's loop over incoming energies (chi.F90:124-159) with ChiData%beta,
// %prob and %integrate (chidata_header.F90:139-493: ACE laws 4/61 by CDF
// differences, 7 Maxwell, 9 evaporation, 11 Watt) and nu_total / nu_delayed
// (fission.F90:18-103).  One thread per incoming energy; the work is closed
// forms per group (microseconds per nuclide), so the kernel exists to keep the
// whole path on the device, not for speed.  Built with the reference's IEEE
// operation order (-DNDPP_FAST=0 -ffp-contract=off); differs from the Fortran
// only through erf/exp/sinh/log last-bit differences.  Reference quirks are
// reproduced and marked (sic).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"
#include "ndpp_math.h"

#if NDPP_FAST
#error "chi_kernels.hip must be compiled with -DNDPP_FAST=0 -ffp-contract=off"
#endif

namespace ndpp {
namespace {

constexpr int NU_POLYNOMIAL = 1, NU_TABULAR = 2;


__device__ double chi_interp(int interp, double x, double x0, double x1, double y0, double y1) {
  double r;
  switch (interp) {
    case 2: r = (x - x0) / (x1 - x0); return (1 - r) * y0 + r * y1;
    case 3: r = (log(x) - log(x0)) / (log(x1) - log(x0)); return (1 - r) * y0 + r * y1;
    case 4: r = (x - x0) / (x1 - x0); return exp((1 - r) * log(y0) + r * log(y1));
    case 5: r = (log(x) - log(x0)) / (log(x1) - log(x0)); return exp((1 - r) * log(y0) + r * log(y1));
    default: return NAN;
  }
}

// interpolate_tab1_array, interpolation.F90:24-123
__device__ double chi_tab1(const double* data, double x) {
  const int n_regions = (int)data[0];
  const int loc_interp = 1 + n_regions;
  const int n_points = (int)data[loc_interp + n_regions];
  const int loc_x = loc_interp + n_regions + 1, loc_y = loc_x + n_points;
  if (x < data[loc_x]) return data[loc_y];
  else if (x > data[loc_x + n_points - 1]) return data[loc_y + n_points - 1];
  const int i = bsearch1_clamped(data + loc_x, n_points, x);
  int interp = 2;
  if (n_regions == 1) interp = (int)data[loc_interp];
  else if (n_regions > 1)
    for (int j = 1; j <= n_regions; ++j)
      if (i < data[j]) { interp = (int)data[loc_interp + j - 1]; break; }
  if (interp == 1) return data[loc_y + i - 1];
  return chi_interp(interp, x, data[loc_x + i - 1], data[loc_x + i], data[loc_y + i - 1], data[loc_y + i]);
}

// interpolate_tab1_object, interpolation.F90:132-208
__device__ double chi_tab1_obj(const ndpp_chi_spectrum& s, double v) {
  if (v < s.pv_x[0]) return s.pv_y[0];
  else if (v > s.pv_x[s.pv_n_pairs - 1]) return s.pv_y[s.pv_n_pairs - 1];
  const int i = bsearch1_clamped(s.pv_x, s.pv_n_pairs, v);
  int interp = 2;
  if (s.pv_n_regions == 1) interp = s.pv_int[0];
  else if (s.pv_n_regions > 1)
    for (int j = 0; j < s.pv_n_regions; ++j)
      if (i < s.pv_nbt[j]) { interp = s.pv_int[j]; break; }
  if (interp == 1) return s.pv_y[i - 1];
  return chi_interp(interp, v, s.pv_x[i - 1], s.pv_x[i], s.pv_y[i - 1], s.pv_y[i]);
}

__device__ double chi_nu_total(const ndpp_chi_nuclide& n, double E) {  // fission.F90:18-45
  if (n.nu_t_type == NU_POLYNOMIAL) {
    const int NC = (int)n.nu_t_data[0];
    double nu = 0.0;
    for (int i = 0; i <= NC - 1; ++i) nu = nu + n.nu_t_data[i + 1] * (i == 0 ? 1.0 : powi(E, i));
    return nu;
  } else if (n.nu_t_type == NU_TABULAR) {
    return chi_tab1(n.nu_t_data, E);
  }
  return NAN;
}

__device__ double chi_nu_delayed(const ndpp_chi_nuclide& n, double E) {  // fission.F90:90-103
  return (n.nu_d_type == NU_TABULAR) ? chi_tab1(n.nu_d_data, E) : 0.0;
}

// chi_prob, chidata_header.F90:154-215
__device__ double chi_prob(const ndpp_chi_nuclide& n, const ndpp_chi_spectrum& s, bool delayed,
                           int grp, double Ein) {
  if (delayed) {
    const double* pd = n.nu_d_precursor_data;
    int lc = 1;
    for (int j = 1; j <= n.n_precursor; ++j) {
      const int NR = (int)pd[lc];
      const int NE = (int)pd[lc + 1 + 2 * NR];
      if (j == grp) break;
      lc = lc + 2 + 2 * NR + 2 * NE + 1;
    }
    return chi_tab1(pd + lc, Ein);
  }
  int j;
  double f, prob;
  if (Ein < n.energy[0]) { j = 1; f = 0.0; }
  else if (Ein >= n.energy[n.n_grid - 1]) { j = n.n_grid - 1; f = 1.0; }
  else {
    j = bsearch1_clamped(n.energy, n.n_grid, Ein);
    f = (Ein - n.energy[j - 1]) / (n.energy[j] - n.energy[j - 1]);
  }
  if (n.energy[j - 1] == n.energy[j]) j = j + 1;
  if (j < s.threshold) prob = 0.0;
  else
    prob = ((1.0 - f) * s.sigma[j - s.threshold] + f * s.sigma[j - s.threshold + 1]) /
           ((1.0 - f) * n.fission[j - 1] + f * n.fission[j]);
  if (s.has_next && s.pv_n_regions > 0) prob = prob * chi_tab1_obj(s, Ein);  // (sic), :210
  return prob;
}

// chi_integrate, chidata_header.F90:221-493
__device__ void chi_integrate(const ndpp_chi_spectrum& s, double Ein, int G, const double* E_bins,
                              double* chis) {
  const double* d = s.data;
  for (int g = 0; g < G; ++g) chis[g] = 0.0;
  int NR = (int)d[0], NE = (int)d[1 + 2 * NR], lc;
  double T, U, I, x, x0;
  switch (s.law) {
    case 4:
    case 61: {
      bool hist = false;
      if (NR == 1 && s.law == 4) hist = (d[2] == 1);
      lc = 2 + 2 * NR;
      int iE;
      if (Ein < d[lc]) { iE = 1; x = 0.0; }
      else if (Ein >= d[lc + NE - 1]) { iE = NE - 1; x = 1.0; }
      else {
        iE = bsearch1_clamped(d + lc, NE, Ein);
        x = (Ein - d[lc + iE - 1]) / (d[lc + iE] - d[lc + iE - 1]);
      }
      if (!hist && x > 0.5) iE = iE + 1;  // nearest row, :294-298
      lc = (int)d[2 + 2 * NR + NE + iE - 1];
      const int NP = (int)d[lc + 1];
      lc = lc + 3;
      int lEout_min = lc;
      double runsum = 0.0;
      for (int g = 1; g <= G; ++g) {
        int k;
        for (k = lEout_min; k <= NP + lc - 2; ++k)
          if (d[k] > E_bins[g]) break;
        if (k == NP + lc - 1) k = k - 1;
        const double interp = (E_bins[g] - d[k - 1]) / (d[k] - d[k - 1]);
        double v = (d[k + 2 * NP - 1] + interp * (d[k + 2 * NP] - d[k + 2 * NP - 1]));
        v = v - runsum;
        runsum = runsum + v;
        chis[g - 1] = v;
        lEout_min = k;
      }
      break;
    }
    case 7:
      T = chi_tab1(d, Ein);
      lc = 2 + 2 * NR + 2 * NE;
      U = d[lc];
      if (Ein - U <= 0.0) return;
      x = (Ein - U) / T;
      I = sqrt(T * T * T) * (sqrt(0.25 * kPi) * erf(x) - x * exp(-x));
      for (int g = 0; g < G; ++g) {
        double Egp1 = E_bins[g + 1];
        if (Egp1 > Ein - U) Egp1 = U;  // (sic), :375
        double v = 0.5 * (sqrt(kPi * T) * erf(sqrt(Egp1 / T)) * exp(Egp1 / T) - 2.0 * sqrt(Egp1)) *
                   T * exp(-Egp1 / T);
        double Eg = E_bins[g];
        if (Eg > Ein - U) Eg = U;
        v = v - (0.5 * (sqrt(kPi * T) * erf(sqrt(Eg / T)) * exp(Eg / T) - 2.0 * sqrt(Eg)) * T *
                 exp(-Eg / T));
        chis[g] = v / I;
      }
      break;
    case 9:
      T = chi_tab1(d, Ein);
      lc = 2 + 2 * NR + 2 * NE;
      U = d[lc];
      x = (Ein - U) / T;
      if (Ein - U <= 0.0) return;
      for (int g = 0; g < G; ++g) {
        double Egp1 = E_bins[g + 1], Eg = E_bins[g];
        if (Egp1 > (Ein - U)) Egp1 = Ein - U;
        if (Eg > (Ein - U)) Eg = Ein - U;
        double v = (Egp1 * exp(x) + T * exp(x)) * exp(-Egp1 / T);
        v = v - (Eg * exp(x) + T * exp(x)) * exp(-Eg / T);
        chis[g] = v / (T * (x - exp(x) + 1.0));
      }
      break;
    case 11: {
      const double Wa = chi_tab1(d, Ein);
      lc = 2 + 2 * (NR + NE);
      double Wb = chi_tab1(d + lc, Ein);
      NR = (int)d[lc];
      NE = (int)d[lc + 1 + 2 * NR];
      lc = lc + 2 + 2 * (NR + NE);
      U = d[lc];
      x = (Ein - U) / Wa;
      if (Ein - U <= 0.0) return;
      x0 = Wa * Wb * 0.25;
      I = 0.25 * sqrt(kPi * powi(Wa, 3) * Wb) * exp(x0) *
              (erf(sqrt(x) - sqrt(x0)) + erf(sqrt(x) + sqrt(x0))) -
          Wa * exp(-x * sinh(Wa * Wb * x));
      Wb = sqrt(Wb);
      x = sqrt(kPi * Wa) * Wb * exp(0.25 * Wa * (Wb * Wb));
      for (int g = 0; g < G; ++g) {
        double Egp1 = E_bins[g + 1];
        if (Egp1 > U) Egp1 = U;
        double v = (-x * erf((Wa * Wb - 2.0 * sqrt(Egp1) / (2.0 * Wa))) +
                    x * erf((Wa * Wb + 2.0 * sqrt(Egp1) / (2.0 * Wa))) -
                    2.0 * (exp(2.0 * Wb * sqrt(Egp1)) * exp(-(Wa * Wb * sqrt(Egp1)) / Wa)));
        double Eg = E_bins[g];
        if (Eg > U) Eg = U;
        v = v - (-x * erf((Wa * Wb - 2.0 * sqrt(Eg) / (2.0 * Wa))) +
                 x * erf((Wa * Wb + 2.0 * sqrt(Eg) / (2.0 * Wa))) -
                 2.0 * (exp(2.0 * Wb * sqrt(Eg)) * exp(-(Wa * Wb * sqrt(Eg)) / Wa)));
        chis[g] = 0.25 * Wa * v / I;
      }
      break;
    }
    default: break;  // other laws: the reference only warns, spectrum stays zero
  }
  I = 0.0;
  for (int g = 0; g < G; ++g) I = I + chis[g];
  if (I != 1.0) {
    I = 1.0 / I;  // all-zero spectrum -> NaN (sic), :482-491
    for (int g = 0; g < G; ++g) chis[g] = chis[g] * I;
  }
}

__device__ double chi_ksum(const double* x, int n) {  // flang SUM: Kahan
  double s = 0.0, c = 0.0;
  for (int i = 0; i < n; ++i) { const double y = x[i] - c, t = s + y; c = (t - s) - y; s = t; }
  return s;
}

struct ChiDev {
  ndpp_chi_nuclide nuc;
  const ndpp_chi_spectrum* prompt;
  const ndpp_chi_spectrum* delay;
  int n_prompt, n_delay, G, NE;
  const double* e_bins;
  const double* e_grid;
  double* scratch;  // [NE][G]
  double* chi_t;    // [NE][G]
  double* chi_p;    // [NE][G]
  double* chi_d;    // [n_delay][NE][G]
};

__global__ __launch_bounds__(64) void chi_kernel(ChiDev D) {
  for (int iE = blockIdx.x * blockDim.x + threadIdx.x; iE < D.NE; iE += gridDim.x * blockDim.x) {
    const int G = D.G;
    const double Ein = D.e_grid[iE];
    double* ct = D.chi_t + (size_t)iE * G;
    double* cpm = D.chi_p + (size_t)iE * G;
    double* cp = D.scratch + (size_t)iE * G;
    for (int g = 0; g < G; ++g) { ct[g] = 0.0; cpm[g] = 0.0; }
    const double beta = chi_nu_delayed(D.nuc, Ein) / chi_nu_total(D.nuc, Ein);
    double prob = 0.0;
    for (int i = 0; i < D.n_prompt; ++i) {
      chi_integrate(D.prompt[i], Ein, G, D.e_bins, cp);
      prob = chi_prob(D.nuc, D.prompt[i], false, 0, Ein);
      for (int g = 0; g < G; ++g) ct[g] = ct[g] + prob * (1.0 - beta) * cp[g];
      for (int g = 0; g < G; ++g) cpm[g] = cpm[g] + prob * cp[g];
    }
    for (int g = 0; g < G; ++g) ct[g] = cpm[g] + prob * cpm[g];  // (sic), chi.F90:135
    for (int i = 0; i < D.n_delay; ++i) {
      double* cd = D.chi_d + ((size_t)i * D.NE + iE) * G;
      chi_integrate(D.delay[i], Ein, G, D.e_bins, cd);
      prob = chi_prob(D.nuc, D.delay[i], true, i + 1, Ein);
      for (int g = 0; g < G; ++g) ct[g] = ct[g] + prob * beta * cd[g];
    }
    double norm = chi_ksum(ct, G);
    if (norm > 0.0) for (int g = 0; g < G; ++g) ct[g] = ct[g] / norm;
    norm = chi_ksum(cpm, G);
    if (norm > 0.0) for (int g = 0; g < G; ++g) cpm[g] = cpm[g] / norm;
    for (int i = 0; i < D.n_delay; ++i) {
      double* cd = D.chi_d + ((size_t)i * D.NE + iE) * G;
      norm = chi_ksum(cd, G);
      if (norm > 0.0) for (int g = 0; g < G; ++g) cd[g] = cd[g] / norm;
    }
  }
}

#define CHI_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(NDPP_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

struct Pool {  // device copies of host arrays, freed together
  std::vector<void*> ptrs;
  ~Pool() { for (void* p : ptrs) dev_free(p); }
  template <class T>
  hipError_t up(const T* h, size_t n, const T** out) {
    *out = nullptr;
    if (!h || n == 0) return hipSuccess;
    void* p = nullptr;
    hipError_t e = dev_alloc(&p, n * sizeof(T));
    if (e != hipSuccess) return e;
    ptrs.push_back(p);
    *out = (const T*)p;
    return hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice);
  }
  template <class T>
  hipError_t alloc(size_t n, T** out) {
    void* p = nullptr;
    hipError_t e = dev_alloc(&p, std::max<size_t>(n, 1) * sizeof(T));
    if (e != hipSuccess) return e;
    ptrs.push_back(p);
    *out = (T*)p;
    return hipSuccess;
  }
};

// a TAB1 block [NR, (NBT, INT) x NR, NE, x(NE), y(NE)] starting at data[0]; returns its
// length in words, or -1 if it does not fit in n
int tab1_words(const double* data, int n) {
  if (!data || n < 2) return -1;
  const int NR = (int)data[0];
  if (NR < 0 || 2 + 2 * NR > n) return -1;
  const int NE = (int)data[1 + 2 * NR];
  if (NE < 1 || 2 + 2 * NR + 2 * NE > n) return -1;
  return 2 + 2 * NR + 2 * NE;
}

int check_spectrum(const ndpp_chi_spectrum& s, bool prompt, int idx) {
  if (!s.data || s.n_data < 4) return fail(NDPP_EINVAL, "spectrum %d: no data", idx);
  const int NR = (int)s.data[0];
  if (NR < 0 || 2 + 2 * NR > s.n_data) return fail(NDPP_EINVAL, "spectrum %d: bad NR", idx);
  if ((s.law == 4 || s.law == 61) && NR > 1)  // reference: fatal_error, chidata_header.F90:275
    return fail(NDPP_EINVAL, "spectrum %d: multiple interpolation regions not supported", idx);
  if (prompt && (!s.sigma || s.n_sigma < 2 || s.threshold < 1))
    return fail(NDPP_EINVAL, "prompt spectrum %d needs sigma and threshold", idx);
  // every index chi_integrate forms from the block must stay inside it
  const int NE = (int)s.data[1 + 2 * NR];
  const int head = 2 + 2 * NR;
  if (NE < 1 || head + NE > s.n_data) return fail(NDPP_EINVAL, "spectrum %d: bad NE=%d", idx, NE);
  if (s.law == 4 || s.law == 61) {
    if (NE < 2 || head + 2 * NE > s.n_data)
      return fail(NDPP_EINVAL, "spectrum %d: law %d table header truncated", idx, s.law);
    for (int k = 0; k < NE; ++k) {
      const int lc = (int)s.data[head + NE + k];
      if (lc < 0 || lc + 2 > s.n_data)
        return fail(NDPP_EINVAL, "spectrum %d: locator of incoming energy %d out of range", idx, k + 1);
      const int NP = (int)s.data[lc + 1];
      if (NP < 2 || lc + 2 + 3 * NP > s.n_data)
        return fail(NDPP_EINVAL, "spectrum %d: table of incoming energy %d truncated", idx, k + 1);
    }
  } else if (s.law == 7 || s.law == 9) {      // TAB1 of T(E), then U
    if (head + 2 * NE + 1 > s.n_data) return fail(NDPP_EINVAL, "spectrum %d: law %d data truncated", idx, s.law);
  } else if (s.law == 11) {                   // TAB1 a(E), TAB1 b(E), U
    const int o = head + 2 * NE;
    if (o + 2 > s.n_data) return fail(NDPP_EINVAL, "spectrum %d: Watt data truncated", idx);
    const int NR2 = (int)s.data[o];
    if (NR2 < 0 || o + 2 + 2 * NR2 > s.n_data) return fail(NDPP_EINVAL, "spectrum %d: Watt b(E) header", idx);
    const int NE2 = (int)s.data[o + 1 + 2 * NR2];
    if (NE2 < 1 || o + 2 + 2 * NR2 + 2 * NE2 + 1 > s.n_data)
      return fail(NDPP_EINVAL, "spectrum %d: Watt data truncated", idx);
  }
  if (s.pv_n_regions < 0 || s.pv_n_pairs < 0 ||
      (s.has_next && s.pv_n_regions > 0 && (s.pv_n_pairs < 1 || !s.pv_x || !s.pv_y || !s.pv_nbt || !s.pv_int)))
    return fail(NDPP_EINVAL, "spectrum %d: p_valid incomplete", idx);
  return NDPP_OK;
}

}  // namespace
}  // namespace ndpp

using namespace ndpp;

extern "C" int ndpp_chi_batch(const ndpp_chi_nuclide* nuc, int n_prompt,
                              const ndpp_chi_spectrum* prompt, int n_delay,
                              const ndpp_chi_spectrum* delay, int G, const double* e_bins,
                              int n_ein, const double* e_grid, double* chi_t, double* chi_p,
                              double* chi_d) {
  if (!nuc || !prompt || n_prompt < 1 || n_delay < 0 || (n_delay > 0 && !delay))
    return fail(NDPP_EINVAL, "need a nuclide and at least one prompt spectrum");
  if (G < 1 || n_ein < 0) return fail(NDPP_EINVAL, "G=%d n_ein=%d", G, n_ein);
  if (n_ein == 0) return NDPP_OK;
  if (!e_bins || !e_grid || !chi_t || !chi_p || (n_delay > 0 && !chi_d))
    return fail(NDPP_EINVAL, "NULL array argument");
  if (nuc->n_grid < 2 || !nuc->energy || !nuc->fission || !nuc->nu_t_data)
    return fail(NDPP_EINVAL, "nuclide grid / nu data missing");
  if (nuc->nu_t_type != NU_POLYNOMIAL && nuc->nu_t_type != NU_TABULAR)
    return fail(NDPP_EINVAL, "no neutron emission data (nu_t_type=%d)", nuc->nu_t_type);  // fission.F90:27
  if (nuc->nu_t_type == NU_POLYNOMIAL) {
    if (nuc->n_nu_t < 1 || (int)nuc->nu_t_data[0] < 0 || (int)nuc->nu_t_data[0] + 1 > nuc->n_nu_t)
      return fail(NDPP_EINVAL, "nu_t polynomial: %d coefficients in %d words", (int)nuc->nu_t_data[0], nuc->n_nu_t);
  } else if (tab1_words(nuc->nu_t_data, nuc->n_nu_t) < 0) {
    return fail(NDPP_EINVAL, "nu_t table malformed");
  }
  if (nuc->nu_d_type == NU_TABULAR && tab1_words(nuc->nu_d_data, nuc->n_nu_d) < 0)
    return fail(NDPP_EINVAL, "nu_d table malformed");
  if (n_delay > 0) {   // per precursor group: lambda, TAB1 of the yield (chidata_header.F90:160-172)
    if (n_delay > nuc->n_precursor || !nuc->nu_d_precursor_data)
      return fail(NDPP_EINVAL, "%d delayed spectra but %d precursor groups", n_delay, nuc->n_precursor);
    int lc = 1;
    for (int j = 0; j < nuc->n_precursor; ++j) {
      const int w = (lc <= nuc->n_prec_data) ? tab1_words(nuc->nu_d_precursor_data + lc, nuc->n_prec_data - lc) : -1;
      if (w < 0) return fail(NDPP_EINVAL, "precursor group %d data malformed", j + 1);
      lc += w + 1;
    }
  }
  for (int i = 0; i < n_prompt; ++i) {
    int rc = check_spectrum(prompt[i], true, i);
    if (rc) return rc;
    // chi_prob reads sigma(j - threshold + 1 .. +1) for j up to n_grid - 1
    if (prompt[i].threshold > nuc->n_grid || prompt[i].n_sigma < nuc->n_grid - prompt[i].threshold + 1)
      return fail(NDPP_EINVAL, "prompt spectrum %d: sigma has %d values, needs %d", i, prompt[i].n_sigma,
                  nuc->n_grid - prompt[i].threshold + 1);
  }
  for (int i = 0; i < n_delay; ++i) { int rc = check_spectrum(delay[i], false, i); if (rc) return rc; }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");

  Pool pool;
  ChiDev D;
  D.nuc = *nuc;
  CHI_TRY(pool.up(nuc->energy, nuc->n_grid, &D.nuc.energy));
  CHI_TRY(pool.up(nuc->fission, nuc->n_grid, &D.nuc.fission));
  CHI_TRY(pool.up(nuc->nu_t_data, nuc->n_nu_t, &D.nuc.nu_t_data));
  CHI_TRY(pool.up(nuc->nu_d_data, nuc->n_nu_d, &D.nuc.nu_d_data));
  CHI_TRY(pool.up(nuc->nu_d_precursor_data, nuc->n_prec_data, &D.nuc.nu_d_precursor_data));
  std::vector<ndpp_chi_spectrum> hp(prompt, prompt + n_prompt), hd(delay, delay + n_delay);
  for (auto* vec : {&hp, &hd})
    for (auto& s : *vec) {
      CHI_TRY(pool.up(s.data, s.n_data, &s.data));
      CHI_TRY(pool.up(s.sigma, s.n_sigma, &s.sigma));
      CHI_TRY(pool.up(s.pv_nbt, s.pv_n_regions, &s.pv_nbt));
      CHI_TRY(pool.up(s.pv_int, s.pv_n_regions, &s.pv_int));
      CHI_TRY(pool.up(s.pv_x, s.pv_n_pairs, &s.pv_x));
      CHI_TRY(pool.up(s.pv_y, s.pv_n_pairs, &s.pv_y));
    }
  CHI_TRY(pool.up(hp.data(), hp.size(), &D.prompt));
  CHI_TRY(pool.up(hd.data(), hd.size(), &D.delay));
  CHI_TRY(pool.up(e_bins, G + 1, &D.e_bins));
  CHI_TRY(pool.up(e_grid, n_ein, &D.e_grid));
  const size_t nrow = (size_t)n_ein * G;
  CHI_TRY(pool.alloc(nrow, &D.scratch));
  CHI_TRY(pool.alloc(nrow, &D.chi_t));
  CHI_TRY(pool.alloc(nrow, &D.chi_p));
  CHI_TRY(pool.alloc(nrow * std::max(n_delay, 1), &D.chi_d));
  D.n_prompt = n_prompt; D.n_delay = n_delay; D.G = G; D.NE = n_ein;
  const int blocks = std::max(1, (n_ein + 63) / 64);
  GpuSpan span(nullptr, kProfChi);
  hipLaunchKernelGGL(chi_kernel, dim3(blocks), dim3(64), 0, 0, D);
  span.end();
  CHI_TRY(hipGetLastError());
  CHI_TRY(hipDeviceSynchronize());
  CHI_TRY(hipMemcpy(chi_t, D.chi_t, sizeof(double) * nrow, hipMemcpyDeviceToHost));
  CHI_TRY(hipMemcpy(chi_p, D.chi_p, sizeof(double) * nrow, hipMemcpyDeviceToHost));
  if (n_delay > 0)
    CHI_TRY(hipMemcpy(chi_d, D.chi_d, sizeof(double) * nrow * n_delay, hipMemcpyDeviceToHost));
  return NDPP_OK;
}
