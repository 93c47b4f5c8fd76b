// sab_kernels.hip -- thermal S(alpha,beta) scattering moments from ACE thermal
// tables: integrate_sab_el (sab.F90:21-109), integrate_sab_inel_disc (:142-245),
// integrate_sab_inel_cont (:253-408), combine_sab_grid (:415-454); i.e. the
// Legendre path of calc_scattsab (scatt.F90:543-596).
//
// Only calc_pn and + - * / are involved, every output element is accumulated by
// one thread in the reference's order, and the TU is always built with
// -DNDPP_FAST=0 -ffp-contract=off: results are bit-identical to the Fortran.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"
#include "ndpp_math.h"

#if NDPP_FAST
#error "sab_kernels.hip must be compiled with -DNDPP_FAST=0 -ffp-contract=off"
#endif

namespace ndpp {
namespace {

constexpr int SAB_SECONDARY_EQUAL = 0, SAB_SECONDARY_SKEWED = 1, SAB_SECONDARY_CONT = 2;
constexpr int SAB_ELASTIC_DISCRETE = 3, SAB_ELASTIC_EXACT = 4;


struct SabDev {
  ndpp_sab_flat t;  // pointers are device pointers
  int NE, G, L;
  const double* ein;
  const double* e_bins;
  const double* wgt;   // [NEo] discrete-mode weights (:167-186)
  double* distro;      // [NEi][G][L] continuous-mode table integrals (:281)
  double* el;          // [NE][G][L]
  double* inel;
  double* mat;
  int* status;
};

// integrate_sab_el: thread per (E_in, order)
__global__ void sab_el_kernel(SabDev D) {
  const long tot = (long)D.NE * D.L;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < tot; q += (long)gridDim.x * blockDim.x) {
    const int l = (int)(q % D.L), i = (int)(q / D.L);
    double* row = D.el + (size_t)i * D.G * D.L;
    for (int g = 0; g < D.G; ++g) row[(size_t)g * D.L + l] = 0.0;
    const ndpp_sab_flat& t = D.t;
    if (t.threshold_elastic == 0.0) continue;
    const double Ein = D.ein[i];
    if (Ein < t.elastic_e_in[0]) continue;
    else if (Ein >= t.threshold_elastic) continue;
    const int isab = bsearch1(t.elastic_e_in, t.n_elastic_e_in, Ein);
    if (isab < 0) { if (l == 0) atomicOr(&D.status[i], NDPP_ST_RANGE); continue; }
    const double f = (Ein - t.elastic_e_in[isab - 1]) / (t.elastic_e_in[isab] - t.elastic_e_in[isab - 1]);
    if (Ein < D.e_bins[0]) continue;
    else if (Ein > D.e_bins[D.G]) continue;
    int g = bsearch1(D.e_bins, D.G + 1, Ein);
    if (g < 1) g = 1;  // NaN energy: stay inside the row
    double sig = 0.0;
    if (t.elastic_mode == SAB_ELASTIC_EXACT) sig = t.elastic_P[isab - 1] / Ein;
    else if (t.elastic_mode == SAB_ELASTIC_DISCRETE)
      sig = (1.0 - f) * t.elastic_P[isab - 1] + f * t.elastic_P[isab];
    double acc = 0.0;
    if (t.n_elastic_mu == 0) {
      const double mu = 1.0 - t.elastic_e_in[isab - 1] / Ein;  // coherent: one Bragg edge, :87
      acc = acc + pn_rt(l, mu);
    } else if (t.elastic_mode == SAB_ELASTIC_DISCRETE) {
      const int NMU = t.n_elastic_mu;
      const double wgt = 1.0 / (double)NMU;
      for (int imu = 0; imu < NMU; ++imu) {
        const double mu = (1.0 - f) * t.elastic_mu[(size_t)(isab - 1) * NMU + imu] +
                          f * t.elastic_mu[(size_t)isab * NMU + imu];
        acc = acc + wgt * pn_rt(l, mu);
      }
    }
    row[(size_t)(g - 1) * D.L + l] = sig * acc;
  }
}

// integrate_sab_inel_disc: thread per (E_in, order); groups are visited in the
// order the discrete outgoing energies fall into them, exactly as :216-242
__global__ void sab_inel_disc_kernel(SabDev D) {
  const long tot = (long)D.NE * D.L;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < tot; q += (long)gridDim.x * blockDim.x) {
    const int l = (int)(q % D.L), i = (int)(q / D.L);
    double* row = D.inel + (size_t)i * D.G * D.L;
    for (int g = 0; g < D.G; ++g) row[(size_t)g * D.L + l] = 0.0;
    const ndpp_sab_flat& t = D.t;
    const int NEo = t.n_inelastic_e_out, NMU = t.n_inelastic_mu, NEi = t.n_inelastic_e_in;
    const double Ein = D.ein[i];
    int isab;
    double f;
    if (Ein < t.inelastic_e_in[0]) { isab = 1; f = 0.0; }
    else if (Ein > t.threshold_inelastic) continue;
    else if (Ein == t.threshold_inelastic) { isab = NEi - 1; f = 1.0; }
    else {
      isab = bsearch1(t.inelastic_e_in, NEi, Ein);
      if (isab < 0) { if (l == 0) atomicOr(&D.status[i], NDPP_ST_RANGE); continue; }
      f = (Ein - t.inelastic_e_in[isab - 1]) / (t.inelastic_e_in[isab] - t.inelastic_e_in[isab - 1]);
    }
    const double sig = (1.0 - f) * t.inelastic_sigma[isab - 1] + f * t.inelastic_sigma[isab];
    for (int io = 0; io < NEo; ++io) {
      const double Eout = (1.0 - f) * t.inelastic_e_out[(size_t)(isab - 1) * NEo + io] +
                          f * t.inelastic_e_out[(size_t)isab * NEo + io];
      if (Eout < D.e_bins[0]) continue;
      else if (Eout >= D.e_bins[D.G]) continue;
      int g = bsearch1(D.e_bins, D.G + 1, Eout);
      if (g < 1) g = 1;  // NaN energy: stay inside the row
      double acc = row[(size_t)(g - 1) * D.L + l];
      const double w = D.wgt[io];
      for (int imu = 0; imu < NMU; ++imu) {
        const double mu = (1.0 - f) * t.inelastic_mu[((size_t)(isab - 1) * NEo + io) * NMU + imu] +
                          f * t.inelastic_mu[((size_t)isab * NEo + io) * NMU + imu];
        acc = acc + pn_rt(l, mu) * w;
      }
      row[(size_t)(g - 1) * D.L + l] = acc;
    }
    for (int g = 0; g < D.G; ++g) row[(size_t)g * D.L + l] = sig * row[(size_t)g * D.L + l];
  }
}

// integrate_sab_inel_cont, stage 1 (:292-378): thread per (table E_in, group, order)
__global__ void sab_cont_table_kernel(SabDev D) {
  const ndpp_sab_flat& t = D.t;
  const int NMU = t.n_inelastic_mu;
  const long tot = (long)t.n_inelastic_e_in * D.G * D.L;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < tot; q += (long)gridDim.x * blockDim.x) {
    const int l = (int)(q % D.L), g = (int)((q / D.L) % D.G), k = (int)(q / ((long)D.L * D.G));
    const int o = t.cont_ptr[k], NEout = t.cont_ptr[k + 1] - o;
    const double *Eo = t.cont_e_out + o, *pd0 = t.cont_pdf + o;
    const double* mu_arr = t.cont_mu + (size_t)o * NMU;
    auto pdf = [&](int j1) -> double {  // 1-based, :299-303
      return (j1 == NEout) ? 0.0 : pd0[j1 - 1] * (Eo[j1] - Eo[j1 - 1]);
    };
    const double eg = D.e_bins[g], eg1 = D.e_bins[g + 1];
    double acc = 0.0;
    int iE_lo = 1, iE_hi = 0;
    bool live = true;
    if (eg < Eo[0]) iE_lo = 1;
    else if (eg >= Eo[NEout - 1]) live = false;
    else {
      iE_lo = bsearch1(Eo, NEout, eg);
      if (iE_lo < 1) iE_lo = 1;
      const double f_lo = (eg - Eo[iE_lo - 1]) / (Eo[iE_lo] - Eo[iE_lo - 1]);
      const double mult = f_lo * pdf(iE_lo);
      for (int imu = 0; imu < NMU; ++imu) {
        const double mu = (1.0 - f_lo) * mu_arr[(size_t)(iE_lo - 1) * NMU + imu] +
                          f_lo * mu_arr[(size_t)iE_lo * NMU + imu];
        acc = acc + pn_rt(l, mu) * mult;
      }
      iE_lo = iE_lo + 1;
    }
    if (live) {
      if (eg1 < Eo[0]) live = false;
      else if (eg1 >= Eo[NEout - 1]) iE_hi = NEout - 1;
      else {
        iE_hi = bsearch1(Eo, NEout, eg1);
        if (iE_hi < 1) iE_hi = 1;
        const double f_hi = (eg1 - Eo[iE_hi - 1]) / (Eo[iE_hi] - Eo[iE_hi - 1]);
        const double mult = f_hi * pdf(iE_hi);
        for (int imu = 0; imu < NMU; ++imu) {
          const double mu = (1.0 - f_hi) * mu_arr[(size_t)(iE_hi - 1) * NMU + imu] +
                            f_hi * mu_arr[(size_t)iE_hi * NMU + imu];
          acc = acc + pn_rt(l, mu) * mult;
        }
        iE_hi = iE_hi - 1;
      }
    }
    if (live) {
      for (int iE = iE_lo; iE <= iE_hi; ++iE) {
        const double w = pdf(iE);
        for (int imu = 0; imu < NMU; ++imu)
          acc = acc + pn_rt(l, mu_arr[(size_t)(iE - 1) * NMU + imu]) * w;
      }
      acc = acc / (double)NMU;
    }
    D.distro[q] = live ? acc : 0.0;
  }
}

// stage 2 (:383-407): thread per output element
__global__ void sab_cont_interp_kernel(SabDev D) {
  const ndpp_sab_flat& t = D.t;
  const int GL = D.G * D.L;
  const long tot = (long)D.NE * GL;
  for (long q = blockIdx.x * (long)blockDim.x + threadIdx.x; q < tot; q += (long)gridDim.x * blockDim.x) {
    const int k = (int)(q % GL), i = (int)(q / GL);
    const double Ein = D.ein[i];
    double v = 0.0;
    if (Ein <= t.inelastic_e_in[0]) {
      v = D.distro[k] * t.inelastic_sigma[0];
    } else if (Ein >= t.threshold_inelastic) {
      v = 0.0;
    } else {
      const int isab = bsearch1(t.inelastic_e_in, t.n_inelastic_e_in, Ein);
      if (isab < 0) { if (k == 0) atomicOr(&D.status[i], NDPP_ST_RANGE); D.inel[q] = 0.0; continue; }
      const double f = (Ein - t.inelastic_e_in[isab - 1]) / (t.inelastic_e_in[isab] - t.inelastic_e_in[isab - 1]);
      const double sig = (1.0 - f) * t.inelastic_sigma[isab - 1] + f * t.inelastic_sigma[isab];
      v = ((1.0 - f) * D.distro[(size_t)(isab - 1) * GL + k] + f * D.distro[(size_t)isab * GL + k]) * sig;
    }
    D.inel[q] = v;
  }
}

// combine_sab_grid (:415-454): thread per E_in; the last point copies its
// neighbour, so thread NE-1 computes row NE-2's values again.
__global__ void sab_combine_kernel(SabDev D) {
  const int GL = D.G * D.L;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < D.NE; i += gridDim.x * blockDim.x) {
    const int src = (i == D.NE - 1 && D.NE >= 2) ? i - 1 : i;
    const double *e = D.el + (size_t)src * GL, *q = D.inel + (size_t)src * GL;
    double* m = D.mat + (size_t)i * GL;
    double s = 0.0, c = 0.0;  // sum(scatt_mat(1,:,iE)): flang's SUM is Kahan-compensated
    for (int g = 0; g < D.G; ++g) {
      const double y = (e[(size_t)g * D.L] + q[(size_t)g * D.L]) - c;
      const double tt = s + y;
      c = (tt - s) - y;
      s = tt;
    }
    if (s > 0.0) {
      s = 1.0 / s;
      for (int k = 0; k < GL; ++k) m[k] = (e[k] + q[k]) * s;
    } else {
      for (int k = 0; k < GL; ++k) m[k] = 0.0;
    }
  }
}

// apply_tol_scatt (scatt.F90:786-818): thread per incoming energy; both sum()s are
// Kahan-compensated like flang's SUM intrinsic.
__global__ void apply_tol_kernel(int L, int G, int n, double* data, double tol) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    double* d = data + (size_t)i * G * L;
    double s = 0.0, c = 0.0;
    for (int g = 0; g < G; ++g) { const double y = d[(size_t)g * L] - c, t = s + y; c = (t - s) - y; s = t; }
    const double orig = s;
    for (int g = 0; g < G; ++g)
      if ((d[(size_t)g * L] > 0.0) && (d[(size_t)g * L] < tol))
        for (int l = 0; l < L; ++l) d[(size_t)g * L + l] = 0.0;
    s = 0.0; c = 0.0;
    for (int g = 0; g < G; ++g) { const double y = d[(size_t)g * L] - c, t = s + y; c = (t - s) - y; s = t; }
    const double norm = (orig > 0.0) ? orig / s : 0.0;
    for (int k = 0; k < G * L; ++k) d[k] = d[k] * norm;
  }
}



#define SAB_TRY(expr)                                                             \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(NDPP_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

}  // namespace
}  // namespace ndpp

using namespace ndpp;

extern "C" int ndpp_sab_batch(const ndpp_params* p, const ndpp_sab_flat* t, int n_ein,
                              const double* ein, int G, const double* e_bins, double* el,
                              double* inel, double* scatt_mat) {
  if (!p || !t) return fail(NDPP_EINVAL, "NULL params/table");
  if (p->order < 1 || p->order > NDPP_MAX_ORDER)
    return fail(NDPP_EINVAL, "order=%d outside 1..%d", p->order, NDPP_MAX_ORDER);
  if (G < 1 || n_ein < 0) return fail(NDPP_EINVAL, "G=%d n_ein=%d", G, n_ein);
  if (n_ein == 0) return NDPP_OK;
  if (!ein || !e_bins || !scatt_mat) return fail(NDPP_EINVAL, "NULL array argument");
  const int NEi = t->n_inelastic_e_in, NEo = t->n_inelastic_e_out, NMU = t->n_inelastic_mu;
  const int mode = t->secondary_mode;
  if (NEi < 2 || NMU < 1) return fail(NDPP_EINVAL, "need >= 2 inelastic E_in and >= 1 cosine");
  if (mode != SAB_SECONDARY_EQUAL && mode != SAB_SECONDARY_SKEWED && mode != SAB_SECONDARY_CONT)
    return fail(NDPP_EINVAL, "secondary_mode=%d", mode);
  if (mode == SAB_SECONDARY_SKEWED && NEo <= 4)  // reference: fatal_error, sab.F90:183
    return fail(NDPP_EINVAL, "skewed weighting needs more than 4 outgoing energies");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");

  const int L = p->order;
  const size_t nout = (size_t)n_ein * G * L;
  // discrete-mode weights, sab.F90:167-186 (sum() as flang evaluates it: Kahan)
  std::vector<double> wgt(std::max(NEo, 1), 0.0);
  if (mode == SAB_SECONDARY_EQUAL) {
    for (int k = 0; k < NEo; ++k) wgt[k] = 1.0 / ((double)NEo * (double)NMU);
  } else if (mode == SAB_SECONDARY_SKEWED) {
    wgt[0] = 0.1; wgt[1] = 0.4;
    for (int k = 2; k < NEo - 2; ++k) wgt[k] = 1.0;
    wgt[NEo - 2] = 0.4; wgt[NEo - 1] = 0.1;
    double s = 0.0, c = 0.0;
    for (int k = 0; k < NEo; ++k) { const double y = wgt[k] - c, tt = s + y; c = (tt - s) - y; s = tt; }
    const double den = s * (double)NMU;
    for (int k = 0; k < NEo; ++k) wgt[k] = wgt[k] / den;
  }
  SabDev D;
  D.t = *t;
  D.NE = n_ein; D.G = G; D.L = L;
  DevBuf<double> d_ei, d_sig, d_eo, d_mu, d_ce, d_cp, d_cm, d_ee, d_eP, d_emu, d_ein, d_bins, d_w,
      d_distro, d_el, d_inel, d_mat;
  DevBuf<int> d_cptr, d_st;
  SAB_TRY(d_ei.upload(t->inelastic_e_in, NEi));
  SAB_TRY(d_sig.upload(t->inelastic_sigma, NEi));
  if (mode != SAB_SECONDARY_CONT) {
    SAB_TRY(d_eo.upload(t->inelastic_e_out, (size_t)NEi * NEo));
    SAB_TRY(d_mu.upload(t->inelastic_mu, (size_t)NEi * NEo * NMU));
  } else {
    if (!t->cont_ptr) return fail(NDPP_EINVAL, "continuous mode without cont_ptr");
    if (t->cont_ptr[0] != 0) return fail(NDPP_EINVAL, "cont_ptr[0] must be 0");
    const size_t tot = (size_t)t->cont_ptr[NEi];
    for (int k = 0; k < NEi; ++k)
      if (t->cont_ptr[k + 1] - t->cont_ptr[k] < 2)
        return fail(NDPP_EINVAL, "continuous row %d has < 2 outgoing energies", k);
    SAB_TRY(d_cptr.upload(t->cont_ptr, NEi + 1));
    SAB_TRY(d_ce.upload(t->cont_e_out, tot));
    SAB_TRY(d_cp.upload(t->cont_pdf, tot));
    SAB_TRY(d_cm.upload(t->cont_mu, tot * NMU));
  }
  if (t->threshold_elastic != 0.0) {
    if (t->n_elastic_e_in < 2) return fail(NDPP_EINVAL, "elastic data needs >= 2 E_in");
    SAB_TRY(d_ee.upload(t->elastic_e_in, t->n_elastic_e_in));
    SAB_TRY(d_eP.upload(t->elastic_P, t->n_elastic_e_in));
    if (t->n_elastic_mu > 0)
      SAB_TRY(d_emu.upload(t->elastic_mu, (size_t)t->n_elastic_e_in * t->n_elastic_mu));
  }
  SAB_TRY(d_ein.upload(ein, n_ein));
  SAB_TRY(d_bins.upload(e_bins, G + 1));
  SAB_TRY(d_w.upload(wgt.data(), wgt.size()));
  SAB_TRY(d_distro.alloc((size_t)NEi * G * L));
  SAB_TRY(d_el.alloc(nout));
  SAB_TRY(d_inel.alloc(nout));
  SAB_TRY(d_mat.alloc(nout));
  SAB_TRY(d_st.alloc(n_ein));
  SAB_TRY(hipMemset(d_st.p, 0, sizeof(int) * n_ein));
  D.t.inelastic_e_in = d_ei.p; D.t.inelastic_sigma = d_sig.p;
  D.t.inelastic_e_out = d_eo.p; D.t.inelastic_mu = d_mu.p;
  D.t.cont_ptr = d_cptr.p; D.t.cont_e_out = d_ce.p; D.t.cont_pdf = d_cp.p; D.t.cont_mu = d_cm.p;
  D.t.elastic_e_in = d_ee.p; D.t.elastic_P = d_eP.p; D.t.elastic_mu = d_emu.p;
  D.ein = d_ein.p; D.e_bins = d_bins.p; D.wgt = d_w.p; D.distro = d_distro.p;
  D.el = d_el.p; D.inel = d_inel.p; D.mat = d_mat.p; D.status = d_st.p;

  GpuSpan span(nullptr, kProfSab);
  hipLaunchKernelGGL(sab_el_kernel, dim3(nblk((long)n_ein * L, 128)), dim3(128), 0, 0, D);
  if (mode == SAB_SECONDARY_CONT) {
    hipLaunchKernelGGL(sab_cont_table_kernel, dim3(nblk((long)NEi * G * L, 64)), dim3(64), 0, 0, D);
    hipLaunchKernelGGL(sab_cont_interp_kernel, dim3(nblk((long)nout, 256)), dim3(256), 0, 0, D);
  } else {
    hipLaunchKernelGGL(sab_inel_disc_kernel, dim3(nblk((long)n_ein * L, 128)), dim3(128), 0, 0, D);
  }
  hipLaunchKernelGGL(sab_combine_kernel, dim3(nblk(n_ein, 128)), dim3(128), 0, 0, D);
  span.end();
  SAB_TRY(hipGetLastError());
  SAB_TRY(hipDeviceSynchronize());
  SAB_TRY(hipMemcpy(scatt_mat, d_mat.p, sizeof(double) * nout, hipMemcpyDeviceToHost));
  if (el) SAB_TRY(hipMemcpy(el, d_el.p, sizeof(double) * nout, hipMemcpyDeviceToHost));
  if (inel) SAB_TRY(hipMemcpy(inel, d_inel.p, sizeof(double) * nout, hipMemcpyDeviceToHost));
  return NDPP_OK;
}

extern "C" int ndpp_apply_tol_scatt(int L, int G, int n, double* data, double tol) {
  if (L < 1 || G < 1 || n < 0) return fail(NDPP_EINVAL, "L=%d G=%d n=%d", L, G, n);
  if (n == 0) return NDPP_OK;
  if (!data) return fail(NDPP_EINVAL, "NULL data");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  DevBuf<double> d;
  const size_t tot = (size_t)n * G * L;
  SAB_TRY(d.upload(data, tot));
  hipLaunchKernelGGL(apply_tol_kernel, dim3(nblk(n, 128)), dim3(128), 0, 0, L, G, n, d.p, tol);
  SAB_TRY(hipGetLastError());
  SAB_TRY(hipDeviceSynchronize());
  SAB_TRY(hipMemcpy(data, d.p, sizeof(double) * tot, hipMemcpyDeviceToHost));
  return NDPP_OK;
}
