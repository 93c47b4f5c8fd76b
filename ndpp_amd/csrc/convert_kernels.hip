// convert_kernels.hip -- ACE -> tabular conversion of one reaction's angular /
// energy-angle distributions onto the uniform mu grid: ScattData%init's shape
// decisions (scattdata_header.F90:78-271) on the host and scatt_convert_distro
// (:325-382) = convert_file4 (:669-760) + convert_file6 (:769-950) on the GPU.
//
// The host walks the raw ACE block once and emits one small "column job" per
// output column (an M-vector f(mu) at one outgoing energy); the kernel gives
// every (column, mu point) its own thread.  The reference scans each table
// forward from the previous mu's match; since mu increases, that is the first
// match from the start of the table, which each thread finds on its own.
// Built with -DNDPP_FAST=0 -ffp-contract=off: isotropic / equiprobable /
// histogram / lin-lin columns are bit-identical to the Fortran; Kalbach-Mann and
// log-interpolated columns differ only through sinh/cosh/log/exp last bits.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "kernels.h"
#include "ndpp_math.h"

namespace ndpp {
namespace {

constexpr double kFpPrecision = 1e-14;  // constants.F90:22
constexpr int kNumEp = 32;              // constants.F90:116

enum ColKind : int { COL_ZERO = 0, COL_ISO = 1, COL_EQUI = 2, COL_TAB = 3, COL_KM = 4 };

struct ColJob {
  int kind;
  int src;        // 0: adist%data, 1: edist%data
  int lc;         // COL_EQUI: location; COL_TAB: 1-based index of the first abscissa
  int np;         // COL_TAB: number of points
  int interp;     // COL_TAB: 1..5
  int pad;
  double kmr, kma;  // COL_KM
};

// thread = (column, mu point); D(i) is the Fortran data(i).  Both data arrays are
// uploaded with one zero word in front and one behind, because the reference's
// index arithmetic can touch data(lc) / data(lc+34) of an equiprobable table.
__global__ void convert_kernel(int ncol, MuGrid grid, const ColJob* jobs, const double* adata,
                               const double* edata, double* f) {
  const int M = grid.M;
  const long tot = (long)ncol * M;
  for (long t = blockIdx.x * (long)blockDim.x + threadIdx.x; t < tot;
       t += (long)gridDim.x * blockDim.x) {
    const int col = (int)(t / M), imu = (int)(t - (long)col * M);
    const ColJob j = jobs[col];
    const double m = grid.at(imu);
    const double* data = j.src ? edata : adata;
#define D(i) data[(i)]
    double v = 0.0;
    if (j.kind == COL_ISO) {
      v = 0.5;
    } else if (j.kind == COL_KM) {  // :822-831
      const double KMconst = 0.5 * j.kma / sinh(j.kma);
      v = KMconst * (cosh(j.kma * m) + j.kmr * sinh(j.kma * m));
    } else if (j.kind == COL_EQUI) {  // :693-710
      for (int idata = j.lc + 1; idata <= j.lc + 1 + kNumEp; ++idata) {
        if (D(idata) >= m) {
          if (imu == 0) v = (1.0 / 32.0) / (D(idata + 1) - D(idata));
          else v = (1.0 / 32.0) / (D(idata) - D(idata - 1));
          break;
        }
      }
    } else if (j.kind == COL_TAB) {  // :711-751, :866-944
      const int NP = j.np, lc = j.lc;
      for (int idata = lc; idata <= lc + NP - 1; ++idata) {
        const double x = D(idata);
        if ((x - m) > kFpPrecision) {
          if (j.interp == 1) {
            v = D(idata - 1 + NP);
          } else if (j.interp == 2) {
            const double r = (m - D(idata - 1)) / (x - D(idata - 1));
            v = D(idata + NP - 1) + r * (D(idata + NP) - D(idata + NP - 1));
          } else if (j.interp == 3) {
            const double r = (log(m) - log(D(idata - 1))) / (log(x) - log(D(idata - 1)));
            v = D(idata + NP - 1) + r * (D(idata + NP) - D(idata - 1 + NP));
          } else if (j.interp == 4) {
            const double r = (m - D(idata - 1)) / (x - D(idata - 1));
            v = exp((1.0 - r) * log(D(idata + NP)) + r * log(D(idata + NP - 1)));  // (sic) :913
          } else {
            const double r = (log(m) - log(D(idata - 1))) / (log(x) - log(D(idata - 1)));
            v = exp((1.0 - r) * log(D(idata + NP)) + r * log(D(idata + NP - 1)));
          }
          break;
        } else if (fabs(x - m) <= kFpPrecision) {
          v = D(idata + NP);
          break;
        }
      }
    }
#undef D
    f[t] = v;
  }
}

// is_valid_scatter, scattdata_header.F90:1502-1515
bool valid_scatter(int MT) {
  if (MT == 2 || (MT >= 11 && MT <= 91))
    return MT != 18 && MT != 19 && MT != 20 && MT != 21 && MT != 38;
  return false;
}

// ScattData after init: which distributions it points at, its law and its grid
struct Shape {
  bool is_init = false;
  bool use_adist = false, use_edist = false, fabricated = false;
  int law = 0;
  int NE = 0;
  long total_np = 0;
  std::vector<int> np;       // per incoming energy
  std::vector<int> lc;       // edist: LDAT location of each incoming energy
  int e_off = 0;             // edist: 1-based index of E_in(1) minus 1
};

int nint_word(double x) { return (int)x; }  // Fortran int()

#define ED(i) r->edata[(i)-1]

// the reads of scatt_init (:225-244) with bounds checks instead of faults
int edist_shape(const ndpp_ace_reaction* r, Shape& s) {
  if (!r->edata || r->n_edata < 2) return fail(NDPP_EINVAL, "edist data missing");
  const int NR = nint_word(ED(1));
  if (NR < 0 || 2 + 2 * NR > r->n_edata) return fail(NDPP_EINVAL, "edist data: bad NR=%d", NR);
  const int NE = nint_word(ED(2 + 2 * NR));
  if (NE < 1 || 2 + 2 * NR + 2 * NE > r->n_edata)
    return fail(NDPP_EINVAL, "edist data: bad NE=%d", NE);
  s.NE = NE;
  s.e_off = 2 + 2 * NR;
  s.np.resize(NE);
  s.lc.resize(NE);
  s.total_np = 0;
  for (int i = 1; i <= NE; ++i) {
    const int lc = nint_word(ED(2 + 2 * NR + NE + i));
    if (lc < 0 || lc + 2 > r->n_edata)
      return fail(NDPP_EINVAL, "edist data: locator %d of incoming energy %d out of range", lc, i);
    const int NP = nint_word(ED(lc + 2));
    if (NP < 1) return fail(NDPP_EINVAL, "edist data: NP=%d at incoming energy %d", NP, i);
    s.lc[i - 1] = lc;
    s.np[i - 1] = NP;
    s.total_np += NP;
  }
  return NDPP_OK;
}

int make_shape(const ndpp_ace_reaction* r, Shape& s) {
  s = Shape();
  if (!valid_scatter(r->MT)) return NDPP_OK;  // :101
  const bool has_edist = r->law != 0;
  if (has_edist && r->law != 3 && r->law != 44 && r->law != 61 && r->law != 9 && r->law != 4)
    return NDPP_OK;  // :105-109
  s.is_init = true;
  if (r->has_angle_dist) {  // :135-149
    s.use_adist = true;
    s.use_edist = has_edist && r->law != 3;
    s.law = has_edist ? r->law : 0;
  } else if (has_edist) {   // :150-196
    if (r->law == 4 || r->law == 3 || r->law == 9) {
      s.use_edist = (r->law == 9 || r->law == 4);
      s.use_adist = true;
      s.fabricated = true;
    } else {
      s.use_edist = true;
    }
    s.law = r->law;
  } else {                  // :197-223
    s.use_adist = true;
    s.fabricated = true;
    s.law = 0;
  }
  if (s.use_adist && !s.fabricated) {
    if (r->n_adist < 1 || !r->adist_energy || !r->adist_type || !r->adist_location)
      return fail(NDPP_EINVAL, "has_angle_dist set but the angular distribution is missing");
  }
  if (s.use_adist && !s.use_edist) {  // :227-239
    s.NE = s.fabricated ? 2 : r->n_adist;
    s.np.assign(s.NE, 1);
    s.total_np = s.NE;
    return NDPP_OK;
  }
  return edist_shape(r, s);           // :240-254
}



#define CV_TRY(expr)                                                              \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(NDPP_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)

// a job for convert_file4(iE) on the reaction's (or the fabricated isotropic) adist
int adist_job(const ndpp_ace_reaction* r, const Shape& s, int iE, ColJob& j) {
  j = ColJob{COL_ZERO, 0, 0, 0, 0, 0, 0.0, 0.0};
  if (s.fabricated) {
    if (iE > 2) return fail(NDPP_EINVAL, "incoming energy %d beyond the isotropic 2-point adist", iE);
    j.kind = COL_ISO;
    return NDPP_OK;
  }
  if (iE < 1 || iE > r->n_adist)
    return fail(NDPP_EINVAL, "incoming energy %d beyond the %d angular tables", iE, r->n_adist);
  const int type = r->adist_type[iE - 1], lc = r->adist_location[iE - 1];
  if (type == 1) {
    j.kind = COL_ISO;
  } else if (type == 2) {
    if (lc < 0 || lc + kNumEp + 1 > r->n_adist_data)
      return fail(NDPP_EINVAL, "equiprobable table %d at %d outside adist data", iE, lc);
    j.kind = COL_EQUI;
    j.lc = lc;
  } else if (type == 3) {
    if (lc < 0 || lc + 2 > r->n_adist_data)
      return fail(NDPP_EINVAL, "tabular table %d at %d outside adist data", iE, lc);
    const int interp = nint_word(r->adist_data[lc]), NP = nint_word(r->adist_data[lc + 1]);
    if (interp == 1 || interp == 2) {   // other codes: zeros, as the reference leaves them
      if (NP < 1 || lc + 2 + 2 * NP > r->n_adist_data)
        return fail(NDPP_EINVAL, "tabular table %d: NP=%d outside adist data", iE, NP);
      j.kind = COL_TAB;
      j.lc = lc + 3;
      j.np = NP;
      j.interp = interp;
    }
  }  // unknown type: zeros (test_scattdata.F90:777-808)
  return NDPP_OK;
}


}  // namespace
}  // namespace ndpp

using namespace ndpp;

extern "C" int ndpp_scattdata_shape(const ndpp_ace_reaction* r, int* is_init, int* sd_law,
                                    int* NE, int* total_np) {
  if (!r || !is_init || !sd_law || !NE || !total_np) return fail(NDPP_EINVAL, "NULL argument");
  Shape s;
  const int rc = make_shape(r, s);
  if (rc) return rc;
  *is_init = s.is_init ? 1 : 0;
  *sd_law = s.law;
  *NE = s.NE;
  *total_np = (int)s.total_np;
  return NDPP_OK;
}

// f: host array the table is copied to, or null with keep: the device array stays and is handed over
static int convert_distro_impl(int mu_bins, const ndpp_ace_reaction* r, int G,
                               const double* e_bins, int NE, int total_np, double* e_grid,
                               int* row_ptr, double* eout, double* pdf, double* cdf, int* intt,
                               double* f, double** keep) {
  if (!r || !e_bins || !e_grid || !row_ptr || !eout || !pdf || !cdf || !intt || (!f && !keep))
    return fail(NDPP_EINVAL, "NULL argument");
  if (mu_bins < 2 || G < 1) return fail(NDPP_EINVAL, "mu_bins=%d G=%d", mu_bins, G);
  Shape s;
  int rc = make_shape(r, s);
  if (rc) return rc;
  if (!s.is_init) return fail(NDPP_EINVAL, "MT=%d law=%d is not a scattering ScattData", r->MT, r->law);
  if (NE != s.NE || total_np != (int)s.total_np)
    return fail(NDPP_EINVAL, "NE=%d total_np=%d but ndpp_scattdata_shape says %d, %ld", NE, total_np,
                s.NE, s.total_np);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");

  // ---- grid, outgoing-energy tables and one job per output column (host) ----
  std::vector<ColJob> jobs((size_t)total_np, ColJob{COL_ZERO, 0, 0, 0, 0, 0, 0.0, 0.0});
  row_ptr[0] = 0;
  for (int k = 0; k < NE; ++k) row_ptr[k + 1] = row_ptr[k] + s.np[k];
  std::fill(eout, eout + total_np, 0.0);
  std::fill(pdf, pdf + total_np, 0.0);
  std::fill(cdf, cdf + total_np, 0.0);
  if (!s.use_edist) {
    if (s.fabricated) {  // :160-170, :200-210
      e_grid[1] = e_bins[G];
      e_grid[0] = (r->threshold_energy > e_bins[0]) ? r->threshold_energy : e_bins[0];
    } else {
      for (int k = 0; k < NE; ++k) e_grid[k] = r->adist_energy[k];
    }
  } else {
    for (int k = 0; k < NE; ++k) e_grid[k] = ED(s.e_off + 1 + k);
  }
  for (int iE = 1; iE <= NE; ++iE) {
    const int o = row_ptr[iE - 1], NP = s.np[iE - 1];
    intt[iE - 1] = 1;  // HISTOGRAM placeholder of convert_file4 (:753-758)
    if (s.law == 0 || s.law == 3 || s.law == 9) {  // :342-348: column 1 only
      rc = adist_job(r, s, iE, jobs[o]);
      if (rc) return rc;
      continue;
    }
    // outgoing energies, pdf, cdf and INTT of convert_file6 (:799-820)
    const int lc = s.lc[iE - 1];
    if (lc + 2 + 3 * NP > r->n_edata)
      return fail(NDPP_EINVAL, "edist data: table of incoming energy %d truncated", iE);
    int it = nint_word(ED(lc + 1));
    if (it > 10) it = it % 10;
    intt[iE - 1] = it;
    for (int k = 1; k <= NP; ++k) {
      eout[o + k - 1] = ED(lc + 2 + k);
      pdf[o + k - 1] = ED(lc + 2 + NP + k);
      cdf[o + k - 1] = ED(lc + 2 + 2 * NP + k);
    }
    if (s.law == 4) {  // :350-370
      int iEa;
      const double E = e_grid[iE - 1];
      if (s.fabricated) {
        iEa = 1;
      } else if (E <= r->adist_energy[0]) {
        iEa = 1;
      } else if (E >= r->adist_energy[r->n_adist - 1]) {
        iEa = r->n_adist;
      } else {
        iEa = bsearch1_clamped(r->adist_energy, r->n_adist, E);
      }
      rc = adist_job(r, s, iEa, jobs[o]);
      if (rc) return rc;
      // (sic) the copy loop runs over the 2-element placeholder Eouts: only the
      // first two outgoing-energy columns receive the angular distribution
      if (NP >= 2) jobs[o + 1] = jobs[o];
    } else if (s.law == 44) {
      if (lc + 2 + 5 * NP > r->n_edata)
        return fail(NDPP_EINVAL, "edist data: Kalbach-Mann table %d truncated", iE);
      for (int k = 1; k <= NP; ++k) {
        ColJob& j = jobs[o + k - 1];
        j.kind = COL_KM;
        j.kmr = ED(lc + 2 + 3 * NP + k);
        j.kma = ED(lc + 2 + 4 * NP + k);
      }
    } else {  // law 61
      if (lc + 2 + 4 * NP > r->n_edata)
        return fail(NDPP_EINVAL, "edist data: law-61 table %d truncated", iE);
      for (int k = 1; k <= NP; ++k) {
        ColJob& j = jobs[o + k - 1];
        const int la = nint_word(ED(lc + 2 + 3 * NP + k));
        if (la == 0) { j.kind = COL_ISO; continue; }
        if (la < 0 || la + 2 > r->n_edata)
          return fail(NDPP_EINVAL, "edist data: angular locator %d out of range", la);
        const int interp = nint_word(ED(la + 1)), NPang = nint_word(ED(la + 2));
        if (interp < 1 || interp > 5)  // reference: fatal_error, :945
          return fail(NDPP_EINVAL, "Unknown interpolation type: %d", interp);
        if (NPang < 1 || la + 2 + 2 * NPang > r->n_edata)
          return fail(NDPP_EINVAL, "edist data: angular table at %d truncated", la);
        j.kind = COL_TAB;
        j.src = 1;
        j.lc = la + 3;
        j.np = NPang;
        j.interp = interp;
      }
    }
  }

  // ---- evaluate all columns on the device ----
  DevBuf<ColJob> d_jobs;
  DevBuf<double> d_a, d_e, d_f;
  CV_TRY(d_jobs.upload(jobs.data(), jobs.size()));
  {
    std::vector<double> pa((size_t)std::max(r->n_adist_data, 0) + 2, 0.0);
    std::vector<double> pe((size_t)std::max(r->n_edata, 0) + 2, 0.0);
    if (r->adist_data) std::copy(r->adist_data, r->adist_data + std::max(r->n_adist_data, 0), pa.begin() + 1);
    if (r->edata) std::copy(r->edata, r->edata + std::max(r->n_edata, 0), pe.begin() + 1);
    CV_TRY(d_a.upload(pa.data(), pa.size()));
    CV_TRY(d_e.upload(pe.data(), pe.size()));
  }
  const size_t nf = (size_t)total_np * mu_bins;
  CV_TRY(d_f.alloc(nf));
  GpuSpan span(nullptr, kProfConvert);
  hipLaunchKernelGGL(convert_kernel, dim3(nblk((long)nf, 256)), dim3(256), 0, 0, total_np,
                     make_mu_grid(mu_bins), d_jobs.p, d_a.p, d_e.p, d_f.p);
  span.end();
  CV_TRY(hipGetLastError());
  CV_TRY(hipDeviceSynchronize());
  if (f) CV_TRY(hipMemcpy(f, d_f.p, sizeof(double) * nf, hipMemcpyDeviceToHost));
  if (keep) { *keep = d_f.p; d_f.p = nullptr; }
  return NDPP_OK;
}

extern "C" int ndpp_convert_distro(int mu_bins, const ndpp_ace_reaction* r, int G,
                                   const double* e_bins, int NE, int total_np, double* e_grid,
                                   int* row_ptr, double* eout, double* pdf, double* cdf, int* intt,
                                   double* f) {
  if (!f) return fail(NDPP_EINVAL, "NULL argument");
  return convert_distro_impl(mu_bins, r, G, e_bins, NE, total_np, e_grid, row_ptr, eout, pdf, cdf, intt, f,
                             nullptr);
}

int ndpp::convert_distro_keep(int mu_bins, const ndpp_ace_reaction* r, int G, const double* e_bins,
                              int NE, int total_np, double* e_grid, int* row_ptr, double* eout,
                              double* pdf, double* cdf, int* intt, double** f_dev) {
  if (!f_dev) return fail(NDPP_EINVAL, "NULL argument");
  *f_dev = nullptr;
  return convert_distro_impl(mu_bins, r, G, e_bins, NE, total_np, e_grid, row_ptr, eout, pdf, cdf, intt,
                             nullptr, f_dev);
}

void ndpp::free_converted(double* f_dev) {
  if (f_dev) dev_free(f_dev);
}
