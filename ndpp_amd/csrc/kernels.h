// kernels.h -- launchers shared between the translation units of libndpp_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/ndpp_hip.h"

namespace ndpp {

// records the message returned by ndpp_last_error() and returns `code`
int fail(int code, const char* fmt, ...);

// hipEvent bracket around the kernels of one batch call: the span between
// construction and end() is what ndpp_last_gpu_ms() reports (uploads, downloads
// and allocation are outside it).  The time is also added to the calling thread's
// per-family profile (ndpp_profile_get): which kernels a whole-nuclide call spent its
// device time in.
enum ProfileFamily { kProfFreegasMu = 0, kProfFreegasOther, kProfFile4, kProfFile6Cm, kProfFile6Lab,
                     kProfLaw9, kProfSab, kProfChi, kProfConvert, kNumProfileFamilies };
void profile_add(int family, double ms);
struct GpuSpan {
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipStream_t s;
  int family;
  explicit GpuSpan(hipStream_t stream = nullptr, int family = -1);
  void end();      // records the closing event; call before the final synchronise
  ~GpuSpan();      // after the device has synchronised: publishes the elapsed time
};

// file4_kernels.hip (always built with the reference's IEEE operation order:
// -DNDPP_FAST=0 -ffp-contract=off, the kernel is bit-identical to the Fortran).
// Thread per (E_in of `list` (or all if null), group): integrate_file4_cm_leg for
// rows_per_ein bracketing rows + blend, written to out[i][g][0..L).
// Mixed-nuclide batches pass nuc_of_ein / nuc_awr / nuc_Q (device arrays; awr and Q
// of incoming energy i are nuc_awr[nuc_of_ein[i]], nuc_Q[...]); otherwise null.
void launch_file4_any(int n, const int* list, int mu_bins, const double* ein,
                      const int* row_lo, const double* w_hi, const double* f_tab,
                      double awr, double Q, int G, int L, const double* e_bins,
                      int rows_per_ein, double* out, hipStream_t s,
                      const int* nuc_of_ein = nullptr, const double* nuc_awr = nullptr,
                      const double* nuc_Q = nullptr);

// fg_strict_stages.hip (always -DNDPP_FAST=0 -ffp-contract=off): the stages of the free-gas
// pipeline in the reference's arithmetic.  `batch` points to the caller's FgBatch (same
// layout in both arithmetic namespaces).
int launch_fg_setup_strict(const void* batch, size_t batch_bytes, hipStream_t s);
int launch_fg_prep_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_mu_strict(const void* batch, size_t batch_bytes, int level, int num_cu,
                        double* gstack, int* counter, hipStream_t s);
int launch_fg_seg_zero_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_combine_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_node_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_reduce_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_assemble_strict(const void* batch, size_t batch_bytes, hipStream_t s);

// Where a batch call leaves its moments when the caller goes on working on the device (the
// nuclide driver's reaction sum): consume() is handed the device array [n][G*L] instead of the
// call copying it to the host; it must only enqueue work on the null stream (the array is freed
// -- which waits for that work -- when the batch call returns).
struct DeviceSink {
  virtual ~DeviceSink() = default;
  virtual int consume(const double* out_d, int n, size_t GL) = 0;
};
// the batch entry points of the C ABI with a sink (null: host array `out`, as the ABI says)
int elastic_leg_batch_sink(const ndpp_params* p, double A, double kT, double freegas_cutoff, double Q,
                           int n_ein, const double* ein, const int* row_lo, const double* w_hi,
                           int n_rows, const double* f_tab, int G, const double* e_bins, double* out,
                           int* status, DeviceSink* sink);
// (f_dev non-null: the table f[sum NP][M] is already on the device -- convert_distro_keep -- and `f`
// is not read)
int file6_leg_batch_sink(const ndpp_params* p, double awr, int frame_cm, int n_ein, const double* ein,
                         const int* row_lo, int n_rows, const double* e_grid, const int* row_ptr,
                         const double* eout, const double* pdf, const int* intt, const double* f, int G,
                         const double* e_bins, double* out, int* status, DeviceSink* sink,
                         const double* f_dev = nullptr);
// ndpp_convert_distro that leaves the table where convert_kernel wrote it: *f_dev receives a
// device array [total_np][mu_bins] (dev_util.h's cached allocator; release with free_converted)
// for a consumer on the same device, and nothing of it crosses to the host
// (scattdata_header.F90:325-382 feeds integrate_distro: both ends are device kernels here).
int convert_distro_keep(int mu_bins, const ndpp_ace_reaction* r, int G, const double* e_bins, int NE,
                        int total_np, double* e_grid, int* row_ptr, double* eout, double* pdf,
                        double* cdf, int* intt, double** f_dev);
void free_converted(double* f_dev);
int law9_leg_batch_sink(const ndpp_params* p, int n_ein, const double* ein, const int* row_lo,
                        const double* w_hi, int n_rows, const double* f_tab, int n_edata,
                        const double* edata, int G, const double* e_bins, double* out, int* status,
                        DeviceSink* sink);
// dst[where[k]][j] += src[k][j] * scale[k] * pv[k]; nudst[where[k]][j] += yield[k] * that
// (scatt_interp_distro's scaling and calc_inelastic_grid's reaction sum, scattdata_header.F90:496,
// scatt.F90:753,:762, in their order of operations; all pointers device; null stream)
void launch_reaction_sum(int nb, size_t GL, const double* src, const int* where, const double* scale,
                         const double* pv, const double* yield, double* dst, double* nudst);

}  // namespace ndpp
