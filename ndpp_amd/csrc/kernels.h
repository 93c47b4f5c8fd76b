// kernels.h -- launchers shared between the translation units of libndpp_hip.so
#pragma once
#include <hip/hip_runtime.h>

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
int launch_fg_mu_strict(const void* batch, size_t batch_bytes, int level, int mu_blocks,
                        double* gstack, double* gtot, hipStream_t s);
int launch_fg_combine_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_node_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_reduce_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s);
int launch_fg_assemble_strict(const void* batch, size_t batch_bytes, hipStream_t s);

}  // namespace ndpp
