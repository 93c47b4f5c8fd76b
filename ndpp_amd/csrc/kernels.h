// kernels.h -- launchers shared between the translation units of libndpp_hip.so
#pragma once
#include <hip/hip_runtime.h>

namespace ndpp {

// records the message returned by ndpp_last_error() and returns `code`
int fail(int code, const char* fmt, ...);

// file4_kernels.hip (always built with the reference's IEEE operation order:
// -DNDPP_FAST=0 -ffp-contract=off, the kernel is bit-identical to the Fortran).
// Thread per (E_in of `list` (or all if null), group): integrate_file4_cm_leg for
// rows_per_ein bracketing rows + blend, written to out[i][g][0..L).
void launch_file4_any(int n, const int* list, int mu_bins, const double* ein,
                      const int* row_lo, const double* w_hi, const double* f_tab,
                      double awr, double Q, int G, int L, const double* e_bins,
                      int rows_per_ein, double* out, hipStream_t s);

}  // namespace ndpp
