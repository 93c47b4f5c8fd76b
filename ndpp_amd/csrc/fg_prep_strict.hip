// fg_prep_strict.hip -- the prep stage of the free-gas pipeline (find_FG_mu: the mu range in
// which the kernel exceeds its threshold, Brent iterations that stop at a tolerance; K at
// the ends and the middle of that range) in the reference's arithmetic, whatever the
// library's own: the mu limits of every inner integral then carry the Fortran's bits
// (median scale-relative difference to the reference on 3072 random cases 3.7e-14 -> 1.6e-14,
// DESIGN.md section 2).  The stage is 0.2 % of a pass.
//
// Always compiled with -DNDPP_FAST=0 -ffp-contract=off (_build.py).
#include <cstring>

#include "../../include/ndpp_hip.h"
#include "fg_device.h"
#include "kernels.h"

#if NDPP_FAST
#error "fg_prep_strict.hip must be compiled with -DNDPP_FAST=0 -ffp-contract=off"
#endif

namespace ndpp {

int launch_fg_prep_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s) {
  if (batch_bytes != sizeof(FgBatch)) return fail(NDPP_EDEVICE, "prep: batch layout mismatch");
  FgBatch B;                      // same layout in both arithmetic namespaces
  memcpy(&B, batch, sizeof B);
  hipLaunchKernelGGL(fg_prep_kernel, dim3(2048), dim3(256), 0, s, B, level);
  return NDPP_OK;
}

}  // namespace ndpp
