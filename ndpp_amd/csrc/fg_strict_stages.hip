// fg_strict_stages.hip -- the stages of the free-gas pipeline in the reference's arithmetic
// (NDPP_FAST=0, no FMA contraction: the operation order of freegas.F90, the code of
// libndpp_hip_strict.so), callable from a library whose own arithmetic is the product's.
//
// Two uses (ndpp_hip.hip, run_batch_d):
//   * the prep stage of EVERY batch: find_FG_mu's mu limits come out of Brent iterations that
//     stop at a tolerance; computed here they carry the Fortran's bits (0.2 % of a pass);
//   * the whole pipeline for incoming energies with E_in < x A kT (x = 1e-3): on heavy
//     targets far below kT the kernel is ~1e9 with a kink at the alpha clamp, the inner
//     adaptive integration runs into its depth limit, and its unconverged remainder follows
//     the last bits of every K value -- only K values with the
//     Fortran's bits land within the reference's own 2e-11.
//
// Always compiled with -DNDPP_FAST=0 -ffp-contract=off (_build.py).  The batch arrives as
// bytes: FgBatch has the same layout in both arithmetic namespaces.  Joint jobs (both
// bracketing rows as one union tree) are walked here too: every row's kernel value is the
// reference expression, the row-independent sub-expressions are shared (ndpp_math.h fg_K_rows).
#include <cstring>

#include "../../include/ndpp_hip.h"
#include "fg_device.h"
#include "kernels.h"

#if NDPP_FAST
#error "fg_strict_stages.hip must be compiled with -DNDPP_FAST=0 -ffp-contract=off"
#endif

namespace ndpp {
namespace {

// the caller's batch as this translation unit's FgBatch, then the stage
template <class Launch>
int strict_stage(const void* batch, size_t batch_bytes, Launch launch) {
  FgBatch B;
  if (batch_bytes != sizeof(FgBatch)) return fail(NDPP_EDEVICE, "strict stage: batch layout mismatch");
  memcpy(&B, batch, sizeof B);
  launch(B);
  return NDPP_OK;
}

}  // namespace

int launch_fg_setup_strict(const void* batch, size_t batch_bytes, hipStream_t s) {
  return strict_stage(batch, batch_bytes, [&](const FgBatch& B) { launch_fg_setup(B, s); });
}
int launch_fg_prep_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s) {
  return strict_stage(batch, batch_bytes, [&](const FgBatch& B) { launch_fg_prep(B, level, s); });
}
int launch_fg_mu_strict(const void* batch, size_t batch_bytes, int level, int num_cu,
                        double* gstack, int* counter, hipStream_t s) {
  return strict_stage(batch, batch_bytes,
                      [&](const FgBatch& B) { launch_mu_any(B, level, num_cu, gstack, counter, s); });
}
int launch_fg_seg_zero_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s) {
  return strict_stage(batch, batch_bytes, [&](const FgBatch& B) { launch_fg_seg_zero(B, level, s); });
}
int launch_fg_combine_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s) {
  return strict_stage(batch, batch_bytes, [&](const FgBatch& B) { launch_fg_combine(B, level, s); });
}
int launch_fg_node_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s) {
  return strict_stage(batch, batch_bytes, [&](const FgBatch& B) { launch_fg_node(B, level, s); });
}
int launch_fg_reduce_strict(const void* batch, size_t batch_bytes, int level, hipStream_t s) {
  return strict_stage(batch, batch_bytes, [&](const FgBatch& B) { launch_fg_reduce(B, level, s); });
}
int launch_fg_assemble_strict(const void* batch, size_t batch_bytes, hipStream_t s) {
  return strict_stage(batch, batch_bytes, [&](const FgBatch& B) { launch_fg_assemble(B, s); });
}

}  // namespace ndpp
