// ndpp_hip.hip -- gfx950 kernels + C ABI (include/ndpp_hip.h) of the NDPP
// scattering-moment hot path.  CDNA4 only: 64-lane wavefronts, FP64 VALU,
// LDS-resident per-lane stacks.  No MFMA: the path is adaptive quadrature
// (exp / sqrt / divide chains with data-dependent control flow), not a dense
// contraction.  See fg_pipeline.h for the algorithm and DESIGN.md for the
// roofline discussion.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/ndpp_hip.h"
#include "dev_util.h"
#include "fg_device.h"
#include "fg_pipeline.h"
#include "kernels.h"

using namespace ndpp;

// =============================================================================
// device side
// =============================================================================
namespace {

// Counting sort of a level's nodes by order mask (fg_task_decode): histogram, scan, scatter.
// Most nodes share a handful of masks, so the lanes of a wave first agree on one slot
// range per distinct mask (one LDS atomic per mask and wave) and a block touches the
// global histogram once per mask and tile.
constexpr int kSortMaxMasks = kSortClasses << 11;    // buckets of the level sort: weight classes x 2^L masks (L <= 11)

// returns this lane's slot within the block's tile for bucket m (m < 0: no item)
__device__ __forceinline__ int sort_block_slot(int m, int* lh) {
  int slot = 0;
  unsigned long long todo = __ballot(m >= 0);
  const int lane = threadIdx.x & (kWave - 1);
  while (todo) {
    const int lead = __ffsll((long long)todo) - 1;
    const int mv = __shfl(m, lead);
    const unsigned long long same = __ballot(m == mv);
    int first = 0;
    if (lane == lead) first = atomicAdd(&lh[mv], __popcll(same));
    first = __shfl(first, lead);
    if (m == mv) slot = first + __popcll(same & ((1ull << lane) - 1ull));
    todo &= ~same;
  }
  return slot;
}

__global__ __launch_bounds__(256) void fg_sort_count_kernel(FgBatch B, int level, int nb, int* hist) {
  __shared__ int lh[kSortMaxMasks];
  if (*B.overflow) return;
  const int base = B.lvl_off(level), nn = B.lvl_cnt[level];
  for (int b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
  __syncthreads();
  for (int i0 = blockIdx.x * blockDim.x; i0 < nn; i0 += gridDim.x * blockDim.x) {
    const int i = i0 + threadIdx.x;
    const int m = i < nn ? fg_node_bucket(B, level, base, base + i, nb / kSortClasses) : -1;
    sort_block_slot(m, lh);
  }
  __syncthreads();
  for (int b = threadIdx.x; b < nb; b += blockDim.x)
    if (lh[b]) atomicAdd(&hist[b], lh[b]);
}

// one block; exclusive scan in place.  The bucket of the empty key is the last one (fewest
// orders last): where it starts is the number of nodes the walk of this class has to visit.
__global__ void fg_sort_scan_kernel(int* hist, int nb, int* mu_nodes) {
  __shared__ int carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b0 = 0; b0 < nb; b0 += blockDim.x) {
    const int i = b0 + threadIdx.x;
    const int v = (i < nb) ? hist[i] : 0;
    // inclusive scan of the block's values by doubling (blockDim.x = 256)
    __shared__ int buf[256];
    buf[threadIdx.x] = v;
    __syncthreads();
    for (int o = 1; o < (int)blockDim.x; o <<= 1) {
      const int add = (threadIdx.x >= (unsigned)o) ? buf[threadIdx.x - o] : 0;
      __syncthreads();
      buf[threadIdx.x] += add;
      __syncthreads();
    }
    if (i < nb) hist[i] = carry + buf[threadIdx.x] - v;
    if (i == nb - 1) *mu_nodes = carry + buf[threadIdx.x] - v;
    __syncthreads();
    if (threadIdx.x == blockDim.x - 1) carry += buf[threadIdx.x];
    __syncthreads();
  }
}

__global__ __launch_bounds__(256) void fg_sort_scatter_kernel(FgBatch B, int level, int nb, int* cursor,
                                                               int* order) {
  __shared__ int lh[kSortMaxMasks];
  if (*B.overflow) return;
  const int base = B.lvl_off(level), nn = B.lvl_cnt[level];
  for (int i0 = blockIdx.x * blockDim.x; i0 < nn; i0 += gridDim.x * blockDim.x) {
    for (int b = threadIdx.x; b < nb; b += blockDim.x) lh[b] = 0;
    __syncthreads();
    const int i = i0 + threadIdx.x;
    const int m = i < nn ? fg_node_bucket(B, level, base, base + i, nb / kSortClasses) : -1;
    const int slot = sort_block_slot(m, lh);
    __syncthreads();
    for (int b = threadIdx.x; b < nb; b += blockDim.x)
      if (lh[b]) lh[b] = atomicAdd(&cursor[b], lh[b]);   // count -> first global slot of the tile
    __syncthreads();
    if (m >= 0) order[lh[m] + slot] = base + i;
    __syncthreads();
  }
}

// ---- batch plumbing ---------------------------------------------------------

// Is a tabulated row linear in mu on the uniform grid?  rough[row] = 1 unless the largest second
// difference |f[i+1] - 2 f[i] + f[i-1]| stays within rho x the row's largest |f| (rounding of a
// linear function: ~4e-16).  One block per row; max is exact in any order, so the host mirror
// (ndpp_freegas_rough_rows) gives the same flags.  See arithmetic_switch for what they decide.
__global__ __launch_bounds__(256) void fg_rough_kernel(int n_rows, int M, const double* __restrict__ f,
                                                        double rho, int* __restrict__ rough) {
  __shared__ double sd2[256], sfm[256];
  for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
    const double* p = f + (size_t)row * M;
    double d2 = 0.0, fm = 0.0;
    bool bad = false;
    for (int i = threadIdx.x; i < M; i += blockDim.x) {
      const double v = p[i];
      bad = bad || !(fabs(v) <= 1.7976931348623157e308);
      fm = fmax(fm, fabs(v));
      if (i > 0 && i < M - 1) d2 = fmax(d2, fabs((p[i + 1] - v) - (v - p[i - 1])));
    }
    sd2[threadIdx.x] = bad ? 1.7976931348623157e308 : d2;
    sfm[threadIdx.x] = fm;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
      if ((int)threadIdx.x < o) {
        sd2[threadIdx.x] = fmax(sd2[threadIdx.x], sd2[threadIdx.x + o]);
        sfm[threadIdx.x] = fmax(sfm[threadIdx.x], sfm[threadIdx.x + o]);
      }
      __syncthreads();
    }
    if (threadIdx.x == 0) rough[row] = (sd2[0] <= rho * sfm[0]) ? 0 : 1;
    __syncthreads();
  }
}

// E_in below the cutoff go to the free-gas pipeline, the rest to file4-CM
// (integrate_distro, scattdata_header.F90:548-564).  Free-gas energies with
// E_in < max(strict_x * A, strict_cold) * kT, and the ones whose bracketing rows are not linear in
// mu (rough != null), go to the list that is integrated in the reference's arithmetic
// (fg_strict_stages.hip).
__global__ void classify_kernel(int n_ein, const double* ein, double cutoff,
                                int* fg_list, int* n_fg, int* f4_list, int* n_f4,
                                const int* nuc_of_ein, const double* nuc_cutoff,
                                double* out, int GL, double strict_x, double strict_cold, double A, double kT,
                                const double* nuc_A, const double* nuc_kT, int* fgs_list,
                                int* n_fgs, const int* rough, const int* row_lo, int rows_per_ein,
                                int n_rows) {
  const int lane = threadIdx.x & (kWave - 1);
  // whole waves iterate together (one atomic per wave and list, lanes take consecutive slots)
  for (int i0 = (blockIdx.x * blockDim.x + threadIdx.x) - lane; i0 < n_ein;
       i0 += gridDim.x * blockDim.x) {
    const int i = i0 + lane;
    int cls = 0;                       // 0: nothing, 1: free gas, 2: file4, 3: free gas, strict
    if (i < n_ein) {
      if (nuc_of_ein) {
        cutoff = nuc_cutoff[nuc_of_ein[i]];
        A = nuc_A[nuc_of_ein[i]];
        kT = nuc_kT[nuc_of_ein[i]];
      }
      // an incoming energy that is not a positive finite number is not integrated (its
      // adaptive trees would never terminate): zero row, NDPP_ST_RANGE from status_kernel
      if (!(ein[i] > 0.0) || !(ein[i] <= 1.7976931348623157e308)) {
        for (int e = 0; e < GL; ++e) out[(size_t)i * GL + e] = 0.0;
      } else if (ein[i] < cutoff) {
        bool r = false;
        if (rough) {
          const int k = row_lo[i];
          if (k >= 0 && k + rows_per_ein <= n_rows)      // (a bad row fails the batch: check_rows_kernel)
            r = (rough[k] | rough[k + rows_per_ein - 1]) != 0;
        }
        cls = (r || ein[i] < fmax(strict_x * A, strict_cold) * kT) ? 3 : 1;
      } else {
        cls = 2;
      }
    }
    for (int c = 1; c <= 3; ++c) {
      const unsigned long long m = __ballot(cls == c);
      if (!m) continue;
      int* cnt = c == 1 ? n_fg : (c == 2 ? n_f4 : n_fgs);
      int* list = c == 1 ? fg_list : (c == 2 ? f4_list : fgs_list);
      const int lead = __ffsll((long long)m) - 1;
      int first = 0;
      if (lane == lead) first = atomicAdd(cnt, __popcll(m));
      first = __shfl(first, lead);
      if (cls == c) list[first + __popcll(m & ((1ull << lane) - 1ull))] = i;
    }
  }
}

// Jobs of one chunk.  rows_per_job = R: job j integrates rows row_lo..row_lo+R-1 of
// incoming energy list[j] jointly; with joint = 0 every (energy, row) pair is its
// own single-row job (calls keep the order energy-major, row-minor either way).
__global__ void fg_set_int_kernel(int* dst, int v) { *dst = v; }

// (the k-th incoming energy of the chunk is list[k * lstride]: a list dealt round-robin to two
// contexts is walked with stride 2)
__global__ void make_jobs_kernel(int n_jobs, int rows_per_ein, int joint, const int* list, int lstride,
                                 const double* ein, const int* row_lo, double* job_ein,
                                 int* job_row, const int* nuc_of_ein, const double* nuc_A,
                                 const double* nuc_kT, double* job_A, double* job_kT) {
  for (int j = blockIdx.x * blockDim.x + threadIdx.x; j < n_jobs; j += gridDim.x * blockDim.x) {
    int i;
    if (joint) {
      i = list[(size_t)j * lstride];
      job_ein[j] = ein[i];
      for (int r = 0; r < rows_per_ein; ++r) job_row[(size_t)j * rows_per_ein + r] = row_lo[i] + r;
    } else {
      i = list[(size_t)(j / rows_per_ein) * lstride];
      job_ein[j] = ein[i];
      job_row[j] = row_lo[i] + (j % rows_per_ein);
    }
    if (nuc_of_ein) {
      job_A[j] = nuc_A[nuc_of_ein[i]];
      job_kT[j] = nuc_kT[nuc_of_ein[i]];
    }
  }
}

__global__ void check_nuc_kernel(int n_ein, const int* nuc_of_ein, int n_nuc, int* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ein;
       i += gridDim.x * blockDim.x)
    if (nuc_of_ein[i] < 0 || nuc_of_ein[i] >= n_nuc) atomicOr(bad, 1);
}

// result = lo*(1-f) + hi*f, scattdata_header.F90:566,:589
__global__ void blend_kernel(int n, const int* list, int lstride, const double* raw,
                             const double* w_hi, int GL, double* out) {
  const long tot = (long)n * GL;
  for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < tot;
       k += (long)gridDim.x * blockDim.x) {
    const int j = (int)(k / GL), e = (int)(k % GL);
    const int i = list[(size_t)j * lstride];
    const double f = w_hi[i];
    const double lo = raw[((size_t)2 * j) * GL + e];
    const double hi = raw[((size_t)2 * j + 1) * GL + e];
    // three separately rounded operations, as the Fortran evaluates it: this file is built
    // with FMA contraction on, and a fused r + hi*f would put rows that the strict stages
    // reproduced bit for bit one rounding away from the reference
    const double r = __dmul_rn(lo, __dsub_rn(1.0, f));
    out[(size_t)i * GL + e] = __dadd_rn(r, __dmul_rn(hi, f));
  }
}

__global__ void copy_raw_kernel(int n, const int* list, int lstride, const double* raw, int GL,
                                double* out) {
  const long tot = (long)n * GL;
  for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < tot;
       k += (long)gridDim.x * blockDim.x) {
    const int j = (int)(k / GL), e = (int)(k % GL);
    out[(size_t)list[(size_t)j * lstride] * GL + e] = raw[(size_t)j * GL + e];
  }
}

__global__ void status_kernel(int n_ein, const double* ein, const double* out, int GL,
                              const int* row_lo, int n_rows, int rows_per_ein,
                              int* status) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ein;
       i += gridDim.x * blockDim.x) {
    int st = 0;
    for (int e = 0; e < GL; ++e) {
      const double v = out[(size_t)i * GL + e];
      if (!(fabs(v) <= 1.7976931348623157e308)) st |= NDPP_ST_NONFINITE;
    }
    if (!(ein[i] > 0.0) || !(ein[i] <= 1.7976931348623157e308)) st |= NDPP_ST_RANGE;
    status[i] = st;
  }
}

__global__ void check_rows_kernel(int n_ein, const int* row_lo, int n_rows,
                                  int rows_per_ein, int* bad) {
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_ein;
       i += gridDim.x * blockDim.x)
    if (row_lo[i] < 0 || row_lo[i] + rows_per_ein > n_rows) atomicOr(bad, 1);
}

}  // namespace

// =============================================================================
// host side
// =============================================================================
namespace {

thread_local char g_err[512] = "";
}  // namespace

namespace ndpp {
int fail(int code, const char* fmt, ...) {
  // Every failure path returns through here BEFORE its locals are destroyed -- and those locals
  // hand their device buffers back to a cache (dev_util.h DevCache) without waiting for the
  // device, on the promise that the caller has synchronised.  An early return has not: kernels
  // launched before the failure may still use the buffers.  So the failure itself waits.
  (void)hipDeviceSynchronize();
  (void)hipGetLastError();
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

static thread_local float g_last_gpu_ms = 0.f;
static thread_local double g_profile_ms[kNumProfileFamilies] = {0};
void profile_add(int family, double ms) {
  if (family >= 0 && family < kNumProfileFamilies) g_profile_ms[family] += ms;
}
GpuSpan::GpuSpan(hipStream_t stream, int fam) : s(stream), family(fam) {
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) { e0 = e1 = nullptr; return; }
  (void)hipEventRecord(e0, s);
}
void GpuSpan::end() {
  if (e1) (void)hipEventRecord(e1, s);
}
GpuSpan::~GpuSpan() {
  float ms = 0.f;
  if (e0 && e1 && hipEventSynchronize(e1) == hipSuccess && hipEventElapsedTime(&ms, e0, e1) == hipSuccess) {
    g_last_gpu_ms = ms;
    profile_add(family, ms);
  }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
}
}  // namespace ndpp

extern "C" void ndpp_profile_reset(void) {
  for (int k = 0; k < ndpp::kNumProfileFamilies; ++k) ndpp::g_profile_ms[k] = 0.0;
}
extern "C" int ndpp_profile_get(double* ms, int n) {
  for (int k = 0; k < n && k < ndpp::kNumProfileFamilies; ++k) ms[k] = ndpp::g_profile_ms[k];
  return ndpp::kNumProfileFamilies;
}
extern "C" float ndpp_last_gpu_ms(void) { return ndpp::g_last_gpu_ms; }

namespace {

#define HIP_TRY(expr)                                                         \
  do {                                                                        \
    hipError_t e_ = (expr);                                                   \
    if (e_ != hipSuccess)                                                     \
      return fail(NDPP_EDEVICE, "%s failed: %s (%s:%d)", #expr,               \
                  hipGetErrorString(e_), __FILE__, __LINE__);                 \
  } while (0)

// Cached workspace, one per device, each with its own lock: batch calls on one device are
// serialised (they share the arena), calls on different devices -- host threads that each
// selected their own GPU -- run concurrently and keep their arenas.
struct Workspace {
  std::mutex mu;
  char* base = nullptr;
  size_t bytes = 0;
  int num_cu = 0;      // multiProcessorCount, queried once (hipGetDeviceProperties costs ~1 ms,
                       // and a library-shaped run makes thousands of small batch calls)
  hipStream_t aux[3] = {nullptr, nullptr, nullptr};   // streams of the pipeline contexts beyond the first (run_batch_d)
};
constexpr int kMaxDevices = 64;
Workspace g_ws_of[kMaxDevices];

// the calling thread's current device and its workspace
int current_workspace(Workspace** ws) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return fail(NDPP_EDEVICE, "device ordinal %d outside 0..%d", dev, kMaxDevices - 1);
  *ws = &g_ws_of[dev];
  return NDPP_OK;
}

// (caller holds ws.mu)
int ensure_workspace(Workspace& ws, size_t bytes) {
  if (ws.base && ws.bytes < bytes) {
    hipFree(ws.base);
    ws.base = nullptr;
    ws.bytes = 0;
  }
  if (!ws.base) {
    hipError_t e = hipMalloc((void**)&ws.base, bytes);
    if (e != hipSuccess) {         // the staging-buffer cache may hold what is missing
      (void)hipGetLastError();
      dev_cache_trim();
      e = hipMalloc((void**)&ws.base, bytes);
    }
    if (e != hipSuccess)
      return fail(NDPP_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    ws.bytes = bytes;
  }
  return NDPP_OK;
}

struct Carver {
  char* p;
  char* end;
  template <class T>
  T* take(size_t n) {
    size_t b = (n * sizeof(T) + 255) & ~(size_t)255;
    T* r = reinterpret_cast<T*>(p);
    p += b;
    return r;
  }
};

int check_params(const ndpp_params* p, int G) {
  if (!p) return fail(NDPP_EINVAL, "params is NULL");
  if (p->order < 1 || p->order > NDPP_MAX_ORDER)
    return fail(NDPP_EINVAL, "order=%d outside 1..%d", p->order, NDPP_MAX_ORDER);
  if (p->mu_bins < 2) return fail(NDPP_EINVAL, "mu_bins=%d < 2", p->mu_bins);
  if (G < 1) return fail(NDPP_EINVAL, "need at least one group");
  if (p->adaptive_mu_its < 0 || p->adaptive_mu_its >= kMaxLevels ||
      p->adaptive_eout_its < 0 || p->adaptive_eout_its >= kMaxLevels)
    return fail(NDPP_EINVAL, "adaptive_*_its must be in 0..%d", kMaxLevels - 1);
  return NDPP_OK;
}

inline int gs_blocks(long n, int threads = 256) {
  return (int)std::max<long>(1, std::min<long>((n + threads - 1) / threads, 4096));
}

// Nodes of the outer trees one call is budgeted in the arena.  H-1 needs ~460
// at P5/G=2; heavier targets up to ~2x.  An overflow is detected on the device
// and the chunk is re-run with half as many calls.
constexpr int kNodesPerCallGuess = 1024;
// The free-gas pipeline of a batch runs as several contexts on their own streams (the product
// and the strict list, each dealt round-robin to more than one when it is long enough): the
// level tails and the small kernels between two levels of one context overlap with the inner
// walk of the others.
constexpr int kNumFgContexts = 4;
constexpr int kTwoContextsMinEin = 128;    // below that a list stays in one context (two pay from a few
                                           // hundred energies on: 512 -> +14 %, 2048 -> +6 %, 4096 -> +16 %)
// ... and so does a lone list of at least this many energies: a second context fills the lanes a
// level's tail leaves idle, but a wave that shares its SIMD runs at half speed, and so does the
// heaviest work item, which is what a level of the split walk waits for.  Once a level holds more
// work than its heaviest item lasts, one context with every level split is faster (H-1, P5, one /
// two contexts: 6 000 energies 336 / 240 ms, 12 500: 570 / 493, 16 667: 618 / 652, 25 000: 722 /
// 828, 50 000: 1156 / 1258, 100 000: 2141 / 2261; U-238-like P7 nuclide 2.64 / 2.79 s; 423-nuclide
// library 8.68 / 8.94 s).  Same bits either way.
constexpr int kTwoContextsMaxEin = 15000;
constexpr int kArenaSpareEin = 64;
size_t bytes_per_node(int nch) {
  return sizeof(double) * (2 + 6 * (size_t)nch) + 4 * sizeof(int)  // node arrays
         + 2 * 8 * sizeof(double)                                   // 2 tasks: limits + 3 K per row
         + 2                                                        // 2 tasks: Gauss-rule flag
         + sizeof(int);                                             // task order
}

// mixed-nuclide batch (ndpp_elastic_leg_multi): device arrays, per nuclide and per E_in
struct NucArrays {
  int n_nuc;
  const double *A, *kT, *cutoff, *Q;   // [n_nuc]
  const int* nuc_of_ein;                // [n_ein]
};

// Which incoming energies the product library integrates in the reference's arithmetic (the
// strict stages, fg_strict_stages.hip: every operation of freegas.F90 in its order, the
// reference's own exp) instead of its own.
//
// The product arithmetic perturbs every kernel value by a few 1e-16, and where the reference's
// inner adaptive integration works at its rounding noise, accept/refine decisions flip on last
// bits.  A flipped decision moves the result by the node's TRUE local error.  Where the integrand
// is smooth inside every node that is nothing (1e-19 of the integral); where the tabulated f(mu)
// has a kink inside the node -- the piecewise-linear interpolant of a curved or stepped table has
// one at every grid point -- the error estimate can vanish by cancellation while the true error
// does not, and the flip shows.  Measured on MI355X at production size against the all-strict walk
// (tools/parity_tail.py, profiles/r04/parity_tail_*; the all-strict walk is pinned to the C oracle
// on the worst 20 energies of every workload, <= 5e-16):
//   * rows LINEAR in mu (isotropic or P1-anisotropic: what a free-gas range sees in practice,
//     s-wave scattering; BASELINE configs 1, 2, 3 and 5 are of this kind): product arithmetic on
//     every energy, 1e-11 MeV ... 400 kT, A = 1 ... 250, G = 2 and 70, P3 ... P10, M = 513 and 2001,
//     627 000 energies in all: max 2.7e-14 at P5 / P7, 2.3e-13 on 70 groups, 8e-12 at P10.  (One
//     energy of one table, 1e-14 MeV = 4e-7 kT on A = 236 with 70 groups, showed 2.4e-8: hence the
//     guard below 1e-4 kT -- everything colder than the coldest energy these measurements cover --
//     which costs nothing: ACE grids start at 1e-11 MeV = 4e-4 kT at room temperature.)
//   * curved rows (f quadratic in mu sampled on the grid; round 3's sweeps): 7.5e-11 up to
//     E_in ~ 10 kT whatever the boundary below, 2.7e-9 below 1e-3 kT;
//   * stepped rows (32 equiprobable cosine bins, scattdata_header.F90:693-710): 5.6e-9 between
//     1e-3 and 0.1 kT -- the 1e-10 bar missed by a factor 56 with round 3's boundaries.
// Hence: the product arithmetic where BOTH bracketing rows are linear in mu to rounding
// (fg_rough_kernel: largest second difference <= 1e-12 x the row's largest value) and
// E_in >= 1e-4 kT; the reference arithmetic -- 6e-16 of the Fortran on 5 376 cases -- for every
// other table, at 1.45x ... 1.7x the cost.
//   NDPP_HIP_STRICT_ROUGH   the second-difference threshold (negative: tables are not looked at)
//   NDPP_HIP_STRICT_COLD    x in "E_in < x kT" (NDPP_HIP_STRICT_MANY: the same for more than two groups)
//   NDPP_HIP_STRICT_BELOW   x in "E_in < x A kT" (0 by default; 1e30 = the reference arithmetic
//                           everywhere; given as 0 it switches all three rules off: experiments)
// A library that is strict itself has nothing to switch.
constexpr double kStrictBelowDefault = 0.0;      // x A kT
constexpr double kStrictColdDefault = 1e-4;      // x kT
constexpr double kStrictRoughDefault = 1e-12;    // x the row's largest |f|
void arithmetic_switch(int G, double& strict_x, double& strict_cold, double& rough_rho) {
  strict_x = 0.0;
  strict_cold = 0.0;
  rough_rho = -1.0;
#if NDPP_FAST
  strict_x = kStrictBelowDefault;
  if (const char* sx = getenv("NDPP_HIP_STRICT_BELOW")) {
    strict_x = atof(sx);
    if (!(strict_x > 0.0)) { strict_x = 0.0; return; }      // "0": no switch at all
  }
  strict_cold = kStrictColdDefault;
  if (const char* sc = getenv("NDPP_HIP_STRICT_COLD")) strict_cold = atof(sc);
  if (G > 2)
    if (const char* sm = getenv("NDPP_HIP_STRICT_MANY")) strict_cold = atof(sm);
  if (!(strict_cold > 0.0)) strict_cold = 0.0;
  rough_rho = kStrictRoughDefault;
  if (const char* sr = getenv("NDPP_HIP_STRICT_ROUGH")) rough_rho = atof(sr);
#else
  (void)G;
#endif
}

// How one batch is laid out in the cached workspace and how many outer-tree nodes fit.
struct BatchPlan {
  int joint, nch;             // joint = 1: one job per E_in walks both rows as one union tree
  int mu_blocks, split_below;
  size_t mu_threads, seg_doubles, gstack_doubles, ctx_fixed, fixed, need;
  size_t nodes_per_ein;       // arena guess per incoming energy
  long ncap;                  // nodes in the arena
  long max_jobs;              // jobs (and calls) of the largest chunk
  long spare_ein;             // arena room beyond the guess, in incoming energies per context
  int contexts;               // pipeline contexts the batch may run side by side (workspace is carved for that many)
  long cap_ein;               // test hook: at most this many incoming energies per chunk (0 = no cap)
  double strict_x, strict_cold, rough_rho;
};

int plan_batch(const ndpp_params* p, int n_ein, int n_rows, int G, int rows_per_ein, Workspace& g_ws, BatchPlan& pl) {
  const int L = p->order, GL = G * L;
  size_t free_b = 0, total_b = 0;
  arithmetic_switch(G, pl.strict_x, pl.strict_cold, pl.rough_rho);
  const char* nj = getenv("NDPP_HIP_NO_JOINT");
  pl.joint = (rows_per_ein == 2 && L <= kJointMaxL && !(nj && nj[0] == '1')) ? 1 : 0;
  pl.nch = (pl.joint ? 2 : 1) * L;
  const size_t per_call_tree = (size_t)G * kSegPerGroup;
  // test hooks: NDPP_HIP_NODES_PER_CALL overrides the arena guess (a small value forces the
  // overflow -> halve-the-chunk path), NDPP_HIP_MAX_CHUNK_EIN caps the chunk (forces chunking)
  const char* e_nodes = getenv("NDPP_HIP_NODES_PER_CALL");
  const char* e_chunk = getenv("NDPP_HIP_MAX_CHUNK_EIN");
  const size_t guess = (e_nodes && atol(e_nodes) > 0) ? (size_t)atol(e_nodes) : (size_t)kNodesPerCallGuess;
  // per call at least 3 nodes per root: the task arrays hold 2 * ncap records and level 0
  // needs 5 per root.  The union tree of two similar rows is barely larger than either.
  pl.contexts = 2;
  pl.spare_ein = (e_nodes && atol(e_nodes) > 0) ? 0 : kArenaSpareEin;   // (the hook means the guess to bind)
  const size_t per_call = std::max<size_t>(guess, 3 * per_call_tree);
  pl.nodes_per_ein = pl.joint ? std::max<size_t>((guess * 5) / 4, 3 * per_call_tree) : per_call * rows_per_ein;
  if (g_ws.num_cu == 0) {
    hipDeviceProp_t prop;
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    HIP_TRY(hipGetDeviceProperties(&prop, dev));
    g_ws.num_cu = prop.multiProcessorCount;
  }
  pl.mu_blocks = g_ws.num_cu * kMuBlocksPerCU;      // the most any walk launches
  pl.mu_threads = (size_t)pl.mu_blocks * kWave;
  // (the global part of the sibling stacks is addressed with 24-bit multiplies: fg_device.h DevMuStack)
  if (pl.mu_threads * DevMuStack<2>::kRecBytes >= ((size_t)1 << 24))
    return fail(NDPP_EDEVICE, "%d compute units: the sibling-stack addressing assumes fewer than 4681 workgroups", g_ws.num_cu);
  // shallow stack levels that do not fit the LDS part: sized for whichever walk needs more
  pl.gstack_doubles = 0;
  for (int R = 1; R <= (pl.joint ? 2 : 1); ++R) {
    const size_t lv = (size_t)std::max(0, p->adaptive_mu_its - (R == 1 ? mu_lds_levels(1) : mu_lds_levels(2)));
    pl.gstack_doubles = std::max(pl.gstack_doubles, lv * ((R == 1 ? mu_stack_fields(1) : mu_stack_fields(2)) + 1) * pl.mu_threads);
  }
  // split mode (fg_pipeline.h kSplitLog2: 16 work items per inner integral) for levels with at most 32
  // inner integrals per lane, i.e. practically every level (items are handed out heaviest first; a
  // split level's lanes are busier -- 0.945 against 0.90 on the headline -- and its tail is a
  // sixteenth of an integral).  Threshold 6 / 16 / 32 per lane, one context: 50 000 energies 1241 /
  // 1160 / 1156 ms, 100 000: 2354 / 2345 / 2141.  The segment slots are sized for what the batch
  // can need (a small call does not reserve gigabytes): at most 128 inner integrals per incoming energy
  // and level are provided for; a level with more is walked unsplit, which gives the same bits.
  const char* ns = getenv("NDPP_HIP_NO_SPLIT");
  double split_x = 32.0;
  if (const char* e = getenv("NDPP_HIP_SPLIT_BELOW_X")) split_x = atof(e);   // measurement hook
  const size_t split_cap = std::min<size_t>((size_t)1 << 22, std::max<size_t>((size_t)n_ein * 128, (size_t)1 << 16));
  pl.split_below = (ns && ns[0] == '1') ? 0 : (int)std::min<size_t>((size_t)(split_x * pl.mu_threads), split_cap);
  pl.seg_doubles = (size_t)pl.split_below * kSplit * pl.nch;
  // per pipeline context (there are two, see run_batch_d): split-walk segments, global stack part,
  // sort histogram, level counters
  pl.ctx_fixed = (pl.seg_doubles + pl.gstack_doubles + 3) * sizeof(double) +
                 sizeof(int) * (((size_t)kSortClasses << L) + 4 * (kMaxLevels + 2) + 128) + 10 * 256;
  pl.fixed = (size_t)n_ein * 3 * sizeof(int) + (size_t)n_rows * sizeof(int) + (1u << 20) +
             sizeof(int) * ((size_t)1 << L) + pl.contexts * pl.ctx_fixed + 4096;
  const size_t node_bytes = bytes_per_node(pl.nch);
  // What the whole batch would take in one chunk.  If the cached workspace already holds that,
  // the free-memory query (~0.1 ms; thousands of small calls in a library-shaped run) is skipped.
  const size_t whole = pl.fixed + ((size_t)n_ein + pl.contexts * pl.spare_ein) * pl.nodes_per_ein * node_bytes +
                       (size_t)n_ein * rows_per_ein * (sizeof(double) * (GL + 3) + sizeof(int) * 2 + 16) + 4096;
  size_t budget;
  if (g_ws.base && whole <= g_ws.bytes) {
    budget = g_ws.bytes;
    free_b = g_ws.bytes;
  } else {
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (g_ws.base) free_b += g_ws.bytes;
    budget = std::min<size_t>((size_t)(free_b * 0.6), (size_t)128 << 30);
  }
  const size_t per_job_bytes = sizeof(double) * (GL + 3) + sizeof(int) * 2 + 16;  // job records + raw row
  // the arena holds ncap nodes; a chunk takes ncap / nodes_per_ein energies
  size_t ncap = (budget > pl.fixed ? budget - pl.fixed : 0) /
                (node_bytes + (per_job_bytes * rows_per_ein + pl.nodes_per_ein - 1) / pl.nodes_per_ein);
  // (nodes_per_ein is a guess: every context keeps room for kArenaSpareEin more energies, so
  // that a small batch of heavy trees is not held to it)
  ncap = std::min<size_t>(ncap, ((size_t)n_ein + pl.contexts * pl.spare_ein) * pl.nodes_per_ein);
  ncap = std::min<size_t>(ncap, (size_t)0x7fffffff / 5);
  pl.cap_ein = (e_chunk && atol(e_chunk) > 0) ? atol(e_chunk) : 0;
  if (pl.cap_ein) ncap = std::min<size_t>(ncap, (size_t)pl.cap_ein * pl.nodes_per_ein);
  if (ncap < pl.nodes_per_ein)
    return fail(NDPP_ENOMEM, "not enough device memory for one incoming energy (free %zu)", free_b);
  pl.ncap = (long)ncap;
  pl.max_jobs = (long)std::min<size_t>((size_t)n_ein, ncap / pl.nodes_per_ein) * rows_per_ein;
  pl.need = pl.fixed + (size_t)pl.ncap * node_bytes + (size_t)pl.max_jobs * per_job_bytes;
  return NDPP_OK;
}

// The device-resident batch (everything *_d).  rows_per_ein = 2 for the
// blended elastic batch, 1 for the single-row B-fine call.
int run_batch_d(const ndpp_params* p, double A, double kT, double cutoff, double Q,
                int n_ein, const double* ein_d, const int* row_lo_d,
                const double* w_hi_d, int n_rows, const double* f_tab_d, int G,
                const double* e_bins_d, double* out_d, int* status_d,
                int rows_per_ein, hipStream_t stream, ndpp_stats* stats,
                const NucArrays* na = nullptr) {
  int rc = check_params(p, G);
  if (rc) return rc;
  if (n_ein < 0 || n_rows < rows_per_ein)
    return fail(NDPP_EINVAL, "n_ein=%d n_rows=%d", n_ein, n_rows);
  if ((size_t)n_rows * (size_t)p->mu_bins * sizeof(double) >= ((size_t)1 << 32))
    return fail(NDPP_EINVAL, "f_tab of %zu bytes: one batch call addresses its table with 32-bit byte offsets "
                "(< 4 GiB); split the batch", (size_t)n_rows * (size_t)p->mu_bins * sizeof(double));
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_ein == 0) return NDPP_OK;
  if (!ein_d || !row_lo_d || !f_tab_d || !e_bins_d || !out_d ||
      (rows_per_ein == 2 && !w_hi_d))
    return fail(NDPP_EINVAL, "NULL array argument");

  Workspace* wsp = nullptr;
  rc = current_workspace(&wsp);
  if (rc) return rc;
  Workspace& g_ws = *wsp;
  std::lock_guard<std::mutex> lock(g_ws.mu);
  const int L = p->order, M = p->mu_bins;
  const int GL = G * L;

  BatchPlan pl;
  rc = plan_batch(p, n_ein, n_rows, G, rows_per_ein, g_ws, pl);
  if (rc) return rc;
  rc = ensure_workspace(g_ws, pl.need);
  if (rc) return rc;
  const int joint = pl.joint, split_below = pl.split_below;
  const int ncap = (int)pl.ncap;

  // (a batch without a free-gas range -- the level reactions' cutoff is 0 -- has nothing to switch)
  const bool look_at_tables = pl.rough_rho >= 0.0 && (na != nullptr || cutoff > 0.0);
  Carver cv{g_ws.base, g_ws.base + g_ws.bytes};
  int* fg_list = cv.take<int>(n_ein);
  int* f4_list = cv.take<int>(n_ein);
  int* fgs_list = cv.take<int>(n_ein);   // free gas, strict stages
  int* rough = cv.take<int>(n_rows);     // per table row: not linear in mu (fg_rough_kernel)
  int* counters = cv.take<int>(64);  // [0]=n_fg [1]=n_f4 [3]=badrow [4]=badnuc [5]=n_fgs
  unsigned long long* dstats = cv.take<unsigned long long>(kNumStats);
  const int nb_masks = 1 << L;                         // sort keys: the orders active in any row ...
  const int nb_sort = kSortClasses * nb_masks;         // ... per weight class (fg_node_bucket)
  int* mask_rank = cv.take<int>(nb_masks);
  // what every pipeline context owns besides its share of the node arena
  struct Slot {
    int *lvl_cnt, *next_task, *overflow, *mask_hist, *mu_nodes;
    double *seg, *gstack;
    hipStream_t s;
  } slot[kNumFgContexts];
  for (int k = 0; k < pl.contexts; ++k) {
    slot[k].lvl_cnt = cv.take<int>(kMaxLevels + 2);
    slot[k].next_task = cv.take<int>(2 * (kMaxLevels + 2));     // one row of counters per order class
    slot[k].overflow = cv.take<int>(64);
    slot[k].mu_nodes = cv.take<int>(64);
    slot[k].mask_hist = cv.take<int>(nb_sort);
    slot[k].seg = cv.take<double>(pl.seg_doubles + 1);
    slot[k].gstack = cv.take<double>(pl.gstack_doubles + 1);
    slot[k].s = stream;
  }
  char* const arena = cv.p;

  const char* ng = getenv("NDPP_HIP_GAUSS");        // 0: every inner integral by the adaptive walk
  const bool gauss_on = NDPP_FAST && look_at_tables && !(ng && ng[0] == '0');
  const char* nsort = getenv("NDPP_HIP_NO_SORT");   // test hook: walk tasks in creation order
  const bool do_sort = !(nsort && nsort[0] == '1');

  hipEvent_t ev0, ev1;
  HIP_TRY(hipEventCreate(&ev0));
  HIP_TRY(hipEventCreate(&ev1));
  struct EvGuard {
    hipEvent_t a, b;
    ~EvGuard() { (void)hipEventDestroy(a); (void)hipEventDestroy(b); }
  } guard{ev0, ev1};

  {
    // bucket of a mask: more active orders first (the longer integrals start first), then by value
    auto upload_rank = [&](int* dst, int nb) -> hipError_t {
      std::vector<int> idx(nb), rank(nb);
      for (int m = 0; m < nb; ++m) idx[m] = m;
      std::stable_sort(idx.begin(), idx.end(), [](int x, int y) {
        return __builtin_popcount((unsigned)x) > __builtin_popcount((unsigned)y);
      });
      for (int k = 0; k < nb; ++k) rank[idx[k]] = k;
      hipError_t e = hipMemcpyAsync(dst, rank.data(), sizeof(int) * nb, hipMemcpyHostToDevice, stream);
      if (e != hipSuccess) return e;
      return hipStreamSynchronize(stream);   // rank[] is a local
    };
    HIP_TRY(upload_rank(mask_rank, nb_masks));
  }
  HIP_TRY(hipEventRecord(ev0, stream));
  HIP_TRY(hipMemsetAsync(counters, 0, 64 * sizeof(int), stream));
  HIP_TRY(hipMemsetAsync(dstats, 0, kNumStats * sizeof(unsigned long long), stream));
  hipLaunchKernelGGL(check_rows_kernel, dim3(gs_blocks(n_ein)), dim3(256), 0, stream,
                     n_ein, row_lo_d, n_rows, rows_per_ein, counters + 3);
  if (na)
    hipLaunchKernelGGL(check_nuc_kernel, dim3(gs_blocks(n_ein)), dim3(256), 0, stream, n_ein,
                       na->nuc_of_ein, na->n_nuc, counters + 4);
  const bool look = look_at_tables;
  if (look)
    hipLaunchKernelGGL(fg_rough_kernel, dim3(std::min(n_rows, 4096)), dim3(256), 0, stream, n_rows, M, f_tab_d,
                       pl.rough_rho, rough);
  hipLaunchKernelGGL(classify_kernel, dim3(gs_blocks(n_ein)), dim3(256), 0, stream,
                     n_ein, ein_d, cutoff, fg_list, counters + 0, f4_list, counters + 1,
                     na ? na->nuc_of_ein : nullptr, na ? na->cutoff : nullptr, out_d, GL,
                     pl.strict_x, pl.strict_cold, A, kT, na ? na->A : nullptr, na ? na->kT : nullptr, fgs_list,
                     counters + 5, look ? rough : nullptr, row_lo_d, rows_per_ein, n_rows);
  int hc[6];
  HIP_TRY(hipMemcpyAsync(hc, counters, sizeof(hc), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  if (hc[3]) return fail(NDPP_EINVAL, "row_lo outside [0, n_rows-%d]", rows_per_ein);
  if (hc[4]) return fail(NDPP_EINVAL, "nuc_of_ein outside [0, n_nuc)");
  const int n_fg_fast = hc[0], n_f4 = hc[1], n_fg_strict = hc[5];

  // ---- file4-CM part ------------------------------------------------------
  hipEvent_t ev_f4;
  HIP_TRY(hipEventCreate(&ev_f4));
  struct Ev1 { hipEvent_t e; ~Ev1() { (void)hipEventDestroy(e); } } guard_f4{ev_f4};
  launch_file4_any(n_f4, f4_list, M, ein_d, row_lo_d, w_hi_d, f_tab_d, A, Q, G,
                   L, e_bins_d, rows_per_ein, out_d, stream, na ? na->nuc_of_ein : nullptr,
                   na ? na->A : nullptr, na ? na->Q : nullptr);
  HIP_TRY(hipEventRecord(ev_f4, stream));

  // ---- free-gas part: pipeline contexts -------------------------------------
  // One context = one list of incoming energies in one arithmetic, walked chunk by chunk through
  // its own share of the node arena on its own stream.  The product and the strict list are two
  // contexts; a lone list of at least kTwoContextsMinEin energies is dealt round-robin to two
  // (bit-identical results: an incoming energy's moments do not depend on what shares its
  // chunk).  The inner walk of one context then fills what the other leaves idle -- the tail of
  // a level, during which a few lanes finish their integrals, and the sort / limits / node
  // kernels between two levels.
  struct FgCtx {
    FgBatch B;
    Slot sl;
    bool strict;
    const int* list;            // this context's k-th energy is list[k * lstride]
    int lstride;
    long n, done, chunk_ein, this_ein;
    int ncap;
    long max_jobs;
    double *job_ein, *job_A, *job_kT;
    int *job_row, *order;
    bool inflight;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev;   // (before, after) each fg_mu_kernel
    std::vector<std::pair<hipEvent_t, hipEvent_t>> gev;  // ... each fg_gauss_kernel
  };
  std::vector<FgCtx> ctx;
  long two_min = kTwoContextsMinEin, two_max = kTwoContextsMaxEin;
  if (const char* e = getenv("NDPP_HIP_TWO_CONTEXTS_MIN")) two_min = atol(e);   // test hook; 0 = one at a time
  if (const char* e = getenv("NDPP_HIP_TWO_CONTEXTS_MAX")) two_max = atol(e);   // test hook
  {
    auto add = [&](const int* list, int parts, long n, bool strict) {
      for (int j = 0; j < parts; ++j) {       // part j: list[j], list[j + parts], ...
        FgCtx c{};
        c.list = list + j; c.lstride = parts; c.n = (n - j + parts - 1) / parts; c.strict = strict;
        if (c.n > 0) ctx.push_back(c);
      }
    };
    // Two contexts in all: the product and the strict list, or a lone list of two_min ... two_max - 1
    // incoming energies dealt to two (kTwoContextsMaxEin).  (Measured with 3 and 4:
    // 12 500 / 25 000 / 100 000 H-1 energies run at 53.3 / 60.4 / 68.7 k E_in*orders/s with two,
    // 52.5 / 57.4 / 67.4 with three, 53.7 / 57.0 / 67.1 with four.)
    const int total = pl.contexts;
    auto parts_of = [&](long n, int room) {
      if (n <= 0) return 0;
      const int k = (two_min > 0 && n >= two_min && n < two_max && n >= room) ? room : 1;
      return std::max(1, k);
    };
    const int k_strict = parts_of(n_fg_strict, n_fg_fast > 0 ? std::max(1, total / 2) : total);
    const int k_fast = parts_of(n_fg_fast, std::max(1, total - k_strict));
    add(fg_list, k_fast, n_fg_fast, false);
    add(fgs_list, k_strict, n_fg_strict, true);
  }
  const int nctx = (int)ctx.size();
  // the contexts run side by side when every one of them gets room for at least one incoming
  // energy, else one after the other through the whole arena
  long n_all = 0;
  for (auto& c : ctx) n_all += c.n;
  const bool side_by_side = nctx > 1 && nctx <= pl.contexts && two_min > 0 &&
                            (size_t)ncap >= (size_t)nctx * pl.nodes_per_ein;
  for (int k = 1; side_by_side && k < nctx; ++k)
    if (!g_ws.aux[k - 1]) HIP_TRY(hipStreamCreateWithFlags(&g_ws.aux[k - 1], hipStreamNonBlocking));
  {
    FgBatch T;
    T.G = G; T.L = L; T.M = M; T.A = A; T.kT = kT;
    T.f_tab = f_tab_d; T.e_bins = e_bins_d;
    T.sab_threshold = p->sab_threshold; T.brent_thresh = p->brent_mu_thresh;
    T.mu_tol = p->adaptive_mu_tol; T.eout_tol = p->adaptive_eout_tol;
    T.mu_its = p->adaptive_mu_its; T.eout_its = p->adaptive_eout_its;
    T.grid = make_mu_grid(M);
    T.R = joint ? rows_per_ein : 1;
    T.mask_rank = mask_rank;
    T.split_below = split_below;
    T.stats = dstats;
    long cap_left = ncap;
    for (int k = 0; k < nctx; ++k) {
      FgCtx& c = ctx[k];
      if (!side_by_side) cv.p = arena;             // every context in turn takes the whole arena
      c.sl = slot[side_by_side ? k : 0];
      if (side_by_side && k > 0) c.sl.s = g_ws.aux[k - 1];
      long share = side_by_side ? std::max<long>((long)pl.nodes_per_ein,
                                                 (long)((double)ncap * (c.n + pl.spare_ein) /
                                                        (n_all + (long)nctx * pl.spare_ein))) : ncap;
      share = std::min<long>(share, (long)((size_t)(c.n + pl.spare_ein) * pl.nodes_per_ein));
      if (side_by_side) {
        share = std::min(share, cap_left - (long)(nctx - 1 - k) * (long)pl.nodes_per_ein);
        cap_left -= share;
      }
      c.ncap = (int)share;
      c.max_jobs = std::min<long>(c.n, share / (long)pl.nodes_per_ein) * rows_per_ein;
      FgBatch& B = c.B;
      B = T;
      B.ncap = c.ncap;
      B.node_a = cv.take<double>(c.ncap);
      B.node_b = cv.take<double>(c.ncap);
      B.node_F = cv.take<double>((size_t)5 * pl.nch * c.ncap);
      B.node_S = cv.take<double>((size_t)pl.nch * c.ncap);
      B.node_info = cv.take<int>((size_t)4 * c.ncap);
      B.tcap = 2 * c.ncap;
      B.t_mulo = cv.take<double>(B.tcap);
      B.t_muhi = cv.take<double>(B.tcap);
      B.t_X = cv.take<double>((size_t)3 * (joint ? 2 : 1) * B.tcap);
      // the Gauss rule for the inner integrals the reference has converged: product-arithmetic
      // contexts only, and only when the batch's tables were looked at (a context of the product
      // arithmetic then holds energies of rows linear in mu only)
      B.t_gl = (gauss_on && !c.strict) ? cv.take<unsigned char>(B.tcap) : nullptr;
      // (experiment knobs; the defaults are what profiles/r04/parity_tail_*.log were measured with)
      if (const char* e = getenv("NDPP_HIP_GAUSS_RATIO")) B.gl_ratio = atof(e);
      if (const char* e = getenv("NDPP_HIP_GAUSS_NEAR")) B.gl_near = atoi(e) != 0;
      if (const char* e = getenv("NDPP_HIP_GAUSS_AMIN")) B.gl_amin = atof(e);
      if (const char* e = getenv("NDPP_HIP_GAUSS_DEPTH")) B.gl_cert_depth = std::min(std::max(atoi(e), 0), 10);
      if (const char* e = getenv("NDPP_HIP_GAUSS_DEPTH_NEAR")) B.gl_cert_depth_near = std::min(std::max(atoi(e), 0), 10);
      if (const char* e = getenv("NDPP_HIP_GAUSS_GRADED")) B.gl_graded = std::min(std::max(atoi(e), 0), 20);
      if (const char* e = getenv("NDPP_HIP_GAUSS_PANELS")) B.gl_panels = std::min(std::max(atoi(e), 8), 1024);
      c.job_ein = cv.take<double>(c.max_jobs);
      c.job_row = cv.take<int>(c.max_jobs);
      B.job_ein = c.job_ein;
      B.job_row = c.job_row;
      c.job_A = cv.take<double>(c.max_jobs);
      c.job_kT = cv.take<double>(c.max_jobs);
      if (na) { B.job_A = c.job_A; B.job_kT = c.job_kT; }
      B.raw = cv.take<double>((size_t)c.max_jobs * GL);
      c.order = cv.take<int>(c.ncap);
      B.order = do_sort ? c.order : nullptr;
      B.seg = split_below ? c.sl.seg : nullptr;
      B.lvl_cnt = c.sl.lvl_cnt;
      B.next_task = c.sl.next_task;
      B.overflow = c.sl.overflow;
      if (cv.p > cv.end)
        return fail(NDPP_ENOMEM, "workspace carve overran (%zu > %zu)",
                    (size_t)(cv.p - g_ws.base), g_ws.bytes);
      c.chunk_ein = std::max<long>(1, c.max_jobs / rows_per_ein);
      if (pl.cap_ein) c.chunk_ein = std::min<long>(c.chunk_ein, pl.cap_ein);
    }
  }
  // (events of a chunk that an error path leaves behind)
  struct EventSweep {
    std::vector<FgCtx>& v;
    ~EventSweep() {
      for (auto& c : v) {
        for (auto& e : c.ev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
        for (auto& e : c.gev) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
      }
    }
  } sweep{ctx};
  // whatever path leaves this function, nothing may still be running in the arena
  struct Drain {
    hipStream_t first;
    hipStream_t* more;
    int n_more;
    ~Drain() {
      (void)hipStreamSynchronize(first);
      for (int k = 0; k < n_more; ++k) (void)hipStreamSynchronize(more[k]);
    }
  } drain{stream, g_ws.aux, side_by_side ? nctx - 1 : 0};

  double mu_sum_ms = 0.0, gauss_sum_ms = 0.0;
  double level_ms[32] = {0};
  int mu_launches = 0;
  std::vector<std::pair<float, float>> mu_spans;   // (start, end) of each fg_mu_kernel, ms after ev0

  // queue one chunk of a context on its stream (nothing here waits for the device)
  auto enqueue = [&](FgCtx& c) -> int {
    FgBatch& B = c.B;
    hipStream_t s = c.sl.s;
    const bool sp = c.strict;
    for (;;) {
      c.this_ein = std::min<long>(c.chunk_ein, c.n - c.done);
      B.n_jobs = joint ? (int)c.this_ein : (int)(c.this_ein * rows_per_ein);
      const int ntrees = B.n_trees();
      if ((long)ntrees * kSegPerGroup <= (long)B.tcap && ntrees <= c.ncap) break;
      // more roots than the arena can even start with: take fewer energies
      if (c.chunk_ein <= 1) return fail(NDPP_EOVERFLOW, "arena of %d nodes is too small for one E_in", c.ncap);
      c.chunk_ein = std::max<long>(1, c.chunk_ein / 2);
    }
    HIP_TRY(hipMemsetAsync(c.sl.lvl_cnt, 0, (kMaxLevels + 2) * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(c.sl.next_task, 0, 2 * (kMaxLevels + 2) * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(c.sl.overflow, 0, sizeof(int), s));
    hipLaunchKernelGGL(fg_set_int_kernel, dim3(1), dim3(1), 0, s, c.sl.lvl_cnt, B.n_trees());
    hipLaunchKernelGGL(make_jobs_kernel, dim3(gs_blocks(B.n_jobs)), dim3(256), 0, s,
                       B.n_jobs, rows_per_ein, joint, c.list + (size_t)c.done * c.lstride, c.lstride,
                       ein_d, row_lo_d, c.job_ein, c.job_row, na ? na->nuc_of_ein : nullptr,
                       na ? na->A : nullptr, na ? na->kT : nullptr, c.job_A, c.job_kT);
    int rc = NDPP_OK;
    if (sp) { rc = launch_fg_setup_strict(&B, sizeof B, s); if (rc) return rc; }
    else launch_fg_setup(B, s);
    const int nlev = B.eout_its + 1;
    for (int level = 0; level < nlev; ++level) {
      // the mu limits come out of Brent iterations that stop at a tolerance: in the product
      // arithmetic they would end ~1e-7 away from the reference's, and every inner integral
      // with them.  So the prep stage of every batch runs in the reference arithmetic
      // (fg_strict_stages.hip; 0.2 % of a pass).
      B.mu_nodes = nullptr;
      rc = launch_fg_prep_strict(&B, sizeof B, level, s);
      if (rc) return rc;
      if (B.t_gl) {
        hipEvent_t ga, gb;
        HIP_TRY(hipEventCreate(&ga));
        HIP_TRY(hipEventCreate(&gb));
        c.gev.emplace_back(ga, gb);
        HIP_TRY(hipEventRecord(ga, s));
        launch_gauss_any(B, level, s);
        HIP_TRY(hipEventRecord(gb, s));
      }
      if (do_sort) {
        // nodes sorted by weight class and the orders still active in any row; nodes with nothing
        // left for the walk last (fg_node_bucket)
        B.mu_nodes = c.sl.mu_nodes;
        HIP_TRY(hipMemsetAsync(c.sl.mask_hist, 0, sizeof(int) * nb_sort, s));
        hipLaunchKernelGGL(fg_sort_count_kernel, dim3(1024), dim3(256), 0, s, B, level, nb_sort, c.sl.mask_hist);
        hipLaunchKernelGGL(fg_sort_scan_kernel, dim3(1), dim3(256), 0, s, c.sl.mask_hist, nb_sort, c.sl.mu_nodes);
        hipLaunchKernelGGL(fg_sort_scatter_kernel, dim3(1024), dim3(256), 0, s, B, level, nb_sort,
                           c.sl.mask_hist, c.order);
      }
      int* counter = c.sl.next_task + level;
      if (B.seg) {
        if (sp) { rc = launch_fg_seg_zero_strict(&B, sizeof B, level, s); if (rc) return rc; }
        else launch_fg_seg_zero(B, level, s);
      }
      hipEvent_t a, b;
      HIP_TRY(hipEventCreate(&a));
      HIP_TRY(hipEventCreate(&b));
      c.ev.emplace_back(a, b);
      HIP_TRY(hipEventRecord(a, s));
      if (sp) {
        rc = launch_fg_mu_strict(&B, sizeof B, level, g_ws.num_cu, c.sl.gstack, counter, s);
        if (rc) return rc;
      } else {
        launch_mu_any(B, level, g_ws.num_cu, c.sl.gstack, counter, s);
      }
      HIP_TRY(hipEventRecord(b, s));
      if (sp) { rc = launch_fg_combine_strict(&B, sizeof B, level, s); if (rc) return rc; }
      else launch_fg_combine(B, level, s);
      B.mu_nodes = nullptr;
      if (sp) { rc = launch_fg_node_strict(&B, sizeof B, level, s); if (rc) return rc; }
      else launch_fg_node(B, level, s);
    }
    for (int level = nlev - 1; level >= 0; --level) {
      if (sp) { rc = launch_fg_reduce_strict(&B, sizeof B, level, s); if (rc) return rc; }
      else launch_fg_reduce(B, level, s);
    }
    if (sp) { rc = launch_fg_assemble_strict(&B, sizeof B, s); if (rc) return rc; }
    else launch_fg_assemble(B, s);
    const int* lst = c.list + (size_t)c.done * c.lstride;
    if (rows_per_ein == 2)
      hipLaunchKernelGGL(blend_kernel, dim3(gs_blocks(c.this_ein * GL)), dim3(256), 0, s,
                         (int)c.this_ein, lst, c.lstride, B.raw, w_hi_d, GL, out_d);
    else
      hipLaunchKernelGGL(copy_raw_kernel, dim3(gs_blocks(c.this_ein * GL)), dim3(256), 0, s,
                         (int)c.this_ein, lst, c.lstride, B.raw, GL, out_d);
    c.inflight = true;
    return NDPP_OK;
  };

  // the chunk of a context has left the device: its timings, and whether it has to be redone
  auto retire = [&](FgCtx& c) -> int {
    c.inflight = false;
    int ovf = 0;
    HIP_TRY(hipMemcpyAsync(&ovf, c.sl.overflow, sizeof(int), hipMemcpyDeviceToHost, c.sl.s));
    HIP_TRY(hipStreamSynchronize(c.sl.s));
    HIP_TRY(hipGetLastError());
    int lvl_i = 0;
    const int ev_per_level = 1;
    for (auto& e : c.ev) {
      float t0 = 0.f, t1 = 0.f;
      if (hipEventElapsedTime(&t0, ev0, e.first) == hipSuccess &&
          hipEventElapsedTime(&t1, ev0, e.second) == hipSuccess) {
        mu_sum_ms += t1 - t0;
        if (lvl_i / ev_per_level < 32) level_ms[lvl_i / ev_per_level] += t1 - t0;
        mu_spans.emplace_back(t0, t1);
      }
      lvl_i++;
      mu_launches++;
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    c.ev.clear();
    for (auto& e : c.gev) {
      // (the Gauss stage of a level is part of its inner integration: counted in mu_busy_ms)
      float t0 = 0.f, t1 = 0.f;
      if (hipEventElapsedTime(&t0, ev0, e.first) == hipSuccess &&
          hipEventElapsedTime(&t1, ev0, e.second) == hipSuccess) {
        gauss_sum_ms += t1 - t0;
        mu_spans.emplace_back(t0, t1);
      }
      (void)hipEventDestroy(e.first);
      (void)hipEventDestroy(e.second);
    }
    c.gev.clear();
    if (ovf) {
      // the adaptive trees outgrew the arena: redo this chunk with half the energies
      if (c.chunk_ein <= 1)
        return fail(NDPP_EOVERFLOW, "outer tree of one E_in exceeds %d nodes", c.ncap);
      c.chunk_ein = std::max<long>(1, c.chunk_ein / 2);
    } else {
      c.done += c.this_ein;
    }
    return NDPP_OK;
  };

  if (side_by_side) {
    for (auto& c : ctx) { rc = enqueue(c); if (rc) return rc; }
    for (;;) {
      int live = 0;
      bool moved = false;
      for (auto& c : ctx) {
        if (!c.inflight) continue;
        ++live;
        const hipError_t q = hipStreamQuery(c.sl.s);
        if (q == hipErrorNotReady) continue;
        if (q != hipSuccess) return fail(NDPP_EDEVICE, "free-gas pipeline failed: %s", hipGetErrorString(q));
        rc = retire(c);
        if (rc) return rc;
        if (c.done < c.n) { rc = enqueue(c); if (rc) return rc; }
        moved = true;
      }
      if (!live) break;
      if (!moved) std::this_thread::sleep_for(std::chrono::microseconds(50));
    }
  } else {
    for (auto& c : ctx)
      while (c.done < c.n) {
        rc = enqueue(c);
        if (!rc) rc = retire(c);
        if (rc) return rc;
      }
  }
  // time during which at least one fg_mu_kernel was in flight
  double mu_ms = 0.0;
  {
    std::sort(mu_spans.begin(), mu_spans.end());
    float end = -1.f;
    for (auto& sp : mu_spans) {
      if (sp.second <= end) continue;
      mu_ms += sp.second - std::max(sp.first, end);
      end = sp.second;
    }
  }

  if (status_d)
    hipLaunchKernelGGL(status_kernel, dim3(gs_blocks(n_ein)), dim3(256), 0, stream, n_ein, ein_d,
                       out_d, GL, row_lo_d, n_rows, rows_per_ein, status_d);
  HIP_TRY(hipEventRecord(ev1, stream));
  unsigned long long hs[kNumStats];
  HIP_TRY(hipMemcpyAsync(hs, dstats, sizeof(hs), hipMemcpyDeviceToHost, stream));
  HIP_TRY(hipStreamSynchronize(stream));
  HIP_TRY(hipGetLastError());
  {
    float ms = 0.f, ms4 = 0.f;
    if (hipEventElapsedTime(&ms, ev0, ev1) == hipSuccess) ndpp::g_last_gpu_ms = ms;
    // the span up to ev_f4 is classification + the file4 kernel; the rest is the free-gas pipeline
    if (hipEventElapsedTime(&ms4, ev0, ev_f4) == hipSuccess) {
      profile_add(kProfFile4, ms4);
      profile_add(kProfFreegasMu, mu_ms);
      profile_add(kProfFreegasOther, std::max(0.0, (double)ms - ms4 - mu_ms));
    }
  }
  if (stats) {
    stats->k_evals = hs[kStatKEvals];
    stats->mu_visits = hs[kStatMuVisits];
    stats->mu_integrals = hs[kStatMuIntegrals];
    stats->eout_nodes = hs[kStatEoutNodes];
    stats->wave_iters = hs[kStatWaveIters];
    stats->lane_iters = hs[kStatLaneIters];
    stats->order_visits = hs[kStatOrderVisits];
    stats->gauss_integrals = hs[kStatGaussIntegrals];
    stats->gauss_ms = gauss_sum_ms;
    for (int k = 0; k < 32; ++k) stats->mu_level_ms[k] = level_ms[k];
    stats->mu_kernel_ms = mu_sum_ms;
    stats->mu_busy_ms = mu_ms;
    stats->mu_kernel_launches = mu_launches;
    stats->contexts = side_by_side ? nctx : 1;
    float ms = 0.f;
    hipEventElapsedTime(&ms, ev0, ev1);
    stats->total_ms = ms;
  }
  return NDPP_OK;
}

// host-pointer front end: stage to device, run, copy back
int run_batch_h(const ndpp_params* p, double A, double kT, double cutoff, double Q,
                int n_ein, const double* ein, const int* row_lo, const double* w_hi,
                int n_rows, const double* f_tab, int G, const double* e_bins,
                double* out, int* status, int rows_per_ein, ndpp_stats* stats,
                ndpp::DeviceSink* sink = nullptr) {
  int rc = check_params(p, G);
  if (rc) return rc;
  if (n_ein < 0) return fail(NDPP_EINVAL, "n_ein=%d", n_ein);
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_ein == 0) return NDPP_OK;
  if (!ein || !row_lo || !f_tab || !e_bins || (!out && !sink) || (rows_per_ein == 2 && !w_hi))
    return fail(NDPP_EINVAL, "NULL array argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  const int L = p->order, M = p->mu_bins;
  const size_t GL = (size_t)G * L;
  double *ein_d = nullptr, *w_d = nullptr, *f_d = nullptr, *eb_d = nullptr, *out_d = nullptr;
  int *row_d = nullptr, *st_d = nullptr;
  auto cleanup = [&]() {
    dev_free(ein_d); dev_free(w_d); dev_free(f_d); dev_free(eb_d); dev_free(out_d);
    dev_free(row_d); dev_free(st_d);
  };
#define TRY_OR_CLEAN(expr)                                                     \
  do {                                                                         \
    hipError_t e_ = (expr);                                                    \
    if (e_ != hipSuccess) {                                                    \
      cleanup();                                                               \
      return fail(NDPP_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    }                                                                          \
  } while (0)
  TRY_OR_CLEAN(dev_alloc((void**)&ein_d, sizeof(double) * n_ein));
  TRY_OR_CLEAN(dev_alloc((void**)&w_d, sizeof(double) * n_ein));
  TRY_OR_CLEAN(dev_alloc((void**)&row_d, sizeof(int) * n_ein));
  TRY_OR_CLEAN(dev_alloc((void**)&st_d, sizeof(int) * n_ein));
  TRY_OR_CLEAN(dev_alloc((void**)&f_d, sizeof(double) * (size_t)n_rows * M));
  TRY_OR_CLEAN(dev_alloc((void**)&eb_d, sizeof(double) * (G + 1)));
  TRY_OR_CLEAN(dev_alloc((void**)&out_d, sizeof(double) * n_ein * GL));
  TRY_OR_CLEAN(hipMemcpy(ein_d, ein, sizeof(double) * n_ein, hipMemcpyHostToDevice));
  if (w_hi) TRY_OR_CLEAN(hipMemcpy(w_d, w_hi, sizeof(double) * n_ein, hipMemcpyHostToDevice));
  TRY_OR_CLEAN(hipMemcpy(row_d, row_lo, sizeof(int) * n_ein, hipMemcpyHostToDevice));
  TRY_OR_CLEAN(hipMemcpy(f_d, f_tab, sizeof(double) * (size_t)n_rows * M, hipMemcpyHostToDevice));
  TRY_OR_CLEAN(hipMemcpy(eb_d, e_bins, sizeof(double) * (G + 1), hipMemcpyHostToDevice));
  rc = run_batch_d(p, A, kT, cutoff, Q, n_ein, ein_d, row_d, w_d, n_rows, f_d, G, eb_d,
                   out_d, st_d, rows_per_ein, nullptr, stats);
  if (rc == NDPP_OK && sink) {
    rc = sink->consume(out_d, n_ein, GL);
    if (rc == NDPP_OK) TRY_OR_CLEAN(hipDeviceSynchronize());
  }
  if (rc == NDPP_OK) {
    if (!sink) TRY_OR_CLEAN(hipMemcpy(out, out_d, sizeof(double) * n_ein * GL, hipMemcpyDeviceToHost));
    if (status)
      TRY_OR_CLEAN(hipMemcpy(status, st_d, sizeof(int) * n_ein, hipMemcpyDeviceToHost));
  }
  cleanup();
  return rc;
#undef TRY_OR_CLEAN
}

int check_mu_grid(const ndpp_params* p, const double* mu) {
  if (!mu) return NDPP_OK;
  MuGrid g = make_mu_grid(p->mu_bins);
  for (int i = 0; i < p->mu_bins; ++i)
    if (mu[i] != g.at(i))
      return fail(NDPP_EINVAL, "mu[%d] is not the uniform grid of scatt_init", i);
  return NDPP_OK;
}

}  // namespace

int ndpp::elastic_leg_batch_sink(const ndpp_params* p, double A, double kT, double freegas_cutoff,
                                 double Q, int n_ein, const double* ein, const int* row_lo,
                                 const double* w_hi, int n_rows, const double* f_tab, int G,
                                 const double* e_bins, double* out, int* status, DeviceSink* sink) {
  return run_batch_h(p, A, kT, freegas_cutoff, Q, n_ein, ein, row_lo, w_hi, n_rows, f_tab, G, e_bins, out,
                     status, 2, nullptr, sink);
}

extern "C" {

void ndpp_default_params(ndpp_params* p) {
  // constants.F90:70-100
  p->order = 6;
  p->mu_bins = 2001;
  p->sab_threshold = 1.0E-6;
  p->brent_mu_thresh = 1.0E-6;
  p->adaptive_mu_tol = 1.0E-7;
  p->adaptive_eout_tol = 1.0E-8;
  p->adaptive_mu_its = 15;
  p->adaptive_eout_its = 15;
  p->ne_per_grp = 20;
  p->sab_epts_per_bin = 10;
  p->extend_pts = 50;
  p->inel_extend_pts = 30;
}

const char* ndpp_version(void) {
#if NDPP_FAST
  // the boundary in force (NDPP_HIP_STRICT_BELOW moves it), not the compiled-in default
  static thread_local char buf[260];
  double x, cold, rho, xm, many, rhom;
  arithmetic_switch(2, x, cold, rho);
  arithmetic_switch(70, xm, many, rhom);
  snprintf(buf, sizeof buf, "ndpp-hip 0.4 (gfx950; free gas: product arithmetic on table rows linear in mu "
           "(second differences <= %g), reference arithmetic on all others and below max(%g A, %g) kT "
           "(more than two groups: max(%g A, %g) kT))", rho, x, cold, xm, many);
  return buf;
#else
  return "ndpp-hip 0.4 (gfx950; free gas: reference arithmetic)";
#endif
}
const char* ndpp_last_error(void) { return g_err; }

int ndpp_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ---- device memory for hosts that use the *_d entry points without linking HIP ----
void* ndpp_dev_alloc(size_t bytes) {
  void* p = nullptr;
  hipError_t e = hipMalloc(&p, bytes ? bytes : 8);
  if (e != hipSuccess) {
    fail(NDPP_ENOMEM, "hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
    return nullptr;
  }
  return p;
}

int ndpp_dev_free(void* p) {
  if (p && hipFree(p) != hipSuccess) return fail(NDPP_EDEVICE, "hipFree failed");
  return NDPP_OK;
}

int ndpp_dev_upload(void* dst_d, const void* src, size_t bytes) {
  if (!bytes) return NDPP_OK;
  if (!dst_d || !src) return fail(NDPP_EINVAL, "NULL pointer");
  HIP_TRY(hipMemcpy(dst_d, src, bytes, hipMemcpyHostToDevice));
  return NDPP_OK;
}

int ndpp_dev_download(void* dst, const void* src_d, size_t bytes) {
  if (!bytes) return NDPP_OK;
  if (!dst || !src_d) return fail(NDPP_EINVAL, "NULL pointer");
  HIP_TRY(hipMemcpy(dst, src_d, bytes, hipMemcpyDeviceToHost));
  return NDPP_OK;
}

int ndpp_dev_synchronize(void) {
  HIP_TRY(hipDeviceSynchronize());
  return NDPP_OK;
}

int ndpp_reserve_workspace(size_t bytes) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  Workspace* ws = nullptr;
  int rc = current_workspace(&ws);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(ws->mu);
  if (bytes == 0) {   // what the largest batch may take: min(60 % of free HBM, 128 GB)
    size_t free_b = 0, total_b = 0;
    HIP_TRY(hipMemGetInfo(&free_b, &total_b));
    if (ws->base) free_b += ws->bytes;
    bytes = std::min<size_t>((size_t)(free_b * 0.6), (size_t)128 << 30);
  }
  if (ws->base && ws->bytes >= bytes) return NDPP_OK;
  return ensure_workspace(*ws, bytes);
}

int ndpp_release_workspace(void) {
  Workspace* ws = nullptr;
  int rc = current_workspace(&ws);
  if (rc) return rc;
  std::lock_guard<std::mutex> lock(ws->mu);
  if (ws->base) hipFree(ws->base);
  ws->base = nullptr;
  ws->bytes = 0;
  dev_cache_trim();          // the cached staging buffers of the batch calls (dev_util.h)
  for (auto& a : ws->aux) {
    if (a) (void)hipStreamDestroy(a);
    a = nullptr;
  }
  return NDPP_OK;
}

double ndpp_freegas_strict_below(int groups, double A, double kT) {
  double sx = 0.0, sc = 0.0, rho = 0.0;
  arithmetic_switch(groups, sx, sc, rho);
  return std::fmax(sx * A, sc) * kT;
}

// host mirror of fg_rough_kernel (cost models, tests): the same expression, and max is exact
int ndpp_freegas_rough_rows(int mu_bins, int n_rows, const double* f_tab, int* rough) {
  if (mu_bins < 2 || n_rows < 0 || (n_rows > 0 && (!f_tab || !rough))) return fail(NDPP_EINVAL, "ndpp_freegas_rough_rows: bad argument");
  double sx = 0.0, sc = 0.0, rho = 0.0;
  arithmetic_switch(2, sx, sc, rho);
  for (int row = 0; row < n_rows; ++row) {
    const double* p = f_tab + (size_t)row * mu_bins;
    double d2 = 0.0, fm = 0.0;
    bool bad = false;
    for (int i = 0; i < mu_bins; ++i) {
      const double v = p[i];
      bad = bad || !(std::fabs(v) <= 1.7976931348623157e308);
      fm = std::fmax(fm, std::fabs(v));
      if (i > 0 && i < mu_bins - 1) d2 = std::fmax(d2, std::fabs((p[i + 1] - v) - (v - p[i - 1])));
    }
    if (bad) d2 = 1.7976931348623157e308;
    rough[row] = (rho >= 0.0 && !(d2 <= rho * fm)) ? 1 : 0;
  }
  return NDPP_OK;
}

int ndpp_set_device(int device) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  if (device < 0 || device >= ndev) return fail(NDPP_EINVAL, "device %d outside 0..%d", device, ndev - 1);
  HIP_TRY(hipSetDevice(device));
  return NDPP_OK;
}

int ndpp_get_device(void) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess) return fail(NDPP_EDEVICE, "hipGetDevice failed");
  return dev;
}

int ndpp_integrate_freegas_leg(const ndpp_params* p, double Ein, double A, double kT,
                               const double* fEmu, const double* mu,
                               const double* E_bins, int n_bins, double* distro) {
  int rc = check_params(p, n_bins - 1);
  if (rc) return rc;
  rc = check_mu_grid(p, mu);
  if (rc) return rc;
  const int row = 0;
  // cutoff = +inf: this entry point IS the free-gas routine
  return run_batch_h(p, A, kT, HUGE_VAL, 0.0, 1, &Ein, &row, nullptr, 1, fEmu,
                     n_bins - 1, E_bins, distro, nullptr, 1, nullptr);
}

int ndpp_integrate_file4_cm_leg(const ndpp_params* p, const double* fw, double Ein,
                                double awr, double Q, const double* E_bins,
                                int n_bins, const double* w, double* distro) {
  int rc = check_params(p, n_bins - 1);
  if (rc) return rc;
  rc = check_mu_grid(p, w);
  if (rc) return rc;
  const int row = 0;
  return run_batch_h(p, awr, 0.0, -HUGE_VAL, Q, 1, &Ein, &row, nullptr, 1, fw,
                     n_bins - 1, E_bins, distro, nullptr, 1, nullptr);
}

int ndpp_elastic_leg_batch(const ndpp_params* p, double A, double kT,
                           double freegas_cutoff, double Q, int n_ein,
                           const double* ein, const int* row_lo, const double* w_hi,
                           int n_rows, const double* f_tab, int G,
                           const double* e_bins, double* out, int* status,
                           ndpp_stats* stats) {
  return run_batch_h(p, A, kT, freegas_cutoff, Q, n_ein, ein, row_lo, w_hi, n_rows,
                     f_tab, G, e_bins, out, status, 2, stats);
}

int ndpp_elastic_leg_batch_d(const ndpp_params* p, double A, double kT,
                             double freegas_cutoff, double Q, int n_ein,
                             const double* ein_d, const int* row_lo_d,
                             const double* w_hi_d, int n_rows, const double* f_tab_d,
                             int G, const double* e_bins_d, double* out_d,
                             int* status_d, void* stream, ndpp_stats* stats) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  return run_batch_d(p, A, kT, freegas_cutoff, Q, n_ein, ein_d, row_lo_d, w_hi_d,
                     n_rows, f_tab_d, G, e_bins_d, out_d, status_d, 2,
                     (hipStream_t)stream, stats);
}

int ndpp_elastic_leg_multi_d(const ndpp_params* p, int n_nuc, const double* A_d,
                             const double* kT_d, const double* cutoff_d, const double* Q_d,
                             int n_ein, const double* ein_d, const int* nuc_of_ein_d,
                             const int* row_lo_d, const double* w_hi_d, int n_rows,
                             const double* f_tab_d, int G, const double* e_bins_d, double* out_d,
                             int* status_d, void* stream, ndpp_stats* stats) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  if (n_nuc < 1 || !A_d || !kT_d || !cutoff_d || !Q_d || (n_ein > 0 && !nuc_of_ein_d))
    return fail(NDPP_EINVAL, "n_nuc=%d or NULL per-nuclide array", n_nuc);
  const NucArrays na{n_nuc, A_d, kT_d, cutoff_d, Q_d, nuc_of_ein_d};
  return run_batch_d(p, 0.0, 0.0, 0.0, 0.0, n_ein, ein_d, row_lo_d, w_hi_d, n_rows, f_tab_d, G,
                     e_bins_d, out_d, status_d, 2, (hipStream_t)stream, stats, &na);
}

int ndpp_elastic_leg_multi(const ndpp_params* p, int n_nuc, const double* A, const double* kT,
                           const double* freegas_cutoff, const double* Q, int n_ein,
                           const double* ein, const int* nuc_of_ein, const int* row_lo,
                           const double* w_hi, int n_rows, const double* f_tab, int G,
                           const double* e_bins, double* out, int* status, ndpp_stats* stats) {
  int rc = check_params(p, G);
  if (rc) return rc;
  if (stats) memset(stats, 0, sizeof(*stats));
  if (n_ein < 0 || n_nuc < 1 || n_rows < 2) return fail(NDPP_EINVAL, "n_ein=%d n_nuc=%d n_rows=%d", n_ein, n_nuc, n_rows);
  if (n_ein == 0) return NDPP_OK;
  if (!A || !kT || !freegas_cutoff || !Q || !ein || !nuc_of_ein || !row_lo || !w_hi || !f_tab || !e_bins || !out)
    return fail(NDPP_EINVAL, "NULL array argument");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0)
    return fail(NDPP_EDEVICE, "no HIP device available (libndpp_hip has no CPU path)");
  const size_t GL = (size_t)G * p->order, M = (size_t)p->mu_bins;
  struct Buf {
    void* p = nullptr;
    ~Buf() { if (p) dev_free(p); }
    hipError_t up(const void* h, size_t bytes) {
      hipError_t e = dev_alloc(&p, bytes ? bytes : 8);
      if (e != hipSuccess || !h) return e;
      return hipMemcpy(p, h, bytes, hipMemcpyHostToDevice);
    }
  } dA, dkT, dcut, dQ, dein, dnuc, drow, dw, df, dbins, dout, dst;
#define MULTI_TRY(expr)                                                           \
  do {                                                                            \
    hipError_t e_ = (expr);                                                       \
    if (e_ != hipSuccess)                                                         \
      return fail(NDPP_EDEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));   \
  } while (0)
  MULTI_TRY(dA.up(A, sizeof(double) * n_nuc));
  MULTI_TRY(dkT.up(kT, sizeof(double) * n_nuc));
  MULTI_TRY(dcut.up(freegas_cutoff, sizeof(double) * n_nuc));
  MULTI_TRY(dQ.up(Q, sizeof(double) * n_nuc));
  MULTI_TRY(dein.up(ein, sizeof(double) * n_ein));
  MULTI_TRY(dnuc.up(nuc_of_ein, sizeof(int) * n_ein));
  MULTI_TRY(drow.up(row_lo, sizeof(int) * n_ein));
  MULTI_TRY(dw.up(w_hi, sizeof(double) * n_ein));
  MULTI_TRY(df.up(f_tab, sizeof(double) * (size_t)n_rows * M));
  MULTI_TRY(dbins.up(e_bins, sizeof(double) * (G + 1)));
  MULTI_TRY(dout.up(nullptr, sizeof(double) * n_ein * GL));
  MULTI_TRY(dst.up(nullptr, sizeof(int) * n_ein));
  rc = ndpp_elastic_leg_multi_d(p, n_nuc, (double*)dA.p, (double*)dkT.p, (double*)dcut.p,
                                (double*)dQ.p, n_ein, (double*)dein.p, (int*)dnuc.p, (int*)drow.p,
                                (double*)dw.p, n_rows, (double*)df.p, G, (double*)dbins.p,
                                (double*)dout.p, (int*)dst.p, nullptr, stats);
  if (rc) return rc;
  MULTI_TRY(hipMemcpy(out, dout.p, sizeof(double) * n_ein * GL, hipMemcpyDeviceToHost));
  if (status) MULTI_TRY(hipMemcpy(status, dst.p, sizeof(int) * n_ein, hipMemcpyDeviceToHost));
  return NDPP_OK;
}

}  // extern "C"
