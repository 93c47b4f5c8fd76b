// fg_device.h -- the device side of the free-gas inner-integral stages (prep, mu, combine)
// and their launchers, shared by the two translation units that instantiate them:
// ndpp_hip.hip (the library's arithmetic: product or strict) and fg_strict_stages.hip (always
// strict: the reference's operation order for the prep stage and for cold incoming energies).
// Everything here is internal linkage; the arithmetic comes from the inline namespace of
// fg_pipeline.h that the including file was compiled for.
#pragma once

#include <hip/hip_runtime.h>

#include "fg_pipeline.h"

namespace ndpp {
namespace {

constexpr int kWave = 64;
// The inner walk is compiled for two waves per SIMD (the register file holds 512 VGPRs per SIMD
// lane: 256 per wave; the joint P5 walk uses 215) and launched as 4 x 2 one-wave workgroups per CU,
// which share the CU's 160 KB of LDS: 20 KB of sibling stack each.  (Three waves -- 168 VGPRs,
// 13 KB: four stack levels -- were built with a spill-free hot loop in round 3 and lost 10 %;
// experiments/README.md.)
constexpr int kMuWavesPerSimd = 2;
constexpr int kMuBlocksPerCU = 4 * kMuWavesPerSimd;  // 1-wave blocks

struct DevAtomics {
  __device__ static int add(int* p, int v) { return atomicAdd(p, v); }
};

// Per-lane direct-mapped stack of right siblings.  The deepest
// lds_levels(R) levels (where >98% of pushes/pops happen) live in LDS,
// lane-interleaved so that a wave's 64 lanes hit 64 distinct banks whatever
// depth each lane is at: [level][field][lane] doubles, ds_read/write_b64 with
// the field as an immediate offset.  Shallower levels spill to a
// lane-interleaved global scratch (coalesced, touched once per ~2^8 nodes).
// An entry has NF = 2 + 2R doubles {b, w, Xb[R], Xe[R]} and the channel mask.
typedef __attribute__((address_space(3))) double lds_f64;
typedef __attribute__((address_space(3))) unsigned lds_u32;

constexpr int mu_stack_fields(int R) { return 2 + 2 * R; }
// stack levels that fit a workgroup's share of the LDS: 8 with one row, 6 with two
constexpr int mu_lds_levels(int R) {
  const int budget = (160 * 1024) / kMuBlocksPerCU - 64;
  const int fit = budget / (kWave * (8 * mu_stack_fields(R) + 4));
  return (R == 1 && fit > kStackLdsLevels) ? kStackLdsLevels : fit;
}

template <int R>
struct DevMuStack {
  static constexpr int NF = mu_stack_fields(R);
  lds_f64* lds;    // [levels][NF][64]  (explicit LDS address space: the compiler
  lds_u32* ldsm;   // [levels][64]       must not fold these with the global path)
  // global part: [d0][nthreads] records of NF doubles + the mask (padded to 8 bytes),
  // addressed as a uniform base + a 32-bit byte offset
  char* gbase;
  unsigned goff, gstride;   // this lane's record of level 0; bytes per level
  int lane, d0;             // levels d0 and deeper live in LDS
  static constexpr unsigned kRecBytes = 8u * (NF + 1);
  __device__ __forceinline__ bool in_lds(int d) const { return d >= d0; }
  __device__ __forceinline__ void push(int d, double b, double w, const double* Xb,
                                       const double* Xe, unsigned m) {
    if (in_lds(d)) {
      const int o = __mul24(d - d0, NF * kWave) + lane;      // (24-bit multiply: full rate)
      lds[o] = b; lds[o + kWave] = w;
#pragma unroll
      for (int r = 0; r < R; ++r) {
        lds[o + (2 + r) * kWave] = Xb[r];
        lds[o + (2 + R + r) * kWave] = Xe[r];
      }
      ldsm[(d - d0) * kWave + lane] = m;
    } else {
      double* p = (double*)(gbase + (size_t)(goff + __umul24((unsigned)d, gstride)));
      p[0] = b; p[1] = w;
#pragma unroll
      for (int r = 0; r < R; ++r) { p[2 + r] = Xb[r]; p[2 + R + r] = Xe[r]; }
      *(unsigned*)(p + NF) = m;
    }
  }
  __device__ __forceinline__ void pop(int d, double& b, double& w, double* Xb,
                                      double* Xe, unsigned& m) const {
    if (in_lds(d)) {
      const int o = __mul24(d - d0, NF * kWave) + lane;      // (24-bit multiply: full rate)
      b = lds[o]; w = lds[o + kWave];
#pragma unroll
      for (int r = 0; r < R; ++r) {
        Xb[r] = lds[o + (2 + r) * kWave];
        Xe[r] = lds[o + (2 + R + r) * kWave];
      }
      m = ldsm[(d - d0) * kWave + lane];
    } else {
      const double* p = (const double*)(gbase + (size_t)(goff + __umul24((unsigned)d, gstride)));
      b = p[0]; w = p[1];
#pragma unroll
      for (int r = 0; r < R; ++r) { Xb[r] = p[2 + r]; Xe[r] = p[2 + R + r]; }
      m = *(const unsigned*)(p + NF);
    }
  }
};

// Once a node kernel has run out of arena it raises *B.overflow and stops creating
// children, but the next level's counter already includes them: every later kernel of
// the chunk must do nothing (the host redoes the chunk with fewer calls).
__global__ void fg_prep_kernel(FgBatch B, int level) {
  if (*B.overflow) return;
  const int base = B.lvl_off(level);
  const int nt = B.n_tasks(level);
  for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < nt;
       t += gridDim.x * blockDim.x)
    fg_prep_task(B, level, base, t);
}

// The hot loop.  Each lane owns one inner integral (kPath: one segment of one) at a
// time and fetches the next from a global counter when done, so a wave only idles
// lanes when the level runs out of work.
template <int R, int LMAX, bool kPath>
__device__ __forceinline__ void mu_wave_loop(const FgBatch& B, int level, int base, int nt,
                                             int* counter, DevMuStack<R>& st) {
  MuLane<R, LMAX> s;
  s.mask = 0;
  const PnConsts pk = make_pn_consts<(R * LMAX <= 12 && LMAX <= 8)>();   // pinned where the register budget allows
  bool active = false, more = true;
  bool counts = true;     // split walk: the item that stands for its integral in the statistics
  unsigned long long n_k = 0, n_v = 0, n_i = 0, n_o = 0;
  unsigned long long w_it = 0, l_it = 0;  // wave-uniform: loop trips, active lanes
  // Tasks are handed out to a WAVE in blocks of consecutive indices (the level's tasks are
  // sorted by order mask): all lanes of a wave then walk integrals with the same active
  // orders however long each one takes, and the per-order blocks of the others are skipped.
  const int kTaskBlock = nt >= (int)(gridDim.x * 16 * kWave) ? 4 * kWave : kWave;
  int blk_next = 0, blk_end = 0;          // wave-uniform
  // Fetching costs the whole wave the set-up of an integral (~half a step).  The single-lane walk
  // pays it every ~50 steps; the split walk's items are 20x shorter, so there the wave waits until
  // kFetchMin lanes are free (or none is busy) and sets them up together.  (12 500-energy shard of
  // the headline grid, 25 items per integral: 1523 ms fetching per free lane, 1421 ms waiting for 8,
  // 1425 for 16, 1506 for 32; 16 equal items: 1481 / 1416 / 1424.)
  constexpr int kFetchMin = kPath ? 8 : 1;
  // (lane masks through the boolean ballot builtin: HIP's __ballot / __any go through an integer
  // select and a second comparison, two vector instructions each, in every iteration)
  auto lanes = [](bool x) -> unsigned long long { return __builtin_amdgcn_ballot_w64(x); };
  for (;;) {
    unsigned long long need = more ? lanes(!active) : 0ull;
    if (kFetchMin > 1 && need && __popcll(need) < kFetchMin && lanes(active)) need = 0;
    if (need) {
      if (blk_next >= blk_end) {
        const int first = __ffsll((long long)need) - 1;      // (wave-uniform, like need itself)
        int b = 0;
        if (threadIdx.x == (unsigned)first) b = atomicAdd(counter, kTaskBlock);
        b = __builtin_amdgcn_readlane(b, first);
        blk_next = b;
        blk_end = (b + kTaskBlock < nt) ? b + kTaskBlock : nt;
        if (b >= nt) more = false;         // the level is handed out
      }
      if (!active && more) {
        const int rank = __popcll(need & ((1ull << threadIdx.x) - 1ull));
        const int t = blk_next + rank;
        if (t < blk_end) {
          if (kPath) mu_init_split<R, LMAX>(B, level, base, t, s);
          else mu_init<R, LMAX>(B, level, base, t, s);
          active = (s.mask != 0);
          if (kPath) counts = (s.path_bits == 0);
          if (active) mu_tot_zero(s);
        }
      }
      const int taken = __popcll(need);
      blk_next = (blk_next + taken < blk_end) ? blk_next + taken : blk_end;
    }
    const unsigned long long act = lanes(active);
    if (!act && !more) break;
    w_it += 1;
    l_it += (unsigned long long)__popcll(act);
    if (active) {
      if (!mu_step<R, LMAX, DevMuStack<R>, kPath>(B, s, st, pk)) {
        mu_finish(B, s, kPath);
        n_k += 2ull * s.visits + 3;
        n_v += s.visits;
        n_o += s.ovisits;
        n_i += counts ? 1 : 0;
        active = false;
      }
    }
  }
  // per-wave totals -> 3 atomics per wave
  for (int o = 32; o > 0; o >>= 1) {
    n_k += __shfl_down(n_k, o);
    n_v += __shfl_down(n_v, o);
    n_i += __shfl_down(n_i, o);
    n_o += __shfl_down(n_o, o);
  }
  if (threadIdx.x == 0) {
    atomicAdd(&B.stats[kStatKEvals], n_k);
    atomicAdd(&B.stats[kStatMuVisits], n_v);
    atomicAdd(&B.stats[kStatMuIntegrals], n_i);
    atomicAdd(&B.stats[kStatOrderVisits], n_o);
    atomicAdd(&B.stats[kStatWaveIters], w_it);
    atomicAdd(&B.stats[kStatLaneIters], l_it);
  }
}

// The hot kernel, one wave per block.  A level with few inner integrals is walked by
// kSplit lanes per integral (otherwise its time is that of its longest integral); the
// two modes give the same bits (fg_pipeline.h kSplitLog2).
template <int R, int LMAX>
__global__ __launch_bounds__(kWave, kMuWavesPerSimd) void fg_mu_kernel(FgBatch B, int level, double* gstack, int* counter) {
  constexpr int NF = mu_stack_fields(R), kLevels = mu_lds_levels(R);
  __shared__ double lds[kLevels * NF * kWave];
  __shared__ unsigned ldsm[kLevels * kWave];
  DevMuStack<R> st;
  st.lds = (lds_f64*)lds;
  st.ldsm = (lds_u32*)ldsm;
  st.gbase = (char*)gstack;
  st.goff = (blockIdx.x * kWave + threadIdx.x) * DevMuStack<R>::kRecBytes;
  st.gstride = gridDim.x * kWave * DevMuStack<R>::kRecBytes;
  st.lane = threadIdx.x;
  // The LDS window: the kLevels deepest levels.  (One level higher -- the very deepest level takes
  // 2 % of the pushes, the one above the window 7 % -- was measured: -6 %.  A lane looks at its top
  // sibling in EVERY visit, and while it works at the bottom of the tree that sibling is one of the
  // deepest: those reads must stay in LDS.)
  st.d0 = B.mu_its > kLevels ? B.mu_its - kLevels : 0;

  if (*B.overflow) return;
  const int base = B.lvl_off(level);
  if (B.split_level(level))
    mu_wave_loop<R, LMAX, true>(B, level, base, B.n_mu_tasks(level) * kSplit, counter, st);
  else
    mu_wave_loop<R, LMAX, false>(B, level, base, B.n_mu_tasks(level), counter, st);
}

// split levels: the segment slots of the level's integrals start at zero (a slot is written by the
// one work item that owns it, or by nobody when everything above it was accepted)
__global__ void fg_seg_zero_kernel(FgBatch B, int level) {
  if (*B.overflow || !B.split_level(level)) return;
  const size_t n = (size_t)B.n_mu_tasks(level) * kSplit * B.nch();
  for (size_t k = blockIdx.x * (size_t)blockDim.x + threadIdx.x; k < n; k += (size_t)gridDim.x * blockDim.x)
    B.seg[k] = 0.0;
}

// One thread per (integral, channel): the channels of an integral's sixteen slots are 8 bytes
// apart, so a wave reads whole 96- or 128-byte runs instead of 64 lines 1.5 KB apart (thread per
// integral: 48 ms of a 2.15 s headline pass).  The sum over the slots is fg_mu_combine_task's, left to right.
__global__ void fg_mu_combine_kernel(FgBatch B, int level) {
  if (*B.overflow || !B.split_level(level)) return;
  const int base = B.lvl_off(level);
  const int nch = B.nch();
  const long tot = (long)B.n_mu_tasks(level) * nch;
  for (long k = blockIdx.x * (long)blockDim.x + threadIdx.x; k < tot; k += (long)gridDim.x * blockDim.x) {
    const int t = (int)(k / nch), ch = (int)(k - (long)t * nch);
    int n, slot;
    fg_task_decode(B, level, base, t, n, slot);
    const unsigned mask = (unsigned)B.node_info[4 * n + 0];
    const int r = ch / B.L, l = ch - r * B.L;
    if (!(mask & chan_bit(r, l))) continue;
    const unsigned gl_rows = B.t_gl ? B.t_gl[B.rec_index(level, base, n, slot)] : 0u;   // done by the Gauss rule
    if (gl_rows >> r & 1u) continue;
    const double* sg = B.seg + (size_t)t * kSplit * nch + ch;
    double sum = 0.0;
    for (int j = 0; j < kSplit; ++j) sum = sum + sg[(size_t)j * nch];
    B.F(slot, ch, n) = sum;    // (slots nobody wrote are zero: x + 0.0 == x)
  }
}

__global__ void fg_setup_kernel(FgBatch B) {
  const int n = B.n_jobs * B.G;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += gridDim.x * blockDim.x)
    fg_setup_group(B, i / B.G, i % B.G);
}

__global__ void fg_node_kernel(FgBatch B, int level) {
  if (*B.overflow) return;
  const int base = B.lvl_off(level);
  const int nn = B.lvl_cnt[level];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nn;
       i += gridDim.x * blockDim.x)
    fg_node_process<DevAtomics>(B, level, base, i);
  if (blockIdx.x == 0 && threadIdx.x == 0)
    atomicAdd(&B.stats[kStatEoutNodes], (unsigned long long)nn);
}

__global__ void fg_reduce_kernel(FgBatch B, int level) {
  if (*B.overflow) return;
  const int base = B.lvl_off(level);
  const int nn = B.lvl_cnt[level];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nn;
       i += gridDim.x * blockDim.x)
    fg_reduce_node(B, base, i);
}

__global__ void fg_assemble_kernel(FgBatch B) {
  if (*B.overflow) return;
  const int n_calls = B.n_jobs * B.R;
  for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < n_calls;
       c += gridDim.x * blockDim.x)
    fg_assemble_call(B, c);
}

// ---- one launcher per stage: the pipeline driver (ndpp_hip.hip) calls the set of the
// library's own arithmetic directly and the strict set through fg_strict_stages.hip ----------
inline int fg_blocks(long n, int threads = 256) {
  long b = (n + threads - 1) / threads;
  return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}
inline void launch_fg_setup(const FgBatch& B, hipStream_t s) {
  hipLaunchKernelGGL(fg_setup_kernel, dim3(fg_blocks((long)B.n_jobs * B.G)), dim3(256), 0, s, B);
}
inline void launch_fg_prep(const FgBatch& B, int level, hipStream_t s) {
  hipLaunchKernelGGL(fg_prep_kernel, dim3(2048), dim3(256), 0, s, B, level);
}
inline void launch_fg_seg_zero(const FgBatch& B, int level, hipStream_t s) {
  hipLaunchKernelGGL(fg_seg_zero_kernel, dim3(4096), dim3(256), 0, s, B, level);
}
inline void launch_fg_combine(const FgBatch& B, int level, hipStream_t s) {
  hipLaunchKernelGGL(fg_mu_combine_kernel, dim3(2048), dim3(256), 0, s, B, level);
}
inline void launch_fg_node(const FgBatch& B, int level, hipStream_t s) {
  hipLaunchKernelGGL(fg_node_kernel, dim3(2048), dim3(256), 0, s, B, level);
}
inline void launch_fg_reduce(const FgBatch& B, int level, hipStream_t s) {
  hipLaunchKernelGGL(fg_reduce_kernel, dim3(2048), dim3(256), 0, s, B, level);
}
inline void launch_fg_assemble(const FgBatch& B, hipStream_t s) {
  hipLaunchKernelGGL(fg_assemble_kernel, dim3(fg_blocks((long)B.n_jobs * B.R)), dim3(256), 0, s, B);
}

// The inner integrals the prep stage flagged for the Gauss rule (fg_pipeline.h mu_gauss_task): one
// thread per task record, before the walk of the level (which skips what is done here).
#if NDPP_FAST
template <int R, int LMAX>
__global__ __launch_bounds__(256) void fg_gauss_kernel(FgBatch B, int level) {
  // Roughly half of a level's inner integrals are in the Gauss zone: each wave gathers the flagged
  // ones of its 64-task chunks in an LDS queue and works on 64 of them at a time, all lanes busy.
  __shared__ int queue[256 / kWave][2 * kWave];
  if (*B.overflow) return;
  const int base = B.lvl_off(level);
  const int nt = B.n_tasks(level);
  const int lane = threadIdx.x & (kWave - 1);
  int* q = queue[threadIdx.x / kWave];
  int pending = 0;                                   // wave-uniform
  unsigned long long nk = 0, ni = 0;
  auto work = [&](int t) {
    const int k = mu_gauss_task<R, LMAX>(B, level, base, t);
    nk += (unsigned long long)k;
    ni += (k > 0 && B.t_gl[t] == (1u << R) - 1u) ? 1ull : 0ull;    // (every row of the job by the rule)
  };
  const int stride = gridDim.x * blockDim.x;
  // far candidates first, then the near ones (deeper certification, graded rule): the lanes of a
  // wave then have work of one kind and similar length
  for (int pass = 0; pass < 2; ++pass) {
    for (int t0 = blockIdx.x * blockDim.x + (threadIdx.x - lane); t0 < nt; t0 += stride) {
      const int t = t0 + lane;
      const unsigned flag = t < nt ? B.t_gl[t] : 0u;
      const bool mine = flag != 0 && ((flag & kGaussNear) != 0) == (pass == 1);
      const unsigned long long m = __builtin_amdgcn_ballot_w64(mine);
      if (mine) q[pending + __popcll(m & ((1ull << lane) - 1ull))] = t;
      pending += __popcll(m);
      __builtin_amdgcn_wave_barrier();
      if (pending >= kWave) {
        pending -= kWave;
        const int t1 = q[pending + lane];
        __builtin_amdgcn_wave_barrier();
        work(t1);
      }
    }
    if (lane < pending) work(q[lane]);
    pending = 0;
    __builtin_amdgcn_wave_barrier();
  }
  for (int o = 32; o > 0; o >>= 1) {
    nk += __shfl_down(nk, o);
    ni += __shfl_down(ni, o);
  }
  if (lane == 0 && (nk | ni)) {
    atomicAdd(&B.stats[kStatKEvals], nk);
    atomicAdd(&B.stats[kStatGaussIntegrals], ni);
  }
}
template <int R, int LMAX>
void launch_gauss(const FgBatch& B, int level, hipStream_t s) {
  hipLaunchKernelGGL((fg_gauss_kernel<R, LMAX>), dim3(4096), dim3(256), 0, s, B, level);
}
#endif
void launch_gauss_any(const FgBatch& B, int level, hipStream_t s) {
#if NDPP_FAST
  if (!B.t_gl) return;
  if (B.R == 2) {
    if (B.L <= 4) launch_gauss<2, 4>(B, level, s);
    else if (B.L <= 6) launch_gauss<2, 6>(B, level, s);
    else launch_gauss<2, 8>(B, level, s);
    return;
  }
  if (B.L <= 4) launch_gauss<1, 4>(B, level, s);
  else if (B.L <= 6) launch_gauss<1, 6>(B, level, s);
  else if (B.L <= 8) launch_gauss<1, 8>(B, level, s);
  else launch_gauss<1, 11>(B, level, s);
#else
  (void)B; (void)level; (void)s;
#endif
}

template <int R, int LMAX>
void launch_mu(const FgBatch& B, int level, int num_cu, double* gs, int* counter, hipStream_t s) {
  const int blocks = num_cu * kMuBlocksPerCU;      // persistent: one wave per block, two per SIMD
  hipLaunchKernelGGL((fg_mu_kernel<R, LMAX>), dim3(blocks), dim3(kWave), 0, s, B, level, gs, counter);
}

// Joint traversal of the two bracketing rows (both arithmetics) for L <= kJointMaxL.
constexpr int kJointMaxL = 8;

// the inner walk of the batch on `level`: one lane type per (rows, orders) shape.  (Separate walks
// for classes of orders -- {0,1,2} / {3,4,5} -- were built in round 3, bit-identical, and measured
// 13 % slower: a lane with half the channels is only ~20 % cheaper per visit and the classes walk
// 1.35x the nodes between them; experiments/README.md.)
void launch_mu_any(const FgBatch& B, int level, int num_cu, double* gs, int* counter, hipStream_t s) {
  if (B.R == 2) {
    if (B.L <= 4) launch_mu<2, 4>(B, level, num_cu, gs, counter, s);
    else if (B.L <= 6) launch_mu<2, 6>(B, level, num_cu, gs, counter, s);
    else launch_mu<2, 8>(B, level, num_cu, gs, counter, s);
    return;
  }
  if (B.L <= 4) launch_mu<1, 4>(B, level, num_cu, gs, counter, s);
  else if (B.L <= 6) launch_mu<1, 6>(B, level, num_cu, gs, counter, s);
  else if (B.L <= 8) launch_mu<1, 8>(B, level, num_cu, gs, counter, s);
  else launch_mu<1, 11>(B, level, num_cu, gs, counter, s);
}

}  // namespace
}  // namespace ndpp
