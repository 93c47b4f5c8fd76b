// dev_util.h -- small helpers shared by the translation units of libndpp_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>
#include <iterator>
#include <map>
#include <mutex>
#include <unordered_map>
#include <utility>

namespace ndpp {

// binary_search_real, search.F90:21-71: 1-based i with a(i) <= v < a(i+1) (v == a(n) gives
// n-1).  Where the reference aborts (v outside [a(1), a(n)], or a NaN) this returns -1; the
// callers either test for it or guard the range first and clamp, so that nothing ever
// indexes with it.
__host__ __device__ inline int bsearch1(const double* a, int n, double v) {
  int L = 1, R = n, it = 0;
  if (v < a[0] || v > a[n - 1]) return -1;
  while (R - L > 1) {
    if (v > a[L - 1] && v < a[L]) return L;
    else if (v > a[R - 2] && v < a[R - 1]) return R - 1;
    const int idx = L + (R - L) / 2;
    const double t = a[idx - 1];
    if (v >= t) L = idx;
    else if (v < t) R = idx;
    if (++it == 64) return -1;
  }
  return L;
}

// same, for callers that have already guarded the range: a NaN stays inside the array
__host__ __device__ inline int bsearch1_clamped(const double* a, int n, double v) {
  const int i = bsearch1(a, n, v);
  return i < 1 ? 1 : i;
}

// grid size for a grid-stride loop over n items
inline int nblk(long n, int threads) {
  return (int)std::max<long>(1, std::min<long>((n + threads - 1) / threads, 1 << 20));
}

// Device memory of the batch calls, cached.  A nuclide is dozens of batch calls (one per level, one
// per reaction, one per table conversion) of a handful of staging buffers each; hipMalloc and
// hipFree cost ~0.1 ms apiece and hipFree drains the device, which on a 423-nuclide library added
// up to 1.5 s of 22.  dev_alloc() hands out a cached block of at least the size asked for (at most
// twice it), dev_free() returns it to the cache WITHOUT waiting for the device -- the callers have
// synchronised before their buffers go out of scope, and everything that touches these buffers
// runs on the null stream or is drained before the call returns; a call that FAILS returns
// through ndpp::fail(), which synchronises the device before the buffers are released.  The cache holds at most
// kDevCacheBytes per device (the largest blocks go first); ndpp_release_workspace() empties it.
// The free-gas workspace is not part of it (ndpp_hip.hip: one block, sized per batch).
struct DevCache {
  static constexpr size_t kDevCacheBytes = (size_t)8 << 30;
  static constexpr int kMaxDev = 64;
  std::mutex mu;
  std::multimap<size_t, void*> idle[kMaxDev];                 // capacity -> block
  std::unordered_map<void*, std::pair<size_t, int>> owner;    // every block handed out or idle: capacity, device
  size_t idle_bytes[kMaxDev] = {};
  static DevCache& get() {
    static DevCache* c = new DevCache;      // (leaked on purpose: the HIP runtime may be gone at exit)
    return *c;
  }
  static int device() {
    int d = 0;
    if (hipGetDevice(&d) != hipSuccess || d < 0 || d >= kMaxDev) d = 0;
    return d;
  }
  void drop_idle(int dev, size_t keep) {     // caller holds mu
    while (idle_bytes[dev] > keep && !idle[dev].empty()) {
      auto it = std::prev(idle[dev].end());
      idle_bytes[dev] -= it->first;
      owner.erase(it->second);
      (void)hipFree(it->second);
      idle[dev].erase(it);
    }
  }
  hipError_t alloc(void** out, size_t bytes) {
    const size_t need = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
    const int dev = device();
    std::lock_guard<std::mutex> lock(mu);
    auto it = idle[dev].lower_bound(need);
    if (it != idle[dev].end() && it->first <= 2 * need + 4096) {
      *out = it->second;
      idle_bytes[dev] -= it->first;
      idle[dev].erase(it);
      return hipSuccess;
    }
    hipError_t e = hipMalloc(out, need);
    if (e != hipSuccess) {                   // out of memory: give back what is cached and try once more
      (void)hipGetLastError();
      drop_idle(dev, 0);
      e = hipMalloc(out, need);
    }
    if (e == hipSuccess) owner[*out] = {need, dev};
    return e;
  }
  void release(void* p) {
    if (!p) return;
    std::lock_guard<std::mutex> lock(mu);
    auto it = owner.find(p);
    if (it == owner.end()) { (void)hipFree(p); return; }     // not ours
    const size_t cap = it->second.first;
    const int dev = it->second.second;
    idle[dev].emplace(cap, p);
    idle_bytes[dev] += cap;
    drop_idle(dev, kDevCacheBytes);
  }
  void trim() {                              // current device
    const int dev = device();
    std::lock_guard<std::mutex> lock(mu);
    drop_idle(dev, 0);
  }
};
inline hipError_t dev_alloc(void** p, size_t bytes) { return DevCache::get().alloc(p, bytes); }
inline void dev_free(void* p) { DevCache::get().release(p); }
inline void dev_cache_trim() { DevCache::get().trim(); }

// owning device buffer; upload() = allocate + copy from the host
template <class T>
struct DevBuf {
  T* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) dev_free(p); }
  hipError_t alloc(size_t n) { return dev_alloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
  hipError_t upload(const T* h, size_t n) {
    hipError_t e = alloc(n);
    if (e != hipSuccess) return e;
    return (n && h) ? hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
  }
};

}  // namespace ndpp
