// dev_util.h -- small helpers shared by the translation units of libndpp_hip.so
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstddef>

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

// owning device buffer; upload() = allocate + copy from the host
template <class T>
struct DevBuf {
  T* p = nullptr;
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
  hipError_t upload(const T* h, size_t n) {
    hipError_t e = alloc(n);
    if (e != hipSuccess) return e;
    return (n && h) ? hipMemcpy(p, h, n * sizeof(T), hipMemcpyHostToDevice) : hipSuccess;
  }
};

}  // namespace ndpp
