// Shared device/host helpers for the gfx950 marker-detection kernels.
// Built with -ffp-contract=off: float64 sequences must round exactly like NumPy's.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "../../include/magnify_hip.h"

#define MG_WAVE 64

// Keyed scoring prefilter (mg_score.hip): super-tiles of MG_SCORE_SUBY x MG_SCORE_SUBX centre tiles (128 x 256
// positions) with their edge window as bytes in LDS, row stride 308 B = 77 dwords (odd: rows rotate through the
// banks); radii up to MG_SCORE_MAX_R (window 180 x 308), perimeters up to 2 * MG_SCORE_MAX_PAIRS points.
#define MG_SCORE_TILE 64
#define MG_SCORE_SUBY 2
#define MG_SCORE_SUBX 4
#define MG_SCORE_MAX_R 26
#define MG_SCORE_MAX_PAIRS 80
#define MG_SCORE_WSTRIDE (MG_SCORE_TILE * MG_SCORE_SUBX + 2 * MG_SCORE_MAX_R)

#define MG_CHECK_LAUNCH()                          \
  do {                                             \
    hipError_t e_ = hipGetLastError();             \
    if (e_ != hipSuccess) return MG_ELAUNCH;       \
  } while (0)

static inline hipStream_t mg_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// Clearing small counters / bitmaps: a kernel, not hipMemsetAsync.  The calls of this library are captured into
// hipGraphs by the host side (hotpath.CircleFinder), and memset nodes of one captured graph were found to be replayed
// with another graph's parameters once a second graph had been captured (ROCm 7.2: a counter came back as 0x10101034);
// kernel nodes carry their own arguments.
static __global__ void mg_k_zero_words(uint32_t* __restrict__ p, int64_t n_words) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_words; i += (int64_t)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline hipError_t mg_zero_async(void* p, size_t bytes, hipStream_t s) {  // bytes: a multiple of 4
  const int64_t n = (int64_t)(bytes / 4);
  if (n == 0) return hipSuccess;
  const int blocks = (int)(n < 256 * 1024 ? (n + 255) / 256 : 1024);
  hipLaunchKernelGGL(mg_k_zero_words, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint32_t*>(p), n);
  return hipGetLastError();
}

__host__ __device__ static inline int mg_elem_size(int dtype) {
  return dtype == MG_U8 ? 1 : dtype == MG_U16 ? 2 : dtype == MG_F32 ? 4 : 8;
}

// cv::borderInterpolate(p, n, BORDER_REFLECT_101)
__device__ __forceinline__ int mg_reflect101(int p, int n) {
  if (n == 1) return 0;
  while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
  return p;
}

// Load element idx of a scalar-or-image operand as float64.
__device__ __forceinline__ double mg_load_f64(const void* p, int dtype, int64_t idx) {
  switch (dtype) {
    case MG_U8: return (double)((const uint8_t*)p)[idx];
    case MG_U16: return (double)((const uint16_t*)p)[idx];
    case MG_F32: return (double)((const float*)p)[idx];
    default: return ((const double*)p)[idx];
  }
}

// NaN-propagating max / min, as np.max / np.min.
__device__ __forceinline__ double mg_nanmax(double a, double b) { return (a != a) ? a : (b != b) ? b : (a > b ? a : b); }
__device__ __forceinline__ double mg_nanmin(double a, double b) { return (a != a) ? a : (b != b) ? b : (a < b ? a : b); }

__device__ __forceinline__ double mg_wave_nanmax(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = mg_nanmax(v, __shfl_xor(v, off));
  return v;
}
__device__ __forceinline__ double mg_wave_nanmin(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v = mg_nanmin(v, __shfl_xor(v, off));
  return v;
}

// Look first, with a load that bypasses the (non-coherent) L1: workgroups that finish together would otherwise all
// see the same stale value and all go through the compare-and-swap, one after the other on one cache line.
__device__ __forceinline__ void mg_atomic_nanmax(double* addr, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    double cur = __longlong_as_double((long long)old);
    if (cur != cur) return;                // already NaN
    if (v == v && !(v > cur)) return;      // not larger (and not NaN)
    unsigned long long seen = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
    if (seen == old) return;
    old = seen;
  }
}
__device__ __forceinline__ void mg_atomic_nanmin(double* addr, double v) {
  unsigned long long* a = reinterpret_cast<unsigned long long*>(addr);
  unsigned long long old = __hip_atomic_load(a, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  while (true) {
    double cur = __longlong_as_double((long long)old);
    if (cur != cur) return;
    if (v == v && !(v < cur)) return;
    unsigned long long seen = atomicCAS(a, old, (unsigned long long)__double_as_longlong(v));
    if (seen == old) return;
    old = seen;
  }
}

// Inclusive prefix sum over the 64 lanes on the VALU's DPP paths: Kogge-Stone inside each row of 16
// (row_shr 1, 2, 4, 8), then the row totals (row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2, 3).
__device__ __forceinline__ int mg_wave_scan_incl_i32(int v) {
  v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xF, 0xF, true);
  v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xA, 0xF, false);
  v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xC, 0xF, false);
  return v;
}
__device__ __forceinline__ int mg_wave_sum_i32(int v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ long long mg_wave_sum_i64(long long v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}
__device__ __forceinline__ double mg_wave_sum_f64(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
  return v;
}

// Exclusive prefix sum over the threads of a block (blockDim.x <= 1024, multiple of 64).
// Returns this thread's exclusive prefix; *total receives the block sum in every thread.
__device__ __forceinline__ int mg_block_exscan(int v, int* total) {
  __shared__ int s_wave[16];
  __shared__ int s_total;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  int incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (wave == 0) {
    const int w = lane < nw ? s_wave[lane] : 0;
    int wi = w;
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
      const int t = __shfl_up(wi, off);
      if (lane >= off) wi += t;
    }
    if (lane < nw) s_wave[lane] = wi - w;
    if (lane == nw - 1) s_total = wi;
  }
  __syncthreads();
  const int res = incl - v + s_wave[wave];
  *total = s_total;
  __syncthreads();
  return res;
}

// float32 gradient angle at an edge pixel: Scharr on the blurred image (BORDER_REFLECT_101),
// arctan2(dy, dx) evaluated in float64 and rounded once (utils.py:118-119, 170).
struct __attribute__((packed)) MgUnaligned32 {
  uint32_t v;
};
__device__ __forceinline__ float mg_edge_angle(const uint8_t* __restrict__ pb, int h, int w, int y, int x) {
  if (y >= 1 && y < h - 1 && x >= 1 && x < w - 2) {
    // interior: one (unaligned) 4-byte load per row covers columns x - 1 .. x + 2
    const uint32_t r0 = reinterpret_cast<const MgUnaligned32*>(pb + (int64_t)(y - 1) * w + x - 1)->v;
    const uint32_t r1 = reinterpret_cast<const MgUnaligned32*>(pb + (int64_t)y * w + x - 1)->v;
    const uint32_t r2 = reinterpret_cast<const MgUnaligned32*>(pb + (int64_t)(y + 1) * w + x - 1)->v;
    const int a = r0 & 0xFF, b = (r0 >> 8) & 0xFF, c = (r0 >> 16) & 0xFF;
    const int d = r1 & 0xFF, f = (r1 >> 16) & 0xFF;
    const int g = r2 & 0xFF, hh = (r2 >> 8) & 0xFF, ii = (r2 >> 16) & 0xFF;
    const int dx = 3 * (c - a) + 10 * (f - d) + 3 * (ii - g);
    const int dy = 3 * (g - a) + 10 * (hh - b) + 3 * (ii - c);
    return (float)atan2((double)dy, (double)dx);
  }
  const int ym = mg_reflect101(y - 1, h), yp = mg_reflect101(y + 1, h);
  const int xm = mg_reflect101(x - 1, w), xp = mg_reflect101(x + 1, w);
  const int a = pb[(int64_t)ym * w + xm], b = pb[(int64_t)ym * w + x], c = pb[(int64_t)ym * w + xp];
  const int d = pb[(int64_t)y * w + xm], f = pb[(int64_t)y * w + xp];
  const int g = pb[(int64_t)yp * w + xm], hh = pb[(int64_t)yp * w + x], ii = pb[(int64_t)yp * w + xp];
  const int dx = 3 * (c - a) + 10 * (f - d) + 3 * (ii - g);
  const int dy = 3 * (g - a) + 10 * (hh - b) + 3 * (ii - c);
  return (float)atan2((double)dy, (double)dx);
}
