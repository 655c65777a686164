// A2 flat-field correction (preprocess.py:83-87) fused with A1 stitch (stitch.py:22-39),
// and the per-plane min/max that feeds to_uint8 (utils.py:24-26).
//
// Roofline: HBM.  Algorithmic bytes per pixel (u16): pass 1 reads 2 B, pass 2 reads 2 B and
// writes 2 B -> 6 B/px.  The two global maxima make the second read compulsory.
#include <math.h>

#include "mg_common.h"

namespace {

template <typename T>
struct VecOf;
template <>
struct VecOf<uint8_t> {
  static constexpr int N = 16;
};
template <>
struct VecOf<uint16_t> {
  static constexpr int N = 8;
};
template <>
struct VecOf<float> {
  static constexpr int N = 4;
};
template <>
struct VecOf<double> {
  static constexpr int N = 2;
};

// Load N consecutive elements; one 16-byte load when the address is aligned.
template <typename T, int N>
__device__ __forceinline__ void load_vec(const T* p, T (&v)[N]) {
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    const uint4 raw = *reinterpret_cast<const uint4*>(p);
    __builtin_memcpy(v, &raw, 16);
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = p[j];
  }
}
template <typename T, int N>
__device__ __forceinline__ void store_vec(T* p, const T (&v)[N]) {
  if ((reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    uint4 raw;
    __builtin_memcpy(&raw, v, 16);
    *reinterpret_cast<uint4*>(p) = raw;
  } else {
#pragma unroll
    for (int j = 0; j < N; ++j) p[j] = v[j];
  }
}

// 16-byte accesses at addresses the caller knows to be aligned; `nt`: nontemporal (streamed once, not kept in cache)
typedef uint32_t mg_u32x4 __attribute__((ext_vector_type(4)));
template <typename T, int N>
__device__ __forceinline__ void load_vec16(const T* p, T (&v)[N], bool nt) {
  const mg_u32x4* q = reinterpret_cast<const mg_u32x4*>(p);
  const mg_u32x4 raw = nt ? __builtin_nontemporal_load(q) : *q;
  __builtin_memcpy(v, &raw, 16);
}
template <typename T, int N>
__device__ __forceinline__ void store_vec16(T* p, const T (&v)[N], bool nt) {
  mg_u32x4 raw;
  __builtin_memcpy(&raw, v, 16);
  mg_u32x4* q = reinterpret_cast<mg_u32x4*>(p);
  if (nt) __builtin_nontemporal_store(raw, q);
  else *q = raw;
}

template <typename T>
__device__ __forceinline__ T cast_trunc(double v);
// NumPy's astype from float64 truncates toward zero.
template <>
__device__ __forceinline__ uint8_t cast_trunc<uint8_t>(double v) {
  return (uint8_t)(unsigned int)v;
}
template <>
__device__ __forceinline__ uint16_t cast_trunc<uint16_t>(double v) {
  return (uint16_t)(unsigned int)v;
}
template <>
__device__ __forceinline__ float cast_trunc<float>(double v) {
  return (float)v;
}
template <>
__device__ __forceinline__ double cast_trunc<double>(double v) {
  return v;
}

template <typename T>
struct IsIntegral {
  static constexpr bool value = false;
};
template <>
struct IsIntegral<uint8_t> {
  static constexpr bool value = true;
};
template <>
struct IsIntegral<uint16_t> {
  static constexpr bool value = true;
};

__device__ __forceinline__ void block_atomic_max2(double m1, double m2, double* out) {
  __shared__ double s1[16], s2[16];
  m1 = mg_wave_nanmax(m1);
  m2 = mg_wave_nanmax(m2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    s1[wave] = m1;
    s2[wave] = m2;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + 63) >> 6;
    for (int i = 1; i < nw; ++i) {
      m1 = mg_nanmax(m1, s1[i]);
      m2 = mg_nanmax(m2, s2[i]);
    }
    mg_atomic_nanmax(out, m1);
    mg_atomic_nanmax(out + 1, m2);
  }
}

// ---- pass 1: global maxima -----------------------------------------------------------
// Division-free filter for M2 = max(t / flat): the float32 reciprocal gives the quotient to
// ~2e-7; only pixels whose approximate quotient is within 1e-6 of the running maximum pay for
// the exact float64 division, so the result is the exact maximum of the exact quotients.
__device__ __forceinline__ bool flat_in_range(double fl) { return fl > 1e-30 && fl < 1e30; }

__device__ __forceinline__ void max_step(double t, double fl, bool fast_m2, double& m1, double& m2) {
  m1 = mg_nanmax(m1, t);
  if (fast_m2) return;
  if (flat_in_range(fl) && t == t) {
    if (t == 0.0) {
      m2 = mg_nanmax(m2, 0.0);
      return;
    }
    const double qa = t * (double)__builtin_amdgcn_rcpf((float)fl);
    if (m2 == m2 && qa < m2 * (1.0 - 1e-6)) return;  // provably below the running maximum
  }
  m2 = mg_nanmax(m2, t / fl);
}

// N consecutive dark/flat operands as float64 (image of float32/float64, or the scalar).
template <int N>
__device__ __forceinline__ void load_field(const void* __restrict__ img, int dt, int64_t p, double scalar,
                                           double (&out)[N]) {
  if (!img) {
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = scalar;
  } else if (dt == MG_F32) {
    const float* f = (const float*)img + p;
    if ((N % 4) == 0 && (reinterpret_cast<uintptr_t>(f) & 15) == 0) {
#pragma unroll
      for (int q = 0; q < N / 4; ++q) {
        const float4 v = reinterpret_cast<const float4*>(f)[q];
        out[4 * q] = v.x;
        out[4 * q + 1] = v.y;
        out[4 * q + 2] = v.z;
        out[4 * q + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < N; ++j) out[j] = f[j];
    }
  } else {
    const double* d = (const double*)img + p;
#pragma unroll
    for (int j = 0; j < N; ++j) out[j] = d[j];
  }
}

// Thread = one chunk of N pixels of the tile grid; it walks over the tiles of its group, so the
// dark/flat operands are loaded once per chunk and reused for every tile (channel) of the group.
template <typename T>
__global__ __launch_bounds__(256) void k_flatfield_max(const T* __restrict__ tiles, int64_t tiles_per_group,
                                                        int64_t tile_elems, double dark,
                                                        const void* __restrict__ d_dark, int dark_dt, double flat,
                                                        const void* __restrict__ d_flat, int flat_dt,
                                                        double* __restrict__ out) {
  constexpr int N = VecOf<T>::N;
  const int group = blockIdx.y;
  tiles += (int64_t)group * tiles_per_group * tile_elems;
  out += 2 * group;
  const bool fast_m2 = (d_flat == nullptr) && (flat > 0.0);  // x / flat is monotone: M2 = M1 / flat
  double m1 = -INFINITY, m2 = -INFINITY;
  const int64_t nvec = tile_elems / N;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    double dk[N], fl[N];
    load_field<N>(d_dark, dark_dt, v * N, dark, dk);
    load_field<N>(d_flat, flat_dt, v * N, flat, fl);
    for (int64_t g = 0; g < tiles_per_group; ++g) {
      T x[N];
      load_vec<T, N>(tiles + g * tile_elems + v * N, x);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        double t = (double)x[j] - dk[j];
        t = t < 0.0 ? 0.0 : t;
        max_step(t, fl[j], fast_m2, m1, m2);
      }
    }
  }
  // tail pixels of every tile (tile_elems not a multiple of N)
  for (int64_t p = nvec * N + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < tile_elems; p += stride) {
    const double dk = d_dark ? mg_load_f64(d_dark, dark_dt, p) : dark;
    const double fl = d_flat ? mg_load_f64(d_flat, flat_dt, p) : flat;
    for (int64_t g = 0; g < tiles_per_group; ++g) {
      double t = (double)tiles[g * tile_elems + p] - dk;
      t = t < 0.0 ? 0.0 : t;
      max_step(t, fl, fast_m2, m1, m2);
    }
  }
  if (fast_m2) m2 = (m1 == -INFINITY) ? m1 : m1 / flat;
  block_atomic_max2(m1, m2, out);
}

// Pass 1 for integer pixels with a scalar dark AND a scalar flat > 0: t = max(x - dark, 0) and t / flat are monotone
// in x, so both maxima follow from the integer maximum of the group -- a pure streaming read (the generic kernel
// spends ~10 float64 operations per pixel on the same answer: 143 us instead of ~30 for one 4 x 4096^2 assay).
template <typename T>
__global__ __launch_bounds__(256) void k_flatfield_max_int(const T* __restrict__ tiles, int64_t group_elems, double dark,
                                                            double flat, double* __restrict__ out) {
  constexpr int N = VecOf<T>::N;
  tiles += (int64_t)blockIdx.y * group_elems;
  out += 2 * blockIdx.y;
  uint32_t xmax = 0;
  const int64_t nvec = group_elems / N;  // the launcher guarantees 16-byte alignment of every group
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; v + 3 * stride < nvec; v += 4 * stride) {  // four loads in flight per lane
    T x[4][N];
#pragma unroll
    for (int q = 0; q < 4; ++q) load_vec<T, N>(tiles + (v + q * stride) * N, x[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int j = 0; j < N; ++j) xmax = max(xmax, (uint32_t)x[q][j]);
  }
  for (; v < nvec; v += stride) {
    T x[N];
    load_vec<T, N>(tiles + v * N, x);
#pragma unroll
    for (int j = 0; j < N; ++j) xmax = max(xmax, (uint32_t)x[j]);
  }
  for (int64_t p = nvec * N + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; p < group_elems; p += stride)
    xmax = max(xmax, (uint32_t)tiles[p]);
  double m1 = (double)xmax - dark;  // every workgroup sees at least one pixel (grid <= nvec / 256 + 1)
  m1 = m1 < 0.0 ? 0.0 : m1;
  block_atomic_max2(m1, m1 / flat, out);
}

// Fast path of pass 1 for integer pixels, scalar dark and a float32 flat image: the test "can this
// pixel beat the running maximum of t / flat?" runs in float32 (reciprocal + multiply, error
// < 1e-6 relative against a 1e-5 margin); only pixels that pass pay for the exact float64 division,
// so the result is still the exact maximum of the exact quotients.  M1 comes from the integer max.
template <typename T>
__global__ __launch_bounds__(256) void k_flatfield_max_fast(const T* __restrict__ tiles, int64_t tiles_per_group,
                                                             int64_t tile_elems, double dark,
                                                             const float* __restrict__ d_flat,
                                                             double* __restrict__ out) {
  constexpr int N = VecOf<T>::N;
  const int group = blockIdx.y;
  tiles += (int64_t)group * tiles_per_group * tile_elems;
  out += 2 * group;
  const float dk_f = (float)dark;
  uint32_t xmax = 0;
  bool any = false;
  double m2 = -INFINITY;
  float thr = -INFINITY;
  const int64_t nvec = tile_elems / N;  // the launcher guarantees tile_elems % N == 0
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    float fl[N], rc[N];
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const float4 f = reinterpret_cast<const float4*>(d_flat + v * N)[q];
      fl[4 * q] = f.x, fl[4 * q + 1] = f.y, fl[4 * q + 2] = f.z, fl[4 * q + 3] = f.w;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) rc[j] = __builtin_amdgcn_rcpf(fl[j]);
    any = true;
    // four tiles (channels) of the group per trip: their loads are issued together -- behind the data-dependent
    // test below the compiler keeps them one round trip apart
    for (int64_t g0 = 0; g0 < tiles_per_group; g0 += 4) {
      T x4[4][N];
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (g0 + q < tiles_per_group) load_vec<T, N>(tiles + (g0 + q) * tile_elems + v * N, x4[q]);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        if (g0 + q >= tiles_per_group) break;
        // the float32 test for all N pixels first, one branch for the chunk: after the first few chunks a pixel that
        // can beat the running maximum is rare, and a branch per pixel cost more than the arithmetic
        bool cand = false;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const uint32_t xi = (uint32_t)x4[q][j];
          xmax = max(xmax, xi);
          const float t_f = fmaxf((float)xi - dk_f, 0.0f);
          const bool in_range = fl[j] > 1e-30f && fl[j] < 1e30f;
          cand |= !(in_range && t_f * rc[j] <= thr);
        }
        if (!cand) continue;
#pragma unroll
        for (int j = 0; j < N; ++j) {
          const uint32_t xi = (uint32_t)x4[q][j];
          const float t_f = fmaxf((float)xi - dk_f, 0.0f);
          const bool in_range = fl[j] > 1e-30f && fl[j] < 1e30f;
          if (in_range && t_f * rc[j] <= thr) continue;  // provably below the running maximum
          double t = (double)xi - dark;
          t = t < 0.0 ? 0.0 : t;
          m2 = mg_nanmax(m2, t / (double)fl[j]);
          thr = (m2 == m2 && m2 < 1e30) ? (float)m2 * (1.0f - 1e-5f) : -INFINITY;
        }
      }
    }
  }
  double m1 = -INFINITY;
  if (any) {
    m1 = (double)xmax - dark;
    m1 = m1 < 0.0 ? 0.0 : m1;
  }
  block_atomic_max2(m1, m2, out);
}

// The same pass with the flat image read at an eighth of its size: d_rcmax[v] = the largest float32 reciprocal of the
// N flat values of chunk v (-1 if one of them is outside the range the float32 test is valid in).  A chunk whose
// largest pixel, times that, cannot beat the running maximum is done after a few integer maxima -- its flat values are
// not even loaded; after the first chunks that is all but a handful.  (Before: 64 MB of flat image re-read for every
// assay, 2 GB of the pass's 10.7 GB of traffic at 64 assays, and a float32 multiply / compare per pixel.)
template <int N>
__global__ __launch_bounds__(256) void k_flat_rcmax(const float* __restrict__ d_flat, int64_t nvec, float* __restrict__ d_rcmax) {
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += (int64_t)gridDim.x * blockDim.x) {
    float m = 0.0f;
    bool ok = true;
#pragma unroll
    for (int q = 0; q < N / 4; ++q) {
      const float4 f = reinterpret_cast<const float4*>(d_flat + v * N)[q];
      const float fl[4] = {f.x, f.y, f.z, f.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        ok = ok && fl[j] > 1e-30f && fl[j] < 1e30f;
        m = fmaxf(m, __builtin_amdgcn_rcpf(fl[j]));
      }
    }
    d_rcmax[v] = ok ? m : -1.0f;
  }
}

template <typename T, bool SHARE>
__global__ __launch_bounds__(256) void k_flatfield_max_lean(const T* __restrict__ tiles, int64_t tiles_per_group,
                                                             int64_t tile_elems, double dark,
                                                             const float* __restrict__ d_flat,
                                                             const float* __restrict__ d_rcmax,
                                                             double* __restrict__ out) {
  constexpr int N = VecOf<T>::N;
  constexpr int UV = 2;  // chunk positions per trip: UV x 4 sixteen-byte loads in flight per lane
  const int group = blockIdx.y;
  tiles += (int64_t)group * tiles_per_group * tile_elems;
  out += 2 * group;
  const float dk_f = (float)dark;
  uint32_t xmax = 0;
  bool any = false;
  double m2 = -INFINITY;
  float thr = -INFINITY;
  const int64_t nvec = tile_elems / N;  // the launcher guarantees tile_elems % N == 0
  // (grid-strided, nontemporal loads.  Measured beside it: contiguous spans per workgroup -- 7.0 TB/s against 5.5 in a
  // plain read, tools/micro/copy_bw.hip -- 1.66 ms here instead of 1.51: four tiles 32 MiB apart walk in step)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t vend = nvec;
  // The threshold a lane tests against comes from its own running maximum AND from the group's (out[1]): the first
  // PUBLISHERS workgroups of a group add the maximum of their first trip to it (a wave at a time, look-first), every
  // lane looks at it on trips 1, 2, 4, 8, ....  A lane of a small batch sees a few dozen chunks: on its own maximum alone a sixth
  // of them took the exact path (one 4 x 4096^2 assay: 81 -> 62 us; 8 assays: 277 -> 239 us).  Any value found there
  // is the exact quotient of a pixel some lane holds in its m2, so a pixel skipped against it cannot be the maximum.
  // SHARE is off where a lane has hundreds of chunks of its own (64 assays: the sharing cost 5 %).
  constexpr int PUBLISHERS = 32;
  float gthr = -INFINITY;
  int trip = 0;
  const bool publish = SHARE && blockIdx.x < PUBLISHERS && (int64_t)(blockIdx.x + 1) * blockDim.x <= nvec;  // (whole waves on trip 0)
  for (int64_t v0 = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v0 < nvec; v0 += UV * stride) {
    float rcm[UV];
#pragma unroll
    for (int u = 0; u < UV; ++u) rcm[u] = d_rcmax[min(v0 + u * stride, nvec - 1)];
    if (SHARE && trip > 0 && (trip & (trip - 1)) == 0) {  // trips 1, 2, 4, 8, ...: a look every trip cost 10 % at 64 assays
      const double gm = __hip_atomic_load(out + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (gm == gm && gm < 1e30 && gm > -INFINITY) gthr = fmaxf(gthr, (float)gm * (1.0f - 1e-5f));
    }
    any = true;
    for (int64_t g0 = 0; g0 < tiles_per_group; g0 += 4) {  // four tiles (channels) of the group per trip, loads together
      T x4[UV][4][N];
#pragma unroll
      for (int u = 0; u < UV; ++u)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          load_vec16<T, N>(tiles + min(g0 + q, tiles_per_group - 1) * tile_elems + min(v0 + u * stride, nvec - 1) * N, x4[u][q],
                           true);
#pragma unroll
      for (int u = 0; u < UV; ++u) {
        const int64_t v = v0 + u * stride;
        if (v >= vend) break;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          if (g0 + q >= tiles_per_group) break;
          uint32_t cm = 0;
#pragma unroll
          for (int j = 0; j < N; ++j) cm = max(cm, (uint32_t)x4[u][q][j]);
          xmax = max(xmax, cm);
          // t -> float(x) - dark -> max(., 0) and the product with a non-negative reciprocal are monotone: if the chunk's
          // largest pixel with the chunk's largest reciprocal stays at or below the threshold, every pixel does
          const float tf = fmaxf((float)cm - dk_f, 0.0f);
          if (rcm[u] >= 0.0f && tf * rcm[u] <= fmaxf(thr, gthr)) continue;
          float fl[N];
#pragma unroll
          for (int k4 = 0; k4 < N / 4; ++k4) {
            const float4 f = reinterpret_cast<const float4*>(d_flat + v * N)[k4];
            fl[4 * k4] = f.x, fl[4 * k4 + 1] = f.y, fl[4 * k4 + 2] = f.z, fl[4 * k4 + 3] = f.w;
          }
#pragma unroll
          for (int j = 0; j < N; ++j) {
            const uint32_t xi = (uint32_t)x4[u][q][j];
            const float t_f = fmaxf((float)xi - dk_f, 0.0f);
            const bool in_range = fl[j] > 1e-30f && fl[j] < 1e30f;
            if (in_range && t_f * __builtin_amdgcn_rcpf(fl[j]) <= fmaxf(thr, gthr)) continue;  // provably below the running maximum
            double t = (double)xi - dark;
            t = t < 0.0 ? 0.0 : t;
            m2 = mg_nanmax(m2, t / (double)fl[j]);
            thr = (m2 == m2 && m2 < 1e30) ? (float)m2 * (1.0f - 1e-5f) : -INFINITY;
          }
        }
      }
    }
    if (trip == 0 && publish) {  // (block-uniform: every lane of the wave is here)
      const double wm = mg_wave_nanmax(m2);
      if ((threadIdx.x & 63) == 0 && wm == wm && wm > -INFINITY) mg_atomic_nanmax(out + 1, wm);
    }
    ++trip;
  }
  double m1 = -INFINITY;
  if (any) {
    m1 = (double)xmax - dark;
    m1 = m1 < 0.0 ? 0.0 : m1;
  }
  block_atomic_max2(m1, m2, out);
}

// ---- pass 2: apply + stitch (+ output min/max) ----------------------------------------
constexpr int ROWS_PER_BLOCK = 32;  // rows of a workgroup at large batches; fewer when the grid would not fill the chip

// out = trunc(((t / fl) * m1) / m2) for an integer output type without the two float64 divisions:
// v = t * rcp(fl) * (m1 / m2) with a Newton-refined reciprocal agrees with the reference's three
// roundings to ~1e-15 relative, so the truncation is the same unless v lies within 1e-6 of an
// integer -- those (rare) pixels take the exact path.
// Newton-refined reciprocal of a flat-field value (float32 seed, two steps in float64: ~1e-16).
// Returns 0 when the value is outside the range in which the fast path is valid.
__device__ __forceinline__ double refined_rcp(double fl) {
  if (!flat_in_range(fl)) return 0.0;
  double r = (double)__builtin_amdgcn_rcpf((float)fl);
  r = r * (2.0 - fl * r);
  r = r * (2.0 - fl * r);
  return r;
}

// out = trunc(((t / fl) * m1) / m2) for an integer output type.  With r = refined_rcp(fl) != 0 and
// k = m1 / m2, v = t * r * k agrees with the reference's three roundings to ~1e-15 relative, so the
// truncation is the same unless v lies within 1e-6 of an integer -- those (rare) pixels, and every
// non-integer output type, take the exact two-division path.
template <typename T>
__device__ __forceinline__ T correct_pixel(double t, double fl, double r, double m1, double m2, double k, bool fast_ok) {
  if (IsIntegral<T>::value && fast_ok && r != 0.0) {
    if (t == 0.0) return (T)0;  // 0 / fl * m1 / m2 == 0 exactly (m1, m2 finite and positive here)
    const double v = t * r * k;
    const double fv = floor(v);
    const double fr = v - fv;
    if (fr > 1e-6 && fr < 1.0 - 1e-6 && v < 4.0e9) return (T)(unsigned int)fv;
  }
  double e = t / fl;
  e = e * m1;
  e = e / m2;
  return cast_trunc<T>(e);
}

// N pixels at once for an integer output type: the fast products of all N first (straight-line code), ONE test whether
// any of them sits too close to an integer (or the operands are out of the fast path's range), and only then -- a few
// chunks in a million -- the exact two-division path for the chunk.  (A branch per pixel made the pass VALU / branch
// bound: ~25 vector and ~6 scalar instructions per pixel; the data path itself is a dozen.)
template <typename T, int N>
__device__ __forceinline__ void correct_chunk(const T (&x)[N], const double (&dk)[N], const double (&fl)[N],
                                              const double (&r)[N], double m1, double m2, double k, bool fast_ok, T (&o)[N]) {
  double t[N];
  bool unsure = !fast_ok || !IsIntegral<T>::value;  // (other output types: always the reference's own operations)
#pragma unroll
  for (int j = 0; j < N; ++j) {
    t[j] = (double)x[j] - dk[j];
    t[j] = t[j] < 0.0 ? 0.0 : t[j];
    if (!IsIntegral<T>::value) continue;
    const double v = t[j] * r[j] * k;
    const double fv = floor(v);
    const double fr = v - fv;
    o[j] = (T)(unsigned int)fv;
    // t == 0 gives exactly 0 (m1, m2 finite and positive under fast_ok); r == 0 marks a flat value outside the range
    unsure |= !((fr > 1e-6 && fr < 1.0 - 1e-6 && v < 4.0e9) || t[j] == 0.0) || r[j] == 0.0;
  }
  if (unsure) {
#pragma unroll
    for (int j = 0; j < N; ++j) {
      double e = t[j] / fl[j];
      e = e * m1;
      e = e / m2;
      o[j] = cast_trunc<T>(e);
    }
  }
}

constexpr int PLANES_PER_BLOCK = 8;

// Block = 256 lanes x N pixels of ROWS_PER_BLOCK output rows, for PLANES_PER_BLOCK consecutive
// planes: the dark/flat operands of a pixel chunk are loaded once and reused across those planes.
template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void k_apply_stitch(const T* __restrict__ tiles, int n_planes, int n_tr, int n_tc,
                                                       int ty, int tx, int clip, int hy, int hx, int planes_per_group,
                                                       double dark, const void* __restrict__ d_dark, int dark_dt,
                                                       double flat, const void* __restrict__ d_flat, int flat_dt,
                                                       const double* __restrict__ d_max2, T* __restrict__ image,
                                                       double* __restrict__ d_minmax, int rows_per_block) {
  constexpr int N = VecOf<T>::N;
  constexpr int PB = PLANES_PER_BLOCK;
  const int plane0 = blockIdx.z * PB;
  const int np = min(PB, n_planes - plane0);
  const int h_out = n_tr * hy, w_out = n_tc * hx;
  const int ox0 = (blockIdx.x * blockDim.x + threadIdx.x) * N;
  double m1[PB], m2[PB], kk[PB];
  bool fast_ok[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    m1[b] = 0.0, m2[b] = 1.0, kk[b] = 1.0, fast_ok[b] = false;
    if (APPLY && b < np) {
      const int group = (plane0 + b) / planes_per_group;
      m1[b] = d_max2[2 * group];
      m2[b] = d_max2[2 * group + 1];
      kk[b] = m1[b] / m2[b];
      fast_ok[b] = kk[b] > 0.0 && kk[b] < 1e30 && m1[b] > 0.0 && m1[b] < 1e300 && m2[b] > 0.0 && m2[b] < 1e300;
    }
  }
  double vmin[PB], vmax[PB];
  uint32_t imin[PB], imax[PB];  // integer outputs: min/max in integer registers
#pragma unroll
  for (int b = 0; b < PB; ++b) vmin[b] = INFINITY, vmax[b] = -INFINITY, imin[b] = 0xFFFFFFFFu, imax[b] = 0u;
  const int64_t tile_elems = (int64_t)ty * tx;
  // a workgroup takes the row groups blockIdx.y, blockIdx.y + gridDim.y, ...: one set of min/max atomics per
  // workgroup however short the row groups are
  if (ox0 < w_out)
  for (int yg = blockIdx.y; yg * rows_per_block < h_out; yg += gridDim.y) {
    const int row_end = min((yg + 1) * rows_per_block, h_out);
    const int tc0 = ox0 / hx;
    const int x0 = ox0 - tc0 * hx + clip;
    const bool one_tile = (ox0 + N <= w_out) && (x0 - clip + N <= hx);
    for (int oy = yg * rows_per_block; oy < row_end; ++oy) {
      const int tr = oy / hy;
      const int y = oy - tr * hy + clip;
      int64_t pix[N], toff[N];  // pixel index inside the tile, element offset of the tile in a plane
      const int cnt = one_tile ? N : min(N, w_out - ox0);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        const int ox = ox0 + (one_tile ? j : min(j, cnt - 1));
        const int tc = one_tile ? tc0 : ox / hx;
        const int xx = one_tile ? x0 + j : ox - tc * hx + clip;
        pix[j] = (int64_t)y * tx + xx;
        toff[j] = ((int64_t)tr * n_tc + tc) * tile_elems;
      }
      double dk[N], fl[N], rr[N];
      if (APPLY) {
        if (one_tile) {
          load_field<N>(d_dark, dark_dt, pix[0], dark, dk);
          load_field<N>(d_flat, flat_dt, pix[0], flat, fl);
        } else {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            dk[j] = d_dark ? mg_load_f64(d_dark, dark_dt, pix[j]) : dark;
            fl[j] = d_flat ? mg_load_f64(d_flat, flat_dt, pix[j]) : flat;
          }
        }
        // the refined reciprocal of flat is shared by all planes of the block
#pragma unroll
        for (int j = 0; j < N; ++j) rr[j] = IsIntegral<T>::value ? refined_rcp(fl[j]) : 0.0;
      }
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        if (b >= np) break;
        const int64_t plane_base = (int64_t)(plane0 + b) * n_tr * n_tc * tile_elems;
        T x[N], o[N];
        if (one_tile) {
          load_vec<T, N>(tiles + plane_base + toff[0] + pix[0], x);
        } else {
#pragma unroll
          for (int j = 0; j < N; ++j) x[j] = tiles[plane_base + toff[j] + pix[j]];
        }
#pragma unroll
        for (int j = 0; j < N; ++j) {
          if (APPLY) {
            double t = (double)x[j] - dk[j];
            t = t < 0.0 ? 0.0 : t;
            o[j] = correct_pixel<T>(t, fl[j], rr[j], m1[b], m2[b], kk[b], fast_ok[b]);
          } else {
            o[j] = x[j];
          }
          if (d_minmax && j < cnt) {
            if (IsIntegral<T>::value) {
              imin[b] = min(imin[b], (uint32_t)o[j]);
              imax[b] = max(imax[b], (uint32_t)o[j]);
            } else {
              const double ov = (double)o[j];
              vmin[b] = mg_nanmin(vmin[b], ov);
              vmax[b] = mg_nanmax(vmax[b], ov);
            }
          }
        }
        T* dst = image + ((int64_t)(plane0 + b) * h_out + oy) * w_out + ox0;
        if (cnt == N) {
          store_vec<T, N>(dst, o);
        } else {
          for (int j = 0; j < cnt; ++j) dst[j] = o[j];
        }
      }
    }
  }
  if (d_minmax) {
    __shared__ double smin[PB][4], smax[PB][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      if (IsIntegral<T>::value && imin[b] <= imax[b]) vmin[b] = (double)imin[b], vmax[b] = (double)imax[b];
      const double a = mg_wave_nanmin(vmin[b]), c = mg_wave_nanmax(vmax[b]);
      if (lane == 0) {
        smin[b][wave] = a;
        smax[b][wave] = c;
      }
    }
    __syncthreads();
    if (threadIdx.x < np) {
      const int b = threadIdx.x;
      double a = smin[b][0], c = smax[b][0];
      for (int i = 1; i < 4; ++i) {
        a = mg_nanmin(a, smin[b][i]);
        c = mg_nanmax(c, smax[b][i]);
      }
      if (!(a == INFINITY && c == -INFINITY)) {
        mg_atomic_nanmin(d_minmax + 2 * (plane0 + b), a);
        mg_atomic_nanmax(d_minmax + 2 * (plane0 + b) + 1, c);
      }
    }
  }
}

// A value that is the same in every lane, held in scalar registers (the compiler keeps uniform float64 values in
// vector registers otherwise: 2 per value and lane).
__device__ __forceinline__ double uniform_f64(double v) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)u), hi = __builtin_amdgcn_readfirstlane((uint32_t)(u >> 32));
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// max(x - d, 0) of N integer pixels in the integer domain (d: an integer-valued scalar dark in [0, 65535]); uint16
// pixels two at a time (v_pk_sub_u16 with clamp).
typedef unsigned short mg_u16x2 __attribute__((ext_vector_type(2)));
template <typename T, int N>
__device__ __forceinline__ void sub_dark_int(const T (&x)[N], uint32_t d, uint32_t (&ti)[N]) {
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const uint32_t xi = (uint32_t)x[j];
    ti[j] = xi > d ? xi - d : 0u;
  }
}
template <>
__device__ __forceinline__ void sub_dark_int<uint16_t, 8>(const uint16_t (&x)[8], uint32_t d, uint32_t (&ti)[8]) {
  uint32_t w[4];
  __builtin_memcpy(w, x, 16);
  const mg_u16x2 dd = {(unsigned short)d, (unsigned short)d};
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const mg_u16x2 r = __builtin_elementwise_sub_sat(__builtin_bit_cast(mg_u16x2, w[q]), dd);
    ti[2 * q] = r.x;
    ti[2 * q + 1] = r.y;
  }
}

// The fast products of a chunk against per-position factors rk = rcp(flat) * (M1 / M2) (made once per position and
// group, not per pixel): v = t * rk agrees with the reference's three roundings to ~1e-15 relative; the integer part
// is the conversion's own truncation (v >= 0), the distance to the next integer comes from v_fract_f64.  Returns
// whether any pixel sits within 1e-6 of an integer (t == 0 gives exactly 0 either way and does not count).
template <typename T, int N>
__device__ __forceinline__ bool fast_chunk_int(const uint32_t (&ti)[N], const double (&rk)[N], T (&o)[N]) {
  bool unsure = false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double v = (double)ti[j] * rk[j];
    const double fr = __builtin_amdgcn_fract(v);
    o[j] = (T)(unsigned int)v;
    unsure |= !(fr > 1e-6 && fr < 1.0 - 1e-6) && ti[j] != 0u;
  }
  return unsure;
}
template <typename T, int N>
__device__ __forceinline__ bool fast_chunk_f64(const double (&t)[N], const double (&rk)[N], T (&o)[N]) {
  bool unsure = false;
#pragma unroll
  for (int j = 0; j < N; ++j) {
    const double v = t[j] * rk[j];
    const double fr = __builtin_amdgcn_fract(v);
    o[j] = (T)(unsigned int)v;
    unsure |= !(fr > 1e-6 && fr < 1.0 - 1e-6) && t[j] != 0.0;
  }
  return unsure;
}

// Lean variant for the aligned case (hx % N == 0 and aligned bases: every N-pixel chunk lies
// inside one tile and all accesses are 16-byte vectors); integer pixel types only.
// Arithmetic of the correction, per pixel: the conversion, ONE float64 product with the position's factor, fract,
// the truncating conversion and three compares (15 vector instructions a pixel, measured, where the per-pixel form --
// subtract, clip, two products, floor, subtract, five compares, the group's M1 / M2 divided anew in every row -- took
// 24: 2.55 -> 1.65 ms of vector issue at 64 assays; what it bought is at small batches, 0.55 -> 0.50 ms at 8 assays --
// at 64 the pass waits for memory, see the workgroup order below).  INT_DARK: an integer-valued scalar dark,
// subtracted in the integer domain (two uint16 pixels per instruction).
template <typename T, bool APPLY, bool INT_DARK>
__global__ __launch_bounds__(256) void k_apply_stitch_aligned(const T* __restrict__ tiles, int n_planes, int n_tr,
                                                               int n_tc, int ty, int tx, int clip, int hy, int hx,
                                                               int planes_per_group, double dark,
                                                               const void* __restrict__ d_dark, int dark_dt,
                                                               double flat, const void* __restrict__ d_flat,
                                                               int flat_dt, const double* __restrict__ d_max2,
                                                               T* __restrict__ image, double* __restrict__ d_minmax,
                                                               int rows_per_block) {
  constexpr int N = VecOf<T>::N;
  constexpr int PB = PLANES_PER_BLOCK;
  // Which part of the grid this workgroup is: the plane groups (z) of one (x, y) part read the same rows of the flat /
  // dark images -- 64 MB of float32 per plane group at 4096^2, 2 GB of the pass's 19 GB at 64 assays when the parts
  // are worked through plane group by plane group (the hardware's order: x, y, then z).  Workgroups are dealt to the
  // eight XCDs in turn, each XCD has its own L2: XCD k takes the (x, y) parts k, k + 8, ... and runs all plane groups
  // of a part one after the other, so the flat rows of a part are fetched once and then found in that XCD's L2.
  int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
  if (APPLY && gridDim.z >= 16 && ((gridDim.x * gridDim.y) & 7) == 0) {  // (4 plane groups: 0.50 -> 0.55 ms; 32: 3.76 -> 3.47)
    const uint32_t lin = blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z);
    const uint32_t xcd = lin & 7, slot = lin >> 3;
    bz = slot % gridDim.z;
    const uint32_t xy = (slot / gridDim.z) * 8 + xcd;
    bx = xy % gridDim.x;
    by = xy / gridDim.x;
  }
  const int plane0 = bz * PB;
  const int np = min(PB, n_planes - plane0);
  const int h_out = n_tr * hy, w_out = n_tc * hx;
  const int ox0 = (bx * blockDim.x + threadIdx.x) * N;
  uint32_t imin[PB], imax[PB];
#pragma unroll
  for (int b = 0; b < PB; ++b) imin[b] = 0xFFFFFFFFu, imax[b] = 0u;
  const int64_t tile_elems = (int64_t)ty * tx, plane_elems = (int64_t)n_tr * n_tc * tile_elems;
  // per plane, once per workgroup (scalar registers): the quotient of the group's maxima and whether the fast path
  // holds for them (the maxima themselves are read again by the rare exact path: 16 more scalar registers spilled)
  double kka[PB];
  uint32_t ok_mask = 0;
#pragma unroll
  for (int b = 0; b < PB; ++b) {
    kka[b] = 1.0;
    if (APPLY && b < np) {
      const int group = (plane0 + b) / planes_per_group;
      const double m1 = d_max2[2 * group], m2 = d_max2[2 * group + 1];
      const double kk = m1 / m2;
      kka[b] = uniform_f64(kk);
      const bool ok = kk > 0.0 && kk < 1e30 && m1 > 0.0 && m1 < 1e300 && m2 > 0.0 && m2 < 1e300;
      ok_mask |= ok ? (1u << b) : 0u;
    }
  }
  ok_mask = __builtin_amdgcn_readfirstlane(ok_mask);
  const uint32_t dark_i = INT_DARK ? (uint32_t)dark : 0u;
  if (ox0 < w_out)
  for (int yg = by; yg * rows_per_block < h_out; yg += gridDim.y) {  // (as in k_apply_stitch)
    const int row_end = min((yg + 1) * rows_per_block, h_out);
    const int tc0 = ox0 / hx;
    const int x0 = ox0 - tc0 * hx + clip;
    for (int oy = yg * rows_per_block; oy < row_end; ++oy) {
      const int tr = oy / hy;
      const int64_t p0 = (int64_t)(oy - tr * hy + clip) * tx + x0;
      const int64_t src0 = ((int64_t)tr * n_tc + tc0) * tile_elems + p0;
      double dk[N], fl[N], rr[N];
      bool flat_bad = false;  // a flat value outside the range the reciprocal path is valid in
      if (APPLY) {
        if (!INT_DARK) load_field<N>(d_dark, dark_dt, p0, dark, dk);
        load_field<N>(d_flat, flat_dt, p0, flat, fl);
#pragma unroll
        for (int j = 0; j < N; ++j) {
          rr[j] = refined_rcp(fl[j]);
          flat_bad |= rr[j] == 0.0;
        }
      }
      // all planes' loads are issued before the first pixel is corrected (more bytes in flight per lane)
      T xin[PB][N];
#pragma unroll
      for (int b = 0; b < PB; ++b)
        if (b < np) load_vec16<T, N>(tiles + (int64_t)(plane0 + b) * plane_elems + src0, xin[b], false);
      double rk[N];
      double kk_of_rk = -1.0;  // the quotient rk was made with (planes of one group follow each other)
      bool rk_large = true;
#pragma unroll
      for (int b = 0; b < PB; ++b) {
        if (b >= np) break;
        T o[N];
        T(&x)[N] = xin[b];
        if (APPLY) {
          if (kka[b] != kk_of_rk) {  // (uniform: a new group)
            kk_of_rk = kka[b];
            rk_large = false;
#pragma unroll
            for (int j = 0; j < N; ++j) {
              rk[j] = rr[j] * kk_of_rk;
              rk_large |= !(rk[j] < 61035.0);  // t <= 65535: v = t rk stays below 4e9 (the unsigned conversion)
            }
          }
          const bool rk_bad = rk_large || flat_bad || !((ok_mask >> b) & 1u);
          bool unsure;
          uint32_t ti[N];
          double t[N];
          if (INT_DARK) {
            sub_dark_int<T, N>(x, dark_i, ti);
            unsure = fast_chunk_int<T, N>(ti, rk, o);
          } else {
#pragma unroll
            for (int j = 0; j < N; ++j) {
              t[j] = (double)x[j] - dk[j];
              t[j] = t[j] < 0.0 ? 0.0 : t[j];
            }
            unsure = fast_chunk_f64<T, N>(t, rk, o);
          }
          if (unsure || rk_bad) {  // a few chunks in a million: the reference's own operations
            const int group = (plane0 + b) / planes_per_group;
            const double m1 = d_max2[2 * group], m2 = d_max2[2 * group + 1];
#pragma unroll
            for (int j = 0; j < N; ++j) {
              const double tj = INT_DARK ? (double)ti[j] : t[j];
              double e = tj / fl[j];
              e = e * m1;
              e = e / m2;
              o[j] = cast_trunc<T>(e);
            }
          }
        } else {
#pragma unroll
          for (int j = 0; j < N; ++j) o[j] = x[j];
        }
        if (d_minmax) {
#pragma unroll
          for (int j = 0; j < N; ++j) {
            imin[b] = min(imin[b], (uint32_t)o[j]);
            imax[b] = max(imax[b], (uint32_t)o[j]);
          }
        }
        store_vec16<T, N>(image + ((int64_t)(plane0 + b) * h_out + oy) * w_out + ox0, o, false);
      }
    }
  }
  if (d_minmax) {
    __shared__ uint32_t smin[PB][4], smax[PB][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int b = 0; b < PB; ++b) {
      uint32_t a = imin[b], c = imax[b];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a = min(a, (uint32_t)__shfl_xor((int)a, off));
        c = max(c, (uint32_t)__shfl_xor((int)c, off));
      }
      if (lane == 0) smin[b][wave] = a, smax[b][wave] = c;
    }
    __syncthreads();
    if (threadIdx.x < np) {
      const int b = threadIdx.x;
      const uint32_t a = min(min(smin[b][0], smin[b][1]), min(smin[b][2], smin[b][3]));
      const uint32_t c = max(max(smax[b][0], smax[b][1]), max(smax[b][2], smax[b][3]));
      if (a <= c) {
        mg_atomic_nanmin(d_minmax + 2 * (plane0 + b), (double)a);
        mg_atomic_nanmax(d_minmax + 2 * (plane0 + b) + 1, (double)c);
      }
    }
  }
}

// ---- per-plane min/max of strided planes ------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_plane_minmax(const T* __restrict__ src, int64_t plane_stride, int h, int w,
                                                       int64_t row_stride, double* __restrict__ d_minmax) {
  constexpr int N = VecOf<T>::N;
  const int plane = blockIdx.y;
  const T* base = src + (int64_t)plane * plane_stride;
  double vmin = INFINITY, vmax = -INFINITY;
  const int vec_per_row = (w + N - 1) / N;
  const int64_t total = (int64_t)h * vec_per_row;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (IsIntegral<T>::value && w % N == 0 && (reinterpret_cast<uintptr_t>(base) & 15) == 0 && (row_stride * (int64_t)sizeof(T)) % 16 == 0) {
    // integer pixels, rows of whole 16-byte vectors: integer min / max, four loads in flight per lane (one 4096^2
    // uint16 plane: 28 -> ~10 us; the float64 compares of the general loop kept the lanes busy, one load at a time)
    uint32_t imin = 0xFFFFFFFFu, imax = 0u;
    for (; i < total; i += 4 * stride) {
      T x[4][N];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int64_t iq = min(i + q * stride, total - 1);  // (a clamped repeat does not change a min / max)
        const int r = (int)(iq / vec_per_row);
        const int c0 = (int)(iq - (int64_t)r * vec_per_row) * N;
        load_vec16<T, N>(base + (int64_t)r * row_stride + c0, x[q], false);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < N; ++j) {
          imin = min(imin, (uint32_t)x[q][j]);
          imax = max(imax, (uint32_t)x[q][j]);
        }
    }
    if (imin <= imax) vmin = (double)imin, vmax = (double)imax;
    i = total;
  }
  for (; i < total; i += stride) {
    const int r = (int)(i / vec_per_row);
    const int c0 = (int)(i - (int64_t)r * vec_per_row) * N;
    const T* p = base + (int64_t)r * row_stride + c0;
    if (c0 + N <= w) {
      T x[N];
      load_vec<T, N>(p, x);
#pragma unroll
      for (int j = 0; j < N; ++j) {
        vmin = mg_nanmin(vmin, (double)x[j]);
        vmax = mg_nanmax(vmax, (double)x[j]);
      }
    } else {
      for (int j = 0; c0 + j < w; ++j) {
        vmin = mg_nanmin(vmin, (double)p[j]);
        vmax = mg_nanmax(vmax, (double)p[j]);
      }
    }
  }
  __shared__ double smin[4], smax[4];
  vmin = mg_wave_nanmin(vmin);
  vmax = mg_wave_nanmax(vmax);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    smin[wave] = vmin;
    smax[wave] = vmax;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int i = 1; i < 4; ++i) {
      vmin = mg_nanmin(vmin, smin[i]);
      vmax = mg_nanmax(vmax, smax[i]);
    }
    if (!(vmin == INFINITY && vmax == -INFINITY)) {
      mg_atomic_nanmin(d_minmax + 2 * plane, vmin);
      mg_atomic_nanmax(d_minmax + 2 * plane + 1, vmax);
    }
  }
}

template <typename T>
int launch_max(const void* d_tiles, int64_t tiles_per_group, int n_groups, int64_t tile_elems, double dark,
               const void* d_dark, int dark_dt, double flat, const void* d_flat, int flat_dt, double* d_max2,
               float* d_scratch, int64_t scratch_floats, hipStream_t s) {
  if (tiles_per_group == 0 || n_groups == 0 || tile_elems == 0) return MG_OK;
  const int64_t nvec = tile_elems / VecOf<T>::N + 1;
  const int per_group = std::min(512, std::max(1, 4096 / n_groups));
  int blocks = (int)std::min<int64_t>((nvec + 255) / 256, per_group);
  const int64_t group_elems = tiles_per_group * tile_elems;
  if (IsIntegral<T>::value && !d_dark && !d_flat && flat > 0.0 && flat < 1e300 && fabs(dark) < 1e300 &&
      (reinterpret_cast<uintptr_t>(d_tiles) & 15) == 0 && (group_elems * (int64_t)sizeof(T)) % 16 == 0) {
    const int64_t gvec = group_elems / VecOf<T>::N;
    // (every workgroup ends with two compare-and-swap maxima on the group's cache line: few, long-running workgroups)
    const int gb = (int)std::max<int64_t>(1, std::min<int64_t>(gvec / 256, std::min(512, std::max(64, 2048 / n_groups))));
    hipLaunchKernelGGL((k_flatfield_max_int<T>), dim3(gb, n_groups), dim3(256), 0, s, (const T*)d_tiles, group_elems, dark,
                       flat, d_max2);
    MG_CHECK_LAUNCH();
    return MG_OK;
  }
  if (IsIntegral<T>::value && !d_dark && d_flat && flat_dt == MG_F32 && tile_elems % VecOf<T>::N == 0 &&
      (reinterpret_cast<uintptr_t>(d_flat) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_tiles) & 15) == 0 &&
      fabs(dark) < 16777216.0 && (double)(float)dark == dark) {
    constexpr int N = VecOf<T>::N;
    if (d_scratch && scratch_floats >= tile_elems / N && (reinterpret_cast<uintptr_t>(d_scratch) & 3) == 0) {
      const int64_t cvec = tile_elems / N;  // (d_scratch: the bound of this flat image, mg_flatfield_bound)
      // one resident round of workgroups in all (74 VGPRs: 6 per CU), however many groups share them: every workgroup
      // ends with two compare-and-swap maxima on its group's cache line
      const int per = (int)std::max<int64_t>(1, std::min<int64_t>((cvec + 255) / 256, std::max(1, 1536 / n_groups)));
      const int64_t chunks_per_lane = (cvec + (int64_t)per * 256 - 1) / ((int64_t)per * 256) * tiles_per_group;
      if (chunks_per_lane < 512)
        hipLaunchKernelGGL((k_flatfield_max_lean<T, true>), dim3(per, n_groups), dim3(256), 0, s, (const T*)d_tiles,
                           tiles_per_group, tile_elems, dark, (const float*)d_flat, d_scratch, d_max2);
      else
        hipLaunchKernelGGL((k_flatfield_max_lean<T, false>), dim3(per, n_groups), dim3(256), 0, s, (const T*)d_tiles,
                           tiles_per_group, tile_elems, dark, (const float*)d_flat, d_scratch, d_max2);
      MG_CHECK_LAUNCH();
      return MG_OK;
    }
    hipLaunchKernelGGL((k_flatfield_max_fast<T>), dim3(blocks, n_groups), dim3(256), 0, s, (const T*)d_tiles,
                       tiles_per_group, tile_elems, dark, (const float*)d_flat, d_max2);
    MG_CHECK_LAUNCH();
    return MG_OK;
  }
  hipLaunchKernelGGL((k_flatfield_max<T>), dim3(blocks, n_groups), dim3(256), 0, s, (const T*)d_tiles, tiles_per_group,
                     tile_elems, dark, d_dark, dark_dt, flat, d_flat, flat_dt, d_max2);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

template <typename T>
int launch_apply(const void* d_tiles, int64_t n_planes, int n_tr, int n_tc, int ty, int tx, int overlap, int apply,
                 int planes_per_group, double dark, const void* d_dark, int dark_dt, double flat, const void* d_flat, int flat_dt,
                 const double* d_max2, void* d_image, double* d_minmax, hipStream_t s) {
  const int clip = overlap / 2, rem = overlap % 2;
  const int hy = ty - 2 * clip - rem, hx = tx - 2 * clip - rem;
  const int h_out = n_tr * hy, w_out = n_tc * hx;
  if (n_planes == 0 || h_out == 0 || w_out == 0) return MG_OK;
  constexpr int N = VecOf<T>::N;
  if (n_planes > 0x7FFFFFF0) return MG_EINVAL;
  // rows per workgroup: 32 when that still gives ~8 workgroups per CU, down to 2 for a single assay
  int rows = ROWS_PER_BLOCK;
  const int64_t cols_planes = (int64_t)((w_out + 256 * N - 1) / (256 * N)) * ((n_planes + PLANES_PER_BLOCK - 1) / PLANES_PER_BLOCK);
  while (rows > 2 && cols_planes * ((h_out + rows - 1) / rows) < 2048) rows /= 2;
  // ... and at most ~1024 workgroups per plane group walk them (each ends with min/max atomics on the plane's one
  // cache line: 2048 workgroups finishing together took 100 us over them)
  int y_blocks = (h_out + rows - 1) / rows;
  if (rows < ROWS_PER_BLOCK) y_blocks = (int)std::min<int64_t>(y_blocks, std::max<int64_t>(1, 1024 / std::max<int64_t>(1, cols_planes)));
  dim3 grid((w_out + 256 * N - 1) / (256 * N), y_blocks,
            (unsigned)((n_planes + PLANES_PER_BLOCK - 1) / PLANES_PER_BLOCK));
  if (grid.y > 65535 || grid.z > 65535) return MG_EINVAL;
  const bool aligned = IsIntegral<T>::value && hx % N == 0 && tx % N == 0 && clip % N == 0 &&
                       (reinterpret_cast<uintptr_t>(d_tiles) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_image) & 15) == 0 &&
                       (!d_flat || (reinterpret_cast<uintptr_t>(d_flat) & 15) == 0) &&
                       (!d_dark || (reinterpret_cast<uintptr_t>(d_dark) & 15) == 0);
  if (aligned) {
    // an integer-valued scalar dark inside the pixel range: subtracted in the integer domain
    const bool int_dark = apply && !d_dark && dark >= 0.0 && dark <= 65535.0 && dark == (double)(uint32_t)dark;
#define MG_ALIGNED(AP, ID) \
    hipLaunchKernelGGL((k_apply_stitch_aligned<T, AP, ID>), grid, dim3(256), 0, s, (const T*)d_tiles, (int)n_planes, n_tr, \
                       n_tc, ty, tx, clip, hy, hx, planes_per_group, dark, d_dark, dark_dt, flat, d_flat, flat_dt, \
                       d_max2, (T*)d_image, d_minmax, rows)
    if (!apply)
      MG_ALIGNED(false, false);
    else if (int_dark)
      MG_ALIGNED(true, true);
    else
      MG_ALIGNED(true, false);
#undef MG_ALIGNED
    MG_CHECK_LAUNCH();
    return MG_OK;
  }
  if (apply)
    hipLaunchKernelGGL((k_apply_stitch<T, true>), grid, dim3(256), 0, s, (const T*)d_tiles, (int)n_planes, n_tr, n_tc,
                       ty, tx, clip, hy, hx, planes_per_group, dark, d_dark, dark_dt, flat, d_flat, flat_dt, d_max2,
                       (T*)d_image, d_minmax, rows);
  else
    hipLaunchKernelGGL((k_apply_stitch<T, false>), grid, dim3(256), 0, s, (const T*)d_tiles, (int)n_planes, n_tr, n_tc,
                       ty, tx, clip, hy, hx, planes_per_group, dark, d_dark, dark_dt, flat, d_flat, flat_dt, d_max2,
                       (T*)d_image, d_minmax, rows);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

template <typename T>
int launch_minmax(const void* d_src, int n_planes, int64_t plane_stride, int h, int w, int64_t row_stride,
                  double* d_minmax, hipStream_t s) {
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  const int64_t total = (int64_t)h * ((w + VecOf<T>::N - 1) / VecOf<T>::N);
  int bx = (int)std::min<int64_t>((total + 255) / 256, 1024);
  hipLaunchKernelGGL((k_plane_minmax<T>), dim3(bx, n_planes), dim3(256), 0, s, (const T*)d_src, plane_stride, h, w,
                     row_stride, d_minmax);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

bool df_dtype_ok(const void* p, int dt) { return p == nullptr || dt == MG_F32 || dt == MG_F64; }

}  // namespace

extern "C" int mg_version(void) { return 1; }

extern "C" int64_t mg_flatfield_max_scratch_floats(int dtype, int ty, int tx) {
  if (ty <= 0 || tx <= 0) return -1;
  const int n = dtype == MG_U8 ? 16 : dtype == MG_U16 ? 8 : 0;  // (only the integer fast path uses the scratch)
  return n ? ((int64_t)ty * tx + n - 1) / n : 0;
}

extern "C" int mg_flatfield_bound(const void* d_flat, int flat_dtype, int dtype, int ty, int tx, float* d_scratch,
                                  int64_t scratch_floats, void* stream) {
  if (!d_flat || !d_scratch || ty <= 0 || tx <= 0 || flat_dtype != MG_F32) return MG_EINVAL;
  const int64_t need = mg_flatfield_max_scratch_floats(dtype, ty, tx);
  const int64_t tile_elems = (int64_t)ty * tx;
  const int n = dtype == MG_U8 ? 16 : dtype == MG_U16 ? 8 : 0;
  if (need <= 0 || scratch_floats < need || tile_elems % n || (reinterpret_cast<uintptr_t>(d_flat) & 15) ||
      (reinterpret_cast<uintptr_t>(d_scratch) & 3))
    return MG_EINVAL;
  hipStream_t s = mg_stream(stream);
  const int64_t cvec = tile_elems / n;
  const dim3 grid((unsigned)std::min<int64_t>((cvec + 255) / 256, 2048));
  if (n == 16)
    hipLaunchKernelGGL((k_flat_rcmax<16>), grid, dim3(256), 0, s, (const float*)d_flat, cvec, d_scratch);
  else
    hipLaunchKernelGGL((k_flat_rcmax<8>), grid, dim3(256), 0, s, (const float*)d_flat, cvec, d_scratch);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_flatfield_max(const void* d_tiles, int dtype, int64_t n_tiles, int n_groups, int ty, int tx,
                                double dark, const void* d_dark, int dark_dtype, double flat, const void* d_flat,
                                int flat_dtype, double* d_max2, float* d_scratch, int64_t scratch_floats, void* stream) {
  if (!d_tiles || !d_max2 || n_tiles < 0 || ty <= 0 || tx <= 0 || n_groups <= 0 || n_groups > 65535) return MG_EINVAL;
  if (n_tiles % n_groups) return MG_EINVAL;
  if (!df_dtype_ok(d_dark, dark_dtype) || !df_dtype_ok(d_flat, flat_dtype)) return MG_EINVAL;
  const int64_t tile_elems = (int64_t)ty * tx, tpg = n_tiles / n_groups;
  hipStream_t s = mg_stream(stream);
#define MG_MAX(T) \
  return launch_max<T>(d_tiles, tpg, n_groups, tile_elems, dark, d_dark, dark_dtype, flat, d_flat, flat_dtype, d_max2, \
                       d_scratch, scratch_floats, s)
  switch (dtype) {
    case MG_U8: MG_MAX(uint8_t);
    case MG_U16: MG_MAX(uint16_t);
    case MG_F32: MG_MAX(float);
    case MG_F64: MG_MAX(double);
  }
#undef MG_MAX
  return MG_EINVAL;
}

extern "C" int mg_flatfield_is_identity(int dtype, double dark, const void* d_dark, double flat, const void* d_flat) {
  return (dtype == MG_U8 || dtype == MG_U16) && !d_dark && !d_flat && dark == 0.0 && flat == 1.0;
}

extern "C" int mg_flatfield_apply_stitch(const void* d_tiles, int dtype, int64_t n_planes, int n_tile_rows,
                                         int n_tile_cols, int ty, int tx, int overlap, int apply_flatfield,
                                         int planes_per_group, double dark, const void* d_dark, int dark_dtype,
                                         double flat, const void* d_flat, int flat_dtype, const double* d_max2,
                                         void* d_image, double* d_minmax, void* stream) {
  if (!d_tiles || !d_image || n_planes < 0 || n_tile_rows <= 0 || n_tile_cols <= 0 || ty <= 0 || tx <= 0)
    return MG_EINVAL;
  if (overlap < 0 || overlap >= ty || overlap >= tx) return MG_EINVAL;
  // Integer pixels, dark 0 and flat 1 (the reference's defaults, preprocess.py:62): ((t / 1) * M1) / M2 with
  // M2 = M1 / 1 is t itself -- the product of two integers below 2^16 is exact in float64 and so is its quotient by
  // one of them; an all-zero group gives 0 either way (NaN -> 0).  Every pixel would otherwise take the exact
  // two-division path (its fast result is an integer), 2.3x the time of a copy.
  if (apply_flatfield && mg_flatfield_is_identity(dtype, dark, d_dark, flat, d_flat)) apply_flatfield = 0;
  if (apply_flatfield && (!d_max2 || planes_per_group <= 0)) return MG_EINVAL;
  if (!df_dtype_ok(d_dark, dark_dtype) || !df_dtype_ok(d_flat, flat_dtype)) return MG_EINVAL;
  hipStream_t s = mg_stream(stream);
#define MG_APPLY(T) \
  return launch_apply<T>(d_tiles, n_planes, n_tile_rows, n_tile_cols, ty, tx, overlap, apply_flatfield, \
                         planes_per_group > 0 ? planes_per_group : 1, dark, d_dark, dark_dtype, flat, d_flat, \
                         flat_dtype, d_max2, d_image, d_minmax, s)
  switch (dtype) {
    case MG_U8: MG_APPLY(uint8_t);
    case MG_U16: MG_APPLY(uint16_t);
    case MG_F32: MG_APPLY(float);
    case MG_F64: MG_APPLY(double);
  }
#undef MG_APPLY
  return MG_EINVAL;
}

extern "C" int mg_plane_minmax(const void* d_src, int dtype, int n_planes, int64_t plane_stride, int h, int w,
                               int64_t row_stride, double* d_minmax, void* stream) {
  if (!d_src || !d_minmax || n_planes < 0 || h < 0 || w < 0) return MG_EINVAL;
  if (n_planes > 65535) return MG_EINVAL;
  hipStream_t s = mg_stream(stream);
  switch (dtype) {
    case MG_U8: return launch_minmax<uint8_t>(d_src, n_planes, plane_stride, h, w, row_stride, d_minmax, s);
    case MG_U16: return launch_minmax<uint16_t>(d_src, n_planes, plane_stride, h, w, row_stride, d_minmax, s);
    case MG_F32: return launch_minmax<float>(d_src, n_planes, plane_stride, h, w, row_stride, d_minmax, s);
    case MG_F64: return launch_minmax<double>(d_src, n_planes, plane_stride, h, w, row_stride, d_minmax, s);
  }
  return MG_EINVAL;
}
