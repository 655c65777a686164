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

template <typename T>
__global__ __launch_bounds__(256) void k_flatfield_max(const T* __restrict__ tiles, int64_t n_per_group,
                                                        int64_t tile_elems, double dark,
                                                        const void* __restrict__ d_dark, int dark_dt, double flat,
                                                        const void* __restrict__ d_flat, int flat_dt,
                                                        double* __restrict__ out) {
  constexpr int N = VecOf<T>::N;
  const int group = blockIdx.y;
  tiles += (int64_t)group * n_per_group;
  out += 2 * group;
  const int64_t n = n_per_group;
  const bool fast_m2 = (d_flat == nullptr) && (flat > 0.0);  // x / flat is monotone: M2 = M1 / flat
  double m1 = -INFINITY, m2 = -INFINITY;
  const int64_t nvec = n / N;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t v = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; v < nvec; v += stride) {
    T x[N];
    load_vec<T, N>(tiles + v * N, x);
    int64_t p = (d_dark || d_flat) ? (v * N) % tile_elems : 0;
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const double dk = d_dark ? mg_load_f64(d_dark, dark_dt, p) : dark;
      const double fl = d_flat ? mg_load_f64(d_flat, flat_dt, p) : flat;
      double t = (double)x[j] - dk;
      t = t < 0.0 ? 0.0 : t;
      max_step(t, fl, fast_m2, m1, m2);
      if (++p == tile_elems) p = 0;
    }
  }
  for (int64_t i = nvec * N + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    const int64_t p = i % tile_elems;
    const double dk = d_dark ? mg_load_f64(d_dark, dark_dt, p) : dark;
    const double fl = d_flat ? mg_load_f64(d_flat, flat_dt, p) : flat;
    double t = (double)tiles[i] - dk;
    t = t < 0.0 ? 0.0 : t;
    max_step(t, fl, fast_m2, m1, m2);
  }
  if (fast_m2) m2 = (m1 == -INFINITY) ? m1 : m1 / flat;
  block_atomic_max2(m1, m2, out);
}

// ---- pass 2: apply + stitch (+ output min/max) ----------------------------------------
constexpr int ROWS_PER_BLOCK = 8;

// out = trunc(((t / fl) * m1) / m2) for an integer output type without the two float64 divisions:
// v = t * rcp(fl) * (m1 / m2) with a Newton-refined reciprocal agrees with the reference's three
// roundings to ~1e-15 relative, so the truncation is the same unless v lies within 1e-6 of an
// integer -- those (rare) pixels take the exact path.
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

template <typename T>
__device__ __forceinline__ T correct_pixel(double t, double fl, double m1, double m2, double k, bool fast_ok) {
  if (IsIntegral<T>::value && fast_ok && flat_in_range(fl)) {
    if (t == 0.0) return (T)0;  // 0 / fl * m1 / m2 == 0 exactly (m1, m2 finite and positive here)
    double r = (double)__builtin_amdgcn_rcpf((float)fl);
    r = r * (2.0 - fl * r);
    r = r * (2.0 - fl * r);
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

template <typename T, bool APPLY>
__global__ __launch_bounds__(256) void k_apply_stitch(const T* __restrict__ tiles, int n_tr, int n_tc, int ty, int tx,
                                                       int clip, int hy, int hx, int planes_per_group, double dark,
                                                       const void* __restrict__ d_dark, int dark_dt, double flat,
                                                       const void* __restrict__ d_flat, int flat_dt,
                                                       const double* __restrict__ d_max2, T* __restrict__ image,
                                                       double* __restrict__ d_minmax) {
  constexpr int N = VecOf<T>::N;
  const int plane = blockIdx.z;
  const int h_out = n_tr * hy, w_out = n_tc * hx;
  const int ox0 = (blockIdx.x * blockDim.x + threadIdx.x) * N;
  double m1 = 0.0, m2 = 1.0, kk = 1.0;
  bool fast_ok = false;
  if (APPLY) {
    const int group = plane / planes_per_group;
    m1 = d_max2[2 * group];
    m2 = d_max2[2 * group + 1];
    kk = m1 / m2;
    fast_ok = kk > 0.0 && kk < 1e30 && m1 > 0.0 && m1 < 1e300 && m2 > 0.0 && m2 < 1e300;
  }
  double vmin = INFINITY, vmax = -INFINITY;
  const int64_t tile_elems = (int64_t)ty * tx;
  const int row_end = min((int)(blockIdx.y + 1) * ROWS_PER_BLOCK, h_out);
  if (ox0 < w_out) {
    const int tc0 = ox0 / hx;
    const int x0 = ox0 - tc0 * hx + clip;
    const bool one_tile = (ox0 + N <= w_out) && (x0 - clip + N <= hx);
    for (int oy = blockIdx.y * ROWS_PER_BLOCK; oy < row_end; ++oy) {
      const int tr = oy / hy;
      const int y = oy - tr * hy + clip;
      T* dst = image + ((int64_t)plane * h_out + oy) * w_out + ox0;
      T x[N], o[N];
      int64_t pix[N];
      int cnt = N;
      if (one_tile) {
        const int64_t tile_base = (((int64_t)plane * n_tr + tr) * n_tc + tc0) * tile_elems;
        const int64_t p0 = (int64_t)y * tx + x0;
        load_vec<T, N>(tiles + tile_base + p0, x);
#pragma unroll
        for (int j = 0; j < N; ++j) pix[j] = p0 + j;
      } else {
        cnt = min(N, w_out - ox0);
        for (int j = 0; j < cnt; ++j) {
          const int ox = ox0 + j;
          const int tc = ox / hx;
          const int xx = ox - tc * hx + clip;
          const int64_t tile_base = (((int64_t)plane * n_tr + tr) * n_tc + tc) * tile_elems;
          pix[j] = (int64_t)y * tx + xx;
          x[j] = tiles[tile_base + pix[j]];
        }
      }
#pragma unroll
      for (int j = 0; j < N; ++j) {
        if (j < cnt) {
          if (APPLY) {
            const double dk = d_dark ? mg_load_f64(d_dark, dark_dt, pix[j]) : dark;
            const double fl = d_flat ? mg_load_f64(d_flat, flat_dt, pix[j]) : flat;
            double t = (double)x[j] - dk;
            t = t < 0.0 ? 0.0 : t;
            o[j] = correct_pixel<T>(t, fl, m1, m2, kk, fast_ok);
          } else {
            o[j] = x[j];
          }
          if (d_minmax) {
            const double ov = (double)o[j];
            vmin = mg_nanmin(vmin, ov);
            vmax = mg_nanmax(vmax, ov);
          }
        }
      }
      if (cnt == N) {
        store_vec<T, N>(dst, o);
      } else {
        for (int j = 0; j < cnt; ++j) dst[j] = o[j];
      }
    }
  }
  if (d_minmax) {
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
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
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
int launch_max(const void* d_tiles, int64_t n_per_group, int n_groups, int64_t tile_elems, double dark,
               const void* d_dark, int dark_dt, double flat, const void* d_flat, int flat_dt, double* d_max2,
               hipStream_t s) {
  if (n_per_group == 0 || n_groups == 0) return MG_OK;
  const int64_t nvec = n_per_group / VecOf<T>::N + 1;
  const int per_group = std::max(1, 2048 / n_groups);
  int blocks = (int)std::min<int64_t>((nvec + 255) / 256, per_group);
  hipLaunchKernelGGL((k_flatfield_max<T>), dim3(blocks, n_groups), dim3(256), 0, s, (const T*)d_tiles, n_per_group,
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
  dim3 grid((w_out + 256 * N - 1) / (256 * N), (h_out + ROWS_PER_BLOCK - 1) / ROWS_PER_BLOCK, (unsigned)n_planes);
  if (grid.y > 65535 || grid.z > 65535) return MG_EINVAL;
  if (apply)
    hipLaunchKernelGGL((k_apply_stitch<T, true>), grid, dim3(256), 0, s, (const T*)d_tiles, n_tr, n_tc, ty, tx, clip,
                       hy, hx, planes_per_group, dark, d_dark, dark_dt, flat, d_flat, flat_dt, d_max2, (T*)d_image,
                       d_minmax);
  else
    hipLaunchKernelGGL((k_apply_stitch<T, false>), grid, dim3(256), 0, s, (const T*)d_tiles, n_tr, n_tc, ty, tx, clip,
                       hy, hx, planes_per_group, dark, d_dark, dark_dt, flat, d_flat, flat_dt, d_max2, (T*)d_image,
                       d_minmax);
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

extern "C" int mg_flatfield_max(const void* d_tiles, int dtype, int64_t n_tiles, int n_groups, int ty, int tx,
                                double dark, const void* d_dark, int dark_dtype, double flat, const void* d_flat,
                                int flat_dtype, double* d_max2, void* stream) {
  if (!d_tiles || !d_max2 || n_tiles < 0 || ty <= 0 || tx <= 0 || n_groups <= 0 || n_groups > 65535) return MG_EINVAL;
  if (n_tiles % n_groups) return MG_EINVAL;
  if (!df_dtype_ok(d_dark, dark_dtype) || !df_dtype_ok(d_flat, flat_dtype)) return MG_EINVAL;
  const int64_t tile_elems = (int64_t)ty * tx, n = (n_tiles / n_groups) * tile_elems;
  hipStream_t s = mg_stream(stream);
#define MG_MAX(T) \
  return launch_max<T>(d_tiles, n, n_groups, tile_elems, dark, d_dark, dark_dtype, flat, d_flat, flat_dtype, d_max2, s)
  switch (dtype) {
    case MG_U8: MG_MAX(uint8_t);
    case MG_U16: MG_MAX(uint16_t);
    case MG_F32: MG_MAX(float);
    case MG_F64: MG_MAX(double);
  }
#undef MG_MAX
  return MG_EINVAL;
}

extern "C" int mg_flatfield_apply_stitch(const void* d_tiles, int dtype, int64_t n_planes, int n_tile_rows,
                                         int n_tile_cols, int ty, int tx, int overlap, int apply_flatfield,
                                         int planes_per_group, double dark, const void* d_dark, int dark_dtype,
                                         double flat, const void* d_flat, int flat_dtype, const double* d_max2,
                                         void* d_image, double* d_minmax, void* stream) {
  if (!d_tiles || !d_image || n_planes < 0 || n_tile_rows <= 0 || n_tile_cols <= 0 || ty <= 0 || tx <= 0)
    return MG_EINVAL;
  if (overlap < 0 || overlap >= ty || overlap >= tx) return MG_EINVAL;
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
