// A3-A7: edge stage of find_circles (utils.py:20-27, 115-142) and grid_array (utils.py:347-377).
//
//   to_uint8 + GaussianBlur 5x5  ->  Scharr + |grad|^2 histogram (x2: coarse, fine)  ->
//   Canny NMS + double threshold ->  hysteresis sweeps -> finalise (edge map, angle map,
//   cell counts) -> CSR edge grid.
//
// All stencils stage a tile (+halo, BORDER_REFLECT_101) through LDS; every kernel is batched
// over planes.  Roofline: HBM (bytes per pixel in DESIGN.md); integer arithmetic throughout,
// so results are bit-exact against the oracle.
#include <math.h>

#include "mg_common.h"

namespace {

constexpr int TW = 128;  // tile width  (pixels)
constexpr int TH = 32;   // tile height (pixels)
constexpr int NT = 256;  // threads per block
constexpr int ROWS_PER_THREAD = TH / (NT / TW);  // 16

// ---- to_uint8 -------------------------------------------------------------------------
struct U8Scale {
  double mn, top, rcp;
  int passthrough;
};

__device__ __forceinline__ U8Scale make_scale(const double* d_minmax, int plane) {
  U8Scale s;
  s.passthrough = (d_minmax == nullptr);
  s.mn = s.passthrough ? 0.0 : d_minmax[2 * plane];
  const double mx = s.passthrough ? 0.0 : d_minmax[2 * plane + 1];
  s.top = mx - s.mn;  // == max(arr - min) because x -> fl(x - mn) is monotone
  s.rcp = s.top > 0.0 ? 1.0 / s.top : 0.0;
  return s;
}

// utils.py:23-27.  Integer inputs: floor(255 * (x - mn) / top) computed exactly: the true
// quotient is either an integer or at least 1/top >= 2^-16 away from one, so the float64
// product with the rounded reciprocal (+2^-20) floors to the same integer as the reference's
// correctly rounded float64 division.
template <typename T>
__device__ __forceinline__ uint8_t to_u8(T x, const U8Scale& s) {
  if (s.passthrough) return (uint8_t)x;
  const double a = (double)x - s.mn;
  if (!(s.top > 0.0)) return (uint8_t)(int)a;
  return (uint8_t)(unsigned int)(255.0 * a * s.rcp + 0x1p-20);
}
template <>
__device__ __forceinline__ uint8_t to_u8<float>(float x, const U8Scale& s) {
  const double a = (double)x - s.mn;
  if (!(s.top > 0.0)) return (uint8_t)(int)a;
  return (uint8_t)(int)(255.0 * a / s.top);
}
template <>
__device__ __forceinline__ uint8_t to_u8<double>(double x, const U8Scale& s) {
  const double a = x - s.mn;
  if (!(s.top > 0.0)) return (uint8_t)(int)a;
  return (uint8_t)(int)(255.0 * a / s.top);
}

// ---- K1: to_uint8 + 5x5 Gaussian ([1 4 6 4 1] x [1 4 6 4 1], (sum + 128) >> 8) ----------
template <typename T>
__global__ __launch_bounds__(NT) void k_u8_blur(const T* __restrict__ src, int64_t plane_stride, int h, int w,
                                                int64_t row_stride, const double* __restrict__ d_minmax,
                                                uint8_t* __restrict__ d_blur, uint8_t* __restrict__ d_u8) {
  constexpr int LW = TW + 4, LH = TH + 4, LS = TW + 8;
  __shared__ uint8_t tile[LH][LS];
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  const T* base = src + (int64_t)plane * plane_stride;
  const U8Scale sc = make_scale(d_minmax, plane);
  for (int i = threadIdx.x; i < LH * LW; i += NT) {
    const int r = i / LW, c = i - r * LW;
    const int gy = mg_reflect101(ty0 + r - 2, h), gx = mg_reflect101(tx0 + c - 2, w);
    const uint8_t v = to_u8<T>(base[(int64_t)gy * row_stride + gx], sc);
    tile[r][c] = v;
    if (d_u8 && r >= 2 && r < TH + 2 && c >= 2 && c < TW + 2 && ty0 + r - 2 < h && tx0 + c - 2 < w)
      d_u8[((int64_t)plane * h + (ty0 + r - 2)) * w + (tx0 + c - 2)] = v;
  }
  __syncthreads();
  const int c = threadIdx.x % TW;             // output column inside the tile
  const int r0 = (threadIdx.x / TW) * ROWS_PER_THREAD;  // first output row of this thread
  if (tx0 + c >= w) return;
  int hsum[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const uint8_t* p = &tile[r0 + k][c];
    hsum[k + 1] = p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4];
  }
  uint8_t* out = d_blur + ((int64_t)plane * h) * w + tx0 + c;
  for (int rr = 0; rr < ROWS_PER_THREAD; ++rr) {
#pragma unroll
    for (int k = 0; k < 4; ++k) hsum[k] = hsum[k + 1];
    const uint8_t* p = &tile[r0 + rr + 4][c];
    hsum[4] = p[0] + 4 * p[1] + 6 * p[2] + 4 * p[3] + p[4];
    const int gy = ty0 + r0 + rr;
    if (gy < h) {
      const int v = hsum[0] + 4 * hsum[1] + 6 * hsum[2] + 4 * hsum[3] + hsum[4];
      out[(int64_t)gy * w] = (uint8_t)((v + 128) >> 8);
    }
  }
}

// ---- shared loader: blurred tile with halo HALO (reflect-101) into LDS -------------------
template <int HALO, int LS>
__device__ __forceinline__ void load_blur_tile(const uint8_t* __restrict__ plane_ptr, int h, int w, int tx0, int ty0,
                                               uint8_t (*tile)[LS]) {
  constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
  for (int i = threadIdx.x; i < LH * LW; i += NT) {
    const int r = i / LW, c = i - r * LW;
    const int gy = mg_reflect101(ty0 + r - HALO, h), gx = mg_reflect101(tx0 + c - HALO, w);
    tile[r][c] = plane_ptr[(int64_t)gy * w + gx];
  }
}

// Scharr at LDS position (r, c): dx = right - left, dy = below - above, weights 3/10/3.
template <int LS>
__device__ __forceinline__ void scharr_at(const uint8_t (*t)[LS], int r, int c, int& dx, int& dy) {
  const int a = t[r - 1][c - 1], b = t[r - 1][c], cc = t[r - 1][c + 1];
  const int d = t[r][c - 1], f = t[r][c + 1];
  const int g = t[r + 1][c - 1], hh = t[r + 1][c], i = t[r + 1][c + 1];
  dx = 3 * (cc - a) + 10 * (f - d) + 3 * (i - g);
  dy = 3 * (g - a) + 10 * (hh - b) + 3 * (i - cc);
}

// ---- K2: histogram of m = dx^2 + dy^2 ------------------------------------------------------
__global__ __launch_bounds__(NT) void k_scharr_hist(const uint8_t* __restrict__ d_blur, int h, int w,
                                                    const uint32_t* __restrict__ d_base, int shift, int n_bins,
                                                    uint32_t* __restrict__ d_hist) {
  constexpr int LS = TW + 4;
  __shared__ uint8_t tile[TH + 2][LS];
  extern __shared__ uint32_t hist[];
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  for (int i = threadIdx.x; i < n_bins; i += NT) hist[i] = 0;
  load_blur_tile<1, LS>(d_blur + (int64_t)plane * h * w, h, w, tx0, ty0, tile);
  __syncthreads();
  const uint32_t base = d_base ? d_base[plane] : 0u;
  const int c = threadIdx.x % TW;
  const int r0 = (threadIdx.x / TW) * ROWS_PER_THREAD;
  constexpr uint32_t NONE = 0xFFFFFFFFu;
  for (int rr = 0; rr < ROWS_PER_THREAD; ++rr) {
    uint32_t bin = NONE;
    if (tx0 + c < w && ty0 + r0 + rr < h) {
      int dx, dy;
      scharr_at<LS>(tile, r0 + rr + 1, c + 1, dx, dy);
      const uint32_t m = (uint32_t)(dx * dx + dy * dy);
      if (m >= base) {
        const uint32_t b = (m - base) >> shift;
        if (b < (uint32_t)n_bins) bin = b;
      }
    }
    // Flat regions put a whole wave into one bin: add 64 once instead of 64 colliding atomics.
    const uint32_t first = (uint32_t)__builtin_amdgcn_readfirstlane((int)bin);
    if (__all(bin == first)) {
      if ((threadIdx.x & 63) == 0 && first != NONE) atomicAdd(&hist[first], 64u);
    } else if (bin != NONE) {
      atomicAdd(&hist[bin], 1u);
    }
  }
  __syncthreads();
  uint32_t* out = d_hist + (int64_t)plane * n_bins;
  for (int i = threadIdx.x; i < n_bins; i += NT) {
    const uint32_t v = hist[i];
    if (v) atomicAdd(&out[i], v);
  }
}

// ---- K3: Canny non-maximum suppression + double threshold ------------------------------------
__global__ __launch_bounds__(NT) void k_canny_nms(const uint8_t* __restrict__ d_blur, int h, int w,
                                                  const int32_t* __restrict__ d_thresh, uint8_t* __restrict__ d_map) {
  constexpr int LS = TW + 4;
  constexpr int MW = TW + 2, MH = TH + 2, MS = TW + 3;
  __shared__ uint8_t tile[TH + 4][LS];
  __shared__ int mag[MH][MS];
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  load_blur_tile<2, LS>(d_blur + (int64_t)plane * h * w, h, w, tx0, ty0, tile);
  __syncthreads();
  // Magnitudes on the tile + 1 halo; zero outside the image (OpenCV's zeroed border rows/cols).
  for (int i = threadIdx.x; i < MH * MW; i += NT) {
    const int r = i / MW, c = i - r * MW;
    const int gy = ty0 + r - 1, gx = tx0 + c - 1;
    int m = 0;
    if (gy >= 0 && gy < h && gx >= 0 && gx < w) {
      int dx, dy;
      scharr_at<LS>(tile, r + 1, c + 1, dx, dy);
      m = dx * dx + dy * dy;
    }
    mag[r][c] = m;
  }
  __syncthreads();
  const int low = d_thresh[2 * plane], high = d_thresh[2 * plane + 1];
  const int c = threadIdx.x % TW;
  const int r0 = (threadIdx.x / TW) * ROWS_PER_THREAD;
  if (tx0 + c >= w) return;
  uint8_t* out = d_map + ((int64_t)plane * h) * w + tx0 + c;
  constexpr int TG22 = 13573;
  for (int rr = 0; rr < ROWS_PER_THREAD; ++rr) {
    const int gy = ty0 + r0 + rr;
    if (gy >= h) break;
    const int mr = r0 + rr + 1, mc = c + 1;
    const int m = mag[mr][mc];
    uint8_t v = 1;
    if (m > low) {
      int xs, ys;
      scharr_at<LS>(tile, mr + 1, mc + 1, xs, ys);
      const int x = abs(xs);
      const int y = abs(ys) << 15;
      const int tg22x = x * TG22;
      bool is_max;
      if (y < tg22x) {
        is_max = m > mag[mr][mc - 1] && m >= mag[mr][mc + 1];
      } else {
        const int tg67x = tg22x + (x << 16);
        if (y > tg67x) {
          is_max = m > mag[mr - 1][mc] && m >= mag[mr + 1][mc];
        } else {
          const int s = (xs ^ ys) < 0 ? -1 : 1;
          is_max = m > mag[mr - 1][mc - s] && m > mag[mr + 1][mc + s];
        }
      }
      if (is_max) v = m > high ? 2 : 0;
    }
    out[(int64_t)gy * w] = v;
  }
}

// ---- K4: hysteresis sweep ------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_hysteresis(uint8_t* __restrict__ d_map, int h, int w,
                                                   uint32_t* __restrict__ d_changed,
                                                   const uint8_t* __restrict__ d_flags_in,
                                                   uint8_t* __restrict__ d_flags_out) {
  constexpr int LS = TW + 4;
  __shared__ uint8_t t[TH + 2][LS];
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  const int ntx = gridDim.x, nty = gridDim.y;
  if (d_flags_in) {
    // A tile can only gain edges if it or one of its 8 neighbours changed in the previous sweep.
    const uint8_t* f = d_flags_in + (int64_t)plane * ntx * nty;
    bool active = false;
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int bx = (int)blockIdx.x + dx, by = (int)blockIdx.y + dy;
        if (bx >= 0 && bx < ntx && by >= 0 && by < nty) active |= f[by * ntx + bx] != 0;
      }
    if (!active) return;
  }
  uint8_t* pm = d_map + (int64_t)plane * h * w;
  int has_weak = 0, has_strong = 0;
  for (int i = threadIdx.x; i < (TH + 2) * (TW + 2); i += NT) {
    const int r = i / (TW + 2), c = i - r * (TW + 2);
    const int gy = ty0 + r - 1, gx = tx0 + c - 1;
    uint8_t v = 1;
    if (gy >= 0 && gy < h && gx >= 0 && gx < w) v = pm[(int64_t)gy * w + gx];
    t[r][c] = v;
    has_weak |= (v == 0);
    has_strong |= (v == 2);
  }
  const int any_weak = __syncthreads_or(has_weak);
  const int any_strong = __syncthreads_or(has_strong);
  if (!any_weak || !any_strong) return;
  const int c = threadIdx.x % TW + 1;
  const int r0 = (threadIdx.x / TW) * ROWS_PER_THREAD + 1;
  uint32_t mine = 0;  // bit rr set: pixel promoted by this thread
  int again;
  do {
    int changed = 0;
    for (int rr = 0; rr < ROWS_PER_THREAD; ++rr) {
      const int r = r0 + rr;
      if (t[r][c] == 0) {
        const bool nb = t[r - 1][c - 1] == 2 || t[r - 1][c] == 2 || t[r - 1][c + 1] == 2 || t[r][c - 1] == 2 ||
                        t[r][c + 1] == 2 || t[r + 1][c - 1] == 2 || t[r + 1][c] == 2 || t[r + 1][c + 1] == 2;
        if (nb) {
          t[r][c] = 2;
          mine |= 1u << rr;
          changed = 1;
        }
      }
    }
    again = __syncthreads_or(changed);
  } while (again);
  for (int rr = 0; rr < ROWS_PER_THREAD; ++rr)
    if (mine & (1u << rr)) pm[(int64_t)(ty0 + r0 - 1 + rr) * w + (tx0 + c - 1)] = 2;
  if (__syncthreads_or(mine != 0) && threadIdx.x == 0) {
    atomicAdd(&d_changed[plane], 1u);
    if (d_flags_out) d_flags_out[(int64_t)plane * ntx * nty + blockIdx.y * ntx + blockIdx.x] = 1;
  }
}

// ---- K5: finalise: edge bitmap (+ optional {0,1} byte map and angle map) ----------------------
// One thread owns 16 consecutive pixels (one 16-byte load of the Canny map); a pair of lanes
// forms one 32-bit word of the bitmap (bit i of word k <-> linear pixel 32 k + i).
__global__ __launch_bounds__(NT) void k_edges_finalize(uint8_t* __restrict__ d_map, const uint8_t* __restrict__ d_blur,
                                                       int h, int w, int64_t words_per_plane,
                                                       uint32_t* __restrict__ d_bits, int write_bytes,
                                                       float* __restrict__ d_angle) {
  const int plane = blockIdx.y;
  const int64_t npix = (int64_t)h * w;
  uint8_t* pm = d_map + plane * npix;
  uint32_t* bits = d_bits + plane * words_per_plane;
  float* pa = d_angle ? d_angle + plane * npix : nullptr;
  const int64_t n_chunks = (npix + 15) / 16;
  const int64_t n_iter = (n_chunks + (int64_t)gridDim.x * NT - 1) / ((int64_t)gridDim.x * NT);
  for (int64_t it = 0; it < n_iter; ++it) {
    const int64_t chunk = (it * gridDim.x + blockIdx.x) * NT + threadIdx.x;  // all lanes stay in the loop
    const int64_t i0 = chunk * 16;
    uint32_t m16 = 0;
    if (i0 < npix) {
      uint8_t v[16];
      if (i0 + 16 <= npix && ((reinterpret_cast<uintptr_t>(pm + i0) & 15) == 0)) {
        const uint4 raw = *reinterpret_cast<const uint4*>(pm + i0);
        __builtin_memcpy(v, &raw, 16);
      } else {
        for (int j = 0; j < 16; ++j) v[j] = (i0 + j < npix) ? pm[i0 + j] : 1;
      }
#pragma unroll
      for (int j = 0; j < 16; ++j) m16 |= (uint32_t)(v[j] == 2) << j;
      if (write_bytes) {
        uint8_t o[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) o[j] = (m16 >> j) & 1;
        if (i0 + 16 <= npix && ((reinterpret_cast<uintptr_t>(pm + i0) & 15) == 0)) {
          uint4 raw;
          __builtin_memcpy(&raw, o, 16);
          *reinterpret_cast<uint4*>(pm + i0) = raw;
        } else {
          for (int j = 0; j < 16 && i0 + j < npix; ++j) pm[i0 + j] = o[j];
        }
      }
      if (pa) {  // inspection only: mark non-edge pixels (edge pixels are written by mg_edge_angles)
        for (int j = 0; j < 16 && i0 + j < npix; ++j)
          if (!((m16 >> j) & 1)) pa[i0 + j] = MG_NO_EDGE;
      }
    }
    const uint32_t hi = (uint32_t)__shfl_down((int)m16, 1);
    if ((threadIdx.x & 1) == 0 && i0 < npix) bits[chunk >> 1] = m16 | (hi << 16);
  }
}

// ---- K6: grid_array from the bitmap: per-cell counts, scan, ordered coordinate fill -------------
__device__ __forceinline__ uint32_t row_bits(const uint32_t* __restrict__ bits, int64_t bit0, int n) {
  // n (<= 32) consecutive bits starting at linear bit index bit0
  const int64_t wi = bit0 >> 5;
  const int sh = (int)(bit0 & 31);
  uint64_t two = bits[wi];
  if (sh + n > 32) two |= (uint64_t)bits[wi + 1] << 32;
  const uint32_t v = (uint32_t)(two >> sh);
  return n >= 32 ? v : (v & ((1u << n) - 1u));
}

__global__ __launch_bounds__(NT) void k_cell_count(const uint32_t* __restrict__ d_bits, int64_t words_per_plane, int h,
                                                   int w, int grid, int gc, int n_cells,
                                                   int32_t* __restrict__ d_counts) {
  const int plane = blockIdx.y;
  const int cell = blockIdx.x * NT + threadIdx.x;
  if (cell >= n_cells) return;
  const uint32_t* bits = d_bits + plane * words_per_plane;
  const int cr = cell / gc, cc = cell - cr * gc;
  const int y0 = cr * grid, x0 = cc * grid;
  const int ch = min(grid, h - y0), cw = min(grid, w - x0);
  int cnt = 0;
  for (int r = 0; r < ch; ++r)
    for (int c0 = 0; c0 < cw; c0 += 32)
      cnt += __popc(row_bits(bits, (int64_t)(y0 + r) * w + x0 + c0, min(32, cw - c0)));
  d_counts[(int64_t)plane * n_cells + cell] = cnt;
}

__global__ __launch_bounds__(1024) void k_cell_scan(const int32_t* __restrict__ d_counts, int n_cells,
                                                    int32_t* __restrict__ d_starts, int32_t* __restrict__ d_num_edges) {
  const int plane = blockIdx.x;
  const int32_t* cnt = d_counts + (int64_t)plane * n_cells;
  int32_t* st = d_starts + (int64_t)plane * n_cells;
  int carry = 0;
  for (int base = 0; base < n_cells; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_cells ? cnt[i] : 0;
    int total;
    const int ex = mg_block_exscan(v, &total);
    if (i < n_cells) st[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) d_num_edges[plane] = carry;
}

__global__ __launch_bounds__(NT) void k_cell_fill(const uint32_t* __restrict__ d_bits, int64_t words_per_plane, int h,
                                                  int w, int grid, int gc, int n_cells,
                                                  const int32_t* __restrict__ d_starts, int32_t* __restrict__ d_coords,
                                                  int64_t coord_cap) {
  const int plane = blockIdx.y;
  const int cell = blockIdx.x * NT + threadIdx.x;
  if (cell >= n_cells) return;
  const uint32_t* bits = d_bits + plane * words_per_plane;
  const int cr = cell / gc, cc = cell - cr * gc;
  const int y0 = cr * grid, x0 = cc * grid;
  const int ch = min(grid, h - y0), cw = min(grid, w - x0);
  int2* out = reinterpret_cast<int2*>(d_coords + (int64_t)plane * coord_cap * 2);
  int64_t pos = d_starts[(int64_t)plane * n_cells + cell];
  for (int r = 0; r < ch; ++r)
    for (int c0 = 0; c0 < cw; c0 += 32) {
      uint32_t v = row_bits(bits, (int64_t)(y0 + r) * w + x0 + c0, min(32, cw - c0));
      while (v) {
        const int b = __ffs(v) - 1;
        v &= v - 1;
        if (pos < coord_cap) out[pos] = make_int2(y0 + r, x0 + c0 + b);
        ++pos;
      }
    }
}

// ---- K6b: gradient angle at every edge pixel (thread per entry of the compact edge list) --------
__global__ __launch_bounds__(NT) void k_edge_angles(const uint8_t* __restrict__ d_blur, int h, int w,
                                                    const int32_t* __restrict__ d_coords, int64_t coord_cap,
                                                    const int32_t* __restrict__ d_num_edges,
                                                    float* __restrict__ d_angle) {
  const int plane = blockIdx.y;
  const int n = min((int64_t)d_num_edges[plane], coord_cap);
  const uint8_t* pb = d_blur + (int64_t)plane * h * w;
  const int2* co = reinterpret_cast<const int2*>(d_coords + (int64_t)plane * coord_cap * 2);
  float* pa = d_angle + (int64_t)plane * h * w;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int2 yx = co[i];
    pa[(int64_t)yx.x * w + yx.y] = mg_edge_angle(pb, h, w, yx.x, yx.y);
  }
}

inline dim3 tile_grid(int h, int w, int n_planes) { return dim3((w + TW - 1) / TW, (h + TH - 1) / TH, n_planes); }

}  // namespace

extern "C" int mg_to_uint8_blur(const void* d_src, int dtype, int n_planes, int64_t plane_stride, int h, int w,
                                int64_t row_stride, const double* d_minmax, uint8_t* d_blur, uint8_t* d_u8,
                                void* stream) {
  if (!d_src || !d_blur || n_planes < 0 || h < 0 || w < 0) return MG_EINVAL;
  if (!d_minmax && dtype != MG_U8) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  const dim3 g = tile_grid(h, w, n_planes);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  hipStream_t s = mg_stream(stream);
  switch (dtype) {
    case MG_U8:
      hipLaunchKernelGGL((k_u8_blur<uint8_t>), g, dim3(NT), 0, s, (const uint8_t*)d_src, plane_stride, h, w,
                         row_stride, d_minmax, d_blur, d_u8);
      break;
    case MG_U16:
      hipLaunchKernelGGL((k_u8_blur<uint16_t>), g, dim3(NT), 0, s, (const uint16_t*)d_src, plane_stride, h, w,
                         row_stride, d_minmax, d_blur, d_u8);
      break;
    case MG_F32:
      hipLaunchKernelGGL((k_u8_blur<float>), g, dim3(NT), 0, s, (const float*)d_src, plane_stride, h, w, row_stride,
                         d_minmax, d_blur, d_u8);
      break;
    case MG_F64:
      hipLaunchKernelGGL((k_u8_blur<double>), g, dim3(NT), 0, s, (const double*)d_src, plane_stride, h, w, row_stride,
                         d_minmax, d_blur, d_u8);
      break;
    default:
      return MG_EINVAL;
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_scharr_hist(const uint8_t* d_blur, int n_planes, int h, int w, const uint32_t* d_base, int shift,
                              int n_bins, uint32_t* d_hist, void* stream) {
  if (!d_blur || !d_hist || n_planes < 0 || h < 0 || w < 0 || shift < 0 || shift > 31 || n_bins <= 0 || n_bins > 8192)
    return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  const dim3 g = tile_grid(h, w, n_planes);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  hipLaunchKernelGGL(k_scharr_hist, g, dim3(NT), n_bins * sizeof(uint32_t), mg_stream(stream), d_blur, h, w, d_base,
                     shift, n_bins, d_hist);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_canny_nms(const uint8_t* d_blur, int n_planes, int h, int w, const int32_t* d_thresh, uint8_t* d_map,
                            void* stream) {
  if (!d_blur || !d_thresh || !d_map || n_planes < 0 || h < 0 || w < 0) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  const dim3 g = tile_grid(h, w, n_planes);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  hipLaunchKernelGGL(k_canny_nms, g, dim3(NT), 0, mg_stream(stream), d_blur, h, w, d_thresh, d_map);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_canny_hysteresis(uint8_t* d_map, int n_planes, int h, int w, uint32_t* d_changed,
                                   const uint8_t* d_flags_in, uint8_t* d_flags_out, void* stream) {
  if (!d_map || !d_changed || n_planes < 0 || h < 0 || w < 0) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  const dim3 g = tile_grid(h, w, n_planes);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  hipLaunchKernelGGL(k_hysteresis, g, dim3(NT), 0, mg_stream(stream), d_map, h, w, d_changed, d_flags_in, d_flags_out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_hysteresis_tiles(int h, int w, int* tiles_x, int* tiles_y) {
  if (!tiles_x || !tiles_y) return MG_EINVAL;
  *tiles_x = (w + TW - 1) / TW;
  *tiles_y = (h + TH - 1) / TH;
  return MG_OK;
}

extern "C" int mg_edges_finalize(uint8_t* d_map, const uint8_t* d_blur, int n_planes, int h, int w,
                                 uint32_t* d_edge_bits, int64_t words_per_plane, int write_bytes, float* d_angle,
                                 void* stream) {
  if (!d_map || !d_blur || !d_edge_bits || n_planes < 0 || h < 0 || w < 0 || n_planes > 65535) return MG_EINVAL;
  const int64_t npix = (int64_t)h * w;
  if (words_per_plane * 32 < ((npix + 31) / 32) * 32 || (words_per_plane & 1)) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  const int64_t n_chunks = (npix + 15) / 16;
  const int bx = (int)std::min<int64_t>((n_chunks + NT - 1) / NT, 2048);
  hipLaunchKernelGGL(k_edges_finalize, dim3(bx, n_planes), dim3(NT), 0, mg_stream(stream), d_map, d_blur, h, w,
                     words_per_plane, d_edge_bits, write_bytes, d_angle);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_edge_grid(const uint32_t* d_edge_bits, int64_t words_per_plane, int n_planes, int h, int w, int grid,
                            int32_t* d_cell_counts, int32_t* d_cell_starts, int32_t* d_num_edges, int32_t* d_coords,
                            int64_t coord_cap, void* stream) {
  if (!d_edge_bits || !d_cell_counts || !d_cell_starts || !d_num_edges || n_planes < 0 || grid <= 0 || coord_cap < 0 ||
      n_planes > 65535)
    return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  const int gr = (h + grid - 1) / grid, gc = (w + grid - 1) / grid, n_cells = gr * gc;
  hipStream_t s = mg_stream(stream);
  if (!d_coords) {  // phase 1: counts + scan (the caller sizes the coordinate list from d_num_edges)
    if (n_cells > 0) {
      hipLaunchKernelGGL(k_cell_count, dim3((n_cells + NT - 1) / NT, n_planes), dim3(NT), 0, s, d_edge_bits,
                         words_per_plane, h, w, grid, gc, n_cells, d_cell_counts);
      MG_CHECK_LAUNCH();
    }
    hipLaunchKernelGGL(k_cell_scan, dim3(n_planes), dim3(1024), 0, s, d_cell_counts, n_cells, d_cell_starts, d_num_edges);
    MG_CHECK_LAUNCH();
    return MG_OK;
  }
  if (n_cells > 0) {  // phase 2: ordered fill
    hipLaunchKernelGGL(k_cell_fill, dim3((n_cells + NT - 1) / NT, n_planes), dim3(NT), 0, s, d_edge_bits,
                       words_per_plane, h, w, grid, gc, n_cells, d_cell_starts, d_coords, coord_cap);
    MG_CHECK_LAUNCH();
  }
  return MG_OK;
}

extern "C" int mg_edge_angles(const uint8_t* d_blur, int n_planes, int h, int w, const int32_t* d_coords,
                              int64_t coord_cap, const int32_t* d_num_edges, float* d_angle, void* stream) {
  if (!d_blur || !d_coords || !d_num_edges || !d_angle || n_planes < 0 || n_planes > 65535 || coord_cap < 0)
    return MG_EINVAL;
  if (n_planes == 0 || coord_cap == 0) return MG_OK;
  const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((coord_cap + NT - 1) / NT, 4096));
  hipLaunchKernelGGL(k_edge_angles, dim3(bx, n_planes), dim3(NT), 0, mg_stream(stream), d_blur, h, w, d_coords,
                     coord_cap, d_num_edges, d_angle);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
