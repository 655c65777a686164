// A3-A7: edge stage of find_circles (utils.py:20-27, 115-142) and grid_array (utils.py:347-377).
//
//   K1 to_uint8 + GaussianBlur 5x5            u16 -> blurred u8          (read 2, write 1 B/px)
//   K2 Scharr + histogram of m = dx^2+dy^2    (exact np.quantile)        (read 1 B/px)
//   K3 Canny NMS + double threshold           -> weak / strong BITMAPS   (read 1, write 1/4 B/px)
//   K4 hysteresis, bit-parallel in LDS, active tiles only                (bits only)
//   K5 grid_array from the strong bitmap, K6 gradient angle per edge pixel
//
// Stencil kernels stage a 256 x 64 tile (+halo, BORDER_REFLECT_101) through LDS with 16-byte
// loads; every lane owns 4 adjacent pixels (one dword) of 16 rows.  Integer arithmetic throughout:
// bit-exact against the oracle.  Roofline: HBM.
#include <math.h>
#include <stdlib.h>

#include <algorithm>

#include "mg_common.h"

namespace {

constexpr int TW = 256;   // tile width  (pixels) = 64 lanes x 4 px
constexpr int TH = 64;    // tile height (pixels) = 4 waves x 16 rows
constexpr int NT = 256;   // threads per block
constexpr int RPW = 16;   // rows per wave
constexpr int LPAD = 16;  // halo chunk on each side of a tile row (one 16-byte chunk)
constexpr int LS = TW + 2 * LPAD;        // LDS row stride in bytes (288)
constexpr int CHUNKS = LS / 16;          // 16-byte chunks per LDS row (18)

// ---- to_uint8 -------------------------------------------------------------------------
struct U8Scale {
  double mn, top, scale, offset;
  int passthrough;
};

__device__ __forceinline__ U8Scale make_scale(const double* d_minmax, int plane) {
  U8Scale s;
  s.passthrough = (d_minmax == nullptr);
  s.mn = s.passthrough ? 0.0 : d_minmax[2 * plane];
  const double mx = s.passthrough ? 0.0 : d_minmax[2 * plane + 1];
  s.top = mx - s.mn;  // == max(arr - min) because x -> fl(x - mn) is monotone
  // integer inputs: floor(255 (x - mn) / top) == floor(x * scale + offset), see to_u8
  s.scale = s.top > 0.0 ? 255.0 / s.top : 0.0;
  s.offset = s.top > 0.0 ? 0x1p-20 - s.mn * s.scale : 0.0;
  return s;
}

// utils.py:23-27.  Integer inputs: the true quotient 255 (x - mn) / top is either an integer or
// at least 1/top >= 2^-16 away from one, while x * scale + offset carries ~1e-10 of rounding on
// top of the deliberate +2^-20, so one fused multiply-add floors to the same integer as the
// reference's correctly rounded float64 division (a constant image gives scale = offset = 0).
template <typename T>
__device__ __forceinline__ uint8_t to_u8(T x, const U8Scale& s) {
  if (s.passthrough) return (uint8_t)x;
  return (uint8_t)(unsigned int)fma((double)x, s.scale, s.offset);
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

// 16 consecutive elements of a row starting at column gx0 (may be outside the image: reflect-101).
template <typename T>
__device__ __forceinline__ void fetch16(const T* __restrict__ rowp, int gx0, int w, const U8Scale& sc, uint8_t (&v)[16]) {
  const T* p = rowp + gx0;
  if (gx0 >= 0 && gx0 + 16 <= w && (reinterpret_cast<uintptr_t>(p) & 15) == 0) {
    constexpr int PER = 16 / sizeof(T);  // elements per 16-byte load
#pragma unroll
    for (int q = 0; q < 16 / PER; ++q) {
      T e[PER];
      const uint4 raw = reinterpret_cast<const uint4*>(p)[q];
      __builtin_memcpy(e, &raw, 16);
#pragma unroll
      for (int j = 0; j < PER; ++j) v[q * PER + j] = to_u8<T>(e[j], sc);
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) v[j] = to_u8<T>(rowp[mg_reflect101(gx0 + j, w)], sc);
  }
}

// Stage tile rows [ty0 - HALO, ty0 + TH + HALO) x cols [tx0 - 16, tx0 + TW + 16) into LDS.
// Interior tiles (the block-uniform common case) take a branch-free path whose loads are all
// issued before the first conversion; border tiles go through the reflecting per-chunk path.
template <typename T, int HALO>
__device__ __forceinline__ void load_tile(const T* __restrict__ base, int64_t row_stride, int h, int w, int tx0, int ty0,
                                          const U8Scale& sc, uint8_t (*tile)[LS]) {
  constexpr int TOTAL = (TH + 2 * HALO) * CHUNKS;
  constexpr int ITER = (TOTAL + NT - 1) / NT;
  constexpr int PER = 16 / sizeof(T);   // elements per 16-byte load
  constexpr int NLD = 16 / PER;         // 16-byte loads per 16-element chunk
  const bool interior = tx0 >= LPAD && tx0 + TW + LPAD <= w && ty0 >= HALO && ty0 + TH + HALO <= h &&
                        ((row_stride * sizeof(T)) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(base + (int64_t)(ty0 - HALO) * row_stride + tx0 - LPAD) & 15) == 0;
  if (interior) {
    uint4 raw[ITER][NLD];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = threadIdx.x + it * NT;
      if (i < TOTAL) {
        const int j = i / CHUNKS, k = i - j * CHUNKS;
        const uint4* p = reinterpret_cast<const uint4*>(base + (int64_t)(ty0 - HALO + j) * row_stride + tx0 - LPAD + 16 * k);
#pragma unroll
        for (int q = 0; q < NLD; ++q) raw[it][q] = p[q];
      }
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = threadIdx.x + it * NT;
      if (i < TOTAL) {
        const int j = i / CHUNKS, k = i - j * CHUNKS;
        uint8_t v[16];
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          T e[PER];
          __builtin_memcpy(e, &raw[it][q], 16);
#pragma unroll
          for (int m = 0; m < PER; ++m) v[q * PER + m] = to_u8<T>(e[m], sc);
        }
        uint4 out;
        __builtin_memcpy(&out, v, 16);
        *reinterpret_cast<uint4*>(&tile[j][16 * k]) = out;
      }
    }
    return;
  }
  // Border tiles: the chunks that lie wholly inside the image (nearly all of them) still take the
  // batched 16-byte loads, rows through reflect-101; only a chunk that crosses or lies beyond an
  // image edge is assembled pixel by pixel -- and only the HALO pixels next to the tile that a stencil
  // can reach (the rest of the 16-pixel pad is never read).  This path used to walk every chunk one
  // after the other with per-byte loads and cost more than all interior tiles together.
  const bool rows_ok = ((row_stride * sizeof(T)) & 15) == 0;
  uint4 raw[ITER][NLD];
  bool fast[ITER];
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = threadIdx.x + it * NT;
    fast[it] = false;
    if (i < TOTAL) {
      const int j = i / CHUNKS, k = i - j * CHUNKS;
      const int gy = mg_reflect101(ty0 - HALO + j, h), gx0 = tx0 - LPAD + 16 * k;
      const T* p = base + (int64_t)gy * row_stride + gx0;
      fast[it] = rows_ok && gx0 >= 0 && gx0 + 16 <= w && (reinterpret_cast<uintptr_t>(p) & 15) == 0;
      if (fast[it]) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) raw[it][q] = reinterpret_cast<const uint4*>(p)[q];
      }
    }
  }
#pragma unroll
  for (int it = 0; it < ITER; ++it) {
    const int i = threadIdx.x + it * NT;
    if (i < TOTAL) {
      const int j = i / CHUNKS, k = i - j * CHUNKS;
      uint8_t v[16];
      if (fast[it]) {
#pragma unroll
        for (int q = 0; q < NLD; ++q) {
          T e[PER];
          __builtin_memcpy(e, &raw[it][q], 16);
#pragma unroll
          for (int m = 0; m < PER; ++m) v[q * PER + m] = to_u8<T>(e[m], sc);
        }
      } else {
        const int gy = mg_reflect101(ty0 - HALO + j, h), gx0 = tx0 - LPAD + 16 * k;
        const T* rowp = base + (int64_t)gy * row_stride;
#pragma unroll
        for (int m = 0; m < 16; ++m) {
          const int c = 16 * k + m;  // column in the LDS row; the tile proper is [LPAD, LPAD + TW)
          const bool needed = c >= LPAD - HALO && c < LPAD + TW + HALO;
          v[m] = needed ? to_u8<T>(rowp[mg_reflect101(gx0 + m, w)], sc) : (uint8_t)0;
        }
      }
      uint4 out;
      __builtin_memcpy(&out, v, 16);
      *reinterpret_cast<uint4*>(&tile[j][16 * k]) = out;
    }
  }
}

// Bytes c0-4 .. c0+7 of a tile row as three dwords (c0 = 4 * lane is the lane's first pixel).
struct Row12 {
  uint32_t d0, d1, d2;
  __device__ __forceinline__ int at(int off) const {  // off in [-4, 8): pixel c0 + off
    const int b = off + 4;
    const uint32_t d = b < 4 ? d0 : (b < 8 ? d1 : d2);
    return (int)((d >> (8 * (b & 3))) & 0xFFu);
  }
};
__device__ __forceinline__ Row12 read_row(const uint8_t (*tile)[LS], int r, int c0) {
  const uint32_t* p = reinterpret_cast<const uint32_t*>(&tile[r][LPAD + c0 - 4]);
  return Row12{p[0], p[1], p[2]};
}

// ---- K1: to_uint8 + 5x5 Gaussian ([1 4 6 4 1] x [1 4 6 4 1], (sum + 128) >> 8) ----------
template <typename T>
__global__ __launch_bounds__(NT) void k_u8_blur(const T* __restrict__ src, int64_t plane_stride, int h, int w,
                                                int64_t row_stride, const double* __restrict__ d_minmax,
                                                uint8_t* __restrict__ d_blur, uint8_t* __restrict__ d_u8) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[TH + 4][LS];
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  const U8Scale sc = make_scale(d_minmax, plane);
  load_tile<T, 2>(src + (int64_t)plane * plane_stride, row_stride, h, w, tx0, ty0, sc, tile);
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = 4 * lane, gx = tx0 + c0;
  if (gx >= w) return;
  int hs[5][4];
  uint8_t* outp = d_blur + (int64_t)plane * h * w;
  uint8_t* rawp = d_u8 ? d_u8 + (int64_t)plane * h * w : nullptr;
  const bool full = (gx + 4 <= w) && ((w & 3) == 0);
#pragma unroll
  for (int jr = 0; jr < RPW + 4; ++jr) {
    const Row12 r = read_row(tile, wave * RPW + jr, c0);
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) hs[k][q] = hs[k + 1][q];
#pragma unroll
    for (int q = 0; q < 4; ++q) hs[4][q] = r.at(q - 2) + 4 * r.at(q - 1) + 6 * r.at(q) + 4 * r.at(q + 1) + r.at(q + 2);
    if (rawp && jr >= 2 && jr < RPW + 2) {
      const int gy = ty0 + wave * RPW + jr - 2;
      if (gy < h) {
        if (full) *reinterpret_cast<uint32_t*>(rawp + (int64_t)gy * w + gx) = r.d1;
        else
          for (int q = 0; q < 4 && gx + q < w; ++q) rawp[(int64_t)gy * w + gx + q] = (uint8_t)r.at(q);
      }
    }
    if (jr >= 4) {
      const int gy = ty0 + wave * RPW + jr - 4;
      if (gy < h) {
        uint32_t packed = 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int v = hs[0][q] + 4 * hs[1][q] + 6 * hs[2][q] + 4 * hs[3][q] + hs[4][q];
          packed |= (uint32_t)((v + 128) >> 8) << (8 * q);
        }
        if (full) *reinterpret_cast<uint32_t*>(outp + (int64_t)gy * w + gx) = packed;
        else
          for (int q = 0; q < 4 && gx + q < w; ++q) outp[(int64_t)gy * w + gx + q] = (uint8_t)(packed >> (8 * q));
      }
    }
  }
}

// Scharr of N adjacent pixels starting at pixel c0 + FIRST from three rows, via the separable
// parts S = 3a + 10b + 3c (vertical smooth) and D = c - a (vertical difference):
//   dx[q] = S[q+1] - S[q-1],  dy[q] = 3 (D[q-1] + D[q+1]) + 10 D[q].
template <int FIRST, int N>
__device__ __forceinline__ void scharr_n(const Row12& a, const Row12& b, const Row12& c, int (&dx)[N], int (&dy)[N]) {
  int S[N + 2], D[N + 2];
#pragma unroll
  for (int j = 0; j < N + 2; ++j) {
    const int o = FIRST - 1 + j;
    S[j] = 3 * (a.at(o) + c.at(o)) + 10 * b.at(o);
    D[j] = c.at(o) - a.at(o);
  }
#pragma unroll
  for (int q = 0; q < N; ++q) {
    dx[q] = S[q + 2] - S[q];
    dy[q] = 3 * (D[q] + D[q + 2]) + 10 * D[q + 1];
  }
}

// ---- K2: histogram of m = dx^2 + dy^2 ------------------------------------------------------
// mode 0 (combined): bins [0, 8192) hold m exactly, bins 8192 + (m >> 13) hold the rest coarsely;
// mode 1 (window):   bin m - base for m in [base, base + 8192).
// m == 0 (flat background, by far the most frequent value) is counted in registers.
constexpr int FINE = 8192;
constexpr int COARSE = 4096;
constexpr uint32_t MG_HIST_SKIP = 0xFFFFFFFFu;  // mode 1: d_base[plane] of a plane that takes no part in the pass

// A workgroup accumulates HIST_TILES vertically adjacent tiles (3 x 16384 pixels: a 16-bit counter
// cannot overflow) before it hands its histogram over: a plain coalesced store of the packed counters
// into the workgroup's own slot of d_partial, summed by k_hist_reduce (or, without d_partial, one
// global atomicAdd per non-empty bin).
// DIRECT (interior tile groups): no LDS tile at all -- a lane loads its own dword of each of the 18
// rows of its wave's strip straight into registers (all loads in flight at once), neighbours' pixels
// come from the adjacent lanes by wave shuffles, so the waves of a workgroup never wait for each
// other and LDS holds only the histogram (6 instead of 3 workgroups per CU).  Border tile groups go
// through the LDS-staged path (reflect-101) in a second launch.
constexpr int HIST_TILES = 3;

typedef short hs2 __attribute__((ext_vector_type(2)));
typedef unsigned short hu2 __attribute__((ext_vector_type(2)));

// The four pixels c0 .. c0+3 of the centre row pb2, from three rows given as pixel PAIRS (c0-1, c0), (c0+1, c0+2),
// (c0+3, c0+4): packed 16-bit arithmetic, two pixels per instruction; |gradient|^2 by one dot2 per pixel (see
// k_canny_nms).
__device__ __forceinline__ void hist_add4_pairs(const hu2 (&pa)[3], const hu2 (&pb2)[3], const hu2 (&pc)[3], int gx, int w,
                                                int mode, uint32_t base, int n_bins, uint32_t* hist, uint32_t& zeros) {
  typedef hs2 s2;
  typedef hu2 u2;
  s2 S[3], D[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    S[k] = __builtin_bit_cast(s2, (u2)((pa[k] + pc[k]) * (unsigned short)3 + pb2[k] * (unsigned short)10));
    D[k] = __builtin_bit_cast(s2, pc[k]) - __builtin_bit_cast(s2, pa[k]);
  }
  uint32_t xy[4];  // (dx | dy << 16) of pixels c0 .. c0+3
#pragma unroll
  for (int jp = 0; jp < 2; ++jp) {
    const s2 dxp = S[jp + 1] - S[jp];
    const s2 mid = __builtin_bit_cast(s2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, D[jp + 1]),
                                                                     __builtin_bit_cast(uint32_t, D[jp]), 16));
    const s2 dyp = (D[jp] + D[jp + 1]) * (short)3 + mid * (short)10;
    xy[2 * jp] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, dyp), __builtin_bit_cast(uint32_t, dxp), 0x05040100u);
    xy[2 * jp + 1] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, dyp), __builtin_bit_cast(uint32_t, dxp), 0x07060302u);
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    if (gx + q >= w) continue;
    const uint32_t m = (uint32_t)__builtin_amdgcn_sdot2(__builtin_bit_cast(s2, xy[q]), __builtin_bit_cast(s2, xy[q]), 0, false);
    if (mode == 0) {
      if (m == 0) ++zeros;
      else {
        const uint32_t b = m < FINE ? m : FINE + (m >> 13);
        atomicAdd(&hist[b >> 1], 1u << (16 * (b & 1)));
      }
    } else if (m >= base && m - base < (uint32_t)n_bins) {
      if (m == base) ++zeros;
      else {
        const uint32_t b = m - base;
        atomicAdd(&hist[b >> 1], 1u << (16 * (b & 1)));
      }
    }
  }
}

__device__ __forceinline__ void hist_add4(const Row12& ra, const Row12& rb, const Row12& rc, int gx, int w, int mode,
                                          uint32_t base, int n_bins, uint32_t* hist, uint32_t& zeros) {
  typedef hu2 u2;
  u2 pa[3], pb2[3], pc[3];
#define MG_UNPACK3(r, o)                                                                    \
  o[0] = __builtin_bit_cast(u2, __builtin_amdgcn_perm(r.d1, r.d0, 0x0C040C03u)); /* c0-1, c0 */ \
  o[1] = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, r.d1, 0x0C020C01u));   /* c0+1, c0+2 */ \
  o[2] = __builtin_bit_cast(u2, __builtin_amdgcn_perm(r.d2, r.d1, 0x0C040C03u)); /* c0+3, c0+4 */
  MG_UNPACK3(ra, pa)
  MG_UNPACK3(rb, pb2)
  MG_UNPACK3(rc, pc)
#undef MG_UNPACK3
  hist_add4_pairs(pa, pb2, pc, gx, w, mode, base, n_bins, hist, zeros);
}

__device__ __forceinline__ bool hist_group_interior(const uint8_t* pb, int h, int w, int tx0, int gy) {
  // "interior" for the register-direct kernel = every tile of the group lies inside the image
  // horizontally (rows and the image's left / right edge are reflected in the kernel itself)
  return gy * HIST_TILES * TH < h && tx0 + TW <= w && (w & 3) == 0 && h >= 2 &&
         (reinterpret_cast<uintptr_t>(pb) & 3) == 0;
}

template <bool DIRECT>
__global__ __launch_bounds__(NT) void k_scharr_hist(const uint8_t* __restrict__ d_blur, int h, int w, int mode,
                                                    const uint32_t* __restrict__ d_base, int n_bins,
                                                    uint32_t* __restrict__ d_hist, uint32_t* __restrict__ d_partial,
                                                    int split) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint8_t (*tile)[LS] = reinterpret_cast<uint8_t (*)[LS]>(smem);
  // 16-bit counters, two per word
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem + (DIRECT ? 0 : (TH + 2) * LS));
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW;
  const uint8_t* pb = d_blur + (int64_t)plane * h * w;
  // DIRECT: grid.y counts tile groups; staged: grid.y counts tiles (one tile per workgroup)
  const bool interior = hist_group_interior(pb, h, w, tx0, DIRECT ? blockIdx.y : blockIdx.y / HIST_TILES);
  uint32_t* slot = (DIRECT && d_partial) ? d_partial + ((int64_t)plane * gridDim.y * gridDim.x +
                                                        (int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (n_bins / 2)
                                         : nullptr;
  const uint32_t base = (mode == 1 && d_base) ? d_base[plane] : 0u;
  if (mode == 1 && base == MG_HIST_SKIP) return;  // this plane needs no window pass (k_hist_reduce skips it too)
  if (DIRECT && !interior) {  // the staged launch handles this group (and adds straight into d_hist)
    if (slot)
      for (int i = threadIdx.x; i < n_bins / 2; i += NT) slot[i] = 0u;
    return;
  }
  if (!DIRECT && split && interior) return;
  for (int i = threadIdx.x; i < n_bins / 2; i += NT) hist[i] = 0;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = 4 * lane, gx = tx0 + c0;
  uint32_t zeros = 0;
  if (DIRECT) {
    __syncthreads();  // histogram zeroed
#pragma unroll 1
    for (int sub = 0; sub < HIST_TILES; ++sub) {
      const int y0 = (blockIdx.y * HIST_TILES + sub) * TH + wave * RPW;  // first row of this wave's strip
      if (y0 >= h) break;  // wave-uniform: this strip lies below the image
      const bool at_left = tx0 == 0, at_right = tx0 + TW == w;  // image edges: BORDER_REFLECT_101
      uint32_t mid[RPW + 2], edge[RPW + 2];
#pragma unroll
      for (int j = 0; j < RPW + 2; ++j) {
        const int ry = mg_reflect101(y0 - 1 + j, h);
        const uint32_t* rowp = reinterpret_cast<const uint32_t*>(pb + (int64_t)ry * w + tx0);
        mid[j] = rowp[lane];
        edge[j] = 0;
        if (lane == 0 && !at_left) edge[j] = rowp[-1];
        else if (lane == 63 && !at_right) edge[j] = rowp[64];
      }
      Row12 ra, rb;
#pragma unroll
      for (int j = 0; j < RPW + 2; ++j) {
        // neighbours' words by whole-wave DPP shifts (wave_shr:1 / wave_shl:1), not LDS permutes
        uint32_t d0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mid[j], 0x138, 0xF, 0xF, false);
        uint32_t d2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mid[j], 0x130, 0xF, 0xF, false);
        // pixel -1 of the image is pixel 1, pixel w is pixel w - 2 (only those two bytes are ever read)
        if (lane == 0) d0 = at_left ? (mid[j] & 0x0000FF00u) << 16 : edge[j];
        if (lane == 63) d2 = at_right ? (mid[j] >> 16) & 0xFFu : edge[j];
        const Row12 rc{d0, mid[j], d2};
        if (j >= 2 && y0 + j - 2 < h) hist_add4(ra, rb, rc, gx, w, mode, base, n_bins, hist, zeros);
        ra = rb;
        rb = rc;
      }
    }
  } else {
    U8Scale sc;
    sc.passthrough = 1;
    const int ty0 = blockIdx.y * TH;
    load_tile<uint8_t, 1>(pb, w, h, w, tx0, ty0, sc, tile);
    __syncthreads();  // tile staged, histogram zeroed
    Row12 ra = read_row(tile, wave * RPW, c0), rb = read_row(tile, wave * RPW + 1, c0);
#pragma unroll 4
    for (int jr = 2; jr < RPW + 2; ++jr) {
      const Row12 rc = read_row(tile, wave * RPW + jr, c0);
      const int gy = ty0 + wave * RPW + jr - 2;
      if (gy < h && gx < w) hist_add4(ra, rb, rc, gx, w, mode, base, n_bins, hist, zeros);
      ra = rb;
      rb = rc;
    }
  }
  zeros = (uint32_t)mg_wave_sum_i32((int)zeros);
  uint32_t* out = d_hist + (int64_t)plane * n_bins;
  // bin 0: counted in registers.  With a slot it joins the workgroup's packed counters (<= 3 x 16384 pixels fit 16
  // bits; one atomic per wave on the plane's one global counter serialised ~1400 deep), else global memory.
  if (lane == 0 && zeros) {
    if (slot) atomicAdd(&hist[0], zeros);
    else atomicAdd(&out[0], zeros);
  }
  __syncthreads();
  if (slot) {
    for (int i = threadIdx.x; i < n_bins / 2; i += NT) slot[i] = hist[i];
    return;
  }
  for (int i = threadIdx.x; i < n_bins / 2; i += NT) {
    const uint32_t v = hist[i];
    if (v & 0xFFFFu) atomicAdd(&out[2 * i], v & 0xFFFFu);
    if (v >> 16) atomicAdd(&out[2 * i + 1], v >> 16);
  }
}

// ---- K1 + K2 in one pass (integer inputs, w % 4 == 0): the blurred rows feed the histogram from registers ----
// Both passes are bound by vector issue, not by their bytes, and the histogram pass read the blurred image back
// (3.3 GB fetched at 64 planes of 4096^2 for 1.07 GB) only to do arithmetic on it.  Here a wave walks down a strip of
// FR rows: a lane loads its four pixels of an input row straight into registers, the neighbours' pixels come by
// whole-wave DPP shifts (no LDS tile, no workgroup barrier in the loop), the 5 x 5 blur runs in packed 16-bit
// arithmetic (every intermediate fits: 16 x 255 per row pass, 256 x 255 per column pass), the blurred dword is
// stored, and the Scharr magnitudes of the row before it -- three blurred rows are kept as pixel pairs -- enter the
// workgroup's histogram in LDS.  Lanes 1 .. 62 own the strip's FW = 248 columns, lanes 0 and 63 recompute the
// neighbours' nearest columns (a strip needs the blurred columns next to it, which belong to another workgroup);
// likewise a wave blurs one row above and below its strip.  BORDER_REFLECT_101 of the blurred image equals the blur
// of the reflected input (the kernel is symmetric), so the halo simply reads reflected input pixels.
constexpr int FW = 248;           // columns owned by a workgroup
constexpr int FR = 64;            // rows per wave
constexpr int FH = (NT / 64) * FR;
constexpr int FB = 5;             // rows per batch of loads; two batches per trip (the 5-row ring returns to its place)
static_assert((FR + 6) % (2 * FB) == 0, "row loop");

template <typename T>
struct alignas(4 * sizeof(T)) Raw4 {
  T v[4];
};

// One wave's strip.  VEC: every lane's four columns lie inside the image row (one aligned load per lane and row);
// else (the first / last strips of a row of strips) every lane reads its four reflected columns one by one.  The
// choice is the workgroup's, not the lane's: a branch around a load inside the loop would end every load with a wait.
template <typename T, bool VEC>
__device__ __forceinline__ void blur_hist_strip(const T* __restrict__ pin, int64_t row_stride, int h, int w, int y0, int cx,
                                                bool owner, const U8Scale& sc, uint8_t* __restrict__ pout, uint32_t* hist,
                                                uint32_t& zeros) {
  typedef hu2 u2;
  constexpr int n_bins = FINE + COARSE;
  const int y_end = min(y0 + FR, h);
  uint32_t col[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) col[q] = VEC ? (uint32_t)(cx + q) : (uint32_t)mg_reflect101(cx + q, w);
  auto load = [&](int j) {  // input row y0 - 3 + j
    int ry = y0 - 3 + j;
    if (ry < 0) ry = -ry;
    if (ry >= h) ry = mg_reflect101(ry, h);
    const T* rowp = pin + (int64_t)ry * row_stride;
    Raw4<T> r;
    if (VEC) {
      r = *reinterpret_cast<const Raw4<T>*>(rowp + col[0]);
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q) r.v[q] = rowp[col[q]];
    }
    return r;
  };
  u2 H[5][2];   // row-pass sums of the five newest input rows, pixels (0, 1) and (2, 3)
  u2 P[3][3];   // the three newest blurred rows as pairs (c-1, c), (c+1, c+2), (c+3, c+4)
#pragma unroll
  for (int k = 0; k < 5; ++k) H[k][0] = H[k][1] = (u2)(0);
#pragma unroll
  for (int k = 0; k < 3; ++k) P[k][0] = P[k][1] = P[k][2] = (u2)(0);
  Raw4<T> cur[FB], nxt[FB];
#pragma unroll
  for (int b = 0; b < FB; ++b) cur[b] = load(b);
#pragma unroll 1
  for (int j0 = 0; j0 < FR + 6; j0 += 2 * FB) {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int jb = j0 + half * FB;
      // the rows of the next batch: in flight while this one is worked on (past the strip: the last row again)
#pragma unroll
      for (int b = 0; b < FB; ++b) nxt[b] = load(min(jb + FB + b, FR + 5));
#pragma unroll
      for (int b = 0; b < FB; ++b) {
        const int j = jb + b;
        // (to_u8's float64 multiply-add is not what this loop waits for: an exact float32 quotient with one integer
        // correction, seven single-rate instructions per pixel, ran 2 % slower)
        const uint32_t d1 = (uint32_t)to_u8<T>(cur[b].v[0], sc) | ((uint32_t)to_u8<T>(cur[b].v[1], sc) << 8) |
                            ((uint32_t)to_u8<T>(cur[b].v[2], sc) << 16) | ((uint32_t)to_u8<T>(cur[b].v[3], sc) << 24);
        const uint32_t d0 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d1, 0x138, 0xF, 0xF, false);  // lane - 1
        const uint32_t d2 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d1, 0x130, 0xF, 0xF, false);  // lane + 1
        // pixel pairs at even and odd offsets: (-2, -1) (0, 1) (2, 3) (4, 5) and (-1, 0) (1, 2) (3, 4)
        const u2 em = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, d0, 0x0C030C02u));
        const u2 e0 = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, d1, 0x0C010C00u));
        const u2 e2 = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, d1, 0x0C030C02u));
        const u2 e4 = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, d2, 0x0C010C00u));
        const u2 om = __builtin_bit_cast(u2, __builtin_amdgcn_perm(d1, d0, 0x0C040C03u));
        const u2 o1 = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, d1, 0x0C020C01u));
        const u2 o3 = __builtin_bit_cast(u2, __builtin_amdgcn_perm(d2, d1, 0x0C040C03u));
#pragma unroll
        for (int k = 0; k < 4; ++k) H[k][0] = H[k + 1][0], H[k][1] = H[k + 1][1];
        H[4][0] = (em + e2) + (om + o1) * (unsigned short)4 + e0 * (unsigned short)6;
        H[4][1] = (e0 + e4) + (o1 + o3) * (unsigned short)4 + e2 * (unsigned short)6;
        // blurred row br (centre of the five newest input rows); meaningful from j = 4 on
        const int br = y0 - 1 + (j - 4);
        u2 B[2];
#pragma unroll
        for (int k = 0; k < 2; ++k)
          B[k] = ((H[0][k] + H[4][k]) + (H[1][k] + H[3][k]) * (unsigned short)4 + H[2][k] * (unsigned short)6 +
                  (unsigned short)128) >> (unsigned short)8;
        const uint32_t b01 = __builtin_bit_cast(uint32_t, B[0]), b23 = __builtin_bit_cast(uint32_t, B[1]);
        if (br >= y0 && br < y_end && owner)
          *reinterpret_cast<uint32_t*>(pout + (int64_t)br * w + cx) = __builtin_amdgcn_perm(b23, b01, 0x06040200u);
        // pairs at odd offsets of the blurred row: (-1, 0) with the left lane's (2, 3), (3, 4) with the right lane's (0, 1)
        const uint32_t l23 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b23, 0x138, 0xF, 0xF, false);
        const uint32_t r01 = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)b01, 0x130, 0xF, 0xF, false);
#pragma unroll
        for (int k = 0; k < 3; ++k) P[0][k] = P[1][k], P[1][k] = P[2][k];
        P[2][0] = __builtin_bit_cast(u2, __builtin_amdgcn_alignbit(b01, l23, 16));
        P[2][1] = __builtin_bit_cast(u2, __builtin_amdgcn_alignbit(b23, b01, 16));
        P[2][2] = __builtin_bit_cast(u2, __builtin_amdgcn_alignbit(r01, b23, 16));
        const int sr = br - 1;  // the Scharr row: centre of the three newest blurred rows
        if (sr >= y0 && sr < y_end && owner) hist_add4_pairs(P[0], P[1], P[2], cx, w, 0, 0u, n_bins, hist, zeros);
      }
#pragma unroll
      for (int b = 0; b < FB; ++b) cur[b] = nxt[b];
    }
  }
}

template <typename T>
__global__ __launch_bounds__(NT) void k_blur_hist(const T* __restrict__ src, int64_t plane_stride, int h, int w,
                                                  int64_t row_stride, const double* __restrict__ d_minmax,
                                                  uint8_t* __restrict__ d_blur, uint32_t* __restrict__ d_partial) {
  extern __shared__ __attribute__((aligned(16))) uint8_t smem[];
  uint32_t* hist = reinterpret_cast<uint32_t*>(smem);  // FINE + COARSE 16-bit counters (<= FW * FH = 63 488 pixels)
  constexpr int n_bins = FINE + COARSE;
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * FW;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // (rows: scalar arithmetic)
  const int y0 = blockIdx.y * FH + wave * FR;
  const int cx = tx0 - 4 + 4 * lane;  // the lane's first column
  const bool owner = lane >= 1 && lane <= 62 && cx < w;
  const U8Scale sc = make_scale(d_minmax, plane);
  const T* pin = src + (int64_t)plane * plane_stride;
  uint8_t* pout = d_blur + (int64_t)plane * h * w;
  for (int i = threadIdx.x; i < n_bins / 2; i += NT) hist[i] = 0;
  __syncthreads();
  uint32_t zeros = 0;
  if (y0 < h) {  // (a strip below the image: nothing to do but the hand-over)
    if (tx0 >= 4 && tx0 + FW + 4 <= w) blur_hist_strip<T, true>(pin, row_stride, h, w, y0, cx, owner, sc, pout, hist, zeros);
    else blur_hist_strip<T, false>(pin, row_stride, h, w, y0, cx, owner, sc, pout, hist, zeros);
  }
  zeros = (uint32_t)mg_wave_sum_i32((int)zeros);
  if (lane == 0 && zeros) atomicAdd(&hist[0], zeros);  // (bin 0 shares the packed counters: <= 63 488 in all)
  __syncthreads();
  uint32_t* slot = d_partial + ((int64_t)plane * gridDim.y * gridDim.x + (int64_t)blockIdx.y * gridDim.x + blockIdx.x) * (n_bins / 2);
  for (int i = threadIdx.x; i < n_bins / 2; i += NT) slot[i] = hist[i];
}

// d_hist[plane][bin] += sum over the plane's workgroup slots (packed 16-bit pairs).  The slots are split over
// blockIdx.z (HIST_SLOTS_PER_BLOCK each, eight loads in flight per lane) and added with atomics: with one block
// per 256 words walking all ~350 slots one dependent load at a time this took 84 us for a single plane.
constexpr int HIST_SLOTS_PER_BLOCK = 32;

__global__ __launch_bounds__(NT) void k_hist_reduce(const uint32_t* __restrict__ d_partial, int slots, int n_bins,
                                                    uint32_t* __restrict__ d_hist, const uint32_t* __restrict__ d_skip_base) {
  const int plane = blockIdx.y;
  if (d_skip_base && d_skip_base[plane] == MG_HIST_SKIP) return;
  const int i = blockIdx.x * NT + threadIdx.x;  // packed word = bins 2i, 2i + 1
  if (i >= n_bins / 2) return;
  const int s0 = blockIdx.z * HIST_SLOTS_PER_BLOCK, s1 = min(s0 + HIST_SLOTS_PER_BLOCK, slots);
  const int64_t step = n_bins / 2;
  const uint32_t* p = d_partial + ((int64_t)plane * slots + s0) * step + i;
  uint32_t lo = 0, hi = 0;
  int s = s0;
  for (; s + 8 <= s1; s += 8, p += 8 * step) {
    uint32_t v[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) v[q] = p[q * step];
#pragma unroll
    for (int q = 0; q < 8; ++q) lo += v[q] & 0xFFFFu, hi += v[q] >> 16;
  }
  for (; s < s1; ++s, p += step) {
    const uint32_t v = *p;
    lo += v & 0xFFFFu;
    hi += v >> 16;
  }
  uint32_t* out = d_hist + (int64_t)plane * n_bins;
  if (lo) atomicAdd(&out[2 * i], lo);  // bin 0 also receives the register-counted zeros by atomicAdd in
  if (hi) atomicAdd(&out[2 * i + 1], hi);  // k_scharr_hist, which has completed: kernels on a stream run in order
}

// ---- K2b: np.quantile + cv::Canny's threshold preparation, from the histogram, on the device -----------------
// One workgroup per plane.  The order statistics of ranks[0..3] (= prev / next index of np.quantile's linear
// interpolation for the low and the high quantile) are the bins where the running sum of the histogram first
// exceeds the rank; a fine bin IS the value m = dx^2 + dy^2.  Then, exactly as NumPy 2.x and OpenCV do it
// (utils.py:120-134): g = sqrt(float32(m)) in float32, _lerp in float32 with the float32 weight gamma, the two
// thresholds ordered, clamped to 32767, squared in float64 and floored.
// A rank that falls into a COARSE bin (strong gradients: noiseless images, the chip's per-chamber windows) is
// resolved by WINDOW passes that stay on the device: the plane's state block records, per rank, the coarse bin and
// the rank's position inside it; pass p re-histograms the plane's p-th distinct coarse bin exactly (mg_scharr_hist
// mode 1 with the base this kernel / k_window_resolve wrote) and k_window_resolve turns the positions into order
// statistics.  d_unresolved[plane] = passes still to run (low byte; 0: thresholds valid) | passes needed in all << 8.
constexpr int TS_ORDER = 0, TS_COARSE = 4, TS_RESID = 8, TS_NWIN = 12, TS_WORDS = 16;  // int32 words of a state block

__device__ __forceinline__ void finish_thresholds(const int32_t* order, float gamma_lo, float gamma_hi, int plane,
                                                  int32_t* __restrict__ d_thresh, float* __restrict__ d_quantiles) {
  float q[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const float ga = (float)sqrt((double)order[2 * k]), gb = (float)sqrt((double)order[2 * k + 1]);  // == float32 sqrt
    const float gamma = k == 0 ? gamma_lo : gamma_hi;
    const float diff = gb - ga;
    q[k] = gamma >= 0.5f ? gb - diff * (1.0f - gamma) : ga + diff * gamma;  // numpy's _lerp, float32 throughout
  }
  double lo = (double)fminf(q[0], q[1]), hi = (double)fmaxf(q[0], q[1]);
  lo = fmin(lo, 32767.0);
  hi = fmin(hi, 32767.0);
  if (lo > 0.0) lo = lo * lo;
  if (hi > 0.0) hi = hi * hi;
  d_thresh[2 * plane] = (int32_t)floor(lo);
  d_thresh[2 * plane + 1] = (int32_t)floor(hi);
  d_quantiles[2 * plane] = q[0];
  d_quantiles[2 * plane + 1] = q[1];
}

// The p-th smallest distinct coarse bin among the four ranks' (entries < 0: resolved), or -1.
__device__ __forceinline__ int nth_window(const int32_t* coarse, int p) {
  int last = -1;
  for (int it = 0; it <= p; ++it) {
    int best = 0x7FFFFFFF;
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (coarse[k] > last && coarse[k] < best) best = coarse[k];
    if (best == 0x7FFFFFFF) return -1;
    last = best;
  }
  return last;
}

// Exclusive prefix of per-thread 64-bit sums over the workgroup (Hillis-Steele in LDS).
__device__ __forceinline__ long long block_exscan_i64(long long mine, long long* s_pre) {
  s_pre[threadIdx.x] = mine;
  __syncthreads();
  for (int off = 1; off < NT; off <<= 1) {
    const long long v = (int)threadIdx.x >= off ? s_pre[threadIdx.x - off] : 0;
    __syncthreads();
    s_pre[threadIdx.x] += v;
    __syncthreads();
  }
  return s_pre[threadIdx.x] - mine;
}

__global__ __launch_bounds__(NT) void k_edge_thresholds(const uint32_t* __restrict__ d_hist, int n_fine, int n_bins,
                                                        long long r0, long long r1, long long r2, long long r3,
                                                        float gamma_lo, float gamma_hi, int32_t* __restrict__ d_thresh,
                                                        float* __restrict__ d_quantiles, int32_t* __restrict__ d_unresolved,
                                                        int32_t* __restrict__ d_state, uint32_t* __restrict__ d_win_base) {
  __shared__ int s_bin[4];
  __shared__ long long s_below[4];
  __shared__ uint32_t s_hist[FINE + COARSE];  // the plane's histogram, staged by coalesced loads (n_bins <= FINE + COARSE)
  const int plane = blockIdx.x;
  const uint32_t* hist_g = d_hist + (int64_t)plane * n_bins;
  for (int i = threadIdx.x; i < n_bins; i += NT) s_hist[i] = hist_g[i];
  __syncthreads();
  const uint32_t* hist = s_hist;
  const int per = (n_bins + NT - 1) / NT;
  const int b0 = threadIdx.x * per, b1 = min(b0 + per, n_bins);
  long long mine = 0;
  for (int b = b0; b < b1; ++b) mine += hist[b];
  // exclusive prefix of the chunk sums (64-bit: a plane holds up to 2^31 pixels)
  __shared__ long long s_pre[NT];
  long long run = block_exscan_i64(mine, s_pre);
  if (threadIdx.x < 4) s_bin[threadIdx.x] = n_bins, s_below[threadIdx.x] = 0;  // rank beyond the data
  __syncthreads();
  const long long ranks[4] = {r0, r1, r2, r3};
  for (int b = b0; b < b1; ++b) {
    const long long nxt = run + hist[b];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (run <= ranks[k] && ranks[k] < nxt) s_bin[k] = b, s_below[k] = run;  // first bin whose running sum exceeds the rank
    run = nxt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t order[4], coarse[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const bool fine = s_bin[k] < n_fine || s_bin[k] >= n_bins;  // (a rank beyond the data cannot occur: n = h w)
      order[k] = fine ? s_bin[k] : 0;
      coarse[k] = fine ? -1 : s_bin[k] - n_fine;
    }
    int n_win = 0;
    while (nth_window(coarse, n_win) >= 0) ++n_win;
    d_unresolved[plane] = n_win | (n_win << 8);  // passes still to run | passes this plane needs in all
    if (d_state) {
      int32_t* st = d_state + (int64_t)plane * TS_WORDS;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        st[TS_ORDER + k] = order[k];
        st[TS_COARSE + k] = coarse[k];
        st[TS_RESID + k] = (int32_t)(ranks[k] - s_below[k]);  // position inside the coarse bin (< its count < 2^31)
      }
      st[TS_NWIN] = n_win;
    }
    if (d_win_base) d_win_base[plane] = n_win > 0 ? (uint32_t)nth_window(coarse, 0) << 13 : MG_HIST_SKIP;
    finish_thresholds(order, gamma_lo, gamma_hi, plane, d_thresh, d_quantiles);  // (valid when n_win == 0)
  }
}

// Window pass `pass`: d_hist_win[plane][0 .. FINE) counts m - base exactly for the plane's pass-th distinct coarse
// bin.  Ranks that sit in that bin get their order statistic; after the plane's last window the thresholds are
// computed and d_unresolved[plane] drops to 0.  Writes the base of the next pass.
__global__ __launch_bounds__(NT) void k_window_resolve(const uint32_t* __restrict__ d_hist_win, int pass, float gamma_lo,
                                                       float gamma_hi, int32_t* __restrict__ d_state,
                                                       uint32_t* __restrict__ d_win_base, int32_t* __restrict__ d_thresh,
                                                       float* __restrict__ d_quantiles, int32_t* __restrict__ d_unresolved) {
  __shared__ int s_pos[4];
  __shared__ long long s_pre[NT];
  const int plane = blockIdx.x;
  int32_t* st = d_state + (int64_t)plane * TS_WORDS;
  const int n_win = st[TS_NWIN];
  if (pass >= n_win) return;  // block-uniform (its base was MG_HIST_SKIP: nothing was counted)
  int32_t coarse[4], resid[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) coarse[k] = st[TS_COARSE + k], resid[k] = st[TS_RESID + k];
  const int cbin = nth_window(coarse, pass);
  const uint32_t* hist = d_hist_win + (int64_t)plane * FINE;
  constexpr int PER = FINE / NT;
  uint32_t mine_v[PER];
  long long mine = 0;
#pragma unroll
  for (int j = 0; j < PER; ++j) mine_v[j] = hist[threadIdx.x * PER + j], mine += mine_v[j];
  long long run = block_exscan_i64(mine, s_pre);
  if (threadIdx.x < 4) s_pos[threadIdx.x] = 0;
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const long long nxt = run + mine_v[j];
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (coarse[k] == cbin && run <= (long long)resid[k] && (long long)resid[k] < nxt) s_pos[k] = threadIdx.x * PER + j;
    run = nxt;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int32_t order[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      order[k] = st[TS_ORDER + k];
      if (coarse[k] == cbin) order[k] = (cbin << 13) + s_pos[k], st[TS_ORDER + k] = order[k];
    }
    const int left = n_win - 1 - pass;
    d_unresolved[plane] = left | (n_win << 8);
    d_win_base[plane] = left > 0 ? (uint32_t)nth_window(coarse, pass + 1) << 13 : MG_HIST_SKIP;
    if (left == 0) finish_thresholds(order, gamma_lo, gamma_hi, plane, d_thresh, d_quantiles);
  }
}

// ---- bit helpers on the linear (y * w + x) bitmaps --------------------------------------------
__device__ __forceinline__ uint32_t bits_at(const uint32_t* __restrict__ bits, int64_t bit0, int n) {
  // n (1..32) consecutive bits starting at linear bit index bit0
  const int64_t wi = bit0 >> 5;
  const int sh = (int)(bit0 & 31);
  uint64_t two = bits[wi];
  if (sh + n > 32) two |= (uint64_t)bits[wi + 1] << 32;
  const uint32_t v = (uint32_t)(two >> sh);
  return n >= 32 ? v : (v & ((1u << n) - 1u));
}
__device__ __forceinline__ void bits_or(uint32_t* __restrict__ bits, int64_t bit0, uint32_t v) {
  // OR a 32-bit group whose bit 0 sits at linear bit index bit0
  if (!v) return;
  const int64_t wi = bit0 >> 5;
  const int sh = (int)(bit0 & 31);
  atomicOr(&bits[wi], v << sh);
  if (sh && (v >> (32 - sh))) atomicOr(&bits[wi + 1], v >> (32 - sh));
}
// Row segment [x0, x0 + 32) of image row y, zero outside the image.
__device__ __forceinline__ uint32_t row_word(const uint32_t* __restrict__ bits, int h, int w, int y, int x0) {
  if (y < 0 || y >= h) return 0u;
  const int lo = max(x0, 0), hi = min(x0 + 32, w);
  if (lo >= hi) return 0u;
  return bits_at(bits, (int64_t)y * w + lo, hi - lo) << (lo - x0);
}

// v = 2 v + (this lane's bit of a wave mask): the mask is the carry-in of one add-with-carry.
__device__ __forceinline__ uint32_t shl1_or(uint32_t v, uint64_t mask) {
  uint32_t out;
  uint64_t carry_out;
  asm("v_addc_co_u32_e64 %0, %1, %2, %2, %3" : "=v"(out), "=s"(carry_out) : "v"(v), "s"(mask));
  return out;
}

// OR over each aligned group of 8 lanes, on the VALU's data-parallel-primitive paths (no LDS
// crossbar trip): xor 1 and xor 2 inside the quads, then lane i <-> 7 - i swaps the two quads.
__device__ __forceinline__ uint32_t or_reduce8(uint32_t v) {
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, true);   // quad_perm [1, 0, 3, 2]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
  v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, true);  // row_half_mirror
  return v;
}

// ---- K3: Canny non-maximum suppression + double threshold -> weak / strong bitmaps --------------
__global__ __launch_bounds__(NT) void k_canny_nms(const uint8_t* __restrict__ d_blur, int h, int w,
                                                  const int32_t* __restrict__ d_thresh, int64_t words_per_plane,
                                                  uint32_t* __restrict__ d_weak, uint32_t* __restrict__ d_strong,
                                                  uint32_t* __restrict__ d_class) {
  __shared__ __attribute__((aligned(16))) uint8_t tile[TH + 4][LS];
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * TH;
  U8Scale sc;
  sc.passthrough = 1;
  load_tile<uint8_t, 2>(d_blur + (int64_t)plane * h * w, w, h, w, tx0, ty0, sc, tile);
  __syncthreads();
  const int low = d_thresh[2 * plane], high = d_thresh[2 * plane + 1];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c0 = 4 * lane, gx = tx0 + c0;
  uint32_t* weak = d_weak + plane * words_per_plane;
  uint32_t* strong = d_strong + plane * words_per_plane;
  // Orientation bin of the gradient, phi = atan2(dy, dx) mod pi in eighths of pi (exact, from the
  // integer gradient): bit planes c0, c1 (quarter index 2 c1 + c0) and c2 (upper half of the quarter),
  // bin = 4 c1 + 2 c0 + c2.  The scoring prefilter bounds every perimeter term by the distance between
  // the point's radial direction and the pixel's bin.
  uint32_t* cls0 = d_class ? d_class + (3 * plane) * words_per_plane : nullptr;
  uint32_t* cls1 = d_class ? d_class + (3 * plane + 1) * words_per_plane : nullptr;
  uint32_t* cls2 = d_class ? d_class + (3 * plane + 2) * words_per_plane : nullptr;
  constexpr int TG22 = 13573;
  const uint32_t in_row = gx + 4 <= w ? 0xFu : ((1u << max(w - gx, 0)) - 1u);  // the lane's pixels left of the row end
  // Packed 16-bit arithmetic, two pixels per instruction: a row is held as four registers of pixel pairs
  // (columns c0-2 .. c0+5); the Scharr sums S = 3 (above + below) + 10 centre <= 4080 and differences fit 16 bits;
  // |gradient|^2 = dot2((dx, dy), (dx, dy)) in one instruction per pixel.
  typedef short s2 __attribute__((ext_vector_type(2)));
  typedef unsigned short u2 __attribute__((ext_vector_type(2)));
  struct Pairs4 {
    u2 p[4];
  };
  auto unpack = [](const Row12& r) {
    Pairs4 o;
    o.p[0] = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, r.d0, 0x0C030C02u));  // pixels c0-2, c0-1
    o.p[1] = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, r.d1, 0x0C010C00u));  // c0, c0+1
    o.p[2] = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, r.d1, 0x0C030C02u));  // c0+2, c0+3
    o.p[3] = __builtin_bit_cast(u2, __builtin_amdgcn_perm(0u, r.d2, 0x0C010C00u));  // c0+4, c0+5
    return o;
  };
  // mag rows: 6 magnitudes (cols c0-1 .. c0+4) of image rows y-1, y, y+1; zero outside the image
  int mg[3][6];
  uint32_t cxy[2][4];  // (dx | dy << 16) of the lane's 4 pixels for the two newest mag rows
  Pairs4 ua = unpack(read_row(tile, wave * RPW, c0)), ub = unpack(read_row(tile, wave * RPW + 1, c0));
#pragma unroll
  for (int jr = 2; jr < RPW + 4; ++jr) {
    const Pairs4 uc = unpack(read_row(tile, wave * RPW + jr, c0));
    // magnitudes of image row ym (centre row of tile rows jr-2, jr-1, jr)
    const int ym = ty0 - 2 + wave * RPW + jr - 1;
#pragma unroll
    for (int k = 0; k < 6; ++k) mg[0][k] = mg[1][k], mg[1][k] = mg[2][k];
#pragma unroll
    for (int q = 0; q < 4; ++q) cxy[0][q] = cxy[1][q];
    {
      s2 S[4], D[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        S[k] = __builtin_bit_cast(s2, (u2)((ua.p[k] + uc.p[k]) * (unsigned short)3 + ub.p[k] * (unsigned short)10));
        D[k] = __builtin_bit_cast(s2, uc.p[k]) - __builtin_bit_cast(s2, ua.p[k]);
      }
      uint32_t xy[6];  // columns c0-1 .. c0+4
#pragma unroll
      for (int jp = 0; jp < 3; ++jp) {
        // pair jp = columns (c0-1+2jp, c0+2jp): dx[x] = S[x+1] - S[x-1], dy[x] = 3 (D[x-1] + D[x+1]) + 10 D[x]
        const s2 dxp = S[jp + 1] - S[jp];
        const s2 mid = __builtin_bit_cast(s2, __builtin_amdgcn_alignbit(__builtin_bit_cast(uint32_t, D[jp + 1]),
                                                                         __builtin_bit_cast(uint32_t, D[jp]), 16));
        const s2 dyp = (D[jp] + D[jp + 1]) * (short)3 + mid * (short)10;
        xy[2 * jp] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, dyp), __builtin_bit_cast(uint32_t, dxp), 0x05040100u);
        xy[2 * jp + 1] = __builtin_amdgcn_perm(__builtin_bit_cast(uint32_t, dyp), __builtin_bit_cast(uint32_t, dxp), 0x07060302u);
      }
      const bool row_in = ym >= 0 && ym < h;
#pragma unroll
      for (int k = 0; k < 6; ++k) {
        const int x = gx + k - 1;
        const int m = __builtin_amdgcn_sdot2(__builtin_bit_cast(s2, xy[k]), __builtin_bit_cast(s2, xy[k]), 0, false);
        mg[2][k] = (row_in && x >= 0 && x < w) ? m : 0;
        if (k >= 1 && k <= 4) cxy[1][k - 1] = xy[k];
      }
    }
    ua = ub;
    ub = uc;
    if (jr < 4) continue;
    // NMS of image row yo = ym - 1 (mag rows 0, 1, 2 = yo - 1, yo, yo + 1; gradients in cd*[0])
    const int yo = ym - 1;
    uint32_t wb = 0, sb = 0, cb0 = 0, cb1 = 0, cb2 = 0;
    if (yo < h) {
      // Every decision is a per-lane boolean = a wave mask in scalar registers: the compares write the masks,
      // the logic runs on the scalar unit, and a plane's bit enters its accumulator as the carry-in of ONE
      // v_addc (v = 2 v + bit).  No value selects: all four direction tests are evaluated and the gradient
      // direction picks the result.  (The select-chain version spent two thirds of its instructions on
      // v_cndmask / shift / or.)
#pragma unroll
      for (int q = 3; q >= 0; --q) {
        const int m = mg[1][q + 1];
        const int xs = (int)(int16_t)(cxy[0][q] & 0xFFFFu), ys = (int)cxy[0][q] >> 16;
        const int x = abs(xs), ay = abs(ys), y = ay << 15;
        const int tg22x = x * TG22;
        const int tg67x = tg22x + (x << 16);
        const uint64_t horiz = __ballot(y < tg22x);
        const uint64_t vert = __ballot(y > tg67x) & ~horiz;
        const bool neg_b = (xs ^ ys) < 0;  // diagonal: s = -1 where the signs differ
        const uint64_t neg = __ballot(neg_b);
        // > towards the previous pixel; >= towards the next one on the axes, > on the diagonals
        const uint64_t ok_h = __ballot(m > mg[1][q]) & __ballot(m >= mg[1][q + 2]);
        const uint64_t ok_v = __ballot(m > mg[0][q + 1]) & __ballot(m >= mg[2][q + 1]);
        const uint64_t ok_d = __ballot(m > mg[0][q]) & __ballot(m > mg[2][q + 2]);      // signs equal
        const uint64_t ok_n = __ballot(m > mg[0][q + 2]) & __ballot(m > mg[2][q]);      // signs differ
        const uint64_t is_max = (horiz & ok_h) | (vert & ok_v) | (~(horiz | vert) & ((neg & ok_n) | (~neg & ok_d)));
        const uint64_t cand = is_max & __ballot(m > low) & __ballot(gx + q < w);
        wb = shl1_or(wb, cand);
        sb = shl1_or(sb, cand & __ballot(m > high));
        // class 0: [0, pi/4), 1: [pi/4, pi/2), 2: [pi/2, 3pi/4), 3: [3pi/4, pi)
        const bool le_b = ay <= x, ge_b = ay >= x;
        const uint64_t le = __ballot(le_b), ge = __ballot(ge_b);
        cb0 = shl1_or(cb0, (neg & le) | (~neg & ge));
        cb1 = shl1_or(cb1, neg);
        // halves of a quarter: psi = atan(|dy| / |dx|) against pi/8 (psi < pi/4: tan = sqrt(2) - 1, i.e.
        // (|dx| + |dy|)^2 > 2 dx^2) or 3 pi/8 (tan = sqrt(2) + 1, i.e. (|dy| - |dx|)^2 > 2 dx^2); phi = psi or
        // pi - psi (signs differ), which mirrors the halves.  Irrational tangents: never an equality.
        const bool lowq = neg_b ? le_b : !ge_b;
        const int sq = lowq ? x + ay : ay - x;
        const uint64_t upper = __ballot(sq * sq > 2 * x * x);
        cb2 = shl1_or(cb2, neg ^ upper);
      }
    }
    // 8 lanes x 4 bits -> one 32-bit word
    wb = or_reduce8(wb << (4 * (lane & 7)));
    sb = or_reduce8(sb << (4 * (lane & 7)));
    if (d_class) {
      cb0 = or_reduce8((cb0 & in_row) << (4 * (lane & 7)));
      cb1 = or_reduce8((cb1 & in_row) << (4 * (lane & 7)));
      cb2 = or_reduce8((cb2 & in_row) << (4 * (lane & 7)));
    }
    if ((lane & 7) == 0 && yo < h && gx < w) {
      const int64_t bit0 = (int64_t)yo * w + gx;
      if ((bit0 & 31) == 0 && gx + 32 <= w) {  // the word belongs to this lane group alone
        weak[bit0 >> 5] = wb;
        strong[bit0 >> 5] = sb;
        if (d_class) {
          cls0[bit0 >> 5] = cb0;
          cls1[bit0 >> 5] = cb1;
          cls2[bit0 >> 5] = cb2;
        }
      } else {
        bits_or(weak, bit0, wb);
        bits_or(strong, bit0, sb);
        if (d_class) {
          bits_or(cls0, bit0, cb0);
          bits_or(cls1, bit0, cb1);
          bits_or(cls2, bit0, cb2);
        }
      }
    }
  }
}

// ---- K4: hysteresis sweep, bit-parallel ---------------------------------------------------------
constexpr int HW = TW / 32;  // interior words per tile row (8)
constexpr int HTH = 256;     // tile rows: taller than the stencil tiles -- growth crosses a tile border only
                             // once per sweep, and a sweep is a launch plus a host check

// A tile is worked on in sweep k + 1 only if a NEIGHBOUR asked for it in sweep k: growth in the neighbour reached a
// weak, not yet strong pixel of this tile (the neighbour sees it in its halo, next to one of its new bits).  A tile's
// own growth never asks for its own next turn -- it has reached its local fixed point, only new bits across a border
// can move it again.  (Until round 3 a tile was active whenever it or any of its 8 neighbours had changed, which in the
// sweeps after the first is nearly every tile: they cost 0.2 ms each at 64 planes where a few per cent of the tiles
// had anything to do.)  A request made from a stale view (the pixel was promoted meanwhile) costs one re-check.
__device__ __forceinline__ uint32_t lds_interior(const uint32_t (*a)[HW + 2], int r, int k) {
  return (r >= 1 && r <= HTH && k >= 1 && k <= HW) ? a[r][k] : 0u;
}

__global__ __launch_bounds__(NT) void k_hysteresis(const uint32_t* __restrict__ d_weak, uint32_t* __restrict__ d_strong,
                                                   int64_t words_per_plane, int h, int w,
                                                   uint32_t* __restrict__ d_changed,
                                                   const uint8_t* __restrict__ d_flags_in,
                                                   uint8_t* __restrict__ d_flags_out) {
  __shared__ uint32_t st[HTH + 2][HW + 2];
  __shared__ uint32_t wk[HTH + 2][HW + 2];  // weak map with halo; its interior is reused for the tile's new bits
  __shared__ uint32_t s_mark;
  const int plane = blockIdx.z;
  const int tx0 = blockIdx.x * TW, ty0 = blockIdx.y * HTH;
  const int ntx = gridDim.x, nty = gridDim.y;
  if (d_flags_in && !d_flags_in[(int64_t)plane * ntx * nty + blockIdx.y * ntx + blockIdx.x]) return;  // nobody asked
  const uint32_t* weak = d_weak + plane * words_per_plane;
  uint32_t* strong = d_strong + plane * words_per_plane;
  int pending = 0;
  if ((w & 31) == 0) {
    // rows of whole words (tiles start on word boundaries): a thread's ~10 words of each map are requested together,
    // then stored -- through row_word every one of them was two loads behind branches, i.e. a chain of round trips
    // (the seven sweeps at 64 planes of 4096^2: 0.86 -> 0.73 ms)
    constexpr int TOTAL = (HTH + 2) * (HW + 2), ITER = (TOTAL + NT - 1) / NT;
    uint32_t sv[ITER], wv[ITER];
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = min((int)threadIdx.x + it * NT, TOTAL - 1);
      const int r = i / (HW + 2), kk = i - r * (HW + 2);
      const int y = ty0 - 1 + r, x = tx0 - 32 + 32 * kk;
      const bool in = y >= 0 && y < h && x >= 0 && x < w;
      const int64_t wi = in ? ((int64_t)y * w + x) >> 5 : 0;
      sv[it] = strong[wi];
      wv[it] = weak[wi];
      if (!in) sv[it] = 0u, wv[it] = 0u;
    }
#pragma unroll
    for (int it = 0; it < ITER; ++it) {
      const int i = threadIdx.x + it * NT;
      if (i < TOTAL) {
        const int r = i / (HW + 2), kk = i - r * (HW + 2);
        st[r][kk] = sv[it];
        wk[r][kk] = wv[it];
        if (r >= 1 && r <= HTH && kk >= 1 && kk <= HW) pending |= (wv[it] & ~sv[it]) != 0;
      }
    }
  } else {
    for (int i = threadIdx.x; i < (HTH + 2) * (HW + 2); i += NT) {
      const int r = i / (HW + 2), kk = i - r * (HW + 2);
      const uint32_t sv = row_word(strong, h, w, ty0 - 1 + r, tx0 - 32 + 32 * kk);
      const uint32_t wv = row_word(weak, h, w, ty0 - 1 + r, tx0 - 32 + 32 * kk);
      st[r][kk] = sv;
      wk[r][kk] = wv;
      if (r >= 1 && r <= HTH && kk >= 1 && kk <= HW) pending |= (wv & ~sv) != 0;
    }
  }
  if (threadIdx.x == 0) s_mark = 0u;
  if (!__syncthreads_or(pending)) return;
  // A thread owns WPT vertically adjacent words (rows WPT g .. WPT g + WPT - 1 of word column k).  Per
  // iteration it grows each of them from the 3 x 3 neighbourhood and then along the word itself until
  // nothing moves (weak runs inside a word are absorbed at once), every row already seeing the update
  // of the row above: fewer block-wide iterations than one Jacobi dilation step per barrier.
  constexpr int WPT = HTH * HW / NT;  // words per thread
  static_assert(WPT * NT == HTH * HW, "the tile's words must divide evenly over the threads");
  const int k = (threadIdx.x & (HW - 1)) + 1, r0 = WPT * (threadIdx.x / HW) + 1;  // st coordinates
  uint32_t first[WPT];  // this thread's strong words before the sweep
#pragma unroll
  for (int j = 0; j < WPT; ++j) first[j] = st[r0 + j][k];
  int again;
  do {
    int changed = 0;
#pragma unroll
    for (int j = 0; j < WPT; ++j) {
      const int r = r0 + j;
      uint32_t cur = st[r][k];
      uint32_t cand = wk[r][k] & ~cur;
      if (!cand) continue;
      uint32_t dil = 0;
#pragma unroll
      for (int dr = -1; dr <= 1; ++dr) {
        const uint32_t c = st[r + dr][k], l = st[r + dr][k - 1], rt = st[r + dr][k + 1];
        dil |= c | (c << 1) | (c >> 1) | (l >> 31) | (rt << 31);
      }
      uint32_t nw = cand & dil;
      if (!nw) continue;
      cur |= nw;
      cand &= ~nw;
      for (uint32_t g = cand & ((cur << 1) | (cur >> 1)); g; g = cand & ((cur << 1) | (cur >> 1))) {
        cur |= g;
        cand &= ~g;
      }
      st[r][k] = cur;
      changed = 1;
    }
    again = __syncthreads_or(changed);
  } while (again);
  // new bits -> global memory and -> the interior of wk (the weak interior is not needed any more)
#pragma unroll
  for (int j = 0; j < WPT; ++j) {
    const uint32_t diff = st[r0 + j][k] & ~first[j];
    wk[r0 + j][k] = diff;
    if (diff) bits_or(strong, (int64_t)(ty0 + r0 + j - 1) * w + tx0 + 32 * (k - 1), diff);  // only in-image bits can be set
  }
  __syncthreads();
  // halo pixels that are weak, not strong (as far as this tile knows) and touch a NEW bit: their tile has work
  for (int i = threadIdx.x; i < 2 * (HW + 2) + 2 * HTH; i += NT) {
    int r, kk;
    if (i < HW + 2) r = 0, kk = i;
    else if (i < 2 * (HW + 2)) r = HTH + 1, kk = i - (HW + 2);
    else if (i < 2 * (HW + 2) + HTH) r = i - 2 * (HW + 2) + 1, kk = 0;
    else r = i - 2 * (HW + 2) - HTH + 1, kk = HW + 1;
    const uint32_t cand = wk[r][kk] & ~st[r][kk];
    if (!cand) continue;
    uint32_t dil = 0;
#pragma unroll
    for (int dr = -1; dr <= 1; ++dr) {
      const uint32_t c = lds_interior(wk, r + dr, kk), l = lds_interior(wk, r + dr, kk - 1), rt = lds_interior(wk, r + dr, kk + 1);
      dil |= c | (c << 1) | (c >> 1) | (l >> 31) | (rt << 31);
    }
    if (cand & dil) {
      const int dy = r == 0 ? -1 : (r == HTH + 1 ? 1 : 0), dx = kk == 0 ? -1 : (kk == HW + 1 ? 1 : 0);
      atomicOr(&s_mark, 1u << ((dy + 1) * 3 + dx + 1));
    }
  }
  __syncthreads();
  if (threadIdx.x == 0 && s_mark) {
    const uint32_t m = s_mark;
    bool any = false;
    for (int d = 0; d < 9; ++d) {
      if (!((m >> d) & 1u)) continue;
      const int nx = (int)blockIdx.x + d % 3 - 1, ny = (int)blockIdx.y + d / 3 - 1;
      if (nx < 0 || nx >= ntx || ny < 0 || ny >= nty) continue;
      any = true;
      if (d_flags_out) d_flags_out[(int64_t)plane * ntx * nty + ny * ntx + nx] = 1;
    }
    // a flag, not a count: a thousand tiles of a plane ask in the first sweep and their atomics would queue on the
    // plane's one counter (the look goes to L2: a stale L1 line would only cost a redundant store)
    if (any && __hip_atomic_load(&d_changed[plane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) d_changed[plane] = 1u;
  }
}

// ---- inspection: bitmap -> {0,1} bytes ------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_unpack_bits(const uint32_t* __restrict__ d_bits, int64_t words_per_plane,
                                                    int64_t npix, uint8_t* __restrict__ d_out) {
  const int plane = blockIdx.y;
  const uint32_t* bits = d_bits + plane * words_per_plane;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < npix; i += (int64_t)gridDim.x * NT)
    d_out[plane * npix + i] = (bits[i >> 5] >> (i & 31)) & 1u;
}

// bits_at without its branch around the second word (the bitmaps carry a spare word): a load behind a lane-level branch
// ends with a wait, and the loads of a wave's groups are meant to be in flight together
__device__ __forceinline__ uint32_t bits_at_nb(const uint32_t* __restrict__ bits, int64_t bit0, int n) {
  const int64_t wi = bit0 >> 5;
  const int sh = (int)(bit0 & 31);
  const uint64_t two = (uint64_t)bits[wi] | ((uint64_t)bits[wi + 1] << 32);
  const uint32_t v = (uint32_t)(two >> sh);
  return n >= 32 ? v : (v & ((1u << n) - 1u));
}

// ---- K5: grid_array from the bitmap: per-cell counts, scan, ordered coordinate fill -------------
__global__ __launch_bounds__(NT) void k_cell_count(const uint32_t* __restrict__ d_bits, int64_t words_per_plane, int h,
                                                   int w, int grid, int gc, int n_cells,
                                                   int32_t* __restrict__ d_counts, unsigned long long* __restrict__ d_state,
                                                   int state_words) {
  const int plane = blockIdx.y;
  const int cell = blockIdx.x * NT + threadIdx.x;
  // the chunk states and the ticket of the scan kernel that follows on the stream start at zero
  if (d_state && cell < state_words) d_state[(int64_t)plane * state_words + cell] = 0ull;
  if (cell >= n_cells) return;
  const uint32_t* bits = d_bits + plane * words_per_plane;
  const int cr = cell / gc, cc = cell - cr * gc;
  const int y0 = cr * grid, x0 = cc * grid;
  const int ch = min(grid, h - y0), cw = min(grid, w - x0);
  int cnt = 0;
#pragma unroll 5
  for (int r = 0; r < ch; ++r)  // (branch-free loads, five rows a trip: in flight together)
    for (int c0 = 0; c0 < cw; c0 += 32)
      cnt += __popc(bits_at_nb(bits, (int64_t)(y0 + r) * w + x0 + c0, min(32, cw - c0)));
  d_counts[(int64_t)plane * n_cells + cell] = cnt;
}

// d_num_edges[plane] = number of edge pixels -- or 0 when it exceeds `limit` (>= 0: the capacity of the coordinate
// list that the fill and every later kernel index with it; the true count goes to d_totals for the host to see).
__device__ __forceinline__ void put_edge_count(int plane, int total, int64_t limit, int32_t* d_num_edges, int32_t* d_totals) {
  if (d_totals) d_totals[plane] = total;
  d_num_edges[plane] = (limit >= 0 && total > limit) ? 0 : total;
}

__global__ __launch_bounds__(1024) void k_cell_scan(const int32_t* __restrict__ d_counts, int n_cells,
                                                    int32_t* __restrict__ d_starts, int32_t* __restrict__ d_num_edges,
                                                    int32_t* __restrict__ d_totals, int64_t limit) {
  const int plane = blockIdx.x;
  const int32_t* cnt = d_counts + (int64_t)plane * n_cells;
  int32_t* st = d_starts + (int64_t)plane * n_cells;
  // every thread owns a contiguous run of cells: run sums, one block-wide scan, run prefixes
  const int per = (n_cells + 1023) / 1024;
  const int lo = min((int)threadIdx.x * per, n_cells), hi = min(lo + per, n_cells);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += cnt[i];
  int total;
  int run = mg_block_exscan(sum, &total);
  for (int i = lo; i < hi; ++i) {
    st[i] = run;
    run += cnt[i];
  }
  if (threadIdx.x == 0) put_edge_count(plane, total, limit, d_num_edges, d_totals);
}

// The same scan with SCAN_CHUNK cells per workgroup (one 16-byte load per thread) and the chunk totals handed from
// workgroup to workgroup through d_state: a workgroup publishes (1 << 63 | total) as soon as it has summed its
// chunk and then adds up the totals of the chunks before it.  The chunk a workgroup takes is a TICKET drawn when it
// starts (not its blockIdx: HIP does not promise a dispatch order), so every total it waits for belongs to a workgroup
// that is already running and whose publication depends on nothing: the wait always ends.  The states and the ticket
// word (d_state[n_chunks] of the plane) are cleared by k_cell_count, the kernel in front of this one -- no launch
// counter in the arguments, so a captured graph can replay the launch.
// One plane: 59 us -> ~8 us; 64 planes use 4096 workgroups instead of 64.
constexpr int SCAN_CHUNK = 4096;

__global__ __launch_bounds__(1024) void k_cell_scan_chunks(const int32_t* __restrict__ d_counts, int n_cells,
                                                           int32_t* __restrict__ d_starts, int32_t* __restrict__ d_num_edges,
                                                           int32_t* __restrict__ d_totals, int64_t limit,
                                                           unsigned long long* __restrict__ d_state) {
  const int plane = blockIdx.y, n_chunks = gridDim.x;
  const int32_t* cnt = d_counts + (int64_t)plane * n_cells;
  int32_t* st = d_starts + (int64_t)plane * n_cells;
  unsigned long long* state = d_state + (int64_t)plane * (n_chunks + 1);
  __shared__ int s_chunk;
  if (threadIdx.x == 0) s_chunk = (int)atomicAdd(&state[n_chunks], 1ull);
  __syncthreads();
  const int chunk = s_chunk;
  const int i0 = chunk * SCAN_CHUNK + 4 * (int)threadIdx.x;
  int c[4] = {0, 0, 0, 0};
  const bool vec = (n_cells & 3) == 0 && i0 + 4 <= n_cells;  // n_cells % 4 == 0 keeps every plane 16-byte aligned
  if (vec) {
    const int4 v = *reinterpret_cast<const int4*>(cnt + i0);
    c[0] = v.x, c[1] = v.y, c[2] = v.z, c[3] = v.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (i0 + j < n_cells) c[j] = cnt[i0 + j];
  }
  int total;
  int run = mg_block_exscan(c[0] + c[1] + c[2] + c[3], &total);
  __shared__ int s_base;
  if (threadIdx.x == 0)
    __hip_atomic_store(&state[chunk], (1ull << 63) | (uint32_t)total, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (threadIdx.x < 64) {  // wave 0: totals of the chunks before this one
    int base = 0;
    for (int j = threadIdx.x; j < chunk; j += 64) {
      unsigned long long v;
      do {
        v = __hip_atomic_load(&state[j], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
      } while (!(v >> 63));
      base += (int)(uint32_t)v;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) base += __shfl_xor(base, off);
    if (threadIdx.x == 0) s_base = base;
  }
  __syncthreads();
  run += s_base;
  if (vec) {
    *reinterpret_cast<int4*>(st + i0) = make_int4(run, run + c[0], run + c[0] + c[1], run + c[0] + c[1] + c[2]);
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (i0 + j < n_cells) st[i0 + j] = run;
      run += c[j];
    }
  }
  if (chunk == n_chunks - 1 && threadIdx.x == 0) put_edge_count(plane, s_base + total, limit, d_num_edges, d_totals);
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
      uint32_t v = bits_at(bits, (int64_t)(y0 + r) * w + x0 + c0, min(32, cw - c0));
      while (v) {
        const int b = __ffs(v) - 1;
        v &= v - 1;
        if (pos < coord_cap) out[pos] = make_int2(y0 + r, x0 + c0 + b);
        ++pos;
      }
    }
}

// The same fill with one lane per cell ROW (64 / grid cells per wave): every lane loads its row's
// bits at once, a segmented wave scan over the rows of a cell gives each row its output position,
// and a lane then writes only its own row's few coordinates -- a short store run instead of a
// thread walking a whole cell (grid rows of dependent loads and ~70 stores).  grid <= 64.
constexpr int FILL_TRIPS = 4;  // groups of cells per wave: the bit rows and starts of all of them are requested before
                               // the first is worked on (a wave per group sat out two global round trips for ~200
                               // entries: 0.69 ms at C4; 3 / 4 / 6 / 8 groups: 0.58 / 0.57 / 0.67 / 0.76)

__global__ __launch_bounds__(NT) void k_cell_fill_rows(const uint32_t* __restrict__ d_bits, int64_t words_per_plane,
                                                       int h, int w, int grid, int gc, int n_cells,
                                                       const int32_t* __restrict__ d_starts,
                                                       int32_t* __restrict__ d_coords, int64_t coord_cap) {
  const int plane = blockIdx.y;
  const int cpw = 64 / grid;  // cells per group (a wave's lanes: one per cell row)
  const int lane = threadIdx.x & 63;
  const int64_t wave = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6;
  const int ci = lane / grid, r = lane - ci * grid;
  const uint32_t* bits = d_bits + plane * words_per_plane;
  uint64_t rowbits[FILL_TRIPS];
  int y[FILL_TRIPS], x0[FILL_TRIPS];
  int64_t pos[FILL_TRIPS];
#pragma unroll
  for (int q = 0; q < FILL_TRIPS; ++q) {
    const int64_t cell = (wave * FILL_TRIPS + q) * cpw + ci;
    rowbits[q] = 0, y[q] = 0, x0[q] = 0, pos[q] = 0;
    if (ci < cpw && cell < n_cells) {
      const int cr = (int)(cell / gc), cc = (int)(cell - (int64_t)cr * gc);
      y[q] = cr * grid + r;
      x0[q] = cc * grid;
      const int cw = min(grid, w - x0[q]);
      if (y[q] < h) {
        const int64_t b0 = (int64_t)y[q] * w + x0[q];
        rowbits[q] = bits_at_nb(bits, b0, min(32, cw));
        if (cw > 32) rowbits[q] |= (uint64_t)bits_at_nb(bits, b0 + 32, cw - 32) << 32;
      }
      pos[q] = d_starts[(int64_t)plane * n_cells + cell];
    }
  }
  // A group's cells are consecutive, so are their runs in the list: ONE contiguous run per group, starting at the
  // first cell's start.  A lane puts its row's coordinates at their places in an LDS copy of that run (wave-wide
  // exclusive scan of the row counts), then the wave stores the run with whole 512-byte instructions (before: every
  // lane stored its own few entries one after the other -- ~10 partly filled store instructions per wave).
  constexpr int CH = 512;  // entries staged per trip (a group of 17 % edge density has ~200; all pixels edges: 1200)
  __shared__ int2 s_stage[NT / 64][CH];
  int2* stage = s_stage[threadIdx.x >> 6];
  int2* out = reinterpret_cast<int2*>(d_coords + (int64_t)plane * coord_cap * 2);
#pragma unroll
  for (int q = 0; q < FILL_TRIPS; ++q) {
    const int cnt = __popcll(rowbits[q]);
    const int incl = mg_wave_scan_incl_i32(cnt);
    const int total = __shfl(incl, 63);
    const int lpos = incl - cnt;
    const int64_t base = __shfl(pos[q], 0);  // (lane 0: the group's first cell, row 0 -- the start of that cell)
    const bool any_cell = (wave * FILL_TRIPS + q) * cpw < n_cells;
    for (int c0 = 0; c0 < total; c0 += CH) {  // wave-uniform
      uint64_t rb = rowbits[q];
      int li = lpos - c0;
      while (rb) {
        const int b = __ffsll((unsigned long long)rb) - 1;
        rb &= rb - 1;
        if (li >= 0 && li < CH) stage[li] = make_int2(y[q], x0[q] + b);
        ++li;
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");  // (LDS operations of a wave complete in order)
      const int n = min(total - c0, CH);
      for (int i = lane; i < n; i += 64) {
        const int64_t p = base + c0 + i;
        if (any_cell && p < coord_cap) out[p] = stage[i];
      }
      __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    }
  }
}

// ---- K6: gradient angle at every edge pixel (thread per entry of the compact edge list) ---------
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
inline bool words_ok(int64_t words_per_plane, int h, int w) {
  return words_per_plane * 32 >= (((int64_t)h * w + 31) / 32) * 32 + 32;  // one spare word: bits_at reads wi + 1
}

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

extern "C" int64_t mg_scharr_hist_scratch_words(int n_planes, int h, int w, int mode) {
  if (n_planes < 0 || h < 0 || w < 0 || (mode != 0 && mode != 1)) return -1;
  const dim3 t = tile_grid(h, w, n_planes);
  const int64_t slots = (int64_t)t.x * ((t.y + HIST_TILES - 1) / HIST_TILES);
  return (int64_t)n_planes * slots * ((mode == 0 ? FINE + COARSE : FINE) / 2);
}

extern "C" int mg_scharr_hist(const uint8_t* d_blur, int n_planes, int h, int w, int mode, const uint32_t* d_base,
                              uint32_t* d_hist, uint32_t* d_scratch, int64_t scratch_words, void* stream) {
  if (!d_blur || !d_hist || n_planes < 0 || h < 0 || w < 0 || (mode != 0 && mode != 1)) return MG_EINVAL;
  if (mode == 1 && !d_base) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  const dim3 t = tile_grid(h, w, n_planes);
  const dim3 g(t.x, (t.y + HIST_TILES - 1) / HIST_TILES, t.z);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  if (d_scratch && scratch_words < mg_scharr_hist_scratch_words(n_planes, h, w, mode)) return MG_EINVAL;
  const int n_bins = mode == 0 ? FINE + COARSE : FINE;
  const size_t lds = (size_t)(TH + 2) * LS + (size_t)n_bins * 2;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_scharr_hist<false>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024) != hipSuccess)
      return MG_ELAUNCH;
    attr_set = true;
  }
  // with scratch: interior tile groups by the register-direct kernel into their slots, border groups
  // by the LDS-staged kernel straight into d_hist; without: the staged kernel does everything
  const int split = d_scratch ? 1 : 0;
  if (split) {
    hipLaunchKernelGGL(k_scharr_hist<true>, g, dim3(NT), (size_t)n_bins * 2, mg_stream(stream), d_blur, h, w, mode,
                       d_base, n_bins, d_hist, d_scratch, split);
    MG_CHECK_LAUNCH();
  }
  hipLaunchKernelGGL(k_scharr_hist<false>, t, dim3(NT), lds, mg_stream(stream), d_blur, h, w, mode, d_base, n_bins,
                     d_hist, d_scratch, split);
  MG_CHECK_LAUNCH();
  if (d_scratch) {
    const int slots = (int)(g.x * g.y);
    hipLaunchKernelGGL(k_hist_reduce, dim3((n_bins / 2 + NT - 1) / NT, n_planes, (slots + HIST_SLOTS_PER_BLOCK - 1) / HIST_SLOTS_PER_BLOCK),
                       dim3(NT), 0, mg_stream(stream), d_scratch, slots, n_bins, d_hist, mode == 1 ? d_base : nullptr);
    MG_CHECK_LAUNCH();
  }
  return MG_OK;
}

// to_uint8 + blur + combined histogram: one pass where the fused kernel applies (integer input, no un-blurred
// copy wanted, w % 4 == 0, aligned rows), else the two passes above.
static bool blur_hist_fused(const void* d_src, int dtype, int64_t plane_stride, int h, int w, int64_t row_stride,
                            const uint8_t* d_blur, const uint8_t* d_u8) {
  if (d_u8 || (dtype != MG_U8 && dtype != MG_U16) || h < 2 || w < 8 || (w & 3)) return false;
  const int64_t esz = dtype == MG_U8 ? 1 : 2;
  return (reinterpret_cast<uintptr_t>(d_src) % (4 * esz)) == 0 && plane_stride % 4 == 0 && row_stride % 4 == 0 &&
         (reinterpret_cast<uintptr_t>(d_blur) & 3) == 0;
}

extern "C" int64_t mg_blur_hist_scratch_words(int n_planes, int h, int w) {
  if (n_planes < 0 || h < 0 || w < 0) return -1;
  const int64_t fused = (int64_t)n_planes * ((w + FW - 1) / FW) * ((h + FH - 1) / FH) * ((FINE + COARSE) / 2);
  return std::max(fused, mg_scharr_hist_scratch_words(n_planes, h, w, 0));
}

extern "C" int mg_to_uint8_blur_hist(const void* d_src, int dtype, int n_planes, int64_t plane_stride, int h, int w,
                                     int64_t row_stride, const double* d_minmax, uint8_t* d_blur, uint8_t* d_u8,
                                     uint32_t* d_hist, uint32_t* d_scratch, int64_t scratch_words, void* stream) {
  if (!d_src || !d_blur || !d_hist || n_planes < 0 || h < 0 || w < 0) return MG_EINVAL;
  if (!d_minmax && dtype != MG_U8) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  if (d_scratch && scratch_words < mg_blur_hist_scratch_words(n_planes, h, w)) return MG_EINVAL;
  static const bool off = getenv("MG_NO_BLUR_HIST") != nullptr;
  if (off || !d_scratch || !blur_hist_fused(d_src, dtype, plane_stride, h, w, row_stride, d_blur, d_u8)) {
    const int rc = mg_to_uint8_blur(d_src, dtype, n_planes, plane_stride, h, w, row_stride, d_minmax, d_blur, d_u8, stream);
    if (rc != MG_OK) return rc;
    return mg_scharr_hist(d_blur, n_planes, h, w, 0, nullptr, d_hist, d_scratch, scratch_words, stream);
  }
  const dim3 g((w + FW - 1) / FW, (h + FH - 1) / FH, n_planes);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  hipStream_t s = mg_stream(stream);
  constexpr int n_bins = FINE + COARSE;
  if (dtype == MG_U8)
    hipLaunchKernelGGL((k_blur_hist<uint8_t>), g, dim3(NT), (size_t)n_bins * 2, s, (const uint8_t*)d_src, plane_stride, h, w,
                       row_stride, d_minmax, d_blur, d_scratch);
  else
    hipLaunchKernelGGL((k_blur_hist<uint16_t>), g, dim3(NT), (size_t)n_bins * 2, s, (const uint16_t*)d_src, plane_stride, h,
                       w, row_stride, d_minmax, d_blur, d_scratch);
  MG_CHECK_LAUNCH();
  const int slots = (int)(g.x * g.y);
  hipLaunchKernelGGL(k_hist_reduce, dim3((n_bins / 2 + NT - 1) / NT, n_planes, (slots + HIST_SLOTS_PER_BLOCK - 1) / HIST_SLOTS_PER_BLOCK),
                     dim3(NT), 0, s, d_scratch, slots, n_bins, d_hist, (const uint32_t*)nullptr);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_edge_thresholds(const uint32_t* d_hist, int n_planes, const int64_t* ranks4, float gamma_low,
                                  float gamma_high, int32_t* d_thresh, float* d_quantiles, int32_t* d_unresolved,
                                  int32_t* d_state, uint32_t* d_win_base, void* stream) {
  if (!d_hist || !ranks4 || !d_thresh || !d_quantiles || !d_unresolved || n_planes < 0) return MG_EINVAL;
  if ((d_state == nullptr) != (d_win_base == nullptr)) return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  hipLaunchKernelGGL(k_edge_thresholds, dim3(n_planes), dim3(NT), 0, mg_stream(stream), d_hist, FINE, FINE + COARSE,
                     (long long)ranks4[0], (long long)ranks4[1], (long long)ranks4[2], (long long)ranks4[3], gamma_low,
                     gamma_high, d_thresh, d_quantiles, d_unresolved, d_state, d_win_base);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_edge_thresholds_window(const uint32_t* d_hist_win, int n_planes, int pass, float gamma_low,
                                         float gamma_high, int32_t* d_state, uint32_t* d_win_base, int32_t* d_thresh,
                                         float* d_quantiles, int32_t* d_unresolved, void* stream) {
  if (!d_hist_win || !d_state || !d_win_base || !d_thresh || !d_quantiles || !d_unresolved || n_planes < 0 || pass < 0 ||
      pass > 3)
    return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  hipLaunchKernelGGL(k_window_resolve, dim3(n_planes), dim3(NT), 0, mg_stream(stream), d_hist_win, pass, gamma_low,
                     gamma_high, d_state, d_win_base, d_thresh, d_quantiles, d_unresolved);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_canny_nms(const uint8_t* d_blur, int n_planes, int h, int w, const int32_t* d_thresh,
                            uint32_t* d_weak, uint32_t* d_strong, uint32_t* d_class, int64_t words_per_plane,
                            void* stream) {
  if (!d_blur || !d_thresh || !d_weak || !d_strong || n_planes < 0 || h < 0 || w < 0) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  if (!words_ok(words_per_plane, h, w)) return MG_EINVAL;
  const dim3 g = tile_grid(h, w, n_planes);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  hipStream_t s = mg_stream(stream);
  // w % 32 == 0: every bitmap word inside the image belongs to one lane group, which stores it whole -- nothing to
  // clear (the spare words behind the image are never written: the caller zeroes the buffers once, when it makes them)
  const size_t bytes = (size_t)n_planes * words_per_plane * 4;
  if ((w & 31) && (mg_zero_async(d_weak, bytes, s) != hipSuccess || mg_zero_async(d_strong, bytes, s) != hipSuccess ||
                   (d_class && mg_zero_async(d_class, 3 * bytes, s) != hipSuccess)))
    return MG_ELAUNCH;
  hipLaunchKernelGGL(k_canny_nms, g, dim3(NT), 0, s, d_blur, h, w, d_thresh, words_per_plane, d_weak, d_strong,
                     d_class);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_canny_hysteresis(const uint32_t* d_weak, uint32_t* d_strong, int64_t words_per_plane, int n_planes,
                                   int h, int w, uint32_t* d_changed, const uint8_t* d_flags_in, uint8_t* d_flags_out,
                                   void* stream) {
  if (!d_weak || !d_strong || !d_changed || n_planes < 0 || h < 0 || w < 0) return MG_EINVAL;
  if (n_planes == 0 || h == 0 || w == 0) return MG_OK;
  if (!words_ok(words_per_plane, h, w)) return MG_EINVAL;
  const dim3 g((w + TW - 1) / TW, (h + HTH - 1) / HTH, n_planes);
  if (g.y > 65535 || g.z > 65535) return MG_EINVAL;
  hipLaunchKernelGGL(k_hysteresis, g, dim3(NT), 0, mg_stream(stream), d_weak, d_strong, words_per_plane, h, w,
                     d_changed, d_flags_in, d_flags_out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_hysteresis_tiles(int h, int w, int* tiles_x, int* tiles_y) {
  if (!tiles_x || !tiles_y) return MG_EINVAL;
  *tiles_x = (w + TW - 1) / TW;
  *tiles_y = (h + HTH - 1) / HTH;
  return MG_OK;
}

extern "C" int mg_unpack_bits(const uint32_t* d_bits, int64_t words_per_plane, int n_planes, int64_t n_bits,
                              uint8_t* d_out, void* stream) {
  if (!d_bits || !d_out || n_planes < 0 || n_planes > 65535 || n_bits < 0 || words_per_plane * 32 < n_bits)
    return MG_EINVAL;
  if (n_planes == 0 || n_bits == 0) return MG_OK;
  const int bx = (int)std::min<int64_t>((n_bits + NT - 1) / NT, 4096);
  hipLaunchKernelGGL(k_unpack_bits, dim3(bx, n_planes), dim3(NT), 0, mg_stream(stream), d_bits, words_per_plane, n_bits,
                     d_out);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int64_t mg_edge_grid_scan_words(int n_planes, int h, int w, int grid) {
  if (n_planes < 0 || h < 0 || w < 0 || grid <= 0) return -1;
  const int64_t n_cells = (int64_t)((h + grid - 1) / grid) * ((w + grid - 1) / grid);
  return (int64_t)n_planes * ((n_cells + SCAN_CHUNK - 1) / SCAN_CHUNK + 1);  // chunk states + the ticket word
}

extern "C" int mg_edge_grid(const uint32_t* d_edge_bits, int64_t words_per_plane, int n_planes, int h, int w, int grid,
                            int32_t* d_cell_counts, int32_t* d_cell_starts, int32_t* d_num_edges, int32_t* d_coords,
                            int64_t coord_cap, uint64_t* d_scan_state, int32_t* d_edge_totals, int phases, void* stream) {
  if (!d_edge_bits || !d_cell_counts || !d_cell_starts || !d_num_edges || n_planes < 0 || grid <= 0 || coord_cap < 0 ||
      n_planes > 65535 || phases < 1 || phases > 3 || ((phases & 2) && !d_coords))
    return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  if (!words_ok(words_per_plane, h, w)) return MG_EINVAL;
  const int gr = (h + grid - 1) / grid, gc = (w + grid - 1) / grid, n_cells = gr * gc;
  hipStream_t s = mg_stream(stream);
  if (phases & 1) {  // counts + scan (a caller without a capacity estimate sizes the coordinate list from d_num_edges)
    const int n_chunks = (n_cells + SCAN_CHUNK - 1) / SCAN_CHUNK;
    const bool chunked = d_scan_state && n_chunks > 1 && n_chunks <= 65535 && n_chunks + 1 <= n_cells;
    if (n_cells > 0) {
      hipLaunchKernelGGL(k_cell_count, dim3((n_cells + NT - 1) / NT, n_planes), dim3(NT), 0, s, d_edge_bits,
                         words_per_plane, h, w, grid, gc, n_cells, d_cell_counts,
                         chunked ? reinterpret_cast<unsigned long long*>(d_scan_state) : nullptr, n_chunks + 1);
      MG_CHECK_LAUNCH();
    }
    const int64_t limit = (phases & 2) ? coord_cap : -1;  // fill follows at once: the list's capacity bounds the count
    if (chunked) {
      hipLaunchKernelGGL(k_cell_scan_chunks, dim3(n_chunks, n_planes), dim3(1024), 0, s, d_cell_counts, n_cells,
                         d_cell_starts, d_num_edges, d_edge_totals, limit, reinterpret_cast<unsigned long long*>(d_scan_state));
    } else {
      hipLaunchKernelGGL(k_cell_scan, dim3(n_planes), dim3(1024), 0, s, d_cell_counts, n_cells, d_cell_starts, d_num_edges,
                         d_edge_totals, limit);
    }
    MG_CHECK_LAUNCH();
  }
  if ((phases & 2) && n_cells > 0) {  // ordered fill (entries beyond coord_cap are dropped; d_num_edges tells)
    if (grid <= 64) {
      const int64_t groups = (n_cells + 64 / grid - 1) / (64 / grid);
      const int64_t waves = (groups + FILL_TRIPS - 1) / FILL_TRIPS;
      hipLaunchKernelGGL(k_cell_fill_rows, dim3((unsigned)((waves + NT / 64 - 1) / (NT / 64)), n_planes), dim3(NT), 0, s,
                         d_edge_bits, words_per_plane, h, w, grid, gc, n_cells, d_cell_starts, d_coords, coord_cap);
    } else {
      hipLaunchKernelGGL(k_cell_fill, dim3((n_cells + NT - 1) / NT, n_planes), dim3(NT), 0, s, d_edge_bits,
                         words_per_plane, h, w, grid, gc, n_cells, d_cell_starts, d_coords, coord_cap);
    }
    MG_CHECK_LAUNCH();
  }
  return MG_OK;
}

extern "C" int mg_edge_angles(const uint8_t* d_blur, int n_planes, int h, int w, const int32_t* d_coords,
                              int64_t coord_cap, const int32_t* d_num_edges, float* d_angle, void* stream) {
  if (!d_blur || !d_coords || !d_num_edges || !d_angle || n_planes < 0 || n_planes > 65535 || coord_cap < 0)
    return MG_EINVAL;
  if (n_planes == 0 || coord_cap == 0) return MG_OK;
  const int bx = (int)std::max<int64_t>(1, std::min<int64_t>((coord_cap + NT - 1) / NT, 1 << 20));  // one edge per thread
  hipLaunchKernelGGL(k_edge_angles, dim3(bx, n_planes), dim3(NT), 0, mg_stream(stream), d_blur, h, w, d_coords,
                     coord_cap, d_num_edges, d_angle);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
