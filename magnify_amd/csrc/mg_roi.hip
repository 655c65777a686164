// A12-A15, A18: ownership labels (utils.py:380-395), ROI windows (utils.py:60-80), fg/bg masks
// and ROI gather (find.py:561-602), masked reductions (README.md:21-22, identify.py:76-80).
//
// Roofline: HBM.  Per marker: L*L*(4 label read + 2 mask write) + C*T*L*L*(2 read + 2 write)
// bytes (u16); the reductions ride along in registers (wavefront shuffles), 0 extra bytes.
#include <math.h>

#include <stdlib.h>

#include "mg_common.h"

namespace {

constexpr int NT = 256;

// ---- circle_labels as a coverage count -------------------------------------------------------
__global__ __launch_bounds__(NT) void k_circle_labels(const int32_t* __restrict__ d_beads, int64_t bead_cap,
                                                      const int32_t* __restrict__ d_num_beads, int h, int w,
                                                      const int32_t* __restrict__ d_halfwidths, int max_r,
                                                      int32_t* __restrict__ d_labels, int reset) {
  const int plane = blockIdx.y;
  const int i = blockIdx.x;
  if (i >= d_num_beads[plane]) return;
  const int32_t* b = d_beads + ((int64_t)plane * bead_cap + i) * 3;
  const int row = b[0], col = b[1], r = b[2];
  if (r < 2 || r > max_r) return;  // undefined in the reference (utils.py:398-430 indexes out of bounds)
  const int32_t* hw = d_halfwidths + (int64_t)r * (2 * max_r + 1);
  int32_t* lab = d_labels + (int64_t)plane * h * w;
  const int side = 2 * r + 1;
  for (int p = threadIdx.x; p < side * side; p += NT) {
    const int dy = p / side - r, dx = p % side - r;
    if (abs(dx) > hw[dy + r]) continue;
    const int y = row + dy, x = col + dx;
    if (y < 0 || y >= h || x < 0 || x >= w) continue;
    int32_t* cell = &lab[(int64_t)y * w + x];
    if (reset) {  // restore the "nobody" value under this disk (lets the caller reuse the map)
      *cell = -1;
      continue;
    }
    const int old = atomicCAS(cell, -1, i);
    if (old != -1 && old != i) *cell = -2;  // a second owner: contested
  }
}

// ---- ROI gather + masks + sums ------------------------------------------------------------------
__device__ __forceinline__ void window(int c, int len, int size, int& lo) {
  // utils.py:64-79 with an integer centre.  Whatever the table holds, the window stays inside the image: a centre
  // beyond +-2^28 (nothing an image can hold; what a NaN turns into when it is cast) is pulled in before the sums below
  // could wrap.
  c = min(max(c, -(1 << 28)), 1 << 28);
  int a = c - len / 2, b = c + (len - len / 2);
  if (a < 0) {
    b -= a;
    a = 0;
  }
  if (b > size) a -= b - size;
  lo = a;
}

// fg/bg segmentation straight from the bead table, without the label map: a window pixel is
//   foreground  <=> it lies in this marker's disk and in no other disk   (labels == i,  find.py:580)
//   background  <=> it lies in no disk at all                            (labels == -1, find.py:582)
// which is what circle_labels' "exactly one owner / contested" rule (utils.py:380-395) yields.
// Row bit masks of the window in LDS: any (covered), multi (covered twice or more), own.
struct DiskMasks {
  uint32_t* any;
  uint32_t* multi;
  uint32_t* own;
  int wpr;  // words per window row
  __device__ __forceinline__ void flags(int ry, int rx, uint32_t& f, uint32_t& b) const {
    const int i = ry * wpr + (rx >> 5), sh = rx & 31;
    f = ((own[i] & ~multi[i]) >> sh) & 1u;
    b = (~any[i] >> sh) & 1u;
  }
};

// OR one row span of a disk into the masks (LDS atomics; any thread, any order).
__device__ __forceinline__ void mask_row(const DiskMasks& m, int ry, int xa, int xb, bool is_own) {
  for (int wd = xa >> 5; wd <= (xb >> 5); ++wd) {
    const int lo = max(xa, 32 * wd) - 32 * wd, hi = min(xb, 32 * wd + 31) - 32 * wd;
    const uint32_t bits = (hi - lo == 31) ? 0xFFFFFFFFu : (((1u << (hi - lo + 1)) - 1u) << lo);
    const uint32_t old = atomicOr(&m.any[ry * m.wpr + wd], bits);
    if (old & bits) atomicOr(&m.multi[ry * m.wpr + wd], old & bits);
    if (is_own) atomicOr(&m.own[ry * m.wpr + wd], bits);
  }
}
// Row ry of the window under the disk (yj, xj, rj), if it intersects the window.
__device__ __forceinline__ void mask_disk_row(const DiskMasks& m, int ry, int len, int top, int left, int yj, int xj,
                                              int rj, bool is_own, const int32_t* __restrict__ hwtab, int max_r) {
  const int dy = top + ry - yj;
  if (dy < -rj || dy > rj) return;
  const int hwid = hwtab[(int64_t)rj * (2 * max_r + 1) + dy + rj];
  if (hwid < 0) return;
  const int xa = max(xj - hwid, left) - left, xb = min(xj + hwid, left + len - 1) - left;
  if (xa <= xb) mask_row(m, ry, xa, xb, is_own);
}

constexpr int NBR_CAP = 96;  // disks overlapping one window that are handled row-parallel

__device__ __forceinline__ DiskMasks build_disk_masks(uint32_t* base, int len, int top, int left,
                                                      const int32_t* __restrict__ beads, int nb, int local,
                                                      const int32_t* __restrict__ hwtab, int max_r) {
  __shared__ int s_nn;
  __shared__ int s_nbr[NBR_CAP][3];
  DiskMasks m;
  m.wpr = (len + 31) >> 5;
  const int words = len * m.wpr;
  m.any = base;
  m.multi = base + words;
  m.own = base + 2 * words;
  for (int i = threadIdx.x; i < 3 * words; i += NT) base[i] = 0u;
  if (threadIdx.x == 0) s_nn = 0;
  __syncthreads();
  // 1. scan the assay's bead table for disks that reach into the window (loads issued in batches:
  //    one memory latency per 8 * NT beads); collect them in LDS
  constexpr int UB = 8;
  for (int b0 = 0; b0 < nb; b0 += NT * UB) {
    int yy[UB], xx[UB], rr[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = b0 + u * NT + (int)threadIdx.x;
      rr[u] = 0;
      if (j < nb) {
        yy[u] = beads[3 * j];
        xx[u] = beads[3 * j + 1];
        rr[u] = beads[3 * j + 2];
      }
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = b0 + u * NT + (int)threadIdx.x;
      const int yj = yy[u], xj = xx[u], rj = rr[u];
      if (rj < 2 || rj > max_r) continue;  // undefined in the reference, no coverage (as k_circle_labels)
      if (yj + rj < top || yj - rj >= top + len || xj + rj < left || xj - rj >= left + len) continue;
      const int k = atomicAdd(&s_nn, 1);
      if (k < NBR_CAP) {
        s_nbr[k][0] = yj;
        s_nbr[k][1] = xj;
        s_nbr[k][2] = rj | (j == local ? 0x10000 : 0);
      } else {  // an extremely crowded window: this thread draws the whole disk itself
        for (int ry = max(yj - rj, top) - top; ry <= min(yj + rj, top + len - 1) - top; ++ry)
          mask_disk_row(m, ry, len, top, left, yj, xj, rj, j == local, hwtab, max_r);
      }
    }
  }
  __syncthreads();
  // 2. one (disk, row) pair per thread
  const int nn = min(s_nn, NBR_CAP), side = 2 * max_r + 1;
  for (int p = threadIdx.x; p < nn * side; p += NT) {
    const int k = p / side, dyi = p - k * side - max_r;
    const int yj = s_nbr[k][0], xj = s_nbr[k][1], rj = s_nbr[k][2] & 0xFFFF;
    const int ry = yj + dyi - top;
    if (dyi < -rj || dyi > rj || ry < 0 || ry >= len) continue;
    mask_disk_row(m, ry, len, top, left, yj, xj, rj, (s_nbr[k][2] & 0x10000) != 0, hwtab, max_r);
  }
  __syncthreads();
  return m;
}

inline __host__ __device__ int mask_words(int len) { return 3 * len * ((len + 31) >> 5); }
inline size_t roi_lds_bytes(int len, bool disks) {
  const size_t fl = ((size_t)len * len + 3) & ~(size_t)3;
  return fl + (disks ? (size_t)3 * len * ((len + 31) >> 5) * 4 : 0);
}

template <typename T, typename ACC>
__global__ __launch_bounds__(NT) void k_roi(const T* __restrict__ d_image, int64_t assay_stride, int n_c, int n_t, int h,
                                            int w, const int32_t* __restrict__ d_beads,
                                            const int32_t* __restrict__ d_marker_assay,
                                            const int32_t* __restrict__ d_marker_local, int len,
                                            const int32_t* __restrict__ d_labels,
                                            const int32_t* __restrict__ d_assay_offsets, int n_assays, int64_t bead_stride, int time_major,
                                            const int32_t* __restrict__ d_order,
                                            const int32_t* __restrict__ d_halfwidths, int max_r, T* __restrict__ d_roi,
                                            uint8_t* __restrict__ d_fg, uint8_t* __restrict__ d_bg,
                                            double* __restrict__ d_sums, int32_t* __restrict__ d_counts) {
  extern __shared__ __attribute__((aligned(4))) uint8_t flags[];
  __shared__ ACC s_red[2][NT / 64];
  __shared__ int s_cnt[2][NT / 64];
  // one block per marker: of a flat list with per-marker assay / local index (label mode), or of the assays'
  // concatenated bead tables (disk mode)
  int g = blockIdx.x, assay, first = 0, local;
  int64_t bead0 = 0, gb = g;  // the assay's first bead / this marker's bead in d_beads
  if (d_assay_offsets) {
    // flat grid over the markers of all assays (the launch may be sized by an upper bound: the rest leaves at once);
    // the marker's assay = the last one that starts at or before it
    if (g >= d_assay_offsets[n_assays]) return;
    if (d_order) g = d_order[g];  // the order the windows are visited in (mg_roi_window_order); outputs stay in place
    int lo = 0, hi = n_assays;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (d_assay_offsets[mid] <= g) lo = mid;
      else hi = mid;
    }
    assay = lo;
    first = d_assay_offsets[assay];
    local = g - first;
    // bead table: compact (markers and beads share the index) or one padded row per assay
    bead0 = bead_stride ? (int64_t)assay * bead_stride : first;
    gb = bead0 + local;
  } else {
    assay = d_marker_assay ? d_marker_assay[g] : 0;
    local = d_marker_local ? d_marker_local[g] : g;
  }
  const int cy = d_beads[3 * gb], cx = d_beads[3 * gb + 1];
  int top, left;
  window(cy, len, h, top);
  window(cx, len, w, left);
  const int n = len * len;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // masks from the label map (find.py:580-584) or straight from the assay's bead table
  const int32_t* lab = d_labels ? d_labels + (int64_t)assay * h * w : nullptr;
  DiskMasks dm{};
  if (d_halfwidths)
    dm = build_disk_masks(reinterpret_cast<uint32_t*>(flags + ((n + 3) & ~3)), len, top, left,
                          d_beads + 3 * bead0, d_assay_offsets[assay + 1] - first, local, d_halfwidths, max_r);
  int cf = 0, cb = 0;
  for (int p = threadIdx.x; p < n; p += NT) {
    const int ry = p / len, rx = p - ry * len;
    uint8_t f = 0, b = 0;
    if (d_halfwidths) {
      uint32_t ff, bb;
      dm.flags(ry, rx, ff, bb);
      f = (uint8_t)ff;
      b = (uint8_t)bb;
    } else if (lab) {
      const int v = lab[(int64_t)(top + ry) * w + (left + rx)];
      f = v == local;
      b = v == -1;
    }
    flags[p] = f | (b << 1);
    if (d_fg) d_fg[(int64_t)g * n + p] = f;
    if (d_bg) d_bg[(int64_t)g * n + p] = b;
    cf += f;
    cb += b;
  }
  cf = mg_wave_sum_i32(cf);
  cb = mg_wave_sum_i32(cb);
  if (lane == 0) {
    s_cnt[0][wave] = cf;
    s_cnt[1][wave] = cb;
  }
  __syncthreads();
  if (threadIdx.x == 0 && d_counts) {
    d_counts[2 * (int64_t)g] = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
    d_counts[2 * (int64_t)g + 1] = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
  }
  // gather every (channel, time) window (find.py:589-602) and reduce under the masks
  const T* img = d_image + (int64_t)assay * assay_stride;
  for (int ct = 0; ct < n_c * n_t; ++ct) {
    // outputs are (channel, time)-ordered; the image block may be stored time-major (t, c, h, w)
    const T* plane = img + (int64_t)(time_major ? (ct % n_t) * n_c + ct / n_t : ct) * h * w;
    T* out = d_roi ? d_roi + ((int64_t)g * n_c * n_t + ct) * n : nullptr;
    ACC sf = 0, sb = 0;
    for (int p = threadIdx.x; p < n; p += NT) {
      const int ry = p / len, rx = p - ry * len;
      const T v = plane[(int64_t)(top + ry) * w + (left + rx)];
      if (out) out[p] = v;
      const uint8_t fl = flags[p];
      if (fl & 1) sf += (ACC)v;
      if (fl & 2) sb += (ACC)v;
    }
    if (d_sums) {
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        sf += __shfl_xor(sf, off);
        sb += __shfl_xor(sb, off);
      }
      __syncthreads();
      if (lane == 0) {
        s_red[0][wave] = sf;
        s_red[1][wave] = sb;
      }
      __syncthreads();
      if (threadIdx.x == 0) {
        double* o = d_sums + ((int64_t)g * n_c * n_t + ct) * 2;
        o[0] = (double)(s_red[0][0] + s_red[0][1] + s_red[0][2] + s_red[0][3]);
        o[1] = (double)(s_red[1][0] + s_red[1][1] + s_red[1][2] + s_red[1][3]);
      }
    }
  }
}

// ---- fast path: uint16 image, even window length <= 126, even image width -------------------------
// One wave per window row, one dword (2 pixels) per lane: aligned 4-byte loads (odd source offsets are funnel-shifted
// from the neighbouring lane's dword, fetched by a whole-wave DPP shift), 4-byte roi stores, 2-byte mask stores.
// The masks live in LDS as two BIT rows per window row (fg, bg); a lane turns its two bits into the 0/1 halves of a
// packed-u16 multiplier once per row and applies it to CTB (channel, time) planes at a time: one v_dot2_u32_u16 per
// dword and mask.  All sums fit 32 bits (126^2 pixels x 65535 < 2^32) up to the final conversion.
// (Before: byte flags in LDS and four conditional 64-bit adds per dword -- 53 wave-instructions per row and plane,
// the VALU 85 % busy at the HBM ceiling.)
typedef unsigned short roi_us2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t roi_dot2(uint32_t v, uint32_t m, uint32_t acc) {
  return __builtin_amdgcn_udot2(__builtin_bit_cast(roi_us2, v), __builtin_bit_cast(roi_us2, m), acc, false);
}
// bits 0, 1 of b -> the 0/1 halves of a packed pair
__device__ __forceinline__ uint32_t roi_pair(uint32_t b) { return (b & 1u) | ((b & 2u) << 15); }

template <int U, int CTB, bool PIPE>
__global__ __launch_bounds__(NT) void k_roi_u16_even(const uint16_t* __restrict__ d_image, int64_t assay_stride,
                                                     int n_c, int n_t, int h, int w,
                                                     const int32_t* __restrict__ d_beads,
                                                     const int32_t* __restrict__ d_marker_assay,
                                                     const int32_t* __restrict__ d_marker_local, int len,
                                                     const int32_t* __restrict__ d_labels,
                                                     const int32_t* __restrict__ d_assay_offsets, int n_assays, int64_t bead_stride, int time_major,
                                                     const int32_t* __restrict__ d_order,
                                                     const int32_t* __restrict__ d_halfwidths, int max_r,
                                                     uint16_t* __restrict__ d_roi, uint8_t* __restrict__ d_fg,
                                                     uint8_t* __restrict__ d_bg, double* __restrict__ d_sums,
                                                     int32_t* __restrict__ d_counts) {
  extern __shared__ __attribute__((aligned(4))) uint8_t smem[];  // three bit-row arrays of len x wpr words
  constexpr int WV = NT / 64;
  __shared__ uint32_t s_red[2][CTB][WV];
  __shared__ int s_cnt[2][WV];
  // one block per marker: of a flat list with per-marker assay / local index (label mode), or of the assays'
  // concatenated bead tables (disk mode)
  int g = blockIdx.x, assay, first = 0, local;
  int64_t bead0 = 0, gb = g;  // the assay's first bead / this marker's bead in d_beads
  if (d_assay_offsets) {
    // flat grid over the markers of all assays (the launch may be sized by an upper bound: the rest leaves at once);
    // the marker's assay = the last one that starts at or before it
    if (g >= d_assay_offsets[n_assays]) return;
    if (d_order) g = d_order[g];  // the order the windows are visited in (mg_roi_window_order); outputs stay in place
    int lo = 0, hi = n_assays;
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (d_assay_offsets[mid] <= g) lo = mid;
      else hi = mid;
    }
    assay = lo;
    first = d_assay_offsets[assay];
    local = g - first;
    // bead table: compact (markers and beads share the index) or one padded row per assay
    bead0 = bead_stride ? (int64_t)assay * bead_stride : first;
    gb = bead0 + local;
  } else {
    assay = d_marker_assay ? d_marker_assay[g] : 0;
    local = d_marker_local ? d_marker_local[g] : g;
  }
  int top, left;
  window(d_beads[3 * gb], len, h, top);
  window(d_beads[3 * gb + 1], len, w, left);
  const int n = len * len, half = len >> 1;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = 2 * lane;
  const bool act = lane < half;
  const int wpr = (len + 31) >> 5, words = len * wpr;
  uint32_t* base = reinterpret_cast<uint32_t*>(smem);
  uint32_t* fgw = base + 2 * words;  // (disk mode: `own`, narrowed in place)
  uint32_t* bgw = base;              // (disk mode: `any`, complemented in place)
  int cf = 0, cb = 0;
  if (d_halfwidths) {
    const DiskMasks dm = build_disk_masks(base, len, top, left, d_beads + 3 * bead0, d_assay_offsets[assay + 1] - first,
                                          local, d_halfwidths, max_r);
    // fg = own and not contested, bg = covered by nobody (the bits of a row's last word beyond the window stay 0)
    for (int i = threadIdx.x; i < words; i += NT) {
      const int wd = i % wpr;
      const uint32_t valid = (len - 32 * wd >= 32) ? 0xFFFFFFFFu : ((1u << (len - 32 * wd)) - 1u);
      const uint32_t f = dm.own[i] & ~dm.multi[i], b = ~dm.any[i] & valid;
      fgw[i] = f;
      bgw[i] = b;
      cf += __popc(f);
      cb += __popc(b);
    }
  } else {
    for (int i = threadIdx.x; i < 3 * words; i += NT) base[i] = 0u;
    __syncthreads();
    if (d_labels) {
      const int32_t* lab = d_labels + (int64_t)assay * h * w;
      for (int ry = wave; ry < len; ry += WV) {
        if (!act) continue;
        const int32_t* lp = lab + (int64_t)(top + ry) * w + left + x;
        const int v0 = lp[0], v1 = lp[1];
        const uint32_t f = (uint32_t)(v0 == local) | ((uint32_t)(v1 == local) << 1);
        const uint32_t b = (uint32_t)(v0 == -1) | ((uint32_t)(v1 == -1) << 1);
        if (f) atomicOr(&fgw[ry * wpr + (x >> 5)], f << (x & 31));
        if (b) atomicOr(&bgw[ry * wpr + (x >> 5)], b << (x & 31));
        cf += __popc(f);
        cb += __popc(b);
      }
    }
  }
  cf = mg_wave_sum_i32(cf);
  cb = mg_wave_sum_i32(cb);
  if (lane == 0) {
    s_cnt[0][wave] = cf;
    s_cnt[1][wave] = cb;
  }
  __syncthreads();  // bit rows complete, counts in place
  if (threadIdx.x == 0 && d_counts) {
    d_counts[2 * (int64_t)g] = s_cnt[0][0] + s_cnt[0][1] + s_cnt[0][2] + s_cnt[0][3];
    d_counts[2 * (int64_t)g + 1] = s_cnt[1][0] + s_cnt[1][1] + s_cnt[1][2] + s_cnt[1][3];
  }
  const int mword = x >> 5, msh = x & 31;  // this lane's two bits inside a bit row
  if (d_fg || d_bg) {
    for (int ry = wave; ry < len; ry += WV) {
      if (!act) continue;
      const uint32_t f = (fgw[ry * wpr + mword] >> msh) & 3u, b = (bgw[ry * wpr + mword] >> msh) & 3u;
      if (d_fg) *reinterpret_cast<uint16_t*>(&d_fg[(int64_t)g * n + ry * len + x]) = (uint16_t)((f & 1u) | ((f & 2u) << 7));
      if (d_bg) *reinterpret_cast<uint16_t*>(&d_bg[(int64_t)g * n + ry * len + x]) = (uint16_t)((b & 1u) | ((b & 2u) << 7));
    }
  }
  const uint16_t* img = d_image + (int64_t)assay * assay_stride;
  const int nct = n_c * n_t;
  // w is even: every row of the window starts at the same parity.  An odd start is read from one element before
  // (aligned dwords) and funnel-shifted; its last dword then ends one element behind the window -- inside the image
  // row, because left + len == w would make left even.
  const int odd = left & 1;
  const uint32_t shift = odd ? 16u : 0u;
  const uint32_t lidx = (uint32_t)min(lane, half - 1 + odd);  // idle lanes repeat the last dword: no branch around a load
  const int64_t plane_elems = (int64_t)h * w;
  for (int ct0 = 0; ct0 < nct; ct0 += CTB) {
    // outputs are (channel, time)-ordered; the image block may be stored time-major (t, c, h, w)
    const uint32_t* plane[CTB];
    uint32_t* out[CTB];
#pragma unroll
    for (int c = 0; c < CTB; ++c) {
      const int ct = min(ct0 + c, nct - 1);
      plane[c] = reinterpret_cast<const uint32_t*>(img + (int64_t)(time_major ? (ct % n_t) * n_c + ct / n_t : ct) * plane_elems);
      out[c] = d_roi ? reinterpret_cast<uint32_t*>(d_roi + ((int64_t)g * nct + ct) * n) : nullptr;
    }
    uint32_t sf[CTB], sb[CTB];
#pragma unroll
    for (int c = 0; c < CTB; ++c) sf[c] = 0u, sb[c] = 0u;
    // U rows x CTB planes per trip and wave, software-pipelined: the loads of the NEXT trip are issued before this
    // trip's dwords are shifted, stored and summed (the gather is latency-bound otherwise: a wave would sit out a
    // full load round trip, and the write acknowledgements of its stores, between two batches of requests)
    auto load_rows = [&](int r0, uint32_t (&dd)[U][CTB]) {
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ry = min(r0 + u * WV, len - 1);
        const uint32_t di = (uint32_t)(((top + ry) * w + left - odd) >> 1) + lidx;  // dword index in the plane (h w < 2^31)
#pragma unroll
        for (int c = 0; c < CTB; ++c) dd[u][c] = plane[c][di];
      }
    };
    uint32_t dd[U][CTB], dn[U][CTB];
    load_rows(wave, dd);
    for (int r0 = wave; r0 < len; r0 += WV * U) {
      if (PIPE) {
        load_rows(r0 + WV * U, dn);  // (rows beyond the window repeat its last row: no branch)
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const int ry = r0 + u * WV;
        if (ry >= len) break;  // wave-uniform
        const uint32_t mf = roi_pair((fgw[ry * wpr + mword] >> msh) & 3u), mb = roi_pair((bgw[ry * wpr + mword] >> msh) & 3u);
        const uint32_t oi = (uint32_t)(ry * half + lane);
#pragma unroll
        for (int c = 0; c < CTB; ++c) {
          if (ct0 + c >= nct) break;  // uniform
          const uint32_t d = dd[u][c];
          const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x130, 0xF, 0xF, false);  // lane + 1
          const uint32_t v = __builtin_amdgcn_alignbit(nx, d, shift);
          if (act) {
            if (out[c]) out[c][oi] = v;
            sf[c] = roi_dot2(v, mf, sf[c]);
            sb[c] = roi_dot2(v, mb, sb[c]);
          }
        }
      }
      if (PIPE) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
          for (int c = 0; c < CTB; ++c) dd[u][c] = dn[u][c];
      } else if (r0 + WV * U < len) {
        load_rows(r0 + WV * U, dd);
      }
    }
    if (d_sums) {
#pragma unroll
      for (int c = 0; c < CTB; ++c) {
        const uint32_t a = (uint32_t)mg_wave_scan_incl_i32((int)sf[c]), b = (uint32_t)mg_wave_scan_incl_i32((int)sb[c]);
        if (lane == 63) {
          s_red[0][c][wave] = a;
          s_red[1][c][wave] = b;
        }
      }
      __syncthreads();
      if ((int)threadIdx.x < 2 * CTB) {
        const int k = threadIdx.x & 1, c = threadIdx.x >> 1;
        if (ct0 + c < nct) {
          uint32_t tot = 0;
#pragma unroll
          for (int q = 0; q < WV; ++q) tot += s_red[k][c][q];
          d_sums[((int64_t)g * nct + ct0 + c) * 2 + k] = (double)tot;
        }
      }
      __syncthreads();
    }
  }
}

// ---- image-centric ROI pass (round 4): every image line is fetched once ----------------------------------------
// The window-centric kernel above reads 200-byte window rows at arbitrary alignment: a row touches 2.56 lines of 128 B
// (measured 15.7 GB fetched for 9.7 GB of window pixels at C4), and pixels shared by overlapping windows are fetched
// again by each of them -- an L2 of 4 MB per XCD does not keep a line for the ~20 us until a neighbouring window of
// another workgroup comes by.  Here a workgroup owns a TILE of RT_H x RT_W pixels of one assay: it loads the tile's
// (channel, time) planes -- RT_CT at a time -- into LDS with full, aligned lines, finds the windows that reach into the
// tile (a scan of the assay's bead table) and serves every one of them its FRAGMENT from LDS: roi pixels and mask
// bytes to their places in the marker's outputs, masked sums and counts by atomic adds (integer sums below 2^53 are
// exact in float64 whatever the order).  Masks as in the window kernel: fg = own disk and no other disk, bg = no disk
// (utils.py:380-395 / find.py:571-586), from tile-wide any / multi bit maps.
constexpr int RT_H = 16, RT_W = 384, RT_WPR = RT_W / 32;  // tile rows / pixels / words per bit row
constexpr int RT_CT = 4;                                  // planes in LDS at a time (RT_CT * RT_H * RT_W * 2 B = 48 KB)
constexpr int RT_F = 16;                                  // fragments per round (their mask rows are held in LDS)
constexpr int RT_IDS = 2048;                              // beads per block of the window scan (one block unless an assay holds more)
// (LDS per workgroup: 48 + 1.5 + 8 + 4 KB and ~2.5 KB of descriptors: two workgroups per CU -- one loads while the
// other serves)
constexpr int RTN = 512;                                  // threads of a tile workgroup (8 waves; two workgroups per CU)
constexpr int RT_DISKS = 128;                             // disks reaching into one tile that are drawn row-parallel

struct RtFrag {
  int g;             // marker (row of the outputs)
  int top, left;     // window origin in the image
  int yj, xj, rj;    // its own disk
  int r0, r1;        // tile rows [r0, r1) the window covers
};

__global__ __launch_bounds__(RTN) void k_roi_tiles_u16(const uint16_t* __restrict__ d_image, int64_t assay_stride, int n_c,
                                                      int n_t, int h, int w, const int32_t* __restrict__ d_beads,
                                                      int64_t bead_stride, const int32_t* __restrict__ d_assay_offsets,
                                                      int time_major, int len, const int32_t* __restrict__ d_halfwidths,
                                                      int max_r, uint16_t* __restrict__ d_roi, uint8_t* __restrict__ d_fg,
                                                      uint8_t* __restrict__ d_bg, double* __restrict__ d_sums,
                                                      int32_t* __restrict__ d_counts) {
  extern __shared__ __attribute__((aligned(16))) uint8_t rt_smem[];
  uint32_t* s_tile = reinterpret_cast<uint32_t*>(rt_smem);                       // [RT_CT][RT_H][RT_W / 2] pixel pairs
  uint32_t* s_any = s_tile + RT_CT * RT_H * (RT_W / 2);                          // [RT_H][RT_WPR]
  uint32_t* s_multi = s_any + RT_H * RT_WPR;                                     // [RT_H][RT_WPR]
  uint32_t* s_fg = s_multi + RT_H * RT_WPR;                                      // [RT_F][RT_H][4]
  uint32_t* s_bg = s_fg + RT_F * RT_H * 4;                                       // [RT_F][RT_H][4]
  uint16_t* s_ids = reinterpret_cast<uint16_t*>(s_bg + RT_F * RT_H * 4);         // [RT_IDS]
  int32_t* s_hw = reinterpret_cast<int32_t*>(s_ids + RT_IDS);                    // [(max_r + 1)][2 max_r + 1] half widths
  __shared__ int s_nfrag, s_ndisk;
  __shared__ int s_disk[RT_DISKS][3];
  __shared__ RtFrag s_frag[RT_F];
  __shared__ int s_cnt[RT_F][2];
  const int assay = blockIdx.z, tx0 = blockIdx.x * RT_W, ty0 = blockIdx.y * RT_H;
  const int tw = min(RT_W, w - tx0), th = min(RT_H, h - ty0);
  const int first = d_assay_offsets[assay], nb = d_assay_offsets[assay + 1] - first;
  if (nb <= 0) return;
  const int32_t* beads = d_beads + 3 * (bead_stride ? (int64_t)assay * bead_stride : (int64_t)first);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int WV = RTN / 64;
  for (int i = threadIdx.x; i < 2 * RT_H * RT_WPR; i += RTN) s_any[i] = 0u;  // (any and multi are adjacent)
  for (int i = threadIdx.x; i < (max_r + 1) * (2 * max_r + 1); i += RTN) s_hw[i] = d_halfwidths[i];
  if (threadIdx.x == 0) s_nfrag = 0, s_ndisk = 0;
  __syncthreads();
  // ---- 1. the assay's beads: whose window reaches into the tile, whose disk does ----
  const int side = 2 * max_r + 1;
  auto draw_row = [&](int r, int yj, int xj, int rj) {  // row r of the tile under the disk
    const int dy = ty0 + r - yj;
    if (dy < -rj || dy > rj) return;
    const int hwid = s_hw[rj * side + dy + rj];
    if (hwid < 0) return;
    const int xa = max(xj - hwid, tx0) - tx0, xb = min(xj + hwid, tx0 + tw - 1) - tx0;
    for (int wd = xa >> 5; xa <= xb && wd <= (xb >> 5); ++wd) {
      const int lo = max(xa, 32 * wd) - 32 * wd, hi = min(xb, 32 * wd + 31) - 32 * wd;
      const uint32_t bits = (hi - lo == 31) ? 0xFFFFFFFFu : (((1u << (hi - lo + 1)) - 1u) << lo);
      const uint32_t old = atomicOr(&s_any[r * RT_WPR + wd], bits);
      if (old & bits) atomicOr(&s_multi[r * RT_WPR + wd], old & bits);
    }
  };
  constexpr int UB = 4;
  for (int b0 = 0; b0 < nb; b0 += RTN * UB) {
    int yy[UB], xx[UB], rr[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = b0 + u * RTN + (int)threadIdx.x;
      yy[u] = xx[u] = 0;
      rr[u] = -1;
      if (j < nb) yy[u] = beads[3 * j], xx[u] = beads[3 * j + 1], rr[u] = beads[3 * j + 2];
    }
#pragma unroll
    for (int u = 0; u < UB; ++u) {
      const int j = b0 + u * RTN + (int)threadIdx.x;
      if (j >= nb) continue;
      const int yj = yy[u], xj = xx[u], rj = rr[u];
      if (j < RT_IDS) {  // (the windows of the first RT_IDS beads: the block loop below scans the later ones itself)
        int top, left;
        window(yj, len, h, top);
        window(xj, len, w, left);
        if (top < ty0 + th && top + len > ty0 && left < tx0 + tw && left + len > tx0) s_ids[atomicAdd(&s_nfrag, 1)] = (uint16_t)j;
      }
      if (rj < 2 || rj > max_r) continue;  // undefined in the reference, no coverage (as k_circle_labels)
      if (yj + rj < ty0 || yj - rj >= ty0 + th || xj + rj < tx0 || xj - rj >= tx0 + tw) continue;
      const int k = atomicAdd(&s_ndisk, 1);
      if (k < RT_DISKS) {
        s_disk[k][0] = yj, s_disk[k][1] = xj, s_disk[k][2] = rj;
      } else {  // an extremely crowded tile: this thread draws the whole disk itself
        for (int r = max(yj - rj, ty0) - ty0; r <= min(yj + rj, ty0 + th - 1) - ty0; ++r) draw_row(r, yj, xj, rj);
      }
    }
  }
  __syncthreads();
  if (nb <= RT_IDS && s_nfrag == 0) return;  // nobody wants this tile: it is not read at all
  {
    const int nd = min(s_ndisk, RT_DISKS);
    for (int p = threadIdx.x; p < nd * RT_H; p += RTN) {
      const int k = p / RT_H, r = p - k * RT_H;
      if (r < th) draw_row(r, s_disk[k][0], s_disk[k][1], s_disk[k][2]);
    }
  }
  __syncthreads();
  // ---- 2. rounds of RT_F windows ----
  const uint16_t* img = d_image + (int64_t)assay * assay_stride;
  const int nct = n_c * n_t, half = len >> 1, n = len * len;
  const int64_t plane_elems = (int64_t)h * w;
  // this thread's 16-byte pieces of a tile plane: piece q = threadIdx.x + RTN * i, row q / (RT_W / 8), 8 pixels from column 8 (q % (RT_W / 8))
  constexpr int PIECES = (RT_H * (RT_W / 8) + RTN - 1) / RTN;
  auto fetch_planes = [&](int ct0, uint4 (&v)[RT_CT][PIECES]) {
#pragma unroll
    for (int c = 0; c < RT_CT; ++c) {
      const int ct = min(ct0 + c, nct - 1);
      const uint16_t* plane = img + (int64_t)(time_major ? (ct % n_t) * n_c + ct / n_t : ct) * plane_elems;
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int q = threadIdx.x + RTN * i, r = q / (RT_W / 8), c8 = q - r * (RT_W / 8);
        v[c][i] = make_uint4(0u, 0u, 0u, 0u);
        if (q < RT_H * (RT_W / 8) && r < th && 8 * c8 < tw) v[c][i] = *reinterpret_cast<const uint4*>(plane + (int64_t)(ty0 + r) * w + tx0 + 8 * c8);
      }
    }
  };
  auto stash_planes = [&](const uint4 (&v)[RT_CT][PIECES]) {
#pragma unroll
    for (int c = 0; c < RT_CT; ++c)
#pragma unroll
      for (int i = 0; i < PIECES; ++i) {
        const int q = threadIdx.x + RTN * i;
        if (q < RT_H * (RT_W / 8)) reinterpret_cast<uint4*>(s_tile + c * RT_H * (RT_W / 2))[q] = v[c][i];
      }
  };
  // blocks of RT_IDS beads (one block unless an assay holds more): the windows of a block that reach into the tile
  for (int blk = 0; blk < nb; blk += RT_IDS) {
  if (blk > 0) {
    __syncthreads();
    if (threadIdx.x == 0) s_nfrag = 0;
    __syncthreads();
    for (int j = blk + threadIdx.x; j < min(blk + RT_IDS, nb); j += RTN) {
      int top, left;
      window(beads[3 * j], len, h, top);
      window(beads[3 * j + 1], len, w, left);
      if (top < ty0 + th && top + len > ty0 && left < tx0 + tw && left + len > tx0) s_ids[atomicAdd(&s_nfrag, 1)] = (uint16_t)(j - blk);
    }
    __syncthreads();
  }
  const int nfrag = s_nfrag;
  for (int f0 = 0; f0 < nfrag; f0 += RT_F) {
    const int nf = min(RT_F, nfrag - f0);
    __syncthreads();  // the previous round's descriptors, masks and tile planes are no longer read
    if ((int)threadIdx.x < nf) {
      const int j = blk + s_ids[f0 + threadIdx.x];
      RtFrag fr;
      fr.g = first + j;
      fr.yj = beads[3 * j], fr.xj = beads[3 * j + 1], fr.rj = beads[3 * j + 2];
      window(fr.yj, len, h, fr.top);
      window(fr.xj, len, w, fr.left);
      fr.r0 = max(fr.top - ty0, 0);
      fr.r1 = min(fr.top + len - ty0, th);
      s_frag[threadIdx.x] = fr;
      s_cnt[threadIdx.x][0] = s_cnt[threadIdx.x][1] = 0;
    }
    __syncthreads();
    // mask rows: (fragment, tile row, 32-column word of the window) -> fg / bg bits of the pixels that lie in this tile
    for (int it = threadIdx.x; it < nf * RT_H * 4; it += RTN) {
      const int f = it / (RT_H * 4), r = (it >> 2) & (RT_H - 1), wd = it & 3;
      const RtFrag fr = s_frag[f];
      uint32_t fgb = 0u, bgb = 0u;
      if (r >= fr.r0 && r < fr.r1 && 32 * wd < len) {
        const int c0 = fr.left - tx0 + 32 * wd;  // tile column of the word's bit 0 (may be negative / beyond the tile)
        // columns of the word that are window columns (< len) AND lie in the tile
        const int lo = max(0, -c0), hi = min(min(32, len - 32 * wd), tw - c0);  // bits [lo, hi)
        if (lo < hi) {
          const uint32_t in = (hi - lo == 32) ? 0xFFFFFFFFu : (((1u << (hi - lo)) - 1u) << lo);
          // any / multi bits of tile columns c0 .. c0 + 31
          const int wi = c0 >> 5, sh = c0 & 31;  // (arithmetic shift: floor)
          auto word_at = [&](const uint32_t* row, int k) { return (k >= 0 && k < RT_WPR) ? row[k] : 0u; };
          const uint32_t* ar = s_any + r * RT_WPR;
          const uint32_t* mr = s_multi + r * RT_WPR;
          const uint64_t a2 = ((uint64_t)word_at(ar, wi + 1) << 32) | word_at(ar, wi);
          const uint64_t m2 = ((uint64_t)word_at(mr, wi + 1) << 32) | word_at(mr, wi);
          const uint32_t anyb = (uint32_t)(a2 >> sh), multib = (uint32_t)(m2 >> sh);
          // the window's own disk in this row
          uint32_t own = 0u;
          const int dy = ty0 + r - fr.yj;
          if (fr.rj >= 2 && fr.rj <= max_r && dy >= -fr.rj && dy <= fr.rj) {
            const int hwid = s_hw[fr.rj * side + dy + fr.rj];
            if (hwid >= 0) {
              const int xa = max(fr.xj - hwid - fr.left - 32 * wd, 0), xb = min(fr.xj + hwid - fr.left - 32 * wd, 31);
              if (xa <= xb) own = (xb - xa == 31) ? 0xFFFFFFFFu : (((1u << (xb - xa + 1)) - 1u) << xa);
            }
          }
          fgb = own & ~multib & in;
          bgb = ~anyb & in;
        }
      }
      s_fg[(f * RT_H + r) * 4 + wd] = fgb;
      s_bg[(f * RT_H + r) * 4 + wd] = bgb;
      if (fgb) atomicAdd(&s_cnt[f][0], __popc(fgb));
      if (bgb) atomicAdd(&s_cnt[f][1], __popc(bgb));
    }
    {
      uint4 v[RT_CT][PIECES];
      fetch_planes(0, v);  // (the first planes' loads are in flight while the masks settle)
      stash_planes(v);
    }
    __syncthreads();
    if ((int)threadIdx.x < 2 * nf && d_counts) {
      const int f = threadIdx.x >> 1, k = threadIdx.x & 1;
      if (s_cnt[f][k]) atomicAdd(&d_counts[2 * (int64_t)s_frag[f].g + k], s_cnt[f][k]);
    }
    // per fragment (a wave each): lane l serves window columns 2 l, 2 l + 1
    const int x = 2 * lane;
    const bool act = lane < half;
    const int mword = x >> 5, msh = x & 31;
    for (int ct0 = 0; ct0 < nct; ct0 += RT_CT) {
      if (ct0) {
        uint4 v[RT_CT][PIECES];
        fetch_planes(ct0, v);
        __syncthreads();  // the planes before are served
        stash_planes(v);
        __syncthreads();
      }
      for (int f = wave; f < nf; f += WV) {
        const RtFrag fr = s_frag[f];
        const int odd = fr.left & 1;  // (tx0 is even: the parity of the window's first column inside the tile)
        const uint32_t shift = odd ? 16u : 0u;
        const int e2 = (fr.left - odd - tx0) >> 1;  // dword of the lane-0 pair's first source pixel (may be negative)
        const int di = min(max(e2 + min(lane, half - 1 + odd), 0), RT_W / 2 - 1);
        const int tc = fr.left - tx0 + x;
        const bool in0 = act && tc >= 0 && tc < tw, in1 = act && tc + 1 >= 0 && tc + 1 < tw;
        uint32_t sf[RT_CT], sb[RT_CT];
#pragma unroll
        for (int c = 0; c < RT_CT; ++c) sf[c] = 0u, sb[c] = 0u;
        const bool masks_out = ct0 == 0 && (d_fg || d_bg);
        auto serve_row = [&](int r, const uint32_t (&dd)[RT_CT]) {
          const uint32_t fb = (s_fg[(f * RT_H + r) * 4 + mword] >> msh) & 3u, bb = (s_bg[(f * RT_H + r) * 4 + mword] >> msh) & 3u;
          const uint32_t mf = roi_pair(fb), mb = roi_pair(bb);
          const int ry = ty0 + r - fr.top;
          const int64_t oi = (int64_t)ry * half + lane;  // dword of the window
          if (masks_out) {  // the mask bytes of the fragment's pixels, once
            const int64_t o = (int64_t)fr.g * n + ry * len + x;
            if (in0 && in1) {
              if (d_fg) *reinterpret_cast<uint16_t*>(&d_fg[o]) = (uint16_t)((fb & 1u) | ((fb & 2u) << 7));
              if (d_bg) *reinterpret_cast<uint16_t*>(&d_bg[o]) = (uint16_t)((bb & 1u) | ((bb & 2u) << 7));
            } else if (in0) {
              if (d_fg) d_fg[o] = (uint8_t)(fb & 1u);
              if (d_bg) d_bg[o] = (uint8_t)(bb & 1u);
            } else if (in1) {
              if (d_fg) d_fg[o + 1] = (uint8_t)(fb >> 1);
              if (d_bg) d_bg[o + 1] = (uint8_t)(bb >> 1);
            }
          }
#pragma unroll
          for (int c = 0; c < RT_CT; ++c) {
            if (ct0 + c >= nct) break;  // uniform
            const uint32_t d = dd[c];
            const uint32_t nx = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)d, 0x130, 0xF, 0xF, false);  // lane + 1
            const uint32_t v = __builtin_amdgcn_alignbit(nx, d, shift);
            if (d_roi) {
              uint32_t* out = reinterpret_cast<uint32_t*>(d_roi + ((int64_t)fr.g * nct + ct0 + c) * n);
              if (in0 && in1) out[oi] = v;
              else if (in0) reinterpret_cast<uint16_t*>(out)[2 * oi] = (uint16_t)v;
              else if (in1) reinterpret_cast<uint16_t*>(out)[2 * oi + 1] = (uint16_t)(v >> 16);
            }
            sf[c] = roi_dot2(v, mf, sf[c]);  // (pixels outside the tile carry mask 0)
            sb[c] = roi_dot2(v, mb, sb[c]);
          }
        };
        for (int r = fr.r0; r < fr.r1; r += 2) {  // two rows per trip: their LDS reads are in flight together
          const int rb = min(r + 1, fr.r1 - 1);
          uint32_t da[RT_CT], db[RT_CT];
#pragma unroll
          for (int c = 0; c < RT_CT; ++c) {
            da[c] = s_tile[(c * RT_H + r) * (RT_W / 2) + di];
            db[c] = s_tile[(c * RT_H + rb) * (RT_W / 2) + di];
          }
          serve_row(r, da);
          if (r + 1 < fr.r1) serve_row(r + 1, db);  // wave-uniform
        }
        if (d_sums) {
#pragma unroll
          for (int c = 0; c < RT_CT; ++c) {
            if (ct0 + c >= nct) break;
            const uint32_t a = (uint32_t)mg_wave_scan_incl_i32((int)sf[c]), b = (uint32_t)mg_wave_scan_incl_i32((int)sb[c]);
            if (lane == 63) {
              double* o = d_sums + ((int64_t)fr.g * nct + ct0 + c) * 2;
              if (a) atomicAdd(o, (double)a);
              if (b) atomicAdd(o + 1, (double)b);
            }
          }
        }
      }
    }
  }
  }
}

constexpr size_t RT_LDS = (size_t)RT_CT * RT_H * RT_W * 2 + 2 * RT_H * RT_WPR * 4 + 2 * RT_F * RT_H * 4 * 4 + RT_IDS * 2;  // + the half-width table

// ---- masked median: byte-wise radix select in LDS -------------------------------------------------
// Keys are the order-preserving unsigned images of the values: an unsigned integer is its own key, an IEEE float has
// its sign bit flipped (positive) or all bits inverted (negative).  A NaN pixel counts as masked out (nanmedian).
template <typename T> struct median_key;
template <> struct median_key<uint8_t> {
  typedef uint32_t type; static constexpr int LEVELS = 1;
  __device__ static bool key(uint8_t x, uint32_t* k) { *k = x; return true; }
  __device__ static double value(uint32_t k) { return (double)k; }
};
template <> struct median_key<uint16_t> {
  typedef uint32_t type; static constexpr int LEVELS = 2;
  __device__ static bool key(uint16_t x, uint32_t* k) { *k = x; return true; }
  __device__ static double value(uint32_t k) { return (double)k; }
};
template <> struct median_key<float> {
  typedef uint32_t type; static constexpr int LEVELS = 4;
  __device__ static bool key(float x, uint32_t* k) {
    const uint32_t b = __float_as_uint(x);
    *k = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
    return x == x;
  }
  __device__ static double value(uint32_t k) {
    return (double)__uint_as_float((k & 0x80000000u) ? (k & 0x7FFFFFFFu) : ~k);
  }
};
template <> struct median_key<double> {
  typedef uint64_t type; static constexpr int LEVELS = 8;
  __device__ static bool key(double x, uint64_t* k) {
    const uint64_t b = (uint64_t)__double_as_longlong(x);
    *k = (b >> 63) ? ~b : (b | 0x8000000000000000ull);
    return x == x;
  }
  __device__ static double value(uint64_t k) {
    return __longlong_as_double((long long)((k >> 63) ? (k & 0x7FFFFFFFFFFFFFFFull) : ~k));
  }
};

// the k-th smallest (0-based) key among the unmasked, non-NaN values: one 256-bin histogram per key byte, most
// significant first, only the values that share the prefix found so far take part
template <typename T>
__device__ typename median_key<T>::type select_kth(const T* __restrict__ v, const uint8_t* __restrict__ mask, int n,
                                                   int k, uint32_t* hist) {
  typedef typename median_key<T>::type K;
  constexpr int LV = median_key<T>::LEVELS;
  __shared__ int s_bin, s_rank;
  K prefix = 0;
  int rank = k;
  for (int level = 0; level < LV; ++level) {
    const int shift = 8 * (LV - 1 - level);
    for (int i = threadIdx.x; i < 256; i += NT) hist[i] = 0;
    __syncthreads();
    for (int p = threadIdx.x; p < n; p += NT) {
      K key;
      if (!mask[p] || !median_key<T>::key(v[p], &key)) continue;
      if (level == 0 || (key >> (shift + 8)) == prefix) atomicAdd(&hist[(uint32_t)(key >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
      int acc = 0, b = 0;
      for (; b < 255; ++b) {
        if (acc + (int)hist[b] > rank) break;
        acc += hist[b];
      }
      s_bin = b;
      s_rank = rank - acc;
    }
    __syncthreads();
    prefix = (prefix << 8) | (K)s_bin;
    rank = s_rank;
    __syncthreads();
  }
  return prefix;
}

// one workgroup per (marker, channel-time): mask of marker g at time t = d_mask + g * mask_stride_m + t * mask_stride_t
template <typename T>
__global__ __launch_bounds__(NT) void k_masked_median(const T* __restrict__ d_roi, const uint8_t* __restrict__ d_mask,
                                                      int64_t mask_stride_m, int64_t mask_stride_t, int n_t, int n_ct,
                                                      int n, double* __restrict__ d_median) {
  __shared__ uint32_t hist[256];
  __shared__ int s_count;
  const int g = blockIdx.x, ct = blockIdx.y;
  const T* v = d_roi + ((int64_t)g * n_ct + ct) * n;
  const uint8_t* mask = d_mask + (int64_t)g * mask_stride_m + (int64_t)(ct % n_t) * mask_stride_t;
  int c = 0;
  for (int p = threadIdx.x; p < n; p += NT) {
    typename median_key<T>::type key;
    c += mask[p] != 0 && median_key<T>::key(v[p], &key);
  }
  int total;
  mg_block_exscan(c, &total);
  if (threadIdx.x == 0) s_count = total;
  __syncthreads();
  const int cnt = s_count;
  double* out = d_median + (int64_t)g * n_ct + ct;
  if (cnt == 0) {
    if (threadIdx.x == 0) *out = __longlong_as_double(0x7FF8000000000000ll);
    return;
  }
  const auto lo = select_kth<T>(v, mask, n, (cnt - 1) / 2, hist);
  auto hi = lo;
  if ((cnt & 1) == 0) hi = select_kth<T>(v, mask, n, cnt / 2, hist);
  // numpy's nanmedian: the mean of the two middle values (here in float64: exact for every type but float64 itself)
  if (threadIdx.x == 0) *out = (median_key<T>::value(lo) + median_key<T>::value(hi)) / 2.0;
}

template <typename T>
int launch_median(const void* d_roi, const uint8_t* d_mask, int64_t sm, int64_t st, int m, int n_c, int n_t, int len,
                  double* d_median, hipStream_t s) {
  hipLaunchKernelGGL((k_masked_median<T>), dim3(m, n_c * n_t), dim3(NT), 0, s, (const T*)d_roi, d_mask, sm, st, n_t,
                     n_c * n_t, len * len, d_median);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

template <typename T, typename ACC>
int launch_roi(const void* d_image, int64_t assay_stride, int n_c, int n_t, int h, int w, const int32_t* d_beads,
               const int32_t* d_marker_assay, const int32_t* d_marker_local, dim3 grid, int len,
               const int32_t* d_labels, const int32_t* d_assay_offsets, int n_assays, int64_t bead_stride, int time_major,
               const int32_t* d_order, const int32_t* d_halfwidths,
               int max_r, void* d_roi, uint8_t* d_fg, uint8_t* d_bg, double* d_sums, int32_t* d_counts, hipStream_t s) {
  hipLaunchKernelGGL((k_roi<T, ACC>), grid, dim3(NT), roi_lds_bytes(len, d_halfwidths != nullptr), s,
                     (const T*)d_image, assay_stride, n_c, n_t, h, w, d_beads, d_marker_assay, d_marker_local, len,
                     d_labels, d_assay_offsets, n_assays, bead_stride, time_major, d_order, d_halfwidths, max_r, (T*)d_roi, d_fg, d_bg, d_sums, d_counts);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

}  // namespace

extern "C" int mg_circle_labels(const int32_t* d_beads, int64_t bead_cap, const int32_t* d_num_beads, int n_planes,
                                int h, int w, const int32_t* d_halfwidths, int max_r, int32_t* d_labels,
                                int reset, void* stream) {
  if (!d_beads || !d_num_beads || !d_halfwidths || !d_labels || n_planes < 0 || n_planes > 65535 || bead_cap < 0 ||
      max_r < 0)
    return MG_EINVAL;
  if (n_planes == 0 || bead_cap == 0) return MG_OK;
  hipLaunchKernelGGL(k_circle_labels, dim3((unsigned)bead_cap, n_planes), dim3(NT), 0, mg_stream(stream), d_beads,
                     bead_cap, d_num_beads, h, w, d_halfwidths, max_r, d_labels, reset);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

namespace {
int roi_dispatch(const void* d_image, int dtype, int64_t assay_stride, int n_c, int n_t, int h, int w,
                 const int32_t* d_beads, const int32_t* d_marker_assay, const int32_t* d_marker_local, int m, int roi_len,
                 const int32_t* d_labels, const int32_t* d_assay_offsets, int64_t bead_stride, int time_major, int n_assays,
                 const int32_t* d_order, const int32_t* d_halfwidths, int max_r, void* d_roi, uint8_t* d_fg, uint8_t* d_bg,
                 double* d_sums, int32_t* d_counts, void* stream) {
  if (!d_image || !d_beads || m < 0 || roi_len <= 0 || n_c <= 0 || n_t <= 0) return MG_EINVAL;
  if (roi_len > h || roi_len > w || roi_lds_bytes(roi_len, d_halfwidths != nullptr) > 60000) return MG_EINVAL;
  if (m == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  const dim3 grid(m);
  // MG_ROI_TILES=1 (looked at on every call: the tests switch it): the image-centric pass of round 4 -- masks from the
  // bead tables of whole assays, uint16, 16-byte aligned rows.  Measured at C4 (profiles/r4_roi_tiles.txt): 21.9 GB of
  // HBM traffic instead of 28.2 (fetches 15.7 -> 8.5 GB: every line once), but 7.05 ms against 4.37 -- 16 waves per CU
  // behind 68 KB of LDS and ten barriers per tile leave its latencies in the open.  Not the default.
  const char* tiles_env = getenv("MG_ROI_TILES");
  const bool tiles = tiles_env && tiles_env[0] == '1';
  if (tiles && dtype == MG_U16 && d_halfwidths && d_assay_offsets && !d_labels && (roi_len & 1) == 0 && roi_len <= 126 &&
      (w & 7) == 0 && (assay_stride & 7) == 0 && (int64_t)h * w < (1LL << 31) && n_assays > 0 && n_assays <= 65535 &&
      (bead_stride ? bead_stride : (int64_t)m) <= 65535 && max_r >= 2 && (max_r + 1) * (2 * max_r + 1) * 4 <= 27 * 53 * 4 + 8192 &&
      (reinterpret_cast<uintptr_t>(d_image) & 15) == 0 && (!d_roi || (reinterpret_cast<uintptr_t>(d_roi) & 3) == 0) &&
      (!d_fg || (reinterpret_cast<uintptr_t>(d_fg) & 1) == 0) && (!d_bg || (reinterpret_cast<uintptr_t>(d_bg) & 1) == 0)) {
    const int nct = n_c * n_t;
    if (d_sums && mg_zero_async(d_sums, (size_t)m * nct * 2 * sizeof(double), s) != hipSuccess) return MG_ELAUNCH;
    if (d_counts && mg_zero_async(d_counts, (size_t)m * 2 * sizeof(int32_t), s) != hipSuccess) return MG_ELAUNCH;
    static bool attr_set = false;
    if (!attr_set) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_roi_tiles_u16), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)RT_LDS + 27 * 53 * 4 + 8192) != hipSuccess)
        return MG_ELAUNCH;
      attr_set = true;
    }
    hipLaunchKernelGGL(k_roi_tiles_u16, dim3((w + RT_W - 1) / RT_W, (h + RT_H - 1) / RT_H, n_assays), dim3(RTN), RT_LDS + (size_t)(max_r + 1) * (2 * max_r + 1) * 4, s,
                       (const uint16_t*)d_image, assay_stride, n_c, n_t, h, w, d_beads, bead_stride, d_assay_offsets,
                       time_major, roi_len, d_halfwidths, max_r, (uint16_t*)d_roi, d_fg, d_bg, d_sums, d_counts);
    MG_CHECK_LAUNCH();
    return MG_OK;
  }
  if (dtype == MG_U16 && (roi_len & 1) == 0 && roi_len <= 126 && (w & 1) == 0 && (assay_stride & 1) == 0 &&
      (int64_t)h * w < (1LL << 31) &&
      (reinterpret_cast<uintptr_t>(d_image) & 3) == 0 && (!d_roi || (reinterpret_cast<uintptr_t>(d_roi) & 3) == 0) &&
      (!d_fg || (reinterpret_cast<uintptr_t>(d_fg) & 1) == 0) && (!d_bg || (reinterpret_cast<uintptr_t>(d_bg) & 1) == 0)) {
    // 2 rows x 4 planes per trip, next trip's loads in flight while this one is worked on (measured at 16 x 4 x 4096^2:
    // 1.14 ms; without the pipelining 1.22, one row per trip 1.22, 4 rows x 2 planes 1.20, 4 x 4 unpipelined 1.24)
    hipLaunchKernelGGL((k_roi_u16_even<2, 4, true>), grid, dim3(NT), (size_t)mask_words(roi_len) * 4, s,
                       (const uint16_t*)d_image, assay_stride, n_c, n_t, h, w, d_beads, d_marker_assay, d_marker_local,
                       roi_len, d_labels, d_assay_offsets, n_assays, bead_stride, time_major, d_order, d_halfwidths, max_r, (uint16_t*)d_roi, d_fg, d_bg,
                       d_sums, d_counts);
    MG_CHECK_LAUNCH();
    return MG_OK;
  }
  switch (dtype) {
    case MG_U8:
      return launch_roi<uint8_t, long long>(d_image, assay_stride, n_c, n_t, h, w, d_beads, d_marker_assay,
                                            d_marker_local, grid, roi_len, d_labels, d_assay_offsets, n_assays, bead_stride, time_major, d_order, d_halfwidths, max_r,
                                            d_roi, d_fg, d_bg, d_sums, d_counts, s);
    case MG_U16:
      return launch_roi<uint16_t, long long>(d_image, assay_stride, n_c, n_t, h, w, d_beads, d_marker_assay,
                                             d_marker_local, grid, roi_len, d_labels, d_assay_offsets, n_assays, bead_stride, time_major, d_order, d_halfwidths, max_r,
                                             d_roi, d_fg, d_bg, d_sums, d_counts, s);
    case MG_F32:
      return launch_roi<float, double>(d_image, assay_stride, n_c, n_t, h, w, d_beads, d_marker_assay, d_marker_local,
                                       grid, roi_len, d_labels, d_assay_offsets, n_assays, bead_stride, time_major, d_order, d_halfwidths, max_r, d_roi, d_fg, d_bg,
                                       d_sums, d_counts, s);
    case MG_F64:
      return launch_roi<double, double>(d_image, assay_stride, n_c, n_t, h, w, d_beads, d_marker_assay, d_marker_local,
                                        grid, roi_len, d_labels, d_assay_offsets, n_assays, bead_stride, time_major, d_order, d_halfwidths, max_r, d_roi, d_fg, d_bg,
                                        d_sums, d_counts, s);
  }
  return MG_EINVAL;
}
}  // namespace

namespace {
// d_offsets[0 .. n] = exclusive prefix of min(d_counts[i], cap): where every assay's markers start in the compact
// outputs of mg_roi_segment_reduce -- on the device, so that the pass can be queued before the host has seen the counts.
__global__ __launch_bounds__(1024) void k_counts_to_offsets(const int32_t* __restrict__ d_counts, int n, int cap,
                                                            int32_t* __restrict__ d_offsets) {
  int carry = 0;
  for (int i0 = 0; i0 < n; i0 += 1024) {  // block-uniform trip count
    const int i = i0 + (int)threadIdx.x;
    const int v = i < n ? min(max(d_counts[i], 0), cap) : 0;
    int total;
    const int ex = mg_block_exscan(v, &total);
    if (i < n) d_offsets[i] = carry + ex;
    carry += total;
  }
  if (threadIdx.x == 0) d_offsets[n] = carry;
}
}  // namespace

extern "C" int mg_counts_to_offsets(const int32_t* d_counts, int n, int cap, int32_t* d_offsets, void* stream) {
  if (!d_counts || !d_offsets || n < 0 || cap < 0) return MG_EINVAL;
  hipLaunchKernelGGL(k_counts_to_offsets, dim3(1), dim3(1024), 0, mg_stream(stream), d_counts, n, cap, d_offsets);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

namespace {
// One row per marker: [assay, row, col, r, fg_count, bg_count, fg_sum[C], bg_sum[C]] (float64: exact for these
// integers) from the bead tables and the ROI pass's counts / sums where they are -- what a rank contributes to the
// all-gather of the marker table (collective 3).
__global__ __launch_bounds__(256) void k_marker_table(const int32_t* __restrict__ d_beads, int64_t bead_stride,
                                                      const int32_t* __restrict__ d_assay_offsets, int n_assays,
                                                      int assay_offset, const int32_t* __restrict__ d_counts,
                                                      const double* __restrict__ d_sums, int n_c, int n_t, int t_index,
                                                      double* __restrict__ d_table) {
  const int total = d_assay_offsets[n_assays];
  const int width = 6 + 2 * n_c;
  for (int g = blockIdx.x * blockDim.x + threadIdx.x; g < total; g += gridDim.x * blockDim.x) {
    int lo = 0, hi = n_assays;  // the assay whose offset range holds g (empty assays have empty ranges)
    while (hi - lo > 1) {
      const int mid = (lo + hi) >> 1;
      if (d_assay_offsets[mid] <= g) lo = mid;
      else hi = mid;
    }
    const int local = g - d_assay_offsets[lo];
    const int32_t* b = bead_stride > 0 ? d_beads + ((int64_t)lo * bead_stride + local) * 3 : d_beads + (int64_t)g * 3;
    double* row = d_table + (int64_t)g * width;
    row[0] = (double)(assay_offset + lo);
    row[1] = (double)b[0], row[2] = (double)b[1], row[3] = (double)b[2];
    row[4] = (double)d_counts[2 * g], row[5] = (double)d_counts[2 * g + 1];
    for (int c = 0; c < n_c; ++c) {
      const double* sm = d_sums + (((int64_t)g * n_c + c) * n_t + t_index) * 2;
      row[6 + c] = sm[0];
      row[6 + n_c + c] = sm[1];
    }
  }
}
}  // namespace

extern "C" int mg_marker_table(const int32_t* d_beads, int64_t bead_stride, const int32_t* d_assay_offsets, int n_assays,
                               int m, int assay_offset, const int32_t* d_counts, const double* d_sums, int n_c, int n_t,
                               int t_index, double* d_table, void* stream) {
  if (!d_assay_offsets || n_assays <= 0 || n_assays > 65535 || m < 0 || bead_stride < 0 || n_c <= 0 || n_t <= 0 ||
      t_index < 0 || t_index >= n_t)
    return MG_EINVAL;
  if (m == 0) return MG_OK;
  if (!d_beads || !d_counts || !d_sums || !d_table) return MG_EINVAL;
  hipLaunchKernelGGL(k_marker_table, dim3((unsigned)std::min((m + 255) / 256, 2048)), dim3(256), 0, mg_stream(stream),
                     d_beads, bead_stride, d_assay_offsets, n_assays, assay_offset, d_counts, d_sums, n_c, n_t, t_index,
                     d_table);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_roi_gather_reduce_batched(const void* d_image, int dtype, int64_t assay_stride, int n_c, int n_t,
                                            int h, int w, const int32_t* d_beads, const int32_t* d_marker_assay,
                                            const int32_t* d_marker_local, int m, int roi_len,
                                            const int32_t* d_labels, void* d_roi, uint8_t* d_fg, uint8_t* d_bg,
                                            double* d_sums, int32_t* d_counts, void* stream) {
  return roi_dispatch(d_image, dtype, assay_stride, n_c, n_t, h, w, d_beads, d_marker_assay, d_marker_local, m, roi_len,
                      d_labels, nullptr, 0, 0, 0, nullptr, nullptr, 0, d_roi, d_fg, d_bg, d_sums, d_counts, stream);
}

namespace {
// ---- the order the windows of an assay are visited in ---------------------------------------------------------
// The beads of an assay arrive in suppression (score) order, spatially random: two windows that share image lines
// (200-byte rows at arbitrary alignment touch 2.56 lines of 128 B; neighbouring windows overlap) are then worked on
// long after each other and every one fetches its lines from HBM again.  Visited band by band (64 rows) and left to
// right inside a band, neighbours are in flight together and meet in the L2s: 4.73 instead of 5.17 ms at C4 in
// tools/roi_order_probe.py.  d_order[g] = the marker the g-th workgroup takes; only the ORDER of the work changes,
// every marker's outputs stay where they were.  ORDER_PARTS workgroups per assay hold the assay's keys in LDS and rank
// slices of 256 markers each: rank = keys before the marker that are <= its key + keys after it that are < (equal
// keys keep their table order) -- two instructions per key except in the few chunks around the wave's own markers.
// (One workgroup of 1024 per assay took 180 us at C4: 2 000^2 compares on one CU.)  An assay with more beads than
// ORDER_CAP keeps its order.
constexpr int ORDER_CAP = 8192;
constexpr int ORDER_BAND = 6;  // log2 of the band height
constexpr int ORDER_PARTS = 8;

__global__ __launch_bounds__(NT) void k_window_order(const int32_t* __restrict__ d_beads, int64_t bead_stride,
                                                     const int32_t* __restrict__ d_assay_offsets, int m,
                                                     int32_t* __restrict__ d_order) {
  __shared__ __attribute__((aligned(16))) uint32_t keys[ORDER_CAP];
  const int assay = blockIdx.x;
  const int first = d_assay_offsets[assay];
  const int n = min(d_assay_offsets[assay + 1], m) - first;  // (markers beyond the launch's bound m are not worked on)
  if (n <= 0) return;
  if (n > ORDER_CAP) {
    for (int i = blockIdx.y * NT + threadIdx.x; i < n; i += ORDER_PARTS * NT) d_order[first + i] = first + i;
    return;
  }
  const int32_t* b = d_beads + 3 * (bead_stride ? (int64_t)assay * bead_stride : (int64_t)first);
  const int n4 = (n + 3) & ~3, nq = n4 >> 2;
  for (int i = threadIdx.x; i < n4; i += NT) {
    uint32_t k = 0xFFFFFFFFu;  // (padding: after every marker, above every key)
    if (i < n) {
      const int row = min(max(b[3 * i], 0), (1 << 20) - 1), col = min(max(b[3 * i + 1], 0), (1 << 17) - 1);
      k = ((uint32_t)(row >> ORDER_BAND) << 17) | (uint32_t)col;
    }
    keys[i] = k;
  }
  __syncthreads();
  const uint4* k4 = reinterpret_cast<const uint4*>(keys);
  for (int i0 = blockIdx.y * NT; i0 < n; i0 += ORDER_PARTS * NT) {
    const int i = i0 + threadIdx.x;
    const int w0 = __builtin_amdgcn_readfirstlane(i0 + (int)(threadIdx.x & ~63u));  // the wave's first marker
    const int q_lo = min(w0 >> 2, nq), q_hi = min((w0 + 63) >> 2, nq - 1);
    const uint32_t ki = keys[min(i, n - 1)];
    int rank = 0;
    for (int q = 0; q < q_lo; ++q) {  // before every marker of the wave
      const uint4 v = k4[q];
      rank += (v.x <= ki) + (v.y <= ki) + (v.z <= ki) + (v.w <= ki);
    }
    for (int q = q_lo; q <= q_hi; ++q) {  // around them
      const uint4 v = k4[q];
      const int j = 4 * q;
      rank += (v.x < ki || (v.x == ki && j < i)) + (v.y < ki || (v.y == ki && j + 1 < i)) +
              (v.z < ki || (v.z == ki && j + 2 < i)) + (v.w < ki || (v.w == ki && j + 3 < i));
    }
    for (int q = q_hi + 1; q < nq; ++q) {  // after
      const uint4 v = k4[q];
      rank += (v.x < ki) + (v.y < ki) + (v.z < ki) + (v.w < ki);
    }
    if (i < n) d_order[first + rank] = first + i;
  }
}
}  // namespace

extern "C" int mg_roi_window_order(const int32_t* d_beads, int64_t bead_stride, const int32_t* d_assay_offsets,
                                   int n_assays, int m, int32_t* d_order, void* stream) {
  if (!d_beads || !d_assay_offsets || !d_order || n_assays <= 0 || n_assays > 65535 || m < 0 || bead_stride < 0)
    return MG_EINVAL;
  if (m == 0) return MG_OK;
  hipLaunchKernelGGL(k_window_order, dim3(n_assays, ORDER_PARTS), dim3(NT), 0, mg_stream(stream), d_beads, bead_stride,
                     d_assay_offsets, m, d_order);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_roi_segment_reduce(const void* d_image, int dtype, int64_t assay_stride, int n_c, int n_t, int h,
                                     int w, int time_major, const int32_t* d_beads, int64_t bead_stride,
                                     const int32_t* d_assay_offsets, int n_assays, int m, const int32_t* d_order,
                                     int roi_len, const int32_t* d_halfwidths, int max_r, void* d_roi, uint8_t* d_fg,
                                     uint8_t* d_bg, double* d_sums, int32_t* d_counts, void* stream) {
  if (!d_assay_offsets || !d_halfwidths || n_assays <= 0 || n_assays > 65535 || max_r < 0 || bead_stride < 0)
    return MG_EINVAL;
  return roi_dispatch(d_image, dtype, assay_stride, n_c, n_t, h, w, d_beads, nullptr, nullptr, m, roi_len, nullptr,
                      d_assay_offsets, bead_stride, time_major, n_assays, d_order, d_halfwidths, max_r, d_roi, d_fg, d_bg,
                      d_sums, d_counts, stream);
}

extern "C" int mg_roi_gather_reduce(const void* d_image, int dtype, int n_c, int n_t, int h, int w,
                                    const int32_t* d_beads, int m, int roi_len, const int32_t* d_labels, void* d_roi,
                                    uint8_t* d_fg, uint8_t* d_bg, double* d_sums, int32_t* d_counts, void* stream) {
  return mg_roi_gather_reduce_batched(d_image, dtype, 0, n_c, n_t, h, w, d_beads, nullptr, nullptr, m, roi_len,
                                      d_labels, d_roi, d_fg, d_bg, d_sums, d_counts, stream);
}

extern "C" int mg_roi_masked_median(const void* d_roi, int dtype, const uint8_t* d_mask, int64_t mask_stride_m,
                                    int64_t mask_stride_t, int m, int n_c, int n_t, int roi_len, double* d_median,
                                    void* stream) {
  if (!d_roi || !d_mask || !d_median || m < 0 || n_c <= 0 || n_t <= 0 || roi_len <= 0) return MG_EINVAL;
  if (mask_stride_m < 0 || mask_stride_t < 0 || (int64_t)n_c * n_t > 65535) return MG_EINVAL;
  if (m == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  switch (dtype) {
    case MG_U8: return launch_median<uint8_t>(d_roi, d_mask, mask_stride_m, mask_stride_t, m, n_c, n_t, roi_len, d_median, s);
    case MG_U16: return launch_median<uint16_t>(d_roi, d_mask, mask_stride_m, mask_stride_t, m, n_c, n_t, roi_len, d_median, s);
    case MG_F32: return launch_median<float>(d_roi, d_mask, mask_stride_m, mask_stride_t, m, n_c, n_t, roi_len, d_median, s);
    case MG_F64: return launch_median<double>(d_roi, d_mask, mask_stride_m, mask_stride_t, m, n_c, n_t, roi_len, d_median, s);
  }
  return MG_EINVAL;
}

extern "C" int mg_roi_masked_median_u16(const uint16_t* d_roi, const uint8_t* d_mask, int m, int n_c, int n_t,
                                        int roi_len, double* d_median, void* stream) {
  return mg_roi_masked_median(d_roi, MG_U16, d_mask, (int64_t)roi_len * roi_len, 0, m, n_c, n_t, roi_len, d_median, stream);
}
