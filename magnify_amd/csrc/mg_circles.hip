// A8-A11: RANSAC circle candidates (utils.py:295-344), filter_circles steps 4-6
// (utils.py:149-199), mean_grad scoring (utils.py:225-251) and the greedy claim-grid
// suppression of filter_neighbors (utils.py:254-292), re-formulated for a GPU:
//
//   candidates -> integer circles -> de-duplication in a bitmap (a circle's score depends
//   only on (row, col, r)) -> ordered compaction (= sort by r, row, col without sorting)
//   -> one thread per unique circle scores it with the reference's sequential float64 sum
//   -> parallel rounds of min-priority claims on the reference's own claim grid.
//
// Roofline: gather/latency bound (random reads of the angle map from L2 / Infinity Cache);
// reported separately from the HBM-streaming stages.
#include <math.h>

#include "mg_common.h"

namespace {

constexpr int NT = 256;

// ---- RNG: splitmix64 finaliser, top 32 bits -------------------------------------------------
__device__ __forceinline__ uint32_t draw32(uint64_t seed, uint64_t it, uint32_t k) {
  uint64_t z = seed + (it * 3ull + k + 1ull) * 0x9E3779B97F4A7C15ull;
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (uint32_t)(z >> 32);
}

// ---- K7: candidate circles --------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_candidates(const int32_t* __restrict__ d_coords, int64_t coord_cap,
                                                   const int32_t* __restrict__ d_starts,
                                                   const int32_t* __restrict__ d_counts,
                                                   const int32_t* __restrict__ d_num_edges, int h, int w, int grid,
                                                   int gc, int n_cells, const uint64_t* __restrict__ d_seeds,
                                                   int64_t num_iter, int min_r, int max_r,
                                                   uint32_t* __restrict__ d_bitmap, int64_t bitmap_words,
                                                   float* __restrict__ d_raw) {
  const int plane = blockIdx.y;
  const uint32_t n_edges = (uint32_t)d_num_edges[plane];
  if (n_edges == 0) return;
  const int32_t* coords = d_coords + (int64_t)plane * coord_cap * 2;
  const int32_t* starts = d_starts + (int64_t)plane * n_cells;
  const int32_t* counts = d_counts + (int64_t)plane * n_cells;
  const uint64_t seed = d_seeds[plane];
  const int hh = h + 2 * max_r, ww = w + 2 * max_r;
  uint32_t* bitmap = d_bitmap + (int64_t)plane * bitmap_words;
  const double eps = (double)1e-20f;
  for (int64_t it = (int64_t)blockIdx.x * NT + threadIdx.x; it < num_iter; it += (int64_t)gridDim.x * NT) {
    const uint32_t u0 = (uint32_t)(((uint64_t)draw32(seed, (uint64_t)it, 0) * n_edges) >> 32);
    const int p0r = coords[2 * (int64_t)u0], p0c = coords[2 * (int64_t)u0 + 1];
    const int cell = (p0r / grid) * gc + (p0c / grid);
    const uint32_t cnt = (uint32_t)counts[cell];
    const int64_t base = starts[cell];
    const int64_t i1 = base + (int64_t)(((uint64_t)draw32(seed, (uint64_t)it, 1) * cnt) >> 32);
    const int64_t i2 = base + (int64_t)(((uint64_t)draw32(seed, (uint64_t)it, 2) * cnt) >> 32);
    // p0-centred integer coordinates (utils.py:319-321); the slope numerator is negated as an
    // integer (as the reference does), so a zero stays +0.0
    const int d1r = coords[2 * i1] - p0r, d1c = coords[2 * i1 + 1] - p0c;
    const int d2r = coords[2 * i2] - p0r, d2c = coords[2 * i2 + 1] - p0c;
    const double q1r = (double)d1r, q1c = (double)d1c, q2r = (double)d2r, q2c = (double)d2c;
    // perpendicular bisectors (utils.py:326-334), float64
    const double m1 = (double)(-d1c) / (q1r + eps);
    const double m2 = (double)(-d2c) / (q2r + eps);
    const double b1 = 0.5 * q1r - m1 * (0.5 * q1c);
    const double b2 = 0.5 * q2r - m2 * (0.5 * q2c);
    // intersection, each store rounds to float32 (utils.py:337-342)
    const float c_col = (float)((b1 - b2) / (m2 - m1 + eps));
    const float c_row = (float)(m1 * (double)c_col + b1);
    const float rad = sqrtf(c_row * c_row + c_col * c_col);
    const float f_row = (float)((double)c_row + (double)p0r);
    const float f_col = (float)((double)c_col + (double)p0c);
    if (d_raw) {
      float* o = d_raw + ((int64_t)plane * num_iter + it) * 3;
      o[0] = f_row;
      o[1] = f_col;
      o[2] = rad;
    }
    // filter_circles step 4 (utils.py:157-166)
    if (!(rad >= (float)min_r && rad <= (float)max_r)) continue;
    const float rr = rintf(f_row), rc = rintf(f_col);
    if (!(fabsf(rr) < 1.0e9f && fabsf(rc) < 1.0e9f)) continue;  // cannot be on the image (and NaN)
    const int ir = (int)rr, ic = (int)rc, irad = (int)rintf(rad);
    if (ir + irad < 0 || ic + irad < 0 || ir - irad >= h || ic - irad >= w) continue;
    const int64_t bit = ((int64_t)(irad - min_r) * hh + (ir + max_r)) * ww + (ic + max_r);
    atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
  }
}

// ---- K8: bitmap -> ordered unique circle list ---------------------------------------------------
constexpr int WORDS_PER_BLOCK = 1024;  // 4 words per thread

__global__ __launch_bounds__(NT) void k_bitmap_count(const uint32_t* __restrict__ d_bitmap, int64_t bitmap_words,
                                                     int n_blocks, uint32_t* __restrict__ d_block_counts) {
  const int plane = blockIdx.y;
  const uint32_t* bm = d_bitmap + (int64_t)plane * bitmap_words;
  const int64_t w0 = (int64_t)blockIdx.x * WORDS_PER_BLOCK + threadIdx.x * 4;
  int c = 0;
  if (w0 + 4 <= bitmap_words && ((reinterpret_cast<uintptr_t>(bm + w0) & 15) == 0)) {
    const uint4 v = *reinterpret_cast<const uint4*>(bm + w0);
    c = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
  } else {
    for (int j = 0; j < 4; ++j)
      if (w0 + j < bitmap_words) c += __popc(bm[w0 + j]);
  }
  int total;
  mg_block_exscan(c, &total);
  if (threadIdx.x == 0) d_block_counts[(int64_t)plane * n_blocks + blockIdx.x] = (uint32_t)total;
}

__global__ __launch_bounds__(1024) void k_block_scan(uint32_t* __restrict__ d_block_counts, int n_blocks,
                                                     int32_t* __restrict__ d_num_circles, int64_t cap) {
  const int plane = blockIdx.x;
  uint32_t* cnt = d_block_counts + (int64_t)plane * n_blocks;
  int carry = 0;
  for (int base = 0; base < n_blocks; base += 1024) {
    const int i = base + threadIdx.x;
    const int v = i < n_blocks ? (int)cnt[i] : 0;
    int total;
    const int ex = mg_block_exscan(v, &total);
    if (i < n_blocks) cnt[i] = (uint32_t)(carry + ex);
    carry += total;
  }
  if (threadIdx.x == 0) d_num_circles[plane] = (int32_t)min((int64_t)carry, cap);
}

__global__ __launch_bounds__(NT) void k_bitmap_emit(uint32_t* __restrict__ d_bitmap, int64_t bitmap_words, int n_blocks,
                                                    const uint32_t* __restrict__ d_block_offsets, int h, int w,
                                                    int min_r, int max_r, int32_t* __restrict__ d_circles,
                                                    int64_t circle_cap) {
  const int plane = blockIdx.y;
  uint32_t* bm = d_bitmap + (int64_t)plane * bitmap_words;
  const int64_t w0 = (int64_t)blockIdx.x * WORDS_PER_BLOCK + threadIdx.x * 4;
  uint32_t v[4] = {0, 0, 0, 0};
  int c = 0;
  for (int j = 0; j < 4; ++j)
    if (w0 + j < bitmap_words) {
      v[j] = bm[w0 + j];
      c += __popc(v[j]);
    }
  int total;
  const int ex = mg_block_exscan(c, &total);
  if (total == 0) return;
  int64_t pos = (int64_t)d_block_offsets[(int64_t)plane * n_blocks + blockIdx.x] + ex;
  const int hh = h + 2 * max_r, ww = w + 2 * max_r;
  int32_t* out = d_circles + (int64_t)plane * circle_cap * 3;
  for (int j = 0; j < 4; ++j) {
    uint32_t bits = v[j];
    if (bits) bm[w0 + j] = 0;  // leave the bitmap clean for the next use
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      const int64_t idx = ((w0 + j) << 5) + b;
      const int col = (int)(idx % ww);
      const int64_t t = idx / ww;
      const int row = (int)(t % hh);
      const int rad = (int)(t / hh);
      if (pos < circle_cap) {
        out[3 * pos] = row - max_r;
        out[3 * pos + 1] = col - max_r;
        out[3 * pos + 2] = rad + min_r;
      }
      ++pos;
    }
  }
}

// ---- K9: scoring ----------------------------------------------------------------------------------
__global__ __launch_bounds__(NT) void k_score(const float* __restrict__ d_angle, int h, int w,
                                              const int32_t* __restrict__ d_circles, int64_t circle_cap,
                                              const int32_t* __restrict__ d_num_circles, int min_r,
                                              const int32_t* __restrict__ d_per_rc,
                                              const double* __restrict__ d_per_expected,
                                              const int32_t* __restrict__ d_per_starts, float min_roundness,
                                              float* __restrict__ d_scores, int32_t* __restrict__ d_alive,
                                              int32_t* __restrict__ d_num_alive, int32_t* __restrict__ d_max_rc) {
  const int plane = blockIdx.y;
  const int n = d_num_circles[plane];
  const float* ang = d_angle + (int64_t)plane * h * w;
  const int32_t* circles = d_circles + (int64_t)plane * circle_cap * 3;
  const double PI = 3.141592653589793;
  for (int64_t i = (int64_t)blockIdx.x * NT + threadIdx.x; i < n; i += (int64_t)gridDim.x * NT) {
    const int row = circles[3 * i], col = circles[3 * i + 1], rad = circles[3 * i + 2];
    const int p0 = d_per_starts[rad - min_r], p1 = d_per_starts[rad - min_r + 1];
    double acc = 0.0;
    for (int p = p0; p < p1; ++p) {
      const int rr = row + d_per_rc[2 * p], cc = col + d_per_rc[2 * p + 1];
      if (rr < 0 || rr >= h || cc < 0 || cc >= w) continue;  // zero padding: no edge (utils.py:172-174)
      const float a = ang[(int64_t)rr * w + cc];
      if (a == MG_NO_EDGE) continue;
      double d = fabs((double)a - d_per_expected[p]);
      if (d > PI) d = d - PI;
      acc += 4.0 * fabs(d - PI / 2.0) / PI - 1.0;
    }
    const float score = (float)acc / (float)(p1 - p0);
    d_scores[(int64_t)plane * circle_cap + i] = score;
    if (score >= min_roundness) {
      const int k = atomicAdd(&d_num_alive[plane], 1);
      d_alive[(int64_t)plane * circle_cap + k] = (int32_t)i;
      atomicMax(&d_max_rc[2 * plane], row);
      atomicMax(&d_max_rc[2 * plane + 1], col);
    }
  }
}

// ---- K10: greedy suppression in parallel rounds ------------------------------------------------------
// Priority key: smaller = earlier in the reference's score-descending order; ties broken by
// the canonical (r, row, col) index.
__device__ __forceinline__ uint64_t nms_key(float score, uint32_t idx) {
  uint32_t b = __float_as_uint(score);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending-sortable
  return ((uint64_t)(~b) << 32) | idx;             // descending score
}

__device__ __forceinline__ int wrap(int v, int n) {
  v %= n;
  return v < 0 ? v + n : v;
}

template <int PHASE>
__global__ __launch_bounds__(NT) void k_nms(const int32_t* __restrict__ d_circles, int64_t circle_cap,
                                            const float* __restrict__ d_scores, const int32_t* __restrict__ d_alive,
                                            const int32_t* __restrict__ d_num_alive,
                                            const int32_t* __restrict__ d_max_rc, int min_dist,
                                            const int32_t* __restrict__ d_ring_rc, int ring_len,
                                            uint64_t* __restrict__ d_grid, int64_t grid_cap,
                                            uint8_t* __restrict__ d_state, int32_t* __restrict__ d_undecided) {
  const int plane = blockIdx.y;
  const int n = d_num_alive[plane];
  if (n == 0) return;
  const int pad = 2 * min_dist + 1;
  const int n_rows = d_max_rc[2 * plane] + 2 * pad, n_cols = d_max_rc[2 * plane + 1] + 2 * pad;
  if ((int64_t)n_rows * n_cols > grid_cap) return;  // caller sized the grid from the image extent
  const int32_t* circles = d_circles + (int64_t)plane * circle_cap * 3;
  const float* scores = d_scores + (int64_t)plane * circle_cap;
  const int32_t* alive = d_alive + (int64_t)plane * circle_cap;
  uint8_t* state = d_state + (int64_t)plane * circle_cap;
  uint64_t* grid = d_grid + (int64_t)plane * grid_cap;
  for (int64_t a = (int64_t)blockIdx.x * NT + threadIdx.x; a < n; a += (int64_t)gridDim.x * NT) {
    const int idx = alive[a];
    if (state[idx] != 0) continue;
    const int row = circles[3 * (int64_t)idx], col = circles[3 * (int64_t)idx + 1];
    const uint64_t key = nms_key(scores[idx], (uint32_t)idx);
    if (PHASE == 0) {  // claim: every undecided circle bids for its ring pixels
      for (int j = 0; j < ring_len; ++j) {
        const int rr = wrap(d_ring_rc[2 * j] + row + pad, n_rows), cc = wrap(d_ring_rc[2 * j + 1] + col + pad, n_cols);
        atomicMin(reinterpret_cast<unsigned long long*>(&grid[(int64_t)rr * n_cols + cc]), (unsigned long long)key);
      }
    } else {  // decide
      bool all_mine = true, hit_kept = false;
      for (int j = 0; j < ring_len; ++j) {
        const int rr = wrap(d_ring_rc[2 * j] + row + pad, n_rows), cc = wrap(d_ring_rc[2 * j + 1] + col + pad, n_cols);
        const uint64_t g = grid[(int64_t)rr * n_cols + cc];
        if (g != key) {
          all_mine = false;
          if (g != ~0ull && reinterpret_cast<volatile uint8_t*>(state)[(uint32_t)g] == 1)
            hit_kept = true;
        }
      }
      if (all_mine) {
        reinterpret_cast<volatile uint8_t*>(state)[idx] = 1;
      } else if (hit_kept) {
        reinterpret_cast<volatile uint8_t*>(state)[idx] = 2;
        // withdraw this circle's bids so that later circles can win these pixels
        for (int j = 0; j < ring_len; ++j) {
          const int rr = wrap(d_ring_rc[2 * j] + row + pad, n_rows), cc = wrap(d_ring_rc[2 * j + 1] + col + pad, n_cols);
          atomicCAS(reinterpret_cast<unsigned long long*>(&grid[(int64_t)rr * n_cols + cc]), (unsigned long long)key,
                    ~0ull);
        }
      } else {
        atomicAdd(&d_undecided[plane], 1);
      }
    }
  }
}

// ---- K11: collect kept circles in priority order -------------------------------------------------------
__global__ __launch_bounds__(NT) void k_collect_list(const int32_t* __restrict__ d_alive,
                                                     const int32_t* __restrict__ d_num_alive,
                                                     const uint8_t* __restrict__ d_state, int keep_all,
                                                     int64_t circle_cap, int64_t out_cap,
                                                     int32_t* __restrict__ d_scratch, int32_t* __restrict__ d_num_out) {
  const int plane = blockIdx.y;
  const int n = d_num_alive[plane];
  for (int64_t a = (int64_t)blockIdx.x * NT + threadIdx.x; a < n; a += (int64_t)gridDim.x * NT) {
    const int idx = d_alive[(int64_t)plane * circle_cap + a];
    if (keep_all || d_state[(int64_t)plane * circle_cap + idx] == 1) {
      const int k = atomicAdd(&d_num_out[plane], 1);
      if (k < out_cap) d_scratch[(int64_t)plane * out_cap + k] = idx;
    }
  }
}

__global__ __launch_bounds__(NT) void k_collect_rank(const int32_t* __restrict__ d_circles, int64_t circle_cap,
                                                     const float* __restrict__ d_scores,
                                                     const int32_t* __restrict__ d_scratch,
                                                     int32_t* __restrict__ d_num_out, int64_t out_cap,
                                                     int32_t* __restrict__ d_out, float* __restrict__ d_out_scores) {
  const int plane = blockIdx.y;
  const int m = (int)min((int64_t)d_num_out[plane], out_cap);
  const int32_t* list = d_scratch + (int64_t)plane * out_cap;
  const float* scores = d_scores + (int64_t)plane * circle_cap;
  for (int64_t a = (int64_t)blockIdx.x * NT + threadIdx.x; a < m; a += (int64_t)gridDim.x * NT) {
    const int idx = list[a];
    const uint64_t key = nms_key(scores[idx], (uint32_t)idx);
    int rank = 0;
    for (int b = 0; b < m; ++b) {
      const int j = list[b];
      rank += nms_key(scores[j], (uint32_t)j) < key;
    }
    const int32_t* c = d_circles + ((int64_t)plane * circle_cap + idx) * 3;
    int32_t* o = d_out + ((int64_t)plane * out_cap + rank) * 3;
    o[0] = c[0];
    o[1] = c[1];
    o[2] = c[2];
    if (d_out_scores) d_out_scores[(int64_t)plane * out_cap + rank] = scores[idx];
  }
}

__global__ void k_clamp_counts(int32_t* d_num_out, int n_planes, int64_t out_cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_planes && d_num_out[i] > out_cap) d_num_out[i] = (int32_t)out_cap;
}

inline int grid_x(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + NT - 1) / NT, 8192)); }

}  // namespace

extern "C" int mg_candidate_circles(const int32_t* d_coords, int64_t coord_cap, const int32_t* d_cell_starts,
                                    const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h,
                                    int w, int grid, const uint64_t* d_seeds, int64_t num_iter, int min_r, int max_r,
                                    uint32_t* d_bitmap, int64_t bitmap_words, float* d_raw, void* stream) {
  if (!d_coords || !d_cell_starts || !d_cell_counts || !d_num_edges || !d_seeds || !d_bitmap) return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || h <= 0 || w <= 0 || grid <= 0 || num_iter < 0 || min_r < 0 || max_r < min_r)
    return MG_EINVAL;
  const int64_t need_bits = (int64_t)(max_r - min_r + 1) * (h + 2 * max_r) * (w + 2 * max_r);
  if (bitmap_words * 32 < need_bits) return MG_EINVAL;
  if (n_planes == 0 || num_iter == 0) return MG_OK;
  const int gr = (h + grid - 1) / grid, gc = (w + grid - 1) / grid;
  hipLaunchKernelGGL(k_candidates, dim3(grid_x(num_iter), n_planes), dim3(NT), 0, mg_stream(stream), d_coords,
                     coord_cap, d_cell_starts, d_cell_counts, d_num_edges, h, w, grid, gc, gr * gc, d_seeds, num_iter,
                     min_r, max_r, d_bitmap, bitmap_words, d_raw);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_bitmap_to_circles(uint32_t* d_bitmap, int64_t bitmap_words, int n_planes, int h, int w, int min_r,
                                    int max_r, uint32_t* d_block_counts, int32_t* d_circles, int64_t circle_cap,
                                    int32_t* d_num_circles, void* stream) {
  if (!d_bitmap || !d_block_counts || !d_circles || !d_num_circles || n_planes < 0 || n_planes > 65535 ||
      bitmap_words <= 0 || circle_cap < 0)
    return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  const int64_t nb64 = (bitmap_words + WORDS_PER_BLOCK - 1) / WORDS_PER_BLOCK;
  if (nb64 > 0x7FFFFFFF) return MG_EINVAL;
  const int nb = (int)nb64;
  hipStream_t s = mg_stream(stream);
  hipLaunchKernelGGL(k_bitmap_count, dim3(nb, n_planes), dim3(NT), 0, s, d_bitmap, bitmap_words, nb, d_block_counts);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_block_scan, dim3(n_planes), dim3(1024), 0, s, d_block_counts, nb, d_num_circles, circle_cap);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_bitmap_emit, dim3(nb, n_planes), dim3(NT), 0, s, d_bitmap, bitmap_words, nb, d_block_counts, h,
                     w, min_r, max_r, d_circles, circle_cap);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_score_circles(const float* d_angle, int n_planes, int h, int w, const int32_t* d_circles,
                                int64_t circle_cap, const int32_t* d_num_circles, int min_r, int max_r,
                                const int32_t* d_per_rc, const double* d_per_expected, const int32_t* d_per_starts,
                                float min_roundness, float* d_scores, int32_t* d_alive, int32_t* d_num_alive,
                                int32_t* d_max_rc, void* stream) {
  if (!d_angle || !d_circles || !d_num_circles || !d_per_rc || !d_per_expected || !d_per_starts || !d_scores ||
      !d_alive || !d_num_alive || !d_max_rc)
    return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || max_r < min_r) return MG_EINVAL;
  if (n_planes == 0 || circle_cap == 0) return MG_OK;
  hipLaunchKernelGGL(k_score, dim3(grid_x(circle_cap), n_planes), dim3(NT), 0, mg_stream(stream), d_angle, h, w,
                     d_circles, circle_cap, d_num_circles, min_r, d_per_rc, d_per_expected, d_per_starts,
                     min_roundness, d_scores, d_alive, d_num_alive, d_max_rc);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_nms_round(const int32_t* d_circles, int64_t circle_cap, const float* d_scores,
                            const int32_t* d_alive, const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes,
                            int min_dist, const int32_t* d_ring_rc, int ring_len, uint64_t* d_grid, int64_t grid_cap,
                            uint8_t* d_state, int32_t* d_undecided, void* stream) {
  if (!d_circles || !d_scores || !d_alive || !d_num_alive || !d_max_rc || !d_ring_rc || !d_grid || !d_state ||
      !d_undecided)
    return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || min_dist <= 0 || ring_len <= 0) return MG_EINVAL;
  if (n_planes == 0 || circle_cap == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  if (hipMemsetAsync(d_undecided, 0, sizeof(int32_t) * n_planes, s) != hipSuccess) return MG_ELAUNCH;
  const dim3 g(grid_x(circle_cap), n_planes);
  hipLaunchKernelGGL((k_nms<0>), g, dim3(NT), 0, s, d_circles, circle_cap, d_scores, d_alive, d_num_alive, d_max_rc,
                     min_dist, d_ring_rc, ring_len, d_grid, grid_cap, d_state, d_undecided);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL((k_nms<1>), g, dim3(NT), 0, s, d_circles, circle_cap, d_scores, d_alive, d_num_alive, d_max_rc,
                     min_dist, d_ring_rc, ring_len, d_grid, grid_cap, d_state, d_undecided);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_collect_circles(const int32_t* d_circles, int64_t circle_cap, const float* d_scores,
                                  const int32_t* d_alive, const int32_t* d_num_alive, const uint8_t* d_state,
                                  int keep_all, int n_planes, int32_t* d_out, float* d_out_scores, int64_t out_cap,
                                  int32_t* d_num_out, int32_t* d_scratch, void* stream) {
  if (!d_circles || !d_scores || !d_alive || !d_num_alive || !d_out || !d_num_out || !d_scratch) return MG_EINVAL;
  if (!keep_all && !d_state) return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || out_cap < 0) return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  if (hipMemsetAsync(d_num_out, 0, sizeof(int32_t) * n_planes, s) != hipSuccess) return MG_ELAUNCH;
  if (circle_cap == 0 || out_cap == 0) return MG_OK;
  hipLaunchKernelGGL(k_collect_list, dim3(grid_x(circle_cap), n_planes), dim3(NT), 0, s, d_alive, d_num_alive, d_state,
                     keep_all, circle_cap, out_cap, d_scratch, d_num_out);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_collect_rank, dim3(grid_x(out_cap), n_planes), dim3(NT), 0, s, d_circles, circle_cap, d_scores,
                     d_scratch, d_num_out, out_cap, d_out, d_out_scores);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_clamp_counts, dim3((n_planes + 255) / 256), dim3(256), 0, s, d_num_out, n_planes, out_cap);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
