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
#include <stdlib.h>

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

// The finaliser alone: draw32(seed, it, k) == mix_top32(seed + (3 it + k + 1) * golden).
constexpr uint64_t GOLDEN = 0x9E3779B97F4A7C15ull;
constexpr uint32_t MG_NO_KEY = 0xFFFFFFFFu;  // candidate rejected by the radius / on-image filter
__device__ __forceinline__ uint64_t mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
__device__ __forceinline__ uint32_t mix_top32(uint64_t z) { return (uint32_t)(mix64(z) >> 32); }
// The three uniforms of iteration `it` (oracle/ref_numeric.py draw_uniform32): u0 = top half of mix64(seed + (3 it + 1) golden),
// u1 and u2 = top and BOTTOM half of mix64(seed + (3 it + 2) golden) -- round 4: one finaliser for the two picks inside
// p0's cell (the finaliser's 64-bit multiplies were a quarter of the candidate kernel's vector time).

// floor(x / d) for an integer-valued double 0 <= x < 2^52 and d < 2^31 with floor(x / d) < 2^32:
// float64 only (full rate), no 64-bit integer multiply.  r = x - q d is exact (a small integer).
__device__ __forceinline__ uint32_t div_f64(double x, double d, double inv_d) {
  uint32_t q = (uint32_t)(x * inv_d);
  const double r = fma(-(double)q, d, x);
  if (r < 0.0) --q;
  else if (r >= d) ++q;
  return q;
}

// floor(n / d) for n < 2^52 via a float64 reciprocal and one correction step (exact).
__device__ __forceinline__ uint64_t div_u64(uint64_t n, uint64_t d, double inv_d) {
  uint64_t q = (uint64_t)((double)n * inv_d);
  int64_t r = (int64_t)(n - q * d);
  if (r < 0) {
    --q;
    r += (int64_t)d;
  }
  if (r >= (int64_t)d) ++q;
  return q;
}

// ---- K7: candidate circles --------------------------------------------------------------------
// FAST: num_iter < 2^31 and h, w <= 65536 and grid > 1 (checked by the launcher).
template <bool FAST>
__global__ __launch_bounds__(NT) void k_candidates(const int32_t* __restrict__ d_coords, int64_t coord_cap,
                                                   const int32_t* __restrict__ d_starts,
                                                   const int32_t* __restrict__ d_counts,
                                                   const int32_t* __restrict__ d_num_edges, int h, int w, int grid,
                                                   int gc, int n_cells, const uint64_t* __restrict__ d_seeds,
                                                   int64_t num_iter, int min_r, int max_r,
                                                   uint32_t* __restrict__ d_bitmap, int64_t bitmap_words,
                                                   float* __restrict__ d_raw, uint32_t* __restrict__ d_keys) {
  const int plane = blockIdx.y;
  const uint32_t n_edges = (uint32_t)d_num_edges[plane];
  if (n_edges == 0) return;
  const int32_t* coords = d_coords + (int64_t)plane * coord_cap * 2;
  const int32_t* starts = d_starts + (int64_t)plane * n_cells;
  const int32_t* counts = d_counts + (int64_t)plane * n_cells;
  const uint64_t seed = d_seeds[plane];
  const int ntc = (w + 2 * max_r + 63) >> 6, nr = max_r - min_r + 1;
  uint32_t* bitmap = d_bitmap ? d_bitmap + (int64_t)plane * bitmap_words : nullptr;
  uint32_t* keys = d_keys ? d_keys + (int64_t)plane * num_iter : nullptr;
  const double eps = (double)1e-20f;
  const double inv_iter = 1.0 / (double)num_iter;
  // strength-reduced bookkeeping (same values as draw32 / floor(it E / K), fewer quarter-rate
  // integer multiplies): the RNG counter advances by a constant per loop trip; E = eq K + er splits
  // the stratum bound into a 32-bit product and a float64 quotient; cells by a magic multiply
  const int64_t it0 = (int64_t)blockIdx.x * NT + threadIdx.x, stride = (int64_t)gridDim.x * NT;
  uint64_t zbase = seed + ((uint64_t)it0 * 3ull + 1ull) * GOLDEN;
  const uint64_t zstep = (uint64_t)stride * 3ull * GOLDEN;
  constexpr bool small = FAST, gfast = FAST;
  const uint32_t eq = small ? n_edges / (uint32_t)num_iter : 0u;
  const double er = small ? (double)(n_edges - eq * (uint32_t)num_iter) : 0.0, dk = (double)num_iter;
  const uint32_t gmagic = grid > 1 ? 0xFFFFFFFFu / (uint32_t)grid + 1u : 0u;  // exact for operands < 2^16
  for (int64_t it = it0; it < num_iter; it += stride, zbase += zstep) {
    // jittered stratified p0: iteration it owns the slice [a, b) of the cell-major edge list
    uint64_t sa, sb;
    if (small) {
      const double x = (double)(uint32_t)it * er;  // exact: < 2^52
      sa = (uint64_t)((uint32_t)it * eq + div_f64(x, dk, inv_iter));
      sb = (uint64_t)(((uint32_t)it + 1u) * eq + div_f64(x + er, dk, inv_iter));
    } else {
      sa = div_u64((uint64_t)it * n_edges, (uint64_t)num_iter, inv_iter);
      sb = div_u64(((uint64_t)it + 1) * n_edges, (uint64_t)num_iter, inv_iter);
    }
    const uint64_t width = sb > sa ? sb - sa : 1;
    const uint32_t u0 = (uint32_t)(sa + (((uint64_t)mix_top32(zbase) * width) >> 32));
    const int p0r = coords[2 * (int64_t)u0], p0c = coords[2 * (int64_t)u0 + 1];
    const int cell = gfast ? (int)(__umulhi((uint32_t)p0r, gmagic) * (uint32_t)gc + __umulhi((uint32_t)p0c, gmagic))
                           : (p0r / grid) * gc + (p0c / grid);
    const uint32_t cnt = (uint32_t)counts[cell];
    const int64_t base = starts[cell];
    const uint64_t z12 = mix64(zbase + GOLDEN);
    const int64_t i1 = base + (int64_t)__umulhi((uint32_t)(z12 >> 32), cnt);
    const int64_t i2 = base + (int64_t)__umulhi((uint32_t)z12, cnt);
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
    uint32_t key = MG_NO_KEY;
    do {
      if (!(rad >= (float)min_r && rad <= (float)max_r)) break;
      const float rr = rintf(f_row), rc = rintf(f_col);
      if (!(fabsf(rr) < 1.0e9f && fabsf(rc) < 1.0e9f)) break;  // cannot be on the image (and NaN)
      const int ir = (int)rr, ic = (int)rc, irad = (int)rintf(rad);
      if (ir + irad < 0 || ic + irad < 0 || ir - irad >= h || ic - irad >= w) break;
      // tile-major de-duplication: 64 x 64 tiles of the padded centre grid, one 4096-bit layer per
      // (tile, radius); the ordered compaction then emits circles grouped by tile
      const int pr = ir + max_r, pc = ic + max_r;
      const int tile = (pr >> 6) * ntc + (pc >> 6);
      if (keys) {
        key = ((uint32_t)tile << 17) | ((uint32_t)(irad - min_r) << 12) | (uint32_t)(((pr & 63) << 6) + (pc & 63));
      } else {
        const int64_t bit = (((int64_t)tile * nr + (irad - min_r)) << 12) + ((pr & 63) << 6) + (pc & 63);
        atomicOr(&bitmap[bit >> 5], 1u << (bit & 31));
      }
    } while (false);
    if (keys) keys[it] = key;  // coalesced plain store; k_tile_dedup builds the bitmap tile by tile in LDS
  }
}

// ---- K7b: the same candidates, cheaper (round 4) ------------------------------------------------
// What the iteration of k_candidates spends its vector instructions on, and what is done about it here:
//  * p1 and p2 come from p0's grid cell, so (d_row, d_col) of a picked point relative to p0 is one of (2 grid - 1)^2
//    integer pairs: the bisector's slope m = -d_col / (d_row + eps) and intercept b = d_row / 2 - m d_col / 2
//    (utils.py:326-334) are TABLE ENTRIES -- computed once per workgroup with the reference's own float64 operations
//    (the same division instruction sequence, so the same bits) and kept in LDS: two of the three float64 divisions
//    and six float64 multiplies / subtracts per iteration become two 16-byte LDS reads;
//  * the third division (the intersection, general operands) keeps the hardware's reciprocal / Newton / residual
//    sequence but drops its range scaling (v_div_scale / v_div_fixup): the operands here lie within 2^+-140, where the
//    scaling is the identity and the remaining operations are instruction for instruction those of `/`;
//    a zero, infinite or NaN divisor takes `/` itself;
//  * the stratum bounds floor(it E / K), floor((it + 1) E / K) are carried from iteration to iteration as
//    (quotient, remainder) pairs -- integer adds and compares instead of two float64 quotients with corrections;
//  * when the stratum holds one edge (E <= K: the usual case) p0 IS that edge and the first hash is not drawn at all
//    (its product with a width of 1 is 0 whatever the hash).
// Bit-identical to k_candidates<true> (tests/test_gpu_kernels.py compares the raw float32 triples with the oracle's,
// NaN / inf included, and both kernels with each other).
constexpr int CT = 1024;
constexpr int MAX_TAB_GRID = 32;  // (2 * 32 - 1)^2 * 16 B = 63.5 KB of LDS

__device__ __forceinline__ double div_unscaled(double num, double den) {
  if (!(fabs(den) > 0.0 && fabs(den) < 1.0e300)) return num / den;  // 0, inf, NaN: the full sequence
  double y = __builtin_amdgcn_rcp(den);
  double e = fma(-den, y, 1.0);
  y = fma(y, e, y);
  e = fma(-den, y, 1.0);
  y = fma(y, e, y);
  double q = num * y;
  const double r = fma(-den, q, num);
  return fma(r, y, q);
}

__global__ __launch_bounds__(CT) void k_candidates_tab(const int32_t* __restrict__ d_coords, int64_t coord_cap,
                                                       const int32_t* __restrict__ d_starts,
                                                       const int32_t* __restrict__ d_counts,
                                                       const int32_t* __restrict__ d_num_edges, int h, int w, int grid,
                                                       int gc, int n_cells, const uint64_t* __restrict__ d_seeds,
                                                       uint32_t num_iter, int min_r, int max_r,
                                                       float* __restrict__ d_raw, uint32_t* __restrict__ d_keys) {
  extern __shared__ double2 s_tab[];  // [(2 grid - 1)^2]: (m, b) of the bisector of p0 and p0 + (d_row, d_col)
  const int plane = blockIdx.y;
  const uint32_t n_edges = (uint32_t)d_num_edges[plane];
  if (n_edges == 0) return;
  const double eps = (double)1e-20f;
  const int g1 = grid - 1, span = 2 * grid - 1;
  for (int i = threadIdx.x; i < span * span; i += CT) {
    const int dr = i / span - g1, dc = i - (i / span) * span - g1;
    const double m = (double)(-dc) / ((double)dr + eps);
    const double b = 0.5 * (double)dr - m * (0.5 * (double)dc);
    s_tab[i] = make_double2(m, b);
  }
  __syncthreads();
  const int32_t* coords = d_coords + (int64_t)plane * coord_cap * 2;
  const int32_t* starts = d_starts + (int64_t)plane * n_cells;
  const int32_t* counts = d_counts + (int64_t)plane * n_cells;
  const uint64_t seed = d_seeds[plane];
  const int ntc = (w + 2 * max_r + 63) >> 6;
  uint32_t* keys = d_keys ? d_keys + (int64_t)plane * num_iter : nullptr;
  const uint32_t K = num_iter;
  const uint32_t it0 = blockIdx.x * CT + threadIdx.x, stride = gridDim.x * CT;
  if (it0 >= K) return;
  // E = eq K + er;  it E = sa K + rem  with  sa = it eq + floor(it er / K),  rem = (it er) mod K
  const uint32_t eq = n_edges / K, er = n_edges - eq * K;
  const double inv_k = 1.0 / (double)K;
  const uint64_t x0 = (uint64_t)it0 * er;                     // < K E < 2^52 (launcher)
  const uint64_t q0 = div_u64(x0, K, inv_k);
  uint32_t sa = it0 * eq + (uint32_t)q0, rem = (uint32_t)(x0 - q0 * K);
  const uint64_t xs = (uint64_t)stride * er;                  // < 2^20 K
  const uint64_t qs = div_u64(xs, K, inv_k);
  const uint32_t sa_step = stride * eq + (uint32_t)qs, rem_step = (uint32_t)(xs - qs * K);
  uint64_t zbase = seed + ((uint64_t)it0 * 3ull + 1ull) * GOLDEN;
  const uint64_t zstep = (uint64_t)stride * 3ull * GOLDEN;
  const uint32_t gmagic = 0xFFFFFFFFu / (uint32_t)grid + 1u;  // exact for operands < 2^16
  for (uint32_t it = it0; it < K; it += stride, zbase += zstep) {
    // jittered stratified p0: iteration it owns the slice [sa, sb) of the cell-major edge list
    const uint32_t sb = sa + eq + (rem + er >= K ? 1u : 0u);  // (rem, er < K < 2^31: no wrap)
    uint32_t u0 = sa;
    if (sb > sa + 1u) u0 = sa + __umulhi(mix_top32(zbase), sb - sa);
    {  // the next iteration's (sa, rem)
      const uint32_t t = rem + rem_step;
      const bool carry = t >= K;
      sa += sa_step + (carry ? 1u : 0u);
      rem = carry ? t - K : t;
    }
    const int2 p0 = reinterpret_cast<const int2*>(coords)[u0];
    const int p0r = p0.x, p0c = p0.y;
    const int cell = (int)(__umulhi((uint32_t)p0r, gmagic) * (uint32_t)gc + __umulhi((uint32_t)p0c, gmagic));
    const uint32_t cnt = (uint32_t)counts[cell];
    const uint32_t base = (uint32_t)starts[cell];
    const uint64_t z12 = mix64(zbase + GOLDEN);
    const uint32_t i1 = base + __umulhi((uint32_t)(z12 >> 32), cnt);
    const uint32_t i2 = base + __umulhi((uint32_t)z12, cnt);
    const int2 p1 = reinterpret_cast<const int2*>(coords)[i1], p2 = reinterpret_cast<const int2*>(coords)[i2];
    const int d1r = p1.x - p0r, d1c = p1.y - p0c, d2r = p2.x - p0r, d2c = p2.y - p0c;
    double m1, b1, m2, b2;
    if ((uint32_t)(d1r + g1) < (uint32_t)span && (uint32_t)(d1c + g1) < (uint32_t)span &&
        (uint32_t)(d2r + g1) < (uint32_t)span && (uint32_t)(d2c + g1) < (uint32_t)span) {
      const double2 t1 = s_tab[(d1r + g1) * span + d1c + g1], t2 = s_tab[(d2r + g1) * span + d2c + g1];
      m1 = t1.x, b1 = t1.y, m2 = t2.x, b2 = t2.y;
    } else {  // (cannot happen with a cell-major list made by mg_edge_grid; kept so that no list can index beyond the table)
      m1 = (double)(-d1c) / ((double)d1r + eps);
      m2 = (double)(-d2c) / ((double)d2r + eps);
      b1 = 0.5 * (double)d1r - m1 * (0.5 * (double)d1c);
      b2 = 0.5 * (double)d2r - m2 * (0.5 * (double)d2c);
    }
    // intersection, each store rounds to float32 (utils.py:337-342)
    const float c_col = (float)div_unscaled(b1 - b2, m2 - m1 + eps);
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
    uint32_t key = MG_NO_KEY;
    do {
      if (!(rad >= (float)min_r && rad <= (float)max_r)) break;
      const float rr = rintf(f_row), rc = rintf(f_col);
      if (!(fabsf(rr) < 1.0e9f && fabsf(rc) < 1.0e9f)) break;  // cannot be on the image (and NaN)
      const int ir = (int)rr, ic = (int)rc, irad = (int)rintf(rad);
      if (ir + irad < 0 || ic + irad < 0 || ir - irad >= h || ic - irad >= w) break;
      const int pr = ir + max_r, pc = ic + max_r;
      const int tile = (pr >> 6) * ntc + (pc >> 6);
      key = ((uint32_t)tile << 17) | ((uint32_t)(irad - min_r) << 12) | (uint32_t)(((pr & 63) << 6) + (pc & 63));
    } while (false);
    if (keys) keys[it] = key;
  }
}

// ---- K8: bitmap -> ordered unique circle list ---------------------------------------------------
// One wave per 4096-bit layer (= one radius of one 64 x 64 centre tile): lane l owns tile row l
// (64 bits).  Emission order = (tile_row, tile_col, r, row, col): the build's canonical order.
constexpr int TS = 64;
constexpr int LAYER_WORDS = 128;

__global__ __launch_bounds__(NT) void k_layer_count(const uint32_t* __restrict__ d_bitmap, int64_t bitmap_words,
                                                    int n_layers, int32_t* __restrict__ d_layer_offsets) {
  const int plane = blockIdx.y;
  const int layer = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (layer >= n_layers) return;
  const int lane = threadIdx.x & 63;
  const uint2 v = reinterpret_cast<const uint2*>(d_bitmap + (int64_t)plane * bitmap_words +
                                                 (int64_t)layer * LAYER_WORDS)[lane];
  const int c = mg_wave_sum_i32(__popc(v.x) + __popc(v.y));
  if (lane == 0) d_layer_offsets[(int64_t)plane * (n_layers + 1) + layer] = c;
}

// Keyed de-duplication (no global atomics): candidates were written as 32-bit keys
// (tile << 17 | radius layer << 12 | position in the 64 x 64 tile) in iteration order.  p0 is a
// stratified draw over the cell-major edge list, so the iterations that can put a centre into a
// given tile are a handful of contiguous iteration ranges -- one per cell row within reach of the
// tile.  One workgroup per tile scans those ranges, sets the bits of its own keys in an LDS copy of
// the tile's layers, counts every layer and stores the layers with plain coalesced stores.
// (Global atomics execute at the memory side on this chip: 320 M scattered atomicOr cost 3.8 ms of
// the 6.1 ms candidates kernel.)
constexpr int MAX_RANGES = 64;
constexpr int DEDUP_TX = 2;

// TX = adjacent tiles of one tile row handled by a workgroup: the scanned cell ranges of neighbours
// overlap by 2 * (max_r + 2) pixels, so wider groups read every key fewer times.
template <int TX>
__global__ __launch_bounds__(NT) void k_tile_dedup(const uint32_t* __restrict__ d_keys, int64_t num_iter,
                                                   const int32_t* __restrict__ d_starts,
                                                   const int32_t* __restrict__ d_counts,
                                                   const int32_t* __restrict__ d_num_edges, int h, int w, int grid,
                                                   int gr, int gc, int ntc, int nr, int max_r,
                                                   uint32_t* __restrict__ d_ukeys, int64_t circle_cap,
                                                   int32_t* __restrict__ d_tile_ranges, int n_tiles,
                                                   int32_t* __restrict__ d_num_circles,
                                                   int32_t* __restrict__ d_layer_starts) {
  extern __shared__ uint32_t lbits[];  // [TX][nr][LAYER_WORDS]
  __shared__ long long s_lo[MAX_RANGES];
  __shared__ int s_pre[MAX_RANGES + 1];
  __shared__ int s_cnt[TX * 32 + 1];  // circles per (tile, layer), then their exclusive prefix (nr <= 32)
  __shared__ int s_base;
  const int plane = blockIdx.z, tr = blockIdx.y, tc0 = blockIdx.x * TX;
  const int ntx = min(TX, ntc - tc0);  // tiles of this group that exist
  const int tile0 = tr * ntc + tc0;
  const int words = nr * LAYER_WORDS;  // per tile
  for (int i = threadIdx.x; i < ntx * words; i += NT) lbits[i] = 0u;
  const long long n_edges = d_num_edges[plane];
  // p0 lies on the circle: within max_r + 1 of the rounded centre (one more for safety)
  const int reach = max_r + 2;
  const int y0 = max(tr * TS - max_r - reach, 0), y1 = min(tr * TS + TS - 1 - max_r + reach, h - 1);
  const int x0 = max(tc0 * TS - max_r - reach, 0), x1 = min((tc0 + ntx) * TS - 1 - max_r + reach, w - 1);
  int n_ranges = 0;
  if (n_edges > 0 && y0 <= y1 && x0 <= x1) {
    const int cr0 = y0 / grid, cr1 = y1 / grid, cc0 = x0 / grid, cc1 = x1 / grid;
    n_ranges = cr1 - cr0 + 1;  // <= MAX_RANGES (checked by the launcher)
    if ((int)threadIdx.x < n_ranges) {
      const int32_t* starts = d_starts + (int64_t)plane * gr * gc;
      const int32_t* counts = d_counts + (int64_t)plane * gr * gc;
      const int cr = cr0 + threadIdx.x;
      const long long ea = starts[cr * gc + cc0], eb = (long long)starts[cr * gc + cc1] + counts[cr * gc + cc1];
      // iterations whose stratum [floor(i E / K), floor((i + 1) E / K)) can touch [ea, eb): a
      // conservative superset (keys carry their own tile, foreign ones are skipped)
      long long lo = 0, hi = 0;
      if (eb > ea) {
        const double kpe = (double)num_iter / (double)n_edges;
        hi = min((long long)((double)eb * kpe) + 3, (long long)num_iter);
        lo = min(max((long long)((double)ea * kpe) - 2, 0ll), hi);
      }
      s_lo[threadIdx.x] = lo;
      s_pre[threadIdx.x + 1] = (int)(hi - lo);
    }
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s_pre[0] = 0;
    for (int r = 0; r < n_ranges; ++r) s_pre[r + 1] += s_pre[r];
  }
  __syncthreads();
  const uint32_t* keys = d_keys + (int64_t)plane * num_iter;
  // Range by range (a handful per workgroup), 8 key loads in flight per thread: inside a range the index
  // arithmetic is 32-bit and loop-invariant (no per-key search for the range a key index belongs to).
  constexpr int KB = 8;
  for (int r = 0; r < n_ranges; ++r) {
    const uint32_t* kr = keys + s_lo[r];
    const int len = s_pre[r + 1] - s_pre[r];
    for (int base = threadIdx.x; base < len; base += NT * KB) {
      uint32_t kv[KB];
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const int i = base + u * NT;
        kv[u] = i < len ? kr[i] : MG_NO_KEY;
      }
#pragma unroll
      for (int u = 0; u < KB; ++u) {
        const uint32_t key = kv[u];
        const uint32_t t = (key >> 17) - (uint32_t)tile0;  // wraps for foreign (and rejected) keys
        if (t < (uint32_t)ntx) atomicOr(&lbits[t * words + ((key & 0x1FFFFu) >> 5)], 1u << (key & 31u));
      }
    }
  }
  __syncthreads();
  // Ordered emission straight from LDS.  The bit index inside a tile IS the low 17 bits of the key
  // ((r << 12) | (row << 6) | col), so walking the tile's words in order yields its keys in ascending order.
  // Every thread owns a run of consecutive words: count its bits, one block-wide exclusive scan of the counts
  // (its entries at the layer boundaries are the per-(tile, radius) starts), reserve the group's slice of the
  // plane's list with one atomicAdd (d_num_circles is the cursor), then every thread stores the keys of its run.
  // (A wave per sparse layer -- ~40 keys in 4096 bits -- spent most of its instructions on empty rows.)
  // Slices of different groups land in arrival order; everything downstream that needs the canonical order
  // compares the keys themselves, never list positions.
  const int n_li = ntx * nr;
  const int n_words = n_li * LAYER_WORDS;
  const int per = (n_words + NT - 1) / NT;
  const int w0 = min((int)threadIdx.x * per, n_words), w1 = min(w0 + per, n_words);
  int mine = 0;
  for (int wd = w0; wd < w1; ++wd) mine += __popc(lbits[wd]);
  int n_unique;
  const int ex = mg_block_exscan(mine, &n_unique);
  {
    int run = ex;  // exclusive prefix at the thread's first word
    for (int wd = w0; wd < w1; ++wd) {
      if ((wd & (LAYER_WORDS - 1)) == 0) s_cnt[wd / LAYER_WORDS] = run;  // first word of a layer
      run += __popc(lbits[wd]);
    }
  }
  if (threadIdx.x == 0) {
    s_cnt[n_li] = n_unique;
    s_base = n_unique ? atomicAdd(&d_num_circles[plane], n_unique) : 0;
  }
  __syncthreads();
  const int64_t base = s_base;
  if ((int)threadIdx.x < ntx) {
    int32_t* tr2 = d_tile_ranges + ((int64_t)plane * n_tiles + tile0 + threadIdx.x) * 2;
    tr2[0] = (int32_t)min(base + s_cnt[threadIdx.x * nr], circle_cap);
    tr2[1] = s_cnt[(threadIdx.x + 1) * nr] - s_cnt[threadIdx.x * nr];
  }
  if (d_layer_starts) {  // first key of every radius of every tile (+ the tile's end): the scoring kernel's chunks
    for (int i = threadIdx.x; i < ntx * (nr + 1); i += NT) {
      const int t = i / (nr + 1), ri = i - t * (nr + 1);
      d_layer_starts[((int64_t)plane * n_tiles + tile0 + t) * (nr + 1) + ri] =
          (int32_t)min(base + s_cnt[t * nr + ri], circle_cap);
    }
  }
  uint32_t* out = d_ukeys + (int64_t)plane * circle_cap;
  int64_t pos = base + ex;
  for (int wd = w0; wd < w1; ++wd) {
    uint32_t bits = lbits[wd];
    if (!bits) continue;
    int t = 0, rem = wd;  // tile of the group; rem * 32 + bit = the key's low 17 bits  (TX - 1 compares, not a division)
#pragma unroll
    for (int k = 1; k < TX; ++k)
      if (rem >= words) rem -= words, ++t;
    const uint32_t hi = ((uint32_t)(tile0 + t) << 17) | ((uint32_t)rem << 5);
    while (bits) {
      const int b = __ffs(bits) - 1;
      bits &= bits - 1;
      if (pos < circle_cap) out[pos] = hi | (uint32_t)b;
      ++pos;
    }
  }
}

__global__ __launch_bounds__(1024) void k_layer_scan(int32_t* __restrict__ d_layer_offsets, int n_layers,
                                                     int32_t* __restrict__ d_num_circles, int64_t cap) {
  // one workgroup per plane; every thread owns a contiguous run of counts: sum it, one block-wide
  // scan of the 1024 run sums, then write the run's exclusive prefixes (2 barriers instead of 2 per
  // 1024 elements -- with ~88 k layers per plane the chunked version spent 0.3 ms per step here)
  const int plane = blockIdx.x;
  int32_t* cnt = d_layer_offsets + (int64_t)plane * (n_layers + 1);
  const int per = (n_layers + 1023) / 1024;
  const int lo = min((int)threadIdx.x * per, n_layers), hi = min(lo + per, n_layers);
  int sum = 0;
  for (int i = lo; i < hi; ++i) sum += cnt[i];
  int total;
  int run = mg_block_exscan(sum, &total);
  for (int i = lo; i < hi; ++i) {
    const int v = cnt[i];
    cnt[i] = run;
    run += v;
  }
  if (threadIdx.x == 0) {
    cnt[n_layers] = total;
    d_num_circles[plane] = (int32_t)min((int64_t)total, cap);
  }
}

__global__ __launch_bounds__(NT) void k_layer_emit(uint32_t* __restrict__ d_bitmap, int64_t bitmap_words, int n_layers,
                                                   const int32_t* __restrict__ d_layer_offsets, int ntc, int nr,
                                                   int min_r, int max_r, int32_t* __restrict__ d_circles,
                                                   int64_t circle_cap) {
  const int plane = blockIdx.y;
  const int layer = blockIdx.x * (NT / 64) + (threadIdx.x >> 6);
  if (layer >= n_layers) return;
  const int lane = threadIdx.x & 63;
  const int32_t* lo = d_layer_offsets + (int64_t)plane * (n_layers + 1);
  if (lo[layer + 1] == lo[layer]) return;  // empty layer (wave-uniform)
  uint2* words = reinterpret_cast<uint2*>(d_bitmap + (int64_t)plane * bitmap_words + (int64_t)layer * LAYER_WORDS);
  const uint2 v = words[lane];
  uint64_t bits = ((uint64_t)v.y << 32) | v.x;
  if (bits) words[lane] = make_uint2(0u, 0u);  // leave the bitmap clean for the next use
  const int c = __popcll(bits);
  int incl = c;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int t = __shfl_up(incl, off);
    if (lane >= off) incl += t;
  }
  int64_t pos = (int64_t)lo[layer] + incl - c;
  const int tile = layer / nr, ri = layer - tile * nr;
  const int row = (tile / ntc) * TS + lane - max_r, col0 = (tile % ntc) * TS - max_r;
  int32_t* out = d_circles + (int64_t)plane * circle_cap * 3;
  while (bits) {
    const int b = __ffsll((unsigned long long)bits) - 1;
    bits &= bits - 1;
    if (pos < circle_cap) {
      out[3 * pos] = row;
      out[3 * pos + 1] = col0 + b;
      out[3 * pos + 2] = min_r + ri;
    }
    ++pos;
  }
}

// ---- K9: scoring, one workgroup per centre tile ------------------------------------------------------
// All circles of a tile (centres in a 64 x 64 block, radius <= max_r) touch only the
// (64 + 2 max_r)^2 window around it, so the window of the 1-bit edge map is staged in LDS once
// and every perimeter test is an LDS read (the global-gather version is bound by one cache-line
// transaction per test).
//   pass A (exact prefilter): every term of the alignment sum is <= 1, so a circle with fewer
//     than min_roundness * P edge pixels on its perimeter cannot reach the threshold;
//   pass B: the reference's float64 sum in perimeter order for the survivors (block-local list),
//     gradient angles gathered from the precomputed edge-angle map.
__device__ __forceinline__ uint32_t bits_at(const uint32_t* __restrict__ bits, int64_t bit0, int n) {
  const int64_t wi = bit0 >> 5;
  const int sh = (int)(bit0 & 31);
  uint64_t two = bits[wi];
  if (sh + n > 32) two |= (uint64_t)bits[wi + 1] << 32;
  const uint32_t v = (uint32_t)(two >> sh);
  return n >= 32 ? v : (v & ((1u << n) - 1u));
}

constexpr int CHUNK = 1024;        // capacity of the LDS survivor list of one prefilter/exact round
constexpr int SURV_IDX_BITS = 20;  // list entry: circle index within the round | hits << 20 (perimeters < 2048 points)

//   Orientation windows: a perimeter point whose radial direction theta lies in the quarter
//     [k pi/4, (k+1) pi/4] (mod pi) gets a term <= 0 from every edge pixel whose gradient
//     orientation class is (k + 2) & 3 (at least pi/4 away), so the prefilter counts it in window
//     W_k = edges & (class != (k + 2) & 3).  The 8 symmetric points of a group always fall into
//     the same quarters (3, 2, 0, 1, 0, 1, 3, 2 for the table's (x > 0, y < 0, x < -y) entries), so
//     with a compile-time slot size WS the window is an immediate offset of the LDS read.
//   KEYED: the tile's circles are the 32-bit keys k_tile_dedup emitted (d_ukeys, d_tile_ranges); the
//     (row, col, r) triple of a circle is written to d_circles only when it passes the threshold.
//     Otherwise: the (row, col, r) list and layer table of mg_bitmap_to_circles.
// (96 SGPRs: the allocation step above would cost one of the 8 waves per SIMD.)
template <int WS, bool KEYED>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_num_sgpr(96))) void k_score_tiles(const float* __restrict__ d_angle,
                                                    const uint32_t* __restrict__ d_bits,
                                                    const uint32_t* __restrict__ d_class, int64_t words_per_plane,
                                                    int h, int w, int32_t* __restrict__ d_circles,
                                                    int64_t circle_cap, const int32_t* __restrict__ d_layer_offsets,
                                                    const uint32_t* __restrict__ d_ukeys,
                                                    const int32_t* __restrict__ d_tile_ranges, int n_tiles,
                                                    int n_layers, int nr, int ntc, int min_r, int max_r,
                                                    const int32_t* __restrict__ d_per_rc, int per_total,
                                                    const double* __restrict__ d_per_expected,
                                                    const int32_t* __restrict__ d_per_starts, float min_roundness,
                                                    int write_skipped, int dedup_centres,
                                                    float* __restrict__ d_scores,
                                                    int32_t* __restrict__ d_alive, int32_t* __restrict__ d_num_alive,
                                                    int32_t* __restrict__ d_max_rc, int32_t* __restrict__ d_num_scored) {
  extern __shared__ uint32_t lds[];
  __shared__ int n_surv, n_pass;
  const int plane = blockIdx.y, tile = blockIdx.x;
  int64_t first, last;
  if (KEYED) {
    const int32_t* tr2 = d_tile_ranges + ((int64_t)plane * n_tiles + tile) * 2;
    first = tr2[0];
    last = min(first + tr2[1], circle_cap);
  } else {
    const int32_t* lo = d_layer_offsets + (int64_t)plane * (n_layers + 1);
    first = lo[(int64_t)tile * nr];
    last = min((int64_t)lo[(int64_t)(tile + 1) * nr], circle_cap);
  }
  if (first >= last) return;
  const int side = TS + 2 * max_r;
  int wsh = 0;
  while ((32 << wsh) < side) ++wsh;  // words per window row, rounded up to a power of two
  const int wpr = 1 << wsh;
  const int slot = WS ? WS : side * wpr;
  uint32_t* win = lds;                                    // [4][slot]: W_0 .. W_3, each [side][wpr]
  int32_t* tab = reinterpret_cast<int32_t*>(lds + 4 * slot);  // packed (dr << 16) | (dc & 0xFFFF)
  int32_t* list = tab + per_total;                        // [CHUNK]
  const int wy0 = (tile / ntc) * TS - 2 * max_r, wx0 = (tile % ntc) * TS - 2 * max_r;
  const uint32_t* bits = d_bits + plane * words_per_plane;
  // planes c0, c1 of mg_canny_nms' three (the quarter of the orientation; c2, its half, is not used here)
  const uint32_t* cls0 = d_class ? d_class + (3 * plane) * words_per_plane : nullptr;
  const uint32_t* cls1 = d_class ? d_class + (3 * plane + 1) * words_per_plane : nullptr;
  for (int i = threadIdx.x; i < side * wpr; i += NT) {
    const int j = i >> wsh, k = i & (wpr - 1);
    const int y = wy0 + j, xs = wx0 + 32 * k;
    uint32_t v = 0, c0 = 0, c1 = 0;
    if (y >= 0 && y < h) {
      const int x_lo = max(xs, 0), x_hi = min(min(xs + 32, wx0 + side), w);
      if (x_lo < x_hi) {
        v = bits_at(bits, (int64_t)y * w + x_lo, x_hi - x_lo) << (x_lo - xs);
        if (d_class && v) {
          c0 = bits_at(cls0, (int64_t)y * w + x_lo, x_hi - x_lo) << (x_lo - xs);
          c1 = bits_at(cls1, (int64_t)y * w + x_lo, x_hi - x_lo) << (x_lo - xs);
        }
      }
    }
    if (d_class) {
      win[i] = v & ~(c1 & ~c0);             // W_0: without class 2
      win[slot + i] = v & ~(c1 & c0);       // W_1: without class 3
      win[2 * slot + i] = v & (c1 | c0);    // W_2: without class 0
      win[3 * slot + i] = v & (c1 | ~c0);   // W_3: without class 1
    } else {
      win[i] = win[slot + i] = win[2 * slot + i] = win[3 * slot + i] = v;
    }
  }
  for (int i = threadIdx.x; i < per_total; i += NT) {
    // (dr << 16) | (quarter << 14) | (dc & 0x3FFF); the quarter of theta = atan2(dr, dc) mod pi by the same
    // integer rule as the pixels' classes (on a boundary either side is valid, but the prefilter's
    // count and the exact pass must use the same window)
    const int dr = d_per_rc[2 * i], dc = d_per_rc[2 * i + 1];
    const int ay = abs(dr), ax = abs(dc);
    const int quarter = ((dr ^ dc) < 0) ? (ay > ax ? 2 : 3) : (ay < ax ? 0 : 1);
    tab[i] = (dr << 16) | (quarter << 14) | (dc & 0x3FFF);
  }
  __syncthreads();
#define MG_TAB_DR(v) ((v) >> 16)
#define MG_TAB_DC(v) (((int)((uint32_t)(v) << 18)) >> 18)
#define MG_TAB_Q(v) (((v) >> 14) & 3)
  int32_t* circles = d_circles + (int64_t)plane * circle_cap * 3;
  const uint32_t* ukeys = KEYED ? d_ukeys + (int64_t)plane * circle_cap : nullptr;
  const int trow0 = (tile / ntc) * TS - max_r, tcol0 = (tile % ntc) * TS - max_r;  // centre of tile position (0, 0)
#define MG_CIRCLE(i, row, col, rad)                                         \
  int row, col, rad;                                                        \
  if (KEYED) {                                                              \
    const uint32_t key_ = ukeys[i];                                         \
    row = trow0 + (int)((key_ >> 6) & 63u);                                 \
    col = tcol0 + (int)(key_ & 63u);                                        \
    rad = min_r + (int)((key_ >> 12) & 31u);                                \
  } else {                                                                  \
    row = circles[3 * (i)], col = circles[3 * (i) + 1], rad = circles[3 * (i) + 2]; \
  }
#define MG_PASS(i, row, col, rad)                                                        \
  {                                                                                      \
    const int k_ = atomicAdd(&d_num_alive[plane], 1);                                    \
    d_alive[(int64_t)plane * circle_cap + k_] = (int32_t)(i);                            \
    if (KEYED) circles[3 * (i)] = (row), circles[3 * (i) + 1] = (col), circles[3 * (i) + 2] = (rad); \
    atomicMax(&d_max_rc[2 * plane], (row));                                              \
    atomicMax(&d_max_rc[2 * plane + 1], (col));                                          \
  }
  const float* ang = d_angle + (int64_t)plane * h * w;
  float* scores = d_scores + (int64_t)plane * circle_cap;
  const double PI = 3.141592653589793, INV_PI = 1.0 / 3.141592653589793;
  // Rounds: prefilter batches of BATCH circles until the survivor list is nearly full (or the tile is
  // done), then the exact pass over the list -- with few survivors per batch the exact pass would
  // otherwise run on mostly idle waves.
  constexpr int BATCH = 2 * NT;
  // dedup_centres: circles that pass are parked in the top PASS_CAP x 3 words of `list` until the tile is
  // done; of those that share a centre only the first in suppression order goes on (see the end)
  constexpr int PASS_CAP = 64, SURV_ROOM = CHUNK - 3 * PASS_CAP;
  if (threadIdx.x == 0) n_pass = 0;
  for (int64_t chunk = first; chunk < last;) {
    if (threadIdx.x == 0) n_surv = 0;
    __syncthreads();
    int64_t pos = chunk;
    for (;;) {
    for (int64_t i = pos + threadIdx.x; i < min(pos + BATCH, last); i += NT) {
      MG_CIRCLE(i, row, col, rad)
      const int p0 = d_per_starts[rad - min_r], p1 = d_per_starts[rad - min_r + 1];
      const int len = p1 - p0;
      // need: hits >= min_roundness * len - 1e-3 (margin far above any rounding of the real sum)
      const int need = (int)ceil((double)min_roundness * len - 1e-3);
      const int by = row - wy0, bx = col - wx0;
#define MG_BIT(k, yy, xx) \
  ((win[(k) * slot + ((by + (yy)) << wsh) + ((bx + (xx)) >> 5)] >> ((bx + (xx)) & 31)) & 1u)
      // The midpoint circle is emitted as 4 axis points, groups of 8 symmetric points sharing one
      // (x, y), and possibly 4 diagonal points (utils.py:441-464): one table read per group.
      // Axis points (quarters by the table's rule): theta = pi, pi/2, 0, pi/2.
      int hits = MG_BIT(3, 0, -rad) + MG_BIT(2, -rad, 0) + MG_BIT(0, 0, rad) + MG_BIT(1, rad, 0);
      int p = p0 + 4;
      for (; p + 8 <= p1; p += 8) {
        const int v = tab[p];
        const int x = MG_TAB_DR(v), y = MG_TAB_DC(v);  // entry (dr, dc) = (x, y), x > 0 > y, x < -y
        hits += MG_BIT(3, x, y) + MG_BIT(2, y, x) + MG_BIT(0, -x, y) + MG_BIT(1, -y, x) + MG_BIT(0, x, -y) +
                MG_BIT(1, y, -x) + MG_BIT(3, -x, -y) + MG_BIT(2, -y, -x);
        // (no early exit: a wave runs as long as its slowest lane anyway, and the test cost more than it saved)
      }
      if (p + 4 == p1) {  // diagonal points: theta = pi/4 (quarter 1) where dr, dc have the same sign, else 3 pi/4 (3)
        const int v = tab[p];
        const int x = MG_TAB_DR(v), y = MG_TAB_DC(v);
        const uint32_t* wa = win + (((x ^ y) < 0) ? 3 : 1) * slot;  // (x, y), (-x, -y)
        const uint32_t* wb = win + (((x ^ y) < 0) ? 1 : 3) * slot;  // (-x, y), (x, -y)
#define MG_BITW(ww, yy, xx) ((ww[((by + (yy)) << wsh) + ((bx + (xx)) >> 5)] >> ((bx + (xx)) & 31)) & 1u)
        hits += MG_BITW(wa, x, y) + MG_BITW(wa, -x, -y) + MG_BITW(wb, -x, y) + MG_BITW(wb, x, -y);
#undef MG_BITW
      }
#undef MG_BIT
      if (hits >= need) {
        list[atomicAdd(&n_surv, 1)] = (int32_t)(i - chunk) | (hits << SURV_IDX_BITS);
      } else if (write_skipped) {
        scores[i] = MG_SCORE_SKIPPED;
      }
    }
      pos += BATCH;
      __syncthreads();
      const int so_far = n_surv;
      if (pos >= last || so_far > SURV_ROOM - BATCH || pos - chunk > (1 << SURV_IDX_BITS) - BATCH) break;
      __syncthreads();  // everyone has read the count before the next batch adds to it
    }
    const int ns = n_surv;
    if (threadIdx.x == 0 && d_num_scored) atomicAdd(&d_num_scored[plane], ns);
    for (int a = threadIdx.x; a < ns; a += NT) {
      const int64_t i = chunk + (list[a] & ((1 << SURV_IDX_BITS) - 1));
      MG_CIRCLE(i, row, col, rad)
      const int p0 = d_per_starts[rad - min_r], p1 = d_per_starts[rad - min_r + 1];
      const int by = row - wy0, bx = col - wx0;
      double acc = 0.0;
      // Early abort (exact): every remaining counted pixel adds at most 1, so once
      // acc + remaining < min_roundness * len - 1e-3 the circle cannot pass any more.  Most circles
      // that survive the count-only prefilter sit just above it and their terms average ~0 (noise
      // edges point anywhere): they are ruled out after a handful of terms.
      double left = (double)(list[a] >> SURV_IDX_BITS);  // prefilter hits on the perimeter not yet summed
      const double floor_sum = (double)min_roundness * (double)(p1 - p0) - 1e-3;
      bool dead = false;
      // 32 perimeter points at a time: hit mask from LDS, then only the hits (in perimeter order)
      // pay for the angle gather and the float64 arithmetic
      for (int base = p0; base < p1 && !dead; base += 32) {
        uint32_t mask = 0, counted = 0;  // edge pixels / those the prefilter counted (its window W_quarter)
        const int cnt = min(32, p1 - base);
        for (int j = 0; j < cnt; ++j) {
          const int v = tab[base + j];
          const int y = by + MG_TAB_DR(v), x = bx + MG_TAB_DC(v);
          const int wi = (y << wsh) + (x >> 5);  // every edge pixel is in W_0 or in W_2
          mask |= (((win[wi] | win[2 * slot + wi]) >> (x & 31)) & 1u) << j;
          counted |= ((win[MG_TAB_Q(v) * slot + wi] >> (x & 31)) & 1u) << j;
        }
        // The gathers of up to PF hits are issued together (their latency, not the arithmetic, is what
        // a lane waits for), then summed in perimeter order.
        constexpr int PF = 8;
        while (mask && !dead) {
          int jj[PF];
          float an[PF];
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            jj[u] = __ffs(mask) - 1;  // -1 once the block's hits are used up
            mask &= mask - 1;
            an[u] = 0.0f;
            if (jj[u] >= 0) {
              const int v = tab[base + jj[u]];
              // hits lie inside the image: 24-bit multiply (h, w < 2^24 guaranteed by the launcher)
              an[u] = ang[(int64_t)(__umul24(row + MG_TAB_DR(v), w) + (col + MG_TAB_DC(v)))];
            }
          }
#pragma unroll
          for (int u = 0; u < PF; ++u) {
            if (jj[u] < 0 || dead) continue;
            const int p = base + jj[u];
            double d = fabs((double)an[u] - d_per_expected[p]);
            if (d > PI) d = d - PI;
            // x / pi, correctly rounded without the division (Markstein: y = RN(1/pi), q0 = RN(x y),
            // r = x - q0 pi exactly by FMA, q = RN(q0 + r y) == RN(x / pi) because pi's significand is
            // not all ones; verified against x / pi on 1e9 operands of exactly this form)
            const double x4 = 4.0 * fabs(d - PI / 2.0);
            const double q0 = x4 * INV_PI;
            const double q = fma(fma(-q0, PI, x4), INV_PI, q0);
            acc += q - 1.0;
            // `left` = the prefilter's hits still to come; the other edge pixels (perpendicular class)
            // add <= 0 (+6e-8 at worst when exactly pi/4 off radial: inside the 1e-3 margin)
            left -= (double)((counted >> jj[u]) & 1u);
            if (acc + left < floor_sum) dead = true;
          }
        }
      }
      if (dead) {
        if (write_skipped) scores[i] = MG_SCORE_SKIPPED;
        continue;
      }
      const float score = (float)acc / (float)(p1 - p0);
      scores[i] = score;
      if (score >= min_roundness) {
        const int slot = dedup_centres ? atomicAdd(&n_pass, 1) : PASS_CAP;
        if (slot < PASS_CAP) {
          list[CHUNK - 1 - 3 * slot] = (int32_t)(i - first);
          list[CHUNK - 2 - 3 * slot] = __float_as_int(score);
          list[CHUNK - 3 - 3 * slot] = (rad << 12) | ((row - trow0) << 6) | (col - tcol0);
        } else {  // no room (or no de-duplication wanted): straight to the plane's list
          MG_PASS(i, row, col, rad)
        }
      }
    }
    __syncthreads();
    chunk = pos;
  }
  // Circles with one centre have one suppression ring (it only depends on the centre and min_dist), so
  // whatever the first of them in (score desc, radius asc) order does -- claim the ring, or fail on a cell
  // that is already claimed -- leaves every later one rejected: they can be dropped here, and on clean
  // images (many radii pass per bead) most of the suppression's bids go with them.
  const int np = min(n_pass, PASS_CAP);
  for (int a = threadIdx.x; a < np; a += NT) {
    const int code = list[CHUNK - 3 - 3 * a];
    const float score = __int_as_float(list[CHUNK - 2 - 3 * a]);
    bool first_at_centre = true;
    for (int b = 0; b < np; ++b) {
      const int code_b = list[CHUNK - 3 - 3 * b];
      const float score_b = __int_as_float(list[CHUNK - 2 - 3 * b]);
      if (((code ^ code_b) & 0xFFF) == 0 && (score_b > score || (score_b == score && code_b < code))) first_at_centre = false;
    }
    if (first_at_centre) {
      const int64_t i = first + list[CHUNK - 1 - 3 * a];
      MG_PASS(i, trow0 + ((code >> 6) & 63), tcol0 + (code & 63), code >> 12)
    }
  }
}

#undef MG_TAB_DR
#undef MG_TAB_DC
#undef MG_TAB_Q
#undef MG_CIRCLE
#undef MG_PASS

// ---- K10: greedy suppression in parallel rounds ------------------------------------------------------
// Priority key: smaller = earlier in the reference's score-descending order; ties broken by the
// canonical (tile, r, row, col) order -- the circle's 32-bit de-duplication key on the keyed path
// (list positions are arrival-ordered there), its list index otherwise.  The score field of a real
// key is never 0, which leaves (0, tie) as the mark a kept circle puts on its ring cells.
__device__ __forceinline__ uint64_t nms_key(float score, uint32_t idx) {
  uint32_t b = __float_as_uint(score);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);  // ascending-sortable
  return ((uint64_t)(~b) << 32) | idx;             // descending score
}

// Python's v % n for n > 0.  Ring coordinates are at most one period outside [0, n) in practice, so
// the integer division (~30 instructions, twice per ring pixel) is kept off the common path.
__device__ __forceinline__ int wrap(int v, int n) {
  if (v >= n) {
    v -= n;
    if (v >= n) v %= n;
  } else if (v < 0) {
    v += n;
    if (v < 0) {
      v %= n;
      if (v < 0) v += n;
    }
  }
  return v;
}

// PHASE 0 (bid), 1 (decide), 2 (cleanup): the lanes of a wave work on the ring cells of one circle together -- its
// loads and atomics are in flight at once and the verdict is two ballots (a thread per circle walked its ring in
// seven dependent steps: a chain of HBM latencies per round at any batch size).  PHASE 3-5: one thread per circle.
// d_undecided[plane] becomes non-zero when a circle stays undecided (a flag, not a count: one atomicAdd per wave on
// one address serialised 18 000 deep in the first round).
template <int PHASE>
__global__ __launch_bounds__(NT) void k_nms(const int32_t* __restrict__ d_circles, int64_t circle_cap,
                                            const float* __restrict__ d_scores, const int32_t* __restrict__ d_alive,
                                            const int32_t* __restrict__ d_num_alive,
                                            const int32_t* __restrict__ d_max_rc, int min_dist,
                                            const int32_t* __restrict__ d_ring_rc, int ring_len,
                                            uint64_t* __restrict__ d_grid, int64_t grid_cap,
                                            uint8_t* __restrict__ d_state, int32_t* __restrict__ d_undecided,
                                            const uint32_t* __restrict__ d_tie, int cpw,
                                            const int32_t* __restrict__ d_skip) {
  const int plane = blockIdx.y;
  const int n = d_num_alive[plane];
  if (n == 0) return;
  if (d_skip && d_skip[plane]) {
    // the plane was decided by mg_nms_sparse: no bids, nothing on the claim grid.  The cleanup only returns the
    // state bytes of the alive circles to 0 (what it does for every circle whose ring it restores)
    if (PHASE == 2)
      for (int64_t a = (int64_t)blockIdx.x * NT + threadIdx.x; a < n; a += (int64_t)gridDim.x * NT)
        d_state[(int64_t)plane * circle_cap + d_alive[(int64_t)plane * circle_cap + a]] = 0;
    return;
  }
  const uint32_t* tie = d_tie ? d_tie + (int64_t)plane * circle_cap : nullptr;
  const int pad = 2 * min_dist + 1;
  const int n_rows = d_max_rc[2 * plane] + 2 * pad, n_cols = d_max_rc[2 * plane + 1] + 2 * pad;
  if ((int64_t)n_rows * n_cols > grid_cap) return;  // caller sized the grid from the image extent
  const int32_t* circles = d_circles + (int64_t)plane * circle_cap * 3;
  const float* scores = d_scores + (int64_t)plane * circle_cap;
  const int32_t* alive = d_alive + (int64_t)plane * circle_cap;
  uint8_t* state = d_state + (int64_t)plane * circle_cap;
  uint64_t* grid = d_grid + (int64_t)plane * grid_cap;
  if (PHASE >= 3) {
    for (int64_t a = (int64_t)blockIdx.x * NT + threadIdx.x; a < n; a += (int64_t)gridDim.x * NT) {
      const int idx = alive[a];
      if (PHASE != 5 && state[idx] != 0) continue;
      const int row = circles[3 * (int64_t)idx], col = circles[3 * (int64_t)idx + 1];
      const uint64_t key = nms_key(scores[idx], tie ? tie[idx] : (uint32_t)idx);
      // Same-centre reduction before the rounds: circles with one centre have one ring, so whatever the first of
      // them in suppression order does -- claim the ring, or fail on a cell that is already claimed -- leaves every
      // later one rejected without ever claiming (utils.py:254-292): only that first one enters the rounds.  One
      // bid on the centre's own cell (3), the losers marked rejected (4), the cell restored (5): three tiny
      // launches instead of a ring of bids, loads and withdrawals per duplicate.
      uint64_t* cell = &grid[(int64_t)wrap(row + pad, n_rows) * n_cols + wrap(col + pad, n_cols)];
      if (PHASE == 3) atomicMin(reinterpret_cast<unsigned long long*>(cell), (unsigned long long)key);
      else if (PHASE == 4) { if (*cell != key) state[idx] = 2; }
      else *cell = ~0ull;
    }
    return;
  }
  // A wave takes cpw consecutive alive circles: lanes 0 .. cpw - 1 look their own circle up (coalesced; most are
  // decided after the first round and the wave is done), then the wave works through the undecided ones NU at a time,
  // all lanes on the ring cells of one circle -- the loads of the NU rings are in flight together, the verdict of a
  // circle is two ballots.  cpw = 64 at large batches (a wave per circle cost ~75 us per launch for 367 000 waves that
  // only looked at a state byte), fewer when the whole batch has few circles (the trips of a wave are a chain).
  constexpr int NU = 4;
  const int lane = threadIdx.x & 63;
  const int64_t n_waves = ((int64_t)gridDim.x * NT) >> 6;
  bool any_undecided = false;
  for (int64_t w0 = ((int64_t)blockIdx.x * NT + threadIdx.x) >> 6; w0 * cpw < n; w0 += n_waves) {
    const int64_t a = w0 * cpw + lane;
    const bool have = lane < cpw && a < n;
    const int my_idx = have ? alive[a] : 0;
    const bool want = have && (PHASE == 2 || state[my_idx] == 0);
    int my_row = 0, my_col = 0;
    uint32_t my_tk = 0, key_lo = 0, key_hi = 0;
    if (want) {
      my_row = circles[3 * (int64_t)my_idx], my_col = circles[3 * (int64_t)my_idx + 1];
      my_tk = tie ? tie[my_idx] : (uint32_t)my_idx;
      const uint64_t k = nms_key(scores[my_idx], my_tk);
      key_lo = (uint32_t)k, key_hi = (uint32_t)(k >> 32);
    }
    uint64_t todo = __ballot(want);
    while (todo) {
      int idx[NU], row[NU], col[NU], cnt = 0;
      uint32_t tk[NU];
      uint64_t key[NU];
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        idx[u] = 0, row[u] = 0, col[u] = 0, tk[u] = 0, key[u] = 0;
        if (todo) {  // wave-uniform
          const int src = __ffsll((unsigned long long)todo) - 1;
          todo &= todo - 1;
          idx[u] = __builtin_amdgcn_readlane(my_idx, src);
          row[u] = __builtin_amdgcn_readlane(my_row, src);
          col[u] = __builtin_amdgcn_readlane(my_col, src);
          tk[u] = (uint32_t)__builtin_amdgcn_readlane((int)my_tk, src);
          key[u] = ((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)key_hi, src) << 32) |
                   (uint32_t)__builtin_amdgcn_readlane((int)key_lo, src);
          cnt = u + 1;
        }
      }
      for (int j0 = 0; j0 < ring_len; j0 += 64) {  // (one trip for rings of up to 64 cells)
        const int j = j0 + lane;
        const bool on = j < ring_len;
        const int dr = on ? d_ring_rc[2 * j] : 0, dc = on ? d_ring_rc[2 * j + 1] : 0;
        uint64_t* cell[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u)
          cell[u] = &grid[(int64_t)wrap(dr + row[u] + pad, n_rows) * n_cols + wrap(dc + col[u] + pad, n_cols)];
        if (PHASE == 2) {  // cleanup (after convergence): restore the all-ones grid under every ring
#pragma unroll
          for (int u = 0; u < NU; ++u)
            if (u < cnt && on) *cell[u] = ~0ull;
        } else if (PHASE == 0) {  // claim: every undecided circle bids for its ring pixels
          // a cell that already holds a smaller key cannot be won (cells only ever decrease during the bids):
          // look first and spare the memory-side atomic for those
          uint64_t cur[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u) cur[u] = (u < cnt && on) ? __builtin_nontemporal_load(cell[u]) : 0ull;
#pragma unroll
          for (int u = 0; u < NU; ++u)
            if (u < cnt && on && cur[u] > key[u]) atomicMin(reinterpret_cast<unsigned long long*>(cell[u]), (unsigned long long)key[u]);
        }
      }
      if (PHASE == 2) {
#pragma unroll
        for (int u = 0; u < NU; ++u)
          if (u < cnt && lane == 0) state[idx[u]] = 0;
      } else if (PHASE == 1) {  // decide
        bool foreign[NU], kept[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) foreign[u] = false, kept[u] = false;
        for (int j = lane; j < ring_len; j += 64) {
          const int dr = d_ring_rc[2 * j], dc = d_ring_rc[2 * j + 1];
          uint64_t g[NU];
#pragma unroll
          for (int u = 0; u < NU; ++u)
            g[u] = u < cnt ? grid[(int64_t)wrap(dr + row[u] + pad, n_rows) * n_cols + wrap(dc + col[u] + pad, n_cols)] : key[u];
#pragma unroll
          for (int u = 0; u < NU; ++u)
            if (g[u] != key[u]) {
              foreign[u] = true;
              if ((g[u] >> 32) == 0) kept[u] = true;  // a kept circle's mark
            }
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
          if (u >= cnt) break;  // wave-uniform
          const bool all_mine = __ballot(foreign[u]) == 0, hit_kept = __ballot(kept[u]) != 0;
          if (all_mine) {
            if (lane == 0) state[idx[u]] = 1;
            // mark the ring as kept: (0, tie) is below every real key, so no later bid replaces it
            for (int j = lane; j < ring_len; j += 64)
              atomicMin(reinterpret_cast<unsigned long long*>(
                            &grid[(int64_t)wrap(d_ring_rc[2 * j] + row[u] + pad, n_rows) * n_cols + wrap(d_ring_rc[2 * j + 1] + col[u] + pad, n_cols)]),
                        (unsigned long long)tk[u]);
          } else if (hit_kept) {
            if (lane == 0) state[idx[u]] = 2;
            // withdraw this circle's bids so that later circles can win these pixels
            for (int j = lane; j < ring_len; j += 64)
              atomicCAS(reinterpret_cast<unsigned long long*>(
                            &grid[(int64_t)wrap(d_ring_rc[2 * j] + row[u] + pad, n_rows) * n_cols + wrap(d_ring_rc[2 * j + 1] + col[u] + pad, n_cols)]),
                        (unsigned long long)key[u], ~0ull);
          } else {
            any_undecided = true;
          }
        }
      }
    }
  }
  if (PHASE == 1 && any_undecided && lane == 0 && d_undecided[plane] == 0) d_undecided[plane] = 1;
}

// ---- K10s: the same greedy suppression, decided per plane in ONE workgroup from the circles alone ----------------
// utils.py:254-292 keeps circle i iff no ring cell of i is claimed by a circle kept before it: iff no kept circle j
// earlier in the order has ring(c_j) and ring(c_i) in common: iff c_i - c_j lies in D = ring (-) ring for no kept
// earlier j.  So the grid is not needed -- only, for every alive circle, the alive circles around it.  The rounds above
// reach that through 64 claim-grid cells per circle and round (memory-side atomics and scattered 8-byte loads over a
// 137 MB grid per plane: 0.9 ms per step at C4 with its same-centre pass and the cleanup); here a plane's ~11 000
// alive circles are sorted into buckets of 32 x 64 in LDS, a circle looks at the 3 x 3 buckets around it, tests
// c_i - c_j against the bitmap of D and compares keys: kept when every conflicting circle before it is rejected,
// rejected when one of them is kept, undecided (another round of the in-kernel loop) otherwise -- the sequential
// greedy choice, whatever the order the threads run in.  A plane the kernel cannot take (more than SP_CAP circles, more
// buckets than SP_MAXB, a centre so far outside the image that the reference's negative index would WRAP, utils.py:271:
// row or col < -(min_dist + 1)) is left to the rounds: d_done[plane] = 0.
constexpr int SP_NT = 1024;
constexpr int SP_CAP = 12288;           // alive circles of a plane held in LDS (SP_PER per thread)
constexpr int SP_PER = SP_CAP / SP_NT;
constexpr int SP_SHIFT_R = 5, SP_SHIFT_C = 6;  // log2 of a bucket's rows / columns: both >= the reach 2 * min_dist
constexpr int SP_BIAS = 64;             // added to coordinates: >= min_dist + 1, so stored coordinates are >= 0
constexpr int SP_MAXB = 9216;           // buckets per plane (16-bit starts: two per LDS word)
constexpr int SP_MAXD = 15;             // min_dist up to this (the bitmap of D has (4 d + 1)^2 bits)
constexpr int SP_DWORDS = ((4 * SP_MAXD + 1) * (4 * SP_MAXD + 1) + 31) / 32;
constexpr int SP_BWORDS = (SP_MAXB + 2 + 1) / 2;
constexpr size_t SP_LDS = (size_t)SP_CAP * (4 + 4 + 2 + 1) + (size_t)SP_BWORDS * 4 + SP_DWORDS * 4;
static_assert(2 * SP_MAXD <= (1 << SP_SHIFT_R) && SP_CAP < 65536, "bucket edge / 16-bit starts");

__global__ __launch_bounds__(SP_NT) void k_nms_sparse(const int32_t* __restrict__ d_circles, int64_t circle_cap,
                                                      const float* __restrict__ d_scores,
                                                      const int32_t* __restrict__ d_alive,
                                                      const int32_t* __restrict__ d_num_alive,
                                                      const int32_t* __restrict__ d_max_rc, int min_dist,
                                                      const uint32_t* __restrict__ d_dbits, uint8_t* __restrict__ d_state,
                                                      const uint32_t* __restrict__ d_tie, int32_t* __restrict__ d_done) {
  extern __shared__ __attribute__((aligned(16))) uint8_t sp_lds[];
  uint32_t* pos = reinterpret_cast<uint32_t*>(sp_lds);          // (row + bias) << 16 | (col + bias)
  uint32_t* skey = pos + SP_CAP;                                // the score half of nms_key: smaller = earlier
  uint32_t* bwords = skey + SP_CAP;                             // [nb + 2] 16-bit entries, two per word
  uint16_t* bstart = reinterpret_cast<uint16_t*>(bwords);
  uint32_t* dbits = bwords + SP_BWORDS;
  uint16_t* sorted = reinterpret_cast<uint16_t*>(dbits + SP_DWORDS);
  uint8_t* st = reinterpret_cast<uint8_t*>(sorted + SP_CAP);
  __shared__ int s_bad;
  const int plane = blockIdx.x;
  const int n = d_num_alive[plane];
  if (threadIdx.x == 0) s_bad = 0;
  const int nbr = ((d_max_rc[2 * plane] + SP_BIAS) >> SP_SHIFT_R) + 1, nbc = ((d_max_rc[2 * plane + 1] + SP_BIAS) >> SP_SHIFT_C) + 1;
  const int nb = nbr * nbc;
  if (n <= 0 || n > SP_CAP || nb > SP_MAXB || nbr <= 0 || nbc <= 0) {  // block-uniform
    if (threadIdx.x == 0) d_done[plane] = (n <= 0);  // (no alive circle: nothing for the rounds either)
    return;
  }
  const int32_t* circles = d_circles + (int64_t)plane * circle_cap * 3;
  const float* scores = d_scores + (int64_t)plane * circle_cap;
  const int32_t* alive = d_alive + (int64_t)plane * circle_cap;
  const uint32_t* tie = d_tie ? d_tie + (int64_t)plane * circle_cap : nullptr;
  const int side = 4 * min_dist + 1, reach = 2 * min_dist;
  for (int i = threadIdx.x; i < (side * side + 31) / 32; i += SP_NT) dbits[i] = d_dbits[i];
  for (int i = threadIdx.x; i < (nb + 3) / 2; i += SP_NT) bwords[i] = 0u;
  __syncthreads();
  // ---- the circles, their buckets, their slots inside the buckets (the gathers of all of a thread's circles in flight
  // together: index, then row / column / score, then the LDS work) ----
  int slot[SP_PER], bkt[SP_PER], idx[SP_PER], row[SP_PER], col[SP_PER];
  float sc[SP_PER];
  bool bad = false;
#pragma unroll
  for (int u = 0; u < SP_PER; ++u) idx[u] = alive[min((int)threadIdx.x + u * SP_NT, n - 1)];
#pragma unroll
  for (int u = 0; u < SP_PER; ++u) {
    row[u] = circles[3 * (int64_t)idx[u]], col[u] = circles[3 * (int64_t)idx[u] + 1];
    sc[u] = scores[idx[u]];
  }
  const int max_row = d_max_rc[2 * plane], max_col = d_max_rc[2 * plane + 1];
#pragma unroll
  for (int u = 0; u < SP_PER; ++u) {
    const int a = threadIdx.x + u * SP_NT;
    slot[u] = 0, bkt[u] = -1;
    if (a < n) {
      if (row[u] < -(min_dist + 1) || col[u] < -(min_dist + 1) || row[u] > max_row || col[u] > max_col) {
        bad = true;
      } else {
        uint32_t b = __float_as_uint(sc[u]);
        b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
        pos[a] = ((uint32_t)(row[u] + SP_BIAS) << 16) | (uint32_t)(col[u] + SP_BIAS);
        skey[a] = ~b;
        st[a] = 0;
        bkt[u] = ((row[u] + SP_BIAS) >> SP_SHIFT_R) * nbc + ((col[u] + SP_BIAS) >> SP_SHIFT_C);
        // entry bkt + 1 counts the bucket (16-bit halves of a word: a 32-bit atomic on the half's word; n < 65536: no carry)
        const int e = bkt[u] + 1, sh = 16 * (e & 1);
        slot[u] = (int)((atomicAdd(&bwords[e >> 1], 1u << sh) >> sh) & 0xFFFFu);
      }
    }
  }
  if (bad) s_bad = 1;
  __syncthreads();
  if (s_bad) {  // block-uniform
    if (threadIdx.x == 0) d_done[plane] = 0;
    return;
  }
  // ---- bucket starts: every thread sums a run of buckets, one block-wide scan, the run's prefixes ----
  {
    const int per = (nb + SP_NT - 1) / SP_NT;
    const int lo = min((int)threadIdx.x * per, nb), hi = min(lo + per, nb);
    constexpr int MAXPER = (SP_MAXB + SP_NT - 1) / SP_NT;
    int cnt[MAXPER];
    int sum = 0;
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
      cnt[k] = lo + k < hi ? (int)bstart[lo + k + 1] : 0;
      sum += cnt[k];
    }
    int total;
    int run = mg_block_exscan(sum, &total);
    __syncthreads();  // every count has been read: the entries become starts (16-bit stores of neighbours share words)
#pragma unroll
    for (int k = 0; k < MAXPER; ++k) {
      if (lo + k < hi) bstart[lo + k] = (uint16_t)run;
      run += cnt[k];
    }
    if (threadIdx.x == 0) bstart[nb] = (uint16_t)n;
  }
  __syncthreads();
#pragma unroll
  for (int u = 0; u < SP_PER; ++u)
    if (bkt[u] >= 0) sorted[bstart[bkt[u]] + slot[u]] = (uint16_t)(threadIdx.x + u * SP_NT);
  __syncthreads();
  // ---- rounds ----
  // A circle that is blocked remembers ONE circle it waits for (the earliest undecided conflicting circle before it):
  // while that one is undecided the circle stays blocked, when it is kept the circle is rejected -- no new scan of
  // the neighbourhood in either case; only when it is rejected does the circle look around again.
  auto scan = [&](int a, int& wait_for) -> int {  // 1 kept, 2 rejected, 0 blocked (wait_for = whom to watch)
    const uint32_t pa = pos[a], ka = skey[a];
    const int ra = (int)(pa >> 16), ca = (int)(pa & 0xFFFFu);
    const int br = ra >> SP_SHIFT_R, bc = ca >> SP_SHIFT_C;
    uint32_t ta = 0, best_k = 0xFFFFFFFFu, best_t = 0xFFFFFFFFu;
    bool have_ta = false;
    wait_for = -1;
    for (int dbr = -1; dbr <= 1; ++dbr) {
      const int r2 = br + dbr;
      if (r2 < 0 || r2 >= nbr) continue;
      const int c_lo = max(bc - 1, 0), c_hi = min(bc + 1, nbc - 1);
      const int k0 = bstart[r2 * nbc + c_lo], k1 = bstart[r2 * nbc + c_hi + 1];  // three buckets of a row are one run
      for (int k = k0; k < k1; ++k) {
        const int j = sorted[k];
        if (j == a) continue;
        const uint32_t pj = pos[j];
        const int dr = ra - (int)(pj >> 16), dc = ca - (int)(pj & 0xFFFFu);
        if (abs(dr) > reach || abs(dc) > reach) continue;
        const int bit = (dr + reach) * side + dc + reach;
        if (!((dbits[bit >> 5] >> (bit & 31)) & 1u)) continue;
        // j's ring meets a's: is j before a?
        const uint32_t kj = skey[j];
        bool before = kj < ka;
        uint32_t tj = 0;
        if (kj == ka) {
          if (!have_ta) ta = tie ? tie[alive[a]] : (uint32_t)alive[a], have_ta = true;
          tj = tie ? tie[alive[j]] : (uint32_t)alive[j];
          before = tj < ta;
        }
        if (!before) continue;
        const uint8_t sj = st[j];
        if (sj == 1) return 2;
        if (sj == 0 && (kj < best_k || (kj == best_k && tj < best_t))) best_k = kj, best_t = tj, wait_for = j;
      }
    }
    return wait_for < 0 ? 1 : 0;
  };
  int wait_for[SP_PER];
#pragma unroll
  for (int u = 0; u < SP_PER; ++u) wait_for[u] = -1;
  int again;
  do {
    int changed = 0;
#pragma unroll
    for (int u = 0; u < SP_PER; ++u) {
      const int a = threadIdx.x + u * SP_NT;
      if (a >= n || st[a] != 0) continue;
      if (wait_for[u] >= 0) {
        const uint8_t sw = st[wait_for[u]];
        if (sw == 0) continue;  // still blocked
        if (sw == 1) {
          st[a] = 2, changed = 1;
          continue;
        }
      }
      const int res = scan(a, wait_for[u]);
      if (res) st[a] = (uint8_t)res, changed = 1;
    }
    again = __syncthreads_or(changed);
  } while (again);
  uint8_t* state = d_state + (int64_t)plane * circle_cap;
  for (int a = threadIdx.x; a < n; a += SP_NT) state[alive[a]] = st[a];
  if (threadIdx.x == 0) d_done[plane] = 1;
}

// ---- K11: collect kept circles in priority order -------------------------------------------------------
__global__ __launch_bounds__(NT) void k_collect_list(const int32_t* __restrict__ d_alive,
                                                     const int32_t* __restrict__ d_num_alive,
                                                     const uint8_t* __restrict__ d_state, int keep_all,
                                                     int64_t circle_cap, int64_t out_cap, const float* __restrict__ d_scores,
                                                     const uint32_t* __restrict__ d_tie,
                                                     int32_t* __restrict__ d_scratch, int32_t* __restrict__ d_num_out) {
  const int plane = blockIdx.y;
  const int n = d_num_alive[plane];
  int32_t* list = d_scratch + (int64_t)plane * 3 * out_cap;  // [out_cap] indices, then [out_cap][2] keys (lo, hi)
  for (int64_t a = (int64_t)blockIdx.x * NT + threadIdx.x; a < n; a += (int64_t)gridDim.x * NT) {
    const int idx = d_alive[(int64_t)plane * circle_cap + a];
    if (keep_all || d_state[(int64_t)plane * circle_cap + idx] == 1) {
      const int k = atomicAdd(&d_num_out[plane], 1);
      if (k < out_cap) {
        const uint64_t key = nms_key(d_scores[(int64_t)plane * circle_cap + idx],
                                     d_tie ? d_tie[(int64_t)plane * circle_cap + idx] : (uint32_t)idx);
        list[k] = idx;
        list[out_cap + 2 * (int64_t)k] = (int32_t)(uint32_t)key;
        list[out_cap + 2 * (int64_t)k + 1] = (int32_t)(uint32_t)(key >> 32);
      }
    }
  }
}

// Rank of every kept circle in the (score desc, index asc) order = number of kept circles with a
// smaller key; the keys (written next to the list by k_collect_list) are staged through LDS in chunks so that
// the m comparisons per circle are LDS broadcasts.
constexpr int RANK_CHUNK = 2048;

__global__ __launch_bounds__(NT) void k_collect_rank(const int32_t* __restrict__ d_circles, int64_t circle_cap,
                                                     const float* __restrict__ d_scores,
                                                     const int32_t* __restrict__ d_scratch,
                                                     int32_t* __restrict__ d_num_out, int64_t out_cap,
                                                     int32_t* __restrict__ d_out, float* __restrict__ d_out_scores) {
  __shared__ __attribute__((aligned(16))) uint64_t keys[RANK_CHUNK];
  const int plane = blockIdx.y;
  const int m = (int)min((int64_t)d_num_out[plane], out_cap);
  const int32_t* list = d_scratch + (int64_t)plane * 3 * out_cap;
  const uint32_t* kw = reinterpret_cast<const uint32_t*>(list + out_cap);
  const float* scores = d_scores + (int64_t)plane * circle_cap;
  // block-uniform trip count: every thread takes part in the staging barriers
  for (int64_t a0 = (int64_t)blockIdx.x * NT; a0 < m; a0 += (int64_t)gridDim.x * NT) {
    const int64_t a = a0 + threadIdx.x;
    const int idx = a < m ? list[a] : 0;
    const uint64_t key = a < m ? (((uint64_t)kw[2 * a + 1] << 32) | kw[2 * a]) : 0;
    int rank = 0;
    for (int c0 = 0; c0 < m; c0 += RANK_CHUNK) {
      const int cn = min(RANK_CHUNK, m - c0);
      __syncthreads();
      for (int b = threadIdx.x; b < cn; b += NT) keys[b] = ((uint64_t)kw[2 * (int64_t)(c0 + b) + 1] << 32) | kw[2 * (int64_t)(c0 + b)];
      __syncthreads();
      // two keys per 128-bit LDS broadcast read, eight keys per trip: the reads of a trip are in flight together
      // (one dependent 64-bit read per key made this loop a chain of LDS latencies: 111 us for 2000 circles)
      const uint4* k4 = reinterpret_cast<const uint4*>(keys);
      int b = 0;
      for (; b + 8 <= cn; b += 8) {
        uint4 q[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) q[u] = k4[(b >> 1) + u];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          rank += (((uint64_t)q[u].y << 32) | q[u].x) < key;
          rank += (((uint64_t)q[u].w << 32) | q[u].z) < key;
        }
      }
      for (; b < cn; ++b) rank += keys[b] < key;
    }
    if (a < m) {
      const int32_t* c = d_circles + ((int64_t)plane * circle_cap + idx) * 3;
      int32_t* o = d_out + ((int64_t)plane * out_cap + rank) * 3;
      o[0] = c[0];
      o[1] = c[1];
      o[2] = c[2];
      if (d_out_scores) d_out_scores[(int64_t)plane * out_cap + rank] = scores[idx];
    }
  }
}

__global__ void k_clamp_counts(int32_t* d_num_out, int n_planes, int64_t out_cap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_planes && d_num_out[i] > out_cap) d_num_out[i] = (int32_t)out_cap;
}

inline int grid_x(int64_t n) { return (int)std::max<int64_t>(1, std::min<int64_t>((n + NT - 1) / NT, 8192)); }

}  // namespace

namespace {
int launch_candidates(const int32_t* d_coords, int64_t coord_cap, const int32_t* d_cell_starts,
                      const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h, int w, int grid,
                      const uint64_t* d_seeds, int64_t num_iter, int min_r, int max_r, uint32_t* d_bitmap,
                      int64_t bitmap_words, float* d_raw, uint32_t* d_keys, void* stream) {
  if (!d_coords || !d_cell_starts || !d_cell_counts || !d_num_edges || !d_seeds || (!d_bitmap && !d_keys))
    return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || h <= 0 || w <= 0 || grid <= 0 || num_iter < 0 || min_r < 0 || max_r < min_r)
    return MG_EINVAL;
  if ((double)num_iter * (double)h * (double)w >= 4.0e15) return MG_EINVAL;  // exact-division range of the strata
  int ntr_, ntc_;
  int64_t n_layers_, need_words_;
  if (mg_dedup_layout(h, w, min_r, max_r, &ntr_, &ntc_, &n_layers_, &need_words_) != MG_OK) return MG_EINVAL;
  if (d_bitmap && bitmap_words < need_words_) return MG_EINVAL;
  if (d_keys && ((int64_t)ntr_ * ntc_ >= 32768 || max_r - min_r + 1 > 32)) return MG_EINVAL;  // key fields
  if (n_planes == 0 || num_iter == 0) return MG_OK;
  const int gr = (h + grid - 1) / grid, gc = (w + grid - 1) / grid;
  const bool fast = num_iter < (1ll << 31) && grid > 1 && h <= 65536 && w <= 65536;
  // keys from the table kernel (round 4) unless MG_CANDIDATES_V1 is set (the tests compare the two) or the shape is
  // outside its ranges: a grid cell above 32 (table size), a coordinate list of 2^31 entries
  static const bool v1 = getenv("MG_CANDIDATES_V1") != nullptr;
  if (fast && !v1 && d_keys && !d_bitmap && grid <= MAX_TAB_GRID && coord_cap < (1ll << 31)) {
    // workgroups of 1024 that run ~32 iterations per lane (the table costs each lane 1.5 entries once), but at least
    // ~1024 of them over all planes so that a single plane still fills the chip
    const int64_t per_plane = (num_iter + CT - 1) / CT;
    int64_t blocks = std::max<int64_t>((num_iter + 32 * CT - 1) / (32 * CT), (1024 + n_planes - 1) / n_planes);
    blocks = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(blocks, per_plane), 1024));
    const int span = 2 * grid - 1;
    static bool tab_attr = false;
    if (!tab_attr) {
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_candidates_tab), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (2 * MAX_TAB_GRID - 1) * (2 * MAX_TAB_GRID - 1) * (int)sizeof(double2)) != hipSuccess)
        return MG_ELAUNCH;
      tab_attr = true;
    }
    hipLaunchKernelGGL(k_candidates_tab, dim3((unsigned)blocks, n_planes), dim3(CT), (size_t)span * span * sizeof(double2),
                       mg_stream(stream), d_coords, coord_cap, d_cell_starts, d_cell_counts, d_num_edges, h, w, grid, gc,
                       gr * gc, d_seeds, (uint32_t)num_iter, min_r, max_r, d_raw, d_keys);
    MG_CHECK_LAUNCH();
    return MG_OK;
  }
  hipLaunchKernelGGL(fast ? k_candidates<true> : k_candidates<false>, dim3((unsigned)std::max<int64_t>(1, std::min<int64_t>((num_iter + NT - 1) / NT, 2048)), n_planes), dim3(NT), 0,
                     mg_stream(stream), d_coords,
                     coord_cap, d_cell_starts, d_cell_counts, d_num_edges, h, w, grid, gc, gr * gc, d_seeds, num_iter,
                     min_r, max_r, d_bitmap, bitmap_words, d_raw, d_keys);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
}  // namespace

extern "C" int mg_candidate_circles(const int32_t* d_coords, int64_t coord_cap, const int32_t* d_cell_starts,
                                    const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h,
                                    int w, int grid, const uint64_t* d_seeds, int64_t num_iter, int min_r, int max_r,
                                    uint32_t* d_bitmap, int64_t bitmap_words, float* d_raw, void* stream) {
  if (!d_bitmap) return MG_EINVAL;
  return launch_candidates(d_coords, coord_cap, d_cell_starts, d_cell_counts, d_num_edges, n_planes, h, w, grid, d_seeds,
                           num_iter, min_r, max_r, d_bitmap, bitmap_words, d_raw, nullptr, stream);
}

extern "C" int mg_candidate_keys(const int32_t* d_coords, int64_t coord_cap, const int32_t* d_cell_starts,
                                 const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h, int w,
                                 int grid, const uint64_t* d_seeds, int64_t num_iter, int min_r, int max_r,
                                 uint32_t* d_keys, float* d_raw, void* stream) {
  if (!d_keys) return MG_EINVAL;
  return launch_candidates(d_coords, coord_cap, d_cell_starts, d_cell_counts, d_num_edges, n_planes, h, w, grid, d_seeds,
                           num_iter, min_r, max_r, nullptr, 0, d_raw, d_keys, stream);
}

extern "C" int mg_dedup_layout(int h, int w, int min_r, int max_r, int* n_tile_rows, int* n_tile_cols,
                               int64_t* n_layers, int64_t* bitmap_words) {
  if (h <= 0 || w <= 0 || min_r < 0 || max_r < min_r || !n_tile_rows || !n_tile_cols || !n_layers || !bitmap_words)
    return MG_EINVAL;
  const int ntr = (h + 2 * max_r + TS - 1) / TS, ntc = (w + 2 * max_r + TS - 1) / TS;
  *n_tile_rows = ntr;
  *n_tile_cols = ntc;
  *n_layers = (int64_t)ntr * ntc * (max_r - min_r + 1);
  *bitmap_words = *n_layers * LAYER_WORDS;
  return MG_OK;
}

extern "C" int mg_bitmap_to_circles(uint32_t* d_bitmap, int64_t bitmap_words, int n_planes, int h, int w, int min_r,
                                    int max_r, int32_t* d_layer_offsets, int32_t* d_circles, int64_t circle_cap,
                                    int32_t* d_num_circles, void* stream) {
  if (!d_bitmap || !d_layer_offsets || !d_circles || !d_num_circles || n_planes < 0 || n_planes > 65535 ||
      circle_cap < 0)
    return MG_EINVAL;
  int ntr, ntc;
  int64_t n_layers, need_words;
  if (mg_dedup_layout(h, w, min_r, max_r, &ntr, &ntc, &n_layers, &need_words) != MG_OK) return MG_EINVAL;
  if (bitmap_words < need_words || n_layers > 0x7FFFFFF0) return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  const int nl = (int)n_layers, nr = max_r - min_r + 1;
  hipStream_t s = mg_stream(stream);
  const dim3 g((nl + 3) / 4, n_planes);
  hipLaunchKernelGGL(k_layer_count, g, dim3(NT), 0, s, d_bitmap, bitmap_words, nl, d_layer_offsets);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_layer_scan, dim3(n_planes), dim3(1024), 0, s, d_layer_offsets, nl, d_num_circles, circle_cap);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_layer_emit, g, dim3(NT), 0, s, d_bitmap, bitmap_words, nl, d_layer_offsets, ntc, nr, min_r, max_r,
                     d_circles, circle_cap);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_keys_to_circles(const uint32_t* d_keys, int64_t num_iter, const int32_t* d_cell_starts,
                                  const int32_t* d_cell_counts, const int32_t* d_num_edges, int n_planes, int h, int w,
                                  int grid, int min_r, int max_r, uint32_t* d_unique_keys, int64_t circle_cap,
                                  int32_t* d_tile_ranges, int32_t* d_num_circles, int32_t* d_layer_starts,
                                  int counters_clear, void* stream) {
  if (!d_keys || !d_cell_starts || !d_cell_counts || !d_num_edges || !d_unique_keys || !d_tile_ranges ||
      !d_num_circles || n_planes < 0 || n_planes > 65535 || circle_cap < 0 || num_iter < 0 || grid <= 0)
    return MG_EINVAL;
  int ntr, ntc;
  int64_t n_layers, need_words;
  if (mg_dedup_layout(h, w, min_r, max_r, &ntr, &ntc, &n_layers, &need_words) != MG_OK) return MG_EINVAL;
  const int nr = max_r - min_r + 1;
  if ((int64_t)ntr * ntc >= 32768 || nr > 32) return MG_EINVAL;  // the 32-bit key: 15 + 5 + 12 bits
  if ((TS + 2 * (max_r + 2)) / grid + 2 > MAX_RANGES || num_iter >= (1ll << 31)) return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  const int gr = (h + grid - 1) / grid, gc = (w + grid - 1) / grid;
  hipStream_t s = mg_stream(stream);
  if (ntr > 65535) return MG_EINVAL;
  if (!counters_clear && mg_zero_async(d_num_circles, (size_t)n_planes * sizeof(int32_t), s) != hipSuccess) return MG_ELAUNCH;
  // (round 4: super-tiles of 2 x 2 / 2 x 4 tiles on 512 / 1024 threads read every key 2.4 / 1.95 times instead of 3.8 and
  // ran 2.81 / 4.12 ms against 1.90: with 43 / 86 KB of layers a CU holds two / one workgroup, and clearing, counting and
  // emitting the layers -- not the key scan -- is where this kernel's instructions go; removed)
  const int tx = DEDUP_TX;  // 1 / 2 / 4 measured: 4.4 / 4.1 / 5.4 ms per step for the whole compaction
  hipLaunchKernelGGL(k_tile_dedup<DEDUP_TX>,
                     dim3((ntc + tx - 1) / tx, ntr, n_planes), dim3(NT), (size_t)tx * nr * LAYER_WORDS * 4, s, d_keys,
                     num_iter, d_cell_starts, d_cell_counts, d_num_edges, h, w, grid, gr, gc, ntc, nr, max_r,
                     d_unique_keys, circle_cap, d_tile_ranges, ntr * ntc, d_num_circles, d_layer_starts);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_clamp_counts, dim3((n_planes + 255) / 256), dim3(256), 0, s, d_num_circles, n_planes, circle_cap);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_score_circles(const float* d_angle, const uint32_t* d_edge_bits, const uint32_t* d_class_bits,
                                int64_t words_per_plane, int n_planes, int h, int w, int32_t* d_circles,
                                int64_t circle_cap, const int32_t* d_layer_offsets, const uint32_t* d_unique_keys,
                                const int32_t* d_tile_ranges, int min_r, int max_r, const int32_t* d_per_rc,
                                const double* d_per_expected, const int32_t* d_per_starts, int per_total,
                                float min_roundness, int write_skipped, int dedup_centres, float* d_scores,
                                int32_t* d_alive, int32_t* d_num_alive, int32_t* d_max_rc, int32_t* d_num_scored,
                                void* stream) {
  if (!d_angle || !d_edge_bits || !d_circles || !d_per_rc || !d_per_expected || !d_per_starts || !d_scores ||
      !d_alive || !d_num_alive || !d_max_rc)
    return MG_EINVAL;
  const bool keyed = d_unique_keys != nullptr;
  if (keyed ? !d_tile_ranges : !d_layer_offsets) return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || per_total <= 0 || max_r > 8000) return MG_EINVAL;
  int ntr, ntc;
  int64_t n_layers, words;
  if (mg_dedup_layout(h, w, min_r, max_r, &ntr, &ntc, &n_layers, &words) != MG_OK) return MG_EINVAL;
  if (keyed && ((int64_t)ntr * ntc >= 32768 || max_r - min_r + 1 > 32)) return MG_EINVAL;
  if (n_planes == 0 || circle_cap == 0) return MG_OK;
  const int side = TS + 2 * max_r;
  int wpr = 1;
  while (32 * wpr < side) wpr <<= 1;
  if (h >= (1 << 24) || w >= (1 << 24) || (int64_t)h * w >= (1LL << 31)) return MG_EINVAL;
  constexpr int SMALL_SLOT = 512;  // words: windows up to 128 x 128 (max_r <= 32)
  const bool small = side * wpr <= SMALL_SLOT;
  const size_t lds_bytes = ((size_t)4 * (small ? SMALL_SLOT : side * wpr) + per_total + CHUNK) * 4;
  if (lds_bytes > 150 * 1024) return MG_EINVAL;  // radii beyond ~150 px: outside this build's envelope
  auto kernel = small ? (keyed ? k_score_tiles<SMALL_SLOT, true> : k_score_tiles<SMALL_SLOT, false>)
                      : (keyed ? k_score_tiles<0, true> : k_score_tiles<0, false>);
  static bool attr_set[2] = {false, false};
  if (lds_bytes > 48 * 1024 && !attr_set[keyed]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            150 * 1024) != hipSuccess)
      return MG_ELAUNCH;
    attr_set[keyed] = true;
  }
  hipLaunchKernelGGL(kernel, dim3(ntr * ntc, n_planes), dim3(NT), lds_bytes, mg_stream(stream), d_angle, d_edge_bits,
                     d_class_bits, words_per_plane, h, w, d_circles, circle_cap, d_layer_offsets, d_unique_keys,
                     d_tile_ranges, ntr * ntc, (int)n_layers, max_r - min_r + 1, ntc, min_r, max_r, d_per_rc, per_total,
                     d_per_expected, d_per_starts, min_roundness, write_skipped, dedup_centres, d_scores, d_alive,
                     d_num_alive,
                     d_max_rc, d_num_scored);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

namespace {
// circles a wave of the suppression rounds looks up at once: 64 when the batch has many (late rounds are then a
// coalesced look at the state bytes), fewer when it has few (their rings are worked through one trip after another)
inline int nms_circles_per_wave(int64_t alive_bound, int n_planes) {
  const int64_t total = alive_bound * std::max(n_planes, 1);
  return total >= 262144 ? 64 : total >= 32768 ? 16 : 4;
}
}  // namespace

extern "C" int mg_nms_rounds(const int32_t* d_circles, int64_t circle_cap, const float* d_scores,
                             const int32_t* d_alive, const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes,
                             int min_dist, const int32_t* d_ring_rc, int ring_len, uint64_t* d_grid, int64_t grid_cap,
                             uint8_t* d_state, int32_t* d_undecided, int64_t undecided_stride, int n_rounds,
                             int counters_clear, const uint32_t* d_tie_keys, int64_t max_alive, const int32_t* d_skip,
                             void* stream) {
  if (!d_circles || !d_scores || !d_alive || !d_num_alive || !d_max_rc || !d_ring_rc || !d_grid || !d_state ||
      !d_undecided)
    return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || min_dist <= 0 || ring_len <= 0 || n_rounds < 0) return MG_EINVAL;
  if (n_rounds > 1 && undecided_stride < n_planes) return MG_EINVAL;
  if (n_planes == 0 || circle_cap == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  // the kernels walk d_alive with a grid-stride loop: max_alive (> 0: the caller's upper bound of
  // d_num_alive) only sizes the grid -- an all-capacity grid of empty blocks costs ~0.1 ms per launch
  const int64_t bound = max_alive > 0 ? std::min(max_alive, circle_cap) : circle_cap;
  const int cpw = nms_circles_per_wave(bound, n_planes);
  const dim3 g(grid_x(bound * (64 / cpw)), n_planes);
  for (int k = 0; k < n_rounds; ++k) {
    int32_t* und = d_undecided + (int64_t)k * undecided_stride;
    if (!counters_clear && mg_zero_async(und, sizeof(int32_t) * n_planes, s) != hipSuccess) return MG_ELAUNCH;
    hipLaunchKernelGGL((k_nms<0>), g, dim3(NT), 0, s, d_circles, circle_cap, d_scores, d_alive, d_num_alive, d_max_rc,
                       min_dist, d_ring_rc, ring_len, d_grid, grid_cap, d_state, und, d_tie_keys, cpw, d_skip);
    MG_CHECK_LAUNCH();
    hipLaunchKernelGGL((k_nms<1>), g, dim3(NT), 0, s, d_circles, circle_cap, d_scores, d_alive, d_num_alive, d_max_rc,
                       min_dist, d_ring_rc, ring_len, d_grid, grid_cap, d_state, und, d_tie_keys, cpw, d_skip);
    MG_CHECK_LAUNCH();
  }
  return MG_OK;
}

extern "C" int mg_nms_same_centre(const int32_t* d_circles, int64_t circle_cap, const float* d_scores,
                                  const int32_t* d_alive, const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes,
                                  int min_dist, uint64_t* d_grid, int64_t grid_cap, uint8_t* d_state,
                                  const uint32_t* d_tie_keys, int64_t max_alive, const int32_t* d_skip, void* stream) {
  if (!d_circles || !d_scores || !d_alive || !d_num_alive || !d_max_rc || !d_grid || !d_state) return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || min_dist <= 0) return MG_EINVAL;
  if (n_planes == 0 || circle_cap == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  const dim3 g(grid_x(max_alive > 0 ? std::min(max_alive, circle_cap) : circle_cap), n_planes);
  hipLaunchKernelGGL((k_nms<3>), g, dim3(NT), 0, s, d_circles, circle_cap, d_scores, d_alive, d_num_alive, d_max_rc,
                     min_dist, (const int32_t*)nullptr, 0, d_grid, grid_cap, d_state, (int32_t*)nullptr, d_tie_keys, 64, d_skip);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL((k_nms<4>), g, dim3(NT), 0, s, d_circles, circle_cap, d_scores, d_alive, d_num_alive, d_max_rc,
                     min_dist, (const int32_t*)nullptr, 0, d_grid, grid_cap, d_state, (int32_t*)nullptr, d_tie_keys, 64, d_skip);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL((k_nms<5>), g, dim3(NT), 0, s, d_circles, circle_cap, d_scores, d_alive, d_num_alive, d_max_rc,
                     min_dist, (const int32_t*)nullptr, 0, d_grid, grid_cap, d_state, (int32_t*)nullptr, d_tie_keys, 64, d_skip);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_nms_cleanup(const int32_t* d_circles, int64_t circle_cap, const float* d_scores,
                              const int32_t* d_alive, const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes,
                              int min_dist, const int32_t* d_ring_rc, int ring_len, uint64_t* d_grid, int64_t grid_cap,
                              uint8_t* d_state, int64_t max_alive, const int32_t* d_skip, void* stream) {
  if (!d_circles || !d_scores || !d_alive || !d_num_alive || !d_max_rc || !d_ring_rc || !d_grid || !d_state)
    return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || min_dist <= 0 || ring_len <= 0) return MG_EINVAL;
  if (n_planes == 0 || circle_cap == 0) return MG_OK;
  const int64_t bound = max_alive > 0 ? std::min(max_alive, circle_cap) : circle_cap;
  const int cpw = nms_circles_per_wave(bound, n_planes);
  hipLaunchKernelGGL((k_nms<2>), dim3(grid_x(bound * (64 / cpw)), n_planes),
                     dim3(NT), 0, mg_stream(stream), d_circles,
                     circle_cap, d_scores, d_alive, d_num_alive, d_max_rc, min_dist, d_ring_rc, ring_len, d_grid,
                     grid_cap, d_state, (int32_t*)nullptr, (const uint32_t*)nullptr, cpw, d_skip);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_nms_sparse(const int32_t* d_circles, int64_t circle_cap, const float* d_scores, const int32_t* d_alive,
                             const int32_t* d_num_alive, const int32_t* d_max_rc, int n_planes, int min_dist,
                             const uint32_t* d_dbits, uint8_t* d_state, const uint32_t* d_tie_keys, int32_t* d_done,
                             void* stream) {
  if (!d_circles || !d_scores || !d_alive || !d_num_alive || !d_max_rc || !d_dbits || !d_state || !d_done) return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || min_dist <= 0) return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  if (min_dist > SP_MAXD || circle_cap == 0) {  // nothing decided here: the rounds take every plane
    if (mg_zero_async(d_done, sizeof(int32_t) * n_planes, s) != hipSuccess) return MG_ELAUNCH;
    return MG_OK;
  }
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_nms_sparse), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)SP_LDS) != hipSuccess)
      return MG_ELAUNCH;
    attr_set = true;
  }
  hipLaunchKernelGGL(k_nms_sparse, dim3(n_planes), dim3(SP_NT), SP_LDS, s, d_circles, circle_cap, d_scores, d_alive,
                     d_num_alive, d_max_rc, min_dist, d_dbits, d_state, d_tie_keys, d_done);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

/* Words of the bitmap of D = ring (-) ring that mg_nms_sparse reads: bit (dr + 2 d) (4 d + 1) + dc + 2 d. */
extern "C" int mg_nms_sparse_max_dist(void) { return SP_MAXD; }

extern "C" int mg_collect_circles(const int32_t* d_circles, int64_t circle_cap, const float* d_scores,
                                  const int32_t* d_alive, const int32_t* d_num_alive, const uint8_t* d_state,
                                  int keep_all, int n_planes, int32_t* d_out, float* d_out_scores, int64_t out_cap,
                                  int32_t* d_num_out, int32_t* d_scratch, const uint32_t* d_tie_keys, int counters_clear,
                                  void* stream) {
  if (!d_circles || !d_scores || !d_alive || !d_num_alive || !d_out || !d_num_out || !d_scratch) return MG_EINVAL;
  if (!keep_all && !d_state) return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || out_cap < 0) return MG_EINVAL;
  if (n_planes == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  if (!counters_clear && mg_zero_async(d_num_out, sizeof(int32_t) * n_planes, s) != hipSuccess) return MG_ELAUNCH;
  if (circle_cap == 0 || out_cap == 0) return MG_OK;
  // out_cap bounds the alive counts in practice; the grid-stride loop covers the rest otherwise
  hipLaunchKernelGGL(k_collect_list, dim3(grid_x(std::min(circle_cap, std::max<int64_t>(out_cap, NT))), n_planes),
                     dim3(NT), 0, s, d_alive, d_num_alive, d_state,
                     keep_all, circle_cap, out_cap, d_scores, d_tie_keys, d_scratch, d_num_out);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_collect_rank, dim3(grid_x(out_cap), n_planes), dim3(NT), 0, s, d_circles, circle_cap, d_scores,
                     d_scratch, d_num_out, out_cap, d_out, d_out_scores);
  MG_CHECK_LAUNCH();
  hipLaunchKernelGGL(k_clamp_counts, dim3((n_planes + 255) / 256), dim3(256), 0, s, d_num_out, n_planes, out_cap);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
