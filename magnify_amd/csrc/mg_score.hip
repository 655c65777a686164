// A9-A10 on the keyed path: mean_grad / len(perimeter) (utils.py:183-192, 225-251) for the unique circles of
// mg_keys_to_circles, as two kernels.
//
//   k_prefilter  exact rejection of the circles that cannot reach min_roundness -- 99.6 % of them on noisy images.
//                A (persistent, 768-thread) workgroup owns a super-tile of MG_SCORE_SUBY x MG_SCORE_SUBX = 2 x 4 centre
//                tiles (128 x 256 positions); its window of the edge map lives in LDS as
//                ONE BYTE per pixel: the pixel's gradient-orientation bin (eighths of pi, decided exactly on the
//                integer gradient by mg_canny_nms) or 0x0C for "no edge".  A lane owns a circle; all 64 circles of
//                a wave have the SAME radius (the keys of a tile are sorted by radius: the block cuts the eight key
//                lists at the radius boundaries and deals 64-circle chunks of one radius to its waves), so the
//                perimeter walk is straight-line code per radius (template <R>, the midpoint circle evaluated at
//                compile time): the offset of a perimeter point is the immediate of its ds_read_u8 and the bound
//                table of the point sits in scalar registers.  Opposite points (p, -p) have the same radial
//                direction mod pi and share a table: their two bytes form the selector of ONE v_perm_b32, which
//                looks both up in the 8 signed bytes of the table (0x0C selects the constant 0), and one
//                v_dot4_i32_i8 adds both to the lane's sum.  A table byte is an upper bound, in 1/64, of the term
//                a pixel of that bin can contribute to the reference's sum; a circle whose bounds add up to less
//                than min_roundness * P cannot pass (every term <= its bound) and is dropped.  Survivors are
//                appended to a per-plane list.  Bounding roofline: LDS (one byte read per perimeter point).
//   k_exact      the survivors' reference sum (float64, sequential in perimeter order) in workgroup-level phases over
//                XS = 64 (16 at small batches) survivors: NT / XS lanes per survivor find the edge pixels on its
//                perimeter, a lane lists them in order, all lanes evaluate (survivor, hit) terms -- gradient angles
//                computed on demand from the blurred image exactly as mg_edge_angles does (the dense angle map and
//                its pass are not needed) --, a lane per survivor adds them in order with the reference-exact early
//                exit; score stored float32; circles that pass go to d_alive.
#include <math.h>

#include <algorithm>

#include "mg_common.h"

namespace {

constexpr int NT = 256;
constexpr int NP = 768;  // prefilter block: 8 waves share a window (4 blocks per CU by LDS: 32 waves per CU)
constexpr int TS = MG_SCORE_TILE;
constexpr int SUBY = MG_SCORE_SUBY, SUBX = MG_SCORE_SUBX, NSUB = SUBY * SUBX;  // centre tiles per super-tile
constexpr int STY = SUBY * TS, STX = SUBX * TS;
constexpr int WSTR = MG_SCORE_WSTRIDE;
constexpr int MAXR = MG_SCORE_MAX_R;
constexpr int MAXP = MG_SCORE_MAX_PAIRS;
constexpr int BIAS = MAXR * WSTR + MAXR;  // immediates BIAS +- (dr * WSTR + dc) are >= 0
constexpr int WBASE = 8192;               // LDS byte offset of the window (>= BIAS: lane addresses stay >= 0)
constexpr int SEGW = 33;                  // radii + 1 per sub-tile in the segment table
static_assert(WBASE >= BIAS && WBASE % 16 == 0, "window base");
static_assert(STX + 2 * MAXR <= WSTR && (WSTR % 4) == 0 && ((WSTR / 4) & 1) == 1, "window stride");
static_assert(2 * BIAS < 65536, "DS immediates are 16 bits");

__device__ __forceinline__ uint32_t bits_at(const uint32_t* __restrict__ bits, int64_t bit0, int n) {
  const int64_t wi = bit0 >> 5;
  const int sh = (int)(bit0 & 31);
  uint64_t two = bits[wi];
  if (sh + n > 32) two |= (uint64_t)bits[wi + 1] << 32;
  const uint32_t v = (uint32_t)(two >> sh);
  return n >= 32 ? v : (v & ((1u << n) - 1u));
}

// First point (dr, dc) of every pair of opposite perimeter points of radius R, in the order of
// mg_score_pair_table (mg_tables.hip: score_pairs) -- the reference's midpoint walk, utils.py:433-465.
template <int R>
struct Pairs {
  int n;
  int dr[MAXP], dc[MAXP];
  constexpr Pairs() : n(0), dr{}, dc{} {
    put(0, -R);
    put(-R, 0);
    int x = 1, y = -R;
    while (x < -y) {
      put(x, y);
      put(y, x);
      put(-x, y);
      put(-y, x);
      if (x * x + y * y - R * R <= 0) {
        ++x;
      } else {
        ++y;
        ++x;
      }
    }
    if (y == -x) {
      put(x, y);
      put(-x, y);
    }
  }
  constexpr void put(int a, int b) {
    dr[n] = a;
    dc[n] = b;
    ++n;
  }
};

typedef unsigned short us2 __attribute__((ext_vector_type(2)));

// Sum of the bounds over the perimeter of the circle whose centre byte sits at lds[vaddr + BIAS].
// `need`: what the sum has to reach (wave-uniform: a wave's circles share the radius).  A bound is at most 64 per
// point, so after k of n pairs a circle with sum + 128 (n - k) < need cannot reach it whatever the rest holds; when
// that is true of ALL 64 circles of the wave -- four checks in the last quarter of the walk: a wave of noise circles
// (81 % of the waves hold no survivor) is out at ~85 % of its perimeter -- the rest is not read.  The sums returned
// then are below `need` like the full ones would be: the same circles are dropped.
template <int R>
__device__ __forceinline__ int score_r(const uint8_t* lds, int vaddr, const uint2* __restrict__ tabs, int need = -(1 << 30),
                                       bool valid = true) {
  constexpr Pairs<R> P{};
  static_assert(P.n <= MAXP, "perimeter too long");
  int sum = 0;
#pragma unroll
  for (int k = 0; k < P.n; ++k) {
    const uint2 t = tabs[R * MAXP + k];  // uniform address: scalar loads
    const int off = P.dr[k] * WSTR + P.dc[k];
    us2 s;
    s.x = lds[vaddr + (BIAS + off)];
    s.y = lds[vaddr + (BIAS - off)];
    const uint32_t q = __builtin_amdgcn_perm(t.y, t.x, __builtin_bit_cast(uint32_t, s));  // bytes 0 and 2: the two bounds
    sum = __builtin_amdgcn_sdot4((int)q, 0x00010001, sum, false);
    const int done = k + 1;
    if (P.n >= 16 && done < P.n && (done == (P.n * 12) / 16 || done == (P.n * 13) / 16 || done == (P.n * 14) / 16 || done == (P.n * 15) / 16)) {
      if (__ballot(valid && sum + 128 * (P.n - done) >= need) == 0) return sum;
    }
  }
  return sum;
}

// ---- prefilter ----------------------------------------------------------------------------------------
// LDS: [0, WBASE) small tables, [WBASE, +side_y * WSTR) the window.
// Survivors are collected per super-tile in LDS and appended to the plane's list with ONE global atomic: a returning
// atomic per wave-with-survivors (~9 000 per plane, all on one address, while only one or two planes are being
// worked on at any time) cost as much as the perimeter walks of a single plane.
constexpr int SURV_OFF = 2048, SURV_LDS = (WBASE - SURV_OFF) / 8;
static_assert((NSUB * SEGW + SEGW + 32 + 8) * 4 <= SURV_OFF, "small tables");

// bits 0..3 of x -> bit 0 of bytes 0..3
__device__ __forceinline__ uint32_t spread4(uint32_t x) { return ((x & 0xFu) * 0x00204081u) & 0x01010101u; }

__global__ __launch_bounds__(NP) void k_prefilter(const uint32_t* __restrict__ d_bits, const uint32_t* __restrict__ d_class,
                                                  int64_t words_per_plane, int h, int w,
                                                  const uint32_t* __restrict__ d_ukeys, int64_t circle_cap,
                                                  const int32_t* __restrict__ d_layer_starts, int n_tiles, int ntr, int ntc,
                                                  int nsc, int n_st, int64_t total_st, int min_r, int max_r, int nr,
                                                  const uint2* __restrict__ d_tabs,
                                                  const int32_t* __restrict__ d_per_starts, float min_roundness,
                                                  int write_skipped, float* __restrict__ d_scores,
                                                  int32_t* __restrict__ d_surv, int64_t surv_cap,
                                                  int32_t* __restrict__ d_num_surv) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  int32_t* seg = reinterpret_cast<int32_t*>(lds);  // [NSUB][SEGW]: index of the first key of radius rho in sub-tile s
  int32_t* chunk0 = seg + NSUB * SEGW;             // [nr + 1]: first chunk of radius rho
  int32_t* need = chunk0 + SEGW;                   // [nr]: threshold on the sum of bounds (1/64)
  int32_t* next = need + 32;                       // the block's chunk counter
  int32_t* n_held = next + 1;                      // survivors of this super-tile held in LDS (may count past SURV_LDS)
  int32_t* g_base = next + 2;                      // where they go in the plane's list
  int2* held = reinterpret_cast<int2*>(lds + SURV_OFF);  // [SURV_LDS] (index in the key list, key)
  uint8_t* win = lds + WBASE;
  const int side_y = STY + 2 * max_r, side_x = STX + 2 * max_r;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < nr; i += NP) {
    const int len = d_per_starts[i + 1] - d_per_starts[i];
    // a circle can only pass with sum(terms) >= min_roundness * len - 1e-3 (margin far above any rounding of the
    // real sum), and sum(terms) <= sum(bounds) / 64
    need[i] = (int)ceil(64.0 * ((double)min_roundness * len - 1e-3));
  }
  const int wgroups = (side_x + 31) >> 5;  // 32-pixel groups per window row
  for (int64_t st = blockIdx.x; st < total_st; st += gridDim.x) {
    const int plane = (int)(st / n_st), sidx = (int)(st - (int64_t)plane * n_st);
    const int sr = sidx / nsc, sc = sidx - sr * nsc;
    __syncthreads();  // the previous super-tile's window and tables are no longer read
    // ---- segment table: first key of every radius in the sub-tiles' sorted lists (mg_keys_to_circles) ----
    if ((int)threadIdx.x < NSUB * (nr + 1)) {
      const int s = threadIdx.x / (nr + 1), q = threadIdx.x - s * (nr + 1);
      const int tr = SUBY * sr + s / SUBX, tc = SUBX * sc + s % SUBX;
      int v = 0;
      if (tr < ntr && tc < ntc) v = d_layer_starts[((int64_t)plane * n_tiles + tr * ntc + tc) * (nr + 1) + q];
      seg[s * SEGW + q] = v;  // a sub-tile beyond the grid: all zero = empty
    }
    if (threadIdx.x == 0) *next = 0, *n_held = 0;
    // ---- the window: orientation bin of every edge pixel, 0x0C elsewhere.  All loads of a thread's (at most
    // WI) 32-pixel groups are issued before the first is used: the block would otherwise wait for two to four
    // global round trips per group, one after the other ----
    const int wy0 = sr * STY - 2 * max_r, wx0 = sc * STX - 2 * max_r;
    {
      const uint32_t* pl[4] = {d_bits + plane * words_per_plane, d_class + (3 * plane) * words_per_plane,
                               d_class + (3 * plane + 1) * words_per_plane, d_class + (3 * plane + 2) * words_per_plane};
      constexpr int WI = ((STY + 2 * MAXR) * ((WSTR + 31) / 32) + NP - 1) / NP;
      uint32_t lo[WI][4], hi[WI][4];
#pragma unroll
      for (int it = 0; it < WI; ++it) {
        const int i = threadIdx.x + it * NP;
        const int j = i / wgroups, k = i - j * wgroups;
        const int y = wy0 + j, xs = wx0 + 32 * k;
        const int x_lo = max(xs, 0), x_hi = min(min(xs + 32, wx0 + side_x), w);
        const bool live = i < side_y * wgroups && y >= 0 && y < h && x_lo < x_hi;
        const int64_t wi = live ? ((int64_t)y * w + x_lo) >> 5 : 0;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          lo[it][c] = pl[c][wi];
          hi[it][c] = pl[c][wi + 1];  // (the bitmaps carry one spare word)
        }
      }
#pragma unroll
      for (int it = 0; it < WI; ++it) {
        const int i = threadIdx.x + it * NP;
        if (i >= side_y * wgroups) break;
        const int j = i / wgroups, k = i - j * wgroups;
        const int y = wy0 + j, xs = wx0 + 32 * k;
        const int x_lo = max(xs, 0), x_hi = min(min(xs + 32, wx0 + side_x), w);
        uint32_t pv[4] = {0u, 0u, 0u, 0u};  // edge bit, c0, c1, c2 of the group's 32 pixels
        if (y >= 0 && y < h && x_lo < x_hi) {
          const int sh = (int)(((int64_t)y * w + x_lo) & 31), n = x_hi - x_lo;
          const uint32_t keep = n >= 32 ? 0xFFFFFFFFu : ((1u << n) - 1u);
#pragma unroll
          for (int c = 0; c < 4; ++c)
            pv[c] = ((uint32_t)(((((uint64_t)hi[it][c]) << 32) | lo[it][c]) >> sh) & keep) << (x_lo - xs);
        }
        uint32_t* rowp = reinterpret_cast<uint32_t*>(win + j * WSTR) + 8 * k;
        const int nd = min(8, (WSTR >> 2) - 8 * k);  // dwords of this group inside the row
#pragma unroll
        for (int q = 0; q < 8; ++q) {
          if (q >= nd) break;
          // four pixels -> four bytes: bin = 4 c1 + 2 c0 + c2 where the pixel is an edge, else 0x0C
          const uint32_t bins = 4u * spread4(pv[2] >> (4 * q)) + 2u * spread4(pv[1] >> (4 * q)) + spread4(pv[3] >> (4 * q));
          const uint32_t m = spread4(pv[0] >> (4 * q)) * 0xFFu;
          rowp[q] = (bins & m) | (0x0C0C0C0Cu & ~m);
        }
      }
    }
    const uint32_t* ukeys = d_ukeys + (int64_t)plane * circle_cap;
    __syncthreads();
    if (wave == 0) {  // chunks of 64 circles per radius
      int cnt = 0;
      if (lane < nr)
        for (int s = 0; s < NSUB; ++s) cnt += seg[s * SEGW + lane + 1] - seg[s * SEGW + lane];
      const int chunks = (cnt + 63) >> 6;
      const int incl = mg_wave_scan_incl_i32(chunks);
      if (lane < nr) chunk0[lane] = incl - chunks;
      if (lane == nr - 1) chunk0[nr] = incl;
    }
    __syncthreads();
    const int total_chunks = chunk0[nr];
    // Chunk loop, software-pipelined: the keys of the NEXT chunk are requested before the current chunk is
    // scored (a wave's chunks are otherwise one global round trip each).
    int rho_n = 0, s_n = 0;
    int64_t i_n = 0;
    bool valid_n = false;
    uint32_t key_n = 0;
    auto fetch = [&]() -> bool {
      int c = 0;
      if (lane == 0) c = atomicAdd(next, 1);
      c = __builtin_amdgcn_readfirstlane(c);
      if (c >= total_chunks) return false;
      // the chunk's radius: the last rho with chunk0[rho] <= c (empty radii share their successor's start)
      rho_n = __builtin_popcountll(__ballot(lane < nr && chunk0[lane] <= c)) - 1;
      int m = 64 * (c - chunk0[rho_n]) + lane;  // position in the concatenation of the sub-tiles' segments of rho
      s_n = -1;
#pragma unroll
      for (int s = 0; s < NSUB; ++s) {
        const int a = seg[s * SEGW + rho_n], cnt = seg[s * SEGW + rho_n + 1] - a;
        if (s_n < 0) {
          if (m < cnt) {
            s_n = s;
            i_n = (int64_t)a + m;
          } else {
            m -= cnt;
          }
        }
      }
      valid_n = s_n >= 0;
      if (!valid_n) s_n = 0, i_n = 0;
      key_n = ukeys[i_n];  // (an idle lane reads a valid address: no branch around the load)
      return true;
    };
    bool have = fetch();
    while (have) {
      const int rho = rho_n, s = s_n;
      const int64_t i = i_n;
      const bool valid = valid_n;
      const uint32_t key = valid ? key_n : 0u;
      have = fetch();
      const int wrow = (s / SUBX) * TS + (int)((key >> 6) & 63u) + max_r, wcol = (s % SUBX) * TS + (int)(key & 63u) + max_r;
      const int vaddr = WBASE + wrow * WSTR + wcol - BIAS;
      int sum = 0;
      const int need_rho = need[rho];
      switch (rho + min_r) {
#define MG_CASE(R) case R: sum = score_r<R>(lds, vaddr, d_tabs, need_rho, valid); break;
          MG_CASE(2) MG_CASE(3) MG_CASE(4) MG_CASE(5) MG_CASE(6) MG_CASE(7) MG_CASE(8) MG_CASE(9) MG_CASE(10)
          MG_CASE(11) MG_CASE(12) MG_CASE(13) MG_CASE(14) MG_CASE(15) MG_CASE(16) MG_CASE(17) MG_CASE(18)
          MG_CASE(19) MG_CASE(20) MG_CASE(21) MG_CASE(22) MG_CASE(23) MG_CASE(24) MG_CASE(25) MG_CASE(26)
#undef MG_CASE
          default: break;
        }
      const bool pass = valid && sum >= need[rho];
      if (write_skipped && valid && !pass) d_scores[(int64_t)plane * circle_cap + i] = MG_SCORE_SKIPPED;
      const uint64_t pm = __ballot(pass);
      if (pm) {  // rare: hold the survivors in LDS until the super-tile is done
        int hbase = 0;
        if (lane == 0) hbase = atomicAdd(n_held, __builtin_popcountll(pm));
        hbase = __builtin_amdgcn_readfirstlane(hbase);
        const int slot = hbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(pm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)pm, 0u));
        if (pass && slot < SURV_LDS) held[slot] = make_int2((int32_t)i, (int32_t)key);
        const uint64_t om = __ballot(pass && slot >= SURV_LDS);
        if (om) {  // LDS full (hundreds of survivors in one super-tile): straight to the plane's list
          int sbase = 0;
          if (lane == 0) sbase = atomicAdd(&d_num_surv[plane], __builtin_popcountll(om));
          sbase = __builtin_amdgcn_readfirstlane(sbase);
          const int rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u));
          if (pass && slot >= SURV_LDS && (int64_t)sbase + rank < surv_cap)
            reinterpret_cast<int2*>(d_surv)[(int64_t)plane * surv_cap + sbase + rank] = make_int2((int32_t)i, (int32_t)key);
        }
      }
    }
    // ---- the super-tile's survivors -> the plane's list ----
    __syncthreads();
    const int n_out = min(*n_held, SURV_LDS);  // block-uniform
    if (n_out > 0) {
      if (threadIdx.x == 0) *g_base = atomicAdd(&d_num_surv[plane], n_out);
      __syncthreads();
      const int64_t gb = *g_base;
      for (int k = threadIdx.x; k < n_out; k += NP)
        if (gb + k < surv_cap) reinterpret_cast<int2*>(d_surv)[(int64_t)plane * surv_cap + gb + k] = held[k];
    }
  }
}

// ---- exact sums of the survivors --------------------------------------------------------------------
// The reference's sum is sequential per circle, but finding the edge pixels on a perimeter and evaluating their
// gradient angles is not, and most survivors are dropped after a handful of terms (sum + remaining hits can no
// longer reach the threshold).  A workgroup takes XS survivors per round:
//   1. four lanes per survivor test its perimeter points against the edge bitmap -> a hit mask per survivor;
//   2. a lane per survivor turns the mask into the list of its hits, in perimeter order;
//   3. all lanes evaluate (survivor, hit) pairs: the next `ch` hits of every survivor that is still undecided --
//      angle on demand from the blurred image, term in float64 -- `ch` grows as the undecided get fewer;
//   4. a lane per survivor adds its terms in order with the early exit, and the undecided are listed again.
// XS survivors per workgroup and round: 64 at large batches; 16 at small ones, where the survivors of a plane would
// otherwise sit in ~150 workgroups whose phases are chains of latencies (NT / XS lanes per survivor in phase 1,
// more hits per step in phase 3).
constexpr int XCH_MAX = 32;  // hits per survivor evaluated per step, at most
constexpr int XPMAX = 2 * MAXP;

template <int XS>
__global__ __launch_bounds__(NT) void k_exact(const uint8_t* __restrict__ d_blur, const float* __restrict__ d_angle,
                                              const uint32_t* __restrict__ d_bits, int64_t words_per_plane, int h, int w,
                                              int32_t* __restrict__ d_circles, int64_t circle_cap, int ntc, int min_r,
                                              int max_r, const int32_t* __restrict__ d_per_rc, int per_total,
                                              const double* __restrict__ d_per_expected,
                                              const int32_t* __restrict__ d_per_starts, float min_roundness,
                                              int write_skipped, float* __restrict__ d_scores,
                                              int32_t* __restrict__ d_alive, int32_t* __restrict__ d_num_alive,
                                              int32_t* __restrict__ d_max_rc, int32_t* __restrict__ d_num_scored,
                                              const int32_t* __restrict__ d_surv, int64_t surv_cap,
                                              const int32_t* __restrict__ d_num_surv) {
  extern __shared__ __attribute__((aligned(16))) int32_t tab[];  // [per_total] (dr << 16) | (dc & 0xFFFF)
  __shared__ double s_term[XS][XCH_MAX];
  __shared__ uint32_t s_mask[XS][XPMAX / 32];
  __shared__ uint8_t s_hits[XS][XPMAX];
  __shared__ int s_row[XS], s_col[XS], s_p0[XS], s_p1[XS], s_nh[XS], s_base[XS], s_list[XS], s_n;
  __shared__ int32_t starts[34];
  const int plane = blockIdx.y;
  const int64_t n = min((int64_t)d_num_surv[plane], surv_cap);
  if ((int64_t)blockIdx.x * XS >= n) return;  // block-uniform
  for (int i = threadIdx.x; i < per_total; i += NT) tab[i] = (d_per_rc[2 * i] << 16) | (d_per_rc[2 * i + 1] & 0xFFFF);
  if ((int)threadIdx.x <= max_r - min_r + 1) starts[threadIdx.x] = d_per_starts[threadIdx.x];
  if (blockIdx.x == 0 && threadIdx.x == 0 && d_num_scored) d_num_scored[plane] = (int32_t)n;
  int32_t* circles = d_circles + (int64_t)plane * circle_cap * 3;
  const uint32_t* bits = d_bits + plane * words_per_plane;
  const uint8_t* blur = d_blur + (int64_t)plane * h * w;
  const float* ang = d_angle ? d_angle + (int64_t)plane * h * w : nullptr;
  float* scores = d_scores + (int64_t)plane * circle_cap;
  const double PI = 3.141592653589793, INV_PI = 1.0 / 3.141592653589793;
  const int t = threadIdx.x, lane = t & 63;
  const int64_t rounds = (n + (int64_t)gridDim.x * XS - 1) / ((int64_t)gridDim.x * XS);
  for (int64_t rd = 0; rd < rounds; ++rd) {
    __syncthreads();  // tables in place / the previous round's lists are no longer read
    // 0. the round's survivors (threads 0..63 = wave 0 own one each)
    int64_t ci = 0;      // index in the plane's key list
    int rad = 0;
    bool active = false, dead = true;
    double acc = 0.0, floor_sum = 0.0;
    int left = 0;
    if (t < XS) {
      const int64_t k = (rd * gridDim.x + blockIdx.x) * XS + t;
      active = k < n;
      // record = (index in the plane's key list, key): written by the prefilter, no dependent load here
      const int2 rec = active ? reinterpret_cast<const int2*>(d_surv)[(int64_t)plane * surv_cap + k] : make_int2(0, 0);
      ci = rec.x;
      const uint32_t key = (uint32_t)rec.y;
      const int tile = (int)(key >> 17);
      rad = min_r + (int)((key >> 12) & 31u);
      s_row[t] = (tile / ntc) * TS - max_r + (int)((key >> 6) & 63u);
      s_col[t] = (tile % ntc) * TS - max_r + (int)(key & 63u);
      s_p0[t] = starts[rad - min_r];
      s_p1[t] = active ? starts[rad - min_r + 1] : s_p0[t];
      floor_sum = (double)min_roundness * (double)(s_p1[t] - s_p0[t]) - 1e-3;
    }
    for (int i = t; i < XS * (XPMAX / 32); i += NT) (&s_mask[0][0])[i] = 0u;
    __syncthreads();
    // 1. hit masks: NT / XS lanes per survivor, four loads in flight per lane
    {
      constexpr int LPS = NT / XS;
      const int s = t / LPS, j0 = t % LPS;
      const int row = s_row[s], col = s_col[s], p0 = s_p0[s], len = s_p1[s] - p0;
      for (int j = j0; j < len; j += 4 * LPS) {
        uint32_t wv[4];
        int bi[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jj = j + LPS * u;
          const int v = tab[p0 + min(jj, len - 1)];
          const int y = row + (v >> 16), x = col + (int)(int16_t)(v & 0xFFFF);
          const bool inb = jj < len && y >= 0 && y < h && x >= 0 && x < w;
          bi[u] = inb ? y * w + x : -1;
          wv[u] = bits[max(bi[u], 0) >> 5];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int jj = j + LPS * u;
          if (bi[u] >= 0 && ((wv[u] >> (bi[u] & 31)) & 1u)) atomicOr(&s_mask[s][jj >> 5], 1u << (jj & 31));
        }
      }
    }
    __syncthreads();
    // 2. hit lists in perimeter order; the undecided survivors
    if (t < XS) {
      int nh = 0;
#pragma unroll
      for (int wd = 0; wd < XPMAX / 32; ++wd) {
        uint32_t m = s_mask[t][wd];
        while (m) {
          s_hits[t][nh++] = (uint8_t)(32 * wd + __ffs(m) - 1);
          m &= m - 1;
        }
      }
      s_nh[t] = nh;
      s_base[t] = 0;
      left = nh;  // edge pixels not yet summed: each adds at most 1 (+1.2e-7, inside the margin)
      dead = !active || (double)left < floor_sum;
      const bool open = !dead && nh > 0;
      const uint64_t om = __ballot(open);
      if (open) s_list[__builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u))] = t;
      if (t == 0) s_n = __builtin_popcountll(om);
    }
    __syncthreads();
    // 3. + 4. until every survivor of the round is decided
    for (int na = s_n; na > 0; na = s_n) {
      const int ch = min(XCH_MAX, max(8, (2 * NT / na) & ~7));
      for (int pair = t; pair < na * ch; pair += NT) {
        const int s = s_list[pair / ch], j = pair % ch;
        const int idx = s_base[s] + j;
        if (idx < s_nh[s]) {
          const int p = s_p0[s] + s_hits[s][idx];
          const int v = tab[p];
          const int y = s_row[s] + (v >> 16), x = s_col[s] + (int)(int16_t)(v & 0xFFFF);
          const float a = ang ? ang[(int64_t)y * w + x] : mg_edge_angle(blur, h, w, y, x);
          double d = fabs((double)a - d_per_expected[p]);
          if (d > PI) d = d - PI;
          // x / pi, correctly rounded without the division (Markstein: y = RN(1/pi), q0 = RN(x y),
          // r = x - q0 pi exactly by FMA, q = RN(q0 + r y) == RN(x / pi); verified against x / pi on 1e9
          // operands of exactly this form)
          const double x4 = 4.0 * fabs(d - PI / 2.0);
          const double q0 = x4 * INV_PI;
          s_term[s][j] = fma(fma(-q0, PI, x4), INV_PI, q0) - 1.0;
        }
      }
      __syncthreads();
      if (t < XS) {
        bool open = false;
        if (!dead && s_base[t] < s_nh[t]) {
          const int cnt = min(ch, s_nh[t] - s_base[t]);
          for (int j = 0; j < cnt && !dead; ++j) {
            acc += s_term[t][j];
            --left;
            if (acc + (double)left < floor_sum) dead = true;  // exact: the remaining hits add <= 1 each
          }
          s_base[t] += cnt;
          open = !dead && s_base[t] < s_nh[t];
        }
        const uint64_t om = __ballot(open);
        if (open) s_list[__builtin_amdgcn_mbcnt_hi((uint32_t)(om >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)om, 0u))] = t;
        if (t == 0) s_n = __builtin_popcountll(om);
      }
      __syncthreads();
    }
    // 5. results
    if (t < XS && active) {
      if (dead) {
        if (write_skipped) scores[ci] = MG_SCORE_SKIPPED;
      } else {
        const float score = (float)acc / (float)(s_p1[t] - s_p0[t]);
        scores[ci] = score;
        if (score >= min_roundness) {
          const int k_ = atomicAdd(&d_num_alive[plane], 1);
          d_alive[(int64_t)plane * circle_cap + k_] = (int32_t)ci;
          circles[3 * ci] = s_row[t], circles[3 * ci + 1] = s_col[t], circles[3 * ci + 2] = rad;
          atomicMax(&d_max_rc[2 * plane], s_row[t]);
          atomicMax(&d_max_rc[2 * plane + 1], s_col[t]);
        }
      }
    }
  }
  (void)lane;
}

}  // namespace

extern "C" int mg_score_keyed_supported(int min_r, int max_r) {
  return (min_r >= 2 && max_r >= min_r && max_r <= MG_SCORE_MAX_R && max_r - min_r + 1 <= 32) ? 1 : 0;
}

extern "C" int mg_score_circles_keyed(const uint8_t* d_blur, const float* d_angle, const uint32_t* d_edge_bits,
                                      const uint32_t* d_class_bits, int64_t words_per_plane, int n_planes, int h, int w,
                                      int32_t* d_circles, int64_t circle_cap, const uint32_t* d_unique_keys,
                                      const int32_t* d_layer_starts, int min_r, int max_r, const int32_t* d_per_rc,
                                      const double* d_per_expected, const int32_t* d_per_starts, int per_total,
                                      const uint64_t* d_pair_table, float min_roundness, int write_skipped,
                                      float* d_scores, int32_t* d_alive, int32_t* d_num_alive, int32_t* d_max_rc,
                                      int32_t* d_num_scored, int32_t* d_surv_list, int64_t surv_cap, int32_t* d_num_surv,
                                      int counters_clear, void* stream) {
  if ((!d_blur && !d_angle) || !d_edge_bits || !d_class_bits || !d_circles || !d_unique_keys || !d_layer_starts ||
      !d_per_rc || !d_per_expected || !d_per_starts || !d_pair_table || !d_scores || !d_alive || !d_num_alive ||
      !d_max_rc || !d_surv_list || !d_num_surv)
    return MG_EINVAL;
  if (n_planes < 0 || n_planes > 65535 || per_total <= 0 || surv_cap < circle_cap) return MG_EINVAL;  // (a survivor per circle)
  if (mg_score_keyed_supported(min_r, max_r) != 1) return MG_EINVAL;
  if (h <= 0 || w <= 0 || h >= (1 << 24) || w >= (1 << 24) || (int64_t)h * w >= (1LL << 31)) return MG_EINVAL;
  const int nr = max_r - min_r + 1;
  int ntr, ntc;
  int64_t n_layers, words;
  if (mg_dedup_layout(h, w, min_r, max_r, &ntr, &ntc, &n_layers, &words) != MG_OK) return MG_EINVAL;
  if ((int64_t)ntr * ntc >= 32768) return MG_EINVAL;  // the 32-bit key
  if ((size_t)per_total * 12 > 48 * 1024 || NSUB * (nr + 1) > NP) return MG_EINVAL;
  hipStream_t s = mg_stream(stream);
  if (!counters_clear && mg_zero_async(d_num_surv, (size_t)std::max(n_planes, 1) * sizeof(int32_t), s) != hipSuccess)
    return MG_ELAUNCH;
  if (n_planes == 0 || circle_cap == 0) return MG_OK;
  const size_t lds_bytes = (size_t)WBASE + (size_t)(STY + 2 * max_r) * WSTR;
  const int nsr = (ntr + SUBY - 1) / SUBY, nsc = (ntc + SUBX - 1) / SUBX, n_st = nsr * nsc;
  const int64_t total_st = (int64_t)n_st * n_planes;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(k_prefilter), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)(WBASE + (STY + 2 * MAXR) * WSTR)) != hipSuccess)
      return MG_ELAUNCH;
    attr_set = true;
  }
  // persistent blocks, super-tiles dealt round-robin (neighbouring super-tiles run at the same time)
  const int blocks = (int)std::min<int64_t>(total_st, 256 * 2 * 8);
  hipLaunchKernelGGL(k_prefilter, dim3(blocks), dim3(NP), lds_bytes, s, d_edge_bits, d_class_bits, words_per_plane, h, w,
                     d_unique_keys, circle_cap, d_layer_starts, ntr * ntc, ntr, ntc, nsc, n_st, total_st, min_r, max_r, nr,
                     reinterpret_cast<const uint2*>(d_pair_table), d_per_starts, min_roundness, write_skipped, d_scores,
                     d_surv_list, surv_cap, d_num_surv);
  MG_CHECK_LAUNCH();
  // blocks per plane: enough to fill the chip at any batch size, few enough to amortise the table load
  const int xblocks = std::max(16, std::min(256, 4096 / std::max(n_planes, 1)));
  if (n_planes <= 4)
    hipLaunchKernelGGL(k_exact<16>, dim3(4 * xblocks, n_planes), dim3(NT), (size_t)per_total * 4, s, d_blur, d_angle,
                       d_edge_bits, words_per_plane, h, w, d_circles, circle_cap, ntc, min_r, max_r, d_per_rc, per_total,
                       d_per_expected, d_per_starts, min_roundness, write_skipped, d_scores, d_alive, d_num_alive, d_max_rc,
                       d_num_scored, d_surv_list, surv_cap, d_num_surv);
  else
    hipLaunchKernelGGL(k_exact<64>, dim3(xblocks, n_planes), dim3(NT), (size_t)per_total * 4, s, d_blur, d_angle,
                       d_edge_bits, words_per_plane, h, w, d_circles, circle_cap, ntc, min_r, max_r, d_per_rc, per_total,
                       d_per_expected, d_per_starts, min_roundness, write_skipped, d_scores, d_alive, d_num_alive, d_max_rc,
                       d_num_scored, d_surv_list, surv_cap, d_num_surv);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
