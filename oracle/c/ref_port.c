/* oracle/c/ref_port.c -- plain-C restatement of the reference's bead hot path.
 *
 * TEST INFRASTRUCTURE (see oracle/__init__.py).  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library; the product never does.
 *
 * It restates, function for function, the NumPy oracle next to it (oracle/ref_numeric.py,
 * ref_opencv.py, ref_pipeline.py), which in turn is pinned bit-for-bit by tests/golden (vectors
 * produced by executing the reference's own utils.py / find.py).  tests/test_oracle_cport.py checks
 * this file against those goldens and against the NumPy oracle; its purpose is (1) an oracle that
 * finishes full-size (4096 x 4096) planes in seconds and (2) the timed CPU baseline of bench.py,
 * run on all host cores with one assay (time slice) per OpenMP thread.
 *
 * Floating point follows the reference's operation order and dtypes; build with
 * -ffp-contract=off and without -ffast-math (oracle/Makefile).  Two spots depend on libm rather
 * than NumPy's SIMD kernels (float64 atan2 of the perimeter offsets, float32 arctan2 of the
 * gradient): they may differ from the NumPy oracle in the last bit, see the test tolerances.
 *
 * Reference lines (src/magnify/...) are cited at every function.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define TG22 13573 /* round(tan(22.5 deg) * 2^15): cv::Canny's fixed-point tangent */
#define CANON_TILE 64

static inline int reflect101(int i, int n) { /* cv::borderInterpolate, BORDER_REFLECT_101 */
  if (n == 1) return 0;
  while (i < 0 || i >= n) i = i < 0 ? -i : 2 * n - 2 - i;
  return i;
}

/* ------------------------------------------------------------------------------------------
 * A2  flatfield_correct (preprocess.py:83-87): float64, two global maxima, truncating cast
 * ---------------------------------------------------------------------------------------- */
int ref_flatfield_correct_u16(const uint16_t* tiles, int64_t n_total, int64_t plane_px, const float* flat_img,
                              double flat_scalar, double dark, uint16_t* out) {
  if (n_total < 0 || plane_px <= 0) return -1;
  double m1 = -INFINITY, m2 = -INFINITY;
  for (int64_t i = 0; i < n_total; ++i) {
    double t = (double)tiles[i] - dark;
    if (t < 0.0) t = 0.0;
    if (t > m1) m1 = t;
    const double f = flat_img ? (double)flat_img[i % plane_px] : flat_scalar;
    const double u = t / f;
    if (u > m2) m2 = u;
  }
  for (int64_t i = 0; i < n_total; ++i) {
    double t = (double)tiles[i] - dark;
    if (t < 0.0) t = 0.0;
    const double f = flat_img ? (double)flat_img[i % plane_px] : flat_scalar;
    const double v = (t / f) * m1 / m2;
    out[i] = (uint16_t)v; /* NaN (all-zero input) is undefined in the reference too */
  }
  return 0;
}

/* ------------------------------------------------------------------------------------------
 * A3  to_uint8 (utils.py:20-27)
 * ---------------------------------------------------------------------------------------- */
void ref_to_uint8_u16(const uint16_t* src, int64_t n, uint8_t* dst) {
  if (n <= 0) return;
  double mn = (double)src[0], mx = (double)src[0];
  for (int64_t i = 1; i < n; ++i) {
    const double v = (double)src[i];
    if (v < mn) mn = v;
    if (v > mx) mx = v;
  }
  const double top = mx - mn;
  for (int64_t i = 0; i < n; ++i) {
    double a = (double)src[i] - mn;
    if (top > 0.0) a = 255.0 * a / top;
    dst[i] = (uint8_t)a;
  }
}

/* ------------------------------------------------------------------------------------------
 * A4  cv.GaussianBlur(u8, (5,5), 0) (call site utils.py:115): [1 4 6 4 1]^2, (sum + 128) >> 8
 * ---------------------------------------------------------------------------------------- */
void ref_gaussian_blur5(const uint8_t* img, int h, int w, uint8_t* out) {
  if (h <= 0 || w <= 0) return;
  static const int k[5] = {1, 4, 6, 4, 1};
  int32_t* horiz = (int32_t*)malloc((size_t)h * w * sizeof(int32_t));
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int j = 0; j < 5; ++j) s += k[j] * img[(int64_t)y * w + reflect101(x + j - 2, w)];
      horiz[(int64_t)y * w + x] = s;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      int s = 0;
      for (int i = 0; i < 5; ++i) s += k[i] * horiz[(int64_t)reflect101(y + i - 2, h) * w + x];
      out[(int64_t)y * w + x] = (uint8_t)((s + 128) >> 8);
    }
  free(horiz);
}

/* ------------------------------------------------------------------------------------------
 * A5  cv.Scharr(u8, CV_32F, 1, 0) / (0, 1) (utils.py:118-119): exact integers, |.| <= 4080
 * ---------------------------------------------------------------------------------------- */
void ref_scharr(const uint8_t* img, int h, int w, int16_t* dx, int16_t* dy) {
  for (int y = 0; y < h; ++y) {
    const uint8_t* r0 = img + (int64_t)reflect101(y - 1, h) * w;
    const uint8_t* r1 = img + (int64_t)y * w;
    const uint8_t* r2 = img + (int64_t)reflect101(y + 1, h) * w;
    for (int x = 0; x < w; ++x) {
      const int xl = reflect101(x - 1, w), xr = reflect101(x + 1, w);
      dx[(int64_t)y * w + x] = (int16_t)(3 * (r0[xr] - r0[xl]) + 10 * (r1[xr] - r1[xl]) + 3 * (r2[xr] - r2[xl]));
      dy[(int64_t)y * w + x] = (int16_t)(3 * (r2[xl] - r0[xl]) + 10 * (r2[x] - r0[x]) + 3 * (r2[xr] - r0[xr]));
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * A6  np.quantile(grad, q) for float32 grad = sqrt(dx^2 + dy^2) (utils.py:120-126)
 * k-th smallest of non-negative floats by a two-level radix histogram on the bit patterns.
 * ---------------------------------------------------------------------------------------- */
static float select_kth(const float* v, int64_t n, int64_t k) {
  int64_t* hist = (int64_t*)calloc(65536, sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) {
    uint32_t b;
    memcpy(&b, &v[i], 4);
    hist[b >> 16]++;
  }
  int64_t acc = 0;
  uint32_t hi = 0;
  for (; hi < 65536; ++hi) {
    if (acc + hist[hi] > k) break;
    acc += hist[hi];
  }
  memset(hist, 0, 65536 * sizeof(int64_t));
  for (int64_t i = 0; i < n; ++i) {
    uint32_t b;
    memcpy(&b, &v[i], 4);
    if ((b >> 16) == hi) hist[b & 0xFFFFu]++;
  }
  uint32_t lo = 0;
  for (; lo < 65536; ++lo) {
    if (acc + hist[lo] > k) break;
    acc += hist[lo];
  }
  free(hist);
  const uint32_t bits = (hi << 16) | lo;
  float r;
  memcpy(&r, &bits, 4);
  return r;
}

/* numpy 2.x np.quantile(float32 array, python float q), method "linear": the virtual index is
 * float32 arithmetic ((n - 1) * float32(q)), interpolation is numpy's _lerp in float32. */
float ref_quantile_f32(const float* v, int64_t n, double q) {
  const float qf = (float)q;
  const float vi = (float)(n - 1) * qf;
  const float prev = floorf(vi);
  int64_t ip, in;
  if (vi >= (float)(n - 1)) ip = in = n - 1;
  else if (vi < 0.0f) ip = in = 0;
  else {
    ip = (int64_t)prev;
    in = ip + 1;
  }
  const float g = vi - prev;
  const float a = select_kth(v, n, ip), b = select_kth(v, n, in);
  const float diff = b - a;
  if (g >= 0.5f) return b - diff * (1.0f - g);
  return a + diff * g;
}

/* Threshold preparation of cv::Canny(dx, dy, t1, t2, L2gradient=true) (utils.py:128-134). */
void ref_canny_thresholds(double lo, double hi, int64_t* low, int64_t* high) {
  if (lo > hi) {
    const double t = lo;
    lo = hi;
    hi = t;
  }
  if (lo > 32767.0) lo = 32767.0;
  if (hi > 32767.0) hi = 32767.0;
  if (lo > 0) lo *= lo;
  if (hi > 0) hi *= hi;
  *low = (int64_t)floor(lo);
  *high = (int64_t)floor(hi);
}

/* cv::Canny NMS + double threshold: map 1 = no edge, 0 = weak candidate, 2 = strong. */
void ref_canny_nms(const int16_t* dx, const int16_t* dy, int h, int w, int64_t low, int64_t high, uint8_t* map) {
  int32_t* mag = (int32_t*)calloc((size_t)(h + 2) * (w + 2), sizeof(int32_t));
  const int64_t ms = w + 2;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int32_t a = dx[(int64_t)y * w + x], b = dy[(int64_t)y * w + x];
      mag[(int64_t)(y + 1) * ms + x + 1] = a * a + b * b;
    }
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const int32_t* c = mag + (int64_t)(y + 1) * ms + x + 1;
      const int32_t m = *c;
      uint8_t v = 1;
      if ((int64_t)m > low) {
        const int32_t xs = dx[(int64_t)y * w + x], ys = dy[(int64_t)y * w + x];
        const int64_t ax = xs < 0 ? -xs : xs, ay = (int64_t)(ys < 0 ? -ys : ys) << 15;
        const int64_t tg22x = ax * TG22, tg67x = tg22x + (ax << 16);
        int is_max;
        if (ay < tg22x) is_max = m > c[-1] && m >= c[1];
        else if (ay > tg67x) is_max = m > c[-ms] && m >= c[ms];
        else {
          const int s = (xs ^ ys) < 0 ? -1 : 1;
          is_max = m > c[-ms - s] && m > c[ms + s];
        }
        if (is_max) v = (int64_t)m > high ? 2 : 0;
      }
      map[(int64_t)y * w + x] = v;
    }
  free(mag);
}

/* 8-connected hysteresis: candidates connected to a strong pixel become edges ({0,1} map). */
void ref_canny_hysteresis(const uint8_t* map, int h, int w, uint8_t* edges) {
  const int64_t n = (int64_t)h * w;
  memset(edges, 0, (size_t)n);
  int64_t* stack = (int64_t*)malloc((size_t)(n > 0 ? n : 1) * sizeof(int64_t));
  int64_t top = 0;
  for (int64_t i = 0; i < n; ++i)
    if (map[i] == 2) {
      edges[i] = 1;
      stack[top++] = i;
    }
  while (top > 0) {
    const int64_t i = stack[--top];
    const int y = (int)(i / w), x = (int)(i % w);
    for (int dy = -1; dy <= 1; ++dy)
      for (int dx = -1; dx <= 1; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
        const int64_t j = (int64_t)yy * w + xx;
        if (!edges[j] && map[j] != 1) {
          edges[j] = 1;
          stack[top++] = j;
        }
      }
  }
  free(stack);
}

/* Steps 1-2 of find_circles (utils.py:115-142).  lohi = the two float32 quantiles. */
void ref_edge_stage(const uint8_t* img, int h, int w, double low_q, double high_q, uint8_t* blur, int16_t* dx,
                    int16_t* dy, uint8_t* edges, float* lohi) {
  const int64_t n = (int64_t)h * w;
  if (n <= 0) return;
  ref_gaussian_blur5(img, h, w, blur);
  ref_scharr(blur, h, w, dx, dy);
  float* grad = (float*)malloc((size_t)n * sizeof(float));
  for (int64_t i = 0; i < n; ++i) {
    const float a = (float)dx[i], b = (float)dy[i];
    const float aa = a * a, bb = b * b;
    grad[i] = sqrtf(aa + bb); /* float32 throughout, as np.sqrt(dx**2 + dy**2) */
  }
  lohi[0] = ref_quantile_f32(grad, n, low_q);
  lohi[1] = ref_quantile_f32(grad, n, high_q);
  free(grad);
  int64_t low, high;
  ref_canny_thresholds((double)lohi[0], (double)lohi[1], &low, &high);
  uint8_t* map = (uint8_t*)malloc((size_t)n);
  ref_canny_nms(dx, dy, h, w, low, high, map);
  ref_canny_hysteresis(map, h, w, edges);
  free(map);
}

/* ------------------------------------------------------------------------------------------
 * A7  grid_array (utils.py:347-377): counts, CSR starts, cell-major coordinates.  Returns E.
 * ---------------------------------------------------------------------------------------- */
int64_t ref_grid_array(const uint8_t* edges, int h, int w, int g, int32_t* coords, int64_t* starts, int64_t* counts) {
  const int gr = (h + g - 1) / g, gc = (w + g - 1) / g;
  memset(counts, 0, (size_t)gr * gc * sizeof(int64_t));
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      if (edges[(int64_t)y * w + x]) counts[(int64_t)(y / g) * gc + x / g]++;
  int64_t acc = 0;
  for (int64_t c = 0; c < (int64_t)gr * gc; ++c) {
    starts[c] = acc;
    acc += counts[c];
  }
  if (coords) {
    int64_t* pos = (int64_t*)malloc((size_t)gr * gc * sizeof(int64_t));
    memcpy(pos, starts, (size_t)gr * gc * sizeof(int64_t));
    for (int y = 0; y < h; ++y) /* row-major scan keeps (row, col) order inside every cell */
      for (int x = 0; x < w; ++x)
        if (edges[(int64_t)y * w + x]) {
          const int64_t p = pos[(int64_t)(y / g) * gc + x / g]++;
          coords[2 * p] = y;
          coords[2 * p + 1] = x;
        }
    free(pos);
  }
  return acc;
}

/* ------------------------------------------------------------------------------------------
 * digital circles (utils.py:433-465, 398-430)
 * ---------------------------------------------------------------------------------------- */
/* Perimeter offsets (row, col) in the reference's emission order; returns the count. */
int ref_circle_points(int r, int four_connected, int32_t* pts /* cap >= 8 * (r + 1) + 4 pairs */) {
  int n = 0;
#define PUT(a, b) (pts[2 * n] = (a), pts[2 * n + 1] = (b), ++n)
  PUT(0, -r);
  PUT(-r, 0);
  PUT(0, r);
  PUT(r, 0);
  int x = 1, y = -r;
  while (x < -y) {
    PUT(x, y);
    PUT(y, x);
    PUT(-x, y);
    PUT(-y, x);
    PUT(x, -y);
    PUT(y, -x);
    PUT(-x, -y);
    PUT(-y, -x);
    if (x * x + y * y - r * r <= 0) x += 1;
    else {
      y += 1;
      if (!four_connected) x += 1;
    }
  }
  if (y == -x) {
    PUT(x, y);
    PUT(-x, -y);
    PUT(-x, y);
    PUT(x, -y);
  }
#undef PUT
  return n;
}

/* Disk = perimeter + row-wise interior fill; returns the count (r >= 2). */
int ref_filled_circle_points(int r, int32_t* pts /* cap >= (2r+1)^2 pairs */) {
  if (r < 2) return -1;
  int n = ref_circle_points(r, 0, pts);
  const int size = 2 * r + 1;
  uint8_t* mask = (uint8_t*)calloc((size_t)size * (size + 1), 1); /* one spare column */
  for (int i = 0; i < n; ++i) mask[(pts[2 * i] + r) * (size + 1) + pts[2 * i + 1] + r] = 1;
  for (int i = 0; i < size; ++i) {
    const uint8_t* row = mask + i * (size + 1);
    int j = 0;
    while (j < size && !row[j]) ++j;
    while (j < size && row[j]) ++j;
    if (j <= r)
      while (j < size && !row[j]) {
        pts[2 * n] = i - r;
        pts[2 * n + 1] = j - r;
        ++n;
        ++j;
      }
  }
  free(mask);
  return n;
}

/* A13 circle_labels (utils.py:380-395): -1 nobody, i exactly bead i, -2 contested. */
void ref_circle_labels(const int32_t* circles, int m, int h, int w, int32_t* labels) {
  for (int64_t i = 0; i < (int64_t)h * w; ++i) labels[i] = -1;
  int32_t* pts = NULL;
  int cached_r = -1, np_ = 0;
  for (int i = 0; i < m; ++i) {
    const int r = circles[3 * i + 2];
    if (r != cached_r) {
      free(pts);
      pts = (int32_t*)malloc((size_t)(2 * r + 1) * (2 * r + 1) * 2 * sizeof(int32_t));
      np_ = ref_filled_circle_points(r, pts);
      cached_r = r;
    }
    for (int k = 0; k < np_; ++k) {
      const int y = circles[3 * i] + pts[2 * k], x = circles[3 * i + 1] + pts[2 * k + 1];
      if (y < 0 || y >= h || x < 0 || x >= w) continue;
      int32_t* l = labels + (int64_t)y * w + x;
      *l = (*l != -1) ? -2 : i;
    }
  }
  free(pts);
}

/* A15 bounding_box (utils.py:60-80): (x, y, L, W, H) -> top, bottom, left, right. */
void ref_bounding_box(int x, int y, int len, int width, int height, int* tblr) {
  const int lo = len / 2, hi = len - len / 2;
  int a = y - lo, b = y + hi;
  if (a < 0) b -= a, a = 0;
  if (b > height) a -= b - height, b = height;
  tblr[0] = a, tblr[1] = b;
  a = x - lo, b = x + hi;
  if (a < 0) b -= a, a = 0;
  if (b > width) a -= b - width, b = width;
  tblr[2] = a, tblr[3] = b;
}

/* ------------------------------------------------------------------------------------------
 * A8  candidate_circles (utils.py:295-344) with the build's explicit RNG stream
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mix64(uint64_t z) {
  z ^= z >> 30;
  z *= 0xBF58476D1CE4E5B9ull;
  z ^= z >> 27;
  z *= 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
/* k-th (0,1,2) 32-bit uniform of iteration it: oracle/ref_numeric.py draw_uniform32 */
uint32_t ref_draw_uniform32(uint64_t seed, uint64_t it, int k) {
  /* k = 0: top half at counter 3 it + 1; k = 1, 2: top / bottom half at counter 3 it + 2 */
  const uint64_t ctr = it * 3ull + (k == 0 ? 1ull : 2ull);
  const uint64_t z = mix64(seed + ctr * 0x9E3779B97F4A7C15ull);
  return k == 2 ? (uint32_t)z : (uint32_t)(z >> 32);
}

/* Circle through p0, p1, p2 in the reference's arithmetic (utils.py:319-342); out = row, col, r. */
void ref_circumcircle(const int32_t* p0, const int32_t* p1, const int32_t* p2, float* out) {
  const int64_t q1r = (int64_t)p1[0] - p0[0], q1c = (int64_t)p1[1] - p0[1];
  const int64_t q2r = (int64_t)p2[0] - p0[0], q2c = (int64_t)p2[1] - p0[1];
  const double eps = (double)1e-20f;
  const double mid1r = 0.5 * (double)q1r, mid1c = 0.5 * (double)q1c;
  const double mid2r = 0.5 * (double)q2r, mid2c = 0.5 * (double)q2c;
  const double m1 = (double)(-q1c) / ((double)q1r + eps);
  const double m2 = (double)(-q2c) / ((double)q2r + eps);
  const double b1 = mid1r - m1 * mid1c;
  const double b2 = mid2r - m2 * mid2c;
  const float c_col = (float)((b1 - b2) / (m2 - m1 + eps));
  const float c_row = (float)(m1 * (double)c_col + b1);
  const float rr = c_row * c_row, cc = c_col * c_col;
  out[2] = sqrtf(rr + cc);
  out[0] = (float)((double)c_row + (double)p0[0]);
  out[1] = (float)((double)c_col + (double)p0[1]);
}

/* ------------------------------------------------------------------------------------------
 * A9-A11  filter_circles (utils.py:149-199), mean_grad (utils.py:225-251),
 *         filter_neighbors (utils.py:254-292)
 * ---------------------------------------------------------------------------------------- */
typedef struct {
  int32_t row, col, r;
  float score;
} Scored;

static int cmp_u64(const void* a, const void* b) {
  const uint64_t x = *(const uint64_t*)a, y = *(const uint64_t*)b;
  return x < y ? -1 : x > y;
}
static int g_max_r_for_sort; /* set per call under the caller's thread: see cmp_canonical */
#ifdef _OPENMP
#pragma omp threadprivate(g_max_r_for_sort)
#endif
/* score descending, then (tile_row, tile_col, r, row, col): oracle/ref_numeric.py canonical_order */
static int cmp_canonical(const void* pa, const void* pb) {
  const Scored* a = (const Scored*)pa;
  const Scored* b = (const Scored*)pb;
  if (a->score != b->score) return a->score > b->score ? -1 : 1;
  const int mr = g_max_r_for_sort;
  const int atr = (a->row + mr) / CANON_TILE, btr = (b->row + mr) / CANON_TILE;
  if (atr != btr) return atr < btr ? -1 : 1;
  const int atc = (a->col + mr) / CANON_TILE, btc = (b->col + mr) / CANON_TILE;
  if (atc != btc) return atc < btc ? -1 : 1;
  if (a->r != b->r) return a->r < b->r ? -1 : 1;
  if (a->row != b->row) return a->row < b->row ? -1 : 1;
  if (a->col != b->col) return a->col < b->col ? -1 : 1;
  return 0;
}

static inline int64_t pymod(int64_t a, int64_t n) {
  const int64_t r = a % n;
  return r < 0 ? r + n : r;
}

/* Greedy suppression on the wrapping claim grid; keep[i] in {0,1}.  circles in score order. */
void ref_filter_neighbors(const int32_t* circles, int n, int min_dist, uint8_t* keep) {
  if (n == 0) return;
  int32_t* ring = (int32_t*)malloc((size_t)(8 * (min_dist + 1) + 4) * 2 * 2 * sizeof(int32_t));
  const int nr = ref_circle_points(min_dist, 1, ring);
  const int64_t pad = 2 * (int64_t)min_dist + 1;
  int64_t max_row = circles[0], max_col = circles[1];
  for (int i = 1; i < n; ++i) {
    if (circles[3 * i] > max_row) max_row = circles[3 * i];
    if (circles[3 * i + 1] > max_col) max_col = circles[3 * i + 1];
  }
  const int64_t n_rows = max_row + 2 * pad, n_cols = max_col + 2 * pad;
  uint8_t* claimed = (uint8_t*)calloc((size_t)(n_rows * n_cols), 1);
  for (int i = 0; i < n; ++i) {
    int hit = 0;
    for (int k = 0; k < nr && !hit; ++k) {
      const int64_t rr = pymod(ring[2 * k] + (int64_t)circles[3 * i] + pad, n_rows);
      const int64_t cc = pymod(ring[2 * k + 1] + (int64_t)circles[3 * i + 1] + pad, n_cols);
      hit = claimed[rr * n_cols + cc];
    }
    keep[i] = !hit;
    if (!hit)
      for (int k = 0; k < nr; ++k) {
        const int64_t rr = pymod(ring[2 * k] + (int64_t)circles[3 * i] + pad, n_rows);
        const int64_t cc = pymod(ring[2 * k + 1] + (int64_t)circles[3 * i + 1] + pad, n_cols);
        claimed[rr * n_cols + cc] = 1;
      }
  }
  free(claimed);
  free(ring);
}

/* mean_grad of one circle: float64 sequential sum in perimeter order, zero-padded borders. */
static inline double sum_one(const float* angle, const uint8_t* edges, int h, int w, int row, int col,
                             const int32_t* per, const double* expected, int nper) {
  double acc = 0.0;
  for (int j = 0; j < nper; ++j) {
    const int y = row + per[2 * j], x = col + per[2 * j + 1];
    if (y < 0 || y >= h || x < 0 || x >= w) continue;
    const int64_t i = (int64_t)y * w + x;
    if (!edges[i]) continue;
    double d = fabs((double)angle[i] - expected[j]);
    if (d > M_PI) d -= M_PI;
    acc += 4.0 * fabs(d - M_PI / 2) / M_PI - 1.0;
  }
  return acc;
}

/* mean_grad (utils.py:225-251) for n centres of one radius; sums stored as float32. */
void ref_mean_grad(const float* angle, const uint8_t* edges, int h, int w, const int32_t* centers, int n, int r,
                   float* sums) {
  int32_t* per = (int32_t*)malloc((size_t)(8 * (r + 1) + 4) * 2 * sizeof(int32_t));
  const int nper = ref_circle_points(r, 0, per);
  double* expected = (double*)malloc((size_t)nper * sizeof(double));
  for (int j = 0; j < nper; ++j) expected[j] = atan2((double)per[2 * j], (double)per[2 * j + 1]);
  for (int i = 0; i < n; ++i)
    sums[i] = (float)sum_one(angle, edges, h, w, centers[2 * i], centers[2 * i + 1], per, expected, nper);
  free(per);
  free(expected);
}

/* candidate_circles (utils.py:295-344) with explicit picks in the reference's indexing: i0 into the
 * row-major edge list, j1 / j2 into p0's cell list. */
void ref_candidate_circles_from_picks(const uint8_t* edges, int h, int w, int grid, const int64_t* i0,
                                      const int64_t* j1, const int64_t* j2, int64_t k, float* out) {
  const int gr = (h + grid - 1) / grid, gc = (w + grid - 1) / grid;
  int64_t* starts = (int64_t*)malloc((size_t)gr * gc * sizeof(int64_t));
  int64_t* counts = (int64_t*)malloc((size_t)gr * gc * sizeof(int64_t));
  const int64_t e = ref_grid_array(edges, h, w, grid, NULL, starts, counts);
  int32_t* gco = (int32_t*)malloc((size_t)(e > 0 ? e : 1) * 2 * sizeof(int32_t));
  int32_t* rco = (int32_t*)malloc((size_t)(e > 0 ? e : 1) * 2 * sizeof(int32_t));
  ref_grid_array(edges, h, w, grid, gco, starts, counts);
  int64_t m = 0;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x)
      if (edges[(int64_t)y * w + x]) rco[2 * m] = y, rco[2 * m + 1] = x, ++m;
  for (int64_t i = 0; i < k; ++i) {
    const int32_t* p0 = rco + 2 * i0[i];
    const int64_t cell = (int64_t)(p0[0] / grid) * gc + p0[1] / grid;
    ref_circumcircle(p0, gco + 2 * (starts[cell] + j1[i]), gco + 2 * (starts[cell] + j2[i]), out + 3 * i);
  }
  free(starts);
  free(counts);
  free(gco);
  free(rco);
}

/* find_circles (utils.py:102-222) after to_uint8.  Returns the number of circles written
 * (row, col, r) + scores, in suppression order; -1 if cap is too small. */
int64_t ref_find_circles(const uint8_t* img, int h, int w, double low_q, double high_q, int grid, int64_t num_iter,
                         int min_r, int max_r, float min_roundness, int min_dist, uint64_t seed,
                         int32_t* out_circles, float* out_scores, int64_t cap) {
  const int64_t n = (int64_t)h * w;
  if (n <= 0) return 0;
  uint8_t* blur = (uint8_t*)malloc((size_t)n);
  uint8_t* edges = (uint8_t*)malloc((size_t)n);
  int16_t* dx = (int16_t*)malloc((size_t)n * 2);
  int16_t* dy = (int16_t*)malloc((size_t)n * 2);
  float lohi[2];
  ref_edge_stage(img, h, w, low_q, high_q, blur, dx, dy, edges, lohi);
  free(blur);

  const int gr = (h + grid - 1) / grid, gc = (w + grid - 1) / grid;
  int64_t* starts = (int64_t*)malloc((size_t)gr * gc * sizeof(int64_t));
  int64_t* counts = (int64_t*)malloc((size_t)gr * gc * sizeof(int64_t));
  const int64_t n_edges = ref_grid_array(edges, h, w, grid, NULL, starts, counts);
  int64_t result = 0;
  if (n_edges == 0 || num_iter <= 0) goto done_early;
  {
    int32_t* coords = (int32_t*)malloc((size_t)n_edges * 2 * sizeof(int32_t));
    ref_grid_array(edges, h, w, grid, coords, starts, counts);

    /* candidates -> radius filter -> round half-to-even -> on-image filter -> packed keys */
    uint64_t* keys = (uint64_t*)malloc((size_t)num_iter * sizeof(uint64_t));
    int64_t nk = 0;
    const int OFF = 1024; /* keeps (row, col) non-negative inside the key */
    for (int64_t it = 0; it < num_iter; ++it) {
      const uint64_t a = ((uint64_t)it * (uint64_t)n_edges) / (uint64_t)num_iter;
      const uint64_t b = (((uint64_t)it + 1) * (uint64_t)n_edges) / (uint64_t)num_iter;
      const uint64_t width = b - a > 1 ? b - a : 1;
      const uint64_t u0 = a + (((uint64_t)ref_draw_uniform32(seed, it, 0) * width) >> 32);
      const int32_t* p0 = coords + 2 * u0;
      const int64_t cell = (int64_t)(p0[0] / grid) * gc + p0[1] / grid;
      const uint64_t cnt = (uint64_t)counts[cell];
      const uint64_t j1 = ((uint64_t)ref_draw_uniform32(seed, it, 1) * cnt) >> 32;
      const uint64_t j2 = ((uint64_t)ref_draw_uniform32(seed, it, 2) * cnt) >> 32;
      float c[3];
      ref_circumcircle(p0, coords + 2 * (starts[cell] + j1), coords + 2 * (starts[cell] + j2), c);
      if (!(c[2] >= (float)min_r && c[2] <= (float)max_r)) continue;
      const int32_t row = (int32_t)rintf(c[0]), col = (int32_t)rintf(c[1]), r = (int32_t)rintf(c[2]);
      if (!(row + r >= 0 && col + r >= 0 && row - r < h && col - r < w)) continue;
      keys[nk++] = ((uint64_t)r << 48) | ((uint64_t)(row + OFF) << 24) | (uint64_t)(col + OFF);
    }
    free(coords);
    /* np.unique + order by (r, row, col) */
    qsort(keys, (size_t)nk, sizeof(uint64_t), cmp_u64);
    int64_t nu = 0;
    for (int64_t i = 0; i < nk; ++i)
      if (i == 0 || keys[i] != keys[i - 1]) keys[nu++] = keys[i];

    /* gradient angle, float32 arctan2(dy, dx) */
    float* angle = (float*)malloc((size_t)n * sizeof(float));
    for (int64_t i = 0; i < n; ++i)
      angle[i] = edges[i] ? (float)atan2((double)dy[i], (double)dx[i]) : 0.0f; /* read on edge pixels only */

    Scored* sc = (Scored*)malloc((size_t)(nu > 0 ? nu : 1) * sizeof(Scored));
    int64_t ng = 0;
    int32_t* per = (int32_t*)malloc((size_t)(8 * (max_r + 1) + 4) * 2 * sizeof(int32_t));
    double* expected = (double*)malloc((size_t)(8 * (max_r + 1) + 4) * sizeof(double));
    int cur_r = -1, nper = 0;
    for (int64_t i = 0; i < nu; ++i) {
      const int r = (int)(keys[i] >> 48);
      const int row = (int)((keys[i] >> 24) & 0xFFFFFF) - OFF, col = (int)(keys[i] & 0xFFFFFF) - OFF;
      if (r != cur_r) {
        nper = ref_circle_points(r, 0, per);
        for (int j = 0; j < nper; ++j) expected[j] = atan2((double)per[2 * j], (double)per[2 * j + 1]);
        cur_r = r;
      }
      const float s = (float)sum_one(angle, edges, h, w, row, col, per, expected, nper) / (float)nper;
      if (s >= min_roundness) {
        sc[ng].row = row, sc[ng].col = col, sc[ng].r = r, sc[ng].score = s;
        ++ng;
      }
    }
    free(per);
    free(expected);
    free(angle);
    free(keys);
    g_max_r_for_sort = max_r;
    qsort(sc, (size_t)ng, sizeof(Scored), cmp_canonical);
    int32_t* flat = (int32_t*)malloc((size_t)(ng > 0 ? ng : 1) * 3 * sizeof(int32_t));
    uint8_t* keep = (uint8_t*)malloc((size_t)(ng > 0 ? ng : 1));
    for (int64_t i = 0; i < ng; ++i) flat[3 * i] = sc[i].row, flat[3 * i + 1] = sc[i].col, flat[3 * i + 2] = sc[i].r;
    if (min_dist > 0) ref_filter_neighbors(flat, (int)ng, min_dist, keep);
    else memset(keep, 1, (size_t)ng);
    for (int64_t i = 0; i < ng; ++i)
      if (keep[i]) {
        if (result >= cap) {
          result = -1;
          break;
        }
        memcpy(out_circles + 3 * result, flat + 3 * i, 3 * sizeof(int32_t));
        out_scores[result] = sc[i].score;
        ++result;
      }
    free(flat);
    free(keep);
    free(sc);
  }
done_early:
  free(starts);
  free(counts);
  free(edges);
  free(dx);
  free(dy);
  return result;
}

/* ------------------------------------------------------------------------------------------
 * A12 BeadFinder.__call__ (find.py:471-605) + A18 ROI reduction, one assay of one time slice.
 * image (C, H, W) uint16.  Detection on the listed search channels (seed + k), cross-channel
 * de-duplication at 2 * min_r, label map, L x L windows, fg/bg masks, masked sums and counts.
 * roi / fg / bg may be NULL (then a scratch window is gathered and reduced all the same).
 * Returns the number of beads (<= cap) or -1.
 * ---------------------------------------------------------------------------------------- */
int64_t ref_bead_assay(const uint16_t* image, int n_c, int h, int w, int min_r, int max_r, int roi_len, double low_q,
                       double high_q, int64_t num_iter, float min_roundness, const int32_t* search_channels,
                       int n_search, uint64_t seed, int32_t* beads, int64_t cap, uint16_t* roi, uint8_t* fg,
                       uint8_t* bg, int64_t* fg_sum, int64_t* bg_sum, int64_t* fg_cnt, int64_t* bg_cnt) {
  const int64_t n = (int64_t)h * w;
  int64_t m = 0;
  uint8_t* u8 = (uint8_t*)malloc((size_t)n);
  int32_t* found = (int32_t*)malloc((size_t)cap * 3 * sizeof(int32_t));
  float* scores = (float*)malloc((size_t)cap * sizeof(float));
  for (int k = 0; k < n_search; ++k) {
    ref_to_uint8_u16(image + (int64_t)search_channels[k] * n, n, u8);
    const int64_t nf = ref_find_circles(u8, h, w, low_q, high_q, 20, num_iter, min_r, max_r, min_roundness, min_r,
                                        seed + (uint64_t)k, found, scores, cap);
    if (nf < 0) {
      m = -1;
      break;
    }
    const int64_t seen = m;
    const double rad2 = (double)(2 * min_r) * (double)(2 * min_r);
    for (int64_t i = 0; i < nf; ++i) {
      int dup = 0;
      for (int64_t j = 0; j < seen && !dup; ++j) {
        const double dr = (double)found[3 * i] - beads[3 * j], dc = (double)found[3 * i + 1] - beads[3 * j + 1];
        dup = dr * dr + dc * dc <= rad2;
      }
      if (dup) continue;
      if (m >= cap) {
        m = -1;
        break;
      }
      memcpy(beads + 3 * m, found + 3 * i, 3 * sizeof(int32_t));
      ++m;
    }
    if (m < 0) break;
  }
  free(u8);
  free(found);
  free(scores);
  if (m <= 0) return m;

  int32_t* labels = (int32_t*)malloc((size_t)n * sizeof(int32_t));
  ref_circle_labels(beads, (int)m, h, w, labels);
  const int64_t win = (int64_t)roi_len * roi_len;
  uint16_t* scratch = roi ? NULL : (uint16_t*)malloc((size_t)n_c * win * sizeof(uint16_t));
  uint8_t* mscratch = (fg && bg) ? NULL : (uint8_t*)malloc((size_t)2 * win);
  for (int64_t i = 0; i < m; ++i) {
    int tblr[4];
    ref_bounding_box(beads[3 * i + 1], beads[3 * i], roi_len, w, h, tblr);
    uint16_t* r_out = roi ? roi + i * n_c * win : scratch;
    uint8_t* f_out = (fg && bg) ? fg + i * win : mscratch;
    uint8_t* b_out = (fg && bg) ? bg + i * win : mscratch + win;
    int64_t fc = 0, bc = 0;
    for (int y = 0; y < roi_len; ++y)
      for (int x = 0; x < roi_len; ++x) {
        const int32_t l = labels[(int64_t)(tblr[0] + y) * w + tblr[2] + x];
        const uint8_t f = l == (int32_t)i, b = l == -1;
        f_out[(int64_t)y * roi_len + x] = f;
        b_out[(int64_t)y * roi_len + x] = b;
        fc += f;
        bc += b;
      }
    fg_cnt[i] = fc;
    bg_cnt[i] = bc;
    for (int c = 0; c < n_c; ++c) {
      int64_t fs = 0, bs = 0;
      for (int y = 0; y < roi_len; ++y) {
        const uint16_t* src = image + (int64_t)c * n + (int64_t)(tblr[0] + y) * w + tblr[2];
        uint16_t* dst = r_out + (int64_t)c * win + (int64_t)y * roi_len;
        memcpy(dst, src, (size_t)roi_len * sizeof(uint16_t));
        for (int x = 0; x < roi_len; ++x) {
          if (f_out[(int64_t)y * roi_len + x]) fs += dst[x];
          if (b_out[(int64_t)y * roi_len + x]) bs += dst[x];
        }
      }
      fg_sum[i * n_c + c] = fs;
      bg_sum[i * n_c + c] = bs;
    }
  }
  free(scratch);
  free(mscratch);
  free(labels);
  return m;
}

/* ------------------------------------------------------------------------------------------
 * Whole stack, mode P (every time slice is its own assay: pipeline.py:18-24): flat-field with the
 * slice's own maxima, bead assay, reductions.  One assay per OpenMP thread.
 * stack (T, C, H, W) uint16.  Per assay outputs: n_beads[t], checksum[t] = sum of fg sums.
 * Returns the total number of beads, or -1.
 * ---------------------------------------------------------------------------------------- */
int64_t ref_run_stack(const uint16_t* stack, int n_t, int n_c, int h, int w, const float* flat_img, double flat_scalar,
                      double dark, int min_r, int max_r, int roi_len, double low_q, double high_q, int64_t num_iter,
                      float min_roundness, const uint64_t* seeds, int64_t cap, int n_threads, int64_t* n_beads,
                      int64_t* checksum) {
  const int64_t plane = (int64_t)h * w, assay = plane * n_c;
  int64_t total = 0;
  int failed = 0;
#ifdef _OPENMP
  if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
#pragma omp parallel for schedule(dynamic, 1) reduction(+ : total)
  for (int t = 0; t < n_t; ++t) {
    uint16_t* img = (uint16_t*)malloc((size_t)assay * sizeof(uint16_t));
    int32_t* beads = (int32_t*)malloc((size_t)cap * 3 * sizeof(int32_t));
    int64_t* sums = (int64_t*)malloc((size_t)cap * (2 * n_c + 2) * sizeof(int64_t));
    ref_flatfield_correct_u16(stack + t * assay, assay, plane, flat_img, flat_scalar, dark, img);
    const int32_t ch0 = 0;
    const int64_t m = ref_bead_assay(img, n_c, h, w, min_r, max_r, roi_len, low_q, high_q, num_iter, min_roundness,
                                     &ch0, 1, seeds[t], beads, cap, NULL, NULL, NULL, sums, sums + cap * n_c,
                                     sums + 2 * cap * n_c, sums + 2 * cap * n_c + cap);
    if (m < 0) {
#pragma omp atomic write
      failed = 1;
    } else {
      int64_t cs = 0;
      for (int64_t i = 0; i < m * n_c; ++i) cs += sums[i];
      n_beads[t] = m;
      checksum[t] = cs;
      total += m;
    }
    free(img);
    free(beads);
    free(sums);
  }
  return failed ? -1 : total;
}

int ref_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
