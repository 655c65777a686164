// Host-side tables for the digital circles of the reference: the midpoint-circle perimeter in
// its emission order (utils.py:433-465; the order fixes the float64 summation order of
// mean_grad), the filled disk of filled_circle_points (utils.py:398-430) as per-row half widths,
// and OpenCV's filled circle (cv::Circle, drawing.cpp; call site utils.py:38).
#include <math.h>

#include <vector>

#include "mg_common.h"

namespace {

void perimeter(int r, bool four, std::vector<int32_t>& out) {
  auto put = [&](int a, int b) {
    out.push_back(a);
    out.push_back(b);
  };
  put(0, -r);
  put(-r, 0);
  put(0, r);
  put(r, 0);
  int x = 1, y = -r;
  while (x < -y) {
    put(x, y);
    put(y, x);
    put(-x, y);
    put(-y, x);
    put(x, -y);
    put(y, -x);
    put(-x, -y);
    put(-y, -x);
    if (x * x + y * y - r * r <= 0) {
      ++x;
    } else {
      ++y;
      if (!four) ++x;
    }
  }
  if (y == -x) {
    put(x, y);
    put(-x, -y);
    put(-x, y);
    put(x, -y);
  }
}

}  // namespace

extern "C" int mg_circle_points(int r, int four_connected, int32_t* out_rc, int cap) {
  if (r < 0) return MG_EINVAL;
  std::vector<int32_t> p;
  perimeter(r, four_connected != 0, p);
  const int n = (int)p.size() / 2;
  if (out_rc)
    for (int i = 0; i < n && i < cap; ++i) {
      out_rc[2 * i] = p[2 * i];
      out_rc[2 * i + 1] = p[2 * i + 1];
    }
  return n;
}

extern "C" int mg_disk_halfwidths(int r, int32_t* out) {
  // The row-wise fill of filled_circle_points covers, in each row, the span between the outermost
  // perimeter pixels; rows that hold a single perimeter run keep just that run.
  if (r < 2 || !out) return MG_EINVAL;
  std::vector<int32_t> p;
  perimeter(r, false, p);
  for (int i = 0; i < 2 * r + 1; ++i) out[i] = -1;
  for (size_t i = 0; i < p.size(); i += 2) {
    const int dy = p[i], dx = abs(p[i + 1]);
    if (dx > out[dy + r]) out[dy + r] = dx;
  }
  return 2 * r + 1;
}

extern "C" int mg_cv_disk_halfwidths(int r, int32_t* out) {
  if (r < 0 || !out) return MG_EINVAL;
  for (int i = 0; i <= r; ++i) out[i] = -1;
  int err = 0, dx = r, dy = 0, plus = 1, minus = (r << 1) - 1;
  while (dx >= dy) {
    if (dx > out[dy]) out[dy] = dx;  // rows cy +- dy: span [cx - dx, cx + dx]
    if (dy > out[dx]) out[dx] = dy;  // rows cy +- dx: span [cx - dy, cx + dy]
    ++dy;
    err += plus;
    plus += 2;
    const int mask = (err <= 0) - 1;
    err -= minus & mask;
    dx += mask;
    minus -= mask & 2;
  }
  return r + 1;
}

extern "C" int mg_perimeter_table(int min_r, int max_r, int32_t* out_rc, double* out_expected, int32_t* out_starts,
                                  int cap) {
  if (min_r < 0 || max_r < min_r) return MG_EINVAL;
  int total = 0;
  std::vector<int32_t> all;
  std::vector<int32_t> starts;
  for (int r = min_r; r <= max_r; ++r) {
    starts.push_back(total);
    std::vector<int32_t> p;
    perimeter(r, false, p);
    total += (int)p.size() / 2;
    all.insert(all.end(), p.begin(), p.end());
  }
  starts.push_back(total);
  if (out_starts)
    for (size_t i = 0; i < starts.size(); ++i) out_starts[i] = starts[i];
  if (total > cap) return total;
  for (int i = 0; i < total; ++i) {
    if (out_rc) {
      out_rc[2 * i] = all[2 * i];
      out_rc[2 * i + 1] = all[2 * i + 1];
    }
    if (out_expected) out_expected[i] = atan2((double)all[2 * i], (double)all[2 * i + 1]);  // utils.py:234
  }
  return total;
}

// Prefilter tables of mg_score_circles_keyed.  The perimeter of radius r is walked as PAIRS of opposite points
// (p, -p): both have the radial direction theta = atan2(dr, dc) mod pi, so they share one table.  Pair order
// (the kernel generates the same sequence at compile time, mg_score.hip Pairs<R>): the first points are
// (0, -r), (-r, 0); for every group (x, y) of the midpoint walk (x, y), (y, x), (-x, y), (-y, x); and for the
// diagonal group (x, y), (-x, y).
// Entry [r][k] (MG_SCORE_MAX_R + 1 radii x MG_SCORE_MAX_PAIRS pairs, unused entries zero): 8 signed bytes, byte b
// = an upper bound, in 1/64 rounded up, of the term 4 |d - pi/2| / pi - 1 of mean_grad (utils.py:244-249) for an
// edge pixel whose gradient orientation lies in bin b (mg_canny_nms' eighths of pi): the term is
// 1 - 4 delta / pi with delta the distance (mod pi) between theta and the pixel's orientation, and the bin pins
// the orientation to [b, b + 1] pi/8, so delta >= dist(theta, bin).  +1e-6 per term covers the float32
// rounding of the reference's angle (<= 2 ulp of pi) and a bin assignment made on the exact integer gradient.
// -48 <= q <= 64.
namespace {
void score_pairs(int r, std::vector<int32_t>& out) {  // first point (dr, dc) of every pair
  auto put = [&](int a, int b) {
    out.push_back(a);
    out.push_back(b);
  };
  put(0, -r);
  put(-r, 0);
  int x = 1, y = -r;
  while (x < -y) {
    put(x, y);
    put(y, x);
    put(-x, y);
    put(-y, x);
    if (x * x + y * y - r * r <= 0) {
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
}  // namespace

extern "C" int mg_score_pair_table(uint64_t* out_entries, int cap) {
  const int total = (MG_SCORE_MAX_R + 1) * MG_SCORE_MAX_PAIRS;
  if (!out_entries || cap < total) return total;
  const double PI = 3.141592653589793;
  for (int i = 0; i < total; ++i) out_entries[i] = 0;
  for (int r = 1; r <= MG_SCORE_MAX_R; ++r) {
    std::vector<int32_t> p;
    score_pairs(r, p);
    const int n = (int)p.size() / 2;
    if (n > MG_SCORE_MAX_PAIRS) return MG_EINVAL;
    for (int k = 0; k < n; ++k) {
      double a = fmod(atan2((double)p[2 * k], (double)p[2 * k + 1]), PI);
      if (a < 0) a += PI;
      uint64_t e = 0;
      for (int b = 0; b < 8; ++b) {
        const double lo = b * PI / 8, hi = (b + 1) * PI / 8;
        double delta = 0.0;
        if (a < lo || a > hi) {
          const double d0 = fabs(a - lo), d1 = fabs(a - hi);
          delta = fmin(fmin(d0, PI - d0), fmin(d1, PI - d1));
        }
        int q = (int)ceil(64.0 * (1.0 - 4.0 * delta / PI + 1e-6));
        if (q > 64) q = 64;
        e |= (uint64_t)(uint8_t)(int8_t)q << (8 * b);
      }
      out_entries[(size_t)r * MG_SCORE_MAX_PAIRS + k] = e;
    }
  }
  return total;
}

// number of pairs of radius r and the first point of pair k (tests)
extern "C" int mg_score_pairs(int r, int32_t* out_rc, int cap) {
  if (r < 1) return MG_EINVAL;
  std::vector<int32_t> p;
  score_pairs(r, p);
  const int n = (int)p.size() / 2;
  if (out_rc)
    for (int i = 0; i < n && i < cap; ++i) out_rc[2 * i] = p[2 * i], out_rc[2 * i + 1] = p[2 * i + 1];
  return n;
}
