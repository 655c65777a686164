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
