// A16/A17: grid fit and per-chamber masks of ButtonFinder (reference: src/magnify/find.py).
//   - cluster_1d (find.py:632-677): cost of every integer offset, one wave per offset;
//   - fg / bg masks of find_rois (find.py:383-400): cv.circle filled disk and annulus rasters;
//   - masked sums with explicit per-marker masks (README.md:21-22 on chip outputs).
#include <math.h>

#include "mg_common.h"

namespace {

constexpr int NT = 256;

__device__ __forceinline__ int lower_bound(const double* __restrict__ a, int n, double v) {
  int lo = 0, hi = n;  // np.searchsorted(a, v, side="left")
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (a[mid] < v) lo = mid + 1;
    else hi = mid;
  }
  return lo;
}

// cost(offset) = sum_k [ var_k * sqrt(ideal_k) + penalty * (ideal_k - n_k)^2 ], var_k = mean squared
// distance to the cluster centre, empty clusters take the maximum var (find.py:647-668).
// One wave per offset, one lane per cluster (64 at a time): the two binary searches and the (sequential, as in the
// reference) sum over a cluster's points run side by side for all clusters; the total is added up by one lane in
// cluster order.  (A thread per offset walked all clusters twice, ~2700 dependent loads: 225 us for the ~400
// offsets of one axis on two workgroups.)
__global__ __launch_bounds__(64) void k_cluster1d(const double* __restrict__ pts, int n_pts, int n_offsets,
                                                  int n_clusters, double cluster_length,
                                                  const double* __restrict__ ideal, double penalty,
                                                  double* __restrict__ costs) {
  extern __shared__ double s_var[];              // [n_clusters] variance of the non-empty clusters
  int* s_cnt = reinterpret_cast<int*>(s_var + n_clusters);  // [n_clusters] points per cluster
  const int off = blockIdx.x, lane = threadIdx.x;
  double vmax = 0.0;  // costs of non-empty clusters are >= 0 and empty ones enter the max as 0
  for (int k = lane; k < n_clusters; k += 64) {
    const double b0 = (double)k * cluster_length + (double)off, b1 = (double)(k + 1) * cluster_length + (double)off;
    const int lo = lower_bound(pts, n_pts, b0), hi = lower_bound(pts, n_pts, b1);
    double var = 0.0;
    if (hi > lo) {
      const double c = (b1 + b0) / 2.0;
      double s = 0.0;
      for (int i = lo; i < hi; ++i) s += (pts[i] - c) * (pts[i] - c);
      var = s / (double)(hi - lo);
      vmax = fmax(vmax, var);
    }
    s_var[k] = var;
    s_cnt[k] = hi - lo;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) vmax = fmax(vmax, __shfl_xor(vmax, o));
  __syncthreads();
  if (lane == 0) {
    double total = 0.0;
    for (int k = 0; k < n_clusters; ++k) {
      const double var = s_cnt[k] > 0 ? s_var[k] : vmax;
      const double miss = ideal[k] - (double)s_cnt[k];
      total += var * sqrt(ideal[k]) + penalty * miss * miss;
    }
    costs[off] = total;
  }
}

// fg = filled cv.circle of radius r_i, bg = annulus(outer_r, inner_r), all centred on
// (cy_i, cx_i) in window coordinates; half-width tables from mg_cv_disk_halfwidths.
__global__ __launch_bounds__(NT) void k_button_masks(const int32_t* __restrict__ d_centers,
                                                     const int32_t* __restrict__ d_radii, int len, int outer_r,
                                                     int inner_r, const int32_t* __restrict__ d_hw, int hw_stride,
                                                     uint8_t* __restrict__ d_fg, uint8_t* __restrict__ d_bg) {
  const int g = blockIdx.x;
  const int cy = d_centers[2 * g], cx = d_centers[2 * g + 1], r = d_radii[g];
  const int32_t* hw_fg = d_hw + (int64_t)r * hw_stride;
  const int32_t* hw_out = d_hw + (int64_t)outer_r * hw_stride;
  const int32_t* hw_in = d_hw + (int64_t)inner_r * hw_stride;
  const int n = len * len;
  for (int p = threadIdx.x; p < n; p += NT) {
    const int y = p / len, x = p - y * len;
    const int ady = abs(y - cy), adx = abs(x - cx);
    const bool f = ady <= r && adx <= hw_fg[ady];
    const bool o = ady <= outer_r && adx <= hw_out[ady];
    const bool i = ady <= inner_r && adx <= hw_in[ady];
    d_fg[(int64_t)g * n + p] = f;
    d_bg[(int64_t)g * n + p] = o && !i;
  }
}

template <typename T, typename ACC>
__global__ __launch_bounds__(NT) void k_masked_sums(const T* __restrict__ d_roi, const uint8_t* __restrict__ d_fg,
                                                    const uint8_t* __restrict__ d_bg, int n_ct, int n,
                                                    double* __restrict__ d_sums, int32_t* __restrict__ d_counts) {
  __shared__ ACC s_f[NT / 64], s_b[NT / 64];
  __shared__ int c_f[NT / 64], c_b[NT / 64];
  const int g = blockIdx.x, ct = blockIdx.y;
  const T* v = d_roi + ((int64_t)g * n_ct + ct) * n;
  const uint8_t* fg = d_fg + (int64_t)g * n;
  const uint8_t* bg = d_bg + (int64_t)g * n;
  ACC sf = 0, sb = 0;
  int cf = 0, cb = 0;
  for (int p = threadIdx.x; p < n; p += NT) {
    const ACC x = (ACC)v[p];
    if (fg[p]) sf += x, ++cf;
    if (bg[p]) sb += x, ++cb;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sf += __shfl_xor(sf, off);
    sb += __shfl_xor(sb, off);
    cf += __shfl_xor(cf, off);
    cb += __shfl_xor(cb, off);
  }
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) s_f[wave] = sf, s_b[wave] = sb, c_f[wave] = cf, c_b[wave] = cb;
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = d_sums + ((int64_t)g * n_ct + ct) * 2;
    o[0] = (double)(s_f[0] + s_f[1] + s_f[2] + s_f[3]);
    o[1] = (double)(s_b[0] + s_b[1] + s_b[2] + s_b[3]);
    if (ct == 0 && d_counts) {
      d_counts[2 * (int64_t)g] = c_f[0] + c_f[1] + c_f[2] + c_f[3];
      d_counts[2 * (int64_t)g + 1] = c_b[0] + c_b[1] + c_b[2] + c_b[3];
    }
  }
}

}  // namespace

extern "C" int mg_cluster1d_costs(const double* d_sorted_points, int n_points, int n_offsets, int n_clusters,
                                  double cluster_length, const double* d_ideal, double penalty, double* d_costs,
                                  void* stream) {
  if (!d_sorted_points || !d_ideal || !d_costs || n_points < 0 || n_offsets < 0 || n_clusters <= 0) return MG_EINVAL;
  if (n_offsets == 0) return MG_OK;
  if (n_clusters > 4096) return MG_EINVAL;  // (the per-cluster terms of an offset are staged in LDS)
  hipLaunchKernelGGL(k_cluster1d, dim3(n_offsets), dim3(64), (size_t)n_clusters * 12 + 8, mg_stream(stream),
                     d_sorted_points, n_points, n_offsets, n_clusters, cluster_length, d_ideal, penalty, d_costs);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_button_masks(const int32_t* d_centers, const int32_t* d_radii, int m, int roi_len, int outer_r,
                               int inner_r, const int32_t* d_cv_halfwidths, int hw_stride, int max_table_r,
                               uint8_t* d_fg, uint8_t* d_bg, void* stream) {
  if (!d_centers || !d_radii || !d_cv_halfwidths || !d_fg || !d_bg || m < 0 || roi_len <= 0) return MG_EINVAL;
  if (outer_r < 0 || inner_r < 0 || outer_r > max_table_r || inner_r > max_table_r || hw_stride <= max_table_r)
    return MG_EINVAL;
  if (m == 0) return MG_OK;
  hipLaunchKernelGGL(k_button_masks, dim3(m), dim3(NT), 0, mg_stream(stream), d_centers, d_radii, roi_len, outer_r,
                     inner_r, d_cv_halfwidths, hw_stride, d_fg, d_bg);
  MG_CHECK_LAUNCH();
  return MG_OK;
}

extern "C" int mg_masked_sums(const void* d_roi, int dtype, const uint8_t* d_fg, const uint8_t* d_bg, int m, int n_ct,
                              int roi_len, double* d_sums, int32_t* d_counts, void* stream) {
  if (!d_roi || !d_fg || !d_bg || !d_sums || m < 0 || n_ct <= 0 || n_ct > 65535 || roi_len <= 0) return MG_EINVAL;
  if (m == 0) return MG_OK;
  const dim3 g(m, n_ct);
  hipStream_t s = mg_stream(stream);
  const int n = roi_len * roi_len;
  switch (dtype) {
    case MG_U8:
      hipLaunchKernelGGL((k_masked_sums<uint8_t, long long>), g, dim3(NT), 0, s, (const uint8_t*)d_roi, d_fg, d_bg, n_ct, n, d_sums, d_counts);
      break;
    case MG_U16:
      hipLaunchKernelGGL((k_masked_sums<uint16_t, long long>), g, dim3(NT), 0, s, (const uint16_t*)d_roi, d_fg, d_bg, n_ct, n, d_sums, d_counts);
      break;
    case MG_F32:
      hipLaunchKernelGGL((k_masked_sums<float, double>), g, dim3(NT), 0, s, (const float*)d_roi, d_fg, d_bg, n_ct, n, d_sums, d_counts);
      break;
    case MG_F64:
      hipLaunchKernelGGL((k_masked_sums<double, double>), g, dim3(NT), 0, s, (const double*)d_roi, d_fg, d_bg, n_ct, n, d_sums, d_counts);
      break;
    default:
      return MG_EINVAL;
  }
  MG_CHECK_LAUNCH();
  return MG_OK;
}
