// Does the power-of-two distance between the planes a workgroup streams together cost bandwidth?  The correction
// pass's access pattern (a workgroup copies the same 32 rows x 2048 uint16 of PB planes, four 16-byte loads then four
// stores per lane and row) with the planes 32 MiB apart (4096^2 uint16, as the stack lies in memory) and with padded
// plane strides, on the source side, the destination side or both; one plane per workgroup beside it.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/plane_copy_bw.hip -o tools/micro/plane_copy_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int PB>
__global__ __launch_bounds__(256) void k_planes(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t src_stride,
                                                int64_t dst_stride, int h, int wvec, int rows) {
  const int x = blockIdx.x * 256 + threadIdx.x;
  if (x >= wvec) return;
  const int plane0 = blockIdx.z * PB;
  const int r0 = blockIdx.y * rows, r1 = min(h, r0 + rows);
  for (int r = r0; r < r1; ++r) {
    u32x4 v[PB];
#pragma unroll
    for (int b = 0; b < PB; ++b) v[b] = src[(int64_t)(plane0 + b) * src_stride + (int64_t)r * wvec + x];
#pragma unroll
    for (int b = 0; b < PB; ++b) dst[(int64_t)(plane0 + b) * dst_stride + (int64_t)r * wvec + x] = v[b];
  }
}

template <int PB>
void run(const char* what, const u32x4* s, u32x4* d, int64_t ss, int64_t ds, int planes, int h, int wvec, int rows) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  dim3 grid((wvec + 255) / 256, (h + rows - 1) / rows, planes / PB);
  hipLaunchKernelGGL((k_planes<PB>), grid, dim3(256), 0, 0, s, d, ss, ds, h, wvec, rows);
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int r = 0; r < 5; ++r) {
    hipEventRecord(a, 0);
    hipLaunchKernelGGL((k_planes<PB>), grid, dim3(256), 0, 0, s, d, ss, ds, h, wvec, rows);
    hipEventRecord(b, 0);
    hipEventSynchronize(b);
    float ms;
    (void)hipEventElapsedTime(&ms, a, b);
    best = std::min(best, ms);
  }
  const double bytes = 2.0 * planes * h * wvec * 16;
  printf("%-40s PB %d rows %3d  %7.3f ms  %7.1f GB/s\n", what, PB, rows, best, bytes / best * 1e-6);
  fflush(stdout);
}

int main() {
  const int planes = 128, h = 4096, wvec = 512;  // 4096 uint16 = 512 x 16 B per row; 32 MiB per plane
  const int64_t plane_vec = (int64_t)h * wvec;
  const int64_t pads[] = {0, 16, 512, 528, 4096 + 16, 65536 + 528};  // in 16-byte units: 0, 256 B, 8 KiB, 8.25 KiB, 64.25 KiB, 1 MiB + 8.25 KiB
  const int64_t max_stride = plane_vec + 65536 + 528;
  u32x4 *s, *d;
  if (hipMalloc(&s, max_stride * planes * 16) != hipSuccess || hipMalloc(&d, max_stride * planes * 16 + (1 << 20)) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  hipMemset(s, 1, max_stride * planes * 16);
  hipMemset(d, 2, max_stride * planes * 16);
  hipDeviceSynchronize();
  for (int rows : {32, 8}) {
    for (int64_t ps : pads)
      for (int64_t pd : {(int64_t)0, ps}) {
        char what[96];
        snprintf(what, sizeof what, "src pad %6lld B, dst pad %6lld B", (long long)ps * 16, (long long)pd * 16);
        run<4>(what, s, d, plane_vec + ps, plane_vec + pd, planes, h, wvec, rows);
        if (ps == 0) break;
      }
    run<1>("one plane per workgroup, no pad", s, d, plane_vec, plane_vec, planes, h, wvec, rows);
    run<2>("two planes per workgroup, no pad", s, d, plane_vec, plane_vec, planes, h, wvec, rows);
    run<1>("one plane per workgroup, dst pad 8.25 KiB", s, d, plane_vec, plane_vec + 528, planes, h, wvec, rows);
    // destination base shifted against the source (src and dst 2^k apart otherwise)
    run<4>("dst base + 8.25 KiB, no pad", s, d + 528, plane_vec, plane_vec, planes, h, wvec, rows);
    run<1>("dst base + 8.25 KiB, one plane, no pad", s, d + 528, plane_vec, plane_vec, planes, h, wvec, rows);
  }
  return 0;
}
