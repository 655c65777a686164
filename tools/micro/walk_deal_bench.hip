// Would the prefilter's perimeter walk gain from DEALING the circles of a radius to the waves by the LDS bank class of
// their centre?  A wave takes a pool of 256 circles (random centres in the 128 x 256 super-tile) and walks them as four
// sub-chunks of 64: (a) as they come, (b) pool sorted by bank class ((address >> 2) & 63) and dealt with stride 4 -- lane
// l of sub-chunk j takes sorted[j + 4 l], so a class with up to four members gives every sub-chunk at most one --,
// (c) the same with 32 classes ((address >> 2) & 31).  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I magnify_amd/csrc ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <vector>
#include "../../magnify_amd/csrc/mg_score.hip"
extern "C" int mg_dedup_layout(int, int, int, int, int*, int*, int64_t*, int64_t*) { return -1; }  // (host entry point not used here)

template <int R>
__global__ __launch_bounds__(512) void kb(const int* __restrict__ centres, const uint2* __restrict__ tabs, int n_iter, int* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  for (int i = threadIdx.x; i < WBASE + 180 * WSTR; i += 512) lds[i] = (uint8_t)((i * 7) % 13 > 8 ? (i & 7) : 0x0C);
  __syncthreads();
  int c[4];
  for (int q = 0; q < 4; ++q) c[q] = centres[(blockIdx.x * 512 + threadIdx.x) * 4 + q];
  int acc = 0;
  for (int it = 0; it < n_iter; ++it) {
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int wrow = 26 + ((c[q] >> 8) + 32 * (it & 3)) % 128, wcol = 26 + (c[q] & 255);  // (+32 rows: every class moves by 32, the pattern stays)
      acc += score_r<R>(lds, WBASE + wrow * WSTR + wcol - BIAS, tabs);
    }
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
  const int blocks = 256 * 8, n_iter = 12;
  const int n = blocks * 512;  // lanes
  std::vector<int> h(4 * n);
  int *d_c, *d_out;
  uint2* d_tabs;
  hipMalloc(&d_c, 4 * n * 4);
  hipMalloc(&d_out, n * 4);
  hipMalloc(&d_tabs, 27 * 80 * 8);
  hipMemset(d_tabs, 0x11, 27 * 80 * 8);
  const char* names[3] = {"pool of 256 as it comes", "pool dealt by 64 bank classes, stride 4", "pool dealt by 32 bank classes, stride 4"};
  for (int pat = 0; pat < 3; ++pat) {
    srand(1);
    for (int wv = 0; wv < n / 64; ++wv) {
      std::vector<int> pool(256);
      for (auto& c : pool) c = ((rand() % 128) << 8) | (rand() % 256);
      if (pat > 0) {
        const int mask = pat == 1 ? 63 : 31;
        std::stable_sort(pool.begin(), pool.end(), [&](int a, int b) {
          auto cls = [&](int c) { return (((26 + (c >> 8)) * WSTR + 26 + (c & 255)) >> 2) & mask; };
          return cls(a) < cls(b);
        });
      }
      for (int l = 0; l < 64; ++l)
        for (int q = 0; q < 4; ++q) h[((size_t)wv * 64 + l) * 4 + q] = pat == 0 ? pool[64 * q + l] : pool[q + 4 * l];
    }
    hipMemcpy(d_c, h.data(), (size_t)4 * n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kb<14>, dim3(blocks), dim3(512), WBASE + 180 * WSTR, 0, d_c, d_tabs, n_iter, d_out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double wave_reads = (double)blocks * 8 * n_iter * 4 * 80;  // r = 14: 80 points
    printf("%-48s %.3f ms, %.2f ns per wave-read per CU (%.1f cycles at 2.4 GHz)\n", names[pat], ms, ms * 1e6 / (wave_reads / 256),
           ms * 1e6 / (wave_reads / 256) * 2.4);
  }
  return 0;
}
