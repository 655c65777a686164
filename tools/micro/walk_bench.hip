// Microbenchmark of the prefilter's perimeter walk (score_r<R> of mg_score.hip) with controlled lane addresses:
// uniform / conflict-free / random centres.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I magnify_amd/csrc ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#define MG_WALK_BENCH 1
#include "../../magnify_amd/csrc/mg_score.hip"
extern "C" int mg_dedup_layout(int, int, int, int, int*, int*, int64_t*, int64_t*) { return -1; }  // (host entry point not used here)

template <int R>
__global__ __launch_bounds__(512) void kb(const int* __restrict__ centres, const uint2* __restrict__ tabs, int n_iter, int* out) {
  extern __shared__ __attribute__((aligned(16))) uint8_t lds[];
  for (int i = threadIdx.x; i < 8192 + 180 * 180; i += 512) lds[i] = (uint8_t)((i * 7) % 13 > 8 ? (i & 7) : 0x0C);
  __syncthreads();
  int c = centres[blockIdx.x * 512 + threadIdx.x];
  int acc = 0;
  for (int it = 0; it < n_iter; ++it) {
    const int wrow = 26 + ((c >> 8) + 32 * (it & 3)) % 128, wcol = 26 + (c & 127);  // (+32 rows: same bank pattern, new address)
    const int vaddr = WBASE + wrow * WSTR + wcol - BIAS;
    acc += score_r<R>(lds, vaddr, tabs);
    c += (acc & 1) * 0;  // keep the address, dependency via acc only
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

int main() {
  const int blocks = 256 * 8, n_iter = 50;
  const int n = blocks * 512;
  int* h = (int*)malloc(n * 4);
  int *d_c, *d_out;
  uint2* d_tabs;
  hipMalloc(&d_c, n * 4);
  hipMalloc(&d_out, n * 4);
  hipMalloc(&d_tabs, 27 * 80 * 8);
  hipMemset(d_tabs, 0x11, 27 * 80 * 8);
  const char* names[4] = {"uniform centre", "lanes 4 columns apart (conflict-free)", "random centres", "bank-sorted random (13 row + col/4 distinct per half-wave)"};
  for (int pat = 0; pat < 4; ++pat) {
    for (int i = 0; i < n; ++i) {
      const int l = i & 63;
      if (pat == 0) h[i] = (40 << 8) | 40;
      else if (pat == 1) h[i] = ((40 + (l >> 5)) << 8) | ((l & 31) * 4);
      else if (pat == 2) h[i] = ((rand() % 128) << 8) | (rand() % 128);
      else {  // choose row at random, then the column so that the bank class equals lane & 31
        const int row = rand() % 128, want = l & 31;
        int col4 = ((want - 13 * (26 + row)) % 32 + 32) % 32;  // (13 wrow + wcol / 4) & 31 == want, wcol / 4 = col4 (+ 32 k)
        int wcol = col4 * 4 + (rand() & 3);
        while (wcol < 26) wcol += 128;
        if (wcol >= 26 + 128) wcol -= 128;
        h[i] = (row << 8) | ((wcol - 26) & 127);
      }
    }
    hipMemcpy(d_c, h, n * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL(kb<14>, dim3(blocks), dim3(512), 8192 + 180 * 180, 0, d_c, d_tabs, n_iter, d_out);
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      hipEventElapsedTime(&ms, e0, e1);
    }
    const double wave_reads = (double)blocks * 8 * n_iter * 80;  // r = 14: 80 points
    printf("%-60s %.3f ms, %.2f ns per wave-read per CU (%.1f cycles at 2.4 GHz)\n", names[pat], ms, ms * 1e6 / (wave_reads / 256),
           ms * 1e6 / (wave_reads / 256) * 2.4);
  }
  return 0;
}
