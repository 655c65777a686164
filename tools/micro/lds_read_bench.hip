// Microbenchmark: LDS read throughput per CU for ds_read_u8 / u16 / b32 with random, strided and uniform addresses.
// build: hipcc -O3 --offload-arch=gfx950 tools/micro/lds_read_bench.hip -o gpurun_out/lds_read_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int MODE>  // 0: u8, 1: u16, 2: b32
__global__ __launch_bounds__(256) void k(const int* __restrict__ offs, int n_iter, int* out) {
  extern __shared__ uint8_t lds[];
  for (int i = threadIdx.x; i < 32768; i += 256) lds[i] = (uint8_t)(i * 7);
  __syncthreads();
  int base = offs[blockIdx.x * 256 + threadIdx.x];
  int acc = 0;
  for (int it = 0; it < n_iter; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      const int a = base + u * 724;  // immediate offsets, like the perimeter walk (181 dwords apart: same bank pattern)
      if (MODE == 0) acc += lds[a & 32767];
      else if (MODE == 1) acc += reinterpret_cast<uint16_t*>(lds)[(a & 32767) >> 1];
      else acc += reinterpret_cast<uint32_t*>(lds)[(a & 32767) >> 2];
    }
    base = (base + acc) & 8191;  // dependency between iterations only
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
  const int blocks = 256 * 8, n_iter = 200;
  int* h = (int*)malloc(blocks * 256 * 4);
  int *d_offs, *d_out;
  hipMalloc(&d_offs, blocks * 256 * 4);
  hipMalloc(&d_out, blocks * 256 * 4);
  const char* names[3] = {"random", "lane*4 (conflict-free dwords)", "uniform"};
  for (int pat = 0; pat < 3; ++pat) {
    for (int i = 0; i < blocks * 256; ++i) h[i] = pat == 0 ? (rand() & 8191) : pat == 1 ? (i & 63) * 4 : 0;
    hipMemcpy(d_offs, h, blocks * 256 * 4, hipMemcpyHostToDevice);
    for (int mode = 0; mode < 3; ++mode) {
      hipEvent_t e0, e1;
      hipEventCreate(&e0);
      hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 32768 + 1024, 0, d_offs, n_iter, d_out);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 32768 + 1024, 0, d_offs, n_iter, d_out);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 32768 + 1024, 0, d_offs, n_iter, d_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
      }
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      const double wave_reads = (double)blocks * 4 * n_iter * 32;
      printf("%-32s %s: %.3f ms, %.2f ns per wave-read per CU (x2.4 = cycles at 2.4 GHz: %.1f)\n", names[pat],
             mode == 0 ? "u8 " : mode == 1 ? "u16" : "b32", ms, ms * 1e6 / (wave_reads / 256), ms * 1e6 / (wave_reads / 256) * 2.4);
    }
  }
  return 0;
}
