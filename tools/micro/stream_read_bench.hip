// How fast does a SHORT streaming-read kernel run?  A max-reduction over a uint16 buffer (what pass 1 of the
// flat-field correction is for one assay) at several buffer sizes, grid sizes, loads in flight per lane and ways of
// handing over the result.  build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/stream_read_bench.hip -o ...
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

template <int INFLIGHT, int FINISH>  // FINISH 0: per-block store, 1: atomicMax u32, 2: f64 CAS loop with look-first
__global__ __launch_bounds__(256) void k_max(const uint4* __restrict__ src, int64_t nvec, uint32_t* out, double* dout) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t m = 0;
  for (; v + (INFLIGHT - 1) * stride < nvec; v += INFLIGHT * stride) {
    uint4 x[INFLIGHT];
#pragma unroll
    for (int q = 0; q < INFLIGHT; ++q) x[q] = src[v + q * stride];
#pragma unroll
    for (int q = 0; q < INFLIGHT; ++q) {
      const uint32_t w[4] = {x[q].x, x[q].y, x[q].z, x[q].w};
#pragma unroll
      for (int j = 0; j < 4; ++j) m = max(m, max(w[j] & 0xFFFFu, w[j] >> 16));
    }
  }
  for (; v < nvec; v += stride) {
    const uint4 x = src[v];
    m = max(m, max(max(x.x & 0xFFFFu, x.x >> 16), max(x.y & 0xFFFFu, x.y >> 16)));
    m = max(m, max(max(x.z & 0xFFFFu, x.z >> 16), max(x.w & 0xFFFFu, x.w >> 16)));
  }
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, off));
  __shared__ uint32_t s[4];
  if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(s[0], s[1]), max(s[2], s[3]));
    if (FINISH == 0) out[blockIdx.x] = m;
    else if (FINISH == 1) atomicMax(out, m);
    else {
      unsigned long long* a = reinterpret_cast<unsigned long long*>(dout);
      unsigned long long old = *a;
      const double val = (double)m;
      while (true) {
        const double cur = __longlong_as_double((long long)old);
        if (!(val > cur)) break;
        const unsigned long long seen = atomicCAS(a, old, (unsigned long long)__double_as_longlong(val));
        if (seen == old) break;
        old = seen;
      }
    }
  }
}

template <int INFLIGHT, int FINISH>
float run(const uint4* src, int64_t nvec, int blocks, uint32_t* out, double* dout, int reps) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  hipLaunchKernelGGL((k_max<INFLIGHT, FINISH>), dim3(blocks), dim3(256), 0, 0, src, nvec, out, dout);
  hipDeviceSynchronize();
  hipEventRecord(a, 0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((k_max<INFLIGHT, FINISH>), dim3(blocks), dim3(256), 0, 0, src, nvec, out, dout);
  hipEventRecord(b, 0);
  hipEventSynchronize(b);
  float ms;
  hipEventElapsedTime(&ms, a, b);
  return ms / reps;
}

int main() {
  const int64_t big = 8ll << 30;
  uint4* src;
  uint32_t* out;
  double* dout;
  hipMalloc(&src, big);
  hipMalloc(&out, 1 << 20);
  hipMalloc(&dout, 64);
  hipMemset(src, 0x21, big);
  hipMemset(out, 0, 1 << 20);
  hipMemset(dout, 0, 64);
  const int64_t sizes[] = {32ll << 20, 128ll << 20, 512ll << 20, 2ll << 30, 8ll << 30};
  const int grids[] = {256, 512, 1024, 2048, 4096, 8192, 16384};
  printf("%8s %7s | GB/s: inflight1/store  inflight4/store  inflight8/store  inflight4/atomicMax  inflight4/f64-CAS | us (inflight4/store)\n", "MiB", "blocks");
  for (int64_t bytes : sizes)
    for (int g : grids) {
      const int64_t nvec = bytes / 16;
      const int reps = bytes >= (2ll << 30) ? 5 : 40;
      const float t1 = run<1, 0>(src, nvec, g, out, dout, reps), t4 = run<4, 0>(src, nvec, g, out, dout, reps),
                  t8 = run<8, 0>(src, nvec, g, out, dout, reps), ta = run<4, 1>(src, nvec, g, out, dout, reps),
                  tc = run<4, 2>(src, nvec, g, out, dout, reps);
      printf("%8lld %7d | %8.0f %8.0f %8.0f %8.0f %8.0f | %8.1f\n", (long long)(bytes >> 20), g, bytes / t1 / 1e6, bytes / t4 / 1e6,
             bytes / t8 / 1e6, bytes / ta / 1e6, bytes / tc / 1e6, t4 * 1e3);
    }
  return 0;
}
