// What does a streaming copy / read / fill reach on this box, and with which launch shape?  16 bytes per lane; knobs:
// loads in flight per lane, cache policy (plain / nontemporal loads / nontemporal stores / both), address layout
// (grid-stride: consecutive workgroups touch consecutive 4 KiB; block-contiguous: a workgroup walks its own contiguous
// span), workgroups (a resident round or one trip per lane), buffer size.  hipMemcpyDtoD beside them.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/micro/copy_bw.hip -o tools/micro/copy_bw
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ u32x4 ld(const u32x4* p) {
  if (NT & 1) return __builtin_nontemporal_load(p);
  return *p;
}
template <int NT>
__device__ __forceinline__ void st(u32x4* p, u32x4 v) {
  if (NT & 2) __builtin_nontemporal_store(v, p);
  else *p = v;
}

// MODE 0 copy, 1 read, 2 fill.  LAYOUT 0: grid-stride; 1: block-contiguous spans
template <int MODE, int INFLIGHT, int NT, int LAYOUT>
__global__ __launch_bounds__(256) void k_stream(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int64_t nvec,
                                                uint32_t* __restrict__ sink) {
  int64_t v, end, step;
  if (LAYOUT == 0) {
    step = (int64_t)gridDim.x * 256;
    v = (int64_t)blockIdx.x * 256 + threadIdx.x;
    end = nvec;
  } else {
    const int64_t span = ((nvec + gridDim.x - 1) / gridDim.x + 255) / 256 * 256;
    v = (int64_t)blockIdx.x * span + threadIdx.x;
    end = std::min<int64_t>(nvec, (int64_t)(blockIdx.x + 1) * span);
    step = 256;
  }
  uint32_t acc = 0;
  for (; v + (INFLIGHT - 1) * step < end; v += INFLIGHT * step) {
    u32x4 x[INFLIGHT];
    if (MODE != 2) {
#pragma unroll
      for (int q = 0; q < INFLIGHT; ++q) x[q] = ld<NT>(src + v + q * step);
    }
#pragma unroll
    for (int q = 0; q < INFLIGHT; ++q) {
      if (MODE == 0) st<NT>(dst + v + q * step, x[q]);
      if (MODE == 1) acc ^= x[q].x ^ x[q].y ^ x[q].z ^ x[q].w;
      if (MODE == 2) st<NT>(dst + v + q * step, u32x4{1u, 2u, 3u, 4u});
    }
  }
  for (; v < end; v += step) {
    if (MODE == 0) st<NT>(dst + v, ld<NT>(src + v));
    if (MODE == 1) {
      const u32x4 x = ld<NT>(src + v);
      acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (MODE == 2) st<NT>(dst + v, u32x4{1u, 2u, 3u, 4u});
  }
  if (MODE == 1 && acc == 0x12345u) sink[0] = acc;
}

struct Case {
  const char* name;
  void (*launch)(const u32x4*, u32x4*, int64_t, uint32_t*, int);
};

template <int MODE, int INFLIGHT, int NT, int LAYOUT>
void launch(const u32x4* s, u32x4* d, int64_t nvec, uint32_t* sink, int blocks) {
  hipLaunchKernelGGL((k_stream<MODE, INFLIGHT, NT, LAYOUT>), dim3(blocks), dim3(256), 0, 0, s, d, nvec, sink);
}

template <int MODE, int INFLIGHT, int NT, int LAYOUT>
void bench(const char* what, const u32x4* s, u32x4* d, int64_t nvec, uint32_t* sink, const std::vector<int>& grids) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const double bytes = (double)nvec * 16 * (MODE == 0 ? 2 : 1);
  for (int blocks : grids) {
    int g = blocks;
    if (g <= 0) g = (int)std::min<int64_t>((nvec + 256LL * INFLIGHT - 1) / (256LL * INFLIGHT), 0x7FFFFFFF);  // one trip per lane
    launch<MODE, INFLIGHT, NT, LAYOUT>(s, d, nvec, sink, g);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(a, 0);
      launch<MODE, INFLIGHT, NT, LAYOUT>(s, d, nvec, sink, g);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      best = std::min(best, ms);
    }
    printf("%-5s inflight %d nt %d layout %d blocks %8d  %7.3f ms  %7.1f GB/s\n", what, INFLIGHT, NT, LAYOUT, g, best,
           bytes / best * 1e-6);
  }
  fflush(stdout);
}

template <int MODE>
void sweep(const char* what, const u32x4* s, u32x4* d, int64_t nvec, uint32_t* sink) {
  const std::vector<int> grids = {1024, 2048, 4096, 8192, 0};
  bench<MODE, 1, 0, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 2, 0, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 4, 0, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 8, 0, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 4, 1, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 4, 2, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 4, 3, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 8, 3, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 2, 3, 0>(what, s, d, nvec, sink, grids);
  bench<MODE, 4, 0, 1>(what, s, d, nvec, sink, grids);
  bench<MODE, 4, 3, 1>(what, s, d, nvec, sink, grids);
  bench<MODE, 8, 3, 1>(what, s, d, nvec, sink, grids);
}

int main(int argc, char** argv) {
  const int64_t mib = argc > 1 ? atoll(argv[1]) : 1024;
  const int64_t bytes = mib << 20, nvec = bytes / 16;
  u32x4 *s, *d;
  uint32_t* sink;
  if (hipMalloc(&s, bytes) != hipSuccess || hipMalloc(&d, bytes) != hipSuccess || hipMalloc(&sink, 4096) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  hipMemset(s, 1, bytes);
  hipMemset(d, 2, bytes);
  hipDeviceSynchronize();
  printf("buffer %lld MiB\n", (long long)mib);
  {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
      hipEventRecord(a, 0);
      hipMemcpyAsync(d, s, bytes, hipMemcpyDeviceToDevice, 0);
      hipEventRecord(b, 0);
      hipEventSynchronize(b);
      float ms;
      hipEventElapsedTime(&ms, a, b);
      best = std::min(best, ms);
    }
    printf("hipMemcpyDtoD  %7.3f ms  %7.1f GB/s (read + written)\n", best, 2.0 * bytes / best * 1e-6);
  }
  sweep<0>("copy", s, d, nvec, sink);
  sweep<1>("read", s, d, nvec, sink);
  sweep<2>("fill", s, d, nvec, sink);
  hipFree(s);
  hipFree(d);
  hipFree(sink);
  return 0;
}
