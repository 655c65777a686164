// Measured streaming ceiling of the box (SURVEY.md 8d asks for it beside the nominal 8 TB/s): the library's own
// 16-byte-per-lane grid-stride kernels -- a read-only pass (what the flat-field maxima pass is), a read + write copy
// (the correction pass), a write-only fill -- in two launch shapes: `blocks` workgroups walking the buffer with four
// loads in flight per lane (a resident round: how the hot path's streaming passes are launched), or, blocks = 0, one
// 16-byte access per lane and as many workgroups as that takes (the shape that moved the most bytes per second in
// tools/micro/copy_bw.hip: 6.2 TB/s copied, 4.6 in a resident round).  bench.py times them with HIP events.
#include "mg_common.h"

namespace {

template <int MODE>  // 0: copy, 1: read, 2: fill
__global__ __launch_bounds__(256) void k_stream(const uint4* __restrict__ src, uint4* __restrict__ dst, int64_t nvec,
                                                uint32_t* __restrict__ d_sink) {
  const int64_t stride = (int64_t)gridDim.x * 256;
  int64_t v = (int64_t)blockIdx.x * 256 + threadIdx.x;
  uint32_t acc = 0;
  for (; v + 3 * stride < nvec; v += 4 * stride) {
    uint4 x[4];
    if (MODE != 2) {
#pragma unroll
      for (int q = 0; q < 4; ++q) x[q] = src[v + q * stride];
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (MODE == 0) dst[v + q * stride] = x[q];
      if (MODE == 1) acc ^= x[q].x ^ x[q].y ^ x[q].z ^ x[q].w;
      if (MODE == 2) dst[v + q * stride] = make_uint4(1u, 2u, 3u, 4u);
    }
  }
  for (; v < nvec; v += stride) {
    if (MODE == 0) dst[v] = src[v];
    if (MODE == 1) {
      const uint4 x = src[v];
      acc ^= x.x ^ x.y ^ x.z ^ x.w;
    }
    if (MODE == 2) dst[v] = make_uint4(1u, 2u, 3u, 4u);
  }
  // (what keeps the loads alive: a store that a lane makes only if the XOR of its words is one particular value)
  if (MODE == 1 && acc == 0x9E3779B9u && d_sink) d_sink[0] = acc;
}

}  // namespace

extern "C" int mg_stream_probe(const void* d_src, void* d_dst, int64_t n_bytes, int mode, uint32_t* d_sink, int blocks,
                               void* stream) {
  if (n_bytes < 0 || (n_bytes & 15) || mode < 0 || mode > 2 || blocks < 0 || blocks > 65535) return MG_EINVAL;
  if (blocks == 0) {  // one trip per lane
    if (n_bytes / 16 / 256 + 1 > 0x7FFFFFF0) return MG_EINVAL;
    blocks = (int)((n_bytes / 16 + 255) / 256);
  }
  if ((mode != 2 && !d_src) || (mode != 1 && !d_dst) || (mode == 1 && !d_sink)) return MG_EINVAL;
  if ((reinterpret_cast<uintptr_t>(d_src) & 15) || (reinterpret_cast<uintptr_t>(d_dst) & 15)) return MG_EINVAL;
  if (n_bytes == 0) return MG_OK;
  hipStream_t s = mg_stream(stream);
  const int64_t nvec = n_bytes / 16;
  if (mode == 0)
    hipLaunchKernelGGL(k_stream<0>, dim3(blocks), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, nvec, d_sink);
  else if (mode == 1)
    hipLaunchKernelGGL(k_stream<1>, dim3(blocks), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, nvec, d_sink);
  else
    hipLaunchKernelGGL(k_stream<2>, dim3(blocks), dim3(256), 0, s, (const uint4*)d_src, (uint4*)d_dst, nvec, d_sink);
  MG_CHECK_LAUNCH();
  return MG_OK;
}
