#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k(int* out) {
  unsigned v;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = (int)v; out[2 * blockIdx.x + 1] = (int)hw; }
}
int main() {
  const int n = 4096;
  int* d; hipMalloc(&d, n * 8);
  hipLaunchKernelGGL(k, dim3(n), dim3(256), 0, 0, d);
  std::vector<int> h(2 * n);
  hipMemcpy(h.data(), d, n * 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 64; ++i) printf("%d:%d ", i, h[2 * i] & 0xF);
  printf("\n");
  int hist[16] = {0}; int rr = 0;
  for (int i = 0; i < n; ++i) { hist[h[2*i] & 0xF]++; rr += ((h[2*i] & 0xF) == (i % 8)); }
  for (int i = 0; i < 16; ++i) printf("xcc%d=%d ", i, hist[i]);
  printf("\nmatches b%%8: %d of %d\n", rr, n);
  return 0;
}
