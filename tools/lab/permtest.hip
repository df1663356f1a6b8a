// Lab: semantics of v_permlane32_swap as exposed by __builtin_amdgcn_permlane32_swap(x, y, fi, bc)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  const unsigned lane = threadIdx.x;
  const unsigned x = 100 + lane, y = 200 + lane;
  const auto sw = __builtin_amdgcn_permlane32_swap(x, y, false, false);
  out[lane] = sw[0];
  out[64 + lane] = sw[1];
}
int main() {
  unsigned* d;
  hipMalloc(&d, 128 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  unsigned h[128];
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("x=100+lane y=200+lane\nsw[0]: lane0 %u lane31 %u lane32 %u lane63 %u\nsw[1]: lane0 %u lane31 %u lane32 %u lane63 %u\n", h[0], h[31], h[32],
         h[63], h[64], h[95], h[96], h[127]);
  return 0;
}
