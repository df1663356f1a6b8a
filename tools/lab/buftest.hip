// Lab: does a raw buffer store / load beyond num_records get dropped / return 0 on gfx950 (voffset vs soffset)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* buf, unsigned nbytes, int mode, unsigned* out) {
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)buf, 0, nbytes, 0x00020000);
  const int lane = threadIdx.x;
  u32x4 v = {7u, 7u, 7u, 7u};
  const int off = lane * 64;                       // lanes 0..63 -> byte offsets 0..4032; nbytes = 2048 -> lanes >= 32 are out of range
  if (mode == 0) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 0);            // all in voffset
  if (mode == 1) __builtin_amdgcn_raw_buffer_store_b128(v, r, lane * 32, lane * 0 + 1024, 0);   // part in soffset (uniform 1024)
  if (mode == 2) __builtin_amdgcn_raw_buffer_store_b128(v, r, off, 0, 2);            // nt
  if (mode == 3) {
    u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, off, 0, 0);
    out[lane] = x[0];
  }
  if (mode == 4) {
    u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(r, lane * 32, 1024, 0);
    out[lane] = x[0];
  }
}
int main() {
  unsigned *d, *o;
  hipMalloc(&d, 8192);
  hipMalloc(&o, 256);
  std::vector<unsigned> h(2048);
  for (int mode = 0; mode < 5; ++mode) {
    hipMemset(d, 0x11, 8192);
    hipMemset(o, 0, 256);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 2048u, mode, o);
    hipDeviceSynchronize();
    hipMemcpy(h.data(), d, 8192, hipMemcpyDeviceToHost);
    int written_in = 0, written_out = 0;
    for (int i = 0; i < 2048; ++i)
      if (h[i] == 7u) (i * 4 < 2048 ? written_in : written_out)++;
    unsigned ho[64];
    hipMemcpy(ho, o, 256, hipMemcpyDeviceToHost);
    printf("mode %d: dwords written inside %d, outside %d | loads: lane0 %x lane31 %x lane32 %x lane63 %x\n", mode, written_in, written_out, ho[0],
           ho[31], ho[32], ho[63]);
  }
  return 0;
}
