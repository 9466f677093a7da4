// Which SIMD does wave w of a workgroup land on?  (gfx950, HW_ID: simd_id = bits [5:4], wave_id = [3:0], cu_id = [11:8])
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + (threadIdx.x >> 6)] = id;
}
int main() {
  unsigned* d; hipMalloc(&d, 4096 * 4);
  for (int nw : {7, 8, 10, 16}) {
    hipLaunchKernelGGL(k, dim3(4), dim3(64 * nw), 100 * 1024, 0, d);
    unsigned h[64]; hipMemcpy(h, d, sizeof(unsigned) * 4 * nw, hipMemcpyDeviceToHost);
    for (int b = 0; b < 2; ++b) {
      printf("waves/wg=%d block %d: simd of wave 0..: ", nw, b);
      for (int w = 0; w < nw; ++w) printf("%u ", (h[b * nw + w] >> 4) & 3);
      printf(" | cu %u\n", (h[b * nw] >> 8) & 15);
    }
  }
  return 0;
}
