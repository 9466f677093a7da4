// Throughput of dword-aligned 16-byte global accesses against 16-byte-aligned ones (the AL = false path of the step kernels):
// each lane copies 8 consecutive floats (two 16-byte accesses), rows of W floats, W % 4 = off.  Build: hipcc -O3 --offload-arch=gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));
template <bool LD, bool ST>
__global__ __launch_bounds__(256) void copy_k(const float* __restrict__ a, float* __restrict__ b, size_t n8, int off) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    const float* p = a + i * 8 + (LD ? off : 0);
    float* q = b + i * 8 + (ST ? off : 0);
    const f4u v0 = *reinterpret_cast<const f4u*>(p), v1 = *reinterpret_cast<const f4u*>(p + 4);
    *reinterpret_cast<f4u*>(q) = v0; *reinterpret_cast<f4u*>(q + 4) = v1;
  }
}
int main() {
  const size_t n = (size_t)1 << 28;        // 1 GiB
  float *a, *b;
  hipMalloc(&a, (n + 64) * 4); hipMalloc(&b, (n + 64) * 4);
  hipMemset(a, 0, (n + 64) * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int off = 0; off < 4; ++off)
    for (int mode = 0; mode < 3; ++mode) {
      float best = 1e9f;
      for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL((copy_k<true, true>), dim3(8192), dim3(256), 0, 0, a, b, n / 8, off);
        else if (mode == 1) hipLaunchKernelGGL((copy_k<true, false>), dim3(8192), dim3(256), 0, 0, a, b, n / 8, off);
        else hipLaunchKernelGGL((copy_k<false, true>), dim3(8192), dim3(256), 0, 0, a, b, n / 8, off);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("offset %d floats, %s: %.3f ms = %.0f GB/s\n", off, mode == 0 ? "load+store shifted" : mode == 1 ? "load shifted" : "store shifted", best,
             2.0 * n * 4 / best * 1e-6);
    }
  return 0;
}
