// Microbenchmark: HBM streaming (read 1 GiB + write 1 GiB) with 4-byte vs 16-byte accesses per lane on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ __launch_bounds__(256) void copy1(const float* __restrict__ x, float* __restrict__ y, size_t n, float a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) y[i] = x[i] * a;
}
__global__ __launch_bounds__(256) void copy4(const float4* __restrict__ x, float4* __restrict__ y, size_t n4, float a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = x[i]; v.x *= a; v.y *= a; v.z *= a; v.w *= a; y[i] = v;
  }
}
// row-tiled: block = 256 threads copies a 32 x 64 tile (like lmc_step_point), dword accesses
__global__ __launch_bounds__(256) void copy_tile(const float* __restrict__ x, float* __restrict__ y, int H, int W, float a) {
  const int tiles_x = W / 64, tiles_y = H / 32;
  const int b = blockIdx.x, chain = b / (tiles_x * tiles_y), t = b % (tiles_x * tiles_y);
  const int ty0 = (t / tiles_x) * 32, tx0 = (t % tiles_x) * 64;
  const size_t base = (size_t)chain * H * W;
  const int tx = threadIdx.x & 63, tq = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < 8; ++j) { const size_t gi = base + (size_t)(ty0 + tq * 8 + j) * W + tx0 + tx; y[gi] = x[gi] * a; }
}
int main() {
  const size_t n = (size_t)1 << 28;  // 1 GiB of floats
  float *x, *y; hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMemset(x, 0, n * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto report = [&](const char* name, float ms) { printf("%-28s %.3f ms  %.0f GB/s (read+write)\n", name, ms, 2.0 * n * 4 / (ms * 1e-3) / 1e9); };
  float ms;
  for (int grid : {2048, 8192, 65536}) {
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(copy1, dim3(grid), dim3(256), 0, 0, x, y, n, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1); }
    hipEventElapsedTime(&ms, e0, e1); char nm[64]; snprintf(nm, 64, "dword   grid=%d", grid); report(nm, ms);
    for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(copy4, dim3(grid), dim3(256), 0, 0, (const float4*)x, (float4*)y, n / 4, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1); }
    hipEventElapsedTime(&ms, e0, e1); snprintf(nm, 64, "dwordx4 grid=%d", grid); report(nm, ms);
  }
  for (int rep = 0; rep < 2; ++rep) { hipEventRecord(e0); hipLaunchKernelGGL(copy_tile, dim3(1024 * 8 * 16), dim3(256), 0, 0, x, y, 512, 512, 1.5f); hipEventRecord(e1); hipEventSynchronize(e1); }
  hipEventElapsedTime(&ms, e0, e1); report("dword 32x64 tiles, 131072 blk", ms);
  return 0;
}
