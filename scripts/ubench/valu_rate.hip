// Microbenchmark: issue rate of v_fma_f32 vs v_pk_fma_f32 (wave64) on gfx950, waves/SIMD = 1, 2, 4.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));
template <bool PK>
__global__ void k(float* out, int iters, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  v2f y0 = {x0, x1}, y1 = {x2, x3}, y2 = {x4, x5}, y3 = {x6, x7}, va = {a, a}, vb = {b, b};
  for (int i = 0; i < iters; ++i) {
    if (PK) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        y0 = __builtin_elementwise_fma(y0, va, vb); y1 = __builtin_elementwise_fma(y1, va, vb);
        y2 = __builtin_elementwise_fma(y2, va, vb); y3 = __builtin_elementwise_fma(y3, va, vb);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
      }
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = PK ? (y0.x + y0.y + y1.x + y1.y + y2.x + y2.y + y3.x + y3.y) : (x0 + x1 + x2 + x3);
}
int main() {
  float* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int wps = 1; wps <= 4; wps *= 2) {       // waves per SIMD: block = 256 threads (1 wave/SIMD), blocks per CU = wps
    for (int pk = 0; pk < 2; ++pk) {
      dim3 grid(256 * wps), block(256);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (pk) hipLaunchKernelGGL(k<true>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
        else hipLaunchKernelGGL(k<false>, grid, block, 0, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
      }
      float ms; hipEventElapsedTime(&ms, e0, e1);
      const double inst_per_wave = (double)iters * 64;            // 64 VALU instructions per loop iteration in both variants
      const double clk = 2.4e9;
      const double cyc_per_inst_per_simd = ms * 1e-3 * clk / (inst_per_wave * wps);
      printf("waves/SIMD=%d %s: %.3f ms -> %.2f cycles per wave-instruction per SIMD (at 2.4 GHz nominal); flops/lane/instr=%d\n", wps,
             pk ? "v_pk_fma_f32" : "v_fma_f32  ", ms, cyc_per_inst_per_simd, pk ? 4 : 2);
    }
  }
  return 0;
}
