// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the instructions that dominate the LMC kernels on gfx950:
// v_xor_b32, v_mad_u64_u32, v_mul_lo_u32, v_mul_hi_u32, v_mul_u32_u24, v_rsq_f32, v_log_f32, v_sin_f32, v_sqrt_f32, v_fma_f32, v_pk_fma_f32,
// v_fma_f64, v_cvt_f32_u32.  One workgroup of 256 threads per CU (1 wave per SIMD), 8 independent chains per lane, clock64 around the loop.
// build: hipcc --offload-arch=gfx950 -O2 -o inst_rate inst_rate.hip ; run: ./inst_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(1024) void rate_kernel(unsigned* out, unsigned long long* cyc, int iters, unsigned seed) {
  unsigned a[8];
  unsigned long long w[8];
  float f[8];
  double d[8];
  typedef float v2f __attribute__((ext_vector_type(2)));
  v2f p[8];
  for (int i = 0; i < 8; ++i) {
    a[i] = seed + threadIdx.x * 8 + i; w[i] = a[i]; f[i] = 1.0f + 1e-3f * (float)(a[i] & 1023); d[i] = f[i]; p[i] = v2f{f[i], f[i] + 1.f};
  }
  const unsigned m = 0xD2511F53u ^ seed;
  const float fm = 1.0001f;
  __syncthreads();
  const unsigned long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#define X_XOR(i) asm volatile("v_xor_b32 %0, %1, %0" : "+v"(a[i]) : "v"(m));
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, s[10:11], %1, %2, %0" : "+v"(w[i]) : "v"(a[i]), "v"(m) : "s10", "s11");
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define X_MUL24(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(a[i]) : "v"(m));
#define X_RSQ(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(f[i]));
#define X_LOG(i) asm volatile("v_log_f32 %0, %0" : "+v"(f[i]));
#define X_SIN(i) asm volatile("v_sin_f32 %0, %0" : "+v"(f[i]));
#define X_SQRT(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(f[i]));
#define X_FMA(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(fm));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i]) : "v"(p[(i + 1) & 7]));
#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(d[i]) : "v"(d[(i + 1) & 7]));
#define X_CVT(i) asm volatile("v_cvt_f32_u32 %0, %1" : "=v"(f[i]) : "v"(a[i]));
    if constexpr (OP == 0) { REP8(X_XOR) REP8(X_XOR) }
    if constexpr (OP == 1) { REP8(X_MAD64) REP8(X_MAD64) }
    if constexpr (OP == 2) { REP8(X_MULLO) REP8(X_MULLO) }
    if constexpr (OP == 3) { REP8(X_MULHI) REP8(X_MULHI) }
    if constexpr (OP == 4) { REP8(X_MUL24) REP8(X_MUL24) }
    if constexpr (OP == 5) { REP8(X_RSQ) REP8(X_RSQ) }
    if constexpr (OP == 6) { REP8(X_LOG) REP8(X_LOG) }
    if constexpr (OP == 7) { REP8(X_SIN) REP8(X_SIN) }
    if constexpr (OP == 8) { REP8(X_SQRT) REP8(X_SQRT) }
    if constexpr (OP == 9) { REP8(X_FMA) REP8(X_FMA) }
    if constexpr (OP == 10) { REP8(X_PKFMA) REP8(X_PKFMA) }
    if constexpr (OP == 11) { REP8(X_FMA64) REP8(X_FMA64) }
    if constexpr (OP == 12) { REP8(X_CVT) REP8(X_CVT) }
  }
  const unsigned long long t1 = clock64();
  unsigned acc = 0;
  for (int i = 0; i < 8; ++i) acc ^= a[i] ^ (unsigned)w[i] ^ __float_as_uint(f[i]) ^ (unsigned)d[i] ^ __float_as_uint(p[i].x);
  out[blockIdx.x * 1024 + threadIdx.x] = acc;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char* name, int threads) {
  const int blocks = 256, iters = 65536;
  unsigned* out; unsigned long long* cyc;
  hipMalloc(&out, blocks * 1024 * 4); hipMalloc(&cyc, blocks * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 16, 1u);
  hipEventRecord(e0);
  hipLaunchKernelGGL(rate_kernel<OP>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters, 1u);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  std::vector<unsigned long long> h(blocks);
  hipMemcpy(h.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
  double mean = 0; for (auto v : h) mean += (double)v; mean /= blocks;
  const double waves_per_simd = threads / 256.0;
  // clock64 (s_memtime) ticks at 100 MHz on this part: report wall-clock ns per instruction per SIMD instead
  const double inst_per_simd = (double)iters * 16 * waves_per_simd;
  printf("%-14s waves/SIMD %.0f  %.3f ms  %.2f ns per wave-instruction per SIMD  (= %.1f cycles at 2.4 GHz)\n", name, waves_per_simd, ms,
         ms * 1e6 / inst_per_simd, ms * 1e6 / inst_per_simd * 2.4);
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512, 1024}) {
    run<9>("v_fma_f32", threads); run<10>("v_pk_fma_f32", threads); run<0>("v_xor_b32", threads); run<1>("v_mad_u64_u32", threads);
    run<2>("v_mul_lo_u32", threads); run<3>("v_mul_hi_u32", threads); run<4>("v_mul_u32_u24", threads); run<12>("v_cvt_f32_u32", threads);
    run<5>("v_rsq_f32", threads); run<6>("v_log_f32", threads); run<7>("v_sin_f32", threads); run<8>("v_sqrt_f32", threads); run<11>("v_fma_f64", threads);
  }
  return 0;
}
