// Microbenchmark: cycles per VALU instruction for the TV-stage instruction mix (registers only).
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ float dpp_l(float v, float e) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false)); }
__device__ __forceinline__ float dpp_r(float v, float e) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false)); }
__device__ __forceinline__ float row_l(float v, float e) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false)); }
__device__ __forceinline__ float row_r(float v, float e) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, e), __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false)); }
constexpr int NS = 5;  // stages per wave (like one wave group)
template <int MODE>   // 0 full, 1 no dpp, 2 no rsq, 3 no dpp no rsq (mov/mul instead), 4 pure fma chain of the same length
__global__ __launch_bounds__(1024) void k(float* out, int iters, float gam, float c, float beta) {
  float rr[NS][2], ss[NS][2], p[NS][2], q[NS][2], sol[NS][2];
  const float t0 = threadIdx.x * 0.001f;
  for (int k = 0; k < NS; ++k) for (int j = 0; j < 2; ++j) { rr[k][j] = t0 + k; ss[k][j] = t0 - k; p[k][j] = 0.1f * k; q[k][j] = 0.2f; sol[k][j] = t0; }
  float xa = t0 * 3.f;
  const int lane = threadIdx.x & 63;
  const bool first = lane == 0, last = lane == 63;
  const int addr_l = ((lane + 63) & 63) * 4, addr_r = ((lane + 1) & 63) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int P = 0; P < 2; ++P) {
#pragma unroll
      for (int k = NS - 1; k >= 1; --k) {
        if (MODE == 4) {
#pragma unroll
          for (int j = 0; j < 10; ++j) { rr[k][P] = fmaf(rr[k][P], gam, c); ss[k][P] = fmaf(ss[k][P], gam, c); }
          continue;
        }
        const float ssc = ss[k - 1][P ^ 1];
        const float ssl = (MODE == 16) ? (first ? xa : __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr_l, __builtin_bit_cast(int, ssc)))) : (MODE == 8) ? row_l(ssc, xa) : ((MODE & 1) ? ssc * 0.5f : dpp_l(ssc, xa));
        const float s_ = fmaf(-gam, (rr[k - 1][P ^ 1] - rr[k - 1][P]) + (ssc - ssl), xa);
        sol[k][P] = s_;
        const float solb = sol[k][P ^ 1];
        const float solr = (MODE == 16) ? (last ? xa : __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr_r, __builtin_bit_cast(int, solb)))) : (MODE == 8) ? row_r(solb, xa) : ((MODE & 1) ? solb * 0.25f : dpp_r(solb, xa));
        const float r = fmaf(-c, s_ - solb, rr[k - 1][P]);
        const float s = fmaf(-c, solr - solb, ss[k - 1][P]);
        const float n2 = fmaxf(fmaf(r, r, s * s), 1.f);
        const float inv = (MODE & 2) ? n2 * 0.37f : __builtin_amdgcn_rsqf(n2);
        const float pn = r * inv, qn = s * inv;
        rr[k][P] = fmaf(beta, pn - p[k - 1][P], pn);
        ss[k][P] = fmaf(beta, qn - q[k - 1][P], qn);
        p[k][P] = pn; q[k][P] = qn;
      }
      xa += 1.f;
    }
  }
  float acc = 0.f;
  for (int k = 0; k < NS; ++k) for (int j = 0; j < 2; ++j) acc += rr[k][j] + ss[k][j] + p[k][j] + q[k][j] + sol[k][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}
template <int MODE> void run(float* out, hipEvent_t e0, hipEvent_t e1, const char* name, int vinst_per_stage) {
  const int iters = 4000;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(1024), 0, 0, out, iters, 0.17f, 0.7f, 0.3f);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double stages = (double)iters * 2 * (NS - 1);            // per wave
  const double cyc = ms * 1e-3 * 2.4e9;                          // nominal
  printf("%-22s %.3f ms : %.1f cycles per stage per wave-slot (4 waves/SIMD => x/4 per SIMD): %.2f cycles/stage/SIMD-wave, ~%d VALU/stage -> %.2f cyc per VALU per SIMD\n",
         name, ms, cyc / stages, cyc / stages / 4, vinst_per_stage, cyc / stages / 4 / vinst_per_stage);
}
int main() {
  float* out; hipMalloc(&out, 1 << 24);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  run<0>(out, e0, e1, "full stage", 20);
  run<1>(out, e0, e1, "no dpp (mul instead)", 20);
  run<2>(out, e0, e1, "no rsq (mul instead)", 20);
  run<3>(out, e0, e1, "no dpp, no rsq", 20);
  run<8>(out, e0, e1, "row_shr/row_shl dpp", 20);
  run<16>(out, e0, e1, "ds_bpermute + cndmask", 20);
  return 0;
}
