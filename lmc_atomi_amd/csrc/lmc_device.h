// Device-side building blocks: Philox4x32-10, Box-Muller, wave/block reductions.
#pragma once
#include <type_traits>
#include "lmc_common.h"

namespace lmc {

// Philox4x32-10 (Salmon et al., SC'11).  One call -> 4 x uint32.
__device__ __forceinline__ void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                              uint32_t k0, uint32_t k1, uint32_t (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
    c1 = (uint32_t)p1;
    c3 = (uint32_t)p0;
    c0 = n0;
    c2 = n2;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// uint32 -> (0,1]: u*2^-32 + 2^-33, one fma (v_cvt_f32_u32 rounds to nearest even).
__device__ __forceinline__ float u01(uint32_t u) {
  return fmaf((float)u, 0x1p-32f, 0x1p-33f);
}

// Two N(0,1) per uint32 pair: r = sqrt(-2 ln u_a); (r sin 2*pi*u_b, r cos 2*pi*u_b).
// Hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
// revolutions, so 2*pi*u_b needs no multiply and no range reduction: u_b is in (0,1]).
// Absolute error of a deviate vs exact arithmetic: < 2e-5 (tests/test_gpu_parity.py), i.e. < 1e-5 of
// the noise scale sqrt(2 tau) -- statistically irrelevant, and identical in every kernel that draws noise.
__device__ __forceinline__ void box_muller(uint32_t ua, uint32_t ub, float& n0, float& n1) {
  const float u1 = u01(ua), u2 = u01(ub);
  const float r = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));  // -2 ln2 log2(u1)
  n0 = r * __builtin_amdgcn_sinf(u2);
  n1 = r * __builtin_amdgcn_cosf(u2);
}

// The 4 normals of quad (row>>2, col) of `chain` at `iteration`: rows 4q+0..3 of column col.
__device__ __forceinline__ void quad_normals(uint32_t key0, uint32_t key1, uint32_t iteration,
                                             uint32_t chain, uint32_t quad, float (&n)[4]) {
  uint32_t o[4];
  philox4x32_10(quad, iteration, chain, kPhiloxStream, key0, key1, o);
  box_muller(o[0], o[1], n[0], n[1]);
  box_muller(o[2], o[3], n[2], n[3]);
}

// 64-lane wavefront sum; result valid in lane 0.
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Block sum via wave shuffles + one LDS slot per wave; result valid in thread 0.
// `scratch` must hold blockDim.x/64 doubles.
__device__ __forceinline__ double block_sum(double v, double* scratch) {
  const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
  v = wave_sum(v);
  if (lane == 0) scratch[wid] = v;
  __syncthreads();
  double tot = 0.0;
  if (threadIdx.x == 0) {
    const int nw = (blockDim.x + kWave - 1) >> 6;
    for (int i = 0; i < nw; ++i) tot += scratch[i];
  }
  __syncthreads();
  return tot;
}

// Bijective XCD-aware remap: hardware deals consecutive block ids round-robin over the 8 XCDs;
// give each XCD a contiguous range of logical ids so tiles of one chain share an L2.
__device__ __forceinline__ int xcd_logical_block(int b, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, xcd = b & 7;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (b >> 3);
}

// ---- MC-TV term of algs.L2_ncvx_tv.grad (algs.py:273-277): A^T( A x / max(|A x|, gamma) ) at pixel (i,j) ----------
// x_rc: x at row offset r (m = -1, 0 = same, p = +1) and column offset c.  has_*: the neighbour exists inside the image.
// |A x| = 0 is mapped to 1e-9 by the reference before min(1/gamma, 1/|A x|): identical to 1/max(|A x|, gamma).
// gamma < 0 (wave-uniform): the ANISOTROPIC branches (algs.py:218-219, 278-279) with |gamma|: A^T( (v - soft(v, gamma)) / gamma ) = A^T of
// v_i / max(|v_i|, gamma) component by component -- the weight of a difference depends on that difference alone.
__device__ __forceinline__ float mc_tv_grad(float xm0, float xmp, float x0m, float x00, float x0p, float xpm, float xp0,
                                            bool has_up, bool has_down, bool has_left, bool has_right, float gamma) {
  const bool an = gamma < 0.f;
  const float g = fabsf(gamma);
  // weights and differences at (i,j), (i-1,j), (i,j-1)
  const float dx00 = has_down ? xp0 - x00 : 0.f, dy00 = has_right ? x0p - x00 : 0.f;
  const float dxm0 = has_up ? x00 - xm0 : 0.f;                       // (i-1,j) always has a row below it
  const float dym0 = (has_up && has_right) ? xmp - xm0 : 0.f;
  const float dy0m = has_left ? x00 - x0m : 0.f;                     // (i,j-1) always has a column to its right
  const float dx0m = (has_left && has_down) ? xpm - x0m : 0.f;
  if (an) {
    const float wx00 = __builtin_amdgcn_rcpf(fmaxf(fabsf(dx00), g)), wy00 = __builtin_amdgcn_rcpf(fmaxf(fabsf(dy00), g));
    const float wxm0 = __builtin_amdgcn_rcpf(fmaxf(fabsf(dxm0), g)), wy0m = __builtin_amdgcn_rcpf(fmaxf(fabsf(dy0m), g));
    return -((wx00 * dx00 - wxm0 * dxm0) + (wy00 * dy00 - wy0m * dy0m));
  }
  const float w00 = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(fmaf(dx00, dx00, dy00 * dy00)), g));
  const float wm0 = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(fmaf(dxm0, dxm0, dym0 * dym0)), g));
  const float w0m = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(fmaf(dx0m, dx0m, dy0m * dy0m)), g));
  // A^T v = -div v
  return -((w00 * dx00 - wm0 * dxm0) + (w00 * dy00 - w0m * dy0m));
}

// Moreau envelope of |.| (Huber) at the gradient (dx, dy) of a pixel: of the pixel norm (gamma > 0, algs.py:173-190) or of each component
// (gamma < 0: the anisotropic branch, |gamma|)
__device__ __forceinline__ double mc_tv_envelope(float dx, float dy, float gamma) {
  const double g = fabsf(gamma);
  auto hub = [&](double e) { return e <= g ? 0.5 * e * e / g : e - 0.5 * g; };
  if (gamma < 0.f) return hub(fabsf(dx)) + hub(fabsf(dy));
  return hub(sqrtf(fmaf(dx, dx, dy * dy)));
}

// ---- shared by the streaming step kernels ---------------------------------------------------------
constexpr int kPad = 8;  // zero columns on both sides of LDS rows (>= kMaxBlur - 1)
#ifdef LMC_BOUNDS_CHECK   // debug build: out-of-range global accesses are recorded and skipped, never issued
static __device__ long long lmc_dbg[8];
__device__ __forceinline__ float ld_chk(const float* base, long long idx, long long n, int tag) {
  if (base == nullptr || idx < 0 || idx >= n) { lmc_dbg[0] = tag; lmc_dbg[1] = idx; lmc_dbg[2] = n; lmc_dbg[3] = (long long)base; return 0.f; }
  return base[idx];
}
__device__ __forceinline__ void st_chk(float* base, long long idx, long long n, float v, int tag) {
  if (base == nullptr || idx < 0 || idx >= n) { lmc_dbg[0] = tag; lmc_dbg[1] = idx; lmc_dbg[2] = n; lmc_dbg[3] = (long long)base; return; }
  base[idx] = v;
}
#define LD(base, idx, n, tag) ld_chk(base, (long long)(idx), (long long)(n), tag)
#define ST(base, idx, n, v, tag) st_chk(base, (long long)(idx), (long long)(n), v, tag)
#else
#define LD(base, idx, n, tag) (base)[idx]
#define ST(base, idx, n, v, tag) (base)[idx] = (v)
#endif

__device__ __forceinline__ float dpp_from_left(float v, float edge) {   // lane i <- v[i-1]; lane 0 <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                              __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float v, float edge) {  // lane i <- v[i+1]; lane 63 <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                              __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

// wave shifts with zero fill through bound_ctrl: no "old" register to initialise (one v_mov less per shift)
// ---- rows that start on 4-byte boundaries only (W % 4 != 0, e.g. the reference's 667 x 877 image) ---------------------------------------
// gfx950 under ROCm runs with unaligned access enabled: a 16-byte global access needs dword alignment only, and the compiler emits
// global_load/store_dwordx4 for a 4-byte-aligned vector type.
typedef float f4u __attribute__((ext_vector_type(4), aligned(4)));

// Pixels c .. c+3 of a row of W >= 4 floats -> d[0..3], one 16-byte access that never leaves the row: the group that holds the row end
// (c < W < c + 4) reads the last four floats of the row and moves them down by 4 - W % 4 (wave-uniform) places; a group past the row reads its
// start.  Elements at columns >= W come back undefined: every caller masks per pixel.
__device__ __forceinline__ void load4_dword_aligned_raw(float& d0, float& d1, float& d2, float& d3, const float* __restrict__ row, int c, int W) {
  const bool part = c < W && c + 4 > W;
  const int cc = c + 4 <= W ? c : (part ? W - 4 : 0);
  const f4u v = *reinterpret_cast<const f4u*>(row + cc);
  d0 = v.x; d1 = v.y; d2 = v.z; d3 = v.w;
}
// ... and the move into place, kept apart from the load: applied where the values are USED, so that a prefetched row stays in flight (a select
// at load time makes the wave wait for the load at once; measured on the 667 x 877 image: pipe 3.13 -> see DESIGN 3.0p, rows 1.02 ms).
__device__ __forceinline__ void unshift4_dword_aligned(float& d0, float& d1, float& d2, float& d3, int c, int W) {
  const bool part = c < W && c + 4 > W;
  const int s = 4 - (W & 3);
  const float v1 = d1, v2 = d2, v3 = d3;
  d0 = part ? (s == 1 ? v1 : (s == 2 ? v2 : v3)) : d0;
  d1 = part ? (s == 1 ? v2 : v3) : v1;
  d2 = part ? v3 : v2;
}
// load and move in one (the row-streaming kernel; written out rather than composed of the two above: the composition costs that register-bound
// kernel 30 more spilled VGPRs)
__device__ __forceinline__ void load4_dword_aligned(float& d0, float& d1, float& d2, float& d3, const float* __restrict__ row, int c, int W) {
  const bool part = c < W && c + 4 > W;
  const int cc = c + 4 <= W ? c : (part ? W - 4 : 0);
  const f4u v = *reinterpret_cast<const f4u*>(row + cc);
  const int s = 4 - (W & 3);
  d0 = part ? (s == 1 ? v.y : (s == 2 ? v.z : v.w)) : v.x;
  d1 = part ? (s == 1 ? v.z : v.w) : v.y;
  d2 = part ? v.w : v.z;
  d3 = v.w;
}
template <int PXL>
__device__ __forceinline__ void unshift_row_dword_aligned(float (&r)[PXL], int c0, int W) {
#pragma unroll
  for (int g = 0; g < PXL / 4; ++g) unshift4_dword_aligned(r[4 * g], r[4 * g + 1], r[4 * g + 2], r[4 * g + 3], c0 + 4 * g, W);
}
// Pixels c .. c+3 -> row, those with columns in [lo, hi) only (hi <= W): one 16-byte store when the group is inside, else pixel by pixel.
__device__ __forceinline__ void store4_dword_aligned(float* __restrict__ row, int c, int lo, int hi, float v0, float v1, float v2, float v3) {
  if (c >= lo && c + 4 <= hi) {
    f4u v; v.x = v0; v.y = v1; v.z = v2; v.w = v3;
    *reinterpret_cast<f4u*>(row + c) = v;
  } else {
    if (c >= lo && c < hi) row[c] = v0;
    if (c + 1 >= lo && c + 1 < hi) row[c + 1] = v1;
    if (c + 2 >= lo && c + 2 < hi) row[c + 2] = v2;
    if (c + 3 >= lo && c + 3 < hi) row[c + 3] = v3;
  }
}

__device__ __forceinline__ float dpp_left0(float v) {    // lane i <- v[i-1]; lane 0 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float dpp_right0(float v) {   // lane i <- v[i+1]; lane 63 <- 0
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// ---- horizontal neighbours at DPP-row (16-lane) granularity -----------------------------------------
// wave_shr / wave_shl DPP moves cost ~12 cycles per wave on gfx950 (measured, scripts/ubench/stage_mix.hip);
// row_shr:1 / row_shl:1 run at the plain VALU rate.  So neighbours inside a 16-lane row come from row
// shifts, and the lane at each row's end takes the value its neighbour row published in LDS one tick
// earlier ("ghost": the values used are one tick old anyway).
__device__ __forceinline__ float row_from_left(float v, float edge) {   // lane i <- v[i-1]; first lane of each row <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                              __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_from_right(float v, float edge) {  // lane i <- v[i+1]; last lane of each row <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                              __builtin_bit_cast(int, v), 0x101, 0xf, 0xf, false));
}
// gather_first<N>(acc, v): in every row, lane N of the result = first lane of v (lanes above N inside N's
// bank are overwritten too and are fixed by later calls with larger N; lanes below N keep acc).
template <int N>
__device__ __forceinline__ float gather_first(float acc, float v) {
  static_assert(N >= 1 && N <= 15, "row_shr distance");
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, acc), __builtin_bit_cast(int, v),
                                                              0x110 + N, 0xf, 1 << (N >> 2), false));
}
// gather_last<N>(acc, v): in every row, lane 15-N of the result = last lane of v (row_shl:N).
template <int N>
__device__ __forceinline__ float gather_last(float acc, float v) {
  static_assert(N >= 1 && N <= 15, "row_shl distance");
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, acc), __builtin_bit_cast(int, v),
                                                              0x100 + N, 0xf, 1 << ((15 - N) >> 2), false));
}

// ---- closed-form elementwise proxes (prox.py:9-85): the functional plugin surface (lmc_prox_elementwise) and the LMC_PRIOR_EPROX priors of the
// fused step kernels ----------------------------------------------------------------------------------------------------------------------
struct EproxParams { float p0, p1; };

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) - (x < 0.f); }
__device__ __forceinline__ float soft(float x, float t) { return copysignf(fmaxf(fabsf(x) - t, 0.f), x); }

__device__ __forceinline__ float eprox(int kind, float x, EproxParams q) {
  const float g = q.p0;
  switch (kind) {
    case LMC_EPROX_LAPLACE: return sgn(x) * fmaxf(fabsf(x) - g, 0.f);
    case LMC_EPROX_UNCENTERED_LAPLACE: { const float d = x - q.p1; return q.p1 + sgn(d) * fmaxf(fabsf(d) - g, 0.f); }
    case LMC_EPROX_GAUSSIAN: return x / (2.f * g + 1.f);
    case LMC_EPROX_GEN_GAUSSIAN_4_3: {
      const float xi = sqrtf(x * x + 256.f * g * g * g / 729.f);
      return x + 4.f * g / (3.f * cbrtf(2.f)) * (cbrtf(xi - x) - cbrtf(xi + x));
    }
    case LMC_EPROX_GEN_GAUSSIAN_3_2:
      return x + 9.f * g * g * sgn(x) * (1.f - sqrtf(1.f + 16.f * fabsf(x) / (9.f * g * g))) / 8.f;
    case LMC_EPROX_GEN_GAUSSIAN_3: return sgn(x) * (sqrtf(1.f + 12.f * g * fabsf(x)) - 1.f) / (6.f * g);
    case LMC_EPROX_GEN_GAUSSIAN_4: {
      const float xi = sqrtf(x * x + 1.f / (27.f * g));
      return cbrtf((xi + x) / (8.f * g)) - cbrtf((xi - x) / (8.f * g));
    }
    case LMC_EPROX_HUBER: {
      const float t = q.p1;
      return fabsf(x) <= g * (2.f * t + 1.f) / sqrtf(2.f * t) ? x / (2.f * t + 1.f) : x - g * sqrtf(2.f * t) * sgn(x);
    }
    case LMC_EPROX_SMOOTHED_LAPLACE: {
      const float ax = fabsf(x), u = g * ax - g * g - 1.f;
      return sgn(x) * (u + sqrtf(u * u + 4.f * g * ax)) / (2.f * g);
    }
    case LMC_EPROX_EXP: return x >= g ? x - g : 0.f;
    case LMC_EPROX_GAMMA: { const float d = x - q.p0; return (d + sqrtf(d * d + 4.f * q.p1)) * 0.5f; }
    case LMC_EPROX_CHI: return (x + sqrtf(x * x + 8.f * q.p0)) * 0.25f;
    case LMC_EPROX_UNIFORM: return fminf(fmaxf(x, -q.p0), q.p0);
    case LMC_EPROX_TRIANGULAR: {
      const float o1 = q.p0, o2 = q.p1;
      if (x < 1.f / o1) return (x + o1 + sqrtf((x - o1) * (x - o1) + 4.f)) * 0.5f;
      if (x > 1.f / o2) return (x + o2 + sqrtf((x - o2) * (x - o2) + 4.f)) * 0.5f;
      return 0.f;
    }
    case LMC_EPROX_LAPLACE_CONJ: {  // x - g * prox_laplace(x/g, 1/g)
      const float z = x / g;
      return x - g * (sgn(z) * fmaxf(fabsf(z) - 1.f / g, 0.f));
    }
  }
  return x;
}

template <int I, int End, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < End) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, End>(f);
  }
}

}  // namespace lmc
