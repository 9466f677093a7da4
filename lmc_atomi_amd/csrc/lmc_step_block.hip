// Fused MYULA update for data terms without a stencil (identity, inpainting mask, none) and a prior whose prox is local to
// an 8 x 8 block -- 3-level Haar-l1 (BASELINE config "inpainting mask + l1-wavelet prox"), l2, l1 or none:
//     out = a*x - t*grad f(x) + b*prox(x) + s*xi                                   (algs.py:569)
//
// One thread owns one 8 x 8 block of one chain: 16 float4 loads of x (the 64 lanes of a wave cover 8 image rows x 512
// contiguous columns); with the Haar prior the block goes to registers (butterflies + soft threshold in place) and a copy to a
// thread-private LDS slab; then, four rows (one Philox quad row-group) at a time, x comes back from the slab (or is simply read,
// for the pointwise priors), the data gradient, prox and noise are combined and stored as float4s.  No barrier, no halo (one
// pixel of halo from memory when the MC-TV term is fused): HBM traffic is x read once + x' written once (8 B per pixel-step);
// y and the mask are shared by all chains and come from L2.
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

#ifndef LMC_BLK_OCC
#define LMC_BLK_OCC 2   // workgroups (4 waves each) per CU the register budget is sized for: 232 VGPRs, no scratch
#endif

__device__ __forceinline__ float soft_thr_b(float v, float thr) { return copysignf(fmaxf(fabsf(v) - thr, 0.f), v); }

// MC: the MC-TV term of L2_ncvx_tv, -lambda * A^T(A x / max(|A x|, gamma)) (algs.py:273-277), added to the gradient: a rolling window
// of three 10-pixel rows (block columns -1 .. 8) around the output row -- the interior from the LDS copy, the halo from the
// neighbouring blocks in memory (cache hits: those lines are being streamed by the neighbouring threads).
template <int DATA, int PRIOR, bool MC = false, int NF = 1>
__global__ __launch_bounds__(256, LMC_BLK_OCC) void myula_step_block_kernel(const StepArgs P) {
  static_assert(NF == 1 || ((NF == 2 || NF == 4) && PRIOR == LMC_PRIOR_HAAR_L1 && !MC), "fused iterations: Haar prior without the MC-TV term");
  const int H = P.H, W = P.W;
  const int nbx = W >> 3;
  const uint32_t blocks_per_img = (uint32_t)nbx * (uint32_t)(H >> 3);
  const size_t bi = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (bi >= (size_t)blocks_per_img * (size_t)P.C) return;
  const uint32_t chain = (uint32_t)(bi / blocks_per_img);
  const uint32_t b = (uint32_t)(bi - (size_t)chain * blocks_per_img);
  const int by = (int)(b / (uint32_t)nbx), bx = (int)(b - (uint32_t)by * (uint32_t)nbx);
  const size_t img = (size_t)H * W;
  const size_t o0 = (size_t)(by * 8) * W + (size_t)bx * 8;      // offset of the block inside an image
  const float* __restrict__ src = P.x_in + (size_t)chain * img + o0;
  float* __restrict__ dst = P.x_out + (size_t)chain * img + o0;

  // Phase 1: the block in registers, prox in place (Haar: butterflies + soft threshold; l2 / l1 / none are pointwise and need no
  // copy at all).  Phase 2 re-reads x row by row (L2 hits: the wave has just streamed these lines) instead of holding a second
  // 64-register copy of the block: 128 VGPRs less, 3-4 waves per SIMD instead of 2.
  float v[8][8];
  __shared__ float4 xs[PRIOR == LMC_PRIOR_HAAR_L1 ? 16 * 256 : 1];    // the thread's own copy of x: [row*2 + half][thread], 64 KB
  // Two iterations per launch (P.fused_iters == 2; Haar prior without the MC-TV term: nothing of the update leaves the thread's block): the first
  // iterate goes back into the slab instead of to memory (and to P.x_mid when the caller keeps it), the second pass reads the slab.
  // (NF = 2: two straight-line copies of the pass -- a run-time loop around it cost 280 spilled VGPRs.)
  float* __restrict__ mid = (NF == 2 && P.x_mid) ? P.x_mid + (size_t)chain * img + o0 : nullptr;      // NF = 4: no iterate in between is kept
  auto pass = [&](auto fitc) __attribute__((always_inline)) {
  constexpr int fit = decltype(fitc)::value;
  constexpr bool last_it = fit + 1 == NF;
  if (PRIOR == LMC_PRIOR_HAAR_L1) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      float4 lo, hi;
      if (fit == 0) {
        lo = *reinterpret_cast<const float4*>(src + (size_t)r * W);
        hi = *reinterpret_cast<const float4*>(src + (size_t)r * W + 4);
      } else {
        lo = xs[(2 * r) * 256 + threadIdx.x];
        hi = xs[(2 * r + 1) * 256 + threadIdx.x];
      }
      v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = lo.z; v[r][3] = lo.w; v[r][4] = hi.x; v[r][5] = hi.y; v[r][6] = hi.z; v[r][7] = hi.w;
      xs[(2 * r) * 256 + threadIdx.x] = lo;
      xs[(2 * r + 1) * 256 + threadIdx.x] = hi;
    }
    const float thr = P.prior_p0;
#pragma unroll
    for (int s = 1; s <= 4; s <<= 1)        // forward: (i, j) holds LL, (i, j+s) LH, (i+s, j) HL, (i+s, j+s) HH of the quad at stride s
#pragma unroll
      for (int i = 0; i < 8; i += 2 * s)
#pragma unroll
        for (int j = 0; j < 8; j += 2 * s) {
          const float a = v[i][j], bb = v[i][j + s], cc = v[i + s][j], d = v[i + s][j + s];
          v[i][j] = 0.5f * (a + bb + cc + d);
          v[i][j + s] = soft_thr_b(0.5f * (a - bb + cc - d), thr);
          v[i + s][j] = soft_thr_b(0.5f * (a + bb - cc - d), thr);
          v[i + s][j + s] = soft_thr_b(0.5f * (a - bb - cc + d), thr);
        }
#pragma unroll
    for (int s = 4; s >= 1; s >>= 1)        // inverse
#pragma unroll
      for (int i = 0; i < 8; i += 2 * s)
#pragma unroll
        for (int j = 0; j < 8; j += 2 * s) {
          const float ll = v[i][j], lh = v[i][j + s], hl = v[i + s][j], hh = v[i + s][j + s];
          v[i][j] = 0.5f * (ll + lh + hl + hh); v[i][j + s] = 0.5f * (ll - lh + hl - hh);
          v[i + s][j] = 0.5f * (ll + lh - hl - hh); v[i + s][j + s] = 0.5f * (ll - lh - hl + hh);
        }
  }
  // Phase 2: out = a*x - t*grad f(x) + b*prox(x) + s*xi, four rows (one Philox quad row-group) at a time
  const float ts = P.t * P.sigma_f;
  // MC-TV window: wr[k][m] = x(row, block column m - 1), rows r-1, r, r+1 in slots (r-1)%3 ... ; addresses are clamped into the image,
  // the has_* flags of mc_tv_grad make the clamped values irrelevant
  // (round 3) v = A x / max(|A x|, gamma) is formed once per pixel, row by row -- (vx, vy) of block row r from window rows r and r + 1, for the block
  // columns -1 .. 7 -- and A^T v at (r, j) = -((vx[r][j] - vx[r-1][j]) + (vy[r][j] - vy[r][j-1])): the same values, operation for operation, as
  // mc_tv_grad recomputes per pixel for its three weights, with a third of the square roots and reciprocals and one window row less in registers.
  float wr[MC ? 2 : 1][MC ? 10 : 1], vxp[MC ? 8 : 1], vxc[MC ? 9 : 1], vyc[MC ? 9 : 1];
  const int gi0 = by * 8, gj0 = bx * 8;
  const float* __restrict__ ximg = P.x_in + (size_t)chain * img;
  // 512 columns = 64 blocks = exactly one wave per block row (and every wave is full: blocks_per_img is a multiple of 64): the one-pixel column halo of the
  // MC-TV window comes from the neighbouring lanes instead of memory (round 3: traffic 1.51 -> see DESIGN 3.0b)
  const bool row_in_wave = MC && nbx == 64;
  auto load_wrow = [&](float (&d)[MC ? 10 : 1], int rr) {      // rr: block-local row -1 .. 8
    if constexpr (MC) {
      const int gi = min(max(gi0 + rr, 0), H - 1);
      const float* row = ximg + (size_t)gi * W;
      if (PRIOR == LMC_PRIOR_HAAR_L1 && rr >= 0 && rr < 8) {
        const float4 lo = xs[(2 * rr) * 256 + threadIdx.x], hi = xs[(2 * rr + 1) * 256 + threadIdx.x];
        d[1] = lo.x; d[2] = lo.y; d[3] = lo.z; d[4] = lo.w; d[5] = hi.x; d[6] = hi.y; d[7] = hi.z; d[8] = hi.w;
      } else {
        const float4 lo = *reinterpret_cast<const float4*>(row + gj0), hi = *reinterpret_cast<const float4*>(row + gj0 + 4);
        d[1] = lo.x; d[2] = lo.y; d[3] = lo.z; d[4] = lo.w; d[5] = hi.x; d[6] = hi.y; d[7] = hi.z; d[8] = hi.w;
      }
      if (row_in_wave) {        // the blocks left and right of this one belong to the neighbouring lanes: their edge pixels by a wave shift (lane 0 / 63 sit
        d[0] = dpp_left0(d[8]);   // at the image edge, where the has-neighbour flags of v_row void the value)
        d[9] = dpp_right0(d[1]);
      } else {                   // two strided dword loads per window row: 64 sectors touched for 256 useful bytes each
        d[0] = row[max(gj0 - 1, 0)];
        d[9] = row[min(gj0 + 8, W - 1)];
      }
    }
  };
  // (vx, vy) of block row rr (-1 .. 7) from window rows `cu` (row rr) and `dn` (row rr + 1): columns m - 1 for m = 0 .. 8
  auto v_row = [&](const float (&cu)[MC ? 10 : 1], const float (&dn)[MC ? 10 : 1], int rr) {
    if constexpr (MC) {
      const int gi = gi0 + rr;
      const bool rowin = gi >= 0, down = gi + 1 < H, an = P.ncvx_gamma < 0.f;
      const float gth = fabsf(P.ncvx_gamma);
#pragma unroll
      for (int m = 0; m < 9; ++m) {
        const int gj = gj0 + m - 1;
        const bool colin = gj >= 0;
        const float dx = (rowin && colin && down) ? dn[m] - cu[m] : 0.f;
        const float dy = (rowin && colin && gj + 1 < W) ? cu[m + 1] - cu[m] : 0.f;
        if (an) {
          vxc[m] = __builtin_amdgcn_rcpf(fmaxf(fabsf(dx), gth)) * dx;
          vyc[m] = __builtin_amdgcn_rcpf(fmaxf(fabsf(dy), gth)) * dy;
        } else {
          const float w = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy)), gth));
          vxc[m] = w * dx;
          vyc[m] = w * dy;
        }
      }
    }
  };
  if constexpr (MC) {
    load_wrow(wr[0], -1); load_wrow(wr[1], 0);
    v_row(wr[0], wr[1], -1);
#pragma unroll
    for (int j = 0; j < 8; ++j) vxp[j] = vxc[j + 1];
  }
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    float nz[8][4];
    if (P.noise_mode == LMC_NOISE_PHILOX) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        quad_normals(P.key0, P.key1, P.iteration + (P.iter_dev ? *P.iter_dev : 0u) + (uint32_t)fit, P.chain_offset + chain, (uint32_t)(by * 2 + q) * (uint32_t)W + (uint32_t)(bx * 8 + j), nz[j]);
        if ((j & 1) == 1) __builtin_amdgcn_sched_barrier(0);   // two Philox calls in flight, not eight: their temporaries decide the occupancy
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int r = 4 * q + k;
      float xr[8], yv[8], mv[8], xi[8];
      {
        const float4 lo = PRIOR == LMC_PRIOR_HAAR_L1 ? xs[(2 * r) * 256 + threadIdx.x] : *reinterpret_cast<const float4*>(src + (size_t)r * W);
        const float4 hi = PRIOR == LMC_PRIOR_HAAR_L1 ? xs[(2 * r + 1) * 256 + threadIdx.x] : *reinterpret_cast<const float4*>(src + (size_t)r * W + 4);
        xr[0] = lo.x; xr[1] = lo.y; xr[2] = lo.z; xr[3] = lo.w; xr[4] = hi.x; xr[5] = hi.y; xr[6] = hi.z; xr[7] = hi.w;
      }
      if (DATA != LMC_DATA_NONE) {
        const float4 lo = *reinterpret_cast<const float4*>(P.y + o0 + (size_t)r * W);
        const float4 hi = *reinterpret_cast<const float4*>(P.y + o0 + (size_t)r * W + 4);
        yv[0] = lo.x; yv[1] = lo.y; yv[2] = lo.z; yv[3] = lo.w; yv[4] = hi.x; yv[5] = hi.y; yv[6] = hi.z; yv[7] = hi.w;
      }
      if (DATA == LMC_DATA_MASK) {
        const float4 lo = *reinterpret_cast<const float4*>(P.mask + o0 + (size_t)r * W);
        const float4 hi = *reinterpret_cast<const float4*>(P.mask + o0 + (size_t)r * W + 4);
        mv[0] = lo.x; mv[1] = lo.y; mv[2] = lo.z; mv[3] = lo.w; mv[4] = hi.x; mv[5] = hi.y; mv[6] = hi.z; mv[7] = hi.w;
      }
      if (P.noise_mode == LMC_NOISE_INJECTED) {
        const float* __restrict__ nzp = P.noise + (size_t)chain * img + o0 + (size_t)r * W;
        const float4 lo = *reinterpret_cast<const float4*>(nzp);
        const float4 hi = *reinterpret_cast<const float4*>(nzp + 4);
        xi[0] = lo.x; xi[1] = lo.y; xi[2] = lo.z; xi[3] = lo.w; xi[4] = hi.x; xi[5] = hi.y; xi[6] = hi.z; xi[7] = hi.w;
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) xi[j] = P.noise_mode == LMC_NOISE_PHILOX ? nz[j][k] : 0.f;
      }
      if constexpr (MC) {        // window rows r (slot (r + 1) & 1) and r + 1 (loaded into the slot row r - 1 has left)
        load_wrow(wr[r & 1], r + 1);
        v_row(wr[(r + 1) & 1], wr[r & 1], r);
      }
      float o[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = xr[j];
        float g = 0.f;                                            // grad f / sigma_f
        if (DATA == LMC_DATA_IDENTITY) g = x - yv[j];
        else if (DATA == LMC_DATA_MASK) g = mv[j] * fmaf(mv[j], x, -yv[j]);
        float px = x;
        if (PRIOR == LMC_PRIOR_HAAR_L1) px = v[r][j];
        else if (PRIOR == LMC_PRIOR_L2) px = x * P.prior_p0;
        else if (PRIOR == LMC_PRIOR_L1) px = soft_thr_b(x, P.prior_p0);
        else if (PRIOR == LMC_PRIOR_EPROX) px = eprox(P.eprox_kind, x, EproxParams{P.prior_p0, P.prior_p1});   // closed forms of prox.py
        o[j] = fmaf(P.b, px, fmaf(P.s, xi[j], fmaf(P.a, x, -ts * g)));
        if constexpr (MC) o[j] = fmaf(P.t * P.ncvx_lambda, -((vxc[j + 1] - vxp[j]) + (vyc[j + 1] - vyc[j])), o[j]);
      }
      if constexpr (MC) {
#pragma unroll
        for (int j = 0; j < 8; ++j) vxp[j] = vxc[j + 1];
      }
      if (last_it) {
        *reinterpret_cast<float4*>(dst + (size_t)r * W) = make_float4(o[0], o[1], o[2], o[3]);
        *reinterpret_cast<float4*>(dst + (size_t)r * W + 4) = make_float4(o[4], o[5], o[6], o[7]);
      } else {        // row r of the slab has been read (above): it takes the new iterate
        xs[(2 * r) * 256 + threadIdx.x] = make_float4(o[0], o[1], o[2], o[3]);
        xs[(2 * r + 1) * 256 + threadIdx.x] = make_float4(o[4], o[5], o[6], o[7]);
        if (fit == 0 && mid) {
          *reinterpret_cast<float4*>(mid + (size_t)r * W) = make_float4(o[0], o[1], o[2], o[3]);
          *reinterpret_cast<float4*>(mid + (size_t)r * W + 4) = make_float4(o[4], o[5], o[6], o[7]);
        }
      }
    }
  }
  };   // pass
  static_for<0, NF>([&](auto fc) { pass(fc); });
}

// two iterations per launch: the block-local combination only (Haar prior, stencil-free data term, no MC-TV term), Philox or no noise
bool block_pair_supported(const StepArgs& a) {
  return block_supported(a) && a.prior_kind == LMC_PRIOR_HAAR_L1 && a.ncvx_kind == LMC_NCVX_NONE &&
         (a.noise_mode == LMC_NOISE_PHILOX || a.noise_mode == LMC_NOISE_NONE);
}

bool block_supported(const StepArgs& a) {
  if ((a.H & 7) || (a.W & 7) || a.H < 8 || a.W < 8) return false;
  if (a.data_kind == LMC_DATA_BLUR || a.extra || a.prox_ext) return false;
  if (a.ncvx_kind == LMC_NCVX_MC_TV && a.prior_kind != LMC_PRIOR_HAAR_L1) return false;   // fused for the C5 combination only
  if (a.ncvx_kind != LMC_NCVX_NONE && a.ncvx_kind != LMC_NCVX_MC_TV) return false;
  if (a.prior_kind == LMC_PRIOR_TV_ISO || a.prior_kind == LMC_PRIOR_TV_ANISO) return false;
  if (a.tv_in || a.tv_out) return false;
  return true;
}

template <int DATA>
static void launch_block_data(const StepArgs& a, int nblk, hipStream_t st) {
  switch (a.prior_kind) {
    case LMC_PRIOR_L2: hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_L2>), dim3(nblk), dim3(256), 0, st, a); break;
    case LMC_PRIOR_L1: hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_L1>), dim3(nblk), dim3(256), 0, st, a); break;
    case LMC_PRIOR_EPROX: hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_EPROX>), dim3(nblk), dim3(256), 0, st, a); break;
    case LMC_PRIOR_HAAR_L1:
      if (a.ncvx_kind == LMC_NCVX_MC_TV) hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_HAAR_L1, true>), dim3(nblk), dim3(256), 0, st, a);
      else if (a.fused_iters == 2) hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_HAAR_L1, false, 2>), dim3(nblk), dim3(256), 0, st, a);
      else if (a.fused_iters == 4) hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_HAAR_L1, false, 4>), dim3(nblk), dim3(256), 0, st, a);
      else hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_HAAR_L1>), dim3(nblk), dim3(256), 0, st, a);
      break;
    default: hipLaunchKernelGGL((myula_step_block_kernel<DATA, LMC_PRIOR_NONE>), dim3(nblk), dim3(256), 0, st, a); break;
  }
}

hipError_t launch_step_block(const StepArgs& a, hipStream_t st) {
  if (!block_supported(a)) return hipErrorInvalidConfiguration;
  if (a.fused_iters != 0 && a.fused_iters != 1 && a.fused_iters != 2 && a.fused_iters != 4) return hipErrorInvalidConfiguration;
  if (a.fused_iters >= 2 && (!block_pair_supported(a) || (a.fused_iters == 4 && a.x_mid))) return hipErrorInvalidConfiguration;
  const size_t nb = (size_t)(a.H >> 3) * (a.W >> 3) * (size_t)a.C;
  const size_t nblk = (nb + 255) / 256;
  if (nblk > 0x7fffffffu) return hipErrorInvalidConfiguration;
  if (a.data_kind == LMC_DATA_MASK) launch_block_data<LMC_DATA_MASK>(a, (int)nblk, st);
  else if (a.data_kind == LMC_DATA_IDENTITY) launch_block_data<LMC_DATA_IDENTITY>(a, (int)nblk, st);
  else launch_block_data<LMC_DATA_NONE>(a, (int)nblk, st);
  return hipGetLastError();
}

}  // namespace lmc
