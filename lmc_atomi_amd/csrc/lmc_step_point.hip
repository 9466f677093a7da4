// Fused MYULA update for closed-form (pointwise) priors -- l2, l1 or none -- with a separable blur, pointwise or no
// data term: out = a*x - t*grad f(x) + b*prox(x) + s*xi   (algs.py:569).
//
// No TV pipeline here, so the update is a plain stencil and the kernel is built to be HBM bound: one workgroup
// (256 threads) = one 32 x 64 output tile of one chain; the tile plus a halo of KT-1 is staged in LDS once, the blur
// gradient sigma_f * H^T(Hx - y) runs as four 1-D passes between two LDS scratch images (the residual is masked to
// the image before H^T: adjoint of a zero-padded "same" convolution), then prox + Philox noise + store.
// HBM traffic: x read (1 + halo) times, x' written once; y comes from L2 (shared by all chains).
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

#ifndef LMC_PT_TW
#define LMC_PT_TW 64
#endif
constexpr int kPtTW = LMC_PT_TW, kPtTH = 32, kPtThreads = 4 * LMC_PT_TW;   // thread -> one column x 8 rows in the combine

template <int KT>   // taps zero-padded to KT (5 or 7); KT == 0: no blur (pointwise / no data term)
__global__ __launch_bounds__(kPtThreads) void myula_step_point_kernel(const StepArgs P) {
  constexpr int HL = KT > 0 ? KT - 1 : 1;             // halo (>= 1 so that the MC-TV stencil has its neighbours)
  constexpr int PW = kPtTW + 2 * HL, PH = kPtTH + 2 * HL;
  constexpr int CW = kPtTW + (KT > 0 ? KT - 1 : 0);   // columns of the intermediate passes
  constexpr int RH = kPtTH + (KT > 0 ? KT - 1 : 0);   // rows of the residual
  __shared__ float xs[PH * PW];
  __shared__ float sa[KT > 0 ? PH * PW : 1];
  __shared__ float sb[KT > 0 ? PH * PW : 1];

  const int tid = threadIdx.x;
  const int logical = xcd_logical_block(blockIdx.x, gridDim.x);
  const int tiles = P.tiles_x * P.tiles_y;
  const int chain = logical / tiles;
  const int tile = logical - chain * tiles;
  const int ty0 = (tile / P.tiles_x) * kPtTH, tx0 = (tile % P.tiles_x) * kPtTW;
  const int H = P.H, W = P.W;
  const size_t img = (size_t)H * W;
  const float* __restrict__ xin = P.x_in + (size_t)chain * img;
  float* __restrict__ xout = P.x_out + (size_t)chain * img;
  const int row0 = ty0 - HL, col0 = tx0 - HL;

  // stage x (zero outside the image).  Every loop below has a compile-time trip count and is fully unrolled so that
  // its loads are all in flight before the first wait (a rolled loop serialises one HBM / LDS latency per trip).
#pragma unroll
  for (int it = 0; it < (PH * PW + kPtThreads - 1) / kPtThreads; ++it) {
    const int e = tid + it * kPtThreads;
    if (e >= PH * PW) break;
    const int r = e / PW, c = e - r * PW;
    const int gr = row0 + r, gc = col0 + c;
    xs[e] = (gr >= 0 && gr < H && gc >= 0 && gc < W) ? xin[(size_t)gr * W + gc] : 0.f;
  }
  if constexpr (KT > 0) {   // stage the observation for the residual region into sb (pass 2 overwrites it in place)
    const int oy = P.blur.oy, ox = P.blur.ox;
#pragma unroll
    for (int it = 0; it < (RH * CW + kPtThreads - 1) / kPtThreads; ++it) {
      const int e = tid + it * kPtThreads;
      if (e >= RH * CW) break;
      const int rr = e / CW, r = HL - oy + rr, c = HL - ox + (e - rr * CW);
      const int gr = row0 + r, gc = col0 + c;
      sb[r * PW + c] = (gr >= 0 && gr < H && gc >= 0 && gc < W) ? P.y[(size_t)gr * W + gc] : 0.f;
    }
  }
  // pointwise data terms: the thread's 8 output pixels of y (and mask), fetched before the barrier
  const int tx = tid % kPtTW, tyq = tid / kPtTW;
  const int gc = tx0 + tx;
  float yv[8], mv[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { yv[j] = 0.f; mv[j] = 0.f; }
  if (KT == 0 && gc < W && (P.data_kind == LMC_DATA_IDENTITY || P.data_kind == LMC_DATA_MASK)) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int gr = ty0 + tyq * 8 + j;
      if (gr < H) {
        yv[j] = P.y[(size_t)gr * W + gc];
        if (P.data_kind == LMC_DATA_MASK) mv[j] = P.mask[(size_t)gr * W + gc];
      }
    }
  }
  __syncthreads();

  if constexpr (KT > 0) {
    const int oy = P.blur.oy, ox = P.blur.ox;
    const float* __restrict__ uv = P.blur.h;   // u[0..KT) then v[0..KT) at h[kMaxBlur..]
    // pass 1: hx[r][c] = sum_b v[b] x[r][c - b + ox], all rows, columns [HL-ox, HL-ox+CW)
#pragma unroll
    for (int it = 0; it < (PH * CW + kPtThreads - 1) / kPtThreads; ++it) {
      const int e = tid + it * kPtThreads;
      if (e >= PH * CW) break;
      const int r = e / CW, c = HL - ox + (e - r * CW);
      float acc = 0.f;
#pragma unroll
      for (int b = 0; b < KT; ++b) acc = fmaf(uv[kMaxBlur + b], xs[r * PW + c - b + ox], acc);
      sa[r * PW + c] = acc;
    }
    __syncthreads();
    // pass 2: R[r][c] = sum_a u[a] hx[r - a + oy][c] - y, rows [HL-oy, HL-oy+RH), zero outside the image
#pragma unroll
    for (int it = 0; it < (RH * CW + kPtThreads - 1) / kPtThreads; ++it) {
      const int e = tid + it * kPtThreads;
      if (e >= RH * CW) break;
      const int rr = e / CW, r = HL - oy + rr, c = HL - ox + (e - rr * CW);
      const int gr = row0 + r, gc = col0 + c;
      float acc = 0.f;
      if (gr >= 0 && gr < H && gc >= 0 && gc < W) {
#pragma unroll
        for (int a = 0; a < KT; ++a) acc = fmaf(uv[a], sa[(r - a + oy) * PW + c], acc);
        acc -= sb[r * PW + c];
      }
      sb[r * PW + c] = acc;
    }
    __syncthreads();
    // pass 3: hR[r][c] = sum_b v[b] R[r][c + b - ox], same rows, output columns [HL, HL+TW)
#pragma unroll
    for (int it = 0; it < (RH * kPtTW + kPtThreads - 1) / kPtThreads; ++it) {
      const int e = tid + it * kPtThreads;
      if (e >= RH * kPtTW) break;
      const int rr = e / kPtTW, r = HL - oy + rr, c = HL + (e - rr * kPtTW);
      float acc = 0.f;
#pragma unroll
      for (int b = 0; b < KT; ++b) acc = fmaf(uv[kMaxBlur + b], sb[r * PW + c + b - ox], acc);
      sa[r * PW + c] = acc;
    }
    __syncthreads();
  }

  // pass 4 + combine: thread -> column tx, rows 8*ty .. 8*ty+7 (two Philox quads)
  if (gc >= W) return;
#pragma unroll
  for (int qd = 0; qd < 2; ++qd) {
    const int lr0 = tyq * 8 + qd * 4;          // tile-local output row of the quad
    const int gr0 = ty0 + lr0;
    if (gr0 >= H) break;
    float xi[4] = {0.f, 0.f, 0.f, 0.f};
    if (P.noise_mode == LMC_NOISE_PHILOX)
      quad_normals(P.key0, P.key1, P.iteration, P.chain_offset + (uint32_t)chain, (uint32_t)(gr0 >> 2) * (uint32_t)W + (uint32_t)gc, xi);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gr = gr0 + j;
      if (gr >= H) break;
      const int r = HL + lr0 + j, c = HL + tx;
      const size_t gi = (size_t)gr * W + gc;
      const float x = xs[r * PW + c];
      float g = 0.f;
      if constexpr (KT > 0) {
        const int oy = P.blur.oy;
        float acc = 0.f;
#pragma unroll
        for (int a = 0; a < KT; ++a) acc = fmaf(P.blur.h[a], sa[(r + a - oy) * PW + c], acc);
        g = P.sigma_f * acc;
      } else if (P.data_kind == LMC_DATA_IDENTITY) {
        g = P.sigma_f * (x - yv[qd * 4 + j]);
      } else if (P.data_kind == LMC_DATA_MASK) {
        const float mk = mv[qd * 4 + j];
        g = P.sigma_f * mk * fmaf(mk, x, -yv[qd * 4 + j]);
      }
      if (P.ncvx_kind == LMC_NCVX_MC_TV)
        g -= P.ncvx_lambda * mc_tv_grad(xs[(r - 1) * PW + c], xs[(r - 1) * PW + c + 1], xs[r * PW + c - 1], x, xs[r * PW + c + 1],
                                        xs[(r + 1) * PW + c - 1], xs[(r + 1) * PW + c], gr > 0, gr + 1 < H, gc > 0, gc + 1 < W,
                                        P.ncvx_gamma);
      if (P.extra) g = fmaf(P.extra_coef, x - P.extra[(size_t)chain * img + gi], g);
      float px = x;
      if (P.prior_kind == LMC_PRIOR_L2) px = x * P.prior_p0;
      else if (P.prior_kind == LMC_PRIOR_L1) px = copysignf(fmaxf(fabsf(x) - P.prior_p0, 0.f), x);
      else if (P.prior_kind == LMC_PRIOR_EPROX) px = eprox(P.eprox_kind, x, EproxParams{P.prior_p0, P.prior_p1});
      if (P.prox_ext) px = P.prox_ext[(size_t)chain * img + gi];
      float nz = xi[j];
      if (P.noise_mode == LMC_NOISE_INJECTED) nz = P.noise[(size_t)chain * img + gi];
      xout[gi] = fmaf(P.a, x, fmaf(-P.t, g, fmaf(P.b, px, P.s * nz)));
    }
  }
}

bool separate_blur_taps(const BlurTaps& T, float* u, float* v);  // lmc_step_rows.hip

bool point_supported(const StepArgs& a) {
  if (a.prior_kind == LMC_PRIOR_TV_ISO) return false;
  if (a.data_kind == LMC_DATA_BLUR) {
    if (a.blur.kh > 7 || a.blur.kw > 7) return false;
    float u[kMaxBlur], v[kMaxBlur];
    if (!separate_blur_taps(a.blur, u, v)) return false;
  }
  return true;
}

hipError_t launch_step_point(StepArgs a, hipStream_t st) {
  int KT = 0;
  if (a.data_kind == LMC_DATA_BLUR) {
    float u[kMaxBlur] = {0}, v[kMaxBlur] = {0};
    if (!separate_blur_taps(a.blur, u, v)) return hipErrorInvalidConfiguration;
    for (int i = 0; i < kMaxBlur; ++i) { a.blur.h[i] = u[i]; a.blur.h[kMaxBlur + i] = v[i]; }
    KT = (a.blur.kh > a.blur.kw ? a.blur.kh : a.blur.kw) <= 5 ? 5 : 7;
  }
  a.tiles_x = (a.W + kPtTW - 1) / kPtTW;
  a.tiles_y = (a.H + kPtTH - 1) / kPtTH;
  const int nblk = a.tiles_x * a.tiles_y * a.C;
  if (KT == 0) hipLaunchKernelGGL(myula_step_point_kernel<0>, dim3(nblk), dim3(kPtThreads), 0, st, a);
  else if (KT == 5) hipLaunchKernelGGL(myula_step_point_kernel<5>, dim3(nblk), dim3(kPtThreads), 0, st, a);
  else hipLaunchKernelGGL(myula_step_point_kernel<7>, dim3(nblk), dim3(kPtThreads), 0, st, a);
  return hipGetLastError();
}

}  // namespace lmc
