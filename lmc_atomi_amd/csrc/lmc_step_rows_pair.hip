// TWO MYULA iterations per launch for a separable uniform-box blur and a closed-form prior (l2 / l1 / none):
//   x_{k+1} = a x_k     - t sigma H^T(H x_k     - y) + b prox(x_k)     + s xi_k
//   x_{k+2} = a x_{k+1} - t sigma H^T(H x_{k+1} - y) + b prox(x_{k+1}) + s xi_{k+1}          (algs.py:569 twice)
// -- one read of x_k and one write of x_{k+2} per TWO iterations (plus x_{k+1} when the posterior moments want it): the single-iteration kernel
// (lmc_step_rows.hip) is bound by its instruction stream at two waves per SIMD, not by HBM, so the iteration pair costs what two waves of work cost
// and the memory system sees half the traffic.  Structure = lmc_cheb_pair.hip: a workgroup is a PAIR of waves on one band of rows of one chain;
// wave 0 (stage 0) streams x_k rows from global memory and publishes x_{k+1} rows in an 8-row LDS ring, wave 1 (stage 1) trails by 8 rows, takes
// its input rows from the ring, keeps them in its register ring for the combine, and stores x_{k+2}.  One barrier every 4 rows; stage 0's band is 8
// rows longer at either end (recomputed: the halo rows of x_{k+1} are the same numbers the neighbouring band computes -- Philox counters are
// global).  Both stages draw their own noise field (iteration k and k + 1).  The same arithmetic as two launches of the single-iteration kernel;
// equal to them to fp32 rounding (the running window sums start at band boundaries, and the bands differ).
#include "lmc_device.h"
#include "lmc_launch.h"

#include <cmath>
#include <cstdlib>

namespace lmc {

template <int PXL>
__device__ __forceinline__ void rpair_gload(float (&dst)[PXL], const float* __restrict__ row, int c0, int W) {
#pragma unroll
  for (int g = 0; g < PXL / 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(row + (c0 + 4 * g < W ? c0 + 4 * g : 0));
    dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
  }
}

template <int PXL, int STAGE>
__device__ __forceinline__ void rows_pair_body(const StepArgs& P, float* __restrict__ x_mid, const int band_rows, const int nbands,
                                               float* __restrict__ ringU, float* __restrict__ nzw) {
  constexpr int HW = 2, LAG = 4, PF = 4, ULO = 0, UHI = 4;
  constexpr int stage = STAGE;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x;
  const int chain = gw / nbands, band = gw - chain * nbands;
  const int H = P.H, W = P.W;
  const int r0 = band * band_rows, r1 = min(r0 + band_rows, H);            // rows this pair writes; r0 % 8 == 0
  const int e0 = max(r0 - 8, 0), e1 = min(r1 + 8, H);                      // rows stage 0 produces
  const int sr0 = stage ? r0 : r0 - 8;                                     // first row of this stage's band
  const int o_lo = stage ? r0 : e0, o_hi = stage ? r1 : e1;                // rows this stage emits
  const int c0 = lane * PXL;
  auto colok = [&](int k) { return c0 + (k & ~3) < W; };
  const size_t img = (size_t)H * W;
  const float* __restrict__ xin = P.x_in + (size_t)chain * img;
  float* __restrict__ xmid = x_mid ? x_mid + (size_t)chain * img : nullptr;
  float* __restrict__ xout = P.x_out + (size_t)chain * img;
  const float cbox = P.blur.h[0] * P.blur.h[kMaxBlur];                     // c_u c_v of the uniform 5-tap box
  const uint32_t iter = P.iteration + (P.iter_dev ? *P.iter_dev : 0u) + (uint32_t)stage;
  float* const myU = ringU + lane * PXL;

  float xr[8][PXL], A[8][PXL], G[8][PXL], Vs[PXL], Ws[PXL], yq[2][PXL];
#pragma unroll
  for (int k = 0; k < PXL; ++k) { Vs[k] = Ws[k] = 0.f; yq[0][k] = yq[1][k] = 0.f; }
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int k = 0; k < PXL; ++k) { xr[s][k] = 0.f; A[s][k] = 0.f; G[s][k] = 0.f; }
#pragma unroll
  for (int k = 0; k < PXL * 4; ++k) nzw[k * 64] = 0.f;                     // the 4 PXL normals of a quad row-group (wave-private slab)

  const int base_first = r0 - 16;
  if constexpr (stage == 0) {          // prime the x ring: rows base_first .. base_first + PF - 1 (slot = row & 7)
    static_for<0, PF>([&](auto pp) {
      constexpr int p = decltype(pp)::value;
      rpair_gload<PXL>(xr[p], xin + (size_t)min(max(base_first + p, 0), H - 1) * W, c0, W);
    });
  }

  // One step: stage 0 takes input row base + J, stage 1 input row base + J - 8 (both have slot J).
  auto step = [&](auto jj, const int base) __attribute__((always_inline)) {
    constexpr int J = decltype(jj)::value;
    const int i = base + J - (stage ? 8 : 0);
    // (0) the observation row of the NEXT step's residual row (slot = step parity), requested first
    rpair_gload<PXL>(yq[(J + 1) & 1], P.y + (size_t)min(max(i + 1 - HW, 0), H - 1) * W, c0, W);
    // (1) input row: zero outside the image (and, stage 1, outside what stage 0 published)
    float xm[PXL];
    if constexpr (stage == 0) {
      const bool rowin = i >= 0 && i < H;
#pragma unroll
      for (int k = 0; k < PXL; ++k) xm[k] = (rowin && colok(k)) ? xr[J][k] : 0.f;
    } else {
      const bool rowin = i >= e0 && i < e1;
#pragma unroll
      for (int g = 0; g < PXL / 4; ++g) {
        const float4 u = *reinterpret_cast<const float4*>(myU + J * 64 * PXL + 4 * g);
        const bool ok = rowin && c0 + 4 * g < W;
        xm[4 * g] = ok ? u.x : 0.f; xm[4 * g + 1] = ok ? u.y : 0.f; xm[4 * g + 2] = ok ? u.z : 0.f; xm[4 * g + 3] = ok ? u.w : 0.f;
        xr[J][4 * g] = u.x; xr[J][4 * g + 1] = u.y; xr[J][4 * g + 2] = u.z; xr[J][4 * g + 3] = u.w;
      }
    }
    // (2) horizontal window sums of the input row -> ring A; vertical running sum Vs = rows i - UHI .. i - ULO; residual row r = i - HW
    float hx[PXL];
    {
      float e[PXL + 2 * HW];
#pragma unroll
      for (int m = 0; m < HW; ++m) e[m] = dpp_left0(xm[PXL - HW + m]);
#pragma unroll
      for (int k = 0; k < PXL; ++k) e[HW + k] = xm[k];
#pragma unroll
      for (int m = 0; m < HW; ++m) e[HW + PXL + m] = dpp_right0(xm[m]);
      float acc = e[2 * HW - UHI];
#pragma unroll
      for (int j = 2 * HW - UHI + 1; j <= 2 * HW - ULO; ++j) acc += e[j];
      hx[0] = acc;
#pragma unroll
      for (int k = 1; k < PXL; ++k) hx[k] = (hx[k - 1] + e[k + 2 * HW - ULO]) - e[k - 1 + 2 * HW - UHI];
    }
#pragma unroll
    for (int k = 0; k < PXL; ++k) A[J][k] = hx[k];
    if constexpr (J == 0) {
#pragma unroll
      for (int k = 0; k < PXL; ++k) {
        float acc = A[(J - ULO + 8) & 7][k];
        static_for<ULO + 1, UHI + 1>([&](auto aa) { acc += A[(J - decltype(aa)::value + 16) & 7][k]; });
        Vs[k] = acc;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PXL; ++k) Vs[k] = (Vs[k] + A[(J - ULO + 8) & 7][k]) - A[(J - 1 - UHI + 16) & 7][k];
    }
    const int r = i - HW;
    float R[PXL];
    {
      const bool rowok = r >= 0 && r < H && r >= sr0 - HW;
#pragma unroll
      for (int k = 0; k < PXL; ++k) R[k] = (rowok && colok(k)) ? fmaf(cbox, Vs[k], -yq[J & 1][k]) : 0.f;
    }
    // (3) the same for the residual row -> ring G, running sum Ws = gradient row o = i - LAG (without sigma c_u c_v)
    float hr[PXL];
    {
      float e[PXL + 2 * HW];
#pragma unroll
      for (int m = 0; m < HW; ++m) e[m] = dpp_left0(R[PXL - HW + m]);
#pragma unroll
      for (int k = 0; k < PXL; ++k) e[HW + k] = R[k];
#pragma unroll
      for (int m = 0; m < HW; ++m) e[HW + PXL + m] = dpp_right0(R[m]);
      float acc = e[ULO];
#pragma unroll
      for (int j = ULO + 1; j <= UHI; ++j) acc += e[j];
      hr[0] = acc;
#pragma unroll
      for (int k = 1; k < PXL; ++k) hr[k] = (hr[k - 1] + e[k + UHI]) - e[k - 1 + ULO];
    }
    constexpr int sR = (J - HW + 8) & 7;
#pragma unroll
    for (int k = 0; k < PXL; ++k) G[sR][k] = hr[k];
    constexpr int sNew = (J - LAG - HW + UHI + 32) & 7, sOld = (J - LAG - 1 - HW + ULO + 32) & 7;
    if constexpr (J == 0) {
#pragma unroll
      for (int k = 0; k < PXL; ++k) {
        float acc = G[(J - LAG - HW + ULO + 32) & 7][k];
        static_for<ULO + 1, UHI + 1>([&](auto aa) { acc += G[(J - LAG - HW + decltype(aa)::value + 32) & 7][k]; });
        Ws[k] = acc;
      }
    } else {
#pragma unroll
      for (int k = 0; k < PXL; ++k) Ws[k] = (Ws[k] + G[sNew][k]) - G[sOld][k];
    }
    // (4) output row o = i - LAG: combine with x[o] (register ring), prior prox, this stage's Philox field
    const int o = i - LAG;
    constexpr int so = (J - LAG + 8) & 7;
    if (o >= o_lo && o < o_hi) {
      if constexpr (((J - LAG + 8) & 3) == 0) {       // first row of a Philox quad (bands and their halos start on multiples of 8)
        if (P.noise_mode == LMC_NOISE_PHILOX) {
#pragma unroll
          for (int k = 0; k < PXL; ++k) {
            float n4[4];
            quad_normals(P.key0, P.key1, iter, P.chain_offset + (uint32_t)chain, (uint32_t)(o >> 2) * (uint32_t)W + (uint32_t)(c0 + k), n4);
#pragma unroll
            for (int q = 0; q < 4; ++q) nzw[(q * PXL + k) * 64] = n4[q];
          }
        }
      }
      constexpr int jq = (J - LAG + 8) & 3;
      const size_t go = (size_t)o * W;
      const bool mine = o >= r0 && o < r1;
#pragma unroll
      for (int g = 0; g < PXL / 4; ++g) {
        if (c0 + 4 * g < W) {
          float ov[4];
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float xi = nzw[(jq * PXL + 4 * g + q) * 64];
            const float x = xr[so][4 * g + q];
            const float gr = (P.sigma_f * cbox) * Ws[4 * g + q];
            float px = x;
            if (P.prior_kind == LMC_PRIOR_L2) px = x * P.prior_p0;
            else if (P.prior_kind == LMC_PRIOR_L1) px = copysignf(fmaxf(fabsf(x) - P.prior_p0, 0.f), x);
            ov[q] = fmaf(P.a, x, fmaf(-P.t, gr, fmaf(P.b, px, P.s * xi)));
          }
          if constexpr (stage == 0) {
            *reinterpret_cast<float4*>(myU + so * 64 * PXL + 4 * g) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            if (mine && xmid) *reinterpret_cast<float4*>(xmid + go + c0 + 4 * g) = make_float4(ov[0], ov[1], ov[2], ov[3]);
          } else {
            *reinterpret_cast<float4*>(xout + go + c0 + 4 * g) = make_float4(ov[0], ov[1], ov[2], ov[3]);
          }
        }
      }
    }
    // (5) stage 0: fetch x row i + PF into the slot row i + PF - 8 has left
    if constexpr (stage == 0) rpair_gload<PXL>(xr[(J + PF) & 7], xin + (size_t)min(max(i + PF, 0), H - 1) * W, c0, W);
  };
  const int r1r = (r1 + 7) & ~7;
  for (int base = base_first; base < r1r + 16; base += 8) {
    static_for<0, 4>([&](auto jj) { step(jj, base); });
    __syncthreads();
    static_for<4, 8>([&](auto jj) { step(jj, base); });
    __syncthreads();
  }
}

template <int PXL>
__global__ __launch_bounds__(128, 2) void myula_step_rows_pair_kernel(const StepArgs P, float* x_mid, const int band_rows, const int nbands) {
  __shared__ float ringU[8 * 64 * PXL];
  __shared__ float nz_lds[2 * PXL * 4 * 64];
  float* const nzw = nz_lds + (threadIdx.x >> 6) * PXL * 4 * 64 + (threadIdx.x & 63);
  // one instantiation per wave: each keeps its own register rings, and both pass the same sequence of barriers
  if ((threadIdx.x >> 6) == 0) rows_pair_body<PXL, 0>(P, x_mid, band_rows, nbands, ringU, nzw);
  else rows_pair_body<PXL, 1>(P, x_mid, band_rows, nbands, ringU, nzw);
}

// Covers: blur data term with the uniform 5-tap box, closed-form prior (l2 / l1 / none), Philox or no noise, 16-byte aligned rows, one wave per
// row (W <= 512); pays when the launch fills the chip with bands of >= 128 rows.
bool rows_pair_supported(const StepArgs& a) {
  if (a.data_kind != LMC_DATA_BLUR || a.ncvx_kind != LMC_NCVX_NONE) return false;
  if (a.prior_kind != LMC_PRIOR_NONE && a.prior_kind != LMC_PRIOR_L2 && a.prior_kind != LMC_PRIOR_L1) return false;
  if (a.tv_in || a.tv_out || a.prox_ext || a.extra || a.dot_out || a.run_count || a.f_out || a.g_out) return false;
  if (a.noise_mode != LMC_NOISE_PHILOX && a.noise_mode != LMC_NOISE_NONE) return false;
  if ((a.W & 3) || a.W < 4 || a.W > 512 || a.H < 1) return false;
  float uc[kMaxBlur] = {0}, vc[kMaxBlur] = {0};
  if (centred_blur_taps(a, uc, vc) != 5) return false;
  for (int i = 0; i < 5; ++i) {
    if (uc[i] == 0.f || vc[i] == 0.f) return false;
    if (std::fabs(uc[i] - uc[0]) > 1e-6f * std::fabs(uc[0]) || std::fabs(vc[i] - vc[0]) > 1e-6f * std::fabs(vc[0])) return false;
  }
  return true;
}

// x_out <- two MYULA iterations from x_in (iterations a.iteration and a.iteration + 1); x_mid (may be NULL) <- the iterate in between.
// x_in, x_mid, x_out: three different arrays (neighbouring bands re-read x_in rows of each other).
hipError_t launch_step_rows_pair(StepArgs a, float* x_mid, hipStream_t st) {
  if (!rows_pair_supported(a) || a.x_in == a.x_out || a.x_in == x_mid || a.x_out == x_mid) return hipErrorInvalidConfiguration;
  float uc[kMaxBlur] = {0}, vc[kMaxBlur] = {0};
  const int KT = centred_blur_taps(a, uc, vc);
  for (int i = 0; i < kMaxBlur; ++i) { a.blur.h[i] = i < KT ? uc[i] : 0.f; a.blur.h[kMaxBlur + i] = i < KT ? vc[i] : 0.f; }
  const char* eb = getenv("LMC_PAIR_BAND");       // read per launch (tests)
  const int env_band = eb ? atoi(eb) : 0;
  int nb = (1024 + a.C - 1) / a.C;
  if (nb < 1) nb = 1;
  int band = env_band > 0 ? env_band : (a.H + nb - 1) / nb;
  if (env_band <= 0 && band < 128) band = 128;
  if (band < 32) band = 32;
  band = (band + 7) & ~7;
  const int nbands = (a.H + band - 1) / band;
  const long long wgs = (long long)a.C * nbands;
  if (wgs > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  if (a.W <= 256) hipLaunchKernelGGL(myula_step_rows_pair_kernel<4>, dim3((unsigned)wgs), dim3(128), 0, st, a, x_mid, band, nbands);
  else hipLaunchKernelGGL(myula_step_rows_pair_kernel<8>, dim3((unsigned)wgs), dim3(128), 0, st, a, x_mid, band, nbands);
  return hipGetLastError();
}

}  // namespace lmc
