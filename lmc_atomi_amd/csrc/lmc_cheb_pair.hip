// Two Chebyshev iterations of the implicit data step (I + ts H^T H) u = rhs (algs.py:250, the solve inside L2.prox / ULPDA algs.py:439,445)
// in ONE launch -- 20 instead of 32 bytes per pixel:
//
//   u_{k+1} = a0 u_k     - t0 sigma H^T H u_k     + b0 rhs + s0 u_{k-1}          (stage 0)
//   u_{k+2} = a1 u_{k+1} - t1 sigma H^T H u_{k+1} + b1 rhs + s1 u_k              (stage 1)
//
// A workgroup is a PAIR of waves working on the same band of rows of one chain, each a row-streaming pipeline of the kind of
// lmc_step_rows.hip (uniform 5 x 5 box, sliding-window form: rings of horizontally filtered rows, running vertical sums):
//   wave 0 (stage 0) reads u_k rows from global memory (prefetched), and for every output row o also rhs[o] and u_{k-1}[o]; it writes
//          u_{k+1}[o] to global memory (the next pair's u_{k-1}) and publishes in LDS  U[o] = u_{k+1}[o]  and  E[o] = a1 u_{k+1}[o] + b1 rhs[o]
//          + s1 u_k[o]  -- everything of stage 1's update that is pointwise;
//   wave 1 (stage 1) trails by 8 rows, takes U rows from LDS as its input rows and keeps E rows in its register ring until its own
//          output row o is complete: u_{k+2}[o] = E[o] - t1 sigma (H^T H U)[o]; it touches global memory only to store.
// Hand-off: an 8-row ring per field, one barrier every 4 rows: in the 4 steps [b, b+4) stage 0 publishes rows [b-4, b) while stage 1 reads
// rows [b-8, b-4) -- opposite halves of the ring.  Both waves run the same number of steps (the block loop below), every step unconditional;
// rows a stage does not need are masked where they are used.  Stage 0's band is 8 rows longer at either end (recomputed, not exchanged).
// Outputs go to buffers that no workgroup reads in this launch (neighbouring bands re-read u_k / u_{k-1} / rhs rows of each other).
#include "lmc_device.h"
#include "lmc_launch.h"

#include <cmath>
#include <cstdlib>

namespace lmc {

// lanes past the row read its start (callers pass a clamped, valid row); masked where the value is used
template <int PXL>
__device__ __forceinline__ void pair_gload(float (&dst)[PXL], const float* __restrict__ row, int c0, int W) {
#pragma unroll
  for (int g = 0; g < PXL / 4; ++g) {
    const float4 v = *reinterpret_cast<const float4*>(row + (c0 + 4 * g < W ? c0 + 4 * g : 0));
    dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
  }
}

template <int PXL, int STAGE, bool DOT>
__device__ __forceinline__ void cheb_pair_body(const ChebPairArgs& P, const int band_rows, const int nbands, const bool last, float* __restrict__ ringU,
                                               float* __restrict__ ringE) {
  constexpr int HW = 2, LAG = 4, PF = 4, ULO = 0, UHI = 4;
  constexpr int stage = STAGE;
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x;
  const int chain = gw / nbands, band = gw - chain * nbands;
  const int H = P.H, W = P.W;
  const int r0 = band * band_rows, r1 = min(r0 + band_rows, H);            // rows this pair writes; r0 % 8 == 0
  const int e0 = max(r0 - 8, 0), e1 = min(r1 + 8, H);                      // rows stage 0 publishes
  const int sr0 = stage ? r0 : r0 - 8;                                     // first row of this stage's band
  const int c0 = lane * PXL;
  auto colok = [&](int k) { return c0 + (k & ~3) < W; };
  const size_t img = (size_t)H * W;
  const float* __restrict__ xin = P.cur + (size_t)chain * img;
  const float* __restrict__ prv = P.prv + (size_t)chain * img;
  const float* __restrict__ rhs = P.rhs + (size_t)chain * img;
  float* __restrict__ out1 = P.f1 + (size_t)chain * img;
  float* __restrict__ out2 = (last ? P.f2_last : P.f2) + (size_t)chain * img;
  const float cbox = P.cbox;

  float* const myU = ringU + lane * PXL;
  float* const myE = ringE + lane * PXL;

  float xr[8][PXL], A[8][PXL], G[8][PXL], Vs[PXL], Ws[PXL];
  // stage 0: rhs / u_{k-1} rows of the NEXT step's output row, requested at the start of a step (slot = row & 1): two steps of arithmetic
  // between request and use
  float rq[stage == 0 ? 2 : 1][PXL], pq[stage == 0 ? 2 : 1][PXL];
#pragma unroll
  for (int k = 0; k < PXL; ++k) { rq[0][k] = 0.f; pq[0][k] = 0.f; if constexpr (stage == 0) { rq[1][k] = 0.f; pq[1][k] = 0.f; } }
  const bool has_prv = P.s0 != 0.f;
#pragma unroll
  for (int k = 0; k < PXL; ++k) Vs[k] = Ws[k] = 0.f;
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int k = 0; k < PXL; ++k) { xr[s][k] = 0.f; A[s][k] = 0.f; G[s][k] = 0.f; }

  const int base_first = r0 - 16;
  if constexpr (stage == 0) {          // prime the x ring: rows base_first .. base_first + PF - 1 (slot = row & 7)
    static_for<0, PF>([&](auto pp) {
      constexpr int p = decltype(pp)::value;
      pair_gload<PXL>(xr[p], xin + (size_t)min(max(base_first + p, 0), H - 1) * W, c0, W);
    });
  }

  double dacc = 0.0, dacc2 = 0.0;       // DOT (stage 0 of a solve's first pair): sum (u_{k+1} - u_k)^2 and sum rhs^2 over the rows this pair writes

  // One step: stage 0 takes input row base + J, stage 1 input row base + J - 8 (both have slot J).
  auto step = [&](auto jj, const int base) __attribute__((always_inline)) {
    constexpr int J = decltype(jj)::value;
    const int i = base + J - (stage ? 8 : 0);
    if constexpr (stage == 0) {
      const size_t gn = (size_t)min(max(i + 1 - LAG, 0), H - 1) * W;
      pair_gload<PXL>(rq[(J + 1) & 1], rhs + gn, c0, W);
      if (has_prv) pair_gload<PXL>(pq[(J + 1) & 1], prv + gn, c0, W);       // iteration 0 has no u_{-1} (s0 = 0): pq stays zero
    }
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
        const float4 e = *reinterpret_cast<const float4*>(myE + J * 64 * PXL + 4 * g);
        const bool ok = rowin && c0 + 4 * g < W;
        xm[4 * g] = ok ? u.x : 0.f; xm[4 * g + 1] = ok ? u.y : 0.f; xm[4 * g + 2] = ok ? u.z : 0.f; xm[4 * g + 3] = ok ? u.w : 0.f;
        xr[J][4 * g] = e.x; xr[J][4 * g + 1] = e.y; xr[J][4 * g + 2] = e.z; xr[J][4 * g + 3] = e.w;
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
    if constexpr (J == 0) {          // re-formed directly every 8th row (and at the band start, where the ring below it is not valid yet)
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
      for (int k = 0; k < PXL; ++k) R[k] = (rowok && colok(k)) ? cbox * Vs[k] : 0.f;
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
    // (4) output row o = i - LAG
    const int o = i - LAG;
    constexpr int so = (J - LAG + 8) & 7;
    if constexpr (stage == 0) {
      if (o >= e0 && o < e1) {
        const size_t go = (size_t)o * W;
        const bool mine = o >= r0 && o < r1;
#pragma unroll
        for (int g = 0; g < PXL / 4; ++g) {
          if (c0 + 4 * g < W) {
            const float rr[4] = {rq[J & 1][4 * g], rq[J & 1][4 * g + 1], rq[J & 1][4 * g + 2], rq[J & 1][4 * g + 3]};
            const float pp[4] = {pq[J & 1][4 * g], pq[J & 1][4 * g + 1], pq[J & 1][4 * g + 2], pq[J & 1][4 * g + 3]};
            float u1[4], ee[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x = xr[so][4 * g + q];
              u1[q] = fmaf(P.a0, x, fmaf(-P.tg0, Ws[4 * g + q], fmaf(P.b0, rr[q], P.s0 * pp[q])));
              ee[q] = fmaf(P.a1, u1[q], fmaf(P.b1, rr[q], P.s1 * x));
              if constexpr (DOT) {
                if (mine) {
                  const double d = (double)u1[q] - (double)x;
                  dacc = fma(d, d, dacc);
                  dacc2 = fma((double)rr[q], (double)rr[q], dacc2);
                }
              }
            }
            *reinterpret_cast<float4*>(myU + so * 64 * PXL + 4 * g) = make_float4(u1[0], u1[1], u1[2], u1[3]);
            *reinterpret_cast<float4*>(myE + so * 64 * PXL + 4 * g) = make_float4(ee[0], ee[1], ee[2], ee[3]);
            if (mine) *reinterpret_cast<float4*>(out1 + go + c0 + 4 * g) = make_float4(u1[0], u1[1], u1[2], u1[3]);
          }
        }
      }
      // (5) fetch x row i + PF into the slot row i + PF - 8 has left
      pair_gload<PXL>(xr[(J + PF) & 7], xin + (size_t)min(max(i + PF, 0), H - 1) * W, c0, W);
    } else {
      if (o >= r0 && o < r1) {
        const size_t go = (size_t)o * W;
#pragma unroll
        for (int g = 0; g < PXL / 4; ++g) {
          if (c0 + 4 * g < W) {
            float u2[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u2[q] = fmaf(-P.tg1, Ws[4 * g + q], xr[so][4 * g + q]);
            *reinterpret_cast<float4*>(out2 + go + c0 + 4 * g) = make_float4(u2[0], u2[1], u2[2], u2[3]);
          }
        }
      }
    }
  };
  const int r1r = (r1 + 7) & ~7;
  for (int base = base_first; base < r1r + 16; base += 8) {
    static_for<0, 4>([&](auto jj) { step(jj, base); });
    __syncthreads();
    static_for<4, 8>([&](auto jj) { step(jj, base); });
    __syncthreads();
  }
  if constexpr (DOT && stage == 0) {
    const double t1 = wave_sum(dacc), t2 = wave_sum(dacc2);
    if (lane == 0) { unsafeAtomicAdd(&P.dot_out[2 * chain], t1); unsafeAtomicAdd(&P.dot_out[2 * chain + 1], t2); }
  }
}


template <int PXL, bool DOT>
__global__ __launch_bounds__(128, 2) void cheb_pair_kernel(const ChebPairArgs P, const int band_rows, const int nbands) {
  if (P.run_count && *P.run_count <= P.run_index) return;      // iterations the warm-started solve turned out not to need (uniform)
  const bool last = P.force_last || (P.run_count && *P.run_count <= P.last_index);
  __shared__ float ringU[8 * 64 * PXL], ringE[8 * 64 * PXL];
  // one instantiation per wave: each keeps its own register rings, and both pass the same sequence of barriers
  if ((threadIdx.x >> 6) == 0) cheb_pair_body<PXL, 0, DOT>(P, band_rows, nbands, last, ringU, ringE);
  else cheb_pair_body<PXL, 1, false>(P, band_rows, nbands, last, ringU, ringE);
}

// The pair kernel covers: the uniform 5-tap box (the reference's 5 x 5 blur, window [0, 4] of the centred taps), 16-byte aligned rows, one wave
// per row (W <= 512).  Everything else runs the single-iteration launches of the row-streaming kernel.
// Worth it when the launch fills the chip with bands of >= 128 rows (C H >= 2^17 rows); below that the single-iteration launches are
// latency-bound and shorter.
bool cheb_pair_pays(int64_t C, int H) { return C * (int64_t)H >= (1 << 17); }

bool cheb_pair_supported(int H, int W, const BlurTaps& taps) {
  if ((W & 3) || W < 4 || W > 512 || H < 1) return false;
  float uc[kMaxBlur] = {0}, vc[kMaxBlur] = {0};
  if (centred_blur_taps(taps, uc, vc) != 5) return false;
  for (int i = 0; i < 5; ++i) {
    if (uc[i] == 0.f || vc[i] == 0.f) return false;
    if (std::fabs(uc[i] - uc[0]) > 1e-6f * std::fabs(uc[0]) || std::fabs(vc[i] - vc[0]) > 1e-6f * std::fabs(vc[0])) return false;
  }
  return true;
}

hipError_t launch_cheb_pair(ChebPairArgs a, const BlurTaps& taps, hipStream_t st) {
  if (!cheb_pair_supported(a.H, a.W, taps)) return hipErrorInvalidConfiguration;
  float uc[kMaxBlur] = {0}, vc[kMaxBlur] = {0};
  centred_blur_taps(taps, uc, vc);
  a.cbox = uc[0] * vc[0];
  a.tg0 *= a.cbox; a.tg1 *= a.cbox;
  // two waves per band: half as many bands as the single-stage kernel for the same number of waves in flight (~4 per SIMD over the launch)
  // One workgroup (wave pair) per band; a pair runs band + 32 steps, so bands are long: about one round of workgroups (4 per CU) over the launch,
  // at least 128 rows each (LMC_PAIR_BAND overrides).
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
  if (a.dot_out) {     // the first pair of a solve: residual statistics of its first iteration (chebyshev_solve's adaptive count)
    if (a.W <= 256) hipLaunchKernelGGL((cheb_pair_kernel<4, true>), dim3((unsigned)wgs), dim3(128), 0, st, a, band, nbands);
    else hipLaunchKernelGGL((cheb_pair_kernel<8, true>), dim3((unsigned)wgs), dim3(128), 0, st, a, band, nbands);
  } else if (a.W <= 256) hipLaunchKernelGGL((cheb_pair_kernel<4, false>), dim3((unsigned)wgs), dim3(128), 0, st, a, band, nbands);
  else hipLaunchKernelGGL((cheb_pair_kernel<8, false>), dim3((unsigned)wgs), dim3(128), 0, st, a, band, nbands);
  return hipGetLastError();
}

}  // namespace lmc
