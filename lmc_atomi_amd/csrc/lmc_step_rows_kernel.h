// Fused MYULA update for a separable blur data term and a prior without a stencil (l2, l1, none, or a prox computed by a
// preceding launch):   out = a*x - t*sigma_f H^T(Hx - y) + b*prox(x) + s*xi        (algs.py:569, grad algs.py:283-284)
//
// Barrier-free row streaming.  ONE wave owns the full width of a band of rows of one chain: lane l holds PXL = 4 or 8
// consecutive pixels of the row (W <= 64*PXL), so the zero boundary of the "same" convolution is the wave's own edge and no
// wave ever talks to another one -- no LDS, no __syncthreads.  The wave walks down its band one input row per step:
//   x row i --h-blur--> scattered into KT residual accumulators (rows i-HW..i+HW)
//   residual row i-HW complete: R = Hx - y (zero outside the image) --h-adjoint--> scattered into KT gradient accumulators
//   gradient row o = i-(KT-1) complete: combine with x[o] (still in the register ring), prox, Philox noise, store.
// Horizontal neighbours: 2*HW wave-shift DPP moves per pass for PXL pixels.  All ring slots are (row & 7) with the row loop
// unrolled by 8, so every index is a compile-time constant and the rings live in VGPRs without rotation moves.
// Bands start KT-1 rows early (recompute instead of exchange); HBM traffic = x read once (+ band overlap) + x' written once.
// (kernel templates: shared by lmc_step_rows.hip -- host side, the general-taps and CG / Chebyshev-statistics instantiations -- and lmc_step_rows_uni.hip --
// the uniform-box instantiations; two translation units so that the ~30 instantiations x 3 copies of the body compile side by side)
#pragma once
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

#ifndef LMC_ROWS_PF
#define LMC_ROWS_PF 4
#endif
#ifndef LMC_ROWS_PF4
#define LMC_ROWS_PF4 4
#endif
#ifndef LMC_ROWS_PF7
#define LMC_ROWS_PF7 2
#endif
#ifndef LMC_ROWS_YD7
#define LMC_ROWS_YD7 0
#endif

template <int PXL, int KT>
struct RowsGeom {
  static constexpr int HW = (KT - 1) / 2;         // taps are centred: window c-HW .. c+HW
  static constexpr int LAG = KT - 1;              // output row = input row - LAG
  // x rows fetched ahead; PF + LAG <= 8 keeps row o's slot intact.  The depth is what the register file allows without spilling (scratch sizes are
  // fenced by tests/test_kernel_resources.py): 4 at 5 taps (8 or 4 pixels per lane).  At 7 taps and 8 pixels per lane three rings of 7-8 rows x 8 pixels leave little: x rows 2 steps ahead
  // and the observation row requested in the step that uses it (measured at 512 x 512 x 1024, 7 x 7 box + l2 prior: 0.600 ms per launch; x 1 ahead /
  // y 1 ahead 0.629; x 1 / y 0: 0.665; round 2's 2 / 1 with 23 spilled VGPRs: 0.647), 1 ahead for the 6 x 6 box (window 0..5), which still spilled at 2
  static constexpr int PFW = (PXL == 8 && KT == 7) ? LMC_ROWS_PF7 : (PXL == 4 ? LMC_ROWS_PF4 : LMC_ROWS_PF);
  static constexpr int PF = 8 - LAG < PFW ? 8 - LAG : PFW;
  static constexpr int YD = (PXL == 8 && KT == 7) ? LMC_ROWS_YD7 : 1;        // observation rows requested this many steps ahead
};

// Row load with zero fill (predicated: here the value must stay untouched until its use several steps later -- a select applied at
// load time would wait for the prefetch at once; measured 0.75 vs 0.65 ms).
template <int PXL>
__device__ __forceinline__ void rows_load(float (&dst)[PXL], const float* __restrict__ row, int c0, int W, bool rowok) {
#pragma unroll
  for (int g = 0; g < PXL / 4; ++g) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (rowok && c0 + 4 * g < W) v = *reinterpret_cast<const float4*>(row + c0 + 4 * g);
    dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
  }
}

// Unconditional load without the zero fill, for values masked where they are used (lanes past the row read its start; callers pass
// a clamped, valid row): no exec-mask branch, and nothing touches the value before its use, so the prefetch stays in flight.
// AL = false: rows that do not start on 16-byte boundaries (W % 4 != 0): dword-aligned 16-byte accesses (lmc_device.h), masks per pixel.
template <int PXL, bool AL = true>
__device__ __forceinline__ void rows_load_raw(float (&dst)[PXL], const float* __restrict__ row, int c0, int W) {
  if constexpr (AL) {
#pragma unroll
    for (int g = 0; g < PXL / 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(row + (c0 + 4 * g < W ? c0 + 4 * g : 0));
      dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
    }
  } else {
#pragma unroll
    // the group that holds the row end is moved into place here, at the load: moving it at the uses instead (as the pipe kernel does) was
    // measured -- 1.03 vs 1.02 ms on 667 x 877 x 512 -- and costs this register-bound kernel 17 to 120 more spilled VGPRs
    for (int g = 0; g < PXL / 4; ++g) load4_dword_aligned(dst[4 * g], dst[4 * g + 1], dst[4 * g + 2], dst[4 * g + 3], row, c0 + 4 * g, W);
  }
}
template <bool AL>
__device__ __forceinline__ void rows_load4(float (&dst)[4], const float* __restrict__ row, int c, int W) {
  if constexpr (AL) {
    const float4 v = *reinterpret_cast<const float4*>(row + c);
    dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; dst[3] = v.w;
  } else load4_dword_aligned(dst[0], dst[1], dst[2], dst[3], row, c, W);
}

// ULO >= 0: UNIFORM-BOX form (the reference's only blurs: ones(k, k) / k^2, prox_lmc_deconv.py:55-69).  The centred taps are c_u on [ULO, UHI] and zero
// elsewhere (same window for rows and columns), so every 1-D pass is a sliding window sum: 2 operations per pixel instead of KT (horizontal: the
// first pixel of a lane directly, the next ones by +new -old; vertical: a running sum over a ring of the horizontally filtered rows, re-formed
// directly every 8th row so that rounding cannot drift).  The scale c_u^2 c_v^2 sigma_f is applied once.  Same update to rounding (tests).
// EP: the prior is one of the closed-form elementwise proxes of prox.py (LMC_PRIOR_EPROX) -- instantiations of their own, so that the fifteen
// closed forms (cube roots, square roots) cost the register-tight l2 / l1 / ready-made-prox kernels nothing.
// XLM: which of the optional per-pixel inputs the combine step LOADS -- bit 0 injected noise, bit 1 a ready-made prox (prox_ext), bit 2 the ME-TV prox
// image (extra) -- fixed at compile time, or -1 = decided at run time.  A load under a run-time condition makes the compiler wait for vmcnt(0) where the
// paths merge -- before every output store, i.e. for every x / y row prefetched for LATER steps and for the wave's previous stores (memory operations retire in
// order through one counter): the prefetch distance was void.  The kernel below picks the copy for the two masks that matter (0: MYULA with Philox noise;
// 3: a Chebyshev step, u_{k-1} through the noise input and the right-hand side as prox_ext) and keeps the run-time form for the rest.
template <int PXL, int KT, bool DOT, int ULO, int UHI, bool AL, bool EP, int XLM>
__device__ __forceinline__ void rows_body(const StepArgs& P, const int band_rows, const int nbands, float* const nz_lds) {
  using Gm = RowsGeom<PXL, KT>;
  constexpr int HW = Gm::HW, LAG = Gm::LAG, PF = (PXL == 8 && KT == 7 && (UHI == 5 || UHI == 6 || EP) && AL) ? 1 : Gm::PF;
  constexpr bool UNI = ULO >= 0;
  static_assert(!UNI || (UHI >= ULO && UHI < KT && !DOT), "uniform-box window");
  if constexpr (DOT) {
    if (P.skip_flag && *P.skip_flag) return;            // CG operator apply after convergence (lmc_capi.hip: cg_solve_fused)
  }
  if (P.run_count && *P.run_count <= P.run_index) return;   // Chebyshev iteration the solve does not need (uniform, one scalar load)
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int H = P.H, W = P.W;
  const bool x_noise = XLM < 0 ? P.noise_mode == LMC_NOISE_INJECTED : (XLM & 1) != 0;
  const bool x_prox = XLM < 0 ? P.prox_ext != nullptr : (XLM & 2) != 0;
  const bool x_extra = XLM < 0 ? P.extra != nullptr : (XLM & 4) != 0;
  // Column strips (W > 64 PXL): a wave covers the columns [strip U - HALO, strip U - HALO + 64 PXL) of its band and writes the interior
  // [strip U, (strip + 1) U); the gradient of a pixel needs x within KT - 1 columns, so the HALO columns either side are recomputed, not exchanged.
  constexpr int HALO = 8, USTRIP = 64 * PXL - 2 * HALO;        // strips only ever run 8 pixels per lane (W > 512)
  static_assert(HALO >= KT - 1, "strip halo");
  const int nstrips = W <= 64 * PXL ? 1 : (W + USTRIP - 1) / USTRIP;
  if (gw >= P.C * nbands * nstrips) return;             // whole waves leave; nothing below synchronises
  const int chain = gw / (nbands * nstrips), bs = gw - chain * (nbands * nstrips);
  const int strip = bs / nbands, band = bs - strip * nbands;
  const int r0 = band * band_rows, r1 = min(r0 + band_rows, H);
  const int st_lo = nstrips > 1 ? strip * USTRIP : 0, st_hi = nstrips > 1 ? min(W, st_lo + USTRIP) : W;
  const int c0 = (strip ? st_lo - HALO : 0) + lane * PXL;        // GLOBAL column of the lane's first pixel
  auto colok = [&](int k) { return AL ? c0 + (k & ~3) < W : c0 + k < W; };
  const size_t img = (size_t)H * W;
  const float* __restrict__ xin = P.x_in + (size_t)chain * img;
  float* __restrict__ xout = P.x_out + (size_t)chain * img;
  const float* __restrict__ uv = P.blur.h;              // u[0..KT) then v[0..KT) at h[kMaxBlur..], centred, zero padded
  const int i_first = r0 - LAG;
  const uint32_t iter = P.iteration + (P.iter_dev ? *P.iter_dev : 0u);      // uniform scalar load; graph replays advance *iter_dev

  // 8 px / lane: the 32 normals of a quad row-group wait in a wave-private LDS slab (each lane reads back only what it
  // wrote, so no barrier) instead of 32 VGPRs -- the register file is the limit at 2 waves / SIMD.
  constexpr bool kNzLds = PXL == 8;        // nz_lds: [4 waves][PXL * 4][64] floats, declared by the kernel (one array for all copies of this body)
  float* const nzw = nz_lds + (kNzLds ? (threadIdx.x >> 6) * PXL * 4 * 64 + lane : 0);
  // general taps: A / G = residual / gradient accumulators of 8 rows in flight; uniform box: A / G = rings of the horizontally filtered rows
  // (of x / of the residual), Vs / Ws = the running vertical window sums
  float xr[8][PXL], A[8][PXL], G[8][PXL], nz[kNzLds ? 1 : PXL][4], yq[4][PXL];
  float Vs[UNI ? PXL : 1], Ws[UNI ? PXL : 1];
#pragma unroll
  for (int k = 0; k < (UNI ? PXL : 1); ++k) Vs[k] = Ws[k] = 0.f;
  const float cbox = UNI ? P.blur.h[ULO] * P.blur.h[kMaxBlur + ULO] : 0.f;       // c_u c_v
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int k = 0; k < PXL; ++k) { xr[s][k] = 0.f; A[s][k] = 0.f; G[s][k] = 0.f; }
#pragma unroll
  for (int k = 0; k < (kNzLds ? 1 : PXL); ++k) nz[k][0] = nz[k][1] = nz[k][2] = nz[k][3] = 0.f;
  if constexpr (kNzLds) {
#pragma unroll
    for (int k = 0; k < PXL * 4; ++k) nzw[k * 64] = 0.f;
  }

  // prime the x ring: rows i_first .. i_first + PF - 1   (i_first = r0 - LAG with r0 % 8 == 0, so slot = (8 - LAG + p) & 7)
  static_for<0, PF>([&](auto pp) {
    constexpr int p = decltype(pp)::value;
    const int i = i_first + p;
    rows_load_raw<PXL, AL>(xr[(8 - LAG + p) & 7], xin + (size_t)min(max(i, 0), H - 1) * W, c0, W);
  });

  // Vector-memory loads return in order, so the load a step consumes must be older than the x rows still in flight for later steps:
  // the observation row is requested kYD steps ahead and first in its step (one step ahead and after the x prefetch, every step
  // waited for an HBM access issued one step earlier).
  constexpr int kYD = Gm::YD;   // 1; 2 was measured: no gain, and the extra ring slots cost the registers the packed build needs
  static_for<0, kYD>([&](auto dd) {   // observation rows of the first kYD steps (residual rows i_first - HW + d, slot (J & 3))
    constexpr int d = decltype(dd)::value;
    const int r = i_first + d - HW;
    rows_load_raw<PXL, AL>(yq[(8 - LAG + d) & 3], P.y + (size_t)min(max(r, 0), H - 1) * W, c0, W);
  });

  double dacc = 0.0;       // sum x_in * x_out over this wave's band (P.dot_out: the p.Ap of CG); dot_mode 1: sum (x_out - x_in)^2
  double dacc2 = 0.0;      // dot_mode 1: sum prox_ext^2

  // One step = input row i = base + J (J = i & 7 is a compile-time constant: every ring slot below is static).
  // No step is conditional, so a ring slot is dead between its last read and the assignment that restarts it.
  auto step = [&](auto jj, const int base) __attribute__((always_inline)) {
    constexpr int J = decltype(jj)::value;
    const int i = base + J;
      // (0) the observation row of the step after next
      rows_load_raw<PXL, AL>(yq[(J + kYD) & 3], P.y + (size_t)min(max(i + kYD - HW, 0), H - 1) * W, c0, W);
      // (1) horizontal blur of x row i
      float hx[PXL];
      {
        float e[PXL + 2 * HW], xm[PXL];
        const bool rowin = i >= 0 && i < H;           // the ring holds raw loads: rows / columns outside the image are zeros HERE
#pragma unroll
        for (int k = 0; k < PXL; ++k) xm[k] = (rowin && colok(k)) ? xr[J][k] : 0.f;
#pragma unroll
        for (int m = 0; m < HW; ++m) e[m] = dpp_left0(xm[PXL - HW + m]);
#pragma unroll
        for (int k = 0; k < PXL; ++k) e[HW + k] = xm[k];
#pragma unroll
        for (int m = 0; m < HW; ++m) e[HW + PXL + m] = dpp_right0(xm[m]);
        if constexpr (UNI) {     // window sum of e[k + 2HW - UHI .. k + 2HW - ULO]
          float acc = e[2 * HW - UHI];
#pragma unroll
          for (int j = 2 * HW - UHI + 1; j <= 2 * HW - ULO; ++j) acc += e[j];
          hx[0] = acc;
#pragma unroll
          for (int k = 1; k < PXL; ++k) hx[k] = (hx[k - 1] + e[k + 2 * HW - ULO]) - e[k - 1 + 2 * HW - UHI];
        } else {
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
          float acc = uv[kMaxBlur] * e[k + 2 * HW];
#pragma unroll
          for (int b = 1; b < KT; ++b) acc = fmaf(uv[kMaxBlur + b], e[k + 2 * HW - b], acc);
          hx[k] = acc;
        }
        }
      }
      const int r = i - HW;
      float R[PXL];
      if constexpr (UNI) {
        // (2u) ring of filtered rows; vertical window of residual row r: filtered rows i - UHI .. i - ULO
#pragma unroll
        for (int k = 0; k < PXL; ++k) A[J][k] = hx[k];
        // The window's newest row enters before the sum is used and its oldest row leaves right after (not one step later): UHI - ULO + 1 ring
        // rows are live between steps instead of UHI - ULO + 2 -- at 7 taps and 8 pixels per lane that is the difference between fitting the
        // 256 VGPRs of two waves per SIMD and spilling.
        if constexpr (J == 0) {          // re-form the sum directly (bounds the rounding drift of the running update)
#pragma unroll
          for (int k = 0; k < PXL; ++k) {
            float acc = A[(J - ULO + 8) & 7][k];
            static_for<ULO + 1, UHI + 1>([&](auto aa) { acc += A[(J - decltype(aa)::value + 16) & 7][k]; });
            Vs[k] = acc;
          }
        } else {
#pragma unroll
          for (int k = 0; k < PXL; ++k) Vs[k] += A[(J - ULO + 8) & 7][k];
        }
        const bool rowok = r >= 0 && r < H && r >= r0 - HW;      // rows before the band's first residual row: partial windows, kept out
#pragma unroll
        for (int k = 0; k < PXL; ++k) R[k] = (rowok && colok(k)) ? fmaf(cbox, Vs[k], -yq[J & 3][k]) : 0.f;
#pragma unroll
        for (int k = 0; k < PXL; ++k) Vs[k] -= A[(J - UHI + 16) & 7][k];
      } else {
      // (2) scatter into the residual accumulators of rows i-HW .. i+HW (the last one starts here)
      static_for<0, KT>([&](auto aa) {
        constexpr int a = decltype(aa)::value;
        constexpr int s = (J + a - HW + 8) & 7;
#pragma unroll
        for (int k = 0; k < PXL; ++k) A[s][k] = (a == KT - 1) ? uv[a] * hx[k] : fmaf(uv[a], hx[k], A[s][k]);
      });
      // (3) residual row r = i - HW is complete
      constexpr int sr = (J - HW + 8) & 7;
      {
        const bool rowok = r >= 0 && r < H;
#pragma unroll
        for (int k = 0; k < PXL; ++k) R[k] = (rowok && colok(k)) ? A[sr][k] - yq[J & 3][k] : 0.f;
      }
      }
      // (4) horizontal adjoint of the residual row
      float hr[PXL];
      {
        float e[PXL + 2 * HW];
#pragma unroll
        for (int m = 0; m < HW; ++m) e[m] = dpp_left0(R[PXL - HW + m]);
#pragma unroll
        for (int k = 0; k < PXL; ++k) e[HW + k] = R[k];
#pragma unroll
        for (int m = 0; m < HW; ++m) e[HW + PXL + m] = dpp_right0(R[m]);
        if constexpr (UNI) {     // window sum of e[k + ULO .. k + UHI]
          float acc = e[ULO];
#pragma unroll
          for (int j = ULO + 1; j <= UHI; ++j) acc += e[j];
          hr[0] = acc;
#pragma unroll
          for (int k = 1; k < PXL; ++k) hr[k] = (hr[k - 1] + e[k + UHI]) - e[k - 1 + ULO];
        } else {
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
          float acc = uv[kMaxBlur] * e[k];
#pragma unroll
          for (int b = 1; b < KT; ++b) acc = fmaf(uv[kMaxBlur + b], e[k + b], acc);
          hr[k] = acc;
        }
        }
      }
      if constexpr (UNI) {
        // (5u) ring of filtered residual rows (slot = residual row & 7); gradient row o = i - LAG: residual rows o - HW + ULO .. o - HW + UHI
        constexpr int sR = (J - HW + 8) & 7;                    // slot of residual row r = i - HW
#pragma unroll
        for (int k = 0; k < PXL; ++k) G[sR][k] = hr[k];
        constexpr int sNew = (J - LAG - HW + UHI + 32) & 7;
        if constexpr (J == 0) {
#pragma unroll
          for (int k = 0; k < PXL; ++k) {
            float acc = G[(J - LAG - HW + ULO + 32) & 7][k];
            static_for<ULO + 1, UHI + 1>([&](auto aa) { acc += G[(J - LAG - HW + decltype(aa)::value + 32) & 7][k]; });
            Ws[k] = acc;
          }
        } else {
#pragma unroll
          for (int k = 0; k < PXL; ++k) Ws[k] += G[sNew][k];
        }
      } else {
      // (5) scatter into the gradient accumulators of rows r+HW .. r-HW, i.e. i .. i-LAG (the first one starts here)
      static_for<0, KT>([&](auto aa) {
        constexpr int a = decltype(aa)::value;
        constexpr int s = (J - a + 8) & 7;
#pragma unroll
        for (int k = 0; k < PXL; ++k) G[s][k] = (a == 0) ? uv[0] * hr[k] : fmaf(uv[a], hr[k], G[s][k]);
      });
      }
      // (6) output row o = i - LAG
      const int o = i - LAG;
      constexpr int so = (J - LAG + 8) & 7;
      if (o >= r0 && o < r1) {
        if constexpr (((J - LAG + 8) & 3) == 0) {       // first row of a Philox quad (r0 % 8 == 0)
          if (!x_noise && P.noise_mode == LMC_NOISE_PHILOX) {
#pragma unroll
            for (int k = 0; k < PXL; ++k) {
              float n4[4];
              quad_normals(P.key0, P.key1, iter, P.chain_offset + (uint32_t)chain,
                           (uint32_t)(o >> 2) * (uint32_t)W + (uint32_t)(c0 + k), n4);
#pragma unroll
              for (int q = 0; q < 4; ++q) {
                if constexpr (kNzLds) nzw[(q * PXL + k) * 64] = n4[q];
                else nz[k][q] = n4[q];
              }
            }
          }
        }
        constexpr int jq = (J - LAG + 8) & 3;
        const size_t go = (size_t)o * W;
#pragma unroll
        for (int g = 0; g < PXL / 4; ++g) {
          if (c0 + 4 * g < st_hi && c0 + 4 * g + 3 >= st_lo) {       // the group touches this wave's interior (AL: inside or outside as a whole)
            float xi[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              if constexpr (kNzLds) xi[q] = nzw[(jq * PXL + 4 * g + q) * 64];
              else xi[q] = nz[4 * g + q][jq];
            }
            if (x_noise) rows_load4<AL>(xi, P.noise + (size_t)chain * img + go, c0 + 4 * g, W);
            float pe[4] = {0.f, 0.f, 0.f, 0.f}, ex[4] = {0.f, 0.f, 0.f, 0.f};
            if (x_prox) rows_load4<AL>(pe, P.prox_ext + (size_t)chain * img + go, c0 + 4 * g, W);
            if (x_extra) rows_load4<AL>(ex, P.extra + (size_t)chain * img + go, c0 + 4 * g, W);
            float ov[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x = xr[so][4 * g + q];
              float gr = UNI ? (P.sigma_f * cbox) * Ws[4 * g + q] : P.sigma_f * G[so][4 * g + q];
              if (x_extra) gr = fmaf(P.extra_coef, x - ex[q], gr);
              float px = x;
              if constexpr (EP) px = eprox(P.eprox_kind, x, EproxParams{P.prior_p0, P.prior_p1});
              else if (P.prior_kind == LMC_PRIOR_L2) px = x * P.prior_p0;
              else if (P.prior_kind == LMC_PRIOR_L1) px = copysignf(fmaxf(fabsf(x) - P.prior_p0, 0.f), x);
              if (x_prox) px = pe[q];
              ov[q] = fmaf(P.a, x, fmaf(-P.t, gr, fmaf(P.b, px, P.s * xi[q])));
              if constexpr (DOT) {
                const bool mine = AL || (c0 + 4 * g + q >= st_lo && c0 + 4 * g + q < st_hi);
                if (!mine) continue;
                if (P.dot_mode == 0) dacc = fma((double)x, (double)ov[q], dacc);
                else {
                  const double d = (double)ov[q] - (double)x;
                  dacc = fma(d, d, dacc);
                  dacc2 = fma((double)pe[q], (double)pe[q], dacc2);
                }
              }
            }
            if constexpr (AL) *reinterpret_cast<float4*>(xout + go + c0 + 4 * g) = make_float4(ov[0], ov[1], ov[2], ov[3]);
            else store4_dword_aligned(xout + go, c0 + 4 * g, st_lo, st_hi, ov[0], ov[1], ov[2], ov[3]);
          }
        }
      }
      if constexpr (UNI) {     // the oldest row of the gradient window leaves (see (2u))
        constexpr int sOldNow = (J - LAG - HW + ULO + 32) & 7;
#pragma unroll
        for (int k = 0; k < PXL; ++k) Ws[k] -= G[sOldNow][k];
      }
      // (7) fetch x row i + PF into the slot row i + PF - 8 has just left (its last use was step (6) above at the latest)
      {
        const int ip = i + PF;
        rows_load_raw<PXL, AL>(xr[(J + PF) & 7], xin + (size_t)min(max(ip, 0), H - 1) * W, c0, W);
      }
  };
  const int r1r = (r1 + 7) & ~7;
  static_for<8 - LAG, 8>([&](auto jj) { step(jj, r0 - 8); });                    // fill: rows r0-LAG .. r0-1, nothing to emit
  for (int base = r0; base < r1r; base += 8) static_for<0, 8>([&](auto jj) { step(jj, base); });
  static_for<0, LAG>([&](auto jj) { step(jj, r1r); });                          // drain: the last LAG output rows
  if constexpr (DOT) {
    const double tot = wave_sum(dacc);
    if (P.dot_mode == 0) {
      if (lane == 0) unsafeAtomicAdd(&P.dot_out[chain], tot);
    } else {
      const double tot2 = wave_sum(dacc2);
      if (lane == 0) { unsafeAtomicAdd(&P.dot_out[2 * chain], tot); unsafeAtomicAdd(&P.dot_out[2 * chain + 1], tot2); }
    }
  }
}

#ifndef LMC_ROWS_X3_OFF7
#define LMC_ROWS_X3_OFF7 0
#endif
template <int PXL, int KT, bool DOT = false, int ULO = -1, int UHI = -1, bool AL = true, bool EP = false>
__global__ __launch_bounds__(256, (PXL == 8 || KT == 7) ? 2 : 3) void myula_step_rows_kernel(const StepArgs P, const int band_rows, const int nbands) {
  __shared__ float nz_lds[PXL == 8 ? 4 * PXL * 4 * 64 : 1];
  const int m = (P.noise_mode == LMC_NOISE_INJECTED ? 1 : 0) | (P.prox_ext ? 2 : 0) | (P.extra ? 4 : 0);      // uniform over the grid
  if (m == 0) rows_body<PXL, KT, DOT, ULO, UHI, AL, EP, 0>(P, band_rows, nbands, nz_lds);
  else if (!EP && !(LMC_ROWS_X3_OFF7 && PXL == 8 && KT == 7) && m == 3) rows_body<PXL, KT, DOT, ULO, UHI, AL, EP, EP ? -1 : 3>(P, band_rows, nbands, nz_lds);
  else rows_body<PXL, KT, DOT, ULO, UHI, AL, EP, -1>(P, band_rows, nbands, nz_lds);
}

}  // namespace lmc
