// Fused MYULA update with the isotropic-TV prox (K FGP dual iterations), "pipe" variant:
//     out = a*x - t*sigma_f H^T(Hx - y) + b*prox_{gamma TV}(x) + s*xi                      (algs.py:569)
//
// Same row pipeline as lmc_step_split.hip (stage k runs on row t-E-2k, one barrier per tick) but laid out the other way
// round: a wavefront owns the FULL WIDTH of the image (lane = PXL consecutive pixels, W <= 64*PXL) and the pipeline STAGES
// are spread over the wavefronts of the workgroup:
//   wave 0      "L": HBM load of row t -> x ring (LDS); blur-gradient pipeline on the ring, one row ahead of the output
//   wave 1..NT  "T": two TV stages each (2j-1, 2j); stage outputs handed to the next wave through LDS (parity double
//                    buffer, read one tick later -- the latency a stage boundary has anyway)
//   wave NT+1   "C": final primal step x - gamma div(rr^K, ss^K), combine, HBM store
//   wave NT+2   "N": Philox normals of the next quad row-group into an LDS slab (read by C four ticks later)
// Because a wave spans the image width, horizontal neighbours are in the same lane (7 of 8) or one wave-shift DPP move
// away (2 per stage per PXL pixels): no row-edge ghost exchange, no per-pixel DPP.  One workgroup = one chain.
#pragma once
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

#ifndef LMC_RT_SPREAD
#define LMC_RT_SPREAD 1
#endif
#ifndef LMC_RT_CHAIN_MAP
#define LMC_RT_CHAIN_MAP 1
#endif
#ifndef LMC_RT_MAP_SPREAD
#define LMC_RT_MAP_SPREAD 3
#endif
#ifndef LMC_WARM_MIN_WAVES
#define LMC_WARM_MIN_WAVES 1
#endif
// Timing experiments of round 2 that were measured and removed again (git history): stage k1's hand-off reads before stage
// k2's arithmetic (1.876 ms), the same with stage k2's stores interleaved with its arithmetic behind scheduling barriers (1.799 ms), and the
// two-team layout -- 16 waves of 4 pixels per lane, the teams' roles on complementary SIMDs, the column seam through LDS; exact -- at commit
// 4183175 (1.811 ms).  Base: 1.749 ms.
#ifdef LMC_EXP_NOBARRIER   // timing experiment: waves free-run (results are wrong); a compiler-only fence keeps every LDS store alive (without it the stores of
                           // a tick that the same wave overwrites two ticks later are dead, and the arithmetic behind them with them)
#define PIPE_TICK_SYNC() asm volatile("" ::: "memory")
#elif defined(LMC_EXP_SLEEP_MASK)   // timing experiment: the waves in the mask start every tick LMC_EXP_SLEEP_N x 64 cycles late (phase shift)
#define PIPE_TICK_SYNC() do { __syncthreads(); if ((LMC_EXP_SLEEP_MASK >> wave) & 1) __builtin_amdgcn_s_sleep(LMC_EXP_SLEEP_N); } while (0)
#else
#define PIPE_TICK_SYNC() __syncthreads()
#endif

template <int K>
struct PipeGeom {
  static constexpr int D = (2 * K + 2 > 10) ? 2 * K + 2 : 10;   // output row lag: o = t - D
  static constexpr int E = D - (2 * K + 2);                     // extra lag of the TV pipeline
  static constexpr int RB = D + 1;                              // x ring rows: t-D .. t
  static constexpr int NT = (K + 1) / 2;                        // TV waves (two stages each; odd K: the last one runs stage K alone)
};

template <int K, int PXL, bool CHAIN = false>
struct PipeLds {
  static constexpr int BW = 64 * PXL;
  static constexpr int o_x = 0;                                        // [RB][BW]
  static constexpr int o_hand = o_x + PipeGeom<K>::RB * BW;            // [NT][2][4][BW]: rr, ss, p, q of the wave's last stage
  // chained launches (more than K dual iterations): the last boundary (T_NT -> C) carries rr, ss only, [2][2][BW], and the 8*BW
  // saved hold the dual state of the previous launch for stage 1, [2][4][BW] (LDS budget: 160 KB)
  static constexpr int o_hand_last = o_hand + (PipeGeom<K>::NT - 1) * 8 * BW;
  static constexpr int o_hand0 = o_hand_last + 4 * BW;
  static constexpr int o_g = CHAIN ? o_hand0 + 8 * BW : o_hand + PipeGeom<K>::NT * 2 * 4 * BW;    // [2][BW] gradient of the output row
  static constexpr int o_slab = o_g + 2 * BW;                          // [2][4][PXL][64] normals of this and the next quad row-group
  static constexpr int total = o_slab + 2 * 4 * PXL * 64;
};

// LDS rows are stored so that every 16-byte access of a wave is contiguous: pixel k of lane l at (k>>2)*256 + 4*l + (k&3).
template <int PXL>
__device__ __forceinline__ void prow_load(float (&v)[PXL], const float* row, int lane) {
#pragma unroll
  for (int g = 0; g < PXL / 4; ++g) {
    const float4 q = *reinterpret_cast<const float4*>(row + g * 256 + lane * 4);
    v[4 * g] = q.x; v[4 * g + 1] = q.y; v[4 * g + 2] = q.z; v[4 * g + 3] = q.w;
  }
}
template <int PXL>
__device__ __forceinline__ void prow_store(float* row, int lane, const float (&v)[PXL]) {
#pragma unroll
  for (int g = 0; g < PXL / 4; ++g)
    *reinterpret_cast<float4*>(row + g * 256 + lane * 4) = make_float4(v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]);
}

// Packed fp32: gfx950 issues one wave64 VALU instruction per 4 cycles per SIMD, and v_pk_fma/mul/add_f32 process two floats
// per lane in that slot.  The TV stages therefore work on pixel PAIRS (2i, 2i+1) held in even-aligned register pairs; only
// the two neighbour-shifted operands of a stage need a v_pk_mov to re-pair, max / rsq stay scalar.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f pk_fma(v2f a, v2f b, v2f c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ v2f pk_set(float a) { return v2f{a, a}; }

template <int NP>   // NP = PXL / 2 pairs
__device__ __forceinline__ void pairs_load(v2f (&v)[NP], const float* row, int lane) {
#pragma unroll
  for (int g = 0; g < NP / 2; ++g) {
    const float4 q = *reinterpret_cast<const float4*>(row + g * 256 + lane * 4);
    v[2 * g] = v2f{q.x, q.y};
    v[2 * g + 1] = v2f{q.z, q.w};
  }
}
template <int NP>
__device__ __forceinline__ void pairs_store(float* row, int lane, const v2f (&v)[NP]) {
#pragma unroll
  for (int g = 0; g < NP / 2; ++g)
    *reinterpret_cast<float4*>(row + g * 256 + lane * 4) = make_float4(v[2 * g].x, v[2 * g].y, v[2 * g + 1].x, v[2 * g + 1].y);
}

template <int NP>
struct DualRow { v2f rr[NP], ss[NP], p[NP], q[NP]; };

// One FGP dual iteration on NP pixel pairs per lane.  r1, s1 = (rr, ss)^{k-1} on row a; in0 = (rr, ss, p, q)^{k-1} on row
// b = a-1; solb = sol^k on row b (in) -> sol^k on row a (out); out = (rr, ss, p, q)^k on row b.
// The horizontal step coefficient per pixel: -c, and 0 for the pixel in the last image column (no difference across it).  LASTLANE: the image
// width is a multiple of the pixels per lane, so that pixel is the last one of a lane (cr_last, a per-lane scalar); otherwise it can be any
// pixel of a lane and the coefficients are a per-lane register array (ncrv).
template <int NP>
struct PipeCr { float cstep, cr_last; v2f ncrv[NP]; };
template <int NP, bool LASTLANE>
__device__ __forceinline__ v2f pipe_ncr(const PipeCr<NP>& c, int i) {
  if constexpr (LASTLANE) return i == NP - 1 ? v2f{-c.cstep, -c.cr_last} : pk_set(-c.cstep);
  else return c.ncrv[i];
}

// Objective by-products of a stage (RT instantiations: the per-chain early exit of the TV prox).  The stage forms the iterate sol = x - gam div(rr, ss)
// on row a and has the one on row b = a - 1 from the previous tick: the pieces of its primal objective 0.5 ||x - sol||^2 + gam TV(sol) are sums of
// what the stage computes anyway -- x - sol = gam T with T = div(rr, ss) on row a, and the forward differences of sol on row b (masked like the dual
// step: none across the last row / column).  sq += sum T^2 (row a), tv += sum |grad sol| (row b), fp32 per row; the caller folds rows into fp64.
struct StageObj { v2f sq, tv; };
template <int NP>
struct ObjMask { float md; v2f mlast; };     // 1 / 0: the row below exists; the column to the right of the lane's last pair exists

template <int NP, bool LASTLANE = true, bool OBJ = false>
__device__ __forceinline__ void pipe_stage(const v2f (&xa)[NP], const v2f (&r1)[NP], const v2f (&s1)[NP], const DualRow<NP>& in0,
                                           v2f (&solb)[NP], float gam, float cdown, const PipeCr<NP>& cr, float beta,
                                           DualRow<NP>& out, StageObj* ob = nullptr, const ObjMask<NP>* om = nullptr) {
  v2f sol[NP];
  const float ssl0 = dpp_left0(s1[NP - 1].y);
  const v2f ngam = pk_set(-gam), ncd = pk_set(-cdown), vb = pk_set(beta);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const v2f ssl = v2f{i == 0 ? ssl0 : s1[i - 1].y, s1[i].x};
    const v2f T = (r1[i] - in0.rr[i]) + (s1[i] - ssl);
    sol[i] = pk_fma(ngam, T, xa[i]);
    if constexpr (OBJ) ob->sq = i == 0 ? T * T : pk_fma(T, T, ob->sq);
  }
  const float solr_last = dpp_right0(solb[0].x);
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const v2f solr = v2f{solb[i].y, i == NP - 1 ? solr_last : solb[i + 1].x};
    const v2f ncr = pipe_ncr<NP, LASTLANE>(cr, i);
    const v2f dxv = sol[i] - solb[i], dyv = solr - solb[i];
    if constexpr (OBJ) {
      static_assert(!OBJ || LASTLANE, "objective by-products: rows whose last column is a lane's last pixel");
      const v2f dxm = dxv * pk_set(om->md), dym = i == NP - 1 ? dyv * om->mlast : dyv;
      const v2f n = pk_fma(dxm, dxm, dym * dym);
      const v2f nr = v2f{__builtin_amdgcn_sqrtf(n.x), __builtin_amdgcn_sqrtf(n.y)};
      ob->tv = i == 0 ? nr : ob->tv + nr;
    }
    const v2f r = pk_fma(ncd, dxv, in0.rr[i]);
    const v2f s = pk_fma(ncr, dyv, in0.ss[i]);
    const v2f n2 = pk_fma(r, r, s * s);
    // min(1, rsq(n2)) == rsq(max(n2, 1)) bit for bit (rsq is monotone, rsq(1) = 1); written as a [0,1] clamp it folds into the
    // output modifier of v_rsq_f32 and the v_max disappears
    const v2f inv = v2f{__builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(n2.x), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(n2.y), 0.f, 1.f)};
    const v2f pn = r * inv, qn = s * inv;
    out.rr[i] = pk_fma(vb, pn - in0.p[i], pn);
    out.ss[i] = pk_fma(vb, qn - in0.q[i], qn);
    out.p[i] = pn;
    out.q[i] = qn;
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) solb[i] = sol[i];
}

// Stage 1 of a launch that starts from the zero dual state: (rr, ss, p, q)^0 = 0, so sol^1 = x and the differences with the previous
// iterate vanish.  Bit-identical to pipe_stage() fed with zeros (x - 0 = x, fma(c, d, 0) = c*d), at ~60 % of its instructions.
template <int NP, bool LASTLANE = true, bool OBJ = false>
__device__ __forceinline__ void pipe_stage_first(const v2f (&xa)[NP], v2f (&solb)[NP], float cdown, const PipeCr<NP>& cr, float beta,
                                                 DualRow<NP>& out, StageObj* ob = nullptr, const ObjMask<NP>* om = nullptr) {
  const float solr_last = dpp_right0(solb[0].x);
  const v2f ncd = pk_set(-cdown), vb = pk_set(beta);
  if constexpr (OBJ) ob->sq = pk_set(0.f);       // sol^0 = x
#pragma unroll
  for (int i = 0; i < NP; ++i) {
    const v2f solr = v2f{solb[i].y, i == NP - 1 ? solr_last : solb[i + 1].x};
    const v2f ncr = pipe_ncr<NP, LASTLANE>(cr, i);
    const v2f dxv = xa[i] - solb[i], dyv = solr - solb[i];
    if constexpr (OBJ) {
      const v2f dxm = dxv * pk_set(om->md), dym = i == NP - 1 ? dyv * om->mlast : dyv;
      const v2f n = pk_fma(dxm, dxm, dym * dym);
      const v2f nr = v2f{__builtin_amdgcn_sqrtf(n.x), __builtin_amdgcn_sqrtf(n.y)};
      ob->tv = i == 0 ? nr : ob->tv + nr;
    }
    const v2f r = ncd * dxv;
    const v2f s = ncr * dyv;
    const v2f n2 = pk_fma(r, r, s * s);
    const v2f inv = v2f{__builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(n2.x), 0.f, 1.f), __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(n2.y), 0.f, 1.f)};
    const v2f pn = r * inv, qn = s * inv;
    out.rr[i] = pk_fma(vb, pn, pn);
    out.ss[i] = pk_fma(vb, qn, qn);
    out.p[i] = pn;
    out.q[i] = qn;
  }
#pragma unroll
  for (int i = 0; i < NP; ++i) solb[i] = xa[i];
}

// Row load with zero fill.  The load itself is unconditional (masked-off lanes read the start of the row, always a valid address:
// callers pass a clamped row) and the mask is applied to the value: a predicated load costs an exec-mask branch per access and
// splits the tick into basic blocks the scheduler cannot move loads across.  `al`: rows start on 16-byte boundaries (W % 4 == 0): one
// float4 per group of four pixels; otherwise (any W, e.g. the reference's 667 x 877 image) dword-aligned 16-byte accesses (lmc_device.h) and per-pixel masks.
template <int PXL>
__device__ __forceinline__ void gload_row(float (&dst)[PXL], const float* __restrict__ row, int c0, int W, bool ok, bool al = true) {
  if (al) {
#pragma unroll
    for (int g = 0; g < PXL / 4; ++g) {
      const bool okg = ok && c0 + 4 * g < W;
      const float4 v = *reinterpret_cast<const float4*>(row + (okg ? c0 + 4 * g : 0));
      dst[4 * g] = okg ? v.x : 0.f; dst[4 * g + 1] = okg ? v.y : 0.f; dst[4 * g + 2] = okg ? v.z : 0.f; dst[4 * g + 3] = okg ? v.w : 0.f;
    }
  } else {
#pragma unroll
    for (int g = 0; g < PXL / 4; ++g) {
      float v[4];
      load4_dword_aligned(v[0], v[1], v[2], v[3], row, c0 + 4 * g, W);
#pragma unroll
      for (int q = 0; q < 4; ++q) dst[4 * g + q] = (ok && c0 + 4 * g + q < W) ? v[q] : 0.f;
    }
  }
}

// The same without the select: for values that are masked where they are USED (a select at load time makes the wave wait for
// the prefetch at once).  Lanes / pixels past the row read its start.
template <int PXL>
__device__ __forceinline__ void gload_raw(float (&dst)[PXL], const float* __restrict__ row, int c0, int W, bool al = true) {
  if (al) {
#pragma unroll
    for (int g = 0; g < PXL / 4; ++g) {
      const float4 v = *reinterpret_cast<const float4*>(row + (c0 + 4 * g < W ? c0 + 4 * g : 0));
      dst[4 * g] = v.x; dst[4 * g + 1] = v.y; dst[4 * g + 2] = v.z; dst[4 * g + 3] = v.w;
    }
  } else {
#pragma unroll
    for (int g = 0; g < PXL / 4; ++g) load4_dword_aligned_raw(dst[4 * g], dst[4 * g + 1], dst[4 * g + 2], dst[4 * g + 3], row, c0 + 4 * g, W);
  }
}
// ... whose consumer moves the group that holds the row end into place first (AL = false only; nothing to do for aligned rows)
template <int PXL, bool AL>
__device__ __forceinline__ void gfix_raw(float (&r)[PXL], int c0, int W) {
  if constexpr (!AL) unshift_row_dword_aligned<PXL>(r, c0, W);
}

// Four pixels of a row to global memory: columns c .. c + 3, of which those in [lo, hi) are written (the interior of a column strip, and < W).
__device__ __forceinline__ void gstore4(float* __restrict__ row, int c, int lo, int hi, bool al, float v0, float v1, float v2, float v3) {
  if (al) {        // W, lo, hi multiples of 4: the group is inside or outside as a whole
    if (c >= lo && c < hi) *reinterpret_cast<float4*>(row + c) = make_float4(v0, v1, v2, v3);
  } else store4_dword_aligned(row, c, lo, hi, v0, v1, v2, v3);
}

// Column strips (images wider than one wave: W > 64 PXL): a workgroup handles the columns [col_start, col_start + 64 PXL) of its chain, of which
// the interior [strip U, (strip + 1) U) is written; the HALO columns on either side are recomputed, not exchanged: the update of a pixel
// depends on x within K + 1 columns (one per dual iteration, one for the final divergence) and within 2 HW columns for the blur gradient.
// Rounded up to the pixels per lane, so that every strip starts on a lane-sized column multiple: 16-byte aligned accesses stay aligned and the
// last image column of an image whose width is a multiple of PXL stays the last pixel of a lane (LASTLANE) in every strip.
__host__ __device__ constexpr int pipe_halo(int K, int KT, int PXL) {
  const int need = (K + 1 > KT - 1 ? K + 1 : KT - 1);
  return (need + PXL - 1) / PXL * PXL;
}

// KT = 0: no data term (pure prox, or t = 0).  CHAIN: the launch is one link of a chain of launches that together run more than K
// dual iterations: stage 1 starts from the dual state A.tv_in of the previous link ([C][4][H][W]: rr, ss, p, q; NULL = zeros), the last
// stage's state goes to A.tv_out (NULL = not stored), and with A.tv_state_only the combine / store of x_out is skipped.
// WARM (with CHAIN): the state is the two-field projected dual carried between MYULA iterations (A.tv_warm) instead of the four-field
// link state -- a template parameter because the L wave's prefetch registers for the state rows set the kernel's VGPR count.
// AL: image rows start on 16-byte boundaries (W % 4 == 0): float4 global accesses; AL = false (any W): pixel-by-pixel accesses with per-pixel bounds
// (instantiated for K = 10 only: the reference's 667 x 877 image with niter_tv = 10 and the 10-iteration links of its ME-TV term).
// RT: per-chain early exit of the prox (StepArgs::rt_*; pyproximal.TV's rtol): the chain of this workgroup runs kc <= K live stages, the stages after
// them pass (rr, ss) through unchanged -- a delay line with the live stages' timing, so the combine wave forms x - gamma div(rr^kc, ss^kc) -- and every
// iterate formed leaves its primal objective behind.  In a chained launch the link a chain leaves in does its combine; the links before it only
// advance the dual state, the ones after it return at once.
// (Round 3, measured and removed again -- git history, commit 49bc3f1 and the three builds after it: the posterior moments of the INPUT state reduced inside this
// kernel, workgroup b summing the 256-pixel slices b, b + C, ... over all chains, two 1 KB reads per tick.  As a ninth wave: three waves on one SIMD cap the kernel
// at 168 VGPRs, 128-184 spilled.  In the combine wave: its conditional stores make the compiler wait for every load in flight each tick, 4.05 ms per iteration.  In the
// Philox wave, sums in registers, loads 4 / 8 ticks ahead: 2.41 / 2.32 ms against 1.92 with the reduction on a side stream -- and with the loads alone, no arithmetic,
// still 2.61 against 2.05 with the arithmetic alone: the CU's outstanding-miss capacity is what the scattered reads take from the L wave's row prefetches.)
template <int K, int PXL, int KT, bool CHAIN = false, bool WARM = false, bool AL = true, bool RT = false>
__global__ __launch_bounds__(64 * ((K + 1) / 2 + 3), (PXL == 8 || CHAIN) ? (WARM ? LMC_WARM_MIN_WAVES : 1) : 2) void myula_step_pipe_kernel(const StepArgs A) {
  static_assert(!WARM || CHAIN, "the warm dual uses the state hand-over of the chained launches");
  static_assert(!RT || (AL && !WARM && (K & 1) == 0), "per-chain exit: aligned rows, cold start, even K");
  using G = PipeGeom<K>;
  using L = PipeLds<K, PXL, CHAIN>;
  constexpr int D = G::D, E = G::E, RB = G::RB, NT = G::NT, BW = L::BW, HW = KT > 0 ? (KT - 1) / 2 : 0;
  static_assert(K >= 1, "at least one dual iteration");
  static_assert(D >= KT + 1, "the blur pipeline reads ring rows at least one tick old");
  extern __shared__ float lds[];
  const int lane = threadIdx.x & 63;
  const int hw_wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int chain = blockIdx.x;
  const int H = A.H, W = A.W;
  // per-chain exit: live stages of this launch, and whether this link only advances the dual state of this chain (it leaves in a later link)
  int kc = K;
  bool state_only = CHAIN && A.tv_state_only;
  if constexpr (RT) {
    const int kc_g = __builtin_amdgcn_readfirstlane(A.rt_kc[chain]);
    if (kc_g <= A.rt_base) return;                   // whole workgroup, before any barrier: left in an earlier link, or no run needed
    if (CHAIN && A.rt_start && A.rt_base < K * (__builtin_amdgcn_readfirstlane(A.rt_start[chain]) & 0xFFFF)) return;   // this link's work of an earlier round stands
    kc = min(kc_g - A.rt_base, K);
    state_only = CHAIN && kc_g > A.rt_base + K;
  }
  // `wave` = the ROLE index used below (0 = L, 1 .. NT = T, NT + 1 = C, NT + 2 = N).  Hardware waves w and w + 4 share a SIMD: L + T4 | T1 + T5 |
  // T2 + C | T3 + N.  In an RT launch the live stages are a prefix: a typical chain runs 3 to 5 of the 10 (waves T1, T2, half of T3), each with the
  // objective sums on top, the other T waves only pass the dual on, and the combine wave forms the objective of the iterate it returns.  A chain
  // with few live stages therefore pairs every heavy wave with a pass-through one: L + N | T1 + T4 | T2 + T3 | C + T5 (measured: DESIGN 3.0r).
  // Which roles share a SIMD was searched exhaustively in round 3 (scripts/round3/perm_search.py: all 105 pairings, 512 x 512 x 1024; the role of hardware
  // wave w is nibble w of the code): fixed K = 10: L + T2 | T1 + T3 | T4 + C | T5 + N 1.736 ms against 1.775 for the plain order and 2.07 for the worst
  // (L + C or L + N with two TV waves together); per-chain exit with one live stage per wave: by live count, below.
  int wave = hw_wave;
  if constexpr (K == 10 && !RT && !CHAIN) {   // 7 taps: L + C | T1 + T2 | T3 + T5 | T4 + N (1.838 vs 1.875); with the MC-TV term in the combine wave: L + T5 | T1 + T3 | T2 + C | T4 + N (1.962 vs 1.983)
    const unsigned code = A.ncvx_kind == LMC_NCVX_MC_TV ? 0x76354210u : (KT == 7 ? 0x75264310u : 0x76325410u);
    wave = (code >> (4 * hw_wave)) & 15;
  }
  if constexpr (K == 10 && !RT && CHAIN) wave = (0x67325410u >> (4 * hw_wave)) & 15;      // links of a fixed-count chain: L + T2 | T1 + T3 | T4 + N | T5 + C (ME-TV, 50 passes: 13.27 vs 13.95 ms)
  if constexpr (RT && K == 10) {
#ifndef LMC_RT_MAP
#define LMC_RT_MAP 0
#endif
    if (LMC_RT_SPREAD && LMC_RT_MAP_SPREAD >= 3 && kc <= NT) {
      // one live stage per wave (t_role: spread)
      // live TV waves beside L, N, C or a pass-through wave, never beside each other (searched on bench.py's data, where the chains settle at 4 passes -- 3 with
      // the 7-tap models: 1.504 / 1.511 ms per iteration against 1.591 / 1.566 for the plain order; the best pairing for one live count is among the worst for another)
      const unsigned code = kc >= 5 ? 0x76543210u : kc == 4 ? 0x67514320u      // kc = 4: L + T1 | T2 + T5 | T3 + N | T4 + C
                                                              : 0x64753210u;     // kc <= 3: L + T5 | T1 + N | T2 + T4 | T3 + C
      wave = (code >> (4 * hw_wave)) & 15;
    } else if (kc <= 4) {
      if (LMC_RT_MAP == 0) wave = hw_wave == 3 ? 6 : hw_wave == 4 ? 7 : hw_wave == 5 ? 4 : hw_wave == 6 ? 3 : hw_wave == 7 ? 5 : hw_wave;
      else if (LMC_RT_MAP == 1) wave = hw_wave == 4 ? 6 : hw_wave == 6 ? 4 : hw_wave;                                              // L + C | T1 + T5 | T2 + T4 | T3 + N
      else wave = hw_wave == 3 ? 6 : hw_wave == 4 ? 5 : hw_wave == 5 ? 4 : hw_wave == 6 ? 3 : hw_wave;                              // L + T5 | T1 + T4 | T2 + T3 | C + N
    }
    else if (CHAIN && LMC_RT_CHAIN_MAP) wave = ((LMC_RT_CHAIN_MAP == 2 ? 0x76345210u : 0x67254310u) >> (4 * hw_wave)) & 15;       // all stages live, a link of a chained prox: L + T5 | T1 + T2 | T3 + N | T4 + C (ME-TV as configured: 15.95 vs 16.73 ms)
    else wave = hw_wave == 4 ? 6 : hw_wave == 6 ? 4 : hw_wave;        // more live stages: L + C | T1 + T5 | T2 + T4 | T3 + N
  }
#ifdef LMC_EXP_PERM   // timing experiment (scripts/round3/perm_search.py): role of hardware wave w = nibble w of A.PH (the tile kernel's field, unused here)
  if (A.PH && (LMC_EXP_PERM == 1 || CHAIN)) wave = (A.PH >> (4 * hw_wave)) & 15;      // LMC_EXP_PERM=2: the chained links only
#endif
  // column strip of this workgroup (blockIdx.y; one strip = the whole row when W <= 64 PXL): c0 is a GLOBAL column, LDS rows are indexed by lane
  constexpr int HALO = pipe_halo(K, KT, PXL);
  const int strip = blockIdx.y;
  const int strip_u = gridDim.y > 1 ? BW - 2 * HALO : W;
  const int c0 = (strip ? strip * strip_u - HALO : 0) + lane * PXL;
  const int st_lo = strip * strip_u, st_hi = min(W, st_lo + strip_u);       // columns this workgroup writes
  constexpr bool al = AL;                                                    // rows are 16-byte aligned (strip_u and HALO are multiples of 4)
  const size_t img = (size_t)H * W;
  const float* __restrict__ xin = A.x_in + (size_t)chain * img;
  float* __restrict__ xout = A.x_out + (size_t)chain * img;

  for (int e = threadIdx.x; e < L::o_slab; e += blockDim.x) lds[e] = 0.f;   // ring rows < 0, hand-offs of tick -1, g
  __syncthreads();

  const int T_end = (H + D + 3) & ~3;          // ticks, rounded up to the unroll factor (extra ticks write nothing)
  float* const xring = lds + L::o_x;
  auto ring_row = [&](int row) -> float* { return xring + ((unsigned)(row + RB) % (unsigned)RB) * BW; };   // row >= -RB

#ifdef LMC_EXP_SKIP   // timing experiment (with LMC_EXP_NOBARRIER): the waves in the bitmask leave at once (results are wrong)
  if ((LMC_EXP_SKIP >> wave) & 1) return;
#endif
  // Issue arbitration on the shared SIMDs: the short, latency-bound waves everybody waits for at the barrier (L publishes the
  // ring row, C frees the hand-off slot) go first, the TV waves next, the Philox wave -- pure arithmetic, a quad row-group ahead
  // of its consumer -- last.  Measured: 2.04 -> 1.98 ms; the other way round (TV waves first) 2.37 ms.
#ifndef LMC_PRIO_L
#define LMC_PRIO_L 3
#define LMC_PRIO_C 3
#define LMC_PRIO_T 1
#define LMC_PRIO_N 0
#endif
  if (wave == 0) __builtin_amdgcn_s_setprio(LMC_PRIO_L);
  else if (wave == NT + 1) __builtin_amdgcn_s_setprio(LMC_PRIO_C);
  else if (wave <= NT) __builtin_amdgcn_s_setprio(LMC_PRIO_T);
  else __builtin_amdgcn_s_setprio(LMC_PRIO_N);
  if (wave == 0) {
    // ---------------- L: loader + blur gradient -------------------------------------------------------------
    const float* __restrict__ uv = A.blur.h;   // centred taps: u[0..KT) then v[0..KT) at h[kMaxBlur..]
    // the windows of the last KT-1 horizontally filtered rows: with KT = 5 they are rings indexed by (tick & 3), static under the
    // x4 unroll (row i-a in slot (U-a)&3, the new row replaces the oldest); otherwise they are rotated by moves
    constexpr bool kRing4 = (KT == 5);
    constexpr int NWIN = KT > 1 ? KT - 1 : 1;
    // x rows: fetched kXPF ticks ahead, slot (tick & 3) (8 ahead was measured: no gain, +50 VGPRs)
    constexpr int kXPF = 4;
    float xpre[kXPF][PXL], hxw[NWIN][PXL], hrw[NWIN][PXL], ypre[4][PXL];   // y rows: fetched kYPF ticks ahead, slot (tick & 3)
    constexpr int kYPF = KT == 7 ? 2 : 3;     // 7 taps: the windows already take 96 registers
#pragma unroll
    for (int a = 0; a < NWIN; ++a)
#pragma unroll
      for (int k = 0; k < PXL; ++k) { hxw[a][k] = 0.f; hrw[a][k] = 0.f; }
#pragma unroll
    for (int u = 0; u < kXPF; ++u) gload_raw<PXL>(xpre[u], xin + (size_t)min(u, H - 1) * W, c0, W, al);
    // Vector-memory loads return in order: waiting for a load also waits for every load issued before it.  So the loads a tick
    // consumes must be the OLDEST in flight: y rows are requested three ticks ahead and, inside a tick, before the x row that is only
    // needed four ticks later (with y one tick ahead and issued after x, every tick waited for a fresh HBM access: ~2000 cycles).
    if constexpr (KT > 0) {   // observation rows of the first kYPF residual rows
#pragma unroll
      for (int u = 0; u < kYPF; ++u) {
        const int r = u + 1 - D + (KT - 1) - HW;
        gload_raw<PXL>(ypre[u], A.y + (size_t)min(max(r, 0), H - 1) * W, c0, W, al);
      }
    }
    // Without a blur: pointwise data terms (identity, diagonal mask).  Their gradient sigma_f m (m x - y) of row t + 1 - D -- the row the
    // combine wave emits next tick -- is formed HERE (this wave issues no stores, so its loads never queue behind stores in vmcnt) from
    // the ring copy of x and the observation / mask rows requested kYPF ticks ahead, and handed over through the same o_g slots as the
    // blur gradient.
    const bool pw_id = KT == 0 && A.data_kind == LMC_DATA_IDENTITY, pw_mask = KT == 0 && A.data_kind == LMC_DATA_MASK;
    float mpre[KT == 0 ? 4 : 1][KT == 0 ? PXL : 1];
    if constexpr (KT == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int k = 0; k < PXL; ++k) { ypre[u][k] = 0.f; mpre[u][k] = 0.f; }
      if (pw_id || pw_mask) {
#pragma unroll
        for (int u = 0; u < kYPF; ++u) {
          const size_t ro = (size_t)min(max(u + 1 - D, 0), H - 1) * W;
          gload_raw<PXL>(ypre[u], A.y + ro, c0, W, al);
          if (pw_mask) gload_raw<PXL>(mpre[u], A.mask + ro, c0, W, al);
        }
      }
    }
    // chained launch: the dual state rows for stage 1, fetched two ticks ahead (row t - E - 1 is published at tick t)
    // (warm dual, A.tv_warm: the state is the projected dual (p, q) of the previous MYULA iteration, two fields; it enters stage 1 as
    // both the extrapolated and the projected iterate -- the momentum restarts, beta_1 = 0)
    constexpr int nsf = CHAIN ? (WARM ? 2 : 4) : 0;           // fields per pixel of the incoming state
    // state rows are fetched kSPF = 2 ticks ahead (4 for the two-field warm dual was measured: K = 1 / 2 / 3: 1.43 / 1.55 / 2.18 ms against
    // 1.44 / 1.41 / 2.13 -- the 32 extra VGPRs cost more than the deeper prefetch gains)
    constexpr int kSPF = 2;
    float spre[CHAIN ? kSPF : 1][CHAIN ? nsf : 1][CHAIN ? PXL : 1];
    const float* const sin = CHAIN && A.tv_in ? A.tv_in + (size_t)chain * nsf * img : nullptr;
    if constexpr (CHAIN) {
#pragma unroll
      for (int u = 0; u < kSPF; ++u) {
        const int rs = u - E - 1;
#pragma unroll
        for (int f = 0; f < nsf; ++f)      // raw: masked where the row is handed over (a select here would wait for the load at once)
          gload_raw<PXL>(spre[u][f], sin ? sin + (size_t)f * img + (size_t)min(max(rs, 0), H - 1) * W : xin, c0, W, al);
      }
    }
    double facc = 0.0;        // sum of squared residuals (A.f_out)
    auto tick = [&](auto uu, const int t) __attribute__((always_inline)) {
      constexpr int U = decltype(uu)::value, P = U & 1;
      if constexpr (KT > 0) {   // observation row of the residual row kYPF ticks from now
        const int r3 = t + kYPF + 1 - D + (KT - 1) - HW;
        gload_raw<PXL>(ypre[(U + kYPF) & 3], A.y + (size_t)min(max(r3, 0), H - 1) * W, c0, W, al);
      } else if (pw_id || pw_mask) {
        const size_t ro = (size_t)min(max(t + kYPF + 1 - D, 0), H - 1) * W;
        gload_raw<PXL>(ypre[(U + kYPF) & 3], A.y + ro, c0, W, al);
        if (pw_mask) gload_raw<PXL>(mpre[(U + kYPF) & 3], A.mask + ro, c0, W, al);
      }
      {   // row t arrives: publish it in the ring (zeros below the image); fetch row t + 4
        float xv[PXL];
        gfix_raw<PXL, AL>(xpre[U], c0, W);
#pragma unroll
        for (int k = 0; k < PXL; ++k) xv[k] = (t < H && c0 + k < W) ? xpre[U][k] : 0.f;
        prow_store<PXL>(ring_row(t), lane, xv);
        gload_raw<PXL>(xpre[U], xin + (size_t)min(t + kXPF, H - 1) * W, c0, W, al);
      }
      if constexpr (CHAIN) {   // dual state row t - E - 1 of the previous link -> stage 1's hand-off slot P (read next tick)
        float* hb = lds + L::o_hand0 + P * 4 * BW;
        constexpr int SP = U & (kSPF - 1);
        const int rh = t - E - 1;                                  // the row fetched kSPF ticks ago
        const bool rowok_h = sin && rh >= 0 && rh < H;
#pragma unroll
        for (int f = 0; f < nsf; ++f) {
          gfix_raw<PXL, AL>(spre[SP][f], c0, W);
          float sv[PXL];
#pragma unroll
          for (int k = 0; k < PXL; ++k) sv[k] = (rowok_h && c0 + (AL ? (k & ~3) : k) < W) ? spre[SP][f][k] : 0.f;
          prow_store<PXL>(hb + f * BW, lane, sv);
        }
        const int rs = t + kSPF - E - 1;
#pragma unroll
        for (int f = 0; f < nsf; ++f)
          gload_raw<PXL>(spre[SP][f], sin ? sin + (size_t)f * img + (size_t)min(max(rs, 0), H - 1) * W : xin, c0, W, al);
      }
      if constexpr (KT > 0) {
      const int i = t + 1 - D + (KT - 1);       // blur input row (<= t-1: published in an earlier tick)
      float hxn[PXL];
      {
        float xi[PXL], e[PXL + 2 * HW];
        prow_load<PXL>(xi, ring_row(i), lane);
#pragma unroll
        for (int m = 0; m < HW; ++m) e[m] = dpp_left0(xi[PXL - HW + m]);
#pragma unroll
        for (int k = 0; k < PXL; ++k) e[HW + k] = xi[k];
#pragma unroll
        for (int m = 0; m < HW; ++m) e[HW + PXL + m] = dpp_right0(xi[m]);
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
          float acc = uv[kMaxBlur] * e[k + 2 * HW];
#pragma unroll
          for (int b = 1; b < KT; ++b) acc = fmaf(uv[kMaxBlur + b], e[k + 2 * HW - b], acc);
          hxn[k] = acc;
        }
      }
      const int r = i - HW;                     // residual row: Hx[r] = sum_a u[a] hx[i - a]
      float R[PXL];
      {
        const bool rowok = r >= 0 && r < H;
        const float rmask = rowok ? 1.f : 0.f;
        gfix_raw<PXL, AL>(ypre[U & 3], c0, W);
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
          float acc = uv[0] * hxn[k];
#pragma unroll
          for (int a = 1; a < KT; ++a) acc = fmaf(uv[a], hxw[kRing4 ? ((U - a) & 3) : a - 1][k], acc);
          // masked by a factor, not a select: a select on (row, column) turns into one exec-masked block per pixel (8 per tick: the wave's longest
          // stretch of unpacked arithmetic and half of its scalar instructions); every operand is finite (clamped rows, zeroed ring rows)
          R[k] = (acc - ypre[U & 3][k]) * ((AL ? c0 < W : c0 + k < W) ? rmask : 0.f);
        }
        if (A.f_out) {
#pragma unroll
          for (int k = 0; k < PXL; ++k) facc = fma((double)R[k], (double)R[k], facc);
        }
        if constexpr (!kRing4) {
#pragma unroll
          for (int a = KT - 2; a >= 1; --a)
#pragma unroll
            for (int k = 0; k < PXL; ++k) hxw[a][k] = hxw[a - 1][k];
        }
#pragma unroll
        for (int k = 0; k < PXL; ++k) hxw[kRing4 ? (U & 3) : 0][k] = hxn[k];
      }
      {   // horizontal adjoint, then G[r - HW] = sum_a u[a] hR[r - 2HW + a]
        float e[PXL + 2 * HW], gout[PXL];
#pragma unroll
        for (int m = 0; m < HW; ++m) e[m] = dpp_left0(R[PXL - HW + m]);
#pragma unroll
        for (int k = 0; k < PXL; ++k) e[HW + k] = R[k];
#pragma unroll
        for (int m = 0; m < HW; ++m) e[HW + PXL + m] = dpp_right0(R[m]);
#pragma unroll
        for (int k = 0; k < PXL; ++k) {
          float hrn = uv[kMaxBlur] * e[k];
#pragma unroll
          for (int b = 1; b < KT; ++b) hrn = fmaf(uv[kMaxBlur + b], e[k + b], hrn);
          float acc = uv[KT - 1] * hrn;
#pragma unroll
          for (int a = 0; a < KT - 1; ++a) acc = fmaf(uv[a], hrw[kRing4 ? ((U - (KT - 1 - a)) & 3) : KT - 2 - a][k], acc);
          if constexpr (!kRing4) {
#pragma unroll
            for (int a = KT - 2; a >= 1; --a) hrw[a][k] = hrw[a - 1][k];
          }
          hrw[kRing4 ? (U & 3) : 0][k] = hrn;
          gout[k] = A.sigma_f * acc;
        }
        prow_store<PXL>(lds + L::o_g + P * BW, lane, gout);      // row t + 1 - D, read by C next tick
      }
      }   // KT > 0
      if constexpr (KT == 0) {
        if (pw_id || pw_mask) {
          const int i = t + 1 - D;                // <= t - 1: published in an earlier tick
          float xi[PXL], gout[PXL];
          prow_load<PXL>(xi, ring_row(i), lane);
          const bool rowok = i >= 0 && i < H;
          gfix_raw<PXL, AL>(ypre[U & 3], c0, W);
          if (pw_mask) gfix_raw<PXL, AL>(mpre[U & 3], c0, W);
#pragma unroll
          for (int k = 0; k < PXL; ++k) {
            float g = 0.f;
            if (rowok && c0 + k < W) {
              if (pw_id) g = A.sigma_f * (xi[k] - ypre[U & 3][k]);
              else g = A.sigma_f * mpre[U & 3][k] * fmaf(mpre[U & 3][k], xi[k], -ypre[U & 3][k]);
            }
            gout[k] = g;
          }
          prow_store<PXL>(lds + L::o_g + P * BW, lane, gout);      // row t + 1 - D, read by C next tick
        }
      }
      PIPE_TICK_SYNC();
    };
    for (int t = 0; t < T_end; t += 4) static_for<0, 4>([&](auto uu) { tick(uu, t + decltype(uu)::value); });
    if (A.f_out) {
      const double tot = wave_sum(facc);
      if (lane == 0) unsafeAtomicAdd(&A.f_out[chain], 0.5 * (double)A.sigma_f * tot);
    }
  } else if (wave <= NT) {
    // ---------------- T: TV stages k1 = 2*wave - 1 and k2 = 2*wave -----------------------------------------
    // Wave 1 of a launch that starts from the zero dual state runs a specialised stage 1 (its own copy of the loop, no branch inside):
    // it shares a SIMD with wave 5, and two full T waves on one SIMD are what bounds the tick.
    auto t_role = [&](auto first_tag, auto single_tag) __attribute__((always_inline)) {
    constexpr bool FIRST = decltype(first_tag)::value;      // stage k1 starts from the zero dual state
    constexpr bool SINGLE = decltype(single_tag)::value;    // odd K, last wave: stage k1 = K only, its output goes straight to the hand-off
    const int k1 = 2 * wave - 1, k2 = 2 * wave;        // the wave's two SLOTS of the row schedule (rows a1, a2 below)
    // RT, at most NT live stages (the chain the reference configures settles at 4): ONE live stage per wave -- slot k1 of wave j runs stage j, slot k2 only
    // hands its state on -- instead of two stages each in the first waves and none in the rest: a wave's tick is then half as long, and what the others wait for at
    // the barrier is no longer the two-stage waves' chain of hand-off reads, two stages and stores.  Same arithmetic on the same data: bit-identical.
    const bool spread = RT && LMC_RT_SPREAD && kc <= NT;
    const int g1 = spread ? wave : k1, g2 = k2;          // the STAGES the slots run (momentum coefficient, objective slot)
    const float gam = A.tv.gamma, cstep = A.tv.c;
    const float beta1 = A.tv.betas[g1 - 1], beta2 = SINGLE ? 0.f : A.tv.betas[g2 - 1];
    PipeCr<PXL / 2> crc;                                              // see PipeCr: AL kernels need W % PXL == 0 (host check), the others take any W
    crc.cstep = cstep;
    crc.cr_last = (c0 + PXL - 1 == W - 1) ? 0.f : cstep;
#pragma unroll
    for (int i = 0; i < PXL / 2; ++i) crc.ncrv[i] = v2f{c0 + 2 * i == W - 1 ? 0.f : -cstep, c0 + 2 * i + 1 == W - 1 ? 0.f : -cstep};
    float* const hout = lds + L::o_hand + (wave - 1) * 8 * BW;       // this wave's hand-off [2][4][BW] ([2][2][BW] for the last one if CHAIN)
    const bool from_state = CHAIN && wave == 1 && A.tv_in != nullptr;
    constexpr bool warm = WARM;
    const float* const hin = from_state ? lds + L::o_hand0 : hout - 8 * BW;   // the previous wave's / the previous link's state
    constexpr int nof = warm ? 2 : 4;                                // fields per pixel of the outgoing state
    float* const sout = CHAIN && wave == NT && A.tv_out && (!RT || state_only) ? A.tv_out + (size_t)chain * nof * img : nullptr;
    constexpr int NP = PXL / 2;
    const bool live1 = !RT || g1 <= kc, live2 = !RT || (!spread && g2 <= kc);      // wave-uniform; live stages are a prefix
    const bool fullpass2 = spread && wave < kc;          // a later wave still runs a live stage: slot k2 hands all four fields on, not (rr, ss) only
    ObjMask<NP> om;
    om.md = 0.f;
    om.mlast = v2f{1.f, (c0 + PXL - 1 == W - 1) ? 0.f : 1.f};
    double osq1 = 0.0, otv1 = 0.0, osq2 = 0.0, otv2 = 0.0;            // RT: objective sums of the iterates stages k1 / k2 form (fp64 over rows)
    DualRow<NP> inb[2], o1[2];
    v2f sol1[NP], sol2[NP];
    v2f xk[2][NP];        // x rows read for stage k1 (row a1 = a2 + 2), reused by stage k2 two ticks later: one ring read per tick
#pragma unroll
    for (int k = 0; k < NP; ++k) {
      sol1[k] = sol2[k] = pk_set(0.f);
      xk[0][k] = xk[1][k] = pk_set(0.f);
#pragma unroll
      for (int pp = 0; pp < 2; ++pp) {
        inb[pp].rr[k] = inb[pp].ss[k] = inb[pp].p[k] = inb[pp].q[k] = pk_set(0.f);
        o1[pp].rr[k] = o1[pp].ss[k] = o1[pp].p[k] = o1[pp].q[k] = pk_set(0.f);
      }
    }
    // the wave's last stage hands its output (dual state of row `brow`) to the next wave / the combine wave, and, in the last wave of a
    // chained launch, to HBM for the next link (all four fields) or the next MYULA iteration (warm dual: p, q)
    auto emit = [&](const DualRow<NP>& out, const int P, const int brow, const bool live) __attribute__((always_inline)) {
      if (RT && !live) {             // pass-through: only (rr, ss) travel on (the combine wave's operands); same slots in both hand-off layouts
        float* hb = hout + P * ((!CHAIN || wave < NT) ? 4 : 2) * BW;
        pairs_store<NP>(hb, lane, out.rr);
        pairs_store<NP>(hb + BW, lane, out.ss);
      } else if (!CHAIN || wave < NT) {
        float* hb = hout + P * 4 * BW;
#ifdef LMC_EXP_NO_HSTORE     // timing experiment: the hand-off stores never execute (results are wrong), the arithmetic stays alive
        if (A.tv.niter == 12345)
#endif
        {
        pairs_store<NP>(hb, lane, out.rr);
        pairs_store<NP>(hb + BW, lane, out.ss);
#ifdef LMC_EXP_HALF_HSTORE   // timing experiment: half the hand-off stores
        if (A.tv.niter == 12345)
#endif
        if (wave < NT) {     // the combine wave reads rr, ss only: the last TV wave (which shares its SIMD with T1) skips half of its hand-off stores
        pairs_store<NP>(hb + 2 * BW, lane, out.p);
        pairs_store<NP>(hb + 3 * BW, lane, out.q);
        }
        }
      } else {
        float* hb = hout + P * 2 * BW;                   // last boundary of a chained launch: rr, ss for the final primal step ...
        pairs_store<NP>(hb, lane, out.rr);
        pairs_store<NP>(hb + BW, lane, out.ss);
        if (sout && brow >= 0 && brow < H) {             // ... and the dual state of row brow for the next link / iteration
#pragma unroll
          for (int g = 0; g < NP / 2; ++g) {
            float* d = sout + (size_t)brow * W;
            const int cg = c0 + 4 * g;
            if (warm) {
              gstore4(d, cg, st_lo, st_hi, al, out.p[2 * g].x, out.p[2 * g].y, out.p[2 * g + 1].x, out.p[2 * g + 1].y);
              gstore4(d + img, cg, st_lo, st_hi, al, out.q[2 * g].x, out.q[2 * g].y, out.q[2 * g + 1].x, out.q[2 * g + 1].y);
            } else {
              gstore4(d, cg, st_lo, st_hi, al, out.rr[2 * g].x, out.rr[2 * g].y, out.rr[2 * g + 1].x, out.rr[2 * g + 1].y);
              gstore4(d + img, cg, st_lo, st_hi, al, out.ss[2 * g].x, out.ss[2 * g].y, out.ss[2 * g + 1].x, out.ss[2 * g + 1].y);
              gstore4(d + 2 * img, cg, st_lo, st_hi, al, out.p[2 * g].x, out.p[2 * g].y, out.p[2 * g + 1].x, out.p[2 * g + 1].y);
              gstore4(d + 3 * img, cg, st_lo, st_hi, al, out.q[2 * g].x, out.q[2 * g].y, out.q[2 * g + 1].x, out.q[2 * g + 1].y);
            }
          }
        }
      }
    };
    auto tick = [&](auto uu, const int t) __attribute__((always_inline)) {
      constexpr int P = decltype(uu)::value & 1;
      const int a2 = t - E - 2 * k2, a1 = t - E - 2 * k1;
      if constexpr (!SINGLE) {   // stage k2 on row a2: inputs are this wave's stage k1, one tick (row a2) and two ticks (row a2-1) old
        const float cdown = ((unsigned)(a2 - 1) >= (unsigned)(H - 1)) ? 0.f : cstep;
        DualRow<NP> out;
        if (live2) {
          StageObj ob;
          om.md = cdown != 0.f ? 1.f : 0.f;
          pipe_stage<NP, AL, RT>(xk[P], o1[P ^ 1].rr, o1[P ^ 1].ss, o1[P], sol2, gam, cdown, crc, beta2, out, &ob, &om);
          if constexpr (RT) { osq2 += (double)(ob.sq.x + ob.sq.y); otv2 += (double)(ob.tv.x + ob.tv.y); }
        } else {                 // pass-through: the state of row a2 - 1 as stage k1 left it
#pragma unroll
          for (int k = 0; k < NP; ++k) { out.rr[k] = o1[P].rr[k]; out.ss[k] = o1[P].ss[k]; }
          if (fullpass2) {
#pragma unroll
            for (int k = 0; k < NP; ++k) { out.p[k] = o1[P].p[k]; out.q[k] = o1[P].q[k]; }
          }
        }
        emit(out, P, a2 - 1, live2 || fullpass2);
      }
      {   // stage k1 on row a1: inputs from the previous wave's hand-off (row a1) and the one read a tick earlier (row a1-1)
        if constexpr (!FIRST) {
          if (k1 > 1 || from_state) {
            const float* hb = hin + (P ^ 1) * 4 * BW;
            pairs_load<NP>(inb[P].rr, hb, lane);
            pairs_load<NP>(inb[P].ss, hb + BW, lane);
            if (from_state && warm) {     // warm dual: the state IS the projected iterate (beta_1 = 0 makes its role as p_old void)
#pragma unroll
              for (int k = 0; k < NP; ++k) { inb[P].p[k] = inb[P].rr[k]; inb[P].q[k] = inb[P].ss[k]; }
            } else if (live1) {           // (a pass-through stage moves rr, ss only)
              pairs_load<NP>(inb[P].p, hb + 2 * BW, lane);
              pairs_load<NP>(inb[P].q, hb + 3 * BW, lane);
            }
          }
        }
        if (live1) {
          pairs_load<NP>(xk[P], ring_row(a1), lane);      // read two ticks ago as row a1 = this tick's a2: consumed above
          const float cdown = ((unsigned)(a1 - 1) >= (unsigned)(H - 1)) ? 0.f : cstep;
          StageObj ob;
          om.md = cdown != 0.f ? 1.f : 0.f;
          if constexpr (FIRST) pipe_stage_first<NP, AL, RT>(xk[P], sol1, cdown, crc, beta1, o1[P], &ob, &om);
          else pipe_stage<NP, AL, RT>(xk[P], inb[P].rr, inb[P].ss, inb[P ^ 1], sol1, gam, cdown, crc, beta1, o1[P], &ob, &om);
          if constexpr (RT) { osq1 += (double)(ob.sq.x + ob.sq.y); otv1 += (double)(ob.tv.x + ob.tv.y); }
        } else {                 // pass-through: the state of row a1 - 1, read one tick ago
#pragma unroll
          for (int k = 0; k < NP; ++k) { o1[P].rr[k] = inb[P ^ 1].rr[k]; o1[P].ss[k] = inb[P ^ 1].ss[k]; }
        }
        if constexpr (SINGLE) emit(o1[P], P, a1 - 1, live1);
      }
      PIPE_TICK_SYNC();
    };
    for (int t = 0; t < T_end; t += 4) static_for<0, 4>([&](auto uu) { tick(uu, t + decltype(uu)::value); });
    if constexpr (RT) {          // objectives of the iterates this wave's live stages formed: sol^{g-1} in stage g = rt_base + k
      const double dg = (double)gam;
      double* const ob = A.rt_obj + (size_t)chain * A.rt_stride + A.rt_base;
      if (live1) {
        const double tot = wave_sum(0.5 * dg * dg * osq1 + dg * otv1);
        if (lane == 0) unsafeAtomicAdd(ob + (g1 - 1), tot);
      }
      if (!SINGLE && live2) {
        const double tot = wave_sum(0.5 * dg * dg * osq2 + dg * otv2);
        if (lane == 0) unsafeAtomicAdd(ob + (g2 - 1), tot);
      }
    }
    };   // t_role
    constexpr bool kOdd = (K & 1) != 0;
    bool ran = false;
    if constexpr (!CHAIN) {
      if (wave == 1) {
        if constexpr (K == 1) t_role(std::true_type{}, std::true_type{});
        else t_role(std::true_type{}, std::false_type{});
        ran = true;
      }
    }
    if constexpr (kOdd) {
      if (!ran && wave == NT) { t_role(std::false_type{}, std::true_type{}); ran = true; }
    }
    if constexpr (K > 1) {
      if (!ran) t_role(std::false_type{}, std::false_type{});
    }
  } else if (wave == NT + 2) {
    // ---------------- N: Philox normals, one quad row-group ahead of C -------------------------------------
    // In the tick of row 4q + NI the normals of pixels NI*PXL/4 .. of quad q + 1 are drawn into the other half of the slab
    // (spread evenly over the ticks: a burst every 4th tick would stall every wave at the barrier).
    float* const slab = lds + L::o_slab + lane;        // normal (row q of the quad, pixel k) at slab[(q*PXL + k)*64]
    const uint32_t iter = A.iteration + (A.iter_dev ? *A.iter_dev : 0u);     // uniform; graph replays advance *iter_dev
    auto tick = [&](auto uu, const int t) __attribute__((always_inline)) {
      constexpr int U = decltype(uu)::value;
      constexpr int NI = ((U - D) % 4 + 4) % 4;        // == o & 3  (t = 4m + U)
      const int o = t - D;
      if (A.noise_mode == LMC_NOISE_PHILOX && !state_only) {
        const int qn = ((o - NI) >> 2) + 1;             // quad row-group being prepared (o - NI is a multiple of 4)
        if (qn >= 0 && 4 * qn < H) {
          float* const sl = slab + (qn & 1) * (4 * PXL * 64);
#pragma unroll
          for (int kk = 0; kk < PXL / 4; ++kk) {
            const int k = NI * (PXL / 4) + kk;
            float n4[4];
            quad_normals(A.key0, A.key1, iter, A.chain_offset + (uint32_t)chain, (uint32_t)qn * (uint32_t)W + (uint32_t)(c0 + k), n4);
#pragma unroll
            for (int q = 0; q < 4; ++q) sl[(q * PXL + k) * 64] = n4[q];
          }
        }
      }
      PIPE_TICK_SYNC();
    };
    for (int t = 0; t < T_end; t += 4) static_for<0, 4>([&](auto uu) { tick(uu, t + decltype(uu)::value); });
  } else {
    // ---------------- C: final primal step, combine, store --------------------------------------------------
    // Two copies of the role (like the T role's): VM = this wave issues global LOADS (rows of the ME-TV prox image, injected noise).  Memory operations
    // retire in order through one counter, so wherever a conditional load merges back the compiler waits for vmcnt(0) -- in a wave that also stores, that
    // is a wait for its own stores of the previous tick, every tick (counters of the combine wave running alone: 55 % of its cycles in s_waitcnt, 529 ns
    // per tick for 126 ns of arithmetic).  The copy without loads has no such wait.
    auto c_role = [&](auto vm_tag) __attribute__((always_inline)) {
    constexpr bool VM = decltype(vm_tag)::value;
    const float gam = A.tv.gamma;
    const float* const hin = lds + L::o_hand + (NT - 1) * 8 * BW;      // [2][4][BW], or [2][2][BW] in a chained launch
    constexpr int HSTR = CHAIN ? 2 * BW : 4 * BW;
    float* const slab = lds + L::o_slab + lane;        // normal (row q of the quad, pixel k) at slab[(q*PXL + k)*64]
    double gacc = 0.0;        // sum |grad x_in| (A.g_out)
    float crr[2][PXL], xprev[PXL];
#pragma unroll
    for (int k = 0; k < PXL; ++k) crr[0][k] = crr[1][k] = xprev[k] = 0.f;
    // RT: primal objective of the iterate this wave returns, sol^kc (the one the exit test of pass kc looks at; not needed when kc is the
    // last pass, whose iterate is returned untested): sum of squared divergences row by row, |grad sol| of row o - 1 once row o is known
    const bool want_obj = RT && !state_only && A.rt_base + kc < A.rt_total;
    double osq = 0.0, otv = 0.0;
    float pprev[RT ? PXL : 1];
#pragma unroll
    for (int k = 0; k < (RT ? PXL : 1); ++k) pprev[k] = 0.f;
    // rows of the ME-TV term's prox image (A.extra), requested three ticks ahead of their use (slot tick & 3): a load issued at
    // its point of use would expose an HBM access per tick
    float exq[4][PXL];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int k = 0; k < PXL; ++k) exq[u][k] = 0.f;
    if (VM && A.extra) {
#pragma unroll
      for (int u = 0; u < 3; ++u) {
        const int r = u - D;
        gload_raw<PXL>(exq[u], A.extra + (size_t)chain * img + (size_t)min(max(r, 0), H - 1) * W, c0, W, al);
      }
    }
    auto tick = [&](auto uu, const int t) __attribute__((always_inline)) {
      constexpr int U = decltype(uu)::value, P = U & 1;
      constexpr int NI = ((U - D) % 4 + 4) % 4;        // == o & 3  (t = 4m + U)
      const int o = t - D;
      if (state_only) { PIPE_TICK_SYNC(); return; }     // this link only advances the dual state (of this chain)
      if (VM && A.extra) {
        const int r3 = o + 3;
        gload_raw<PXL>(exq[(U + 3) & 3], A.extra + (size_t)chain * img + (size_t)min(max(r3, 0), H - 1) * W, c0, W, al);
      }
      float css[PXL], xo[PXL], gv[PXL], prox[PXL];
      prow_load<PXL>(crr[P], hin + (P ^ 1) * HSTR, lane);            // rr^K on row o (written last tick)
      prow_load<PXL>(css, hin + (P ^ 1) * HSTR + BW, lane);
      prow_load<PXL>(xo, ring_row(o), lane);
      if (KT > 0 || A.data_kind == LMC_DATA_IDENTITY || A.data_kind == LMC_DATA_MASK) {
        prow_load<PXL>(gv, lds + L::o_g + (P ^ 1) * BW, lane);
      } else {                                  // no data term: o_g is never written (stale LDS could hold NaN bit patterns)
#pragma unroll
        for (int j = 0; j < PXL; ++j) gv[j] = 0.f;
      }
      const float ssl0 = dpp_left0(css[PXL - 1]);
      float dvs = 0.f;
#pragma unroll
      for (int j = 0; j < PXL; ++j) {
        const float ssl = j == 0 ? ssl0 : css[j - 1];
        const float dv = (crr[P][j] - crr[P ^ 1][j]) + (css[j] - ssl);
        prox[j] = fmaf(-gam, dv, xo[j]);
        if constexpr (RT) dvs = fmaf(dv, dv, dvs);
      }
      if constexpr (RT) {
        if (want_obj && o >= 0 && o < H) {
          osq += (double)dvs;
          if (o >= 1) {               // row o - 1: the row below it is this one
            const float pr_last = dpp_right0(pprev[0]);
            float tvs = 0.f;
#pragma unroll
            for (int j = 0; j < PXL; ++j) {
              const float dx = prox[j] - pprev[j];
              const float dy = (c0 + j + 1 < W) ? (j == PXL - 1 ? pr_last : pprev[j + 1]) - pprev[j] : 0.f;
              tvs += __builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
            }
            otv += (double)tvs;
          }
#pragma unroll
          for (int j = 0; j < PXL; ++j) pprev[j] = prox[j];
        }
      }
      if (A.ncvx_kind == LMC_NCVX_MC_TV) {   // - lambda * A^T(A x / max(|A x|, gamma))  (algs.py:273-277, 291), added to the gradient
        // v = A x / max(|A x|, gamma) is formed ONCE per pixel, row by row: (vx, vy) of row o from ring rows o, o + 1; A^T v at (o, j) = -((vx[o][j] -
        // vx[o-1][j]) + (vy[o][j] - vy[o][j-1])) with vx of the previous row kept in registers.  The same values, operation for operation, as
        // mc_tv_grad (lmc_device.h) recomputes per pixel for its three weights -- a third of the square roots and reciprocals (round 3: the
        // per-pixel form made this wave the slowest of the workgroup, 3.0 ms per launch against 1.75 without the term).
        float xp[PXL], vx[PXL], vy[PXL];
        prow_load<PXL>(xp, ring_row(o + 1), lane);
        const float x0_r = dpp_right0(xo[0]);
        const bool an = A.ncvx_gamma < 0.f, rowin = o >= 0 && o < H, down = o + 1 < H;
        const float gth = fabsf(A.ncvx_gamma);
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
          const float dx = (rowin && down) ? xp[j] - xo[j] : 0.f;
          const float dy = (rowin && c0 + j + 1 < W) ? (j == PXL - 1 ? x0_r : xo[j + 1]) - xo[j] : 0.f;
          if (an) {
            vx[j] = __builtin_amdgcn_rcpf(fmaxf(fabsf(dx), gth)) * dx;
            vy[j] = __builtin_amdgcn_rcpf(fmaxf(fabsf(dy), gth)) * dy;
          } else {
            const float w = __builtin_amdgcn_rcpf(fmaxf(__builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy)), gth));
            vx[j] = w * dx;
            vy[j] = w * dy;
          }
        }
        const float vy_l = dpp_left0(vy[PXL - 1]);
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
          gv[j] -= A.ncvx_lambda * -((vx[j] - xprev[j]) + (vy[j] - (j == 0 ? vy_l : vy[j - 1])));
          xprev[j] = vx[j];          // (xprev: vx of the previous row)
        }
      }
      if (A.g_out && o >= 0 && o < H) {   // isotropic TV of the input image, row o: forward differences, zero across the last row / column
        float xq[PXL];
        prow_load<PXL>(xq, ring_row(o + 1), lane);
        const float xr_last = dpp_right0(xo[0]);
        const bool down = o + 1 < H;
#pragma unroll
        for (int j = 0; j < PXL; ++j) {
          const float dx = down ? xq[j] - xo[j] : 0.f;
          const float dy = (c0 + j + 1 < W) ? (j == PXL - 1 ? xr_last : xo[j + 1]) - xo[j] : 0.f;
          gacc += (double)__builtin_amdgcn_sqrtf(fmaf(dx, dx, dy * dy));
        }
      }
      if (o >= 0 && o < H) {
        const float* const slr = slab + ((o >> 2) & 1) * (4 * PXL * 64);
        const size_t go = (size_t)o * W;
#pragma unroll
        for (int g = 0; g < PXL / 4; ++g) {
          if (c0 + 4 * g < st_hi && c0 + 4 * g + 3 >= st_lo) {      // the group touches this workgroup's interior
            float xi[4] = {0.f, 0.f, 0.f, 0.f}, ex[4] = {0.f, 0.f, 0.f, 0.f};
            if (A.noise_mode == LMC_NOISE_PHILOX) {
#pragma unroll
              for (int q = 0; q < 4; ++q) xi[q] = slr[(NI * PXL + 4 * g + q) * 64];
            } else if (VM && A.noise_mode == LMC_NOISE_INJECTED) {
              const float* nrow = A.noise + (size_t)chain * img + go;
              if (al) {
                const float4 v = *reinterpret_cast<const float4*>(nrow + c0 + 4 * g);
                xi[0] = v.x; xi[1] = v.y; xi[2] = v.z; xi[3] = v.w;
              } else load4_dword_aligned(xi[0], xi[1], xi[2], xi[3], nrow, c0 + 4 * g, W);
            }
            if (VM && A.extra) {
              ex[0] = exq[U][4 * g]; ex[1] = exq[U][4 * g + 1]; ex[2] = exq[U][4 * g + 2]; ex[3] = exq[U][4 * g + 3];
              if constexpr (!AL) unshift4_dword_aligned(ex[0], ex[1], ex[2], ex[3], c0 + 4 * g, W);
            }
            float ov[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
              const float x = xo[4 * g + q];
              float gr = gv[4 * g + q];
              if (VM && A.extra) gr = fmaf(A.extra_coef, x - ex[q], gr);
              ov[q] = fmaf(A.a, x, fmaf(-A.t, gr, fmaf(A.b, prox[4 * g + q], A.s * xi[q])));
            }
            gstore4(xout + go, c0 + 4 * g, st_lo, st_hi, al, ov[0], ov[1], ov[2], ov[3]);
          }
        }
      }
      PIPE_TICK_SYNC();
    };
    for (int t = 0; t < T_end; t += 4) static_for<0, 4>([&](auto uu) { tick(uu, t + decltype(uu)::value); });
    if (A.g_out) {
      const double tot = wave_sum(gacc);
      if (lane == 0) unsafeAtomicAdd(&A.g_out[chain], (double)A.g_scale * tot);
    }
    if constexpr (RT) {
      if (want_obj) {               // the last image row: no row below it, horizontal differences only
        const float pr_last = dpp_right0(pprev[0]);
        float tvs = 0.f;
#pragma unroll
        for (int j = 0; j < PXL; ++j) tvs += (c0 + j + 1 < W) ? fabsf((j == PXL - 1 ? pr_last : pprev[j + 1]) - pprev[j]) : 0.f;
        otv += (double)tvs;
        const double dg = (double)gam;
        const double tot = wave_sum(0.5 * dg * dg * osq + dg * otv);
        if (lane == 0) unsafeAtomicAdd(A.rt_obj + (size_t)chain * A.rt_stride + A.rt_base + kc, tot);
      }
    }
    };   // c_role
    if (A.extra != nullptr || A.noise_mode == LMC_NOISE_INJECTED) c_role(std::true_type{});
    else c_role(std::false_type{});
  }
}


template <int K, int PXL, int KT, bool CHAIN = false>
static constexpr size_t pipe_lds_bytes() { return sizeof(float) * (size_t)PipeLds<K, PXL, CHAIN>::total; }

template <int PXL, int KT, bool CHAIN, int K = 10, bool WARM = false, bool AL = true, bool RT = false>
static hipError_t pipe_launch_one(const StepArgs& a, hipStream_t st) {
  auto kern = myula_step_pipe_kernel<K, PXL, KT, CHAIN, WARM, AL, RT>;
  constexpr size_t lb = pipe_lds_bytes<K, PXL, KT, CHAIN>();
  static bool attr_set[64] = {};        // per device: the attribute belongs to the function's code object on that device
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (dev < 0 || dev >= 64 || !attr_set[dev]) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lb);
    if (e != hipSuccess) return e;
    if (dev >= 0 && dev < 64) attr_set[dev] = true;
  }
  const int BWk = 64 * PXL, U = BWk - 2 * pipe_halo(K, KT, PXL);
  const int nstrips = a.W <= BWk ? 1 : (a.W + U - 1) / U;       // wider than one wave: column strips with recomputed halos
#ifdef LMC_EXP_PERM
  StepArgs b = a;
  if (const char* e = getenv("LMC_EXP_PERM")) b.PH = (int)strtoul(e, nullptr, 16);
  hipLaunchKernelGGL(kern, dim3(a.C, nstrips), dim3(64 * ((K + 1) / 2 + 3)), lb, st, b);
  return hipGetLastError();
#endif
  hipLaunchKernelGGL(kern, dim3(a.C, nstrips), dim3(64 * ((K + 1) / 2 + 3)), lb, st, a);
  return hipGetLastError();
}

// one launch with K dual iterations: blur taps 5 / 7 / none (KT), 8 pixels per lane above 256 columns, else 4
// rows not 16-byte aligned (W % 4 != 0): the pixel-by-pixel instantiations, K = 10 only (see pipe_links)
template <int K, bool CHAIN>
static hipError_t pipe_dispatch_unaligned(const StepArgs& a, int KT, hipStream_t st) {
  if (a.W > 256) {
    if (KT == 5) return pipe_launch_one<8, 5, CHAIN, K, false, false>(a, st);
    if (KT == 7) return pipe_launch_one<8, 7, CHAIN, K, false, false>(a, st);
    return pipe_launch_one<8, 0, CHAIN, K, false, false>(a, st);
  }
  if (KT == 5) return pipe_launch_one<4, 5, CHAIN, K, false, false>(a, st);
  if (KT == 7) return pipe_launch_one<4, 7, CHAIN, K, false, false>(a, st);
  return pipe_launch_one<4, 0, CHAIN, K, false, false>(a, st);
}

template <int K, bool CHAIN, bool WARM = false>
static hipError_t pipe_dispatch_k(const StepArgs& a, int KT, hipStream_t st) {
  const bool lastlane = (a.W & (a.W > 256 ? 7 : 3)) == 0;       // the last image column is the last pixel of a lane, rows are 16-byte aligned
  if constexpr (K == 10 && !WARM) {
    if (!lastlane) return pipe_dispatch_unaligned<K, CHAIN>(a, KT, st);
  }
  if (!lastlane) return hipErrorInvalidConfiguration;
  if (a.W > 256) {
    if (KT == 5) return pipe_launch_one<8, 5, CHAIN, K, WARM>(a, st);
    if (KT == 7) return pipe_launch_one<8, 7, CHAIN, K, WARM>(a, st);
    return pipe_launch_one<8, 0, CHAIN, K, WARM>(a, st);
  }
  if (KT == 5) return pipe_launch_one<4, 5, CHAIN, K, WARM>(a, st);
  if (KT == 7) return pipe_launch_one<4, 7, CHAIN, K, WARM>(a, st);
  return pipe_launch_one<4, 0, CHAIN, K, WARM>(a, st);
}

// lmc_step_pipe.hip: geometry / data term the pipe kernels cover, and the centred taps (returns KT)
bool pipe_geometry_ok(const StepArgs& a);
int pipe_taps(StepArgs& a);

// lmc_step_pipe_chain.hip: the CHAIN instantiations (dual state in / out through HBM): links of a chained launch (K = 9, 10) and the
// warm-started prox (a.tv_warm: K = 1, 2, 3)
hipError_t pipe_dispatch_chain(const StepArgs& a, int K, int KT, hipStream_t st);

}  // namespace lmc
