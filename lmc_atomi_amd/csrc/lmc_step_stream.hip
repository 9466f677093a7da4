// Fused MYULA update, streaming variant ("stream"): one workgroup = one chain, lane = image column,
// the image streams through the workgroup top to bottom, one row per tick.
//
//   out = a*x - t*grad f(x) + b*prox_{TV}(x) + s*xi           (algs.py:569)
//
// The K fast-gradient-projection iterations of the TV prox are laid out as a software pipeline over
// rows: dual iteration k works 2 rows behind iteration k-1, so at tick t stage k computes the primal
// iterate sol^k on row a_k = t - E - 2k ("A") and the projected dual on row a_k - 1 ("B").  All
// per-column state (dual rr/ss, projections p/q, sol; two ticks deep) lives in VGPRs, indexed by
// tick parity so nothing is ever moved.  Horizontal neighbours come from DPP wave shifts inside a
// wavefront and from a 2-word-per-stage LDS ghost exchange between wavefronts; ONE barrier per tick.
// Compared with overlapped tiles this recomputes nothing: redundancy is the pipeline fill
// (D = max(2K+2,10) rows per image) instead of ((T+2K)/T)^2.
//
// x lives in an LDS ring of RB = D+2 rows, stored twice (slots s and s+RB) so that every read of
// row t-c is `base(t) + immediate` with no wrap arithmetic.  The blur gradient
// sigma_f * H^T(Hx - y) (separable taps, zero-padded to KT) runs beside the TV pipeline: the two
// horizontal passes read LDS rows with immediate offsets, the two vertical passes use register
// windows.  Noise: Philox4x32-10 + Box-Muller, one call per lane every 4 rows.
// Ticks whose rows all lie strictly inside the image run a predicate-free body (EDGE = false).
// HBM traffic: x read once, x' written once (8 B per pixel per iteration).
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

#ifndef LMC_SCHED_GROUP
#define LMC_SCHED_GROUP 2
#endif
constexpr int SCHED_GROUP = LMC_SCHED_GROUP;  // TV stages per scheduling region (0 = unrestricted)

template <int K>
struct StreamGeom {
  static constexpr int D = (2 * K + 2 > 10) ? 2 * K + 2 : 10;  // output row lag: o = t - D
  static constexpr int E = D - (2 * K + 2);                    // extra lag of the TV pipeline
  static constexpr int RB = D + 2;                             // x ring rows (each stored twice)
};

template <int K, int KT>
struct StreamState {
  float rr[K + 1][2], ss[K + 1][2], p[K + 1][2], q[K + 1][2], sol[K + 2][2];
  float hxw[KT - 1], hrw[KT - 1];   // vertical windows of the two blur passes (newest first)
  float xpre[2], ypre[2], nz[4];
};

struct StreamCtx {
  float* xring;   // [2*RB][BWP]
  float* rrow;    // [2][BWP]   residual row, by tick parity
  float* gsol;    // [2][K+2][NW] right-neighbour sol for lane 63 of wave w
  float* gss;     // [2][K+2][NW] left-neighbour ss for lane 0 of wave w
  const float* xin;
  float* xout;
  int chain, col, colc, lane, wave;
  float cright;
};

// One tick.  U = t mod 4 (static): parity P = U & 1, noise slot = (U - D) & 3.
// EDGE = false: every row touched by this tick is strictly inside the image -> no predicates.
template <int K, int NW, int KT, int U, bool EDGE>
__device__ __forceinline__ void stream_tick(const StepArgs& A, const int t, const int tm, const StreamCtx& c,
                                            StreamState<K, KT>& S) {
  using G = StreamGeom<K>;
  constexpr int P = U & 1;
  constexpr int BW = 64 * NW, BWP = BW + 2 * kPad;
  const int H = A.H, W = A.W;
  const bool incol = c.col < W;
  float* const xb = c.xring + tm * BWP + kPad + c.col;  // row t-cc is xb[(RB - cc) * BWP]

  // ---- 0. row t arrives: publish it (twice) in the x ring; fetch row t+2 -------------------------
  {
    float xv = S.xpre[P];
    if (EDGE) xv = (t < H) ? xv : 0.f;
    xv = incol ? xv : 0.f;
    xb[0] = xv;
    xb[G::RB * BWP] = xv;
    int tn = t + 2;
    if (EDGE) tn = tn < H ? tn : H - 1;
    S.xpre[P] = LD(c.xin, (size_t)tn * W + c.colc, (size_t)H * W, 1);
  }

  const int o = t - G::D;  // output row of this tick
  float prox_o = 0.f;

  // ---- 1. TV pipeline, stages K+1 .. 1 (decreasing: stage k overwrites what stage k+1 just read) -
  if constexpr (K > 0) {
    const float gam = A.tv.gamma, cstep = A.tv.c;
    const float* gssr = c.gss + (P ^ 1) * (K + 2) * NW + c.wave;
    const float* gsolr = c.gsol + (P ^ 1) * (K + 2) * NW + c.wave;
#pragma unroll
    for (int k = K + 1; k >= 1; --k) {
      // A_k: sol^k[a] = x[a] - gamma * div(rr^{k-1}, ss^{k-1})[a],  a = t - E - 2k
      const float xa = xb[(G::RB - (G::E + 2 * k)) * BWP];
      float sol;
      if (k == 1) {
        sol = xa;
      } else {
        const float ssc = S.ss[k - 1][P ^ 1];
        const float ssl = dpp_from_left(ssc, gssr[(k - 1) * NW]);
        const float dv = (S.rr[k - 1][P ^ 1] - S.rr[k - 1][P]) + (ssc - ssl);
        sol = fmaf(-gam, dv, xa);
      }
      S.sol[k][P] = sol;
      if (k == K + 1) {
        prox_o = sol;
      } else {
        // B_k: dual update on row b = a - 1
        const float solb = S.sol[k][P ^ 1];
        const float solr = dpp_from_right(solb, gsolr[k * NW]);
        float cdown = cstep;
        if (EDGE) {  // wave-uniform: no vertical difference across the last row, nor from row -1
          const int b = t - G::E - 2 * k - 1;
          cdown = ((unsigned)b >= (unsigned)(H - 1)) ? 0.f : cstep;
        }
        const float dx = sol - solb, dy = solr - solb;
        const float r = fmaf(-cdown, dx, S.rr[k - 1][P]);
        const float s = fmaf(-c.cright, dy, S.ss[k - 1][P]);
        const float inv = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(fmaf(r, r, s * s)), 0.f, 1.f);   // == rsq(max(., 1)); folds into v_rsq ... clamp
        const float pn = r * inv, qn = s * inv;
        const float beta = A.tv.betas[k - 1];
        S.rr[k][P] = fmaf(beta, pn - S.p[k - 1][P], pn);
        S.ss[k][P] = fmaf(beta, qn - S.q[k - 1][P], qn);
        S.p[k][P] = pn;
        S.q[k][P] = qn;
      }
      // keep the scheduler from hoisting every stage's LDS reads to the top of the tick (VGPR pressure)
      if (SCHED_GROUP > 0 && (k % SCHED_GROUP) == 1 && k > 1) __builtin_amdgcn_sched_barrier(0);
    }
    // ghost exchange for the next tick: lane 0 publishes sol (right neighbour of the previous wave's
    // lane 63), lane 63 publishes ss (left neighbour of the next wave's lane 0)
    if (NW > 1) {
      if (c.lane == 0 && c.wave > 0) {
        float* g = c.gsol + P * (K + 2) * NW + c.wave - 1;
#pragma unroll
        for (int k = 1; k <= K; ++k) g[k * NW] = S.sol[k][P];
      }
      if (c.lane == 63 && c.wave < NW - 1) {
        float* g = c.gss + P * (K + 2) * NW + c.wave + 1;
#pragma unroll
        for (int k = 1; k <= K; ++k) g[k * NW] = S.ss[k][P];
      }
    }
  }

  // ---- 2. blur gradient pipeline: g[o] = sigma_f * H^T (H x - y) [o] -----------------------------
  float g_o = 0.f;
  if (A.data_kind == LMC_DATA_BLUR) {
    const int oy = A.blur.oy, ox = A.blur.ox;
    const float* __restrict__ uv = A.blur.h;  // separable taps: u[0..KT) then v[0..KT) at h[kMaxBlur..], zero padded
    // (a) horizontal pass of x row ru = o + KT:  hx = sum_b v[b] x[ru][col - b + ox]
    float hxn = 0.f;
    {
      const float* xr = xb + (G::RB - (G::D - KT)) * BWP + ox;
#pragma unroll
      for (int b = 0; b < KT; ++b) hxn = fmaf(uv[kMaxBlur + b], xr[-b], hxn);
    }
    // (b) residual row i = ru - oy:  R = (Hx)[i] - y[i], zero outside the image
    const int i = o + KT - oy;
    {
      float acc = uv[0] * hxn;
#pragma unroll
      for (int a = 1; a < KT; ++a) acc = fmaf(uv[a], S.hxw[a - 1], acc);
#pragma unroll
      for (int a = KT - 2; a >= 1; --a) S.hxw[a] = S.hxw[a - 1];
      S.hxw[0] = hxn;
      float rv = acc - S.ypre[P];
      if (EDGE) rv = ((i >= 0) & (i < H)) ? rv : 0.f;
      rv = incol ? rv : 0.f;
      c.rrow[P * BWP + kPad + c.col] = rv;
      int in2 = i + 2;
      if (EDGE) in2 = in2 < 0 ? 0 : (in2 < H ? in2 : H - 1);
      S.ypre[P] = LD(A.y, (size_t)in2 * W + c.colc, (size_t)H * W, 2);
    }
    // (c) horizontal adjoint of the residual row published in the previous tick (row i - 1)
    float hrn = 0.f;
    {
      const float* rp = c.rrow + (P ^ 1) * BWP + kPad + c.col - ox;
#pragma unroll
      for (int b = 0; b < KT; ++b) hrn = fmaf(uv[kMaxBlur + b], rp[b], hrn);
    }
    // (d) vertical adjoint: g[o] = sum_a u[a] hR[o + a - oy]; the newest row i - 1 is tap a = KT - 1
    {
      float acc = uv[KT - 1] * hrn;
#pragma unroll
      for (int a = 0; a < KT - 1; ++a) acc = fmaf(uv[a], S.hrw[KT - 2 - a], acc);
#pragma unroll
      for (int a = KT - 2; a >= 1; --a) S.hrw[a] = S.hrw[a - 1];
      S.hrw[0] = hrn;
      g_o = A.sigma_f * acc;
    }
  }

  // ---- 3. noise: one Philox call per lane every 4 rows ---------------------------------------------
  constexpr int NI = ((U - G::D) % 4 + 4) % 4;  // == o & 3
  if (A.noise_mode == LMC_NOISE_PHILOX) {
    if (NI == 0 && (!EDGE || (o >= 0 && o < H))) {
      quad_normals(A.key0, A.key1, A.iteration, A.chain_offset + (uint32_t)c.chain,
                   (uint32_t)(o >> 2) * (uint32_t)W + (uint32_t)c.col, S.nz);
    }
  }

  // ---- 4. combine and store row o --------------------------------------------------------------------
  if ((!EDGE || (o >= 0 && o < H)) && incol) {
    const size_t gi = (size_t)o * W + c.col;
    const float x = xb[(G::RB - G::D) * BWP];
    float g = g_o;
    if (A.data_kind == LMC_DATA_IDENTITY) {
      g = A.sigma_f * (x - LD(A.y, gi, (size_t)H * W, 3));
    } else if (A.data_kind == LMC_DATA_MASK) {
      const float mk = LD(A.mask, gi, (size_t)H * W, 4);
      g = A.sigma_f * mk * fmaf(mk, x, -LD(A.y, gi, (size_t)H * W, 5));
    }
    float px;
    if (K > 0) {
      px = prox_o;
    } else if (A.prior_kind == LMC_PRIOR_L2) {
      px = x * A.prior_p0;
    } else if (A.prior_kind == LMC_PRIOR_L1) {
      px = copysignf(fmaxf(fabsf(x) - A.prior_p0, 0.f), x);
    } else {
      px = x;
    }
    float xi = S.nz[NI];
    if (A.noise_mode == LMC_NOISE_INJECTED) xi = LD(A.noise, (size_t)c.chain * H * W + gi, (size_t)A.C * H * W, 6);
    if (A.noise_mode == LMC_NOISE_NONE) xi = 0.f;
    ST(c.xout, gi, (size_t)H * W, fmaf(A.a, x, fmaf(-A.t, g, fmaf(A.b, px, A.s * xi))), 7);
  }
  __syncthreads();
}

template <int K, int NW, int KT, bool EDGE>
__device__ __forceinline__ void stream_group(const StepArgs& A, const int t0, int& tm, const StreamCtx& c,
                                             StreamState<K, KT>& S) {
  constexpr int RB = StreamGeom<K>::RB;
  stream_tick<K, NW, KT, 0, EDGE>(A, t0 + 0, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;
  stream_tick<K, NW, KT, 1, EDGE>(A, t0 + 1, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;
  stream_tick<K, NW, KT, 2, EDGE>(A, t0 + 2, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;
  stream_tick<K, NW, KT, 3, EDGE>(A, t0 + 3, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;
}

// 2 waves per SIMD at NW = 8 (one 512-thread workgroup per CU): at most 256 VGPRs; the same cap is
// requested for the narrower variants so that every instantiation is register-allocated alike.
template <int K, int NW, int KT>
__global__ __launch_bounds__(64 * NW, 2) void myula_step_stream_kernel(const StepArgs A) {
  extern __shared__ float lds[];
  using G = StreamGeom<K>;
  constexpr int BW = 64 * NW, BWP = BW + 2 * kPad;
  const int tid = threadIdx.x;
  const int H = A.H, W = A.W;
  const size_t img = (size_t)H * W;

  StreamCtx c;
  c.chain = blockIdx.x;
  c.col = tid;
  c.colc = tid < W ? tid : W - 1;   // clamped column for unconditional prefetch loads
  c.lane = tid & 63;
  c.wave = tid >> 6;
  c.xin = A.x_in + (size_t)c.chain * img;
  c.xout = A.x_out + (size_t)c.chain * img;
  c.xring = lds;
  c.rrow = c.xring + 2 * G::RB * BWP;
  c.gsol = c.rrow + 2 * BWP;
  c.gss = c.gsol + 2 * (K + 2) * NW;
  c.cright = (tid >= W - 1) ? 0.f : A.tv.c;  // no horizontal difference across the last column
  constexpr int lds_floats = 2 * G::RB * BWP + 2 * BWP + 4 * (K + 2) * NW;
  for (int i = tid; i < lds_floats; i += BW) lds[i] = 0.f;

  StreamState<K, KT> S;
#pragma unroll
  for (int k = 0; k <= K; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j) S.rr[k][j] = S.ss[k][j] = S.p[k][j] = S.q[k][j] = 0.f;
#pragma unroll
  for (int k = 0; k <= K + 1; ++k) S.sol[k][0] = S.sol[k][1] = 0.f;
#pragma unroll
  for (int a = 0; a < KT - 1; ++a) S.hxw[a] = S.hrw[a] = 0.f;
  S.nz[0] = S.nz[1] = S.nz[2] = S.nz[3] = 0.f;
  S.xpre[0] = LD(c.xin, c.colc, (size_t)H * W, 8);
  S.xpre[1] = LD(c.xin, (size_t)(H > 1 ? 1 : 0) * W + c.colc, (size_t)H * W, 9);
  S.ypre[0] = S.ypre[1] = 0.f;
  if (A.data_kind == LMC_DATA_BLUR) {
    // residual row of tick t is i(t) = t - D + KT - oy; ypre[t & 1] must hold y[i(t)] when i(t) is a row
    const int i0 = -G::D + KT - A.blur.oy;
    const int r0 = i0 < 0 ? 0 : (i0 < H ? i0 : H - 1), r1 = i0 + 1 < 0 ? 0 : (i0 + 1 < H ? i0 + 1 : H - 1);
    S.ypre[0] = LD(A.y, (size_t)r0 * W + c.colc, (size_t)H * W, 10);
    S.ypre[1] = LD(A.y, (size_t)r1 * W + c.colc, (size_t)H * W, 11);
  }
  __syncthreads();

  // ticks 0 .. T-1 produce output rows -D .. H-1; groups of 4 keep U = t & 3 static
  const int T = H + G::D;
  constexpr int t_lo = (G::D + 3) & ~3;  // first group whose 4 ticks all have o >= 0
  int tm = 0;
  for (int t0 = 0; t0 < T; t0 += 4) {
    // predicate-free body needs: o >= 0, row t+2 < H (prefetch), residual rows and y prefetch inside
    const bool steady = false;   // single general path: the two-path loop inflates the live state (168 -> 282 VGPRs)
    (void)t_lo;
    if (steady) stream_group<K, NW, KT, false>(A, t0, tm, c, S);
    else stream_group<K, NW, KT, true>(A, t0, tm, c, S);
  }
}

// ---- host side ----------------------------------------------------------------------------------

template <int K>
static size_t stream_lds_bytes(int NW) {
  using G = StreamGeom<K>;
  const int BW = 64 * NW, BWP = BW + 2 * kPad;
  return sizeof(float) * (size_t)(2 * G::RB * BWP + 2 * BWP + 4 * (K + 2) * NW);
}

template <int K>
static bool stream_fits(int NW) {
  using G = StreamGeom<K>;
  const int BWP = 64 * NW + 2 * kPad;
  // ds_read immediate offsets are 16 bit: the farthest read is RB rows from the tick base
  return (size_t)G::RB * BWP * sizeof(float) + 64 <= 65535 && stream_lds_bytes<K>(NW) <= 160 * 1024;
}

static int stream_nw(int W) { return W <= 64 ? 1 : (W <= 128 ? 2 : (W <= 256 ? 4 : (W <= 512 ? 8 : 0))); }

template <int K, int NW, int KT>
static hipError_t launch_knw(const StepArgs& a, hipStream_t st) {
  auto k = myula_step_stream_kernel<K, NW, KT>;
  const size_t lds = stream_lds_bytes<K>(NW);
  static thread_local bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    configured = true;
  }
  hipLaunchKernelGGL(k, dim3(a.C), dim3(64 * NW), lds, st, a);
  return hipGetLastError();
}

template <int K, int KT>
static hipError_t launch_k(const StepArgs& a, hipStream_t st) {
  switch (stream_nw(a.W)) {
    case 1: return launch_knw<K, 1, KT>(a, st);
    case 2: return launch_knw<K, 2, KT>(a, st);
    case 4: return launch_knw<K, 4, KT>(a, st);
    case 8: return launch_knw<K, 8, KT>(a, st);
  }
  return hipErrorInvalidConfiguration;
}

template <int K>
static hipError_t launch_kt(const StepArgs& a, int KT, hipStream_t st) {
  if (KT <= 5) return launch_k<K, 5>(a, st);
  return launch_k<K, 7>(a, st);
}

// Rank-1 factorisation h = u v^T of the blur taps (u: kh, v: kw).  Returns false if h is not separable.
bool separate_blur_taps(const BlurTaps& T, float* u, float* v) {
  int pa = 0, pb = 0;
  float best = 0.f;
  for (int a = 0; a < T.kh; ++a)
    for (int b = 0; b < T.kw; ++b)
      if (fabsf(T.h[a * T.kw + b]) > best) { best = fabsf(T.h[a * T.kw + b]); pa = a; pb = b; }
  if (best == 0.f) return false;
  const float piv = T.h[pa * T.kw + pb];
  for (int a = 0; a < T.kh; ++a) u[a] = T.h[a * T.kw + pb] / piv;
  for (int b = 0; b < T.kw; ++b) v[b] = T.h[pa * T.kw + b];
  for (int a = 0; a < T.kh; ++a)
    for (int b = 0; b < T.kw; ++b)
      if (fabsf(u[a] * v[b] - T.h[a * T.kw + b]) > 1e-6f * best) return false;
  return true;
}

static int stream_k(const StepArgs& a) { return a.prior_kind == LMC_PRIOR_TV_ISO ? a.tv.niter : 0; }

// True if the streaming kernel covers this configuration (else the caller uses the tile kernel).
bool stream_supported(const StepArgs& a) {
  const int NW = stream_nw(a.W);
  if (NW == 0 || a.H < 1 || a.ncvx_kind != LMC_NCVX_NONE || a.extra || a.prox_ext) return false;
  bool fits = false;
  switch (stream_k(a)) {
#ifndef LMC_ONLY_K10
    case 0: fits = stream_fits<0>(NW); break;
    case 1: fits = stream_fits<1>(NW); break;
    case 2: fits = stream_fits<2>(NW); break;
    case 3: fits = stream_fits<3>(NW); break;
    case 4: fits = stream_fits<4>(NW); break;
    case 5: fits = stream_fits<5>(NW); break;
    case 6: fits = stream_fits<6>(NW); break;
    case 8: fits = stream_fits<8>(NW); break;
    case 12: fits = stream_fits<12>(NW); break;
#endif
    case 10: fits = stream_fits<10>(NW); break;
    default: return false;
  }
  if (!fits) return false;
  if (a.data_kind == LMC_DATA_BLUR) {
    if (a.blur.kh > 7 || a.blur.kw > 7) return false;
    float u[kMaxBlur], v[kMaxBlur];
    if (!separate_blur_taps(a.blur, u, v)) return false;
  }
  return true;
}

hipError_t launch_step_stream(StepArgs a, hipStream_t st) {
  int KT = 5;
  if (a.data_kind == LMC_DATA_BLUR) {
    float u[kMaxBlur] = {0}, v[kMaxBlur] = {0};
    if (!separate_blur_taps(a.blur, u, v)) return hipErrorInvalidConfiguration;
    for (int i = 0; i < kMaxBlur; ++i) { a.blur.h[i] = u[i]; a.blur.h[kMaxBlur + i] = v[i]; }  // zero padded
    KT = (a.blur.kh > a.blur.kw ? a.blur.kh : a.blur.kw) <= 5 ? 5 : 7;
  }
  switch (stream_k(a)) {
#ifndef LMC_ONLY_K10
    case 0: return launch_kt<0>(a, KT, st);
    case 1: return launch_kt<1>(a, KT, st);
    case 2: return launch_kt<2>(a, KT, st);
    case 3: return launch_kt<3>(a, KT, st);
    case 4: return launch_kt<4>(a, KT, st);
    case 5: return launch_kt<5>(a, KT, st);
    case 6: return launch_kt<6>(a, KT, st);
    case 8: return launch_kt<8>(a, KT, st);
    case 12: return launch_kt<12>(a, KT, st);
#endif
    case 10: return launch_kt<10>(a, KT, st);
  }
  return hipErrorInvalidConfiguration;
}

}  // namespace lmc

#ifdef LMC_BOUNDS_CHECK
extern "C" int lmc_debug_read(long long* out8) {
  return (int)hipMemcpyFromSymbol(out8, HIP_SYMBOL(lmc::lmc_dbg), sizeof(long long) * 8);
}
#endif
