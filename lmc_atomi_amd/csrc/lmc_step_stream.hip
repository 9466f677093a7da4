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
// The blur gradient sigma_f * H^T(Hx - y) (separable taps) runs as four 1-D passes on LDS row
// rings beside the TV pipeline; x itself is kept in an LDS ring of >= D+2 rows (it is needed once per
// stage), filled from HBM two ticks ahead.  Noise: Philox4x32-10 + Box-Muller, one call per lane
// every 4 rows.  HBM traffic: x read once, x' written once (8 B per pixel per iteration).
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

constexpr int kPad = 8;  // zero columns on both sides of LDS rows (>= kMaxBlur - 1)

__device__ __forceinline__ float dpp_from_left(float v, float edge) {   // lane i <- v[i-1]; lane 0 <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                              __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ float dpp_from_right(float v, float edge) {  // lane i <- v[i+1]; lane 63 <- edge
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, edge),
                                                              __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
}

template <int K>
struct StreamGeom {
  static constexpr int D = (2 * K + 2 > 10) ? 2 * K + 2 : 10;  // output row lag: o = t - D
  static constexpr int E = D - (2 * K + 2);                    // extra lag of the TV pipeline
  static constexpr int RB = D + 2 <= 16 ? 16 : (D + 2 <= 32 ? 32 : 64);  // x ring rows (power of two)
};

struct StreamLds {
  float* xring;   // [RB][BWP]
  float* hx;      // [16][BW]   horizontal pass of x
  float* hr;      // [16][BW]   horizontal adjoint pass of R
  float* rrow;    // [2][BWP]   residual row, by tick parity
  float* gsol;    // [2][K+2][NW] right-neighbour sol for lane 63 of wave w
  float* gss;     // [2][K+2][NW] left-neighbour ss for lane 0 of wave w
};

template <int K>
struct TvState {
  float rr[K + 1][2], ss[K + 1][2], p[K + 1][2], q[K + 1][2], sol[K + 2][2];
};

// One tick.  U = t mod 4 (static): parity P = U & 1, noise slot = (U - D) & 3.
template <int K, int NW, int U>
__device__ __forceinline__ void stream_tick(const StepArgs& P_, const int t, const int chain, const int col,
                                            const int lane, const int wave, const StreamLds& L,
                                            TvState<K>& S, float (&xpre)[2], float (&ypre)[2], float (&nz)[4],
                                            const float cright, const float* __restrict__ xin,
                                            float* __restrict__ xout) {
  using G = StreamGeom<K>;
  constexpr int P = U & 1;
  constexpr int BW = 64 * NW, BWP = BW + 2 * kPad;
  const int H = P_.H, W = P_.W;
  const bool incol = col < W;

  // ---- 0. row t arrives: publish it in the x ring; fetch row t+2 -------------------------------
  {
    const float xv = (t < H && incol) ? xpre[P] : 0.f;
    L.xring[(t & (G::RB - 1)) * BWP + kPad + col] = xv;
    const int tn = t + 2;
    if (tn < H && incol) xpre[P] = xin[(size_t)tn * W + col];
  }

  const int o = t - G::D;  // output row of this tick
  float prox_o = 0.f;

  // ---- 1. TV pipeline, stages K+1 .. 1 (decreasing: stage k overwrites what stage k+1 just read) -
  if constexpr (K > 0) {
    const float gam = P_.tv.gamma, cstep = P_.tv.c;
#pragma unroll
    for (int k = K + 1; k >= 1; --k) {
      const int a = t - G::E - 2 * k;  // A-row of stage k
      // A_k: sol^k[a] = x[a] - gamma * div(rr^{k-1}, ss^{k-1})[a]
      const float xa = L.xring[(a & (G::RB - 1)) * BWP + kPad + col];
      float sol;
      if (k == 1) {
        sol = xa;
      } else {
        const float ssc = S.ss[k - 1][P ^ 1];
        const float ssl = dpp_from_left(ssc, L.gss[(P ^ 1) * (K + 2) * NW + (k - 1) * NW + wave]);
        const float dv = (S.rr[k - 1][P ^ 1] - S.rr[k - 1][P]) + (ssc - ssl);
        sol = fmaf(-gam, dv, xa);
      }
      S.sol[k][P] = sol;
      if (k == K + 1) {
        prox_o = sol;
      } else {
        // B_k: dual update on row b = a - 1
        const int b = a - 1;
        const float solb = S.sol[k][P ^ 1];
        const float solr = dpp_from_right(solb, L.gsol[(P ^ 1) * (K + 2) * NW + k * NW + wave]);
        // wave-uniform: no vertical difference across the last row, nor from row -1 (pipeline fill)
        const float cdown = ((unsigned)b >= (unsigned)(H - 1)) ? 0.f : cstep;
        const float dx = sol - solb, dy = solr - solb;
        const float r = fmaf(-cdown, dx, S.rr[k - 1][P]);
        const float s = fmaf(-cright, dy, S.ss[k - 1][P]);
        const float inv = rsqrtf(fmaxf(fmaf(r, r, s * s), 1.f));
        const float pn = r * inv, qn = s * inv;
        const float beta = P_.tv.betas[k - 1];
        S.rr[k][P] = fmaf(beta, pn - S.p[k - 1][P], pn);
        S.ss[k][P] = fmaf(beta, qn - S.q[k - 1][P], qn);
        S.p[k][P] = pn;
        S.q[k][P] = qn;
      }
    }
    // ghost exchange for the next tick: lane 0 publishes sol (right neighbour of the previous wave's
    // lane 63), lane 63 publishes ss (left neighbour of the next wave's lane 0)
    if (NW > 1) {
      if (lane == 0 && wave > 0) {
#pragma unroll
        for (int k = 1; k <= K; ++k) L.gsol[P * (K + 2) * NW + k * NW + wave - 1] = S.sol[k][P];
      }
      if (lane == 63 && wave < NW - 1) {
#pragma unroll
        for (int k = 1; k <= K; ++k) L.gss[P * (K + 2) * NW + k * NW + wave + 1] = S.ss[k][P];
      }
    }
  }

  // ---- 2. blur gradient pipeline: g[o] = sigma_f * H^T (H x - y) [o] -----------------------------
  float g_o = 0.f;
  if (P_.data_kind == LMC_DATA_BLUR) {
    const int kh = P_.blur.kh, kw = P_.blur.kw, oy = P_.blur.oy, ox = P_.blur.ox;
    const float* __restrict__ uv = P_.blur.h;            // separable taps: u[0..kh) then v[0..kw) at h[kMaxBlur..]
    const int ru = o + kh;                               // newest x row entering the horizontal pass
    // (a) hx[ru] = sum_b v[b] x[ru][col - b + ox]
    float hxn = 0.f;
    {
      const float* xr = L.xring + (ru & (G::RB - 1)) * BWP + kPad + col + ox;
      for (int b = 0; b < kw; ++b) hxn = fmaf(uv[kMaxBlur + b], xr[-b], hxn);
      L.hx[(ru & 15) * BW + col] = hxn;
    }
    // (b) residual row i = ru - oy:  R = (Hx)[i] - y[i], zero outside the image
    const int i = ru - oy;
    {
      float acc = uv[0] * hxn;
      for (int a = 1; a < kh; ++a) acc = fmaf(uv[a], L.hx[((ru - a) & 15) * BW + col], acc);
      const bool ok = (i >= 0) & (i < H) & incol;
      L.rrow[P * BWP + kPad + col] = ok ? acc - ypre[P] : 0.f;
      const int in2 = i + 2;
      if (in2 >= 0 && in2 < H && incol) ypre[P] = P_.y[(size_t)in2 * W + col];
    }
    // (c) horizontal adjoint of the residual row published in the previous tick (row i - 1)
    float hrn = 0.f;
    {
      const float* rp = L.rrow + (P ^ 1) * BWP + kPad + col - ox;
      for (int b = 0; b < kw; ++b) hrn = fmaf(uv[kMaxBlur + b], rp[b], hrn);
      L.hr[((i - 1) & 15) * BW + col] = hrn;
    }
    // (d) vertical adjoint: g[o] = sum_a u[a] hR[o + a - oy]; the newest row i-1 is tap a = kh-1
    {
      float acc = uv[kh - 1] * hrn;
      for (int a = 0; a < kh - 1; ++a) acc = fmaf(uv[a], L.hr[((o + a - oy) & 15) * BW + col], acc);
      g_o = P_.sigma_f * acc;
    }
  }

  // ---- 3. noise: one Philox call per lane every 4 rows ---------------------------------------------
  constexpr int NI = ((U - G::D) % 4 + 4) % 4;  // == o & 3 for o >= 0
  if (P_.noise_mode == LMC_NOISE_PHILOX) {
    if (NI == 0 && o >= 0 && o < H) {
      quad_normals(P_.key0, P_.key1, P_.iteration, P_.chain_offset + (uint32_t)chain,
                   (uint32_t)(o >> 2) * (uint32_t)W + (uint32_t)col, nz);
    }
  }

  // ---- 4. combine and store row o --------------------------------------------------------------------
  if (o >= 0 && o < H && incol) {
    const size_t gi = (size_t)o * W + col;
    const float x = L.xring[(o & (G::RB - 1)) * BWP + kPad + col];
    float g = g_o;
    if (P_.data_kind == LMC_DATA_IDENTITY) {
      g = P_.sigma_f * (x - P_.y[gi]);
    } else if (P_.data_kind == LMC_DATA_MASK) {
      const float mk = P_.mask[gi];
      g = P_.sigma_f * mk * fmaf(mk, x, -P_.y[gi]);
    }
    float px;
    if (K > 0) {
      px = prox_o;
    } else if (P_.prior_kind == LMC_PRIOR_L2) {
      px = x * P_.prior_p0;
    } else if (P_.prior_kind == LMC_PRIOR_L1) {
      px = copysignf(fmaxf(fabsf(x) - P_.prior_p0, 0.f), x);
    } else {
      px = x;
    }
    float xi = nz[NI];
    if (P_.noise_mode == LMC_NOISE_INJECTED) xi = P_.noise[(size_t)chain * H * W + gi];
    if (P_.noise_mode == LMC_NOISE_NONE) xi = 0.f;
    xout[gi] = fmaf(P_.a, x, fmaf(-P_.t, g, fmaf(P_.b, px, P_.s * xi)));
  }
  __syncthreads();
}

template <int K, int NW>
__global__ __launch_bounds__(64 * NW) void myula_step_stream_kernel(const StepArgs P_) {
  extern __shared__ float lds[];
  using G = StreamGeom<K>;
  constexpr int BW = 64 * NW, BWP = BW + 2 * kPad;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int chain = blockIdx.x;
  const int H = P_.H, W = P_.W;
  const size_t img = (size_t)H * W;
  const float* __restrict__ xin = P_.x_in + (size_t)chain * img;
  float* __restrict__ xout = P_.x_out + (size_t)chain * img;

  StreamLds L;
  L.xring = lds;
  L.hx = L.xring + G::RB * BWP;
  L.hr = L.hx + 16 * BW;
  L.rrow = L.hr + 16 * BW;
  L.gsol = L.rrow + 2 * BWP;
  L.gss = L.gsol + 2 * (K + 2) * NW;
  const int lds_floats = G::RB * BWP + 32 * BW + 2 * BWP + 4 * (K + 2) * NW;
  for (int i = tid; i < lds_floats; i += BW) lds[i] = 0.f;

  TvState<K> S;
#pragma unroll
  for (int k = 0; k <= K; ++k)
#pragma unroll
    for (int j = 0; j < 2; ++j) S.rr[k][j] = S.ss[k][j] = S.p[k][j] = S.q[k][j] = 0.f;
#pragma unroll
  for (int k = 0; k <= K + 1; ++k) S.sol[k][0] = S.sol[k][1] = 0.f;

  const int col = tid;
  const bool incol = col < W;
  const float cright = (col >= W - 1) ? 0.f : P_.tv.c;  // no gradient across the last column
  float xpre[2] = {0.f, 0.f}, ypre[2] = {0.f, 0.f}, nz[4] = {0.f, 0.f, 0.f, 0.f};
  if (incol) {
    xpre[0] = xin[col];
    if (H > 1) xpre[1] = xin[(size_t)W + col];
  }
  if (P_.data_kind == LMC_DATA_BLUR && incol) {
    // residual row of tick t is i(t) = t - D + kh - oy; ypre[t & 1] must hold y[i(t)]
    const int i0 = -G::D + P_.blur.kh - P_.blur.oy;
    if (i0 >= 0 && i0 < H) ypre[0] = P_.y[(size_t)i0 * W + col];
    if (i0 + 1 >= 0 && i0 + 1 < H) ypre[1] = P_.y[(size_t)(i0 + 1) * W + col];
  }
  __syncthreads();

  const int T = H + G::D;  // ticks 0 .. T-1 produce output rows -D .. H-1
  for (int t0 = 0; t0 < T; t0 += 4) {
    stream_tick<K, NW, 0>(P_, t0 + 0, chain, col, lane, wave, L, S, xpre, ypre, nz, cright, xin, xout);
    stream_tick<K, NW, 1>(P_, t0 + 1, chain, col, lane, wave, L, S, xpre, ypre, nz, cright, xin, xout);
    stream_tick<K, NW, 2>(P_, t0 + 2, chain, col, lane, wave, L, S, xpre, ypre, nz, cright, xin, xout);
    stream_tick<K, NW, 3>(P_, t0 + 3, chain, col, lane, wave, L, S, xpre, ypre, nz, cright, xin, xout);
  }
}

// ---- host side ----------------------------------------------------------------------------------

template <int K>
static size_t stream_lds_bytes(int NW) {
  using G = StreamGeom<K>;
  const int BW = 64 * NW, BWP = BW + 2 * kPad;
  return sizeof(float) * (size_t)(G::RB * BWP + 32 * BW + 2 * BWP + 4 * (K + 2) * NW);
}

template <int K, int NW>
static hipError_t launch_knw(const StepArgs& a, hipStream_t st) {
  auto k = myula_step_stream_kernel<K, NW>;
  const size_t lds = stream_lds_bytes<K>(NW);
  if (lds > 160 * 1024) return hipErrorInvalidConfiguration;
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

template <int K>
static hipError_t launch_k(const StepArgs& a, hipStream_t st) {
  if (a.W <= 64) return launch_knw<K, 1>(a, st);
  if (a.W <= 128) return launch_knw<K, 2>(a, st);
  if (a.W <= 256) return launch_knw<K, 4>(a, st);
  if (a.W <= 512) return launch_knw<K, 8>(a, st);
  return hipErrorInvalidConfiguration;
}

// Rank-1 factorisation h = u v^T of the blur taps (u: kh, v: kw).  Returns false if h is not separable.
static bool separate_taps(const BlurTaps& T, float* u, float* v) {
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

// True if the streaming kernel covers this configuration (else the caller uses the tile kernel).
bool stream_supported(const StepArgs& a) {
  if (a.W > 512 || a.H < 1) return false;
  const int K = a.prior_kind == LMC_PRIOR_TV_ISO ? a.tv.niter : 0;
  switch (K) {
    case 0: case 1: case 2: case 3: case 4: case 5: case 6: case 8: case 10: case 12: case 16: case 20: break;
    default: return false;
  }
  if (a.data_kind == LMC_DATA_BLUR) {
    float u[kMaxBlur], v[kMaxBlur];
    if (!separate_taps(a.blur, u, v)) return false;
  }
  return true;
}

hipError_t launch_step_stream(StepArgs a, hipStream_t st) {
  if (a.data_kind == LMC_DATA_BLUR) {
    float u[kMaxBlur] = {0}, v[kMaxBlur] = {0};
    if (!separate_taps(a.blur, u, v)) return hipErrorInvalidConfiguration;
    for (int i = 0; i < kMaxBlur; ++i) { a.blur.h[i] = u[i]; a.blur.h[kMaxBlur + i] = v[i]; }
  }
  const int K = a.prior_kind == LMC_PRIOR_TV_ISO ? a.tv.niter : 0;
  switch (K) {
    case 0: return launch_k<0>(a, st);
    case 1: return launch_k<1>(a, st);
    case 2: return launch_k<2>(a, st);
    case 3: return launch_k<3>(a, st);
    case 4: return launch_k<4>(a, st);
    case 5: return launch_k<5>(a, st);
    case 6: return launch_k<6>(a, st);
    case 8: return launch_k<8>(a, st);
    case 10: return launch_k<10>(a, st);
    case 12: return launch_k<12>(a, st);
    case 16: return launch_k<16>(a, st);
    case 20: return launch_k<20>(a, st);
  }
  return hipErrorInvalidConfiguration;
}

}  // namespace lmc
