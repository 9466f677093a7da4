// Fused MYULA update, split streaming variant ("split"): the row pipeline of lmc_step_stream.hip with the
// TV stages divided between TWO wave groups of the same workgroup, so that each thread carries half of the
// per-column state (<= 128 VGPRs) and 4 wavefronts per SIMD are resident instead of 2.
//
//   group A (threads 0 .. BW-1)     : HBM load of row t, x ring, blur-gradient pipeline (one row early),
//                                     TV stages 1 .. KA            (KA = ceil(K/2))
//   group B (threads BW .. 2BW-1)   : TV stages KA+1 .. K, final primal step (prox), Philox noise, combine, store
//
// Hand-off A -> B (LDS, parity double buffered, read one tick after it is written -- the same latency a
// stage boundary has inside one group): the four outputs (rr, ss, p, q) of stage KA for its row, and the
// gradient value g of the output row.  Everything else (x ring, ghost exchange, one barrier per tick,
// predicate-free steady ticks, 8 B of HBM traffic per pixel-iteration) is as in the one-group kernel.
// Reference update: algs.py:569.
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

#ifdef LMC_EXP_NOLDS   // timing experiment: no LDS traffic at all (results are wrong)
__device__ __forceinline__ float exp_keep(float v) { asm volatile("" : "+v"(v)); return v; }
#define LR(ptr_expr, alt) exp_keep(alt)
#define LW(lhs, val) do { float v_ = (val); asm volatile("" ::"v"(v_)); } while (0)
#else
#define LR(ptr_expr, alt) (ptr_expr)
#define LW(lhs, val) (lhs) = (val)
#endif

template <int K>
struct SplitGeom {
  static constexpr int D = (2 * K + 2 > 10) ? 2 * K + 2 : 10;  // output row lag: o = t - D
  static constexpr int E = D - (2 * K + 2);                    // extra lag of the TV pipeline
  static constexpr int RB = D + 2;                             // x ring rows (each stored twice)
  static constexpr int KA = (K + 1) / 2;                       // stages owned by group A
};

template <int K, int NW>
struct SplitLds {
  static constexpr int BW = 64 * NW, BWP = BW + 2 * kPad;
  static constexpr int RB = SplitGeom<K>::RB;
  static constexpr int o_xring = 0;                       // [2*RB][BWP]
  static constexpr int o_rrow = o_xring + 2 * RB * BWP;   // [2][BWP]  residual row
  static constexpr int o_hand = o_rrow + 2 * BWP;         // [2][4][BWP] stage-KA outputs rr, ss, p, q
  static constexpr int o_garr = o_hand + 8 * BWP;         // [2][BWP]  gradient of the output row
  // row-edge exchange: per parity a linear array of 64*(NW+1) slots, 16 per DPP row.
  //   sol: first lane of (wave w, row r), stage k -> slot 64w + 16r + k      ; read by the row to its left
  //   ss : last lane of (wave w, row r), stage k  -> slot 64(w+1) + 16r + 15-k; read by the row to its right
  // Each wave group has its own arrays (both store all 64 lanes of a block every tick).
  static constexpr int GB = 64 * (NW + 1);
  static constexpr int o_gsol = o_garr + 2 * BWP;         // [group][2][NW+1][64]
  static constexpr int o_gss = o_gsol + 4 * GB;           // [group][2][NW+1][64]
  static constexpr int total = o_gss + 4 * GB;
};

// Broadcast-load ghost slots [4*V0 .. 4*V1+3] of a 64-slot block with 16-byte LDS reads (one per 4 stages).
template <int V0, int V1>
__device__ __forceinline__ void load_ghost(const float* blk, float (&g)[16]) {
  static_for<V0, V1 + 1>([&](auto vv) {
    constexpr int v = decltype(vv)::value;
#ifdef LMC_EXP_NOLDS
    const float4 q = {1.f, 2.f, 3.f, 4.f};
#else
    const float4 q = *reinterpret_cast<const float4*>(blk + 4 * v);
#endif
    g[4 * v + 0] = q.x; g[4 * v + 1] = q.y; g[4 * v + 2] = q.z; g[4 * v + 3] = q.w;
  });
}

// Horizontal neighbours: DPP row shifts + row-edge ghosts (lmc_device.h).  ds_bpermute was measured slower here
// (2.94 vs 2.38 ms per launch: it competes with the ghost / ring traffic for the LDS pipe).
#define NB_LEFT(c, v, edge) row_from_left(v, edge)
#define NB_RIGHT(c, v, edge) row_from_right(v, edge)

struct SplitCtx {
  float* lds;
  const float* xin;
  float* xout;
  int chain, col, colc, lane, wave;
  float cright;
};

template <int K, int KT>
struct StateA {
  static constexpr int KA = SplitGeom<K>::KA;
  float rr[KA + 1][2], ss[KA + 1][2], p[KA + 1][2], q[KA + 1][2], sol[KA + 2][2];
  float hxw[KT > 1 ? KT - 1 : 1], hrw[KT > 1 ? KT - 1 : 1], xpre[4], ypre[4];   // x / y rows in flight: fetched 4 ticks ahead, slot t & 3
};

template <int K>
struct StateB {
  float hrr[2], hss[2], hp[2], hq[2];   // stage KA's outputs, two ticks deep (filled from the LDS hand-off)
  float rr[K + 1][2], ss[K + 1][2], p[K + 1][2], q[K + 1][2], sol[K + 2][2];
  float nz[4];
  float ypre[4], mpre[4];   // y / mask rows of the pointwise data terms, fetched 4 ticks ahead (slot t & 3)
};

// ---- group A tick --------------------------------------------------------------------------------------
template <int K, int NW, int KT, int U, bool EDGE>
__device__ __forceinline__ void split_tick_a(const StepArgs& A, const int t, const int tm, const SplitCtx& c,
                                             StateA<K, KT>& S) {
  using G = SplitGeom<K>;
  using L = SplitLds<K, NW>;
  constexpr int P = U & 1, BWP = L::BWP, KA = G::KA;
  const int H = A.H, W = A.W;
  const bool incol = c.col < W;
  float* const xb = c.lds + L::o_xring + tm * BWP + kPad + c.col;  // row t-cc is xb[(RB - cc) * BWP]

  {  // row t arrives: publish it (twice) in the x ring; fetch row t+2
    float xv = S.xpre[U];
    if (EDGE) xv = (t < H) ? xv : 0.f;
    xv = incol ? xv : 0.f;
    LW(xb[0], xv);
    LW(xb[G::RB * BWP], xv);
    int tn = t + 4;
    if (EDGE) tn = tn < H ? tn : H - 1;
    S.xpre[U] = LD(c.xin, (size_t)tn * W + c.colc, (size_t)H * W, 1);
  }

  if constexpr (KA > 0) {
    const float gam = A.tv.gamma, cstep = A.tv.c;
    // ghost values of the previous tick: sol of the right neighbour wave's lane 0 (slot k of block wave+1),
    // ss of the left neighbour wave's lane 63 (slot 63-k of block wave, i.e. index 15-k of its last 16 slots)
    float gsolv[16], gssv[16];
    load_ghost<0, KA / 4>(c.lds + L::o_gsol + (P ^ 1) * L::GB + c.wave * 64 + (c.lane & 48) + 16, gsolv);
    if constexpr (KA > 1) load_ghost<(15 - (KA - 1)) / 4, 3>(c.lds + L::o_gss + (P ^ 1) * L::GB + c.wave * 64 + (c.lane & 48) + 48, gssv);
#pragma unroll
    for (int k = KA; k >= 1; --k) {
      const float xa = LR(xb[(G::RB - (G::E + 2 * k)) * BWP], S.xpre[0] + (float)k);
      float sol;
      if (k == 1) {
        sol = xa;
      } else {
        const float ssc = S.ss[k - 1][P ^ 1];
#ifdef LMC_EXP_NOGHOST
        const float ssl = NB_LEFT(c, ssc, 0.f);
#else
        const float ssl = NB_LEFT(c, ssc, gssv[15 - (k - 1)]);
#endif
        sol = fmaf(-gam, (S.rr[k - 1][P ^ 1] - S.rr[k - 1][P]) + (ssc - ssl), xa);
      }
      S.sol[k][P] = sol;
      const float solb = S.sol[k][P ^ 1];
#ifdef LMC_EXP_NOGHOST
      const float solr = NB_RIGHT(c, solb, 0.f);
#else
      const float solr = NB_RIGHT(c, solb, gsolv[k]);
#endif
      float cdown = cstep;
      if (EDGE) {
        const int b = t - G::E - 2 * k - 1;
        cdown = ((unsigned)b >= (unsigned)(H - 1)) ? 0.f : cstep;
      }
      const float r = fmaf(-cdown, sol - solb, S.rr[k - 1][P]);
      const float s = fmaf(-c.cright, solr - solb, S.ss[k - 1][P]);
      const float inv = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(fmaf(r, r, s * s)), 0.f, 1.f);   // == rsq(max(., 1)); folds into v_rsq ... clamp
      const float pn = r * inv, qn = s * inv;
      const float beta = A.tv.betas[k - 1];
      const float rn = fmaf(beta, pn - S.p[k - 1][P], pn);
      const float sn = fmaf(beta, qn - S.q[k - 1][P], qn);
      if (k == KA) {  // hand the stage over to group B
        float* hb = c.lds + L::o_hand + P * 4 * BWP + kPad + c.col;
        LW(hb[0], rn); LW(hb[BWP], sn); LW(hb[2 * BWP], pn); LW(hb[3 * BWP], qn);
      } else {
        S.rr[k][P] = rn; S.ss[k][P] = sn; S.p[k][P] = pn; S.q[k][P] = qn;
      }
#ifdef LMC_SPLIT_SCHED
      if ((k % LMC_SPLIT_SCHED) == 0) __builtin_amdgcn_sched_barrier(0);
#endif
    }
#ifndef LMC_EXP_NOGHOST
    {   // one store per direction: the row-edge values of every stage gathered into lanes by DPP, all lanes store
      float gs = 0.f, gq = 0.f;
      static_for<1, KA + 1>([&](auto kk) { constexpr int k = decltype(kk)::value; gs = gather_first<k>(gs, S.sol[k][P]); });
      LW(c.lds[L::o_gsol + P * L::GB + c.wave * 64 + c.lane], gs);
      if constexpr (KA > 1) {
        static_for<1, KA>([&](auto kk) { constexpr int k = decltype(kk)::value; gq = gather_last<k>(gq, S.ss[k][P]); });
        LW(c.lds[L::o_gss + P * L::GB + (c.wave + 1) * 64 + c.lane], gq);
      }
    }
#endif
  }

  // blur gradient pipeline, one row ahead of the output: g[o+1] -> garr[P] (group B reads it next tick)
  if constexpr (KT > 0) {   // KT == 0: pointwise data term, handled by group B
    const int oy = A.blur.oy, ox = A.blur.ox;
    const float* __restrict__ uv = A.blur.h;  // u[0..KT) then v[0..KT) at h[kMaxBlur..], zero padded
    const int o1 = t - G::D + 1;
    float hxn = 0.f;
    {
      const float* xr = xb + (G::RB - (G::D - 1 - KT)) * BWP + ox;   // x row o1 + KT
#pragma unroll
      for (int b = 0; b < KT; ++b) hxn = fmaf(uv[kMaxBlur + b], LR(xr[-b], S.xpre[1] + (float)b), hxn);
    }
    const int i = o1 + KT - oy;   // residual row
    {
      float acc = uv[0] * hxn;
#pragma unroll
      for (int a = 1; a < KT; ++a) acc = fmaf(uv[a], S.hxw[a - 1], acc);
#pragma unroll
      for (int a = KT - 2; a >= 1; --a) S.hxw[a] = S.hxw[a - 1];
      S.hxw[0] = hxn;
      float rv = acc - S.ypre[U];
      if (EDGE) rv = ((i >= 0) & (i < H)) ? rv : 0.f;
      rv = incol ? rv : 0.f;
      LW(c.lds[L::o_rrow + P * BWP + kPad + c.col], rv);
      int in2 = i + 4;
      if (EDGE) in2 = in2 < 0 ? 0 : (in2 < H ? in2 : H - 1);
      S.ypre[U] = LD(A.y, (size_t)in2 * W + c.colc, (size_t)H * W, 2);
    }
    float hrn = 0.f;
    {
      const float* rp = c.lds + L::o_rrow + (P ^ 1) * BWP + kPad + c.col - ox;
#pragma unroll
      for (int b = 0; b < KT; ++b) hrn = fmaf(uv[kMaxBlur + b], LR(rp[b], S.ypre[1] + (float)b), hrn);
    }
    {
      float acc = uv[KT - 1] * hrn;
#pragma unroll
      for (int a = 0; a < KT - 1; ++a) acc = fmaf(uv[a], S.hrw[KT - 2 - a], acc);
#pragma unroll
      for (int a = KT - 2; a >= 1; --a) S.hrw[a] = S.hrw[a - 1];
      S.hrw[0] = hrn;
      LW(c.lds[L::o_garr + P * BWP + kPad + c.col], A.sigma_f * acc);
    }
  }
#ifndef LMC_EXP_NOBARRIER
  __syncthreads();
#endif
}

// ---- group B tick --------------------------------------------------------------------------------------
template <int K, int NW, int KT, int U, bool EDGE>
__device__ __forceinline__ void split_tick_b(const StepArgs& A, const int t, const int tm, const SplitCtx& c,
                                             StateB<K>& S) {
  using G = SplitGeom<K>;
  using L = SplitLds<K, NW>;
  constexpr int P = U & 1, BWP = L::BWP, KA = G::KA;
  const int H = A.H, W = A.W;
  const bool incol = c.col < W;
  const float* const xb = c.lds + L::o_xring + tm * BWP + kPad + c.col;
  const int o = t - G::D;
  float prox_o = 0.f;

  if constexpr (K > 0) {
    // stage KA's outputs of the previous tick, and the left neighbour of its ss
    const float* hb = c.lds + L::o_hand + (P ^ 1) * 4 * BWP + kPad + c.col;
    S.hrr[P ^ 1] = LR(hb[0], S.hrr[P] + 1.f);
    S.hss[P ^ 1] = LR(hb[BWP], S.hss[P] + 1.f);
    S.hp[P ^ 1] = LR(hb[2 * BWP], S.hp[P] + 1.f);
    S.hq[P ^ 1] = LR(hb[3 * BWP], S.hq[P] + 1.f);
    const float hssl = LR(hb[BWP - 1], S.hss[P] + 2.f);
    const float gam = A.tv.gamma, cstep = A.tv.c;
    float gsolv[16], gssv[16];
    if constexpr (K > KA) {
      load_ghost<(KA + 1) / 4, K / 4>(c.lds + L::o_gsol + (2 + (P ^ 1)) * L::GB + c.wave * 64 + (c.lane & 48) + 16, gsolv);
      load_ghost<(15 - K) / 4, (15 - (KA + 1)) / 4>(c.lds + L::o_gss + (2 + (P ^ 1)) * L::GB + c.wave * 64 + (c.lane & 48) + 48, gssv);
    }
#pragma unroll
    for (int k = K + 1; k > KA; --k) {
      const float xa = LR(xb[(G::RB - (G::E + 2 * k)) * BWP], S.nz[0] + (float)k);
      float rr1, rr2, ssc, ssl;   // rr^{k-1} on rows a, a-1 ; ss^{k-1} on row a and its left neighbour
      if (k - 1 == KA) {
        rr1 = S.hrr[P ^ 1]; rr2 = S.hrr[P]; ssc = S.hss[P ^ 1]; ssl = hssl;
      } else {
        rr1 = S.rr[k - 1][P ^ 1]; rr2 = S.rr[k - 1][P]; ssc = S.ss[k - 1][P ^ 1];
#ifdef LMC_EXP_NOGHOST
        ssl = NB_LEFT(c, ssc, 0.f);
#else
        ssl = NB_LEFT(c, ssc, gssv[15 - (k - 1)]);
#endif
      }
      const float sol = fmaf(-gam, (rr1 - rr2) + (ssc - ssl), xa);
      S.sol[k][P] = sol;
      if (k == K + 1) {
        prox_o = sol;
      } else {
        const float solb = S.sol[k][P ^ 1];
#ifdef LMC_EXP_NOGHOST
        const float solr = NB_RIGHT(c, solb, 0.f);
#else
        const float solr = NB_RIGHT(c, solb, gsolv[k]);
#endif
        float cdown = cstep;
        if (EDGE) {
          const int b = t - G::E - 2 * k - 1;
          cdown = ((unsigned)b >= (unsigned)(H - 1)) ? 0.f : cstep;
        }
        float rrb, ssb, pb, qb;   // stage k-1 on row b (two ticks old)
        if (k - 1 == KA) { rrb = S.hrr[P]; ssb = S.hss[P]; pb = S.hp[P]; qb = S.hq[P]; }
        else { rrb = S.rr[k - 1][P]; ssb = S.ss[k - 1][P]; pb = S.p[k - 1][P]; qb = S.q[k - 1][P]; }
        const float r = fmaf(-cdown, sol - solb, rrb);
        const float s = fmaf(-c.cright, solr - solb, ssb);
        const float inv = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rsqf(fmaf(r, r, s * s)), 0.f, 1.f);   // == rsq(max(., 1)); folds into v_rsq ... clamp
        const float pn = r * inv, qn = s * inv;
        const float beta = A.tv.betas[k - 1];
        S.rr[k][P] = fmaf(beta, pn - pb, pn);
        S.ss[k][P] = fmaf(beta, qn - qb, qn);
        S.p[k][P] = pn;
        S.q[k][P] = qn;
      }
#ifdef LMC_SPLIT_SCHED
      if ((k % LMC_SPLIT_SCHED) == 0) __builtin_amdgcn_sched_barrier(0);
#endif
    }
#ifndef LMC_EXP_NOGHOST
    if constexpr (K > KA) {
      float gs = 0.f, gq = 0.f;
      static_for<KA + 1, K + 1>([&](auto kk) { constexpr int k = decltype(kk)::value; gs = gather_first<k>(gs, S.sol[k][P]); });
      LW(c.lds[L::o_gsol + (2 + P) * L::GB + c.wave * 64 + c.lane], gs);
      static_for<KA + 1, K + 1>([&](auto kk) { constexpr int k = decltype(kk)::value; gq = gather_last<k>(gq, S.ss[k][P]); });
      LW(c.lds[L::o_gss + (2 + P) * L::GB + (c.wave + 1) * 64 + c.lane], gq);
    }
#endif
  }

  constexpr int NI = ((U - G::D) % 4 + 4) % 4;  // == o & 3
  if (A.noise_mode == LMC_NOISE_PHILOX) {
    if (NI == 0 && (!EDGE || (o >= 0 && o < H))) {
      quad_normals(A.key0, A.key1, A.iteration, A.chain_offset + (uint32_t)c.chain,
                   (uint32_t)(o >> 2) * (uint32_t)W + (uint32_t)c.col, S.nz);
    }
  }

  if ((!EDGE || (o >= 0 && o < H)) && incol) {
    const size_t gi = (size_t)o * W + c.col;
    const float x = LR(xb[(G::RB - G::D) * BWP], S.nz[1] + 3.f);
    float g = 0.f;
    if constexpr (KT > 0) {
      g = LR(c.lds[L::o_garr + (P ^ 1) * BWP + kPad + c.col], S.nz[2] + 1.f);
    } else if (A.data_kind == LMC_DATA_IDENTITY) {
      g = A.sigma_f * (x - S.ypre[U]);
    } else if (A.data_kind == LMC_DATA_MASK) {
      const float mk = S.mpre[U];
      g = A.sigma_f * mk * fmaf(mk, x, -S.ypre[U]);
    }
    if (A.ncvx_kind == LMC_NCVX_MC_TV) {   // - lambda * A^T(A x / max(|A x|, gamma))  (algs.py:273-277, 291)
      const float* xm = xb + (G::RB - G::D - 1) * BWP;   // row o-1
      const float* xp = xb + (G::RB - G::D + 1) * BWP;   // row o+1
      const float* x0 = xb + (G::RB - G::D) * BWP;
      g -= A.ncvx_lambda * mc_tv_grad(xm[0], xm[1], x0[-1], x, x0[1], xp[-1], xp[0], o > 0, o + 1 < H, c.col > 0,
                                      c.col + 1 < W, A.ncvx_gamma);
    }
    if (A.extra) g = fmaf(A.extra_coef, x - A.extra[(size_t)c.chain * H * W + gi], g);   // ME-TV term (algs.py:282)
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
    if (A.prox_ext) px = A.prox_ext[(size_t)c.chain * H * W + gi];
    float xi = S.nz[NI];
    if (A.noise_mode == LMC_NOISE_INJECTED) xi = LD(A.noise, (size_t)c.chain * H * W + gi, (size_t)A.C * H * W, 6);
    if (A.noise_mode == LMC_NOISE_NONE) xi = 0.f;
    ST(c.xout, gi, (size_t)H * W, fmaf(A.a, x, fmaf(-A.t, g, fmaf(A.b, px, A.s * xi))), 7);
  }
  if (KT == 0 && (A.data_kind == LMC_DATA_IDENTITY || A.data_kind == LMC_DATA_MASK)) {   // rows o+4 of y (and mask) for tick t+4
    int on = o + 4;
    on = on < 0 ? 0 : (on < H ? on : H - 1);
    S.ypre[U] = LD(A.y, (size_t)on * W + c.colc, (size_t)H * W, 3);
    if (A.data_kind == LMC_DATA_MASK) S.mpre[U] = LD(A.mask, (size_t)on * W + c.colc, (size_t)H * W, 4);
  }
#ifndef LMC_EXP_NOBARRIER
  __syncthreads();
#endif
}

template <int K, int NW, int KT>
__global__ __launch_bounds__(128 * NW) void myula_step_split_kernel(const StepArgs A) {
  extern __shared__ float lds[];
  using G = SplitGeom<K>;
  using L = SplitLds<K, NW>;
  constexpr int BW = 64 * NW, RB = G::RB;
  const int tid = threadIdx.x;
  const int H = A.H, W = A.W;
  const size_t img = (size_t)H * W;
  const bool is_a = tid < BW;

  SplitCtx c;
  c.lds = lds;
  c.chain = blockIdx.x;
  c.col = is_a ? tid : tid - BW;
  c.colc = c.col < W ? c.col : W - 1;
  c.lane = tid & 63;
  c.wave = c.col >> 6;
  c.xin = A.x_in + (size_t)c.chain * img;
  c.xout = A.x_out + (size_t)c.chain * img;
  c.cright = (c.col >= W - 1) ? 0.f : A.tv.c;
  for (int i = tid; i < L::total; i += 2 * BW) lds[i] = 0.f;

  const int T = H + G::D;
  constexpr int t_lo = (G::D + 3) & ~3;
#ifdef LMC_NO_STEADY
  constexpr bool kSteady = false;
#else
  constexpr bool kSteady = true;
#endif
#ifdef LMC_SPLIT_ONLY_B
  if (false) {
#else
  if (is_a) {
#endif
    StateA<K, KT> S;
#pragma unroll
    for (int k = 0; k <= G::KA; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) S.rr[k][j] = S.ss[k][j] = S.p[k][j] = S.q[k][j] = 0.f;
#pragma unroll
    for (int k = 0; k <= G::KA + 1; ++k) S.sol[k][0] = S.sol[k][1] = 0.f;
#pragma unroll
    for (int a = 0; a < KT - 1; ++a) S.hxw[a] = S.hrw[a] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      S.xpre[j] = LD(c.xin, (size_t)(j < H ? j : H - 1) * W + c.colc, img, 8);
      S.ypre[j] = 0.f;
    }
    if constexpr (KT > 0) {
      // residual row of tick t is i(t) = t - D + 1 + KT - oy; ypre[t & 3] holds y[i(t)] when i(t) is a row
      const int i0 = -G::D + 1 + KT - A.blur.oy;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = i0 + j < 0 ? 0 : (i0 + j < H ? i0 + j : H - 1);
        S.ypre[j] = LD(A.y, (size_t)r * W + c.colc, img, 10);
      }
    }
    __syncthreads();
    int tm = 0;
#define LMC_A_GROUP(EDGE_)                                                                                   \
    split_tick_a<K, NW, KT, 0, EDGE_>(A, t0 + 0, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;               \
    split_tick_a<K, NW, KT, 1, EDGE_>(A, t0 + 1, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;               \
    split_tick_a<K, NW, KT, 2, EDGE_>(A, t0 + 2, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;               \
    split_tick_a<K, NW, KT, 3, EDGE_>(A, t0 + 3, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;
    // three loops in sequence: fill (general ticks), steady (predicate-free ticks), drain (general ticks)
    int t0 = 0;
    const int t_hi = kSteady ? ((H - 5 - 3) & ~3) : 0;   // last group start with t0 + 3 + 4 < H (prefetch distance 4)
    for (; t0 < T && (t0 < t_lo || !kSteady || t0 > t_hi); t0 += 4) { LMC_A_GROUP(true) }
    if (kSteady) for (; t0 <= t_hi; t0 += 4) { LMC_A_GROUP(false) }
    for (; t0 < T; t0 += 4) { LMC_A_GROUP(true) }
#undef LMC_A_GROUP
#ifdef LMC_SPLIT_ONLY_A
  } else if (false) {
#else
  } else {
#endif
    StateB<K> S;
    S.hrr[0] = S.hrr[1] = S.hss[0] = S.hss[1] = S.hp[0] = S.hp[1] = S.hq[0] = S.hq[1] = 0.f;
#pragma unroll
    for (int k = 0; k <= K; ++k)
#pragma unroll
      for (int j = 0; j < 2; ++j) S.rr[k][j] = S.ss[k][j] = S.p[k][j] = S.q[k][j] = 0.f;
#pragma unroll
    for (int k = 0; k <= K + 1; ++k) S.sol[k][0] = S.sol[k][1] = 0.f;
    S.nz[0] = S.nz[1] = S.nz[2] = S.nz[3] = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {   // output row of tick j is j - D (< 0): slots are refilled before they are used
      S.ypre[j] = 0.f;
      S.mpre[j] = 0.f;
    }
    if (KT == 0 && (A.data_kind == LMC_DATA_IDENTITY || A.data_kind == LMC_DATA_MASK)) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        int o0 = j - G::D;
        o0 = o0 < 0 ? 0 : (o0 < H ? o0 : H - 1);
        S.ypre[j] = LD(A.y, (size_t)o0 * W + c.colc, img, 3);
        if (A.data_kind == LMC_DATA_MASK) S.mpre[j] = LD(A.mask, (size_t)o0 * W + c.colc, img, 4);
      }
    }
    __syncthreads();
    int tm = 0;
#define LMC_B_GROUP(EDGE_)                                                                                   \
    split_tick_b<K, NW, KT, 0, EDGE_>(A, t0 + 0, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;               \
    split_tick_b<K, NW, KT, 1, EDGE_>(A, t0 + 1, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;               \
    split_tick_b<K, NW, KT, 2, EDGE_>(A, t0 + 2, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;               \
    split_tick_b<K, NW, KT, 3, EDGE_>(A, t0 + 3, tm, c, S); tm = (tm + 1 == RB) ? 0 : tm + 1;
    int t0 = 0;
    const int t_hi = kSteady ? ((H - 5 - 3) & ~3) : 0;
    for (; t0 < T && (t0 < t_lo || !kSteady || t0 > t_hi); t0 += 4) { LMC_B_GROUP(true) }
    if (kSteady) for (; t0 <= t_hi; t0 += 4) { LMC_B_GROUP(false) }
    for (; t0 < T; t0 += 4) { LMC_B_GROUP(true) }
#undef LMC_B_GROUP
  }
}

// ---- host side -------------------------------------------------------------------------------------------

static int split_nw(int W) { return W <= 64 ? 1 : (W <= 128 ? 2 : (W <= 256 ? 4 : (W <= 512 ? 8 : 0))); }

template <int K>
static bool split_fits(int NW) {
  const int BWP = 64 * NW + 2 * kPad;
  const size_t total = NW == 1 ? SplitLds<K, 1>::total : NW == 2 ? SplitLds<K, 2>::total
                     : NW == 4 ? SplitLds<K, 4>::total : SplitLds<K, 8>::total;
  return (size_t)SplitGeom<K>::RB * BWP * sizeof(float) + 64 <= 65535 && total * sizeof(float) <= 160 * 1024;
}

template <int K, int NW, int KT>
static hipError_t launch_split_knw(const StepArgs& a, hipStream_t st) {
  auto k = myula_step_split_kernel<K, NW, KT>;
  constexpr size_t lds = sizeof(float) * (size_t)SplitLds<K, NW>::total;
  static thread_local bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)lds);
    if (e != hipSuccess) return e;
    configured = true;
  }
  hipLaunchKernelGGL(k, dim3(a.C), dim3(128 * NW), lds, st, a);
  return hipGetLastError();
}

template <int K, int KT>
static hipError_t launch_split_k(const StepArgs& a, hipStream_t st) {
  switch (split_nw(a.W)) {
    case 1: return launch_split_knw<K, 1, KT>(a, st);
    case 2: return launch_split_knw<K, 2, KT>(a, st);
    case 4: return launch_split_knw<K, 4, KT>(a, st);
    case 8: return launch_split_knw<K, 8, KT>(a, st);
  }
  return hipErrorInvalidConfiguration;
}

template <int K>
static hipError_t launch_split_kt(const StepArgs& a, int KT, hipStream_t st) {
  if (KT == 0) return launch_split_k<K, 0>(a, st);   // no blur: pointwise (identity / mask / none) data term
  if (KT <= 5) return launch_split_k<K, 5>(a, st);
  return launch_split_k<K, 7>(a, st);
}

bool separate_blur_taps(const BlurTaps& T, float* u, float* v);  // lmc_step_rows.hip

static int split_k(const StepArgs& a) { return a.prior_kind == LMC_PRIOR_TV_ISO ? a.tv.niter : 0; }

bool split_supported(const StepArgs& a) {
  const int NW = split_nw(a.W);
  if (NW == 0 || a.H < 1) return false;
  bool fits = false;
  switch (split_k(a)) {
#ifndef LMC_ONLY_K10
    case 0: fits = split_fits<0>(NW); break;
    case 1: fits = split_fits<1>(NW); break;
    case 2: fits = split_fits<2>(NW); break;
    case 3: fits = split_fits<3>(NW); break;
    case 4: fits = split_fits<4>(NW); break;
    case 5: fits = split_fits<5>(NW); break;
    case 6: fits = split_fits<6>(NW); break;
    case 8: fits = split_fits<8>(NW); break;
    case 9: fits = split_fits<9>(NW); break;      // the lagged reading of niter = 10
    case 12: fits = split_fits<12>(NW); break;
#endif
    case 10: fits = split_fits<10>(NW); break;
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

hipError_t launch_step_split(StepArgs a, hipStream_t st) {
  int KT = 0;
  if (a.data_kind == LMC_DATA_BLUR) {
    float u[kMaxBlur] = {0}, v[kMaxBlur] = {0};
    if (!separate_blur_taps(a.blur, u, v)) return hipErrorInvalidConfiguration;
    for (int i = 0; i < kMaxBlur; ++i) { a.blur.h[i] = u[i]; a.blur.h[kMaxBlur + i] = v[i]; }
    KT = (a.blur.kh > a.blur.kw ? a.blur.kh : a.blur.kw) <= 5 ? 5 : 7;
  }
  switch (split_k(a)) {
#ifndef LMC_ONLY_K10
    case 0: return launch_split_kt<0>(a, KT, st);
    case 1: return launch_split_kt<1>(a, KT, st);
    case 2: return launch_split_kt<2>(a, KT, st);
    case 3: return launch_split_kt<3>(a, KT, st);
    case 4: return launch_split_kt<4>(a, KT, st);
    case 5: return launch_split_kt<5>(a, KT, st);
    case 6: return launch_split_kt<6>(a, KT, st);
    case 8: return launch_split_kt<8>(a, KT, st);
    case 9: return launch_split_kt<9>(a, KT, st);
    case 12: return launch_split_kt<12>(a, KT, st);
#endif
    case 10: return launch_split_kt<10>(a, KT, st);
  }
  return hipErrorInvalidConfiguration;
}

}  // namespace lmc
