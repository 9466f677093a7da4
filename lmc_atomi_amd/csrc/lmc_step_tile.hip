// Fused MYULA update, LDS-tiled variant ("tile"): one workgroup = one (TH x TW) output tile of
// one chain.  out = a*x - t*grad f(x) + b*prox_g(x) + s*xi, everything between the HBM read of
// x (tile + halo) and the HBM write of x' stays on chip:
//   * blur residual R = Hx - y and its adjoint H^T R from LDS-staged stencils,
//   * K fast-gradient-projection iterations of the TV prox with the dual field in LDS/registers,
//   * Philox4x32-10 + Box-Muller noise in registers (one call per 4 vertically adjacent pixels).
// Reference update: algs.py:569 (MoreauYosidaUnadjustedLangevin).
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

constexpr int kStepThreads = 1024;

// LDS image: (PH + 2) rows of PW floats; row -1 and row PH are pad rows so that the +-1 neighbour
// reads of tile-border pixels stay inside the allocation (their values never reach the interior:
// information moves one pixel per dual iteration and the halo is >= niter).
template <int NP, bool TV>
__global__ __launch_bounds__(kStepThreads) void myula_step_tile_kernel(const StepArgs P) {
  extern __shared__ float lds[];
  const int tid = threadIdx.x;
  const int logical = xcd_logical_block(blockIdx.x, gridDim.x);
  const int tiles = P.tiles_x * P.tiles_y;
  const int chain = logical / tiles;
  const int tile = logical - chain * tiles;
  const int ty = tile / P.tiles_x, tx = tile - ty * P.tiles_x;
  const int H = P.H, W = P.W, PW = P.PW, PH = P.PH, HL = P.HL;
  const int row0 = ty * P.TH - HL, col0 = tx * P.TW - HL;  // image coords of tile pixel (0,0)
  const int npix = PH * PW;
  const int arr = (PH + 2) * PW;
  float* xs = lds + PW;   // x tile (+halo), zero outside the image
  float* S = xs + arr;    // blur residual, then TV primal iterate ("sol")
  float* A = S + arr;     // dual, row component    (TV only)
  float* B = A + arr;     // dual, column component (TV only)

  const size_t img = (size_t)H * W;
  const float* __restrict__ xin = P.x_in + (size_t)chain * img;

  // ---- phase 0: stage x tile; each thread owns pixels p = tid + m*1024 -------------------
  float xv[NP];
  int flags[NP];  // bit0 in image, bit1 has-down, bit2 has-right, bit3 p < npix
#pragma unroll
  for (int m = 0; m < NP; ++m) {
    const int p = tid + m * kStepThreads;
    float v = 0.f;
    int f = 0;
    if (p < npix) {
      const int r = p / PW, c = p - r * PW;
      const int gr = row0 + r, gc = col0 + c;
      const bool in = (gr >= 0) & (gr < H) & (gc >= 0) & (gc < W);
      if (in) v = xin[(size_t)gr * W + gc];
      f = 8 | (in ? 1 : 0) | ((in && gr + 1 < H) ? 2 : 0) | ((in && gc + 1 < W) ? 4 : 0);
      xs[p] = v;
      if (TV) { A[p] = 0.f; B[p] = 0.f; }
    }
    xv[m] = v;
    flags[m] = f;
  }
  if (TV) {  // zero the pad rows of the dual arrays
    for (int i = tid; i < PW; i += kStepThreads) {
      A[-PW + i] = 0.f; B[-PW + i] = 0.f; A[npix + i] = 0.f; B[npix + i] = 0.f;
      S[-PW + i] = 0.f; S[npix + i] = 0.f;
    }
  }
  __syncthreads();

  // interior ownership for gradient / combine: thread -> column ci, rows 4*rg .. 4*rg+3
  const int ci = tid % P.TW, rg = tid / P.TW;
  const bool own_int = rg < (P.TH >> 2);
  float gv[4] = {0.f, 0.f, 0.f, 0.f};

  // ---- phase 1+2: grad f = sigma_f * H^T (H x - y) ----------------------------------------
  if (P.data_kind == LMC_DATA_BLUR) {
    const int kh = P.blur.kh, kw = P.blur.kw, oy = P.blur.oy, ox = P.blur.ox;
    const int r_lo = HL - oy, r_hi = HL + P.TH + kh - 1 - oy;  // rows of R needed by H^T
    const int c_lo = HL - ox, c_hi = HL + P.TW + kw - 1 - ox;
#pragma unroll
    for (int m = 0; m < NP; ++m) {
      const int p = tid + m * kStepThreads;
      if (p < npix) {
        const int r = p / PW, c = p - r * PW;
        if (r >= r_lo && r < r_hi && c >= c_lo && c < c_hi) {
          float acc = 0.f;
          if (flags[m] & 1) {
            for (int a = 0; a < kh; ++a)
              for (int b = 0; b < kw; ++b)
                acc = fmaf(P.blur.h[a * kw + b], xs[(r - a + oy) * PW + (c - b + ox)], acc);
            acc -= P.y[(size_t)(row0 + r) * W + (col0 + c)];
          }
          S[p] = acc;  // residual, zero outside the image (zero-padded adjoint)
        }
      }
    }
    __syncthreads();
    if (own_int) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = HL + 4 * rg + j, c = HL + ci;
        float acc = 0.f;
        for (int a = 0; a < kh; ++a)
          for (int b = 0; b < kw; ++b)
            acc = fmaf(P.blur.h[a * kw + b], S[(r + a - oy) * PW + (c + b - ox)], acc);
        gv[j] = P.sigma_f * acc;
      }
    }
    __syncthreads();
  }

  // ---- phase 3: TV prox, K fast-gradient-projection dual iterations -----------------------
  if (TV) {
    float rrv[NP], ssv[NP], pv[NP], qv[NP], solv[NP];
#pragma unroll
    for (int m = 0; m < NP; ++m) { rrv[m] = ssv[m] = pv[m] = qv[m] = 0.f; }
    if (P.tv_in) {   // resume: dual state of the previous launch (tile + halo), zero outside the image
      const float* __restrict__ st = P.tv_in + (size_t)chain * 4 * img;
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        const int p = tid + m * kStepThreads;
        if (p < npix && (flags[m] & 1)) {
          const int r = p / PW, c = p - r * PW;
          const size_t gi = (size_t)(row0 + r) * W + (col0 + c);
          rrv[m] = st[gi]; ssv[m] = st[img + gi]; pv[m] = st[2 * img + gi]; qv[m] = st[3 * img + gi];
          A[p] = rrv[m]; B[p] = ssv[m];
        }
      }
      __syncthreads();
    }
    const float gam = P.tv.gamma, cstep = P.tv.c;
    for (int k = 0; k <= P.tv.niter; ++k) {
      // A-phase: sol = x - gamma * div(rr, ss)
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        const int p = tid + m * kStepThreads;
        if (p < npix) {
          const float dv = (rrv[m] - A[p - PW]) + (ssv[m] - B[p - 1]);
          solv[m] = fmaf(-gam, dv, xv[m]);
          S[p] = solv[m];
        }
      }
      __syncthreads();
      if (k == P.tv.niter) break;
      const float beta = P.tv.betas[k];
      // B-phase: dual ascent step, projection onto the unit ball, momentum
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        const int p = tid + m * kStepThreads;
        if (p < npix) {
          const float dx = (flags[m] & 2) ? S[p + PW] - solv[m] : 0.f;
          const float dy = (flags[m] & 4) ? S[p + 1] - solv[m] : 0.f;
          const float r = fmaf(-cstep, dx, rrv[m]);
          const float s = fmaf(-cstep, dy, ssv[m]);
          const float inv = rsqrtf(fmaxf(fmaf(r, r, s * s), 1.f));
          const float pn = r * inv, qn = s * inv;
          rrv[m] = fmaf(beta, pn - pv[m], pn);
          ssv[m] = fmaf(beta, qn - qv[m], qn);
          pv[m] = pn;
          qv[m] = qn;
          A[p] = rrv[m];
          B[p] = ssv[m];
        }
      }
      __syncthreads();
    }
    if (P.tv_out) {   // store the dual state of the tile interior for the next launch
      float* __restrict__ st = P.tv_out + (size_t)chain * 4 * img;
#pragma unroll
      for (int m = 0; m < NP; ++m) {
        const int p = tid + m * kStepThreads;
        if (p < npix && (flags[m] & 1)) {
          const int r = p / PW, c = p - r * PW;
          if (r >= HL && r < HL + P.TH && c >= HL && c < HL + P.TW) {
            const size_t gi = (size_t)(row0 + r) * W + (col0 + c);
            st[gi] = rrv[m]; st[img + gi] = ssv[m]; st[2 * img + gi] = pv[m]; st[3 * img + gi] = qv[m];
          }
        }
      }
    }
  }
  if (P.tv_state_only) return;

  // ---- phase 4: combine + noise + store ----------------------------------------------------
  if (!own_int) return;
  const int gc = col0 + HL + ci;
  const int gr0 = row0 + HL + 4 * rg;
  if (gc >= W || gr0 >= H) return;
  float xi[4] = {0.f, 0.f, 0.f, 0.f};
  if (P.noise_mode == LMC_NOISE_PHILOX) {
    // gr0 is a multiple of 4 because TH is: the quad of rows gr0..gr0+3
    quad_normals(P.key0, P.key1, P.iteration, P.chain_offset + (uint32_t)chain,
                 (uint32_t)(gr0 >> 2) * (uint32_t)W + (uint32_t)gc, xi);
  }
  float* __restrict__ xout = P.x_out + (size_t)chain * img;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int gr = gr0 + j;
    if (gr >= H) break;
    const int p = (HL + 4 * rg + j) * PW + HL + ci;
    const size_t gi = (size_t)gr * W + gc;
    const float x = xs[p];
    float g = gv[j];
    if (P.data_kind == LMC_DATA_IDENTITY) {
      g = P.sigma_f * (x - P.y[gi]);
    } else if (P.data_kind == LMC_DATA_MASK) {
      const float mk = P.mask[gi];
      g = P.sigma_f * mk * fmaf(mk, x, -P.y[gi]);
    }
    if (P.ncvx_kind == LMC_NCVX_MC_TV) {   // - lambda * A^T(A x / max(|A x|, gamma))  (algs.py:273-277, 291)
      g -= P.ncvx_lambda * mc_tv_grad(xs[p - PW], xs[p - PW + 1], xs[p - 1], x, xs[p + 1], xs[p + PW - 1], xs[p + PW],
                                      gr > 0, gr + 1 < H, gc > 0, gc + 1 < W, P.ncvx_gamma);
    }
    if (P.extra) g = fmaf(P.extra_coef, x - P.extra[(size_t)chain * img + gi], g);   // ME-TV: -lambda/gamma (x - prox_{gamma TV}(x))
    float px;
    if (TV) {
      px = S[p];
    } else if (P.prior_kind == LMC_PRIOR_L2) {
      px = x * P.prior_p0;
    } else if (P.prior_kind == LMC_PRIOR_L1) {
      px = copysignf(fmaxf(fabsf(x) - P.prior_p0, 0.f), x);
    } else {
      px = x;
    }
    if (P.prox_ext) px = P.prox_ext[(size_t)chain * img + gi];
    float nz = xi[j];
    if (P.noise_mode == LMC_NOISE_INJECTED) nz = P.noise[(size_t)chain * img + gi];
    xout[gi] = fmaf(P.a, x, fmaf(-P.t, g, fmaf(P.b, px, P.s * nz)));
  }
}

// ---- host-side launcher --------------------------------------------------------------------

struct TilePlan {
  int TH, TW, HL, PH, PW, NP;
  size_t lds_bytes;
};

static bool plan_tiles(int H, int W, int halo, bool tv, size_t lds_limit, TilePlan& out) {
  const int n_arr = tv ? 4 : 2;
  // candidate output tiles, largest first; TW*TH/4 <= 1024 threads, TH % 4 == 0
  static const int cand[][2] = {{64, 64}, {48, 64}, {32, 64}, {32, 32}, {16, 32}, {16, 16}, {8, 16}, {4, 16}};
  for (auto& c : cand) {
    const int TH = c[0], TW = c[1];
    const int PH = TH + 2 * halo, PW = TW + 2 * halo;
    const size_t bytes = (size_t)n_arr * (PH + 2) * PW * sizeof(float) + 64;
    const int NP = (PH * PW + kStepThreads - 1) / kStepThreads;
    if (bytes <= lds_limit && NP <= 8) {
      out = {TH, TW, halo, PH, PW, NP, bytes};
      return true;
    }
  }
  return false;
}

template <int NP, bool TV>
static hipError_t launch_np(const StepArgs& a, size_t lds, hipStream_t st) {
  auto k = myula_step_tile_kernel<NP, TV>;
  static thread_local size_t configured = 0;
  if (lds > configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    configured = lds;
  }
  const int nblk = a.tiles_x * a.tiles_y * a.C;
  hipLaunchKernelGGL(k, dim3(nblk), dim3(kStepThreads), lds, st, a);
  return hipGetLastError();
}

template <bool TV>
static hipError_t launch_tv(const StepArgs& a, int NP, size_t lds, hipStream_t st) {
  switch (NP) {
    case 1: return launch_np<1, TV>(a, lds, st);
    case 2: return launch_np<2, TV>(a, lds, st);
    case 3: return launch_np<3, TV>(a, lds, st);
    case 4: return launch_np<4, TV>(a, lds, st);
    case 5: return launch_np<5, TV>(a, lds, st);
    case 6: return launch_np<6, TV>(a, lds, st);
    case 7: return launch_np<7, TV>(a, lds, st);
    case 8: return launch_np<8, TV>(a, lds, st);
  }
  return hipErrorInvalidValue;
}

// Fills the tile geometry of `a` (needs H, W, data/prior fields set) and launches.
// Returns hipErrorInvalidConfiguration if no tile fits (halo too large).
hipError_t launch_step_tile(StepArgs a, hipStream_t st) {
  const bool tv = a.prior_kind == LMC_PRIOR_TV_ISO;
  int halo = 0;
  if (a.data_kind == LMC_DATA_BLUR) halo = max(a.blur.kh, a.blur.kw) - 1;
  // Dual iterations spread one pixel per iteration in every direction: K iterations from the ZERO dual state leave sol^K valid on the tile when the
  // halo is K (the first primal iterate is x itself).  A chunk that RESUMES from a stored state first forms sol^0 = x - gamma div(state), which already
  // reaches one pixel up / left: it needs K + 1.  (Round 3: with a halo of K the final chunk's prox was wrong along the top row / left column of every
  // tile by an amount that decays ~6x per iteration of that chunk -- 1e-3 for a last chunk of one iteration (K = 17), 1e-9 for a full one.)
  if (tv) halo = max(halo, a.tv.niter + (a.tv_in ? 1 : 0));
  if (a.ncvx_kind != LMC_NCVX_NONE) halo = max(halo, 1);
  TilePlan tp;
  if (!plan_tiles(a.H, a.W, halo, tv, 160 * 1024, tp)) return hipErrorInvalidConfiguration;
  a.TH = tp.TH; a.TW = tp.TW; a.HL = tp.HL; a.PH = tp.PH; a.PW = tp.PW;
  a.tiles_x = (a.W + tp.TW - 1) / tp.TW;
  a.tiles_y = (a.H + tp.TH - 1) / tp.TH;
  return tv ? launch_tv<true>(a, tp.NP, tp.lds_bytes, st) : launch_tv<false>(a, tp.NP, tp.lds_bytes, st);
}


// TV prox with more dual iterations than one launch's halo allows: chunks of <= kTvChunk iterations chained exactly
// through the dual state (rr, ss, p, q) in HBM (two ping-pong buffers of [C][4][H][W] floats each).  Non-final chunks
// only advance the state; the final chunk also does the blur gradient, the combine and the store.
constexpr int kTvChunk = 8;

bool tile_needs_chunks(const StepArgs& a) {
  return a.prior_kind == LMC_PRIOR_TV_ISO && a.tv.niter > 12;
}

hipError_t launch_step_tile_chunked(const StepArgs& a, float* state0, float* state1, hipStream_t st) {
  const int K = a.tv.niter;
  const int n_chunks = (K + kTvChunk - 1) / kTvChunk;
  float* state[2] = {state0, state1};
  for (int ch = 0; ch < n_chunks; ++ch) {
    const int k0 = ch * kTvChunk, kn = (K - k0 < kTvChunk) ? K - k0 : kTvChunk;
    const bool last = ch == n_chunks - 1;
    StepArgs b = a;
    b.tv.niter = kn;
    for (int i = 0; i < kn; ++i) b.tv.betas[i] = a.tv.betas[k0 + i];
    b.tv_in = ch > 0 ? state[(ch - 1) & 1] : nullptr;
    b.tv_out = last ? nullptr : state[ch & 1];
    b.tv_state_only = last ? 0 : 1;
    if (!last) { b.data_kind = LMC_DATA_NONE; b.ncvx_kind = LMC_NCVX_NONE; b.extra = nullptr; }
    hipError_t e = launch_step_tile(b, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

// per-image sum of squared differences ||a_i - b_i||^2 (ME-TV envelope value), one atomic per block
__global__ __launch_bounds__(256) void sqdiff_kernel(const float* __restrict__ a, const float* __restrict__ b, size_t img,
                                                     double* __restrict__ out) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.x;      // images on gridDim.x (no 65535 limit)
  double acc = 0.0;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x) {
    const double d = (double)a[c * img + k] - (double)b[c * img + k];
    acc += d * d;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&out[c], t);
}

__global__ void axpy_env_kernel(double* f, const double* tvv, const double* sq, int64_t n, float lambda, float gamma) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[i] -= (double)lambda * (tvv[i] + sq[i] / (2.0 * (double)gamma));
}

hipError_t launch_axpy_env(double* f, const double* tvv, const double* sq, int64_t n, float lambda, float gamma, hipStream_t st) {
  hipLaunchKernelGGL(axpy_env_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, f, tvv, sq, n, lambda, gamma);
  return hipGetLastError();
}

hipError_t launch_sqdiff(const float* a, const float* b, int64_t n_img, size_t img, double* out, hipStream_t st) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(double) * n_img, st);
  if (e != hipSuccess) return e;
  int gx = (int)((img + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(sqdiff_kernel, dim3((unsigned)n_img, gx), dim3(256), 0, st, a, b, img, out);
  return hipGetLastError();
}

}  // namespace lmc
