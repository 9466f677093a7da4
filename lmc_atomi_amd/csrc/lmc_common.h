// Shared host/device declarations for liblmc_atomi (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "lmc_atomi.h"

namespace lmc {

constexpr int kMaxBlur = LMC_MAX_BLUR;
constexpr int kMaxTvIters = LMC_MAX_TV_ITERS;
constexpr int kWave = 64;  // CDNA4 wavefront

// Blur taps live in the kernel-argument segment (scalar loads, uniform over the grid).
struct BlurTaps {
  int kh, kw, oy, ox;
  float h[kMaxBlur * kMaxBlur];
};

struct TvIter {
  int niter;
  float gamma;   // prox parameter * prior_sigma
  float c;       // dual step = tv_step / gamma
  float betas[kMaxTvIters];
};

// out = a*x - t*grad f(x) + b*prox(x) + s*xi
struct StepArgs {
  int H, W, C;
  int tiles_x, tiles_y, TH, TW, HL, PH, PW;
  int data_kind;
  float sigma_f;
  const float* y;
  const float* mask;
  BlurTaps blur;
  int prior_kind;
  float prior_p0;     // L2: 1/(1+t*sigma) ; L1: threshold t*sigma ; EPROX: first parameter of the closed form
  float prior_p1;     // EPROX: second parameter
  int eprox_kind;     // EPROX: lmc_eprox_kind
  TvIter tv;
  int ncvx_kind;             // LMC_NCVX_*: extra term of the data gradient (algs.py:270-291)
  float ncvx_lambda, ncvx_inv_gamma, ncvx_gamma;
  float a, t, b, s;
  int noise_mode;
  const float* noise;        // [C][H][W] of this iteration (injected)
  uint32_t key0, key1;       // Philox key = seed
  uint32_t iteration;        // Philox counter word 1 (+ *iter_dev when that is set)
  const uint32_t* iter_dev;  // rows / block / pipe kernels: device-resident iteration base added to `iteration` -- lets a captured hipGraph of
                             // iterations be replayed (kernel arguments are frozen in a graph; the counter word must advance)
  uint32_t chain_offset;     // global id of chain 0 (counter word 2 = chain_offset + c)
  const float* x_in;
  float* x_out;
  // resumable TV prox (tile kernel): dual state [C][4][H][W] = (rr, ss, p, q) carried between launches so that
  // more dual iterations than one launch's halo allows can be chained exactly; tv.betas is already offset.
  const float* tv_in;        // NULL: start from zero
  float* tv_out;             // NULL: do not store
  int tv_state_only;         // != 0: store the state and skip the combine / x_out (non-final chunk)
  int tv_warm;               // pipe kernel: tv_in / tv_out are [C][2][H][W] = the projected dual (p, q) carried from one MYULA
                             // iteration to the next (warm-started TV prox, momentum restarted) instead of the 4-field link state
  // extra gradient term g += extra_coef * (x - extra[c][i][j])   (ME-TV: extra = prox_{gamma TV}(x), algs.py:282)
  const float* extra;
  float extra_coef;
  // rows kernel only: dot_out[c] += sum_ij x_in[c][i][j] * x_out[c][i][j] (the p.Ap of a CG iteration, fused into the operator apply)
  double* dot_out;
  const int* skip_flag;
  // rows kernel only: dot_mode 1 turns the fused reduction into the residual statistics of a Chebyshev step (lmc_capi.hip:
  // chebyshev_solve): dot_out[2c] += sum (x_out - x_in)^2, dot_out[2c+1] += sum prox_ext^2.  run_count / run_index: the launch returns at
  // once when *run_count <= run_index (the iterations a warm-started solve turned out not to need).
  int dot_mode;
  const int* run_count;
  int run_index;
  // pipe kernel only: energies of x_in as by-products of the update (MYMALA): f_out[c] += sigma_f/2 ||H x - y||^2 (the residual rows
  // of the blur pipeline), g_out[c] += g_scale * TV_iso(x) (from the ring rows of the combine wave)
  double* f_out;
  double* g_out;
  float g_scale;      // rows kernel only: the launch returns at once when *skip_flag != 0 (inner solver already converged)
  // prox computed by a preceding launch (Haar-l1 wavelet prior): px = prox_ext[c][i][j]; the kernel's own prior is NONE
  const float* prox_ext;
  // block kernel only (Haar prior, no MC-TV term): fused_iters = 2 runs TWO iterations on the thread's 8 x 8 block before it goes back to memory
  // (iterations `iteration` and `iteration + 1`; x_out <- x_{k+2}); x_mid (may be NULL, may be x_in itself: the update is block-local) <- x_{k+1}
  int fused_iters;
  float* x_mid;
  // pipe kernel, RT instantiations only: per-chain early exit of the TV prox (pyproximal.TV's rtol; lmc_problem.tv_rtol / ncvx_rtol).  A chain's
  // workgroup runs rt_kc[c] dual updates in all (counted over the links of a chained prox; this launch holds updates rt_base + 1 .. rt_base + K),
  // the later stages pass the dual through; it returns at once when rt_kc[c] <= rt_base (the chain left in an earlier link, or needs no run:
  // rt_kc[c] < 0), or when this link lies before rt_start[c] (its result of an earlier round is still valid).  By-products: the primal objective 0.5 ||x - sol_j||^2 + gamma TV(sol_j) of every iterate formed -- stage g adds that of
  // its input iterate to rt_obj[c][g - 1], the combine wave that of the iterate it returns to rt_obj[c][rt_kc[c]] (unless that is rt_total,
  // the iterate returned untested).  fp64 sums; rt_obj must be zero where this launch adds.
  const int* rt_kc;
  const int* rt_start;     // [C] (chained prox) first link chain c needs in this round: the links before it return at once -- their work of an earlier round stands
  double* rt_obj;
  int rt_stride, rt_base, rt_total;
};

constexpr uint32_t kPhiloxStream = 0x4C4D4301u;  // counter word 3 (noise field)
constexpr uint32_t kPhiloxAccept = 0x4C4D4302u;  // counter word 3 (Metropolis uniforms: ctr = (0, iteration, chain, this))

}  // namespace lmc
