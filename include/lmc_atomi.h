/* lmc_atomi.h -- C ABI of the MI355X-native LMC hot path (liblmc_atomi.so).
 *
 * The reference (192459/lmc-atomi) is pure Python and has no FFI: its hot path sits behind
 * duck-typed Python protocols.  This header is the boundary a maintainer of the reference
 * would bind (ctypes stub in INTEGRATION.md); each entry point names the reference
 * interface it replaces (file:line relative to the reference root).
 *
 * Conventions
 *  - plain C, no HIP/torch types: `stream` is a hipStream_t passed as void* (NULL = default
 *    stream); pointers ending in _dev are device (HBM) pointers, _host are host pointers.
 *  - every function returns 0 (LMC_OK) or a negative lmc_status; lmc_last_error() gives the
 *    thread-local message of the last failure.  No exception crosses the boundary.
 *  - images are fp32, row-major [H][W] (W fastest); chain states are [C][H][W];
 *    the stacked gradient field is [2][H][W] per image (row-differences first), as the
 *    reference's 2n vectors (algs.py:427).
 *  - the caller owns every buffer it passes; a sampler owns its state/scratch buffers.
 *  - calls on one sampler handle must be serialised by the caller; all work is enqueued on
 *    the given stream; functions that return host values synchronise that stream.
 *  - devices: a sampler lives on the device that is current when it is created and every call on the handle runs there
 *    (the library switches to that device for the duration of the call and restores the caller's; `stream` and the buffers
 *    passed must belong to it).  The stateless operator entry points run on the caller's current device.
 */
#ifndef LMC_ATOMI_H
#define LMC_ATOMI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LMC_ATOMI_ABI_VERSION 3

typedef enum lmc_status {
  LMC_OK = 0,
  LMC_E_INVALID = -1,     /* bad argument / inconsistent configuration */
  LMC_E_UNSUPPORTED = -2, /* valid request this build has no kernel for */
  LMC_E_HIP = -3,         /* a HIP runtime call failed */
  LMC_E_NOMEM = -4,
  LMC_E_STATE = -5        /* call not allowed in the handle's current state */
} lmc_status;

/* data-fidelity term f(x) = sigma_f/2 * || Op x - y ||^2  (pyproximal.L2 as built at
 * prox_lmc_deconv.py:101-103; same formulae in-repo at algs.py:182-187, 283-288) */
typedef enum lmc_data_kind {
  LMC_DATA_NONE = 0,     /* f = 0 */
  LMC_DATA_IDENTITY = 1, /* Op = I */
  LMC_DATA_BLUR = 2,     /* Op = zero-padded "same" convolution, pylops Convolve2D (prox_lmc_deconv.py:55-69) */
  LMC_DATA_MASK = 3      /* Op = diag(mask), inpainting (BASELINE config 5) */
} lmc_data_kind;

/* prior g whose prox enters the MYULA update (algs.py:569) */
typedef enum lmc_prior_kind {
  LMC_PRIOR_NONE = 0,   /* prox = identity */
  LMC_PRIOR_L2 = 1,     /* g = sigma/2 ||x||^2      : prox_t = x / (1 + t*sigma)          */
  LMC_PRIOR_L1 = 2,     /* g = sigma ||x||_1        : soft threshold t*sigma  (prox.py:18) */
  LMC_PRIOR_TV_ISO = 3, /* g = sigma TV_iso(x)      : tv_niter FGP dual iterations (pyproximal.TV, prox_lmc_deconv.py:122);
                         *  in ULPDA: g o A with g = sigma*L21 (prox_lmc_deconv.py:116), dual prox = l2-ball projection */
  LMC_PRIOR_TV_ANISO = 4, /* ULPDA / energies only: g o A with g = sigma*L1 (prox_lmc_deconv.py:119), dual prox = clip */
  LMC_PRIOR_HAAR_L1 = 5, /* g = sigma * || detail coefficients of the 3-level orthonormal Haar transform of x ||_1 (BASELINE
                          * config 5; no counterpart in the reference): prox = W^T soft(W x, t*sigma); H, W multiples of 8 */
  LMC_PRIOR_EPROX = 6    /* (ABI 3) a separable prior whose prox is one of the closed forms of prox.py (lmc_eprox_kind below; prox.py:18-85, used
                          * inside the reference's samplers at prox_lmc.py:106,115): prox(x) = prox_X(x; p0, p1) pixel by pixel, evaluated inside
                          * the fused step kernel.  lmc_problem.eprox_kind / eprox_p0 / eprox_p1; eprox_scale_mask bit i set = parameter i is
                          * multiplied by the prox parameter (epsg * gamma in MYULA) -- e.g. prox_laplace(x, gamma * lam): p0 = lam, mask = 1.
                          * MYULA / MYMALA steps and lmc_fused_eval; its value g(x) is not defined for every family: lmc_energies returns g = 0. */
} lmc_prior_kind;

typedef enum lmc_ncvx_kind {
  LMC_NCVX_NONE = 0,
  LMC_NCVX_MC_TV = 1,   /* minimax-concave TV (Moreau envelope of l1 composed with the gradient), isotropic */
  LMC_NCVX_ME_TV = 2,   /* Moreau envelope of isotropic TV itself (Op2 = None): grad env = (x - prox_{gamma TV}(x))/gamma,
                         * prox by ncvx_niter FGP iterations (niter_l2 = 50 at prox_lmc_deconv.py:111; algs.py:169,282) */
  LMC_NCVX_MC_TV_ANISO = 3,  /* the anisotropic MC-TV branches (isotropic = False, Op2 = gradient; algs.py:218-219, 278-279): the Moreau
                              * envelope of l1 on every component of the gradient: grad = A^T clip(A x / gamma, -1, 1) */
  LMC_NCVX_ME_TV_ANISO = 4   /* (ABI 3) the anisotropic ME-TV branch (isotropic = False, Op2 = None; algs.py:170): the Moreau envelope of the 1-D TV of
                              * the FLATTENED image (row-major, differences across row ends included), inner prox by ncvx_niter 1-D FGP iterations with
                              * the early exit ncvx_rtol.  Plain coverage (one pass over the images per dual iteration; with ncvx_rtol > 0 the host reads
                              * a counter after every pass): no model of the reference's driver uses it. */
} lmc_ncvx_kind;

typedef enum lmc_noise_mode {
  LMC_NOISE_PHILOX = 0,   /* counter-based Philox4x32-10 + Box-Muller, keyed by (seed, iteration, global chain, pixel) */
  LMC_NOISE_INJECTED = 1, /* caller supplies xi (parity tests: replaces algs.py:565) */
  LMC_NOISE_NONE = 2
} lmc_noise_mode;

#define LMC_MAX_BLUR 9       /* kernels up to 9x9 */
#define LMC_MAX_TV_ITERS 64

/* Geometry + potential U(x) = f(x) + eps*g(x).  Plain data; copied by the callee. */
typedef struct lmc_problem {
  uint32_t struct_size;     /* = sizeof(lmc_problem) */
  int32_t H, W;
  /* data term */
  int32_t data_kind;        /* lmc_data_kind */
  float sigma_f;            /* multiplies the L2 term: 1/sigma^2 at prox_lmc_deconv.py:101 */
  const float* y_dev;       /* [H][W] observation (borrowed; must outlive every use) */
  const float* mask_dev;    /* [H][W], LMC_DATA_MASK only */
  int32_t kh, kw, oy, ox;   /* LMC_DATA_BLUR: kernel size and origin (pylops `offset`) */
  const float* h_host;      /* kh*kw taps, row-major, host memory */
  /* prior */
  int32_t prior_kind;       /* lmc_prior_kind */
  float prior_sigma;        /* multiplicative coefficient of g (tau_reg = 0.3 at prox_lmc_deconv.py:116-122) */
  int32_t tv_niter;         /* LMC_PRIOR_TV_ISO: number of dual iterations (niter_tv = 10) */
  float tv_step;            /* dual step is tv_step / (prox parameter); 0 -> 1/8 */
  const float* tv_betas_host; /* tv_niter momentum coefficients, host; NULL -> UNLocBoX/pyproximal sequence */
  /* non-log-concave data term of algs.L2_ncvx_tv (algs.py:22-291): f(x) = sigma_f/2||Hx-y||^2 - ncvx_lambda * env_gamma(TV)(x).
   * LMC_NCVX_MC_TV: Op2 = gradient, isotropic (algs.py:273-277): grad f = sigma_f H^T(Hx-y) - lambda * A^T( A x / max(|A x|, gamma) ) */
  int32_t ncvx_kind;        /* lmc_ncvx_kind */
  float ncvx_lambda;        /* lamda (= tau_reg at prox_lmc_deconv.py:106) */
  float ncvx_gamma;         /* gamma (= gamma_mc = gamma_me = 15 at prox_lmc_deconv.py:40,106,111) */
  int32_t ncvx_niter;       /* LMC_NCVX_ME_TV: dual iterations of the inner TV prox */
  /* ---- ABI 2 ---- */
  /* Which truncated iterate the TV prox returns.  pyproximal.TV.prox forms `sol = x - gamma div(r, s)` at the TOP of each loop pass and
   * returns the `sol` of the pass it leaves in; whether a run of `niter` reflects niter or niter - 1 dual updates depends on the loop
   * bound of the (un-vendored, un-pinned) upstream version -- it cannot be settled in this image.
   * 0 (default): after tv_niter dual updates; 1: after tv_niter - 1 ("lagged": one pipeline stage fewer).  Applies to the
   * LMC_PRIOR_TV_ISO prox and to the inner prox of the ME-TV term. */
  int32_t tv_lagged_output;
  /* pyproximal.TV's per-image early exit on the relative change of the primal objective (its default rtol = 1e-4, which the reference's
   * call at prox_lmc_deconv.py:122 does not override): at the top of loop pass j >= 1 the iterate x - gamma div(r_j) is returned when
   * |obj_j - obj_{j-1}| / obj_j < rtol.  0 (default): off -- every image runs tv_niter dual iterations in one fused launch.  > 0: every image
   * (chain) leaves in the pass the reference's loop leaves it in.  Two implementations, chosen by tv_exit_path:
   *  - on the device, without synchronisation (ABI 3; the default where the full-width pipeline covers the problem: 128 < W <= 512,
   *    W % 4 == 0 up to 256 columns and W % 8 == 0 above, tv_niter <= 60): the fused launch runs every chain with a PREDICTED pass count
   *    (the pass it left in at the previous call), the stages past it hand the dual through, and the primal objectives of all the iterates
   *    formed are by-products; a one-thread-per-chain kernel replays the exit test on them and the chains whose prediction was wrong
   *    run again (at most three more rounds settle every chain; the launches of settled chains return at once).
   *  - pass by pass (ABI 2; everything else): one launch per loop pass for the iterate, one for its objective; the host reads the number
   *    of images still iterating after every pass, so this path SYNCHRONISES the stream.
   * Not with tv_warm or MYMALA. */
  float tv_rtol;
  /* Step-kernel variant for launches configured from this problem: 0 = the library default (lmc_set_step_variant, itself "auto"
   * unless changed), 1..7 as listed at lmc_set_step_variant. */
  int32_t step_variant;
  /* Relative residual |r| <= tol |b| of the implicit data step (lmc_l2_prox, ULPDA): 0 = the library default
   * (lmc_set_cg_tolerance, itself 1e-6 unless changed), > 0 explicit, < 0 disabled (always all iterations, CG). */
  float implicit_tol;
  /* MYULA samplers only (build extension; NOT the reference's algorithm, whose pyproximal.TV.prox starts every call from a zero dual):
   * != 0 carries the projected TV dual (p, q) from one MYULA iteration to the next -- tv_niter in {1, 2, 3} dual iterations per
   * MYULA iteration, momentum restarted, +16 B per pixel and iteration of HBM traffic for the dual field.  SURVEY section 8(d) "K in
   * {1,3} warm-dual reported too".  lmc_sampler_set_state resets the dual to zero.  Needs the full-width pipeline kernel
   * (W > 128, W % 4 == 0 up to 256 columns and W % 8 == 0 above; separable blur / pointwise / no data term), otherwise LMC_E_UNSUPPORTED. */
  int32_t tv_warm;
  /* ---- ABI 3 ---- */
  /* The same early exit for the inner prox of the ME-TV term: algs.L2_ncvx_tv builds it as TV(dims, 1., niter, rtol) with the class's own
   * default rtol = 1e-4 (algs.py:130,169), used by value, gradient and prox of models M3 / M6 / M9 (prox_lmc_deconv.py:111-113).
   * 0: fixed ncvx_niter updates.  > 0: the device path above over the chained launches (ncvx_niter <= 60, same widths); elsewhere
   * LMC_E_UNSUPPORTED. */
  float ncvx_rtol;
  int32_t tv_exit_path;           /* tv_rtol > 0: 0 = the device path where it covers the problem, 1 = always pass by pass (synchronises) */
  /* Launch policy of samplers created from this problem (0 = the library decides).  Each field has an environment variable that supplies the
   * process-wide default when the field is 0; both are read ONCE, when the sampler is created -- never inside lmc_sampler_step.
   *  iterations_per_launch  (LMC_ITERS_PER_LAUNCH, and the older LMC_ROWS_PAIR / LMC_BLOCK_PAIR / LMC_CHEB_PAIR): 0 = several iterations per
   *      launch where a kernel covers the configuration and the launch is large enough to pay (two MYULA iterations: separable 5 x 5 box +
   *      closed-form prior, mask + Haar; two Chebyshev iterations of the implicit step); 1 = always one; 2 = wherever covered (tests).
   *  moments_overlap  (LMC_MOMENTS_OVERLAP): posterior-moment reductions on a side stream under the next step kernel: 0 = on (default), 1 = on, -1 = off.
   *  moments_bg_workgroups  (LMC_MOMENTS_BG_WGS): workgroups of that background reduction, 0 = by size.
   *  graph_replay  (LMC_GRAPH): 1 = replay captured hipGraphs of 8 iterations. */
  int32_t iterations_per_launch;
  int32_t moments_overlap;
  int32_t moments_bg_workgroups;
  int32_t graph_replay;
  /* LMC_PRIOR_EPROX */
  int32_t eprox_kind;             /* lmc_eprox_kind */
  float eprox_p0, eprox_p1;
  int32_t eprox_scale_mask;
  /* Array-valued epsg (algs.py:509,539-542: "float or np.ndarray"; the prox parameter at algs.py:569 is then the array epsg * gamma, which the
   * closed-form proxes broadcast -- one weight per right-hand side, i.e. per chain here, or per pixel).  NULL = scalar epsg only.  Device array;
   * the prox parameter of chain c, pixel i is  epsg * gamma * prox_scale[c * prox_scale_chain_stride + i * prox_scale_pixel_stride]
   * (strides in elements: (0, 1) per pixel, (1, 0) per chain, (H*W, 1) both).  MYULA with LMC_PRIOR_L2 / L1 / EPROX only (the prox is evaluated by
   * one launch before the fused step, which consumes it); LMC_E_UNSUPPORTED for the other priors and samplers.  Must outlive the sampler. */
  const float* prox_scale;
  int64_t prox_scale_chain_stride;
  int32_t prox_scale_pixel_stride;
} lmc_problem;

/* ---- library ------------------------------------------------------------------------- */
int lmc_version(void);                 /* LMC_ATOMI_ABI_VERSION of the loaded library */
const char* lmc_last_error(void);      /* thread-local; valid until the next failing call on this thread */
int lmc_device_info(int* device, int* n_cu, size_t* lds_bytes, size_t* hbm_bytes);
/* Measured HBM streaming bandwidth of the current device: a state-shaped copy (read `bytes`, write `bytes`, 16 B per lane), the best of
 * `reps` timed passes after one warm-up over ten launch shapes, in GB/s counting both directions -- the figure bench.py prints beside the 8 TB/s spec peak
 * (SURVEY section 8(d)).  Allocates and frees 2 x bytes of HBM; synchronises `stream`. */
int lmc_hbm_copy_probe(size_t bytes, int32_t reps, float* gbs_out, void* stream);

/* ---- operator-level entry points (the prox / linear-operator plugin protocol) -------- */

/* out = H x (adjoint == 0) or H^T x (adjoint != 0) for n_img images.
 * Replaces Convolve2D.matvec / .rmatvec (prox_lmc_deconv.py:58; algs.py:284). */
int lmc_blur(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W,
             const float* h_host, int32_t kh, int32_t kw, int32_t oy, int32_t ox, int32_t adjoint, void* stream);

/* out[2][H][W] = forward-difference gradient of x (zero in last row/col); pylops.Gradient.matvec
 * (prox_lmc_deconv.py:98; algs.py:436,448). */
int lmc_gradient(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, void* stream);
/* out[H][W] = Gradient.rmatvec(y[2][H][W]) = -div (algs.py:437,443). */
int lmc_gradient_adjoint(const float* y_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, void* stream);

/* out = a*x - t*grad f(x) + b*prox_{pt * g}(x)  for n_img images, one fused launch.
 *   (a,t,b) = (0,-1,0): out = grad f(x)          -> proxf.grad        (algs.py:569; 283-284)
 *   (a,t,b) = (0, 0,1): out = prox_{pt*g}(x)     -> proxg.prox(x, pt) (algs.py:569)
 * `pt` is the prox parameter (epsg*gamma in MYULA). */
int lmc_fused_eval(const lmc_problem* prob, const float* x_dev, float* out_dev, int64_t n_img,
                   float a, float t, float b, float pt, void* stream);

/* Implicit data step  out = prox_{tau f}(x) = (I + tau*sigma_f*Op^T Op)^{-1} (x + tau*sigma_f*Op^T y)  for n_img images:
 * pyproximal.L2.prox / algs.py:224-256 (row a8).  LMC_DATA_BLUR: iterative, started from `out` as given if warm != 0, else from zero,
 * to the relative residual of lmc_set_cg_tolerance within at most `niter` iterations (the reference: LSQR, niter=50, warm start).
 * Build-specified solver: the Chebyshev semi-iteration on the known spectrum [1, 1 + tau*sigma_f*(sum|h|)^2] where the row-streaming
 * kernel covers the blur (separable, <= 7 centred taps, any width) and the tolerance is reachable within `niter` (ULPDA samplers run two
 * iterations per launch where that pays: uniform 5 x 5 box, W % 4 == 0, W <= 512, n_chains * H >= 2^17);
 * conjugate gradients otherwise (then a cap that binds returns CG's truncated iterate).
 * IDENTITY / MASK / NONE: closed form.  workspace_dev: 5*n_img*H*W floats followed (8-byte aligned) by 4*n_img+1 doubles;
 * lmc_l2_prox_workspace_bytes gives the size. */
size_t lmc_l2_prox_workspace_bytes(int64_t n_img, int32_t H, int32_t W);
int lmc_l2_prox(const lmc_problem* prob, const float* x_dev, float* out_dev, int64_t n_img, float tau, int32_t niter,
                int32_t warm, void* workspace_dev, void* stream);

/* f_out[i] = f(x_i), g_out[i] = g(x_i) (double, device, n_img each; either may be NULL).
 * Replaces proxf(x), proxg(x) of the energy log (algs.py:461-466, 578-582). */
int lmc_energies(const lmc_problem* prob, const float* x_dev, int64_t n_img,
                 double* f_out_dev, double* g_out_dev, void* stream);

/* out = prox_{thr * ||W_detail . ||_1}(x): 3-level orthonormal Haar, soft threshold of the detail coefficients.  H, W % 8 == 0. */
int lmc_haar_l1_prox(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, float thr, void* stream);

/* Chain probes for convergence diagnostics across chains (split R-hat / effective sample size; SURVEY 8(f).3 -- absent in the
 * reference, whose only diagnostics are the per-iterate scalars of prox_lmc_deconv.py:128-133): every image is reduced to a
 * ph x pw grid of block means, out[i][a][b] = mean of x_i over rows [a*H/ph, (a+1)*H/ph) x columns [b*W/pw, (b+1)*W/pw)
 * (integer division).  1 <= ph <= H, 1 <= pw <= W, W <= 8192.  out_dev: n_img*ph*pw floats. */
int lmc_chain_probes(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, int32_t ph, int32_t pw, void* stream);

/* Per-pixel projection of the stacked field y[2][H][W] onto the l2 ball (isotropic != 0) or the
 * box (isotropic == 0) of radius `radius`: L21.proxdual / L1.proxdual (algs.py:436,448). */
int lmc_dual_project(const float* y_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W,
                     float radius, int32_t isotropic, void* stream);

/* Closed-form elementwise proxes of prox.py (the functional plugin surface), n elements. */
typedef enum lmc_eprox_kind {
  LMC_EPROX_LAPLACE = 0,            /* prox.py:18  params: gamma                */
  LMC_EPROX_UNCENTERED_LAPLACE = 1, /* prox.py:22  params: gamma, mu            */
  LMC_EPROX_GAUSSIAN = 2,           /* prox.py:26  params: gamma                */
  LMC_EPROX_GEN_GAUSSIAN_4_3 = 3,   /* prox.py:31  params: gamma                */
  LMC_EPROX_GEN_GAUSSIAN_3_2 = 4,   /* prox.py:34                               */
  LMC_EPROX_GEN_GAUSSIAN_3 = 5,     /* prox.py:36                               */
  LMC_EPROX_GEN_GAUSSIAN_4 = 6,     /* prox.py:38                               */
  LMC_EPROX_HUBER = 7,              /* prox.py:44  params: gamma, tau           */
  LMC_EPROX_SMOOTHED_LAPLACE = 8,   /* prox.py:52  params: gamma                */
  LMC_EPROX_EXP = 9,                /* prox.py:56  params: gamma                */
  LMC_EPROX_GAMMA = 10,             /* prox.py:60  params: omega, kappa         */
  LMC_EPROX_CHI = 11,               /* prox.py:64  params: kappa                */
  LMC_EPROX_UNIFORM = 12,           /* prox.py:68  params: omega                */
  LMC_EPROX_TRIANGULAR = 13,        /* prox.py:78  params: omega1, omega2       */
  LMC_EPROX_LAPLACE_CONJ = 14       /* prox.py:9 applied to prox_laplace; params: gamma */
} lmc_eprox_kind;
int lmc_prox_elementwise(int32_t kind, const float* x_dev, float* out_dev, int64_t n,
                         const float* params_host, int32_t n_params, void* stream);

/* ---- MYULA sampler (replaces algs.MoreauYosidaUnadjustedLangevin, algs.py:477-587) --- */

typedef struct lmc_sampler lmc_sampler; /* opaque */

typedef struct lmc_myula_config {
  uint32_t struct_size;     /* = sizeof(lmc_myula_config) */
  lmc_problem problem;
  int32_t n_chains;         /* chains resident on this GPU */
  int64_t chain_offset;     /* global id of local chain 0 (keys the RNG; sharding-invariant noise) */
  float tau, gamma, epsg;   /* step, Moreau smoothing, scaling of g (algs.py:477-478) */
  uint64_t seed;
  int32_t noise_mode;       /* lmc_noise_mode */
  /* posterior moments: sum / sum of squares over chains and kept iterations (generalises the
   * mean over iterates at prox_lmc_deconv.py:474 to many chains, burn-in and thinning) */
  int32_t moments;          /* 0 = off */
  int32_t burn_in;          /* iterations (0-based index < burn_in) excluded */
  int32_t thin;             /* keep every thin-th iteration after burn-in (>= 1) */
} lmc_myula_config;

int lmc_myula_create(const lmc_myula_config* cfg, lmc_sampler** out);
void lmc_sampler_destroy(lmc_sampler* s);

/* MYMALA -- the Metropolis-adjusted MYULA of prox_lmc.py:134-158 at image scale, every chain at once, same configuration
 * struct.  Per iteration and chain: x' = m(x) + sqrt(2 tau) xi (the MYULA move, :150); accepted with probability
 * min(1, pi(x') q(x|x') / (pi(x) q(x'|x))), pi = exp(-f - epsg*g), q(.|b) = N(m(b), 2 tau I) (:139-143,151-154); u ~ U(0,1) from
 * Philox (ctr = (0, iteration, global chain id, 0x4C4D4302), key = seed).  A rejected chain keeps its state, which is
 * counted again by the moment accumulators (the reference's toy version drops rejected iterations from its output list).
 * All lmc_sampler_* calls apply. */
int lmc_mymala_create(const lmc_myula_config* cfg, lmc_sampler** out);
/* accepted_dev [n_chains] uint64: accepted proposals so far; last_log_alpha_dev [n_chains] f64 (nullable): log acceptance
 * ratio of the latest iteration.  Device buffers. */
int lmc_sampler_get_acceptance(lmc_sampler* s, uint64_t* accepted_dev, double* last_log_alpha_dev, void* stream);

/* x_dev: [n_chains][H][W].  x0 of algs.py:559 (copied). */
int lmc_sampler_set_state(lmc_sampler* s, const float* x_dev, void* stream);
int lmc_sampler_get_state(lmc_sampler* s, float* x_dev, void* stream);
/* Run n_iters iterations of algs.py:564-570 on every chain.  noise_dev is
 * [n_iters][n_chains][H][W] when noise_mode == LMC_NOISE_INJECTED, else NULL.
 * All work is enqueued on `stream`.  The posterior-moment reductions of all but the last iteration of a call run on an internal side stream
 * under the following step kernel (ABI 3: at every size; lmc_problem.moments_overlap = -1 / LMC_MOMENTS_OVERLAP=0 keeps them in line, and so
 * does lmc_sampler_enable_timing); the call still returns with everything it enqueued ordered before whatever the caller enqueues on
 * `stream` next. */
int lmc_sampler_step(lmc_sampler* s, int32_t n_iters, const float* noise_dev, void* stream);
/* iteration counter (number of completed iterations since creation / last set_iteration) */
int64_t lmc_sampler_iteration(const lmc_sampler* s);
int lmc_sampler_set_iteration(lmc_sampler* s, int64_t it);

/* sum_dev, sumsq_dev: [H][W] double (device).  count = samples accumulated (chains*kept its). */
int lmc_sampler_get_moments(lmc_sampler* s, double* sum_dev, double* sumsq_dev, uint64_t* count, void* stream);
int lmc_sampler_reset_moments(lmc_sampler* s, void* stream);
/* per-chain energies f(x_c), g(x_c) of the current state (device double [n_chains]) */
int lmc_sampler_energies(lmc_sampler* s, double* f_out_dev, double* g_out_dev, void* stream);
/* the noise field xi[n_chains][H][W] the sampler draws at `iteration` (parity rung R3) */
int lmc_sampler_noise(lmc_sampler* s, int64_t iteration, float* out_dev, void* stream);
/* HIP-event timing of the step kernels: when enabled, lmc_sampler_step brackets EACH step-kernel
 * launch with its own event pair on the launch stream (interleaved moment reductions stay outside, and run in line -- not on the side
 * stream -- so that the step kernel is timed alone).
 * last_step_timing returns the summed kernel milliseconds and the number of launches of the LAST
 * lmc_sampler_step call (bench.py's roofline leg); it synchronises on the last event. */
int lmc_sampler_enable_timing(lmc_sampler* s, int32_t on);
int lmc_sampler_last_step_timing(lmc_sampler* s, float* total_ms, int32_t* n_launches);
/* name of the step kernel variant selected for this configuration (for profiles/) */
const char* lmc_sampler_kernel_name(const lmc_sampler* s);
/* Early-exit statistics of the latest TV prox evaluated with tv_rtol > 0 (which = 0) or of the ME-TV inner prox with ncvx_rtol > 0 (which = 1)
 * on the device path: passes_dev [n_chains] int32 (device, nullable) = the loop pass each chain left in (tv_niter: it ran out of passes);
 * reruns_host[4] (host, nullable) = chains whose run had to be repeated after round 1 / 2 / 3 / 4, summed over all calls since creation
 * (round 4's count stays 0 by construction).  Synchronises `stream`.  LMC_E_STATE when the sampler does not use the device path. */
int lmc_sampler_tv_exit_stats(lmc_sampler* s, int32_t which, int32_t* passes_dev, uint64_t* reruns_host, void* stream);

/* ---- multi-GPU: the one collective of the path ----------------------------------------------
 * Chains are sharded over GPUs by global chain id (chain_offset), one process per GPU, no data-path collective.  The only
 * exchange is the sum of the posterior-moment accumulators over the ranks -- what generalises the `.mean(axis=0)` over iterates
 * of prox_lmc_deconv.py:474 to a multi-GPU job.
 *
 * lmc_allreduce_moments: packs this sampler's {sum x [H][W], sum x^2 [H][W], count} into one float64 buffer, runs ONE
 * ncclAllReduce(ncclSum) (RCCL over xGMI) on `stream`, writes the job-wide sums to sum_dev / sumsq_dev (device, [H][W] double,
 * nullable), synchronises `stream` and returns the job-wide count.  The sampler's own accumulators are left untouched.
 * rccl_comm: an ncclComm_t passed as void* -- from lmc_rccl_comm_create below or any communicator the host already has (e.g.
 * torch.distributed's ProcessGroupNCCL) -- with one rank per sampler; NULL = a job of one rank (plain copy).
 * librccl is dlopen'd on first use (LMC_RCCL_LIB overrides the search: librccl.so.1 already in the process, then the default
 * paths); without it these calls return LMC_E_UNSUPPORTED and everything else works. */
#define LMC_RCCL_UNIQUE_ID_BYTES 128
int lmc_rccl_available(void);
/* ncclGetUniqueId on one rank; the host distributes the 128 bytes to the others (MPI, a file, torch.distributed ...). */
int lmc_rccl_unique_id(void* id128_host);
/* ncclCommInitRank on the CURRENT device (one process per GPU).  Collective: every rank of the job calls it. */
int lmc_rccl_comm_create(void** comm_out, int32_t world, int32_t rank, const void* id128_host);
int lmc_rccl_comm_destroy(void* comm);
int lmc_allreduce_moments(lmc_sampler* s, void* rccl_comm, double* sum_dev, double* sumsq_dev, uint64_t* count, void* stream);

/* ---- ULPDA sampler (replaces algs.UnadjustedLangevinPrimalDual, algs.py:295-474) --------------------
 *   x    <- prox_{tau f}(x - tau (A^T y + z)) + sqrt(2 tau) xi      (algs.py:440/446)
 *   xhat <- x + theta (x - x_old)                                   (:441/447)
 *   y    <- prox_{mu g*}(y + mu A xhat)                             (:436/448)   order set by gfirst (:435)
 * A = forward-difference gradient (prox_lmc_deconv.py:98); g = problem.prior (TV_ISO -> L21, TV_ANISO -> L1);
 * f = problem data term; for LMC_DATA_BLUR the implicit step (I + tau*sigma_f*H^T H)^{-1} is solved per chain, warm-started, to the
 * tolerance of lmc_set_cg_tolerance within at most `cg_niter` iterations (Chebyshev semi-iteration or CG, see lmc_l2_prox; the
 * reference: <= 50 LSQR iterations, algs.py:247-256).
 * The handle is an lmc_sampler: set_state / get_state / step / energies / moments / noise / destroy apply. */
typedef struct lmc_ulpda_config {
  uint32_t struct_size;     /* = sizeof(lmc_ulpda_config) */
  lmc_problem problem;
  int32_t n_chains;
  int64_t chain_offset;
  float tau, mu, theta;     /* algs.py:295-296 (scalars; per-iteration arrays: lmc_sampler_set_steps between calls) */
  int32_t gfirst;           /* algs.py:296, 435 */
  int32_t cg_niter;         /* inner iterations of the implicit data step (niter=50 at prox_lmc_deconv.py:101) */
  int32_t warm;             /* warm-start the inner solver from the previous solve (warm=True at :101) */
  const float* z_dev;       /* optional additional vector z [H][W] (algs.py:438-439), NULL = none */
  uint64_t seed;
  int32_t noise_mode;
  int32_t moments, burn_in, thin;
} lmc_ulpda_config;

int lmc_ulpda_create(const lmc_ulpda_config* cfg, lmc_sampler** out);
/* dual variable y [n_chains][2][H][W] (y0 of algs.py:427; y_samples of :451) */
int lmc_sampler_set_dual(lmc_sampler* s, const float* y_dev, void* stream);
int lmc_sampler_get_dual(lmc_sampler* s, float* y_dev, void* stream);
/* change (tau, mu) for the following iterations (the reference accepts per-iteration arrays, algs.py:402-408) */
int lmc_sampler_set_steps(lmc_sampler* s, float tau, float mu);

/* Library-wide DEFAULT (used when lmc_problem.implicit_tol is 0; per-handle / per-call value: lmc_problem.implicit_tol) of the
 * relative residual |r| <= tol |b| at which the inner solver of the implicit data step (lmc_l2_prox, ULPDA) stops before
 * cg_niter / niter iterations -- the stopping rule of the reference's solver (scipy lsqr's btol, default 1e-6, at algs.py:250).
 * Decided on the device for the whole batch of chains (all must satisfy it); tol = 0 disables it (always CG, all iterations).
 * Default 1e-6.  Returns the previous value; a negative argument only queries. */
float lmc_set_cg_tolerance(float tol);

/* Library-wide DEFAULT of the step-kernel variant, used by launches whose lmc_problem.step_variant is 0 (per-handle / per-call
 * selection: lmc_problem.step_variant; this switch is process-global and not thread-safe -- A/B tests only): 0 = auto (default),
 * 1 = LDS-tiled, 2 = (removed in ABI 2: the one-group streaming pipeline; LMC_E_INVALID), 3 = the row pipeline split over two wave groups,
 * 4 = HBM-bound tiled kernel for closed-form priors, 5 = register-block kernel (stencil-free data term, prox local to
 * 8 x 8 blocks: Haar-l1 / l2 / l1 / none; H, W multiples of 8), 6 = barrier-free row streaming (separable blur + closed-form
 * prior; any width: column strips above 512 columns, dword-aligned 16-byte accesses when W % 4 != 0), 7 = stage-parallel full-width TV pipeline
 * (isotropic TV with 10, 20, ... 60 dual iterations, separable blur or no data term, W > 128; any width for these counts, other counts -- 2, 6, 8, 9,
 * warm-dual 1, 2, 3 -- need W % 4 == 0 up to 256 columns and W % 8 == 0 above: the default of the headline configuration).  Returns the previous setting (>= 0) or a
 * negative lmc_status.  All variants compute the same update; the switch exists for A/B tests and profiles. */
int lmc_set_step_variant(int32_t variant);

#ifdef __cplusplus
}
#endif
#endif /* LMC_ATOMI_H */
