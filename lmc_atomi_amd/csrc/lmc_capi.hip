// extern "C" boundary of liblmc_atomi.so (see include/lmc_atomi.h).  Argument checks happen
// HERE, on the host, before any kernel sees a pointer or a shape.
#include <dlfcn.h>
#include <rccl/rccl.h>   // types and enums only: the library itself is dlopen'd (liblmc_atomi loads without RCCL)

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "lmc_launch.h"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t e_ = (expr);                                                                    \
    if (e_ != hipSuccess) return fail(LMC_E_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)

inline hipStream_t S(void* s) { return reinterpret_cast<hipStream_t>(s); }

int fill_taps(lmc::BlurTaps& T, const float* h, int kh, int kw, int oy, int ox) {
  if (!h) return fail(LMC_E_INVALID, "blur kernel pointer is NULL");
  if (kh < 1 || kw < 1 || kh > lmc::kMaxBlur || kw > lmc::kMaxBlur)
    return fail(LMC_E_UNSUPPORTED, "blur kernel %dx%d outside 1..%d", kh, kw, lmc::kMaxBlur);
  if (oy < 0 || oy >= kh || ox < 0 || ox >= kw) return fail(LMC_E_INVALID, "blur offset (%d,%d) outside kernel", oy, ox);
  T.kh = kh; T.kw = kw; T.oy = oy; T.ox = ox;
  std::memset(T.h, 0, sizeof T.h);
  std::memcpy(T.h, h, sizeof(float) * kh * kw);
  return LMC_OK;
}

// default momentum table: t_k = (1 + sqrt(4 t_{k-1}^2))/2 (pyproximal.TV / UNLocBoX), beta_k = (t_{k-1}-1)/t_k
void default_betas(float* b, int n) {
  double t = 1.0;
  for (int k = 0; k < n; ++k) {
    const double tn = (1.0 + std::sqrt(4.0 * t * t)) / 2.0;
    b[k] = (float)((t - 1.0) / tn);
    t = tn;
  }
}

// Validated, self-contained copy of an lmc_problem.
struct Problem {
  int H = 0, W = 0;
  int data_kind = 0;
  float sigma_f = 0.f;
  const float* y = nullptr;
  const float* mask = nullptr;
  lmc::BlurTaps taps{};
  int prior_kind = 0;
  float prior_sigma = 0.f;
  int tv_niter = 0;
  float tv_step = 0.125f;
  float betas[lmc::kMaxTvIters] = {};
  int ncvx_kind = 0;
  float ncvx_lambda = 0.f, ncvx_gamma = 1.f;
  int ncvx_niter = 0;
  int ncvx_aniso = 0;       // ME-TV: the 1-D TV of the flattened image (LMC_NCVX_ME_TV_ANISO)
  int tv_warm = 0;
  float tv_rtol = 0.f;      // > 0: pyproximal.TV's per-image early exit (device path tv_prox_rt, or the pass-by-pass path tv_prox_rtol)
  float ncvx_rtol = 0.f;    // > 0: the same for the inner prox of the ME-TV term (device path only)
  int tv_exit_path = 0;     // 1: always pass by pass
  int iters_per_launch = 0, moments_overlap = 0, moments_bg_wgs = 0, graph_replay = 0;   // launch policy (0 = library decides; fixed at sampler creation)
  int cheb_pair = 1;        // two Chebyshev iterations per launch: 0 never, 1 where they pay, 2 wherever covered
  int eprox_kind = 0, eprox_mask = 0;   // LMC_PRIOR_EPROX: closed form, which parameters scale with the prox parameter
  float eprox_p0 = 0.f, eprox_p1 = 0.f;
  const float* prox_scale = nullptr;   // array-valued epsg: per-chain / per-pixel multiplier of the prox parameter (closed-form priors of MYULA only)
  int64_t prox_scale_cs = 0, prox_scale_ps = 0;
  int variant = 0;          // 0: the library default (g_variant)
  float implicit_tol = 0.f; // 0: the library default (g_cg_tol); < 0: disabled
};

// Buffers of the device-side early exit of a TV prox (tv_prox_rt): per chain the pass count of the current round (kc; -1 = settled), the pass it left
// in at the previous call (pred: the prediction of the next one), the primal objectives of the iterates [n][stride], and three counters of re-runs.
struct RtState {
  int* kc = nullptr;
  int* start = nullptr;
  int* pred = nullptr;
  double* obj = nullptr;
  unsigned long long* reruns = nullptr;     // [4]: chains that had to run again after rounds 1 .. 4 (the last stays 0 by construction)
  int stride = 0;
  size_t n = 0;
  hipError_t need(size_t n_img, int niter) {
    if (n_img <= n && niter + 1 <= stride) return hipSuccess;
    release();
    hipError_t e = hipMalloc(&kc, sizeof(int) * n_img);
    if (e == hipSuccess) e = hipMalloc(&start, sizeof(int) * n_img);
    if (e == hipSuccess) e = hipMalloc(&pred, sizeof(int) * n_img);
    if (e == hipSuccess) e = hipMalloc(&obj, sizeof(double) * n_img * (size_t)(niter + 1));
    if (e == hipSuccess) e = hipMalloc(&reruns, sizeof(unsigned long long) * 4);
    if (e == hipSuccess) e = hipMemset(pred, 0, sizeof(int) * n_img);           // 0 = no prediction yet: the first call runs every pass
    if (e == hipSuccess) e = hipMemset(reruns, 0, sizeof(unsigned long long) * 4);
    if (e == hipSuccess) { n = n_img; stride = niter + 1; }
    return e;
  }
  void release() {
    if (kc) (void)hipFree(kc);
    if (start) (void)hipFree(start);
    if (pred) (void)hipFree(pred);
    if (obj) (void)hipFree(obj);
    if (reruns) (void)hipFree(reruns);
    kc = start = pred = nullptr; obj = nullptr; reruns = nullptr; n = 0; stride = 0;
  }
};

// Device scratch for the stateless entry points (grown on demand, kept for the life of the process; calls are
// serialised by the caller, see lmc_atomi.h).  Samplers own their buffers instead.
struct Scratch {
  float* state[2] = {nullptr, nullptr};   // TV dual state ping-pong, [n][4][H][W] each
  float* extra = nullptr;                 // ME-TV inner prox, [n][H][W]
  float* prox = nullptr;                  // Haar-l1 prox / early-exit TV prox, [n][H][W]
  float* rtmp = nullptr;                  // early-exit TV prox: the iterate of the current pass, [n][H][W]
  size_t n_rtmp = 0;
  hipError_t need_rtmp(size_t n) {
    if (n <= n_rtmp) return hipSuccess;
    if (rtmp) (void)hipFree(rtmp);
    rtmp = nullptr; n_rtmp = 0;
    hipError_t e = hipMalloc(&rtmp, sizeof(float) * n);
    if (e == hipSuccess) n_rtmp = n;
    return e;
  }
  double* dbl = nullptr;                  // 2*n doubles
  RtState rt_tv, rt_me;                   // device-side early exit of the TV prior's prox / of the ME-TV inner prox
  size_t n_state = 0, n_extra = 0, n_dbl = 0, n_prox = 0;
  hipError_t need_prox(size_t n) {
    if (n <= n_prox) return hipSuccess;
    if (prox) (void)hipFree(prox);
    prox = nullptr; n_prox = 0;
    hipError_t e = hipMalloc(&prox, sizeof(float) * n);
    if (e == hipSuccess) n_prox = n;
    return e;
  }
  hipError_t need_state(size_t n) {
    if (n <= n_state) return hipSuccess;
    for (float*& p : state) { if (p) (void)hipFree(p); p = nullptr; }
    n_state = 0;
    hipError_t e = hipMalloc(&state[0], sizeof(float) * n);
    if (e == hipSuccess) e = hipMalloc(&state[1], sizeof(float) * n);
    if (e == hipSuccess) n_state = n;
    return e;
  }
  hipError_t need_extra(size_t n) {
    if (n <= n_extra) return hipSuccess;
    if (extra) (void)hipFree(extra);
    extra = nullptr; n_extra = 0;
    hipError_t e = hipMalloc(&extra, sizeof(float) * n);
    if (e == hipSuccess) n_extra = n;
    return e;
  }
  hipError_t need_dbl(size_t n) {
    if (n <= n_dbl) return hipSuccess;
    if (dbl) (void)hipFree(dbl);
    dbl = nullptr; n_dbl = 0;
    hipError_t e = hipMalloc(&dbl, sizeof(double) * n);
    if (e == hipSuccess) n_dbl = n;
    return e;
  }
  void release() {
    for (float*& p : state) { if (p) (void)hipFree(p); p = nullptr; }
    if (extra) (void)hipFree(extra);
    if (dbl) (void)hipFree(dbl);
    extra = nullptr; dbl = nullptr; n_state = n_extra = n_dbl = 0;
  }
};
// one per device: a process may drive several GPUs (one sampler handle each); the stateless entry points run on the current device
constexpr int kMaxDevices = 64;
Scratch g_scratch_dev[kMaxDevices];
Scratch& scratch_here() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return g_scratch_dev[(dev >= 0 && dev < kMaxDevices) ? dev : 0];
}
#define g_scratch (scratch_here())

// Makes `dev` the current device for the duration of a call on a handle that lives there, and restores the caller's device.
struct DeviceGuard {
  int prev = -1;
  bool switched = false;
  explicit DeviceGuard(int dev) {
    if (dev < 0) return;
    if (hipGetDevice(&prev) == hipSuccess && prev != dev) switched = hipSetDevice(dev) == hipSuccess;
  }
  ~DeviceGuard() { if (switched) (void)hipSetDevice(prev); }
};

int load_problem(const lmc_problem* p, Problem& q) {
  if (!p) return fail(LMC_E_INVALID, "lmc_problem is NULL");
  if (p->struct_size != sizeof(lmc_problem))
    return fail(LMC_E_INVALID, "lmc_problem.struct_size %u != %zu (ABI mismatch)", p->struct_size, sizeof(lmc_problem));
  if (p->H < 1 || p->W < 1 || (int64_t)p->H * p->W > (int64_t)1 << 30) return fail(LMC_E_INVALID, "bad image size %dx%d", p->H, p->W);
  q.H = p->H; q.W = p->W;
  q.data_kind = p->data_kind;
  q.sigma_f = p->sigma_f;
  q.y = p->y_dev;
  q.mask = p->mask_dev;
  switch (p->data_kind) {
    case LMC_DATA_NONE: break;
    case LMC_DATA_IDENTITY:
      if (!p->y_dev) return fail(LMC_E_INVALID, "data term needs y_dev");
      break;
    case LMC_DATA_MASK:
      if (!p->y_dev || !p->mask_dev) return fail(LMC_E_INVALID, "mask data term needs y_dev and mask_dev");
      break;
    case LMC_DATA_BLUR: {
      if (!p->y_dev) return fail(LMC_E_INVALID, "data term needs y_dev");
      int rc = fill_taps(q.taps, p->h_host, p->kh, p->kw, p->oy, p->ox);
      if (rc) return rc;
      break;
    }
    default: return fail(LMC_E_INVALID, "unknown data_kind %d", p->data_kind);
  }
  q.prior_kind = p->prior_kind;
  q.prior_sigma = p->prior_sigma;
  switch (p->prior_kind) {
    case LMC_PRIOR_NONE: case LMC_PRIOR_L2: case LMC_PRIOR_L1: case LMC_PRIOR_TV_ANISO: break;
    case LMC_PRIOR_EPROX:
      if (p->eprox_kind < 0 || p->eprox_kind > LMC_EPROX_LAPLACE_CONJ) return fail(LMC_E_INVALID, "unknown eprox_kind %d", p->eprox_kind);
      if (p->eprox_scale_mask < 0 || p->eprox_scale_mask > 3) return fail(LMC_E_INVALID, "eprox_scale_mask must be 0..3");
      if (!(p->eprox_p0 == p->eprox_p0) || !(p->eprox_p1 == p->eprox_p1)) return fail(LMC_E_INVALID, "eprox parameter is NaN");
      q.eprox_kind = p->eprox_kind; q.eprox_mask = p->eprox_scale_mask; q.eprox_p0 = p->eprox_p0; q.eprox_p1 = p->eprox_p1;
      break;
    case LMC_PRIOR_HAAR_L1:
      if ((p->H & 7) || (p->W & 7)) return fail(LMC_E_UNSUPPORTED, "the Haar-l1 prior needs H and W to be multiples of 8 (got %dx%d)", p->H, p->W);
      break;
    case LMC_PRIOR_TV_ISO:
      if (p->tv_niter < 1 || p->tv_niter > lmc::kMaxTvIters)
        return fail(LMC_E_UNSUPPORTED, "tv_niter %d outside 1..%d", p->tv_niter, lmc::kMaxTvIters);
      if (!(p->tv_rtol >= 0.f) || p->tv_rtol >= 1.f) return fail(LMC_E_INVALID, "tv_rtol must be in [0, 1)");
      q.tv_rtol = p->tv_rtol;
      q.tv_niter = p->tv_niter - (p->tv_lagged_output ? 1 : 0);     // lagged: the iterate after tv_niter - 1 dual updates (0: prox = x)
      q.tv_step = p->tv_step > 0.f ? p->tv_step : 0.125f;
      if (p->tv_betas_host) std::memcpy(q.betas, p->tv_betas_host, sizeof(float) * p->tv_niter);
      else default_betas(q.betas, p->tv_niter);
      break;
    default: return fail(LMC_E_INVALID, "unknown prior_kind %d", p->prior_kind);
  }
  if (p->prior_kind != LMC_PRIOR_NONE && p->prior_kind != LMC_PRIOR_EPROX && !(p->prior_sigma >= 0.f)) return fail(LMC_E_INVALID, "prior_sigma must be >= 0");
  if (p->ncvx_kind != LMC_NCVX_NONE) {
    if (p->ncvx_kind != LMC_NCVX_MC_TV && p->ncvx_kind != LMC_NCVX_ME_TV && p->ncvx_kind != LMC_NCVX_MC_TV_ANISO && p->ncvx_kind != LMC_NCVX_ME_TV_ANISO)
      return fail(LMC_E_INVALID, "unknown ncvx_kind %d", p->ncvx_kind);
    if (!(p->ncvx_gamma > 0.f)) return fail(LMC_E_INVALID, "ncvx_gamma must be > 0");
    const bool me = p->ncvx_kind == LMC_NCVX_ME_TV || p->ncvx_kind == LMC_NCVX_ME_TV_ANISO;
    if (me && (p->ncvx_niter < 1 || p->ncvx_niter > lmc::kMaxTvIters))
      return fail(LMC_E_INVALID, "ncvx_niter %d outside 1..%d", p->ncvx_niter, lmc::kMaxTvIters);
    q.ncvx_kind = p->ncvx_kind; q.ncvx_lambda = p->ncvx_lambda; q.ncvx_gamma = p->ncvx_gamma;
    // anisotropic ME-TV: inside the library the ME-TV kind with the 1-D inner prox (every code path that adds the term's gradient serves both)
    if (p->ncvx_kind == LMC_NCVX_ME_TV_ANISO) { q.ncvx_kind = LMC_NCVX_ME_TV; q.ncvx_aniso = 1; }
    // anisotropic MC-TV: inside the library the same kind with a NEGATIVE gamma -- mc_tv_grad (lmc_device.h) and the energy kernels take
    // the sign as "component-wise weights 1 / max(|d|, gamma)" instead of the pixel norm; every MC-TV code path serves both
    if (p->ncvx_kind == LMC_NCVX_MC_TV_ANISO) { q.ncvx_kind = LMC_NCVX_MC_TV; q.ncvx_gamma = -p->ncvx_gamma; }
    q.ncvx_niter = p->ncvx_niter - ((me && p->tv_lagged_output) ? 1 : 0);
    if (me) {
      if (!(p->ncvx_rtol >= 0.f) || p->ncvx_rtol >= 1.f) return fail(LMC_E_INVALID, "ncvx_rtol must be in [0, 1)");
      q.ncvx_rtol = p->ncvx_rtol;
    }
  }
  if (p->tv_exit_path != 0 && p->tv_exit_path != 1) return fail(LMC_E_INVALID, "tv_exit_path must be 0 (device path where covered) or 1 (pass by pass)");
  q.tv_exit_path = p->tv_exit_path;
  if (p->iterations_per_launch < 0 || p->iterations_per_launch > 2) return fail(LMC_E_INVALID, "iterations_per_launch must be 0 (auto), 1 or 2");
  if (p->moments_overlap < -1 || p->moments_overlap > 1) return fail(LMC_E_INVALID, "moments_overlap must be 0 (auto), 1 (on) or -1 (off)");
  if (p->moments_bg_workgroups < 0 || p->graph_replay < 0 || p->graph_replay > 1) return fail(LMC_E_INVALID, "bad moments_bg_workgroups / graph_replay");
  q.iters_per_launch = p->iterations_per_launch; q.moments_overlap = p->moments_overlap;
  q.moments_bg_wgs = p->moments_bg_workgroups; q.graph_replay = p->graph_replay;
  {
    const char* e = getenv("LMC_CHEB_PAIR");
    const char* e2 = getenv("LMC_ITERS_PER_LAUNCH");
    const int ipl = q.iters_per_launch ? q.iters_per_launch : (e2 ? atoi(e2) : 0);
    q.cheb_pair = ipl == 1 ? 0 : (ipl == 2 ? 2 : (e ? atoi(e) : 1));
  }
  if (p->step_variant < 0 || p->step_variant > 7 || p->step_variant == 2)
    return fail(LMC_E_INVALID, "step_variant %d: 0 (library default), 1 tile, 3 split, 4 point, 5 block, 6 rows, 7 pipe", p->step_variant);
  q.variant = p->step_variant;
  if (p->prox_scale) {
    if (p->prior_kind != LMC_PRIOR_L2 && p->prior_kind != LMC_PRIOR_L1 && p->prior_kind != LMC_PRIOR_EPROX)
      return fail(LMC_E_UNSUPPORTED, "prox_scale (array-valued epsg) is built for the closed-form priors (l2, l1, prox.py closed forms) only");
    if (p->prox_scale_chain_stride < 0 || p->prox_scale_pixel_stride < 0) return fail(LMC_E_INVALID, "prox_scale strides must be >= 0");
    q.prox_scale = p->prox_scale; q.prox_scale_cs = p->prox_scale_chain_stride; q.prox_scale_ps = p->prox_scale_pixel_stride;
  }
  q.tv_warm = (p->tv_warm != 0 && p->prior_kind == LMC_PRIOR_TV_ISO) ? 1 : 0;
  if (!(p->implicit_tol == p->implicit_tol)) return fail(LMC_E_INVALID, "implicit_tol is NaN");
  q.implicit_tol = p->implicit_tol;
  return LMC_OK;
}

// StepArgs for: out = a*x - t*grad f + b*prox_{pt*g}(x) + s*xi
int make_step_args(const Problem& q, float a, float t, float b, float pt, float s, lmc::StepArgs& A) {
  std::memset(&A, 0, sizeof A);
  A.H = q.H; A.W = q.W;
  A.data_kind = (t == 0.f) ? LMC_DATA_NONE : q.data_kind;   // skip the stencil work if its weight is zero
  A.sigma_f = q.sigma_f;
  A.y = q.y; A.mask = q.mask;
  A.blur = q.taps;
  A.prior_kind = (b == 0.f) ? LMC_PRIOR_NONE : q.prior_kind;
  if (A.prior_kind == LMC_PRIOR_TV_ISO && q.tv_niter == 0) A.prior_kind = LMC_PRIOR_NONE;   // lagged output of a 1-iteration prox: x itself
  if (A.prior_kind == LMC_PRIOR_HAAR_L1) A.prior_p0 = pt * q.prior_sigma;  // soft threshold of the detail coefficients
  if (A.prior_kind == LMC_PRIOR_L2) A.prior_p0 = 1.f / (1.f + pt * q.prior_sigma);
  if (A.prior_kind == LMC_PRIOR_L1) A.prior_p0 = pt * q.prior_sigma;
  if (A.prior_kind == LMC_PRIOR_EPROX) {     // prox.py closed forms: the parameters the mask names scale with the prox parameter
    A.eprox_kind = q.eprox_kind;
    A.prior_p0 = (q.eprox_mask & 1) ? pt * q.eprox_p0 : q.eprox_p0;
    A.prior_p1 = (q.eprox_mask & 2) ? pt * q.eprox_p1 : q.eprox_p1;
  }
  if (A.prior_kind == LMC_PRIOR_TV_ISO) {
    const float gam = pt * q.prior_sigma;
    if (!(gam > 0.f)) return fail(LMC_E_INVALID, "TV prox parameter must be > 0 (got %g)", (double)gam);
    A.tv.niter = q.tv_niter;
    A.tv.gamma = gam;
    A.tv.c = q.tv_step / gam;
    std::memcpy(A.tv.betas, q.betas, sizeof(float) * q.tv_niter);
  }
  if (t != 0.f && q.ncvx_kind == LMC_NCVX_MC_TV) {
    A.ncvx_kind = q.ncvx_kind; A.ncvx_lambda = q.ncvx_lambda; A.ncvx_gamma = q.ncvx_gamma; A.ncvx_inv_gamma = 1.f / q.ncvx_gamma;
  }
  // LMC_NCVX_ME_TV: the caller runs me_tv_prox first and sets A.extra / A.extra_coef
  A.a = a; A.t = t; A.b = b; A.s = s;
  A.noise_mode = LMC_NOISE_NONE;
  return LMC_OK;
}

// Pointers that the configuration does not use are pointed at the input state (always valid for the
// index ranges the kernels form) so that no kernel ever holds a null pointer it could dereference.
void sanitize_pointers(lmc::StepArgs& A) {
  if (!A.y) A.y = A.x_in;
  if (!A.mask) A.mask = A.x_in;
  if (!A.noise) A.noise = A.x_in;
}

// Library-wide DEFAULTS only (lmc_set_step_variant / lmc_set_cg_tolerance): every launch takes its variant and tolerance from the
// lmc_problem it was configured from (step_variant / implicit_tol) and falls back to these when that field is 0.
int g_variant = 0;  // 0 auto, 1 tile, (2: removed) 3 split, 4 point, 5 block, 6 rows, 7 pipe
float g_cg_tol = 1e-6f;   // relative residual at which the inner solver stops (0: always cg_niter iterations)
int variant_of(const Problem& q) { return q.variant ? q.variant : g_variant; }
float tol_of(const Problem& q) { return q.implicit_tol > 0.f ? q.implicit_tol : (q.implicit_tol < 0.f ? 0.f : g_cg_tol); }

// Picks the step-kernel variant.  auto: the split streaming pipeline (two wave groups, 4 waves/SIMD) when
// it covers the configuration (W <= 512, separable blur <= 7x7, supported K), else the LDS-tiled kernel.
hipError_t launch_step(const lmc::StepArgs& A_in, int variant, hipStream_t st, const char** name, float* state0 = nullptr,
                       float* state1 = nullptr, float* pxbuf = nullptr) {
  int v = variant;
  // no stencil in the data term and a prox local to 8 x 8 blocks (Haar-l1, l2, l1, none): the register-block kernel
  if ((v == 0 || v == 5) && lmc::block_supported(A_in)) {
    if (name) *name = "myula_step_block_kernel";
    return lmc::launch_step_block(A_in, st);
  }
  if ((v == 0 || v == 5) && A_in.ncvx_kind == LMC_NCVX_MC_TV) {
    // stencil-free data term + block-local prox + MC-TV term: the block kernel without the term, then one stencil
    // pass that adds t * lambda * A^T(A x / max(|A x|, gamma)) to its output
    lmc::StepArgs B = A_in;
    B.ncvx_kind = LMC_NCVX_NONE;
    if (lmc::block_supported(B)) {
      if (name) *name = "myula_step_block_kernel";
      hipError_t e = lmc::launch_step_block(B, st);
      if (e != hipSuccess) return e;
      return lmc::launch_mc_tv_add(A_in.x_in, A_in.x_out, A_in.C, A_in.H, A_in.W, A_in.t * A_in.ncvx_lambda, A_in.ncvx_gamma, st);
    }
  }
  if (v == 5) return hipErrorInvalidConfiguration;
  lmc::StepArgs A = A_in;
  if (A.prior_kind == LMC_PRIOR_HAAR_L1) {   // other data terms: the block-wavelet prox first, consumed by the fused step kernel
    if (!pxbuf) return hipErrorInvalidConfiguration;
    hipError_t e = lmc::launch_haar_prox(A.x_in, pxbuf, A.C, A.H, A.W, A.prior_p0, st);
    if (e != hipSuccess) return e;
    A.prior_kind = LMC_PRIOR_NONE;
    A.prox_ext = pxbuf;
  }
  // separable blur + closed-form prior (no TV pipeline): barrier-free row streaming, one wave per band of rows
  if ((v == 0 || v == 6) && lmc::rows_supported(A)) {
    if (name) *name = "myula_step_rows_kernel";
    return lmc::launch_step_rows(A, st);
  }
  if (v == 6) return hipErrorInvalidConfiguration;
  // TV K = 10 on a 264..512-wide image with a separable blur: the stage-parallel full-width pipeline
  if (v == 0 || v == 7) {
    const int links = lmc::pipe_links(A);
    if (links == 1 || (links > 1 && state0 && state1)) {
      if (name) *name = "myula_step_pipe_kernel";
      return lmc::launch_step_pipe(A, st, state0, state1);
    }
  }
  if (v == 7) return hipErrorInvalidConfiguration;
  // auto: split pipeline when it covers the configuration (W <= 512); for wider images the tiled kernels:
  // "point" for closed-form priors with a separable blur, else the general LDS-tiled kernel
  // a closed-form elementwise prior (LMC_PRIOR_EPROX) that reaches this point (a non-log-concave term, or a blur the row kernel does not cover): the split and
  // tiled kernels have no functor for it -- the point kernel evaluates it in place; where that does not cover the data term the prox is formed by one elementwise
  // launch and consumed as a ready-made prox.  (Round 3's configuration-matrix test found these combinations running with prox = identity.)
  if (A.prior_kind == LMC_PRIOR_EPROX) {
    if ((v == 0 || v == 4) && lmc::point_supported(A)) v = 4;
    else {
      if (!pxbuf) return hipErrorInvalidConfiguration;
      hipError_t e = lmc::launch_eprox(A.eprox_kind, A.x_in, pxbuf, (int64_t)A.C * A.H * A.W, A.prior_p0, A.prior_p1, st);
      if (e != hipSuccess) return e;
      A.prior_kind = LMC_PRIOR_NONE;
      A.prox_ext = pxbuf;
    }
  }
  if (v == 0) v = lmc::split_supported(A) ? 3 : (lmc::point_supported(A) ? 4 : 1);
  if (v == 4) {
    if (!lmc::point_supported(A)) return hipErrorInvalidConfiguration;
    if (name) *name = "myula_step_point_kernel";
    return lmc::launch_step_point(A, st);
  }
  if (v == 3) {
    if (!lmc::split_supported(A)) return hipErrorInvalidConfiguration;
    if (name) *name = "myula_step_split_kernel";
    return lmc::launch_step_split(A, st);
  }
  if (name) *name = "myula_step_tile_kernel";
  if (lmc::tile_needs_chunks(A)) {
    if (!state0 || !state1) return hipErrorInvalidConfiguration;
    return lmc::launch_step_tile_chunked(A, state0, state1, st);
  }
  return lmc::launch_step_tile(A, st);
}

// Conjugate gradients on (I + ts H^T H) u = rhs for every chain, `niter` iterations from the current u.
// The operator q = p + ts H^T H p is ONE launch of the fused step kernel (out = 1*p - t*grad f(p) with y = 0 and
// t = -ts/sigma_f), i.e. the same blur pipeline as the sampler; zero_y is an all-zero [H][W] image.
// scal: 4C + 1 doubles (rs, pq, rs_new, |rhs|^2 per chain, and the "converged" flag).
// Chebyshev semi-iteration for (I + ts H^T H) u = rhs.  The spectrum is known: H^T H lies in [0, (sum |h|)^2] (zero-padded
// convolution, Young's inequality), so A lies in [1, 1 + ts (sum |h|)^2] and the three-term recurrence (Saad, Iterative Methods,
// alg. 12.1)   u_{k+1} = u_k + alpha_k (rhs - A u_k) + beta_k (u_k - u_{k-1})   needs no inner products at all.  One iteration is ONE
// launch of the row-streaming step kernel:  out = a x - t sigma_f H^T H x + b ext + s prev  with x = u_k, ext = rhs, prev = u_{k-1} read
// through the injected-noise input and overwritten in place by u_{k+1} (pointwise read-then-write by the same lane): 16 B per pixel
// and iteration instead of the 44 B and six launches of a CG iteration.  The residual of the k-th iterate is max|p_k| |r_0| with
// max|p_k| <= 2 c^k, c = (sqrt(kappa) - 1) / (sqrt(kappa) + 1): the iteration count for the reference's stopping rule |r| <= tol |b|
// (scipy lsqr btol, algs.py:250) is known in advance -- no convergence test, no flags, no host synchronisation.
// Returns hipErrorInvalidConfiguration when the row-streaming kernel does not cover the problem (caller falls back to CG).
// pb (optional): two scratch arrays and the array that receives the solution, all [C][H][W] and distinct from u / tmp / rhs.  With them, and
// where lmc_cheb_pair.hip covers the problem, the iterations after the first run TWO per launch (20 instead of 32 B per pixel); the solution
// then arrives in pb->out, u (the starting guess) is used as scratch, and *result says which of the two holds it.
struct ChebPairBufs { float* b1; float* b2; float* out; };
static hipError_t chebyshev_solve(const Problem& q, float ts, float* u, const float* rhs, float* tmp, double* scal, int64_t C, int niter_cap,
                                  float tol, const float* zero_y, hipStream_t st, const ChebPairBufs* pb = nullptr, float** result = nullptr) {
  if (result) *result = u;
  lmc::StepArgs A;
  std::memset(&A, 0, sizeof A);
  A.H = q.H; A.W = q.W; A.C = (int)C;
  A.data_kind = LMC_DATA_BLUR; A.sigma_f = q.sigma_f; A.blur = q.taps;
  A.y = zero_y; A.mask = zero_y;
  A.prior_kind = LMC_PRIOR_NONE;
  A.prox_ext = rhs;
  A.noise = zero_y;
  A.noise_mode = LMC_NOISE_NONE;
  A.x_in = u; A.x_out = tmp;
  if (!lmc::rows_supported(A)) return hipErrorInvalidConfiguration;
  double hsum = 0.0;
  for (int i = 0; i < q.taps.kh * q.taps.kw; ++i) hsum += std::fabs((double)q.taps.h[i]);
  const double lmin = 1.0, lmax = 1.0 + (double)ts * hsum * hsum * 1.0001;       // a hair of slack for the fp32 taps
  const double theta = 0.5 * (lmax + lmin), delta = 0.5 * (lmax - lmin);
  int k_need = 1;
  if (delta > 1e-12 * theta) {
    const double sk = std::sqrt(lmax / lmin), c = (sk - 1.0) / (sk + 1.0);
    k_need = (int)std::ceil(std::log(2.0 / (double)tol) / std::log(1.0 / c)) + 1;   // +1: a warm start may begin with |r_0| > |b|
  }
  // A cap below what the tolerance needs makes the answer depend on the solver (a truncated iterate): leave that case to CG, whose
  // truncated iterates are the ones pinned by the tests; here every solve reaches the tolerance.
  if (k_need > niter_cap) return hipErrorInvalidConfiguration;
  static const int env_k = [] { const char* e = getenv("LMC_CHEB_K"); return e ? atoi(e) : 0; }();   // experiments: fixed count, no adaptation
  int K = env_k > 0 ? env_k : k_need;
  // A warm start begins with |r_0| << |rhs|: the first launch measures |r_0| and |rhs| on the fly, a one-block kernel turns them into the
  // number of launches needed (even, <= the a-priori count), and the launches beyond it return at their first instruction.
  // (the adaptive count is even; when rounding up would exceed the caller's cap the a-priori count runs as it is)
  const bool adaptive = env_k <= 0 && K > 2 && delta > 1e-12 * theta && ((K + 1) & ~1) <= niter_cap;
  if (adaptive) {
    K = (K + 1) & ~1;
    hipError_t e = hipMemsetAsync(scal, 0, sizeof(double) * (4 * C + 1), st);
    if (e != hipSuccess) return e;
  }
  // Chain chunks: the K launches of one chunk run back to back, so u_k, u_{k-1} and rhs of that chunk (3 arrays) are still in the 256 MB
  // memory-side cache when the next launch reads them.  LMC_CHEB_CHUNK = chains per chunk (0 / unset: one chunk).
  static const int64_t env_chunk = [] { const char* e = getenv("LMC_CHEB_CHUNK"); return e ? (int64_t)atoll(e) : 0; }();
  const int64_t chunk = env_chunk > 0 && env_chunk < C ? env_chunk : C;
  const size_t img = (size_t)q.H * q.W;
  // lmc_problem.iterations_per_launch / LMC_CHEB_PAIR (resolved when the problem is loaded: sampler creation, or the stateless call): 0 = single-iteration
  // launches only, 2 = pairs wherever the kernel covers the problem (tests), default = where they pay
  const int pair_mode = q.cheb_pair;
  const bool pair_on = pair_mode == 2 || (pair_mode == 1 && lmc::cheb_pair_pays(C, q.H));
  if (pb && result && pair_on && chunk == C && K >= 4 && 2 * ((K + 1) / 2) <= niter_cap && delta > 1e-12 * theta &&
      lmc::cheb_pair_supported(q.H, q.W, q.taps)) {      // (pairs never run more iterations than the caller's cap)
    // pairs p = 1 .. M of iterations 2p - 2, 2p - 1.  The first one also forms the residual statistics of iteration 0 (the adaptive count, known
    // after it); the second always runs (the solution has to arrive in pb->out, and the first cannot know whether it is the last); pair p >= 3
    // returns at once when count <= 2p - 2; the last pair that runs writes to pb->out.
    const int M = (K + 1) / 2;
    const int n_it = 2 * M;
    std::vector<double> al(n_it), be(n_it);
    {
      double rho = delta / theta;
      const double sigma1 = theta / delta;
      al[0] = 1.0 / theta; be[0] = 0.0;
      for (int k = 1; k < n_it; ++k) {
        const double rho_new = 1.0 / (2.0 * sigma1 - rho);
        al[k] = 2.0 * rho_new / delta;
        be[k] = rho_new * rho;
        rho = rho_new;
      }
    }
    double* stat = scal;
    int* count = reinterpret_cast<int*>(scal + 4 * C);
    const float* cur = u;
    const float* prv = u;          // iteration 0 has no u_{-1} (s0 = 0): any valid array
    float* f1 = tmp;
    float* f2 = pb->b1;
    float* spare = pb->b2;
    for (int p = 1; p <= M; ++p) {
      const int k0 = 2 * p - 2, k1 = 2 * p - 1;
      lmc::ChebPairArgs P;
      std::memset(&P, 0, sizeof P);
      P.H = q.H; P.W = q.W; P.C = (int)C;
      P.cur = cur; P.prv = prv; P.rhs = rhs; P.f1 = f1; P.f2 = f2; P.f2_last = p == 1 ? f2 : pb->out;
      P.a0 = (float)(1.0 - al[k0] + be[k0]); P.tg0 = (float)(al[k0] * (double)ts); P.b0 = (float)al[k0]; P.s0 = (float)(-be[k0]);
      P.a1 = (float)(1.0 - al[k1] + be[k1]); P.tg1 = (float)(al[k1] * (double)ts); P.b1 = (float)al[k1]; P.s1 = (float)(-be[k1]);
      P.run_count = adaptive && p >= 2 ? count : nullptr;
      P.run_index = p <= 2 ? -1 : k0;           // *count <= k0: iterations k0, k0 + 1 are not needed
      P.last_index = k1 + 1;                    // no later pair runs when *count <= 2p
      P.force_last = p == M;
      P.dot_out = adaptive && p == 1 ? stat : nullptr;
      hipError_t e = lmc::launch_cheb_pair(P, q.taps, st);
      if (e != hipSuccess) return e;
      if (adaptive && p == 1) {
        const double sk = std::sqrt(lmax / lmin), c = (sk - 1.0) / (sk + 1.0);
        e = lmc::cheb_count(C, stat, 1.0 / ((double)P.b0 * (double)P.b0), (double)tol, 1.0 / std::log(1.0 / c), 2 * M, count, st);
        if (e != hipSuccess) return e;
      }
      // u_{k+2} = f2 and u_{k+1} = f1 are the next pair's inputs; the arrays it read are free again (u itself from the second pair on)
      float* free_a = p == 1 ? spare : const_cast<float*>(cur);
      float* free_b = p == 1 ? u : const_cast<float*>(prv);
      cur = f2; prv = f1; f1 = free_a; f2 = free_b;
    }
    *result = pb->out;
    return hipSuccess;
  }
  int ci = 0;
  for (int64_t c0 = 0; c0 < C; c0 += chunk, ++ci) {
    const int64_t nc = C - c0 < chunk ? C - c0 : chunk;
    if (ci >= 2 * C) return hipErrorInvalidConfiguration;
    double* stat = scal + 2 * c0;                            // [2 nc] of the [2C] block
    int* count = reinterpret_cast<int*>(scal + 2 * C) + ci;  // the solver's flag word of this chunk (the second [2C] block is free here)
    A.C = (int)nc;
    A.prox_ext = rhs + c0 * img;
    float* cur = u + c0 * img;
    float* oth = tmp + c0 * img;
    double rho = delta > 0 ? delta / theta : 0.0;      // rho_0 = 1 / sigma_1
    const double sigma1 = delta > 0 ? theta / delta : 0.0;
    for (int k = 0; k < K; ++k) {
      double alpha, beta;
      if (k == 0 || !(delta > 1e-12 * theta)) { alpha = 1.0 / theta; beta = 0.0; }
      else {
        const double rho_new = 1.0 / (2.0 * sigma1 - rho);
        alpha = 2.0 * rho_new / delta;
        beta = rho_new * rho;
        rho = rho_new;
      }
      A.x_in = cur; A.x_out = oth;
      A.a = (float)(1.0 - alpha + beta);
      A.t = (float)(alpha * (double)ts / (double)q.sigma_f);
      A.b = (float)alpha;
      if (beta != 0.0) { A.noise_mode = LMC_NOISE_INJECTED; A.noise = oth; A.s = (float)(-beta); }   // oth holds u_{k-1} and receives u_{k+1}
      else { A.noise_mode = LMC_NOISE_NONE; A.noise = zero_y; A.s = 0.f; }
      A.dot_out = nullptr; A.dot_mode = 0; A.run_count = nullptr; A.run_index = 0;
      if (adaptive) {
        if (k == 0) { A.dot_out = stat; A.dot_mode = 1; }
        else { A.run_count = count; A.run_index = k; }
      }
      hipError_t e = lmc::launch_step_rows(A, st);
      if (e != hipSuccess) return e;
      if (adaptive && k == 0) {
        const double sk = std::sqrt(lmax / lmin), c = (sk - 1.0) / (sk + 1.0);
        e = lmc::cheb_count(nc, stat, 1.0 / ((double)A.b * (double)A.b), (double)tol, 1.0 / std::log(1.0 / c), K, count, st);
        if (e != hipSuccess) return e;
      }
      float* t = cur; cur = oth; oth = t;
    }
    if (cur != u + c0 * img) {
      hipError_t e = hipMemcpyAsync(u + c0 * img, cur, sizeof(float) * (size_t)nc * img, hipMemcpyDeviceToDevice, st);
      if (e != hipSuccess) return e;
    }
  }
  return hipSuccess;
}

// alt_out / result (optional, both or none): an extra [C][H][W] array the solution may arrive in instead of u (*result tells); u is then scratch.
int cg_solve_fused(const Problem& q, float ts, float* u, const float* rhs, float* r, float* p, float* qq, double* scal,
                   int64_t C, int niter, const float* zero_y, hipStream_t st, float* alt_out = nullptr, float** result = nullptr) {
  if (result) *result = u;
  const size_t img = (size_t)q.H * q.W;
  // LMC_IMPLICIT_SOLVER=cg keeps the conjugate-gradient path below (A/B runs); default: Chebyshev whenever a tolerance is set
  static const bool want_cheb = [] { const char* e = getenv("LMC_IMPLICIT_SOLVER"); return !(e && std::strcmp(e, "cg") == 0); }();
  const float cg_tol = tol_of(q);
  if (want_cheb && cg_tol > 0.f) {
    const ChebPairBufs pb{p, qq, alt_out};
    hipError_t e = chebyshev_solve(q, ts, u, rhs, r, scal, C, niter, cg_tol, zero_y, st, alt_out && result ? &pb : nullptr, result);
    if (e == hipSuccess) return LMC_OK;
    if (e != hipErrorInvalidConfiguration) HIP_TRY(e);
  }
  double *rs = scal, *pq = scal + C, *rs_new = scal + 2 * C, *b2 = scal + 3 * C;
  int* done = reinterpret_cast<int*>(scal + 4 * C);
  const bool early = cg_tol > 0.f;
  const double tol2 = (double)cg_tol * (double)cg_tol;
  lmc::StepArgs A;
  std::memset(&A, 0, sizeof A);
  A.H = q.H; A.W = q.W; A.C = (int)C;
  A.data_kind = LMC_DATA_BLUR; A.sigma_f = q.sigma_f; A.blur = q.taps;
  A.y = zero_y; A.mask = zero_y; A.noise = zero_y;
  A.prior_kind = LMC_PRIOR_NONE;
  A.a = 1.f; A.t = -ts / q.sigma_f; A.b = 0.f; A.s = 0.f;
  A.noise_mode = LMC_NOISE_NONE;
  const char* kname = nullptr;
  auto apply = [&](const float* in, float* out) -> hipError_t { A.x_in = in; A.x_out = out; return launch_step(A, variant_of(q), st, &kname); };
  HIP_TRY(hipMemsetAsync(scal, 0, sizeof(double) * (4 * C + 1), st));
  HIP_TRY(apply(u, qq));
  A.dot_out = pq;            // the row-streaming kernel accumulates p.Ap while it writes Ap (one pass less per iteration)
  if (early) A.skip_flag = done;
  HIP_TRY(lmc::cg_init(rhs, qq, r, p, C, img, rs, b2, st));
  // Stopping rule = the reference's: its solver (scipy lsqr, algs.py:250) ends at |r| <= btol |b| with btol = 1e-6 by default,
  // or after niter iterations.  Here: when EVERY chain of the batch satisfies it.  The test runs on the device; once the flag is
  // set the kernels of the remaining iterations return at their first instruction (no host synchronisation anywhere).
  if (early) HIP_TRY(lmc::cg_check(C, rs, b2, tol2, done, st));
  for (int it = 0; it < niter; ++it) {
    HIP_TRY(hipMemsetAsync(pq, 0, sizeof(double) * 2 * C, st));     // pq and rs_new are adjacent
    HIP_TRY(apply(p, qq));
    if (!kname || std::strcmp(kname, "myula_step_rows_kernel") != 0) HIP_TRY(lmc::cg_dot(p, qq, C, img, pq, early ? done : nullptr, st));
    HIP_TRY(lmc::cg_update(u, r, p, qq, C, img, rs, pq, rs_new, early ? done : nullptr, st));
    if (early) HIP_TRY(lmc::cg_check(C, rs_new, b2, tol2, done, st));
    HIP_TRY(lmc::cg_dir(p, r, C, img, rs, rs_new, early ? done : nullptr, st));
    HIP_TRY(hipMemcpyAsync(rs, rs_new, sizeof(double) * C, hipMemcpyDeviceToDevice, st));
  }
  return LMC_OK;
}

bool needs_tv_state(const Problem& q) {
  return (q.prior_kind == LMC_PRIOR_TV_ISO && (q.tv_niter > 12 || (q.tv_rtol > 0.f && q.tv_niter > 10))) ||
         (q.ncvx_kind == LMC_NCVX_ME_TV && (q.ncvx_aniso || q.ncvx_niter > 12 || (q.ncvx_rtol > 0.f && q.ncvx_niter > 10)));
}

// The TV prox inside A (a complete StepArgs: prox only, or the whole fused update when A.tv.niter <= 10) with upstream's per-image early exit,
// decided on the device: every chain runs with the pass count it left in last time (rt.pred), the launch leaves the primal objectives of
// the iterates behind, tv_rt_decide replays upstream's test on them, and the chains whose prediction was wrong run again -- with the exact
// count when the objectives already show it, else with one pass more, then with all passes (whose objectives show it) and then once more.
// Four rounds settle every chain; the workgroups of settled chains return at once, so the later rounds cost a few microseconds when the
// predictions hold, and a chained prox re-runs from the link the change lies in, not from its first.  No host synchronisation.
// (lmc_ops.hip: tv_rt_begin / tv_rt_decide; lmc_step_pipe_rt.hip.)
constexpr int kRtRounds = 4;
int tv_prox_rt(lmc::StepArgs A, RtState& rt, float rtol, float* st0, float* st1, hipStream_t st) {
  const int niter = A.tv.niter;
  if (!rt.kc || rt.n < (size_t)A.C || rt.stride < niter + 1) return fail(LMC_E_STATE, "early-exit buffers are missing");
  A.rt_kc = rt.kc; A.rt_start = rt.start; A.rt_obj = rt.obj; A.rt_stride = rt.stride;
  HIP_TRY(lmc::launch_tv_rt_begin(A.C, rt.pred, rt.kc, rt.start, rt.obj, rt.stride, niter, st));
  for (int round = 0; round < kRtRounds; ++round) {
    hipError_t e = lmc::launch_step_pipe_rt(A, st, st0, st1);
    if (e == hipErrorInvalidConfiguration) return fail(LMC_E_UNSUPPORTED, "the device-side early exit of the TV prox does not cover this configuration");
    HIP_TRY(e);
    HIP_TRY(lmc::launch_tv_rt_decide(A.C, rt.kc, rt.start, rt.pred, rt.obj, rt.stride, niter, (double)rtol, round, rt.reruns + round, st));
  }
  return LMC_OK;
}

int tv_prox_rtol(const Problem& q, float pt, const float* x, float* sol, float* tmp, double* obj, int* flag, int64_t n, float* st0, float* st1,
                 hipStream_t st);

// out <- prox_{gam TV_1D}(x) of the n flattened images (N = H W entries each) by `niter` 1-D FGP iterations (lmc_ops.hip: tv1d_*), with upstream's early
// exit when rtol > 0 (pass by pass: the host reads the number of images still iterating after every pass).  buf: 4 n N floats (dual, dual, projected dual,
// iterate); obj: 2 n doubles; flag: n + 1 ints.
int tv1d_prox(const float* x, float* out, int64_t n, size_t N, float gam, int niter, float rtol, float* buf, double* obj, int* flag, hipStream_t st) {
  const size_t tot = (size_t)n * N;
  float *rr[2] = {buf, buf + tot}, *p = buf + 2 * tot, *tmp = buf + 3 * tot;
  float betas[lmc::kMaxTvIters];
  default_betas(betas, niter);
  const float cstep = 0.25f / gam;
  double *prev = obj, *cur = obj + n;
  int* n_active = flag + n;
  HIP_TRY(hipMemsetAsync(buf, 0, sizeof(float) * 3 * tot, st));
  HIP_TRY(hipMemsetAsync(obj, 0, sizeof(double) * 2 * n, st));
  HIP_TRY(hipMemsetAsync(flag, 0xFF, sizeof(int) * n, st));
  for (int j = 0; j <= niter; ++j) {
    const float* r_now = rr[j & 1];
    if (rtol > 0.f || j == niter) HIP_TRY(lmc::launch_tv1d_sol(x, r_now, tmp, n, N, gam, flag, st));
    if (j == niter) {                                                        // out of passes: the rest take sol_niter untested
      HIP_TRY(lmc::launch_tv_rtol_select(tmp, out, flag, -1, n, N, st));
      break;
    }
    if (rtol > 0.f) {
      HIP_TRY(lmc::launch_tv1d_objective(x, tmp, n, N, gam, flag, cur, st));
      HIP_TRY(hipMemsetAsync(n_active, 0, sizeof(int), st));
      HIP_TRY(lmc::launch_tv_rtol_decide(n, prev, cur, flag, j, (double)rtol, n_active, st));
      if (j > 0) HIP_TRY(lmc::launch_tv_rtol_select(tmp, out, flag, j, n, N, st));
      int active = 0;
      HIP_TRY(hipMemcpyAsync(&active, n_active, sizeof(int), hipMemcpyDeviceToHost, st));
      HIP_TRY(hipStreamSynchronize(st));
      if (active == 0) break;
    }
    HIP_TRY(lmc::launch_tv1d_iter(x, r_now, p, rr[(j & 1) ^ 1], n, N, gam, cstep, betas[j], flag, st));
  }
  return LMC_OK;
}

// extra <- prox_{gamma TV}(x) with ncvx_niter dual iterations (the inner prox of the ME-TV term, algs.py:169,282)
int me_tv_prox(const Problem& q, const float* x, float* extra, int64_t n_img, float* state0, float* state1, hipStream_t st, RtState* rt = nullptr) {
  lmc::StepArgs A;
  std::memset(&A, 0, sizeof A);
  A.H = q.H; A.W = q.W; A.C = (int)n_img;
  A.data_kind = LMC_DATA_NONE;
  A.prior_kind = LMC_PRIOR_TV_ISO;
  A.tv.niter = q.ncvx_niter;
  A.tv.gamma = q.ncvx_gamma;          // g_gamma = TV(dims, sigma = 1) evaluated at prox parameter gamma (algs.py:169,282)
  A.tv.c = 0.125f / q.ncvx_gamma;
  default_betas(A.tv.betas, q.ncvx_niter);
  A.a = 0.f; A.t = 0.f; A.b = 1.f; A.s = 0.f;
  A.noise_mode = LMC_NOISE_NONE;
  A.x_in = x; A.x_out = extra;
  A.y = x; A.mask = x; A.noise = x;
  if (q.ncvx_niter == 0) {     // lagged output of a 1-iteration prox: x itself
    HIP_TRY(hipMemcpyAsync(extra, x, sizeof(float) * (size_t)n_img * q.H * q.W, hipMemcpyDeviceToDevice, st));
    return LMC_OK;
  }
  if (q.ncvx_aniso) {          // algs.py:170: the 1-D TV of the flattened image
    if (!state0) return fail(LMC_E_STATE, "anisotropic ME-TV: work buffers are missing");
    Scratch& sc = g_scratch;
    HIP_TRY(sc.need_dbl(3 * (size_t)n_img + 2));
    return tv1d_prox(x, extra, n_img, (size_t)q.H * q.W, q.ncvx_gamma, q.ncvx_niter, q.ncvx_rtol, state0, sc.dbl, reinterpret_cast<int*>(sc.dbl + 2 * n_img), st);
  }
  if (q.ncvx_rtol > 0.f) {     // the class's own rtol (algs.py:130,169): per-chain early exit
    if (rt && q.tv_exit_path == 0 && lmc::pipe_rt_supported(A)) return tv_prox_rt(A, *rt, q.ncvx_rtol, state0, state1, st);     // on the device
    // elsewhere (narrow / wide / unaligned images): pass by pass, as the TV prior's prox (synchronises the stream)
    Problem qt;
    qt.H = q.H; qt.W = q.W;
    qt.prior_kind = LMC_PRIOR_TV_ISO; qt.prior_sigma = 1.f; qt.tv_niter = q.ncvx_niter; qt.tv_step = 0.125f; qt.tv_rtol = q.ncvx_rtol;
    default_betas(qt.betas, q.ncvx_niter);
    qt.variant = q.variant;
    Scratch& sc = g_scratch;
    HIP_TRY(sc.need_rtmp((size_t)n_img * q.H * q.W));
    HIP_TRY(sc.need_dbl(3 * (size_t)n_img + 2));
    return tv_prox_rtol(qt, q.ncvx_gamma, x, extra, sc.rtmp, sc.dbl, reinterpret_cast<int*>(sc.dbl + 2 * n_img), n_img, state0, state1, st);
  }
  hipError_t e = launch_step(A, variant_of(q), st, nullptr, state0, state1);
  if (e == hipErrorInvalidConfiguration) return fail(LMC_E_UNSUPPORTED, "no kernel covers the inner TV prox of the ME-TV term");
  HIP_TRY(e);
  return LMC_OK;
}


// prox_{pt g}(x), g = sigma TV, with upstream's per-image early exit (lmc_problem.tv_rtol > 0; pyproximal.TV.prox as restated by the
// CPU checker's tv_prox_fgp): at the top of pass j the iterate sol_j = x - gam div(r_j) (j dual updates) and its primal objective
// are formed; an image leaves with sol_j as soon as the relative change of the objective drops below rtol (never in pass 0); after
// tv_niter updates the iterate is returned untested.  Exact, pass by pass, for the whole batch: pass j's iterate is ONE fused launch
// with j dual iterations from the zero dual (the stages of a longer launch compute the same values), the objective a second one; images
// that have left keep their iterate (flag / select).  The host reads the number of images still iterating after every pass -- this
// path synchronises the stream, the fixed-count path (tv_rtol = 0) never does.  Typical MYULA iterates leave after 2-4 passes.
// sol, tmp: [n][H][W]; obj: 2n doubles (previous, current); flag: n + 1 ints (pass an image left in, -1 = iterating; then the counter).
int tv_prox_rtol(const Problem& q, float pt, const float* x, float* sol, float* tmp, double* obj, int* flag, int64_t n, float* st0, float* st1,
                 hipStream_t st) {
  const float gam = pt * q.prior_sigma;
  if (!(gam > 0.f)) return fail(LMC_E_INVALID, "TV prox parameter must be > 0 (got %g)", (double)gam);
  const size_t img = (size_t)q.H * q.W;
  double *prev = obj, *cur = obj + n;
  int* n_active = flag + n;
  HIP_TRY(hipMemsetAsync(obj, 0, sizeof(double) * 2 * n, st));
  HIP_TRY(hipMemsetAsync(flag, 0xFF, sizeof(int) * n, st));                  // -1: every image iterating
  const int K = q.tv_niter;
  for (int j = 0; j <= K; ++j) {
    const float* it = x;                                                     // pass 0: sol_0 = x
    if (j > 0) {
      Problem qj = q;
      qj.tv_niter = j;
      qj.ncvx_kind = LMC_NCVX_NONE;
      lmc::StepArgs A;
      int rc = make_step_args(qj, 0.f, 0.f, 1.f, pt, 0.f, A);
      if (rc) return rc;
      A.C = (int)n; A.x_in = x; A.x_out = tmp;
      sanitize_pointers(A);
      hipError_t e = launch_step(A, 0 /* auto: the passes have 1, 2, 3 ... dual iterations, no single variant covers them all */, st, nullptr, st0, st1);
      if (e == hipErrorInvalidConfiguration) return fail(LMC_E_UNSUPPORTED, "no step-kernel variant covers a TV prox with %d dual iterations", j);
      HIP_TRY(e);
      it = tmp;
    }
    if (j == K) {                                                            // out of passes: the rest take sol_K untested
      HIP_TRY(lmc::launch_tv_rtol_select(it, sol, flag, -1, n, img, st));
      break;
    }
    HIP_TRY(lmc::launch_tv_objective(x, it, n, q.H, q.W, gam, flag, cur, st));
    HIP_TRY(hipMemsetAsync(n_active, 0, sizeof(int), st));
    HIP_TRY(lmc::launch_tv_rtol_decide(n, prev, cur, flag, j, (double)q.tv_rtol, n_active, st));
    if (j > 0) HIP_TRY(lmc::launch_tv_rtol_select(it, sol, flag, j, n, img, st));
    int active = 0;
    HIP_TRY(hipMemcpyAsync(&active, n_active, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_TRY(hipStreamSynchronize(st));
    if (active == 0) break;
  }
  return LMC_OK;
}

// The TV PRIOR's prox with the early exit on the device.  A: the complete fused update (x_in, x_out, data term, noise ...).  Returns 1 when the
// update has been computed (up to 10 passes: the exit lives inside the fused step launch), 0 when the prox alone went to `proxbuf` and A now
// consumes it as a ready-made prox (more than 10 passes, or a data term the pipeline does not cover: the caller launches the step), 2 when the
// device path does not cover the problem (the caller takes the pass-by-pass path), or a negative status.
int tv_prior_rt(const Problem& q, float pt, lmc::StepArgs& A, RtState& rt, float* proxbuf, float* st0, float* st1, hipStream_t st) {
  if (A.tv.niter <= 10 && lmc::pipe_rt_supported(A)) {
    const int rc = tv_prox_rt(A, rt, q.tv_rtol, st0, st1, st);
    return rc ? rc : 1;
  }
  Problem qp = q;
  qp.ncvx_kind = LMC_NCVX_NONE;
  lmc::StepArgs P;
  int rc = make_step_args(qp, 0.f, 0.f, 1.f, pt, 0.f, P);
  if (rc) return rc;
  P.C = A.C; P.x_in = A.x_in; P.x_out = proxbuf;
  sanitize_pointers(P);
  if (!proxbuf || !lmc::pipe_rt_supported(P)) return 2;
  rc = tv_prox_rt(P, rt, q.tv_rtol, st0, st1, st);
  if (rc) return rc;
  A.prior_kind = LMC_PRIOR_NONE;
  A.prox_ext = proxbuf;
  return 0;
}
// which of the two the sampler / call will take: 1 fused, 0 prox alone, 2 not covered
int tv_prior_rt_mode(const Problem& q, const lmc::StepArgs& A_probe, float pt) {
  if (q.tv_exit_path != 0) return 2;
  if (A_probe.tv.niter <= 10 && lmc::pipe_rt_supported(A_probe)) return 1;
  Problem qp = q;
  qp.ncvx_kind = LMC_NCVX_NONE;
  lmc::StepArgs P;
  if (make_step_args(qp, 0.f, 0.f, 1.f, pt, 0.f, P)) return 2;
  P.C = A_probe.C;
  return lmc::pipe_rt_supported(P) ? 0 : 2;
}

int env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
}  // namespace

struct lmc_sampler {
  int kind = 0;   // 0 MYULA, 1 ULPDA, 2 MYMALA
  int device = -1;   // the device the handle's buffers live on (current at creation); every call on the handle runs there
  // ULPDA state (kind == 1)
  float mu = 0, theta = 1;
  int gfirst = 0, cg_niter = 0, warm = 1;
  const float* z = nullptr;
  float* xhat = nullptr; float* ydual = nullptr; float* uw = nullptr; float* rhs = nullptr;
  float* ydual2 = nullptr;    // ULPDA: the dual array the fused dual + right-hand-side pass writes (swapped with ydual after it)
  bool rhs_ready = false;     // ULPDA: s->rhs already holds the right-hand side of the NEXT iteration for (rhs_tau, rhs_ts), formed by that pass
  float rhs_tau = 0.f, rhs_ts = 0.f;
  float* uw2 = nullptr;       // ULPDA: the other home of the implicit-step solution (two Chebyshev iterations per launch deliver it there)
  float* cr = nullptr; float* cp = nullptr; float* cq = nullptr; float* ctmp = nullptr; float* xi = nullptr;
  float* htb = nullptr; double* scal = nullptr; float* zero_y = nullptr;
  float* tvstate[2] = {nullptr, nullptr};   // dual-state ping-pong for chunked TV proxes (K > 12, ME-TV)
  float* tvwarm[2] = {nullptr, nullptr};    // warm-started TV prox: projected dual (p, q) of the previous / this MYULA iteration, [C][2][H][W]
  int wcur = 0;
  float* extra = nullptr;                   // ME-TV inner prox
  float* pxbuf = nullptr;                   // Haar-l1 prox / early-exit TV prox of the current state
  float* rtmp = nullptr; double* robj = nullptr; int* rflag = nullptr;   // early-exit TV prox (tv_rtol > 0), pass-by-pass path: pass iterate, objectives, flags
  RtState rt_tv, rt_me;                     // early-exit TV prox on the device (tv_prox_rt): of the TV prior / of the ME-TV inner prox
  // launch policy, fixed at creation (lmc_problem fields; their environment variables supply the defaults): see lmc_atomi.h
  int pol_pair = 1;          // 0 never, 1 where it pays, 2 wherever covered: two MYULA iterations per launch (rows kernel)
  int pol_blockpair = 1;     // 0 / 1: two or four iterations per launch on the block kernel
  int pol_overlap = 0;       // 0 by size, 1 on, -1 off
  int pol_bg_wgs = -1;       // workgroups of the background reduction, -1 by size
  int pol_graph = 0;
  int pol_side_lowprio = 1;
  bool pol_ulpda_dual_rhs = false;   // ULPDA, opt-in experiment (LMC_ULPDA_DUAL_RHS=1 at creation): dual update fused with the next right-hand side
  Problem prob;
  int C = 0;
  int64_t chain_offset = 0;
  float tau = 0, gamma = 0, epsg = 1;
  uint64_t seed = 0;
  int noise_mode = 0;
  int moments = 0, burn_in = 0, thin = 1;
  int64_t iteration = 0;
  uint64_t count = 0;
  float* x[2] = {nullptr, nullptr};
  float* xspare = nullptr;    // third state array of the two-iterations-per-launch MYULA path (allocated at its first use)
  // Moment reductions of the pair launches on the side stream: the iterate in between goes to one of two arrays of its own (a pair launch two
  // launches later is the first that may write where launch n's reductions read: ev_pair[n & 1] orders that)
  float* xmid[2] = {nullptr, nullptr};
  hipEvent_t ev_pair[2] = {nullptr, nullptr};
  bool pair_pending[2] = {false, false};
  uint64_t pair_n = 0;
  int cur = 0;
  double* s1 = nullptr;
  double* s2 = nullptr;
  double* packed = nullptr;              // [2 H W + 1]: the send / receive buffer of lmc_allreduce_moments
  lmc::StepArgs base{};
  // MYMALA state (kind == 2): proposal mean of the current state, proposal, its mean, energies, decisions
  float* mx = nullptr; float* xp = nullptr; float* mxp = nullptr;
  double* mala_d = nullptr;              // [5C]: U(x), f(x'), g(x'), ||x'-m(x)||^2, ||x-m(x')||^2 ; then [C] log alpha
  int* flag = nullptr;
  unsigned long long* nacc = nullptr;
  bool mala_fresh = false;               // mx / U match x[cur]
  // moment reductions on a side stream, overlapping the next step kernel (HBM-bound reduction under a VALU-bound step kernel)
  hipStream_t side = nullptr;
  hipEvent_t ev_step = nullptr;              // "the step that wrote x[cur] is done" (recorded on the caller's stream)
  hipEvent_t ev_mom[2] = {nullptr, nullptr}; // "the reduction that reads x[i] is done" (recorded on the side stream)
  bool mom_pending[2] = {false, false};
  // hipGraph replay of kGraphIters iterations at a time (small configurations, where launch gaps and the serial moment reduction
  // are a large part of an iteration): step kernels on the caller's stream, the moment reduction of iteration k on a captured side
  // branch under the step kernel of iteration k + 1; the Philox iteration word comes from device memory (StepArgs.iter_dev)
  hipGraphExec_t gexec[2] = {nullptr, nullptr};   // by `cur` at the start of the block of iterations
  uint32_t* iter_dev = nullptr;
  hipStream_t gmain = nullptr, gside = nullptr;   // capture streams (the caller's stream may be the legacy default stream, which cannot capture)
  std::vector<hipEvent_t> gev;
  bool plain_done = false;      // at least one ordinary launch has happened (function attributes set, kernel known)
  std::vector<hipEvent_t> ev;   // pairs (begin, end) around each step-kernel launch of the last step() call
  bool timing = false;
  bool timed = false;
  int last_launches = 0;
  std::string kernel_name;
};

extern "C" {

int lmc_version(void) { return LMC_ATOMI_ABI_VERSION; }

const char* lmc_last_error(void) { return g_err.c_str(); }

int lmc_device_info(int* device, int* n_cu, size_t* lds_bytes, size_t* hbm_bytes) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  hipDeviceProp_t pr;
  HIP_TRY(hipGetDeviceProperties(&pr, dev));
  if (device) *device = dev;
  if (n_cu) *n_cu = pr.multiProcessorCount;
  if (lds_bytes) *lds_bytes = pr.maxSharedMemoryPerMultiProcessor;
  if (hbm_bytes) *hbm_bytes = pr.totalGlobalMem;
  return LMC_OK;
}

int lmc_hbm_copy_probe(size_t bytes, int32_t reps, float* gbs_out, void* stream) {
  if (!gbs_out || reps < 1 || bytes < (1u << 20)) return fail(LMC_E_INVALID, "bad arguments (at least 1 MiB, reps >= 1)");
  const size_t n = (bytes / 16) * 4;                       // floats, a multiple of 4
  hipStream_t st = S(stream);
  float *x = nullptr, *y = nullptr;
  hipEvent_t e0 = nullptr, e1 = nullptr;
  hipError_t e = hipMalloc(&x, n * sizeof(float));
  if (e == hipSuccess) e = hipMalloc(&y, n * sizeof(float));
  if (e == hipSuccess) e = hipMemsetAsync(x, 0, n * sizeof(float), st);
  if (e == hipSuccess) e = hipEventCreate(&e0);
  if (e == hipSuccess) e = hipEventCreate(&e1);
  float best = 0.f;
  for (int shape = 0; shape < lmc::hbm_copy_probe_shapes(); ++shape)     // launch shapes: what a plain copy reaches depends on them
    for (int r = 0; r < reps + 1 && e == hipSuccess; ++r) {   // the first pass warms up (page mapping, clocks)
      e = hipEventRecord(e0, st);
      if (e == hipSuccess) e = lmc::launch_hbm_copy_probe(x, y, n, shape, st);
      if (e == hipSuccess) e = hipEventRecord(e1, st);
      if (e == hipSuccess) e = hipEventSynchronize(e1);
      float ms = 0.f;
      if (e == hipSuccess) e = hipEventElapsedTime(&ms, e0, e1);
      if (e == hipSuccess && r > 0 && ms > 0.f) best = fmaxf(best, (float)(2.0 * (double)n * sizeof(float) / ((double)ms * 1e-3) / 1e9));
    }
  if (e0) (void)hipEventDestroy(e0);
  if (e1) (void)hipEventDestroy(e1);
  if (x) (void)hipFree(x);
  if (y) (void)hipFree(y);
  if (e != hipSuccess) return fail(e == hipErrorOutOfMemory ? LMC_E_NOMEM : LMC_E_HIP, "HBM probe failed: %s", hipGetErrorString(e));
  *gbs_out = best;
  return LMC_OK;
}

int lmc_blur(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, const float* h_host, int32_t kh,
             int32_t kw, int32_t oy, int32_t ox, int32_t adjoint, void* stream) {
  if (!x_dev || !out_dev) return fail(LMC_E_INVALID, "NULL image pointer");
  if (x_dev == out_dev) return fail(LMC_E_INVALID, "lmc_blur cannot run in place");
  if (n_img < 1 || H < 1 || W < 1) return fail(LMC_E_INVALID, "bad shape n_img=%lld H=%d W=%d", (long long)n_img, H, W);
  lmc::BlurTaps T;
  int rc = fill_taps(T, h_host, kh, kw, oy, ox);
  if (rc) return rc;
  HIP_TRY(lmc::launch_blur(x_dev, out_dev, n_img, H, W, T, adjoint != 0, S(stream)));
  return LMC_OK;
}

int lmc_gradient(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, void* stream) {
  if (!x_dev || !out_dev || x_dev == out_dev) return fail(LMC_E_INVALID, "bad pointers");
  if (n_img < 1 || H < 1 || W < 1) return fail(LMC_E_INVALID, "bad shape");
  HIP_TRY(lmc::launch_gradient(x_dev, out_dev, n_img, H, W, false, S(stream)));
  return LMC_OK;
}

int lmc_gradient_adjoint(const float* y_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, void* stream) {
  if (!y_dev || !out_dev || y_dev == out_dev) return fail(LMC_E_INVALID, "bad pointers");
  if (n_img < 1 || H < 1 || W < 1) return fail(LMC_E_INVALID, "bad shape");
  HIP_TRY(lmc::launch_gradient(y_dev, out_dev, n_img, H, W, true, S(stream)));
  return LMC_OK;
}

int lmc_fused_eval(const lmc_problem* prob, const float* x_dev, float* out_dev, int64_t n_img, float a, float t, float b,
                   float pt, void* stream) {
  Problem q;
  int rc = load_problem(prob, q);
  if (rc) return rc;
  if (!x_dev || !out_dev || x_dev == out_dev) return fail(LMC_E_INVALID, "bad pointers (in-place not allowed)");
  if (n_img < 1 || n_img > (1 << 24)) return fail(LMC_E_INVALID, "bad n_img %lld", (long long)n_img);
  if (q.prox_scale) return fail(LMC_E_UNSUPPORTED, "prox_scale (array-valued epsg) belongs to the MYULA sampler");
  lmc::StepArgs A;
  rc = make_step_args(q, a, t, b, pt, 0.f, A);
  if (rc) return rc;
  A.C = (int)n_img;
  A.x_in = x_dev;
  A.x_out = out_dev;
  sanitize_pointers(A);
  const size_t npx = (size_t)n_img * q.H * q.W;
  if (needs_tv_state(q)) HIP_TRY(g_scratch.need_state(4 * npx));
  if (t != 0.f && q.ncvx_kind == LMC_NCVX_ME_TV) {
    HIP_TRY(g_scratch.need_extra(npx));
    if (q.ncvx_rtol > 0.f) HIP_TRY(g_scratch.rt_me.need((size_t)n_img, q.ncvx_niter));
    rc = me_tv_prox(q, x_dev, g_scratch.extra, n_img, g_scratch.state[0], g_scratch.state[1], S(stream), &g_scratch.rt_me);
    if (rc) return rc;
    A.extra = g_scratch.extra;
    A.extra_coef = -q.ncvx_lambda / q.ncvx_gamma;
  }
  if (A.prior_kind == LMC_PRIOR_HAAR_L1 || A.prior_kind == LMC_PRIOR_EPROX) HIP_TRY(g_scratch.need_prox(npx));
  if (A.prior_kind == LMC_PRIOR_TV_ISO && q.tv_rtol > 0.f && tv_prior_rt_mode(q, A, pt) != 2) {     // the early exit decided on the device
    Scratch& sc = g_scratch;
    HIP_TRY(sc.need_prox(npx));
    HIP_TRY(sc.rt_tv.need((size_t)n_img, q.tv_niter));
    if (needs_tv_state(q)) HIP_TRY(sc.need_state(4 * npx));
    rc = tv_prior_rt(q, pt, A, sc.rt_tv, sc.prox, sc.state[0], sc.state[1], S(stream));
    if (rc < 0) return rc;
    if (rc == 1) return LMC_OK;
    if (rc == 2) return fail(LMC_E_UNSUPPORTED, "early exit of the TV prox: no path covers this configuration");
  } else if (A.prior_kind == LMC_PRIOR_TV_ISO && q.tv_rtol > 0.f) {     // the early-exit prox first, then the fused update with it as a ready-made prox
    Scratch& sc = g_scratch;
    HIP_TRY(sc.need_prox(npx));
    HIP_TRY(sc.need_rtmp(npx));
    HIP_TRY(sc.need_dbl(3 * (size_t)n_img + 2));
    rc = tv_prox_rtol(q, pt, x_dev, sc.prox, sc.rtmp, sc.dbl, reinterpret_cast<int*>(sc.dbl + 2 * n_img), n_img, sc.state[0], sc.state[1], S(stream));
    if (rc) return rc;
    A.prior_kind = LMC_PRIOR_NONE;
    A.prox_ext = sc.prox;
  }
  hipError_t e = launch_step(A, variant_of(q), S(stream), nullptr, g_scratch.state[0], g_scratch.state[1], g_scratch.prox);
  if (e == hipErrorInvalidConfiguration) return fail(LMC_E_UNSUPPORTED, "no step-kernel variant covers this configuration");
  HIP_TRY(e);
  return LMC_OK;
}

static lmc::EnergyArgs energy_args(const Problem& q) {
  lmc::EnergyArgs E;
  E.H = q.H; E.W = q.W; E.data_kind = q.data_kind; E.sigma_f = q.sigma_f; E.y = q.y; E.mask = q.mask;
  E.blur = q.taps; E.prior_kind = q.prior_kind; E.prior_sigma = q.prior_sigma;
  E.ncvx_kind = q.ncvx_kind == LMC_NCVX_MC_TV ? LMC_NCVX_MC_TV : LMC_NCVX_NONE;   // ME-TV envelope: me_tv_energy
  E.ncvx_lambda = q.ncvx_lambda; E.ncvx_gamma = q.ncvx_gamma;
  return E;
}

// f_out -= lambda * ( TV(prox) + ||x - prox||^2 / (2 gamma) ),  prox = prox_{gamma TV}(x)    (algs.py:178-190, ME-TV)
static int me_tv_energy(const Problem& q, const float* x, int64_t n_img, double* f_out, float* extra, float* st0, float* st1,
                        double* dbl /* 2*n_img */, hipStream_t st, RtState* rt) {
  int rc = me_tv_prox(q, x, extra, n_img, st0, st1, st, rt);
  if (rc) return rc;
  lmc::EnergyArgs E;
  std::memset(&E, 0, sizeof E);
  E.H = q.H; E.W = q.W; E.data_kind = LMC_DATA_NONE; E.prior_kind = LMC_PRIOR_TV_ISO; E.prior_sigma = 1.f;
  if (q.ncvx_aniso) {                                                                    // TV_1D(prox) of the flattened image
    HIP_TRY(hipMemsetAsync(dbl, 0, sizeof(double) * n_img, st));
    HIP_TRY(lmc::launch_tv1d_objective(extra, extra, n_img, (size_t)q.H * q.W, 1.f, nullptr, dbl, st));
  } else
  HIP_TRY(lmc::launch_energies(extra, n_img, E, nullptr, dbl, st));                       // TV(prox)
  HIP_TRY(lmc::launch_sqdiff(x, extra, n_img, (size_t)q.H * q.W, dbl + n_img, st));      // ||x - prox||^2
  HIP_TRY(lmc::launch_axpy_env(f_out, dbl, dbl + n_img, n_img, q.ncvx_lambda, q.ncvx_gamma, st));
  return LMC_OK;
}

int lmc_energies(const lmc_problem* prob, const float* x_dev, int64_t n_img, double* f_out_dev, double* g_out_dev,
                 void* stream) {
  Problem q;
  int rc = load_problem(prob, q);
  if (rc) return rc;
  if (!x_dev || n_img < 1) return fail(LMC_E_INVALID, "bad arguments");
  HIP_TRY(lmc::launch_energies(x_dev, n_img, energy_args(q), f_out_dev, g_out_dev, S(stream)));
  if (q.prior_kind == LMC_PRIOR_HAAR_L1 && g_out_dev) HIP_TRY(lmc::launch_haar_value(x_dev, n_img, q.H, q.W, q.prior_sigma, g_out_dev, S(stream)));
  if (q.ncvx_kind == LMC_NCVX_ME_TV && f_out_dev) {
    const size_t npx = (size_t)n_img * q.H * q.W;
    HIP_TRY(g_scratch.need_state(4 * npx));
    HIP_TRY(g_scratch.need_extra(npx));
    HIP_TRY(g_scratch.need_dbl(3 * (size_t)n_img + 2));     // (the pass-by-pass fallback of the inner prox shares the buffer: sized for it up front)
    if (q.ncvx_rtol > 0.f) HIP_TRY(g_scratch.rt_me.need((size_t)n_img, q.ncvx_niter));
    rc = me_tv_energy(q, x_dev, n_img, f_out_dev, g_scratch.extra, g_scratch.state[0], g_scratch.state[1], g_scratch.dbl, S(stream), &g_scratch.rt_me);
    if (rc) return rc;
  }
  return LMC_OK;
}

size_t lmc_l2_prox_workspace_bytes(int64_t n_img, int32_t H, int32_t W) {
  const size_t n = (size_t)n_img * H * W;
  return ((5 * n * sizeof(float) + 7) / 8) * 8 + (4 * (size_t)n_img + 1) * sizeof(double);
}

int lmc_l2_prox(const lmc_problem* prob, const float* x_dev, float* out_dev, int64_t n_img, float tau, int32_t niter,
                int32_t warm, void* workspace_dev, void* stream) {
  Problem q;
  int rc = load_problem(prob, q);
  if (rc) return rc;
  if (!x_dev || !out_dev || x_dev == out_dev || n_img < 1) return fail(LMC_E_INVALID, "bad arguments (in-place not allowed)");
  if (!(tau > 0.f)) return fail(LMC_E_INVALID, "tau must be > 0");
  hipStream_t st = S(stream);
  const float ts = tau * q.sigma_f;
  const size_t n = (size_t)n_img * q.H * q.W;
  if (q.data_kind != LMC_DATA_BLUR) {
    const float* xin = x_dev;
    if (q.ncvx_kind != LMC_NCVX_NONE && q.data_kind != LMC_DATA_NONE) {
      // L2_ncvx_tv.prox with a pointwise data term: the pre-step of algs.py:213-223 first (x + tau lambda A^T(Ax / max(|Ax|, gamma)), or
      // x + tau lambda / gamma (x - prox_{gamma TV}(x))), then the closed-form solve.  (Round 3's matrix test found this path applying the solve to x itself.)
      HIP_TRY(g_scratch.need_prox(n));
      if (q.ncvx_kind == LMC_NCVX_MC_TV) {
        HIP_TRY(lmc::ulpda_ncvx_rhs(x_dev, q.y, g_scratch.prox, n_img, q.H, q.W, tau * q.ncvx_lambda, q.ncvx_gamma, 0.f, st));     // ts = 0: the H^T b slot adds nothing
      } else {
        HIP_TRY(g_scratch.need_extra(n));
        if (needs_tv_state(q)) HIP_TRY(g_scratch.need_state(4 * n));
        if (q.ncvx_rtol > 0.f) HIP_TRY(g_scratch.rt_me.need((size_t)n_img, q.ncvx_niter));
        rc = me_tv_prox(q, x_dev, g_scratch.extra, n_img, g_scratch.state[0], g_scratch.state[1], st, &g_scratch.rt_me);
        if (rc) return rc;
        HIP_TRY(lmc::ulpda_me_rhs(x_dev, g_scratch.extra, q.y, g_scratch.prox, n_img, q.H, q.W, tau * q.ncvx_lambda / q.ncvx_gamma, 0.f, st));
      }
      xin = g_scratch.prox;
    }
    HIP_TRY(lmc::ulpda_pointwise_prox(xin, out_dev, q.y, q.mask, n_img, q.H, q.W, ts, q.data_kind, st));
    return LMC_OK;
  }
  if (niter < 1) return fail(LMC_E_INVALID, "niter must be >= 1");
  if (!workspace_dev) return fail(LMC_E_INVALID, "workspace_dev is NULL (see lmc_l2_prox_workspace_bytes)");
  float* w = static_cast<float*>(workspace_dev);
  float *rhs = w, *r = w + n, *p = w + 2 * n, *qq = w + 3 * n, *tmp = w + 4 * n;
  double* scal = reinterpret_cast<double*>(static_cast<char*>(workspace_dev) + ((5 * n * sizeof(float) + 7) / 8) * 8);
  // rhs = x + ts * H^T y  : H^T y into tmp (one image), then broadcast-add through the rhs kernel with y == 0
  HIP_TRY(lmc::launch_blur(q.y, tmp, 1, q.H, q.W, q.taps, 1, st));
  // ulpda_rhs computes x - tau*(A^T ydual) + ts*htb; with a zero dual field it is x + ts*htb.  The dual field needs 2n
  // floats: use r and p (both overwritten later by the solver) as a zeroed [n_img][2][H][W] field.
  if (q.ncvx_kind == LMC_NCVX_MC_TV) {
    HIP_TRY(lmc::ulpda_ncvx_rhs(x_dev, tmp, rhs, n_img, q.H, q.W, tau * q.ncvx_lambda, q.ncvx_gamma, ts, st));
  } else if (q.ncvx_kind == LMC_NCVX_ME_TV) {
    HIP_TRY(g_scratch.need_extra(n));
    if (needs_tv_state(q)) HIP_TRY(g_scratch.need_state(4 * n));
    if (q.ncvx_rtol > 0.f) HIP_TRY(g_scratch.rt_me.need((size_t)n_img, q.ncvx_niter));
    rc = me_tv_prox(q, x_dev, g_scratch.extra, n_img, g_scratch.state[0], g_scratch.state[1], st, &g_scratch.rt_me);
    if (rc) return rc;
    HIP_TRY(lmc::ulpda_me_rhs(x_dev, g_scratch.extra, tmp, rhs, n_img, q.H, q.W, tau * q.ncvx_lambda / q.ncvx_gamma, ts, st));
  } else {
    HIP_TRY(hipMemsetAsync(r, 0, sizeof(float) * 2 * n, st));
    HIP_TRY(lmc::ulpda_rhs(x_dev, r, nullptr, tmp, rhs, n_img, q.H, q.W, 0.f, ts, st));
  }
  if (!warm) HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float) * n, st));
  HIP_TRY(hipMemsetAsync(tmp, 0, sizeof(float) * (size_t)q.H * q.W, st));   // tmp (H^T y) is consumed: reuse its first image as the zero observation
  rc = cg_solve_fused(q, ts, out_dev, rhs, r, p, qq, scal, n_img, niter, tmp, st);
  if (rc) return rc;
  return LMC_OK;
}

int lmc_haar_l1_prox(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, float thr, void* stream) {
  if (!x_dev || !out_dev || n_img < 1 || H < 8 || W < 8) return fail(LMC_E_INVALID, "bad arguments");
  if ((H & 7) || (W & 7)) return fail(LMC_E_UNSUPPORTED, "H and W must be multiples of 8 (got %dx%d)", H, W);
  if (!(thr >= 0.f)) return fail(LMC_E_INVALID, "threshold must be >= 0");
  HIP_TRY(lmc::launch_haar_prox(x_dev, out_dev, n_img, H, W, thr, S(stream)));
  return LMC_OK;
}

int lmc_chain_probes(const float* x_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, int32_t ph, int32_t pw, void* stream) {
  if (!x_dev || !out_dev || n_img < 1 || H < 1 || W < 1) return fail(LMC_E_INVALID, "bad arguments");
  if (ph < 1 || pw < 1 || ph > H || pw > W) return fail(LMC_E_INVALID, "probe grid %dx%d does not fit a %dx%d image", ph, pw, H, W);
  if (W > 8192) return fail(LMC_E_UNSUPPORTED, "W <= 8192 (got %d)", W);
  HIP_TRY(lmc::launch_chain_probes(x_dev, out_dev, n_img, H, W, ph, pw, S(stream)));
  return LMC_OK;
}

int lmc_dual_project(const float* y_dev, float* out_dev, int64_t n_img, int32_t H, int32_t W, float radius,
                     int32_t isotropic, void* stream) {
  if (!y_dev || !out_dev) return fail(LMC_E_INVALID, "NULL pointer");
  if (n_img < 1 || H < 1 || W < 1 || !(radius > 0.f)) return fail(LMC_E_INVALID, "bad shape or radius");
  HIP_TRY(lmc::launch_dual_project(y_dev, out_dev, n_img, H, W, radius, isotropic, S(stream)));
  return LMC_OK;
}

int lmc_prox_elementwise(int32_t kind, const float* x_dev, float* out_dev, int64_t n, const float* params_host,
                         int32_t n_params, void* stream) {
  static const int need[] = {1, 2, 1, 1, 1, 1, 1, 2, 1, 1, 2, 1, 1, 2, 1};
  if (kind < 0 || kind > LMC_EPROX_LAPLACE_CONJ) return fail(LMC_E_INVALID, "unknown elementwise prox %d", kind);
  if (!x_dev || !out_dev || n < 1) return fail(LMC_E_INVALID, "bad arguments");
  if (n_params != need[kind] || !params_host)
    return fail(LMC_E_INVALID, "elementwise prox %d takes %d parameter(s), got %d", kind, need[kind], n_params);
  const float p0 = params_host[0], p1 = n_params > 1 ? params_host[1] : 0.f;
  HIP_TRY(lmc::launch_eprox(kind, x_dev, out_dev, n, p0, p1, S(stream)));
  return LMC_OK;
}

// ---- sampler ---------------------------------------------------------------------------------

int lmc_myula_create(const lmc_myula_config* cfg, lmc_sampler** out) {
  if (!cfg || !out) return fail(LMC_E_INVALID, "NULL argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(lmc_myula_config))
    return fail(LMC_E_INVALID, "lmc_myula_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(lmc_myula_config));
  if (cfg->n_chains < 1) return fail(LMC_E_INVALID, "n_chains must be >= 1");
  if (cfg->chain_offset < 0 || cfg->chain_offset + cfg->n_chains > 0xFFFFFFFFLL)
    return fail(LMC_E_INVALID, "global chain ids must fit 32 bits");
  if (!(cfg->tau > 0.f) || !(cfg->gamma > 0.f)) return fail(LMC_E_INVALID, "tau and gamma must be > 0");
  if (cfg->noise_mode < LMC_NOISE_PHILOX || cfg->noise_mode > LMC_NOISE_NONE) return fail(LMC_E_INVALID, "bad noise_mode");
  if (cfg->moments && cfg->thin < 1) return fail(LMC_E_INVALID, "thin must be >= 1");
  lmc_sampler* s = new (std::nothrow) lmc_sampler();
  if (!s) return fail(LMC_E_NOMEM, "host allocation failed");
  int rc = load_problem(&cfg->problem, s->prob);
  if (rc) { delete s; return rc; }
  if (hipGetDevice(&s->device) != hipSuccess) { delete s; return fail(LMC_E_HIP, "hipGetDevice failed"); }
  s->C = cfg->n_chains;
  s->chain_offset = cfg->chain_offset;
  s->tau = cfg->tau; s->gamma = cfg->gamma; s->epsg = cfg->epsg;
  s->seed = cfg->seed;
  s->noise_mode = cfg->noise_mode;
  s->moments = cfg->moments; s->burn_in = cfg->burn_in; s->thin = cfg->thin < 1 ? 1 : cfg->thin;
  // x <- (1 - tau/gamma) x - tau grad f(x) + (tau/gamma) prox_{epsg*gamma*g}(x) + sqrt(2 tau) xi   (algs.py:569)
  rc = make_step_args(s->prob, 1.f - s->tau / s->gamma, s->tau, s->tau / s->gamma, s->epsg * s->gamma,
                      std::sqrt(2.f * s->tau), s->base);
  if (rc) { delete s; return rc; }
  s->base.C = s->C;
  s->base.noise_mode = s->noise_mode;
  s->base.key0 = (uint32_t)(s->seed & 0xFFFFFFFFu);
  s->base.key1 = (uint32_t)(s->seed >> 32);
  s->base.chain_offset = (uint32_t)s->chain_offset;
  const size_t nbytes = sizeof(float) * (size_t)s->C * s->prob.H * s->prob.W;
  hipError_t e = hipMalloc(&s->x[0], nbytes);
  if (e == hipSuccess) e = hipMalloc(&s->x[1], nbytes);
  if (e == hipSuccess) e = hipMemset(s->x[0], 0, nbytes);
  if (e == hipSuccess && needs_tv_state(s->prob)) {
    e = hipMalloc(&s->tvstate[0], 4 * nbytes);
    if (e == hipSuccess) e = hipMalloc(&s->tvstate[1], 4 * nbytes);
  }
  if (e == hipSuccess && s->prob.ncvx_kind == LMC_NCVX_ME_TV) e = hipMalloc(&s->extra, nbytes);
  if (e == hipSuccess && (s->prob.prior_kind == LMC_PRIOR_HAAR_L1 || s->prob.prior_kind == LMC_PRIOR_EPROX || s->prob.prox_scale)) e = hipMalloc(&s->pxbuf, nbytes);
  if (e == hipSuccess && s->prob.tv_rtol > 0.f && s->base.prior_kind == LMC_PRIOR_TV_ISO) {
    if (s->prob.tv_warm) { lmc_sampler_destroy(s); return fail(LMC_E_UNSUPPORTED, "tv_rtol > 0 and tv_warm exclude each other"); }
    const int mode = tv_prior_rt_mode(s->prob, s->base, s->epsg * s->gamma);     // 1: inside the fused launch, 0: prox alone, 2: pass by pass
    if (mode != 2) e = s->rt_tv.need((size_t)s->C, s->prob.tv_niter);
    if (e == hipSuccess && mode != 1 && !s->pxbuf) e = hipMalloc(&s->pxbuf, nbytes);
    if (mode == 2) {
      if (e == hipSuccess) e = hipMalloc(&s->rtmp, nbytes);
      if (e == hipSuccess) e = hipMalloc(&s->robj, sizeof(double) * 2 * (size_t)s->C);
      if (e == hipSuccess) e = hipMalloc(&s->rflag, sizeof(int) * ((size_t)s->C + 1));
    }
  }
  if (e == hipSuccess && s->prob.ncvx_kind == LMC_NCVX_ME_TV && s->prob.ncvx_rtol > 0.f) e = s->rt_me.need((size_t)s->C, s->prob.ncvx_niter);
  {   // launch policy: the lmc_problem fields, their environment variables where a field is 0 -- read here, once, never inside lmc_sampler_step
    const Problem& q = s->prob;
    const int ipl = q.iters_per_launch ? q.iters_per_launch : env_int("LMC_ITERS_PER_LAUNCH", 0);
    s->pol_pair = ipl == 1 ? 0 : (ipl == 2 ? 2 : env_int("LMC_ROWS_PAIR", 1));
    s->pol_blockpair = ipl == 1 ? 0 : (ipl == 2 ? 1 : (env_int("LMC_BLOCK_PAIR", 1) != 0));
    s->pol_overlap = q.moments_overlap ? q.moments_overlap : (getenv("LMC_MOMENTS_OVERLAP") ? (env_int("LMC_MOMENTS_OVERLAP", 0) ? 1 : -1) : 0);
    s->pol_bg_wgs = q.moments_bg_wgs > 0 ? q.moments_bg_wgs : env_int("LMC_MOMENTS_BG_WGS", -1);
    s->pol_graph = q.graph_replay ? 1 : (env_int("LMC_GRAPH", 0) == 1);
    s->pol_side_lowprio = env_int("LMC_MOMENTS_SIDE_PRIO", 1) != 0;
    if (q.prox_scale) { s->pol_pair = 0; s->pol_blockpair = 0; s->pol_graph = 0; }   // array-valued epsg: the prox is its own launch before every step
  }
  if (e == hipSuccess && s->prob.tv_warm) {
    lmc::StepArgs probe = s->base;
    probe.x_in = s->x[0];
    if (s->base.prior_kind != LMC_PRIOR_TV_ISO || !lmc::pipe_warm_supported(probe)) {
      lmc_sampler_destroy(s);
      return fail(LMC_E_UNSUPPORTED, "tv_warm: needs tv_niter in {1, 2, 3} (after tv_lagged_output) and the full-width pipeline kernel "
                  "(W > 128, W %% 4 == 0 (%% 8 above 256), separable blur <= 7 taps / pointwise / no data term)");
    }
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipMalloc(&s->tvwarm[i], 2 * nbytes);
      if (e == hipSuccess) e = hipMemset(s->tvwarm[i], 0, 2 * nbytes);
    }
  }
  if (e == hipSuccess && s->moments) {
    const size_t mb = sizeof(double) * (size_t)s->prob.H * s->prob.W;
    e = hipMalloc(&s->s1, mb);
    if (e == hipSuccess) e = hipMalloc(&s->s2, mb);
    if (e == hipSuccess) e = hipMemset(s->s1, 0, mb);
    if (e == hipSuccess) e = hipMemset(s->s2, 0, mb);
  }
  if (e != hipSuccess) {
    rc = fail(e == hipErrorOutOfMemory ? LMC_E_NOMEM : LMC_E_HIP, "sampler allocation failed: %s", hipGetErrorString(e));
    lmc_sampler_destroy(s);
    return rc;
  }
  s->kernel_name = "(no step launched yet)";
  *out = s;
  return LMC_OK;
}

void lmc_sampler_destroy(lmc_sampler* s) {
  if (!s) return;
  DeviceGuard dg(s->device);
  for (float* b : {s->xmid[0], s->xmid[1], s->ydual2, s->xspare, s->zero_y, s->xhat, s->ydual, s->uw, s->uw2, s->rhs, s->cr, s->cp, s->cq, s->ctmp, s->xi, s->htb, s->tvstate[0], s->tvstate[1], s->extra, s->pxbuf,
                   s->mx, s->xp, s->mxp, s->tvwarm[0], s->tvwarm[1], s->rtmp})
    if (b) (void)hipFree(b);
  if (s->robj) (void)hipFree(s->robj);
  if (s->rflag) (void)hipFree(s->rflag);
  s->rt_tv.release();
  s->rt_me.release();
  if (s->mala_d) (void)hipFree(s->mala_d);
  if (s->flag) (void)hipFree(s->flag);
  if (s->nacc) (void)hipFree(s->nacc);
  if (s->scal) (void)hipFree(s->scal);
  if (s->x[0]) (void)hipFree(s->x[0]);
  if (s->x[1]) (void)hipFree(s->x[1]);
  if (s->s1) (void)hipFree(s->s1);
  if (s->s2) (void)hipFree(s->s2);
  if (s->packed) (void)hipFree(s->packed);
  for (hipEvent_t e : s->ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->gev) (void)hipEventDestroy(e);
  for (hipGraphExec_t g : s->gexec) if (g) (void)hipGraphExecDestroy(g);
  if (s->gside) (void)hipStreamDestroy(s->gside);
  if (s->gmain) (void)hipStreamDestroy(s->gmain);
  if (s->iter_dev) (void)hipFree(s->iter_dev);
  if (s->side) { (void)hipStreamSynchronize(s->side); (void)hipStreamDestroy(s->side); }
  if (s->ev_step) (void)hipEventDestroy(s->ev_step);
  for (hipEvent_t e : s->ev_mom) if (e) (void)hipEventDestroy(e);
  for (hipEvent_t e : s->ev_pair) if (e) (void)hipEventDestroy(e);
  delete s;
}

int lmc_sampler_set_state(lmc_sampler* s, const float* x_dev, void* stream) {
  if (!s || !x_dev) return fail(LMC_E_INVALID, "NULL argument");
  DeviceGuard dg(s->device);
  const size_t nbytes = sizeof(float) * (size_t)s->C * s->prob.H * s->prob.W;
  HIP_TRY(hipMemcpyAsync(s->x[s->cur], x_dev, nbytes, hipMemcpyDeviceToDevice, S(stream)));
  if (s->kind == 1) HIP_TRY(hipMemcpyAsync(s->xhat, x_dev, nbytes, hipMemcpyDeviceToDevice, S(stream)));   // xhat = x (algs.py:426)
  if (s->tvwarm[s->wcur]) HIP_TRY(hipMemsetAsync(s->tvwarm[s->wcur], 0, 2 * nbytes, S(stream)));           // a new start: zero dual
  s->mala_fresh = false;
  return LMC_OK;
}

int lmc_sampler_get_state(lmc_sampler* s, float* x_dev, void* stream) {
  if (!s || !x_dev) return fail(LMC_E_INVALID, "NULL argument");
  DeviceGuard dg(s->device);
  const size_t nbytes = sizeof(float) * (size_t)s->C * s->prob.H * s->prob.W;
  HIP_TRY(hipMemcpyAsync(x_dev, s->x[s->cur], nbytes, hipMemcpyDeviceToDevice, S(stream)));
  return LMC_OK;
}

static int ulpda_step(lmc_sampler* s, int32_t n_iters, const float* noise_dev, hipStream_t st);
static int mymala_step(lmc_sampler* s, int32_t n_iters, const float* noise_dev, hipStream_t st);

// f(x_c), g(x_c) of `x` ([C][H][W]) with the sampler's problem and scratch buffers
static int sampler_energies_at(lmc_sampler* s, const float* x, double* f_out_dev, double* g_out_dev, hipStream_t st) {
  HIP_TRY(lmc::launch_energies(x, s->C, energy_args(s->prob), f_out_dev, g_out_dev, st));
  if (s->prob.prior_kind == LMC_PRIOR_HAAR_L1 && g_out_dev)
    HIP_TRY(lmc::launch_haar_value(x, s->C, s->prob.H, s->prob.W, s->prob.prior_sigma, g_out_dev, st));
  if (s->prob.ncvx_kind == LMC_NCVX_ME_TV && f_out_dev) {
    HIP_TRY(g_scratch.need_dbl(3 * (size_t)s->C + 2));
    int rc = me_tv_energy(s->prob, x, s->C, f_out_dev, s->extra, s->tvstate[0], s->tvstate[1], g_scratch.dbl, st, &s->rt_me);
    if (rc) return rc;
  }
  return LMC_OK;
}

// out = base update of `x_in` with the sampler's coefficients; noise_scale 0 gives the proposal mean m(x_in)
static int sampler_update(lmc_sampler* s, const float* x_in, float* x_out, bool with_noise, const float* noise, uint32_t iteration,
                          hipStream_t st, const char** kname, double* f_out = nullptr, double* g_out = nullptr, bool* fused = nullptr) {
  lmc::StepArgs A = s->base;
  if (fused) *fused = false;
  if (f_out && g_out && (variant_of(s->prob) == 0 || variant_of(s->prob) == 7) && s->prob.ncvx_kind == LMC_NCVX_NONE && s->prob.prior_kind == LMC_PRIOR_TV_ISO) {
    lmc::StepArgs probe = A;
    probe.x_in = x_in;
    if (lmc::pipe_supported(probe)) {   // the pipe kernel returns f(x_in), g(x_in) as by-products
      HIP_TRY(hipMemsetAsync(f_out, 0, sizeof(double) * s->C, st));
      HIP_TRY(hipMemsetAsync(g_out, 0, sizeof(double) * s->C, st));
      A.f_out = f_out; A.g_out = g_out; A.g_scale = s->prob.prior_sigma;
      if (fused) *fused = true;
    }
  }
  A.x_in = x_in;
  A.x_out = x_out;
  A.iteration = iteration;
  A.noise = noise;
  if (!with_noise) { A.s = 0.f; A.noise_mode = LMC_NOISE_NONE; A.noise = nullptr; }
  sanitize_pointers(A);
  if (s->prob.ncvx_kind == LMC_NCVX_ME_TV) {   // inner prox of the Moreau-envelope term, then the fused step
    int rc = me_tv_prox(s->prob, A.x_in, s->extra, s->C, s->tvstate[0], s->tvstate[1], st, &s->rt_me);
    if (rc) return rc;
    A.extra = s->extra;
    A.extra_coef = -s->prob.ncvx_lambda / s->prob.ncvx_gamma;
  }
  hipError_t e = launch_step(A, variant_of(s->prob), st, kname, s->tvstate[0], s->tvstate[1], s->pxbuf);
  if (e == hipErrorInvalidConfiguration) return fail(LMC_E_UNSUPPORTED, "no step-kernel variant covers this configuration");
  HIP_TRY(e);
  return LMC_OK;
}


// ---- hipGraph replay of MYULA iterations (small configurations) -------------------------------------------------------------------
constexpr int kGraphIters = 8;     // iterations per graph launch (even: the ping-pong buffers are back in place)

static bool graph_wanted(const lmc_sampler* s) {
  // Opt-in (lmc_problem.graph_replay / LMC_GRAPH=1, fixed when the sampler is created).  Measured on ROCm 7.2 / MI355X at BASELINE config 2 (256 x 256 x 128, rows kernel 25 us + reduction 15.6 us): plain
  // launches 40.2 us per iteration -- the queue is never empty, there are no launch gaps to recover -- graph replay 42.0 us (the side branch
  // does not run under the next step kernel); what does help is the reduction on a second HIP stream (LMC_MOMENTS_OVERLAP, default for small
  // configurations): 37.1 us.  Kept because the replay is exact (tests/test_gpu_graph.py) and may pay on another runtime.
  return s->pol_graph != 0;
}

// Captures kGraphIters iterations starting from buffer s->cur into an executable graph.
static int build_graph(lmc_sampler* s) {
  if (!s->iter_dev) {
    HIP_TRY(hipMalloc(&s->iter_dev, sizeof(uint32_t)));
    HIP_TRY(hipStreamCreateWithFlags(&s->gmain, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&s->gside, hipStreamNonBlocking));
    s->gev.resize(2 * kGraphIters);
    for (hipEvent_t& e : s->gev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  }
  const bool mom = s->moments != 0;
  int cur = s->cur;
  hipStream_t st = s->gmain;           // nothing runs here: the launches below are recorded, the graph is replayed on the caller's stream
  HIP_TRY(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
  int rc = LMC_OK;
  hipError_t e = hipSuccess;
  for (int k = 0; k < kGraphIters && e == hipSuccess; ++k) {
    lmc::StepArgs A = s->base;
    A.x_in = s->x[cur];
    A.x_out = s->x[cur ^ 1];
    A.iteration = (uint32_t)k;
    A.iter_dev = s->iter_dev;
    A.noise = nullptr;
    sanitize_pointers(A);
    if (mom && k >= 2) e = hipStreamWaitEvent(st, s->gev[2 * (k - 2) + 1], 0);     // this step overwrites what reduction k - 2 reads
    const char* kname = nullptr;
    if (e == hipSuccess) e = launch_step(A, variant_of(s->prob), st, &kname, s->tvstate[0], s->tvstate[1], s->pxbuf);
    cur ^= 1;
    if (mom && e == hipSuccess) {       // reduction of the new state on the side branch, under the next step kernel
      e = hipEventRecord(s->gev[2 * k], st);
      if (e == hipSuccess) e = hipStreamWaitEvent(s->gside, s->gev[2 * k], 0);
      if (e == hipSuccess) e = lmc::launch_moments(s->x[cur], s->C, s->prob.H, s->prob.W, s->s1, s->s2, s->gside);
      if (e == hipSuccess) e = hipEventRecord(s->gev[2 * k + 1], s->gside);
    }
  }
  if (mom && e == hipSuccess) {         // join: the last two reductions (the earlier ones were waited for by later steps)
    for (int k = kGraphIters - 2; k < kGraphIters && e == hipSuccess; ++k) e = hipStreamWaitEvent(st, s->gev[2 * k + 1], 0);
  }
  if (e == hipSuccess) e = lmc::launch_bump_u32(s->iter_dev, (uint32_t)kGraphIters, st);
  hipGraph_t graph = nullptr;
  const hipError_t e2 = hipStreamEndCapture(st, &graph);       // always end the capture, also after a failed launch
  if (e != hipSuccess || e2 != hipSuccess) {
    if (graph) (void)hipGraphDestroy(graph);
    rc = fail(LMC_E_HIP, "graph capture failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    return rc;
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  (void)hipGraphDestroy(graph);
  if (e != hipSuccess) return fail(LMC_E_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  s->gexec[s->cur] = exec;
  return LMC_OK;
}

int lmc_sampler_step(lmc_sampler* s, int32_t n_iters, const float* noise_dev, void* stream) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  DeviceGuard dg(s->device);
  if (s->kind == 2) {
    if (n_iters < 0) return fail(LMC_E_INVALID, "n_iters < 0");
    if (s->noise_mode == LMC_NOISE_INJECTED && !noise_dev && n_iters > 0)
      return fail(LMC_E_INVALID, "noise_mode is INJECTED but noise_dev is NULL");
    if (s->noise_mode != LMC_NOISE_INJECTED && noise_dev)
      return fail(LMC_E_INVALID, "noise_dev given but noise_mode is not INJECTED");
    if (s->iteration + n_iters > 0xFFFFFFFFLL) return fail(LMC_E_STATE, "iteration counter would exceed 32 bits");
    return mymala_step(s, n_iters, noise_dev, S(stream));
  }
  if (s->kind == 1) {
    if (n_iters < 0) return fail(LMC_E_INVALID, "n_iters < 0");
    if (s->noise_mode == LMC_NOISE_INJECTED && !noise_dev && n_iters > 0)
      return fail(LMC_E_INVALID, "noise_mode is INJECTED but noise_dev is NULL");
    if (s->noise_mode != LMC_NOISE_INJECTED && noise_dev)
      return fail(LMC_E_INVALID, "noise_dev given but noise_mode is not INJECTED");
    return ulpda_step(s, n_iters, noise_dev, S(stream));
  }
  if (n_iters < 0) return fail(LMC_E_INVALID, "n_iters < 0");
  if (s->noise_mode == LMC_NOISE_INJECTED && !noise_dev && n_iters > 0)
    return fail(LMC_E_INVALID, "noise_mode is INJECTED but noise_dev is NULL");
  if (s->noise_mode != LMC_NOISE_INJECTED && noise_dev)
    return fail(LMC_E_INVALID, "noise_dev given but noise_mode is not INJECTED");
  if (s->iteration + n_iters > 0xFFFFFFFFLL) return fail(LMC_E_STATE, "iteration counter would exceed 32 bits");
  hipStream_t st = S(stream);
  const size_t per_iter = (size_t)s->C * s->prob.H * s->prob.W;
  s->timed = false;
  s->last_launches = 0;
  if (n_iters == 0) return LMC_OK;
  if (s->timing) {  // one event pair per step-kernel launch: moment reductions stay outside the brackets
    while ((int)s->ev.size() < 2 * n_iters) {
      hipEvent_t e;
      HIP_TRY(hipEventCreate(&e));
      s->ev.push_back(e);
    }
  }
  // LMC_MOMENTS_OVERLAP=1: the moment reductions run on a side stream under the following step kernel (0: serially on the caller's stream).
  // Default: on for small configurations (<= 32 Mi pixel-updates per iteration; BASELINE config 2 at 256 x 256 x 128: the 15 us reduction is
  // 40 % of a serial iteration, 40.2 -> 35.9 us per iteration with 128 background workgroups), off for large ones: at the headline size the
  // reduction under the step kernel costs that kernel 6 % (1.76 -> 1.89 ms per launch) and saves its own 0.22 ms -- 1.976 -> 1.90-1.92 ms per
  // iteration with 256 workgroups (16: 3.19, 64: 2.07, 128: 1.92, 256: 1.90, 512: 1.94, 1024: 1.98 ms; too few and the reduction outlasts the
  // step kernel) -- a 3.5 % gain that is left opt-in so that the step kernel's launch time in bench.py / profiles/ is that of the kernel alone.
  // (policy: lmc_problem.moments_overlap / moments_bg_workgroups, LMC_MOMENTS_OVERLAP / LMC_MOMENTS_BG_WGS as defaults; fixed at creation)
  // Round 3: on by default at every size -- what bench.py's `value` measures -- and off while the launches are being event-timed
  // (lmc_sampler_enable_timing: the roofline leg wants the step kernel alone).
  const bool want_overlap = !s->timing && (s->pol_overlap ? s->pol_overlap > 0 : true);
  const int bg_wgs = s->pol_bg_wgs >= 0 ? s->pol_bg_wgs : ((long long)s->C * s->prob.H * s->prob.W <= (1LL << 25) ? 128 : 256);   // 0: the full-speed kernel
  bool overlap = want_overlap && s->moments && n_iters > 1;
  if (overlap && !s->side) {
    int prio_least = 0, prio_greatest = 0;     // lowest priority: the step kernel's workgroups go first
    HIP_TRY(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    const bool low_prio = s->pol_side_lowprio != 0;
    if (low_prio) HIP_TRY(hipStreamCreateWithPriority(&s->side, hipStreamNonBlocking, prio_least));
    else HIP_TRY(hipStreamCreateWithFlags(&s->side, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&s->ev_step, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&s->ev_mom[i], hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&s->ev_pair[i], hipEventDisableTiming));
  }
  // the same for launches that advance two iterations: both iterates are reduced on the side stream under the NEXT pair launch
  auto pair_wait = [&](int p) -> int {        // the pair launch about to be enqueued writes where launch n - 2's reductions read
    if (s->pair_pending[p]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_pair[p], 0)); s->pair_pending[p] = false; }
    return LMC_OK;
  };
  auto pair_reduce = [&](int p, const float* mid, const float* outp) -> int {   // after the pair launch: its kept iterates, on the side stream
    HIP_TRY(hipEventRecord(s->ev_step, st));
    HIP_TRY(hipStreamWaitEvent(s->side, s->ev_step, 0));
    if (mid) HIP_TRY(lmc::launch_moments_bg(mid, s->C, s->prob.H, s->prob.W, s->s1, s->s2, bg_wgs, s->side));
    if (outp) HIP_TRY(lmc::launch_moments_bg(outp, s->C, s->prob.H, s->prob.W, s->s1, s->s2, bg_wgs, s->side));
    HIP_TRY(hipEventRecord(s->ev_pair[p], s->side));
    s->pair_pending[p] = true;
    return LMC_OK;
  };
  // graph replay: whole blocks of kGraphIters iterations whose every iteration is kept by the moment accumulators (or none is)
  const bool graph_ok = !s->timing && (!overlap || s->pol_graph) && !noise_dev && s->noise_mode != LMC_NOISE_INJECTED && s->prob.ncvx_kind != LMC_NCVX_ME_TV &&
                        !s->tvwarm[0] && !s->rtmp && !s->rt_tv.kc && (!s->moments || s->thin == 1) && graph_wanted(s);
  bool graph_enabled = false;
  const int pair_mode = s->pol_pair;
  const bool blockpair_on = s->pol_blockpair != 0;
  for (int k = 0; k < n_iters; ++k) {
    if (graph_ok && s->plain_done && n_iters - k >= kGraphIters && (!s->moments || s->iteration >= s->burn_in) &&
        (s->kernel_name == "myula_step_rows_kernel" || s->kernel_name == "myula_step_block_kernel" || s->kernel_name == "myula_step_pipe_kernel")) {
      if (!s->gexec[s->cur]) { int rc = build_graph(s); if (rc) return rc; }
      if (!graph_enabled) {             // the device-side iteration base of this call
        HIP_TRY(hipMemsetD32Async(reinterpret_cast<hipDeviceptr_t>(s->iter_dev), (int)(uint32_t)s->iteration, 1, st));
        graph_enabled = true;
      }
      for (int i = 0; i < 2; ++i)         // reductions still running on the side stream read buffers the replay overwrites
        if (s->mom_pending[i]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_mom[i], 0)); s->mom_pending[i] = false; }
      HIP_TRY(hipGraphLaunch(s->gexec[s->cur], st));
      s->iteration += kGraphIters;
      if (s->moments) s->count += (uint64_t)s->C * kGraphIters;
      s->last_launches += kGraphIters;
      k += kGraphIters - 1;
      continue;
    }
    if (graph_enabled) {   // plain launches after graph replays take their iteration word by value again: nothing to do (iter_dev is unused)
    }
    // Two MYULA iterations per launch (lmc_step_rows_pair.hip) where that kernel covers the configuration and the launch is large enough for its
    // long bands: x_{k+2} goes to a third array (neighbouring bands re-read x_k), x_{k+1} is stored only when the moment accumulators keep it.
    // lmc_problem.iterations_per_launch / LMC_ROWS_PAIR: 0 = never, 2 = wherever covered (tests), default = where it pays (n_chains * H >= 2^17).
    if (pair_mode && n_iters - k >= 2 && !noise_dev && !graph_ok && !s->tvwarm[0] && !s->rtmp && s->prob.ncvx_kind == LMC_NCVX_NONE &&
        (variant_of(s->prob) == 0 || variant_of(s->prob) == 6) && (pair_mode == 2 || (long long)s->C * s->prob.H >= (1 << 17))) {
      lmc::StepArgs A = s->base;
      A.x_in = s->x[s->cur];
      A.iteration = (uint32_t)s->iteration;
      A.noise = nullptr;
      sanitize_pointers(A);
      if (lmc::rows_pair_supported(A)) {
        if (!s->xspare) {
          HIP_TRY(hipMalloc(&s->xspare, sizeof(float) * per_iter));
        }
        auto kept = [&](int64_t it) { return s->moments && it >= s->burn_in && (it - s->burn_in) % s->thin == 0; };
        const bool keep_mid = kept(s->iteration), keep_out = kept(s->iteration + 1);
        const int pp = (int)(s->pair_n & 1);
        float* mid = keep_mid ? s->x[s->cur ^ 1] : nullptr;
        if (overlap && (keep_mid || keep_out)) {     // the reductions of this launch run under the next one: the iterate in between gets an array of its own
          if (keep_mid) {
            if (!s->xmid[pp]) HIP_TRY(hipMalloc(&s->xmid[pp], sizeof(float) * per_iter));
            mid = s->xmid[pp];
          }
          int rc = pair_wait(pp);
          if (rc) return rc;
        }
        for (int i = 0; i < 2; ++i)                 // (single launches of this call that ran before: their side-stream reductions read x[0] / x[1])
          if (s->mom_pending[i]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_mom[i], 0)); s->mom_pending[i] = false; }
        A.x_out = s->xspare;
        if (s->timing) HIP_TRY(hipEventRecord(s->ev[2 * s->last_launches], st));
        HIP_TRY(lmc::launch_step_rows_pair(A, mid, st));
        if (s->timing) HIP_TRY(hipEventRecord(s->ev[2 * s->last_launches + 1], st));
        s->kernel_name = "myula_step_rows_pair_kernel";
        s->plain_done = true;
        if (overlap && (keep_mid || keep_out)) {
          int rc = pair_reduce(pp, keep_mid ? mid : nullptr, keep_out ? s->xspare : nullptr);
          if (rc) return rc;
          s->count += (uint64_t)s->C * ((keep_mid ? 1 : 0) + (keep_out ? 1 : 0));
        } else {
          if (keep_mid) { HIP_TRY(lmc::launch_moments(mid, s->C, s->prob.H, s->prob.W, s->s1, s->s2, st)); s->count += (uint64_t)s->C; }
          if (keep_out) { HIP_TRY(lmc::launch_moments(s->xspare, s->C, s->prob.H, s->prob.W, s->s1, s->s2, st)); s->count += (uint64_t)s->C; }
        }
        ++s->pair_n;
        std::swap(s->x[s->cur], s->xspare);        // x[cur] = x_{k+2}; the array that held x_k is the spare now
        for (int i = 0; i < 2; ++i)                // captured graphs hold the old pointers
          if (s->gexec[i]) { (void)hipGraphExecDestroy(s->gexec[i]); s->gexec[i] = nullptr; }
        s->iteration += 2;
        ++s->last_launches;
        ++k;
        continue;
      }
    }
    // Two iterations per launch on the register-block kernel (Haar prior, stencil-free data term: BASELINE config 5): the update never leaves a
    // thread's 8 x 8 block, so the second iteration runs on the block while it is on chip; x_{k+1} is written (in place, over x_k) only when the
    // moment accumulators keep it.  LMC_BLOCK_PAIR=0 turns it off.  Same arithmetic, same noise: bit-identical to two launches.
    if (blockpair_on && n_iters - k >= 2 && !noise_dev && !graph_ok && !s->tvwarm[0] && !s->rtmp && s->prob.ncvx_kind == LMC_NCVX_NONE &&
        (variant_of(s->prob) == 0 || variant_of(s->prob) == 5)) {
      lmc::StepArgs A = s->base;
      A.x_in = s->x[s->cur];
      A.x_out = s->x[s->cur ^ 1];
      A.iteration = (uint32_t)s->iteration;
      A.noise = nullptr;
      sanitize_pointers(A);
      if (lmc::block_pair_supported(A)) {
        auto kept = [&](int64_t it) { return s->moments && it >= s->burn_in && (it - s->burn_in) % s->thin == 0; };
        // four iterations on chip when none of the three iterates in between is kept (moments off, burn-in, thinning by >= 4)
        const bool four = n_iters - k >= 4 && !kept(s->iteration) && !kept(s->iteration + 1) && !kept(s->iteration + 2);
        const int nf = four ? 4 : 2;
        const bool keep_mid = !four && kept(s->iteration), keep_out = kept(s->iteration + nf - 1);
        A.fused_iters = nf;
        A.x_mid = keep_mid ? s->x[s->cur] : nullptr;          // in place over x_k (the update is block-local) ...
        const int pp = (int)(s->pair_n & 1);
        const bool side_red = overlap && (keep_mid || keep_out);
        if (side_red) {                                       // ... unless its reduction runs under the next launch, which writes x_{k+3} there
          if (keep_mid) {
            if (!s->xmid[pp]) HIP_TRY(hipMalloc(&s->xmid[pp], sizeof(float) * per_iter));
            A.x_mid = s->xmid[pp];
          }
          int rc = pair_wait(pp);
          if (rc) return rc;
        }
        for (int i = 0; i < 2; ++i)
          if (s->mom_pending[i]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_mom[i], 0)); s->mom_pending[i] = false; }
        if (s->timing) HIP_TRY(hipEventRecord(s->ev[2 * s->last_launches], st));
        HIP_TRY(lmc::launch_step_block(A, st));
        if (s->timing) HIP_TRY(hipEventRecord(s->ev[2 * s->last_launches + 1], st));
        s->kernel_name = four ? "myula_step_block_kernel(4 iterations)" : "myula_step_block_kernel(2 iterations)";
        s->plain_done = true;
        if (side_red) {
          int rc = pair_reduce(pp, keep_mid ? A.x_mid : nullptr, keep_out ? s->x[s->cur ^ 1] : nullptr);
          if (rc) return rc;
          s->count += (uint64_t)s->C * ((keep_mid ? 1 : 0) + (keep_out ? 1 : 0));
          s->cur ^= 1;
        } else {
          if (keep_mid) { HIP_TRY(lmc::launch_moments(s->x[s->cur], s->C, s->prob.H, s->prob.W, s->s1, s->s2, st)); s->count += (uint64_t)s->C; }
          s->cur ^= 1;
          if (keep_out) { HIP_TRY(lmc::launch_moments(s->x[s->cur], s->C, s->prob.H, s->prob.W, s->s1, s->s2, st)); s->count += (uint64_t)s->C; }
        }
        ++s->pair_n;
        s->iteration += nf;
        ++s->last_launches;
        k += nf - 1;
        continue;
      }
    }
    lmc::StepArgs A = s->base;
    A.x_in = s->x[s->cur];
    A.x_out = s->x[s->cur ^ 1];
    A.iteration = (uint32_t)s->iteration;
    A.noise = noise_dev ? noise_dev + (size_t)k * per_iter : nullptr;
    sanitize_pointers(A);
    if (s->prob.ncvx_kind == LMC_NCVX_ME_TV) {   // inner prox of the Moreau-envelope term, then the fused step
      int rc = me_tv_prox(s->prob, A.x_in, s->extra, s->C, s->tvstate[0], s->tvstate[1], st, &s->rt_me);
      if (rc) return rc;
      A.extra = s->extra;
      A.extra_coef = -s->prob.ncvx_lambda / s->prob.ncvx_gamma;
    }
    for (int i = 0; i < 2; ++i)          // reductions of earlier pair launches of this call may still read the array this step writes
      if (s->pair_pending[i]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_pair[i], 0)); s->pair_pending[i] = false; }
    if (s->mom_pending[s->cur ^ 1]) {   // this step overwrites x[cur ^ 1]: the reduction that still reads it must be done
      HIP_TRY(hipStreamWaitEvent(st, s->ev_mom[s->cur ^ 1], 0));
      s->mom_pending[s->cur ^ 1] = false;
    }
    if (s->timing) HIP_TRY(hipEventRecord(s->ev[2 * s->last_launches], st));
    const char* kname = nullptr;
    hipError_t e;
    bool stepped = false;
    if (s->rt_tv.kc && A.prior_kind == LMC_PRIOR_TV_ISO) {   // early exit of the TV prox decided on the device: inside the fused launch, or the prox alone first
      const int rc = tv_prior_rt(s->prob, s->epsg * s->gamma, A, s->rt_tv, s->pxbuf, s->tvstate[0], s->tvstate[1], st);
      if (rc < 0) return rc;
      if (rc == 2) return fail(LMC_E_STATE, "the device-side early exit no longer covers this sampler");
      if (rc == 1) { stepped = true; kname = "myula_step_pipe_kernel(per-chain exit)"; }
    }
    if (s->rtmp && A.prior_kind == LMC_PRIOR_TV_ISO) {   // early-exit TV prox first (exact pass-by-pass path), consumed as a ready-made prox
      int rc = tv_prox_rtol(s->prob, s->epsg * s->gamma, A.x_in, s->pxbuf, s->rtmp, s->robj, s->rflag, s->C, s->tvstate[0], s->tvstate[1], st);
      if (rc) return rc;
      A.prior_kind = LMC_PRIOR_NONE;
      A.prox_ext = s->pxbuf;
    }
    if (s->prob.prox_scale && A.prior_kind != LMC_PRIOR_NONE) {   // array-valued epsg: prox_{epsg[c,i] gamma g} first, consumed as a ready-made prox
      const Problem& q = s->prob;
      HIP_TRY(lmc::launch_prior_prox_scaled(q.prior_kind, q.eprox_kind, A.x_in, s->pxbuf, s->C, (int64_t)q.H * q.W, q.prox_scale, q.prox_scale_cs, q.prox_scale_ps,
                                            s->epsg * s->gamma, q.prior_sigma, q.eprox_p0, q.eprox_p1, q.eprox_mask, st));
      A.prior_kind = LMC_PRIOR_NONE;
      A.prox_ext = s->pxbuf;
    }
    if (stepped) {
      e = hipSuccess;
    } else if (s->tvwarm[0]) {     // warm-started TV prox: the dual of the previous iteration in, this iteration's out
      A.tv_in = s->tvwarm[s->wcur];
      A.tv_out = s->tvwarm[s->wcur ^ 1];
      e = lmc::launch_step_pipe_warm(A, st);
      kname = "myula_step_pipe_kernel(warm)";
      s->wcur ^= 1;
    } else {
      e = launch_step(A, variant_of(s->prob), st, &kname, s->tvstate[0], s->tvstate[1], s->pxbuf);
    }
    if (e == hipErrorInvalidConfiguration) return fail(LMC_E_UNSUPPORTED, "no step-kernel variant covers this configuration");
    HIP_TRY(e);
    if (kname) s->kernel_name = kname;
    s->plain_done = true;
    if (s->timing) HIP_TRY(hipEventRecord(s->ev[2 * s->last_launches + 1], st));
    s->cur ^= 1;
    if (s->moments && s->iteration >= s->burn_in && (s->iteration - s->burn_in) % s->thin == 0) {
      // ... unless the step kernel is a single launch of an HBM-heavy closed-form-prior kernel at full size: the two then share the memory system and the
      // reduction outlasts the kernel whatever its workgroup count (blur + l2, 512 x 512 x 1024: in line 0.714 ms per iteration, beside it 0.73-0.83)
      const bool beside = overlap && (s->pol_overlap > 0 || s->base.prior_kind == LMC_PRIOR_TV_ISO || (long long)per_iter <= (1LL << 25));
      if (beside && k + 1 < n_iters) {   // reduce x[cur] on the side stream while the next step kernel runs
        HIP_TRY(hipEventRecord(s->ev_step, st));
        HIP_TRY(hipStreamWaitEvent(s->side, s->ev_step, 0));
        HIP_TRY(lmc::launch_moments_bg(s->x[s->cur], s->C, s->prob.H, s->prob.W, s->s1, s->s2, bg_wgs, s->side));
        HIP_TRY(hipEventRecord(s->ev_mom[s->cur], s->side));
        s->mom_pending[s->cur] = true;
      } else {
        for (int i = 0; i < 2; ++i)       // accumulators are shared: stay behind the side stream's reductions
          if (s->mom_pending[i]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_mom[i], 0)); s->mom_pending[i] = false; }
        HIP_TRY(lmc::launch_moments(s->x[s->cur], s->C, s->prob.H, s->prob.W, s->s1, s->s2, st));
      }
      s->count += (uint64_t)s->C;
    }
    ++s->iteration;
    ++s->last_launches;
  }
  for (int i = 0; i < 2; ++i) {           // everything this call enqueued is ordered before whatever the caller enqueues next
    if (s->mom_pending[i]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_mom[i], 0)); s->mom_pending[i] = false; }
    if (s->pair_pending[i]) { HIP_TRY(hipStreamWaitEvent(st, s->ev_pair[i], 0)); s->pair_pending[i] = false; }
  }
  s->timed = s->timing;
  return LMC_OK;
}

// ---- MYMALA: Metropolis-adjusted MYULA at image scale (generalises prox_lmc.py:134-158) -------------------------------
int lmc_mymala_create(const lmc_myula_config* cfg, lmc_sampler** out) {
  int rc = lmc_myula_create(cfg, out);
  if (rc) return rc;
  lmc_sampler* s = *out;
  *out = nullptr;
  if (s->tvwarm[0]) { lmc_sampler_destroy(s); return fail(LMC_E_UNSUPPORTED, "MYMALA needs a proposal mean that is a function of x alone: tv_warm is not allowed"); }
  if (s->rtmp || s->rt_tv.kc) { lmc_sampler_destroy(s); return fail(LMC_E_UNSUPPORTED, "MYMALA with tv_rtol > 0 is not built (use the fixed-count prox, tv_rtol = 0)"); }
  if (s->prob.prox_scale) { lmc_sampler_destroy(s); return fail(LMC_E_UNSUPPORTED, "MYMALA takes a scalar epsg (the reference's array-valued epsg is MYULA's, algs.py:509)"); }
  s->kind = 2;
  const size_t nbytes = sizeof(float) * (size_t)s->C * s->prob.H * s->prob.W;
  hipError_t e = hipMalloc(&s->mx, nbytes);
  if (e == hipSuccess) e = hipMalloc(&s->xp, nbytes);
  if (e == hipSuccess) e = hipMalloc(&s->mxp, nbytes);
  if (e == hipSuccess) e = hipMalloc(&s->mala_d, sizeof(double) * 6 * (size_t)s->C);
  if (e == hipSuccess) e = hipMalloc(&s->flag, sizeof(int) * (size_t)s->C);
  if (e == hipSuccess) e = hipMalloc(&s->nacc, sizeof(unsigned long long) * (size_t)s->C);
  if (e == hipSuccess) e = hipMemset(s->nacc, 0, sizeof(unsigned long long) * (size_t)s->C);
  if (e == hipSuccess) e = hipMemset(s->mala_d, 0, sizeof(double) * 6 * (size_t)s->C);
  if (e != hipSuccess) {
    rc = fail(e == hipErrorOutOfMemory ? LMC_E_NOMEM : LMC_E_HIP, "sampler allocation failed: %s", hipGetErrorString(e));
    lmc_sampler_destroy(s);
    return rc;
  }
  *out = s;
  return LMC_OK;
}

static int mymala_step(lmc_sampler* s, int32_t n_iters, const float* noise_dev, hipStream_t st) {
  const size_t img = (size_t)s->prob.H * s->prob.W, per_iter = (size_t)s->C * img;
  const int C = s->C;
  double *U = s->mala_d, *fp = U + C, *gp = U + 2 * C, *d1 = U + 3 * C, *d2 = U + 4 * C, *la = U + 5 * C;
  float* x = s->x[s->cur];
  s->timed = false;
  s->last_launches = 0;
  const char* kname = nullptr;
  if (!s->mala_fresh && n_iters > 0) {   // m(x) and U(x) = f(x) + g(x) of the current state (after create / set_state)
    bool fused = false;
    int rc = sampler_update(s, x, s->mx, false, nullptr, (uint32_t)s->iteration, st, &kname, fp, gp, &fused);
    if (rc) return rc;
    if (!fused) rc = sampler_energies_at(s, x, fp, gp, st);
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(d1, 0, sizeof(double) * C, st));
    HIP_TRY(lmc::launch_axpy_env(fp, gp, d1, C, -s->epsg, 1.f, st));    // fp += epsg * gp  (f -= (-epsg) * (g + d1/2) with d1 = 0): U = f + epsg g, the
                                                                        // potential MYULA's drift is built from (algs.py:569, 582)
    HIP_TRY(hipMemcpyAsync(U, fp, sizeof(double) * C, hipMemcpyDeviceToDevice, st));
    s->mala_fresh = true;
  }
  for (int k = 0; k < n_iters; ++k) {
    const float* xi = noise_dev ? noise_dev + (size_t)k * per_iter : s->xi;
    if (s->noise_mode == LMC_NOISE_PHILOX) {   // x' = m(x) + s xi with the Philox field drawn inside the proposal kernel, and ||x' - m(x)||^2
      HIP_TRY(lmc::mala_propose_philox(s->mx, s->xp, C, s->prob.H, s->prob.W, s->base.s, s->base.key0, s->base.key1, (uint32_t)s->iteration,
                                       s->base.chain_offset, d1, st));
    } else if (s->noise_mode == LMC_NOISE_NONE) {   // deterministic proposal x' = m(x): d1 = 0
      HIP_TRY(hipMemcpyAsync(s->xp, s->mx, sizeof(float) * per_iter, hipMemcpyDeviceToDevice, st));
      HIP_TRY(hipMemsetAsync(d1, 0, sizeof(double) * C, st));
    } else {
      HIP_TRY(lmc::mala_propose(s->mx, xi, s->xp, C, img, s->base.s, d1, st));          // x' and ||x' - m(x)||^2
    }
    bool fused = false;
    int rc = sampler_update(s, s->xp, s->mxp, false, nullptr, (uint32_t)s->iteration, st, &kname, fp, gp, &fused);   // m(x') [+ f, g]
    if (rc) return rc;
    if (!fused) rc = sampler_energies_at(s, s->xp, fp, gp, st);                           // f(x'), g(x')
    if (rc) return rc;
    HIP_TRY(lmc::launch_sqdiff(x, s->mxp, C, img, d2, st));                               // ||x - m(x')||^2
    HIP_TRY(lmc::mala_accept(C, U, fp, gp, s->epsg, d1, d2, s->tau, s->base.key0, s->base.key1, (uint32_t)s->iteration, s->base.chain_offset,
                             s->flag, s->nacc, la, st));
    // accepted chains: x <- x', m(x) <- m(x').  (The other direction -- keep the proposal buffers and give the rejected chains their old
    // state back -- was measured: 3.0 instead of 3.7 ms at 98 % acceptance, but 3.7 instead of 3.0 ms at 48 %; the choice would have to
    // follow the acceptance rate, which the host does not see without a synchronisation.)
    HIP_TRY(lmc::mala_select(s->flag, x, s->mx, s->xp, s->mxp, C, img, 1, st));
    if (s->moments && s->iteration >= s->burn_in && (s->iteration - s->burn_in) % s->thin == 0) {
      HIP_TRY(lmc::launch_moments(x, C, s->prob.H, s->prob.W, s->s1, s->s2, st));
      s->count += (uint64_t)C;
    }
    ++s->iteration;
  }
  if (kname) s->kernel_name = kname;
  return LMC_OK;
}

int lmc_sampler_get_acceptance(lmc_sampler* s, uint64_t* accepted_dev, double* last_log_alpha_dev, void* stream) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  DeviceGuard dg(s->device);
  if (s->kind != 2) return fail(LMC_E_STATE, "not a MYMALA sampler");
  if (accepted_dev) HIP_TRY(hipMemcpyAsync(accepted_dev, s->nacc, sizeof(uint64_t) * (size_t)s->C, hipMemcpyDeviceToDevice, S(stream)));
  if (last_log_alpha_dev)
    HIP_TRY(hipMemcpyAsync(last_log_alpha_dev, s->mala_d + 5 * (size_t)s->C, sizeof(double) * (size_t)s->C, hipMemcpyDeviceToDevice, S(stream)));
  return LMC_OK;
}

int lmc_sampler_enable_timing(lmc_sampler* s, int32_t on) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  s->timing = on != 0;
  s->timed = false;
  return LMC_OK;
}

// ---- ULPDA ---------------------------------------------------------------------------------------

static int ulpda_step(lmc_sampler* s, int32_t n_iters, const float* noise_dev, hipStream_t st) {
  const int H = s->prob.H, W = s->prob.W;
  const int64_t C = s->C;
  const size_t per_iter = (size_t)C * H * W;
  const int iso = s->prob.prior_kind == LMC_PRIOR_TV_ISO;
  s->timed = false;
  s->last_launches = 0;
  if (s->iteration + n_iters > 0xFFFFFFFFLL) return fail(LMC_E_STATE, "iteration counter would exceed 32 bits");
  // gfirst = false: finish and dual update as ONE row-streaming pass (x ping-pongs between two buffers, xhat stays in registers; 28 instead
  // of 36 B per pixel).  Opt-in (LMC_ULPDA_FUSE=1): measured at 512 x 512 x 1024 the fused pass takes 1.85 ms (4.1 TB/s; software-pipelined
  // loads: the same) against 0.76 + 0.94 ms for the two flat passes (5.6 TB/s each) -- 8.0 ms per iteration either way.  Exact (test_gpu_ulpda.py).
  static const bool fuse_env = [] { const char* e = getenv("LMC_ULPDA_FUSE"); return e && atoi(e) != 0; }();
  const bool fuse_fd = fuse_env && !s->gfirst && s->x[1] && lmc::ulpda_finish_dual_supported(H, W);
  bool used_pairs = false;
  // LMC_ULPDA_DUAL_RHS=1 (opt-in): the dual update fused with the next iteration's right-hand side (ulpda_dual_rhs4_kernel: 28 instead of 36 B per
  // pixel, bit-identical -- tests/test_gpu_ulpda.py).  Measured at 512 x 512 x 1024 on one box, alternating: 5.61 / 5.61 ms per iteration fused
  // against 5.57 / 5.66 ms with the two passes -- the neighbour recomputation costs what the 8 B save; left off.  Inside one call only
  // (k + 1 < n_iters): between calls the caller may change the steps, the state or the dual.
  const bool fuse_dr = s->pol_ulpda_dual_rhs && !s->gfirst && !fuse_fd && s->ydual2 && s->prob.ncvx_kind == LMC_NCVX_NONE &&
                       lmc::ulpda_dual_rhs_supported(H, W);
  s->rhs_ready = false;
  for (int k = 0; k < n_iters; ++k) {
    float* x = s->x[s->cur];
    const float ts = s->tau * s->prob.sigma_f;
    // pre-step of L2_ncvx_tv.prox: with a blur the right-hand side also takes tau sigma H^T b; the pointwise solves add their tau sigma m b themselves
    const float* htb_nc = s->prob.data_kind == LMC_DATA_BLUR ? s->htb : s->prob.y;
    const float ts_nc = s->prob.data_kind == LMC_DATA_BLUR ? ts : 0.f;
    if (s->gfirst)   // y <- proxdual(y + mu A xhat)   (algs.py:436)
      HIP_TRY(lmc::ulpda_dual_update(s->xhat, s->ydual, C, H, W, s->mu, s->prob.prior_sigma, iso, st));
    // v = x - tau (A^T y + z) [+ tau sigma H^T b]      (algs.py:437-440 / 443-446)
    if (s->prob.ncvx_kind == LMC_NCVX_MC_TV) {   // L2_ncvx_tv.prox pre-step (algs.py:213-217), then + tau sigma H^T b (:225)
      HIP_TRY(lmc::ulpda_rhs(x, s->ydual, s->z, nullptr, s->ctmp, C, H, W, s->tau, ts, st));
      HIP_TRY(lmc::ulpda_ncvx_rhs(s->ctmp, htb_nc, s->rhs, C, H, W, s->tau * s->prob.ncvx_lambda, s->prob.ncvx_gamma, ts_nc, st));
    } else if (s->prob.ncvx_kind == LMC_NCVX_ME_TV) {   // x += tau*lamda/gamma (x - prox_{gamma TV}(x))  (algs.py:221-223)
      HIP_TRY(lmc::ulpda_rhs(x, s->ydual, s->z, nullptr, s->ctmp, C, H, W, s->tau, ts, st));
      int rc = me_tv_prox(s->prob, s->ctmp, s->extra, C, s->tvstate[0], s->tvstate[1], st, &s->rt_me);
      if (rc) return rc;
      HIP_TRY(lmc::ulpda_me_rhs(s->ctmp, s->extra, htb_nc, s->rhs, C, H, W, s->tau * s->prob.ncvx_lambda / s->prob.ncvx_gamma, ts_nc, st));
    } else if (s->rhs_ready && s->rhs_tau == s->tau && s->rhs_ts == ts) {
      // formed together with the previous iteration's dual update (fuse_dr below)
    } else
    HIP_TRY(lmc::ulpda_rhs(x, s->ydual, s->z, s->prob.data_kind == LMC_DATA_BLUR ? s->htb : nullptr, s->rhs, C, H, W, s->tau, ts, st));
    s->rhs_ready = false;
    const float* u = s->rhs;
    if (s->prob.data_kind == LMC_DATA_BLUR) {
      if (!s->warm) HIP_TRY(hipMemsetAsync(s->uw, 0, sizeof(float) * per_iter, st));
      float* where = s->uw;
      { int rc = cg_solve_fused(s->prob, ts, s->uw, s->rhs, s->cr, s->cp, s->cq, s->scal, C, s->cg_niter, s->zero_y, st, s->uw2, &where); if (rc) return rc; }
      if (where != s->uw) { s->uw2 = s->uw; s->uw = where; used_pairs = true; }       // the pair launches deliver the solution in the other array
      u = s->uw;
    } else if (s->prob.data_kind != LMC_DATA_NONE) {
      HIP_TRY(lmc::ulpda_pointwise_prox(s->rhs, s->uw, s->prob.y, s->prob.mask, C, H, W, ts, s->prob.data_kind, st));
      u = s->uw;
    }
    // x <- u + sqrt(2 tau) xi ; xhat <- x + theta (x - x_old)     (algs.py:440-441 / 446-447)
    if (fuse_fd) {       // ... and y <- proxdual(y + mu A xhat) (algs.py:448) in the same pass
      const float* xi = s->noise_mode == LMC_NOISE_INJECTED ? noise_dev + (size_t)k * per_iter : nullptr;
      HIP_TRY(lmc::ulpda_finish_dual(x, s->x[s->cur ^ 1], u, s->ydual, xi, C, H, W, std::sqrt(2.f * s->tau), s->theta, s->mu, s->prob.prior_sigma, iso,
                                     s->noise_mode == LMC_NOISE_PHILOX, s->base.key0, s->base.key1, (uint32_t)s->iteration, s->base.chain_offset, st));
      s->cur ^= 1;
      x = s->x[s->cur];
    } else
    if (s->noise_mode == LMC_NOISE_PHILOX && (W & 3) == 0) {     // the Philox field is drawn inside the pass
      HIP_TRY(lmc::ulpda_finish_philox(x, s->xhat, u, C, H, W, std::sqrt(2.f * s->tau), s->theta, s->base.key0, s->base.key1,
                                       (uint32_t)s->iteration, s->base.chain_offset, st));
    } else {
      const float* xi = nullptr;
      if (s->noise_mode == LMC_NOISE_INJECTED) xi = noise_dev + (size_t)k * per_iter;
      else if (s->noise_mode == LMC_NOISE_PHILOX) {
        HIP_TRY(lmc::launch_noise(s->xi, (int)C, H, W, s->base.key0, s->base.key1, (uint32_t)s->iteration, s->base.chain_offset, st));
        xi = s->xi;
      }
      HIP_TRY(lmc::ulpda_finish(x, s->xhat, u, xi, C, H, W, std::sqrt(2.f * s->tau), s->theta, st));
    }
    if (!s->gfirst && !fuse_fd) {  // (algs.py:448)
      if (fuse_dr && k + 1 < n_iters) {
        // ... fused with the right-hand side of the next iteration (same tau unless lmc_sampler_set_steps intervenes: then it is formed again):
        // 28 instead of 20 + 16 B per pixel, bit-identical to the two passes
        HIP_TRY(lmc::ulpda_dual_rhs(s->xhat, s->ydual, s->ydual2, x, s->z, s->prob.data_kind == LMC_DATA_BLUR ? s->htb : nullptr, s->rhs, C, H, W, s->mu,
                                    s->prob.prior_sigma, iso, s->tau, ts, st));
        std::swap(s->ydual, s->ydual2);
        s->rhs_ready = true; s->rhs_tau = s->tau; s->rhs_ts = ts;
      } else {
        HIP_TRY(lmc::ulpda_dual_update(s->xhat, s->ydual, C, H, W, s->mu, s->prob.prior_sigma, iso, st));
      }
    }
    if (s->moments && s->iteration >= s->burn_in && (s->iteration - s->burn_in) % s->thin == 0) {
      HIP_TRY(lmc::launch_moments(x, s->C, H, W, s->s1, s->s2, st));
      s->count += (uint64_t)s->C;
    }
    ++s->iteration;
  }
  s->kernel_name = used_pairs ? "ulpda (multi-kernel, chebyshev pairs)" : "ulpda (multi-kernel)";
  return LMC_OK;
}

int lmc_ulpda_create(const lmc_ulpda_config* cfg, lmc_sampler** out) {
  if (!cfg || !out) return fail(LMC_E_INVALID, "NULL argument");
  *out = nullptr;
  if (cfg->struct_size != sizeof(lmc_ulpda_config))
    return fail(LMC_E_INVALID, "lmc_ulpda_config.struct_size %u != %zu (ABI mismatch)", cfg->struct_size, sizeof(lmc_ulpda_config));
  if (cfg->n_chains < 1) return fail(LMC_E_INVALID, "n_chains must be >= 1");
  if (cfg->chain_offset < 0 || cfg->chain_offset + cfg->n_chains > 0xFFFFFFFFLL)
    return fail(LMC_E_INVALID, "global chain ids must fit 32 bits");
  if (!(cfg->tau > 0.f) || !(cfg->mu > 0.f)) return fail(LMC_E_INVALID, "tau and mu must be > 0");
  if (cfg->noise_mode < LMC_NOISE_PHILOX || cfg->noise_mode > LMC_NOISE_NONE) return fail(LMC_E_INVALID, "bad noise_mode");
  if (cfg->problem.prior_kind != LMC_PRIOR_TV_ISO && cfg->problem.prior_kind != LMC_PRIOR_TV_ANISO)
    return fail(LMC_E_UNSUPPORTED, "ULPDA needs g o A with g = L21 (LMC_PRIOR_TV_ISO) or L1 (LMC_PRIOR_TV_ANISO)");
  if (!(cfg->problem.prior_sigma > 0.f)) return fail(LMC_E_INVALID, "prior_sigma (dual ball radius) must be > 0");
  if (cfg->problem.data_kind == LMC_DATA_BLUR && cfg->cg_niter < 1) return fail(LMC_E_INVALID, "cg_niter must be >= 1");
  if (cfg->problem.ncvx_kind != LMC_NCVX_NONE && cfg->problem.data_kind == LMC_DATA_NONE)
    return fail(LMC_E_UNSUPPORTED, "the non-convex term belongs to a data term (blur, identity or mask)");
  lmc_sampler* s = new (std::nothrow) lmc_sampler();
  if (!s) return fail(LMC_E_NOMEM, "host allocation failed");
  lmc_problem pr = cfg->problem;
  if (pr.prior_kind == LMC_PRIOR_TV_ISO && pr.tv_niter < 1) pr.tv_niter = 1;   // unused by ULPDA; keeps the loader happy
  int rc = load_problem(&pr, s->prob);
  if (rc) { delete s; return rc; }
  if (hipGetDevice(&s->device) != hipSuccess) { delete s; return fail(LMC_E_HIP, "hipGetDevice failed"); }
  s->kind = 1;
  s->C = cfg->n_chains;
  s->chain_offset = cfg->chain_offset;
  s->tau = cfg->tau; s->mu = cfg->mu; s->theta = cfg->theta;
  s->gfirst = cfg->gfirst != 0; s->cg_niter = cfg->cg_niter; s->warm = cfg->warm != 0;
  s->z = cfg->z_dev;
  s->seed = cfg->seed;
  s->noise_mode = cfg->noise_mode;
  s->moments = cfg->moments; s->burn_in = cfg->burn_in; s->thin = cfg->thin < 1 ? 1 : cfg->thin;
  s->base.key0 = (uint32_t)(s->seed & 0xFFFFFFFFu);
  s->base.key1 = (uint32_t)(s->seed >> 32);
  s->base.chain_offset = (uint32_t)s->chain_offset;
  const size_t n = (size_t)s->C * s->prob.H * s->prob.W, img = (size_t)s->prob.H * s->prob.W;
  hipError_t e = hipSuccess;
  auto alloc = [&](float** p, size_t count) { if (e == hipSuccess) e = hipMalloc(p, sizeof(float) * count); if (e == hipSuccess) e = hipMemset(*p, 0, sizeof(float) * count); };
  alloc(&s->x[0], n); alloc(&s->xhat, n); alloc(&s->ydual, 2 * n); alloc(&s->uw, n); alloc(&s->rhs, n);
  s->pol_ulpda_dual_rhs = env_int("LMC_ULPDA_DUAL_RHS", 0) != 0;     // read once, here
  if (s->pol_ulpda_dual_rhs && !s->gfirst && s->prob.ncvx_kind == LMC_NCVX_NONE && lmc::ulpda_dual_rhs_supported(s->prob.H, s->prob.W)) alloc(&s->ydual2, 2 * n);
  if (!s->gfirst && lmc::ulpda_finish_dual_supported(s->prob.H, s->prob.W) && getenv("LMC_ULPDA_FUSE") && atoi(getenv("LMC_ULPDA_FUSE")))
    alloc(&s->x[1], n);     // ping-pong target of the fused finish + dual pass (opt-in, see ulpda_step)
  if (s->noise_mode == LMC_NOISE_PHILOX) alloc(&s->xi, n);
  if (s->prob.data_kind == LMC_DATA_BLUR) {
    alloc(&s->cr, n); alloc(&s->cp, n); alloc(&s->cq, n); alloc(&s->ctmp, n); alloc(&s->htb, img); alloc(&s->zero_y, img);
    if (lmc::cheb_pair_supported(s->prob.H, s->prob.W, s->prob.taps)) alloc(&s->uw2, n);
    if (e == hipSuccess) e = hipMalloc(&s->scal, sizeof(double) * (4 * (size_t)s->C + 1));
    if (e == hipSuccess) e = lmc::launch_blur(s->prob.y, s->htb, 1, s->prob.H, s->prob.W, s->prob.taps, 1, nullptr);   // H^T b
    if (e == hipSuccess) e = hipDeviceSynchronize();
  }
  if (s->prob.ncvx_kind != LMC_NCVX_NONE && !s->ctmp) alloc(&s->ctmp, n);      // pointwise data terms: the pre-step of L2_ncvx_tv.prox needs its own array
  if (s->prob.ncvx_kind == LMC_NCVX_ME_TV) {
    alloc(&s->extra, n);
    if (needs_tv_state(s->prob)) { alloc(&s->tvstate[0], 4 * n); alloc(&s->tvstate[1], 4 * n); }
    if (e == hipSuccess && s->prob.ncvx_rtol > 0.f) e = s->rt_me.need((size_t)s->C, s->prob.ncvx_niter);     // early exit of the inner prox, decided on the device
  }
  if (e == hipSuccess && s->moments) {
    const size_t mb = sizeof(double) * img;
    e = hipMalloc(&s->s1, mb);
    if (e == hipSuccess) e = hipMalloc(&s->s2, mb);
    if (e == hipSuccess) e = hipMemset(s->s1, 0, mb);
    if (e == hipSuccess) e = hipMemset(s->s2, 0, mb);
  }
  if (e != hipSuccess) {
    rc = fail(e == hipErrorOutOfMemory ? LMC_E_NOMEM : LMC_E_HIP, "sampler allocation failed: %s", hipGetErrorString(e));
    lmc_sampler_destroy(s);
    return rc;
  }
  s->kernel_name = "(no step launched yet)";
  *out = s;
  return LMC_OK;
}

int lmc_sampler_set_dual(lmc_sampler* s, const float* y_dev, void* stream) {
  if (!s || !y_dev) return fail(LMC_E_INVALID, "NULL argument");
  DeviceGuard dg(s->device);
  if (s->kind != 1) return fail(LMC_E_STATE, "not a ULPDA sampler");
  HIP_TRY(hipMemcpyAsync(s->ydual, y_dev, sizeof(float) * 2 * (size_t)s->C * s->prob.H * s->prob.W, hipMemcpyDeviceToDevice, S(stream)));
  return LMC_OK;
}

int lmc_sampler_get_dual(lmc_sampler* s, float* y_dev, void* stream) {
  if (!s || !y_dev) return fail(LMC_E_INVALID, "NULL argument");
  DeviceGuard dg(s->device);
  if (s->kind != 1) return fail(LMC_E_STATE, "not a ULPDA sampler");
  HIP_TRY(hipMemcpyAsync(y_dev, s->ydual, sizeof(float) * 2 * (size_t)s->C * s->prob.H * s->prob.W, hipMemcpyDeviceToDevice, S(stream)));
  return LMC_OK;
}

int lmc_sampler_set_steps(lmc_sampler* s, float tau, float mu) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  if (s->kind != 1) return fail(LMC_E_STATE, "not a ULPDA sampler");
  if (!(tau > 0.f) || !(mu > 0.f)) return fail(LMC_E_INVALID, "tau and mu must be > 0");
  s->tau = tau; s->mu = mu;
  return LMC_OK;
}

int64_t lmc_sampler_iteration(const lmc_sampler* s) { return s ? s->iteration : -1; }

int lmc_sampler_set_iteration(lmc_sampler* s, int64_t it) {
  if (!s || it < 0 || it > 0xFFFFFFFFLL) return fail(LMC_E_INVALID, "bad iteration");
  s->iteration = it;
  return LMC_OK;
}

int lmc_sampler_get_moments(lmc_sampler* s, double* sum_dev, double* sumsq_dev, uint64_t* count, void* stream) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  DeviceGuard dg(s->device);
  if (!s->moments) return fail(LMC_E_STATE, "sampler was created with moments = 0");
  const size_t mb = sizeof(double) * (size_t)s->prob.H * s->prob.W;
  if (sum_dev) HIP_TRY(hipMemcpyAsync(sum_dev, s->s1, mb, hipMemcpyDeviceToDevice, S(stream)));
  if (sumsq_dev) HIP_TRY(hipMemcpyAsync(sumsq_dev, s->s2, mb, hipMemcpyDeviceToDevice, S(stream)));
  HIP_TRY(hipStreamSynchronize(S(stream)));
  if (count) *count = s->count;
  return LMC_OK;
}

int lmc_sampler_reset_moments(lmc_sampler* s, void* stream) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  DeviceGuard dg(s->device);
  if (!s->moments) return fail(LMC_E_STATE, "sampler was created with moments = 0");
  const size_t mb = sizeof(double) * (size_t)s->prob.H * s->prob.W;
  HIP_TRY(hipMemsetAsync(s->s1, 0, mb, S(stream)));
  HIP_TRY(hipMemsetAsync(s->s2, 0, mb, S(stream)));
  s->count = 0;
  return LMC_OK;
}

int lmc_sampler_energies(lmc_sampler* s, double* f_out_dev, double* g_out_dev, void* stream) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  DeviceGuard dg(s->device);
  return sampler_energies_at(s, s->x[s->cur], f_out_dev, g_out_dev, S(stream));
}

int lmc_sampler_noise(lmc_sampler* s, int64_t iteration, float* out_dev, void* stream) {
  if (!s || !out_dev) return fail(LMC_E_INVALID, "NULL argument");
  DeviceGuard dg(s->device);
  if (iteration < 0 || iteration > 0xFFFFFFFFLL) return fail(LMC_E_INVALID, "bad iteration");
  HIP_TRY(lmc::launch_noise(out_dev, s->C, s->prob.H, s->prob.W, s->base.key0, s->base.key1, (uint32_t)iteration,
                            s->base.chain_offset, S(stream)));
  return LMC_OK;
}

int lmc_sampler_last_step_timing(lmc_sampler* s, float* total_ms, int32_t* n_launches) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  DeviceGuard dg(s->device);
  if (!s->timed || s->last_launches < 1)
    return fail(LMC_E_STATE, "no timed lmc_sampler_step call to report (lmc_sampler_enable_timing first)");
  HIP_TRY(hipEventSynchronize(s->ev[2 * s->last_launches - 1]));
  float ms = 0.f;
  for (int k = 0; k < s->last_launches; ++k) {
    float d = 0.f;
    HIP_TRY(hipEventElapsedTime(&d, s->ev[2 * k], s->ev[2 * k + 1]));
    ms += d;
  }
  if (total_ms) *total_ms = ms;
  if (n_launches) *n_launches = s->last_launches;
  return LMC_OK;
}

const char* lmc_sampler_kernel_name(const lmc_sampler* s) { return s ? s->kernel_name.c_str() : ""; }

int lmc_sampler_tv_exit_stats(lmc_sampler* s, int32_t which, int32_t* passes_dev, uint64_t* reruns_host, void* stream) {
  if (!s || (which != 0 && which != 1)) return fail(LMC_E_INVALID, "bad arguments");
  DeviceGuard dg(s->device);
  RtState& rt = which == 0 ? s->rt_tv : s->rt_me;
  if (!rt.kc) return fail(LMC_E_STATE, "this sampler does not run the device-side early exit for that prox (tv_rtol / ncvx_rtol = 0, or the pass-by-pass path)");
  hipStream_t st = S(stream);
  if (passes_dev) HIP_TRY(hipMemcpyAsync(passes_dev, rt.pred, sizeof(int) * (size_t)s->C, hipMemcpyDeviceToDevice, st));
  unsigned long long r[4] = {0, 0, 0, 0};
  if (reruns_host) HIP_TRY(hipMemcpyAsync(r, rt.reruns, sizeof r, hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (reruns_host) for (int i = 0; i < 4; ++i) reruns_host[i] = r[i];
  return LMC_OK;
}

// ---- multi-GPU: the one collective of the path -----------------------------------------------------------------
// RCCL is reached through dlopen so that the library (and every single-GPU use) does not depend on it.  When the host process has
// RCCL loaded already (PyTorch-ROCm ships its own librccl.so.1) that instance is the one bound, so a communicator created by the host
// framework is valid here.
namespace {
struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string why;
};
RcclApi* rccl_api() {
  static RcclApi api = [] {
    RcclApi a;
    const char* env = getenv("LMC_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      a.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);          // the instance the process already has, if any
      if (!a.lib) a.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (a.lib) break;
    }
    if (!a.lib) { const char* e = dlerror(); a.why = std::string("librccl not found (set LMC_RCCL_LIB): ") + (e ? e : ""); return a; }
    auto sym = [&](const char* n) { void* p = dlsym(a.lib, n); if (!p && a.why.empty()) a.why = std::string("librccl lacks ") + n; return p; };
    a.GetUniqueId = reinterpret_cast<decltype(a.GetUniqueId)>(sym("ncclGetUniqueId"));
    a.CommInitRank = reinterpret_cast<decltype(a.CommInitRank)>(sym("ncclCommInitRank"));
    a.CommDestroy = reinterpret_cast<decltype(a.CommDestroy)>(sym("ncclCommDestroy"));
    a.CommCount = reinterpret_cast<decltype(a.CommCount)>(sym("ncclCommCount"));
    a.AllReduce = reinterpret_cast<decltype(a.AllReduce)>(sym("ncclAllReduce"));
    a.GetErrorString = reinterpret_cast<decltype(a.GetErrorString)>(sym("ncclGetErrorString"));
    if (!a.why.empty()) { dlclose(a.lib); a.lib = nullptr; }
    return a;
  }();
  return &api;
}
#define RCCL_TRY(api, expr)                                                                                       \
  do {                                                                                                            \
    ncclResult_t r_ = (expr);                                                                                     \
    if (r_ != ncclSuccess) return fail(LMC_E_HIP, "%s failed: %s", #expr, (api)->GetErrorString(r_));            \
  } while (0)
}  // namespace

int lmc_rccl_available(void) { return rccl_api()->lib ? 1 : 0; }

int lmc_rccl_unique_id(void* id128_host) {
  RcclApi* R = rccl_api();
  if (!R->lib) return fail(LMC_E_UNSUPPORTED, "%s", R->why.c_str());
  if (!id128_host) return fail(LMC_E_INVALID, "NULL argument");
  static_assert(sizeof(ncclUniqueId) == LMC_RCCL_UNIQUE_ID_BYTES, "ncclUniqueId size");
  ncclUniqueId id;
  RCCL_TRY(R, R->GetUniqueId(&id));
  std::memcpy(id128_host, &id, sizeof id);
  return LMC_OK;
}

int lmc_rccl_comm_create(void** comm_out, int32_t world, int32_t rank, const void* id128_host) {
  RcclApi* R = rccl_api();
  if (!R->lib) return fail(LMC_E_UNSUPPORTED, "%s", R->why.c_str());
  if (!comm_out || !id128_host) return fail(LMC_E_INVALID, "NULL argument");
  if (world < 1 || rank < 0 || rank >= world) return fail(LMC_E_INVALID, "bad rank %d of %d", rank, world);
  ncclUniqueId id;
  std::memcpy(&id, id128_host, sizeof id);
  ncclComm_t comm = nullptr;
  RCCL_TRY(R, R->CommInitRank(&comm, world, id, rank));      // on the current device: one process per GPU
  *comm_out = comm;
  return LMC_OK;
}

int lmc_rccl_comm_destroy(void* comm) {
  RcclApi* R = rccl_api();
  if (!R->lib) return fail(LMC_E_UNSUPPORTED, "%s", R->why.c_str());
  if (!comm) return LMC_OK;
  RCCL_TRY(R, R->CommDestroy(static_cast<ncclComm_t>(comm)));
  return LMC_OK;
}

int lmc_allreduce_moments(lmc_sampler* s, void* rccl_comm, double* sum_dev, double* sumsq_dev, uint64_t* count, void* stream) {
  if (!s) return fail(LMC_E_INVALID, "NULL sampler");
  DeviceGuard dg(s->device);
  if (!s->moments) return fail(LMC_E_STATE, "sampler was created with moments = 0");
  hipStream_t st = S(stream);
  const size_t n = (size_t)s->prob.H * s->prob.W;
  if (!rccl_comm) return lmc_sampler_get_moments(s, sum_dev, sumsq_dev, count, stream);   // a job of one rank
  RcclApi* R = rccl_api();
  if (!R->lib) return fail(LMC_E_UNSUPPORTED, "%s", R->why.c_str());
  if (!s->packed) HIP_TRY(hipMalloc(&s->packed, sizeof(double) * (2 * n + 1)));
  // one packed buffer {sum x, sum x^2, count}: ONE ncclAllReduce(sum) over xGMI (4 MiB at 512 x 512), in place
  HIP_TRY(hipMemcpyAsync(s->packed, s->s1, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  HIP_TRY(hipMemcpyAsync(s->packed + n, s->s2, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  const double cnt = (double)s->count;                      // exact below 2^53 samples
  HIP_TRY(hipMemcpyAsync(s->packed + 2 * n, &cnt, sizeof(double), hipMemcpyHostToDevice, st));
  RCCL_TRY(R, R->AllReduce(s->packed, s->packed, 2 * n + 1, ncclFloat64, ncclSum, static_cast<ncclComm_t>(rccl_comm), st));
  if (sum_dev) HIP_TRY(hipMemcpyAsync(sum_dev, s->packed, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  if (sumsq_dev) HIP_TRY(hipMemcpyAsync(sumsq_dev, s->packed + n, sizeof(double) * n, hipMemcpyDeviceToDevice, st));
  double total = 0.0;
  HIP_TRY(hipMemcpyAsync(&total, s->packed + 2 * n, sizeof(double), hipMemcpyDeviceToHost, st));
  HIP_TRY(hipStreamSynchronize(st));
  if (count) *count = (uint64_t)(total + 0.5);
  return LMC_OK;
}

float lmc_set_cg_tolerance(float tol) {
  const float prev = g_cg_tol;
  if (tol >= 0.f) g_cg_tol = tol;
  return prev;
}

int lmc_set_step_variant(int32_t variant) {
  if (variant < 0 || variant > 7 || variant == 2)
    return fail(LMC_E_INVALID, "variant must be 0 (auto), 1 (tile), 3 (split), 4 (point), 5 (block), 6 (rows) or 7 (pipe); 2 (the one-group "
                "streaming kernel of ABI 1) was removed");
  const int prev = g_variant;
  g_variant = variant;
  return prev;
}

}  // extern "C"
