// ULPDA (Unadjusted Langevin Primal-Dual, algs.py:425-449) building blocks, batched over chains.
//
//   x    <- prox_{tau f}(x - tau (A^T y + z)) + sqrt(2 tau) xi          f = sigma/2 ||H x - b||^2
//   xhat <- x + theta (x - x_old)
//   y    <- prox_{mu g*}(y + mu A xhat)                                  A = forward-difference gradient
//
// prox_{tau f}(v) = (I + tau sigma H^T H)^{-1} (v + tau sigma H^T b) is solved per chain by a fixed number of
// conjugate-gradient iterations, warm-started from the previous solve (build-specified inner solver; the reference
// calls LSQR through pyproximal.L2.prox / algs.py:247-256 -- parity unpinned there).  All scalars of the CG
// recurrences live on the device ([C] arrays), so a step enqueues a fixed sequence of launches with no host sync.
// These kernels are plain coalesced global-memory passes (first correct version of the "next" row f1).
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

static inline int grid1d(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  return (int)(g < 16384 ? (g ? g : 1) : 16384);
}

// y <- proj(y + mu * grad(xhat)):  iso: per-pixel l2 ball of radius `radius`; aniso: clip to [-radius, radius]
__global__ __launch_bounds__(256) void ulpda_dual_kernel(const float* __restrict__ xhat, float* __restrict__ y, int H, int W,
                                                         int64_t C, float mu, float radius, int iso) {
  const size_t img = (size_t)H * W, total = img * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c = i / img, p = i - c * img;
    const int r = (int)(p / W), col = (int)(p - (size_t)r * W);
    const float v = xhat[i];
    const float dx = (r + 1 < H) ? xhat[i + W] - v : 0.f;
    const float dy = (col + 1 < W) ? xhat[i + 1] - v : 0.f;
    float a = fmaf(mu, dx, y[c * 2 * img + p]);
    float b = fmaf(mu, dy, y[c * 2 * img + img + p]);
    if (iso) {
      const float sc = 1.f / fmaxf(1.f, sqrtf(fmaf(a, a, b * b)) / radius);
      a *= sc; b *= sc;
    } else {
      a = fminf(fmaxf(a, -radius), radius);
      b = fminf(fmaxf(b, -radius), radius);
    }
    y[c * 2 * img + p] = a;
    y[c * 2 * img + img + p] = b;
  }
}

// rhs = x - tau (A^T y + z) + ts * Htb     (A^T y = -div y; Htb = H^T b, shared by all chains; may be null)
__global__ __launch_bounds__(256) void ulpda_rhs_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                        const float* __restrict__ z, const float* __restrict__ htb,
                                                        float* __restrict__ rhs, int H, int W, int64_t C, float tau, float ts) {
  const size_t img = (size_t)H * W, total = img * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c = i / img, p = i - c * img;
    const int r = (int)(p / W), col = (int)(p - (size_t)r * W);
    const float* yr = y + c * 2 * img;
    const float* yc = yr + img;
    float aty = 0.f;
    if (r + 1 < H) aty -= yr[p];
    if (r > 0) aty += yr[p - W];
    if (col + 1 < W) aty -= yc[p];
    if (col > 0) aty += yc[p - 1];
    if (z) aty += z[p];
    float v = fmaf(-tau, aty, x[i]);
    if (htb) v = fmaf(ts, htb[p], v);
    rhs[i] = v;
  }
}

// rhs = v + coef * A^T( A v / max(|A v|, gamma) ) + ts * Htb : the in-place pre-step of L2_ncvx_tv.prox (algs.py:213-217)
// followed by the right-hand side of its linear solve (:225).  v and rhs must be different buffers.
__global__ __launch_bounds__(256) void ulpda_ncvx_rhs_kernel(const float* __restrict__ v, const float* __restrict__ htb,
                                                             float* __restrict__ rhs, int H, int W, int64_t C, float coef,
                                                             float gamma, float ts) {
  const size_t img = (size_t)H * W, total = img * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % img;
    const int r = (int)(p / W), c = (int)(p - (size_t)r * W);
    const bool up = r > 0, down = r + 1 < H, left = c > 0, right = c + 1 < W;
    const float x00 = v[i];
    const float xm0 = up ? v[i - W] : 0.f, xmp = (up && right) ? v[i - W + 1] : 0.f;
    const float x0m = left ? v[i - 1] : 0.f, x0p = right ? v[i + 1] : 0.f;
    const float xpm = (down && left) ? v[i + W - 1] : 0.f, xp0 = down ? v[i + W] : 0.f;
    float out = fmaf(coef, mc_tv_grad(xm0, xmp, x0m, x00, x0p, xpm, xp0, up, down, left, right, gamma), x00);
    if (htb) out = fmaf(ts, htb[p], out);
    rhs[i] = out;
  }
}

// rhs = v + coef * (v - extra) + ts * Htb : ME-TV pre-step of L2_ncvx_tv.prox (algs.py:221-223), extra = prox_{gamma TV}(v)
__global__ __launch_bounds__(256) void ulpda_me_rhs_kernel(const float* __restrict__ v, const float* __restrict__ extra,
                                                           const float* __restrict__ htb, float* __restrict__ rhs, size_t img,
                                                           int64_t C, float coef, float ts) {
  const size_t total = img * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float x = v[i];
    float out = fmaf(coef, x - extra[i], x);
    if (htb) out = fmaf(ts, htb[i % img], out);
    rhs[i] = out;
  }
}

hipError_t ulpda_me_rhs(const float* v, const float* extra, const float* htb, float* rhs, int64_t C, int H, int W, float coef,
                        float ts, hipStream_t st) {
  hipLaunchKernelGGL(ulpda_me_rhs_kernel, dim3(grid1d((size_t)H * W * C, 256)), dim3(256), 0, st, v, extra, htb, rhs,
                     (size_t)H * W, C, coef, ts);
  return hipGetLastError();
}

hipError_t ulpda_ncvx_rhs(const float* v, const float* htb, float* rhs, int64_t C, int H, int W, float coef, float gamma, float ts,
                          hipStream_t st) {
  hipLaunchKernelGGL(ulpda_ncvx_rhs_kernel, dim3(grid1d((size_t)H * W * C, 256)), dim3(256), 0, st, v, htb, rhs, H, W, C, coef,
                     gamma, ts);
  return hipGetLastError();
}

// q = p + ts * (second operand already holds H^T H p): q = p + ts*hthp ; accumulate dot(p, q) per chain
__global__ __launch_bounds__(256) void cg_q_kernel(const float* __restrict__ p, float* __restrict__ q /* in: HtHp, out: q */,
                                                   size_t img, float ts, double* __restrict__ pq) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.y;
  double acc = 0.0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.x * blockDim.x) {
    const float pv = p[c * img + k];
    const float qv = fmaf(ts, q[c * img + k], pv);
    q[c * img + k] = qv;
    acc += (double)pv * (double)qv;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&pq[c], t);
}

// r = rhs - q ; p = r ; rs = dot(r, r)       (CG start, q = A u0)
__global__ __launch_bounds__(256) void cg_init_kernel(const float* __restrict__ rhs, const float* __restrict__ q,
                                                      float* __restrict__ r, float* __restrict__ p, size_t img,
                                                      double* __restrict__ rs, double* __restrict__ b2) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.y;
  double acc = 0.0, accb = 0.0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.x * blockDim.x) {
    const float bv = rhs[c * img + k];
    const float rv = bv - q[c * img + k];
    r[c * img + k] = rv;
    p[c * img + k] = rv;
    acc += (double)rv * (double)rv;
    accb += (double)bv * (double)bv;
  }
  const double t = block_sum(acc, scratch);
  __syncthreads();
  const double tb = block_sum(accb, scratch);
  if (threadIdx.x == 0) { unsafeAtomicAdd(&rs[c], t); if (b2) unsafeAtomicAdd(&b2[c], tb); }
}

// done[0] = 1 when every chain satisfies |r|^2 <= tol^2 |b|^2 (the stopping rule of the reference's solver: scipy lsqr's
// btol, algs.py:250 with the default 1e-6).  Once set, the kernels of the remaining iterations return at once.
__global__ __launch_bounds__(256) void cg_check_kernel(int64_t C, const double* __restrict__ rsv, const double* __restrict__ b2,
                                                       double tol2, int* __restrict__ done) {
  if (*done) return;
  int ok = 1;
  for (int64_t c = threadIdx.x; c < C; c += blockDim.x) ok &= (rsv[c] <= tol2 * b2[c]) ? 1 : 0;
  ok = __syncthreads_and(ok);
  if (threadIdx.x == 0 && ok) *done = 1;
}

// Number of Chebyshev launches a solve needs, from the statistics of its first launch (stat[2c] = sum (u_1 - u_0)^2 = alpha_0^2 |r_0|^2,
// stat[2c+1] = |rhs|^2): the k-th iterate has |r_k| <= 2 c^k |r_0|, so k = ceil(log(2 |r_0| / (tol |rhs|)) / log(1/c)), at least the one
// launch already done, rounded up to even (the ping-pong then ends in the caller's buffer), at most kmax; the maximum over the chains.
__global__ __launch_bounds__(256) void cheb_count_kernel(int64_t C, const double* __restrict__ stat, double inv_alpha2, double tol,
                                                         double inv_log_inv_c, int kmax, int* __restrict__ count) {
  int need = 2;
  for (int64_t c = threadIdx.x; c < C; c += blockDim.x) {
    const double r2 = stat[2 * c] * inv_alpha2, b2 = stat[2 * c + 1];
    int k = 1;
    if (!(b2 > 0.0) || !(r2 == r2)) k = kmax;                       // no scale to compare with (or NaN): the a-priori count
    else if (r2 > tol * tol * b2) k = (int)ceil(log(2.0 * sqrt(r2 / b2) / tol) * inv_log_inv_c);
    k = (k + 1) & ~1;
    need = max(need, min(k, kmax));
  }
  atomicMax(count, need);
}

hipError_t cheb_count(int64_t C, const double* stat, double inv_alpha2, double tol, double inv_log_inv_c, int kmax, int* count,
                      hipStream_t st) {
  hipLaunchKernelGGL(cheb_count_kernel, dim3(1), dim3(256), 0, st, C, stat, inv_alpha2, tol, inv_log_inv_c, kmax, count);
  return hipGetLastError();
}

// alpha = rs/pq ; u += alpha p ; r -= alpha q ; rs_new = dot(r, r)
__global__ __launch_bounds__(256) void cg_update_kernel(float* __restrict__ u, float* __restrict__ r, const float* __restrict__ p,
                                                        const float* __restrict__ q, size_t img, const double* __restrict__ rs,
                                                        const double* __restrict__ pq, double* __restrict__ rs_new,
                                                        const int* __restrict__ done) {
  __shared__ double scratch[4];
  if (done && *done) return;
  const size_t c = blockIdx.y;
  const double den = pq[c];
  const float alpha = (rs[c] > 0.0 && den != 0.0) ? (float)(rs[c] / den) : 0.f;   // rs == 0: converged, freeze
  double acc = 0.0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.x * blockDim.x) {
    u[c * img + k] = fmaf(alpha, p[c * img + k], u[c * img + k]);
    const float rv = fmaf(-alpha, q[c * img + k], r[c * img + k]);
    r[c * img + k] = rv;
    acc += (double)rv * (double)rv;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&rs_new[c], t);
}

// p = r + (rs_new/rs) p
__global__ __launch_bounds__(256) void cg_dir_kernel(float* __restrict__ p, const float* __restrict__ r, size_t img,
                                                     double* __restrict__ rs, const double* __restrict__ rs_new,
                                                     const int* __restrict__ done) {
  if (done && *done) return;
  const size_t c = blockIdx.y;
  const float beta = rs[c] > 0.0 ? (float)(rs_new[c] / rs[c]) : 0.f;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.x * blockDim.x)
    p[c * img + k] = fmaf(beta, p[c * img + k], r[c * img + k]);
}

// closed-form data steps: identity  u = (v + ts b)/(1 + ts) ; mask  u = (v + ts m b)/(1 + ts m^2) ; none  u = v
__global__ __launch_bounds__(256) void ulpda_pointwise_prox_kernel(const float* __restrict__ v, float* __restrict__ u,
                                                                   const float* __restrict__ b, const float* __restrict__ m,
                                                                   size_t img, int64_t C, float ts, int kind) {
  const size_t total = img * C;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t p = i % img;
    float out = v[i];
    if (kind == LMC_DATA_IDENTITY) out = fmaf(ts, b[p], out) / (1.f + ts);
    else if (kind == LMC_DATA_MASK) { const float mk = m[p]; out = fmaf(ts * mk, b[p], out) / fmaf(ts * mk, mk, 1.f); }
    u[i] = out;
  }
}

// x_new = u + s*xi ; xhat = x_new + theta (x_new - x_old) ; warm <- u ; x <- x_new     (xi may be null: no noise)
__global__ __launch_bounds__(256) void ulpda_finish_kernel(float* __restrict__ x, float* __restrict__ xhat,
                                                           const float* __restrict__ u, const float* __restrict__ xi,
                                                           size_t total, float s, float theta) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const float xo = x[i];
    const float xn = xi ? fmaf(s, xi[i], u[i]) : u[i];
    x[i] = xn;
    xhat[i] = fmaf(theta, xn - xo, xn);
  }
}

// pq[c] = dot(p_c, q_c)
__global__ __launch_bounds__(256) void cg_dot_kernel(const float* __restrict__ p, const float* __restrict__ q, size_t img,
                                                     double* __restrict__ pq, const int* __restrict__ done) {
  __shared__ double scratch[4];
  if (done && *done) return;
  const size_t c = blockIdx.y;
  double acc = 0.0;
  for (size_t k = (size_t)blockIdx.x * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.x * blockDim.x)
    acc += (double)p[c * img + k] * (double)q[c * img + k];
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&pq[c], t);
}

static inline dim3 cg_grid(size_t img, int64_t C) {
  int gx = (int)((img + 255) / 256);
  if (gx > 128) gx = 128;
  return dim3(gx, (unsigned)C);
}

hipError_t cg_dot(const float* p, const float* q, int64_t C, size_t img, double* pq, const int* done, hipStream_t st) {
  hipLaunchKernelGGL(cg_dot_kernel, cg_grid(img, C), dim3(256), 0, st, p, q, img, pq, done);
  return hipGetLastError();
}
hipError_t cg_init(const float* rhs, const float* q, float* r, float* p, int64_t C, size_t img, double* rs, double* b2, hipStream_t st) {
  hipLaunchKernelGGL(cg_init_kernel, cg_grid(img, C), dim3(256), 0, st, rhs, q, r, p, img, rs, b2);
  return hipGetLastError();
}
hipError_t cg_update(float* u, float* r, const float* p, const float* q, int64_t C, size_t img, const double* rs, const double* pq,
                     double* rs_new, const int* done, hipStream_t st) {
  hipLaunchKernelGGL(cg_update_kernel, cg_grid(img, C), dim3(256), 0, st, u, r, p, q, img, rs, pq, rs_new, done);
  return hipGetLastError();
}
hipError_t cg_dir(float* p, const float* r, int64_t C, size_t img, double* rs, const double* rs_new, const int* done, hipStream_t st) {
  hipLaunchKernelGGL(cg_dir_kernel, cg_grid(img, C), dim3(256), 0, st, p, r, img, rs, rs_new, done);
  return hipGetLastError();
}
hipError_t cg_check(int64_t C, const double* rsv, const double* b2, double tol2, int* done, hipStream_t st) {
  hipLaunchKernelGGL(cg_check_kernel, dim3(1), dim3(256), 0, st, C, rsv, b2, tol2, done);
  return hipGetLastError();
}

// ---- host-side sequences --------------------------------------------------------------------------------

hipError_t ulpda_dual_update(const float* xhat, float* y, int64_t C, int H, int W, float mu, float radius, int iso,
                             hipStream_t st) {
  hipLaunchKernelGGL(ulpda_dual_kernel, dim3(grid1d((size_t)H * W * C, 256)), dim3(256), 0, st, xhat, y, H, W, C, mu, radius, iso);
  return hipGetLastError();
}

hipError_t ulpda_rhs(const float* x, const float* y, const float* z, const float* htb, float* rhs, int64_t C, int H, int W,
                     float tau, float ts, hipStream_t st) {
  hipLaunchKernelGGL(ulpda_rhs_kernel, dim3(grid1d((size_t)H * W * C, 256)), dim3(256), 0, st, x, y, z, htb, rhs, H, W, C, tau, ts);
  return hipGetLastError();
}

hipError_t ulpda_pointwise_prox(const float* v, float* u, const float* b, const float* m, int64_t C, int H, int W, float ts,
                                int kind, hipStream_t st) {
  hipLaunchKernelGGL(ulpda_pointwise_prox_kernel, dim3(grid1d((size_t)H * W * C, 256)), dim3(256), 0, st, v, u, b, m,
                     (size_t)H * W, C, ts, kind);
  return hipGetLastError();
}

hipError_t ulpda_finish(float* x, float* xhat, const float* u, const float* xi, int64_t C, int H, int W, float s, float theta,
                        hipStream_t st) {
  const size_t total = (size_t)H * W * C;
  hipLaunchKernelGGL(ulpda_finish_kernel, dim3(grid1d(total, 256)), dim3(256), 0, st, x, xhat, u, xi, total, s, theta);
  return hipGetLastError();
}

// Solve (I + ts H^T H) u = rhs for every chain with `niter` CG iterations from the current content of u.
// Scratch: r, p, q, tmp [C][H][W]; scal: 3*C doubles (rs, pq, rs_new).
hipError_t ulpda_cg_solve(float* u, const float* rhs, float* r, float* p, float* q, float* tmp, double* scal, int64_t C, int H,
                          int W, const BlurTaps& T, float ts, int niter, hipStream_t st) {
  const size_t img = (size_t)H * W;
  double* rs = scal;
  double* pq = scal + C;
  double* rs_new = scal + 2 * C;
  int gx = (int)((img + 255) / 256);
  if (gx > 128) gx = 128;
  const dim3 grid(gx, (unsigned)C), block(256);
  hipError_t e;
  if ((e = hipMemsetAsync(scal, 0, sizeof(double) * 3 * C, st)) != hipSuccess) return e;
  // q = A u
  if ((e = launch_blur(u, tmp, C, H, W, T, 0, st)) != hipSuccess) return e;
  if ((e = launch_blur(tmp, q, C, H, W, T, 1, st)) != hipSuccess) return e;
  hipLaunchKernelGGL(cg_q_kernel, grid, block, 0, st, u, q, img, ts, pq);
  hipLaunchKernelGGL(cg_init_kernel, grid, block, 0, st, rhs, q, r, p, img, rs, (double*)nullptr);
  for (int it = 0; it < niter; ++it) {
    if ((e = hipMemsetAsync(pq, 0, sizeof(double) * C, st)) != hipSuccess) return e;
    if ((e = hipMemsetAsync(rs_new, 0, sizeof(double) * C, st)) != hipSuccess) return e;
    if ((e = launch_blur(p, tmp, C, H, W, T, 0, st)) != hipSuccess) return e;
    if ((e = launch_blur(tmp, q, C, H, W, T, 1, st)) != hipSuccess) return e;
    hipLaunchKernelGGL(cg_q_kernel, grid, block, 0, st, p, q, img, ts, pq);
    hipLaunchKernelGGL(cg_update_kernel, grid, block, 0, st, u, r, p, q, img, rs, pq, rs_new, (const int*)nullptr);
    hipLaunchKernelGGL(cg_dir_kernel, grid, block, 0, st, p, r, img, rs, rs_new, (const int*)nullptr);
    if ((e = hipMemcpyAsync(rs, rs_new, sizeof(double) * C, hipMemcpyDeviceToDevice, st)) != hipSuccess) return e;
  }
  return hipGetLastError();
}

}  // namespace lmc
