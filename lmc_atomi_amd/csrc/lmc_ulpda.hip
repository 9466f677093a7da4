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

// r = rhs - q ; p = r ; rs = dot(r, r)       (CG start, q = A u0)
__global__ __launch_bounds__(256) void cg_init_kernel(const float* __restrict__ rhs, const float* __restrict__ q,
                                                      float* __restrict__ r, float* __restrict__ p, size_t img,
                                                      double* __restrict__ rs, double* __restrict__ b2) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.x;
  double acc = 0.0, accb = 0.0;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x) {
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
  const size_t c = blockIdx.x;
  const double den = pq[c];
  const float alpha = (rs[c] > 0.0 && den != 0.0) ? (float)(rs[c] / den) : 0.f;   // rs == 0: converged, freeze
  double acc = 0.0;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x) {
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
  const size_t c = blockIdx.x;
  const float beta = rs[c] > 0.0 ? (float)(rs_new[c] / rs[c]) : 0.f;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x)
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
  const size_t c = blockIdx.x;
  double acc = 0.0;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x)
    acc += (double)p[c * img + k] * (double)q[c * img + k];
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&pq[c], t);
}

// chains on gridDim.x (no 65535 limit), pixel blocks of one image on gridDim.y
static inline dim3 cg_grid(size_t img, int64_t C) {
  int gy = (int)((img + 255) / 256);
  if (gy > 128) gy = 128;
  return dim3((unsigned)C, gy);
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

// ---- 4 pixels per thread (W % 4 == 0): b128 accesses, one 32-bit division per thread instead of two 64-bit ones per pixel ----------
// grid: x over groups of 4 pixels of one image, y over chains (chunks of 65535)

__global__ __launch_bounds__(256) void ulpda_dual4_kernel(const float* __restrict__ xhat, float* __restrict__ y, int H, int W,
                                                          float mu, float radius, int iso) {
  const unsigned img = (unsigned)H * (unsigned)W;
  const unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  if (p >= img) return;
  const unsigned r = p / (unsigned)W, col = p - r * (unsigned)W;
  const float* xc = xhat + (size_t)blockIdx.y * img;
  float* yr = y + (size_t)blockIdx.y * 2 * img;
  float* yc = yr + img;
  const float4 v = *reinterpret_cast<const float4*>(xc + p);
  const bool down = r + 1 < (unsigned)H;
  const float4 vd = down ? *reinterpret_cast<const float4*>(xc + p + W) : v;       // last row: dx = v - v = 0
  const float vr = (col + 4 < (unsigned)W) ? xc[p + 4] : v.w;                          // last column: dy = 0
  const float4 a4 = *reinterpret_cast<const float4*>(yr + p);
  const float4 b4 = *reinterpret_cast<const float4*>(yc + p);
  float a[4] = {fmaf(mu, vd.x - v.x, a4.x), fmaf(mu, vd.y - v.y, a4.y), fmaf(mu, vd.z - v.z, a4.z), fmaf(mu, vd.w - v.w, a4.w)};
  float b[4] = {fmaf(mu, v.y - v.x, b4.x), fmaf(mu, v.z - v.y, b4.y), fmaf(mu, v.w - v.z, b4.z), fmaf(mu, vr - v.w, b4.w)};
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (iso) {
      const float sc = 1.f / fmaxf(1.f, sqrtf(fmaf(a[k], a[k], b[k] * b[k])) / radius);
      a[k] *= sc; b[k] *= sc;
    } else {
      a[k] = fminf(fmaxf(a[k], -radius), radius);
      b[k] = fminf(fmaxf(b[k], -radius), radius);
    }
  }
  *reinterpret_cast<float4*>(yr + p) = make_float4(a[0], a[1], a[2], a[3]);
  *reinterpret_cast<float4*>(yc + p) = make_float4(b[0], b[1], b[2], b[3]);
}

__global__ __launch_bounds__(256) void ulpda_rhs4_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                         const float* __restrict__ z, const float* __restrict__ htb,
                                                         float* __restrict__ rhs, int H, int W, float tau, float ts) {
  const unsigned img = (unsigned)H * (unsigned)W;
  const unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  if (p >= img) return;
  const unsigned r = p / (unsigned)W, col = p - r * (unsigned)W;
  const float* yr = y + (size_t)blockIdx.y * 2 * img;
  const float* yc = yr + img;
  const float4 z0 = make_float4(0.f, 0.f, 0.f, 0.f);
  const float4 ra = (r + 1 < (unsigned)H) ? *reinterpret_cast<const float4*>(yr + p) : z0;
  const float4 rb = (r > 0) ? *reinterpret_cast<const float4*>(yr + p - W) : z0;
  float4 ca = *reinterpret_cast<const float4*>(yc + p);
  const float cl = (col > 0) ? yc[p - 1] : 0.f;
  if (col + 4 >= (unsigned)W) ca.w = 0.f;                                             // last column of the column component counts as zero
  // A^T y = -div y, terms in the order of the scalar kernel: -yr[p] + yr[p-W] - yc[p] + yc[p-1]
  float aty[4] = {((0.f - ra.x) + rb.x - ca.x) + cl, ((0.f - ra.y) + rb.y - ca.y) + ca.x, ((0.f - ra.z) + rb.z - ca.z) + ca.y,
                  ((0.f - ra.w) + rb.w - ca.w) + ca.z};
  if (z) {
    const float4 zz = *reinterpret_cast<const float4*>(z + p);
    aty[0] += zz.x; aty[1] += zz.y; aty[2] += zz.z; aty[3] += zz.w;
  }
  const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)blockIdx.y * img + p);
  float o[4] = {fmaf(-tau, aty[0], xv.x), fmaf(-tau, aty[1], xv.y), fmaf(-tau, aty[2], xv.z), fmaf(-tau, aty[3], xv.w)};
  if (htb) {
    const float4 hb = *reinterpret_cast<const float4*>(htb + p);
    o[0] = fmaf(ts, hb.x, o[0]); o[1] = fmaf(ts, hb.y, o[1]); o[2] = fmaf(ts, hb.z, o[2]); o[3] = fmaf(ts, hb.w, o[3]);
  }
  *reinterpret_cast<float4*>(rhs + (size_t)blockIdx.y * img + p) = make_float4(o[0], o[1], o[2], o[3]);
}

// Dual update of iteration k FUSED with the right-hand side of iteration k + 1 (gfirst = false: algs.py:448 followed by :443-445 of the next pass):
//   y' = proj(y + mu grad(xhat))      rhs = x - tau (A^T y' + z) + ts H^T b
// A^T y' at a pixel needs y' at the pixel, at the pixel above (row component) and at the pixel to the left (column component): those two dual
// updates are recomputed here (their inputs are cache hits: the neighbouring threads stream the same lines), y goes from y_in to y_out (other
// threads still read y_in at their neighbours: not in place).  28 instead of 20 + 16 B per pixel.  Same formulas in the same order as
// ulpda_dual4_kernel and ulpda_rhs4_kernel: bit-identical to running the two.
__device__ __forceinline__ void dual_proj(float& a, float& b, float radius, int iso) {
  if (iso) {
    const float sc = 1.f / fmaxf(1.f, sqrtf(fmaf(a, a, b * b)) / radius);
    a *= sc; b *= sc;
  } else {
    a = fminf(fmaxf(a, -radius), radius);
    b = fminf(fmaxf(b, -radius), radius);
  }
}
__global__ __launch_bounds__(256) void ulpda_dual_rhs4_kernel(const float* __restrict__ xhat, const float* __restrict__ y_in, float* __restrict__ y_out,
                                                              const float* __restrict__ x, const float* __restrict__ z, const float* __restrict__ htb,
                                                              float* __restrict__ rhs, int H, int W, float mu, float radius, int iso, float tau, float ts) {
  const unsigned img = (unsigned)H * (unsigned)W;
  const unsigned p = (blockIdx.x * blockDim.x + threadIdx.x) * 4u;
  if (p >= img) return;
  const unsigned r = p / (unsigned)W, col = p - r * (unsigned)W;
  const float* xc = xhat + (size_t)blockIdx.y * img;
  const float* yr = y_in + (size_t)blockIdx.y * 2 * img;
  const float* yc = yr + img;
  float* yro = y_out + (size_t)blockIdx.y * 2 * img;
  float* yco = yro + img;
  const bool down = r + 1 < (unsigned)H, up = r > 0, left = col > 0, right = col + 4 < (unsigned)W;
  // own pixels (as ulpda_dual4_kernel)
  const float4 v = *reinterpret_cast<const float4*>(xc + p);
  const float4 vd = down ? *reinterpret_cast<const float4*>(xc + p + W) : v;
  const float vr = right ? xc[p + 4] : v.w;
  const float4 a4 = *reinterpret_cast<const float4*>(yr + p);
  const float4 b4 = *reinterpret_cast<const float4*>(yc + p);
  float a[4] = {fmaf(mu, vd.x - v.x, a4.x), fmaf(mu, vd.y - v.y, a4.y), fmaf(mu, vd.z - v.z, a4.z), fmaf(mu, vd.w - v.w, a4.w)};
  float b[4] = {fmaf(mu, v.y - v.x, b4.x), fmaf(mu, v.z - v.y, b4.y), fmaf(mu, v.w - v.z, b4.z), fmaf(mu, vr - v.w, b4.w)};
#pragma unroll
  for (int k = 0; k < 4; ++k) dual_proj(a[k], b[k], radius, iso);
  *reinterpret_cast<float4*>(yro + p) = make_float4(a[0], a[1], a[2], a[3]);
  *reinterpret_cast<float4*>(yco + p) = make_float4(b[0], b[1], b[2], b[3]);
  // the row above: its row component (its "down" neighbour is this row, so it is never a last row)
  float ua[4] = {0.f, 0.f, 0.f, 0.f};
  if (up) {
    const float4 vu = *reinterpret_cast<const float4*>(xc + p - W);
    const float vur = right ? xc[p - W + 4] : vu.w;
    const float4 ua4 = *reinterpret_cast<const float4*>(yr + p - W);
    const float4 ub4 = *reinterpret_cast<const float4*>(yc + p - W);
    float ub[4] = {fmaf(mu, vu.y - vu.x, ub4.x), fmaf(mu, vu.z - vu.y, ub4.y), fmaf(mu, vu.w - vu.z, ub4.z), fmaf(mu, vur - vu.w, ub4.w)};
    ua[0] = fmaf(mu, v.x - vu.x, ua4.x); ua[1] = fmaf(mu, v.y - vu.y, ua4.y); ua[2] = fmaf(mu, v.z - vu.z, ua4.z); ua[3] = fmaf(mu, v.w - vu.w, ua4.w);
#pragma unroll
    for (int k = 0; k < 4; ++k) dual_proj(ua[k], ub[k], radius, iso);
  }
  // the pixel to the left: its column component (its "right" neighbour is this thread's first pixel, so it is never a last column)
  float lb = 0.f;
  if (left) {
    const float vl = xc[p - 1];
    const float vld = down ? xc[p - 1 + W] : vl;
    float la = fmaf(mu, vld - vl, yr[p - 1]);
    lb = fmaf(mu, v.x - vl, yc[p - 1]);
    dual_proj(la, lb, radius, iso);
  }
  // right-hand side (as ulpda_rhs4_kernel): A^T y = -div y, terms in the same order: -yr[p] + yr[p-W] - yc[p] + yc[p-1]; the row component of the
  // last row and the column component of the last column count as zero
  const float ra[4] = {down ? a[0] : 0.f, down ? a[1] : 0.f, down ? a[2] : 0.f, down ? a[3] : 0.f};
  const float ca[4] = {b[0], b[1], b[2], right ? b[3] : 0.f};
  float aty[4] = {((0.f - ra[0]) + ua[0] - ca[0]) + lb, ((0.f - ra[1]) + ua[1] - ca[1]) + ca[0], ((0.f - ra[2]) + ua[2] - ca[2]) + ca[1],
                  ((0.f - ra[3]) + ua[3] - ca[3]) + ca[2]};
  if (z) {
    const float4 zz = *reinterpret_cast<const float4*>(z + p);
    aty[0] += zz.x; aty[1] += zz.y; aty[2] += zz.z; aty[3] += zz.w;
  }
  const float4 xv = *reinterpret_cast<const float4*>(x + (size_t)blockIdx.y * img + p);
  float o[4] = {fmaf(-tau, aty[0], xv.x), fmaf(-tau, aty[1], xv.y), fmaf(-tau, aty[2], xv.z), fmaf(-tau, aty[3], xv.w)};
  if (htb) {
    const float4 hb = *reinterpret_cast<const float4*>(htb + p);
    o[0] = fmaf(ts, hb.x, o[0]); o[1] = fmaf(ts, hb.y, o[1]); o[2] = fmaf(ts, hb.z, o[2]); o[3] = fmaf(ts, hb.w, o[3]);
  }
  *reinterpret_cast<float4*>(rhs + (size_t)blockIdx.y * img + p) = make_float4(o[0], o[1], o[2], o[3]);
}

__global__ __launch_bounds__(256) void ulpda_finish4_kernel(float4* __restrict__ x, float4* __restrict__ xhat, const float4* __restrict__ u,
                                                            const float4* __restrict__ xi, size_t total4, float s, float theta) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (size_t)gridDim.x * blockDim.x) {
    const float4 xo = x[i];
    float4 xn = u[i];
    if (xi) { const float4 n = xi[i]; xn.x = fmaf(s, n.x, xn.x); xn.y = fmaf(s, n.y, xn.y); xn.z = fmaf(s, n.z, xn.z); xn.w = fmaf(s, n.w, xn.w); }
    x[i] = xn;
    xhat[i] = make_float4(fmaf(theta, xn.x - xo.x, xn.x), fmaf(theta, xn.y - xo.y, xn.y), fmaf(theta, xn.z - xo.z, xn.z),
                          fmaf(theta, xn.w - xo.w, xn.w));
  }
}

// finish with the Philox field drawn in place (no noise buffer, no separate noise pass): one thread = a 4 x 4 block = four quads of the
// counter layout shared by every kernel (ctr = (quad, iteration, global chain, stream tag), one quad = 4 rows of one column).
__global__ __launch_bounds__(256) void ulpda_finish_philox_kernel(float* __restrict__ x, float* __restrict__ xhat, const float* __restrict__ u,
                                                                  int H, int W, float s, float theta, uint32_t key0, uint32_t key1,
                                                                  uint32_t iteration, uint32_t chain_offset) {
  const unsigned w4 = (unsigned)W >> 2, nq = ((unsigned)H + 3u) >> 2;
  const unsigned t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= w4 * nq) return;
  const unsigned q = t / w4, g = t - q * w4;
  const size_t base = (size_t)blockIdx.y * H * W;
  // all loads first (in flight under the Philox arithmetic), then all stores: stores count in vmcnt like loads and complete in order with
  // them, so a load issued after a store of this lane would wait for that store
  float4 xo[4], uu[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned r = min(4u * q + j, (unsigned)H - 1u);
    const size_t i = base + (size_t)r * W + 4u * g;
    xo[j] = *reinterpret_cast<const float4*>(x + i);
    uu[j] = *reinterpret_cast<const float4*>(u + i);
  }
  float n[4][4];                                   // [column][row]
#pragma unroll
  for (int c = 0; c < 4; ++c) quad_normals(key0, key1, iteration, chain_offset + blockIdx.y, q * (unsigned)W + 4u * g + c, n[c]);
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned r = 4u * q + j;
    if (r < (unsigned)H) {
      const size_t i = base + (size_t)r * W + 4u * g;
      const float4 xn = make_float4(fmaf(s, n[0][j], uu[j].x), fmaf(s, n[1][j], uu[j].y), fmaf(s, n[2][j], uu[j].z), fmaf(s, n[3][j], uu[j].w));
      *reinterpret_cast<float4*>(x + i) = xn;
      *reinterpret_cast<float4*>(xhat + i) = make_float4(fmaf(theta, xn.x - xo[j].x, xn.x), fmaf(theta, xn.y - xo[j].y, xn.y),
                                                         fmaf(theta, xn.z - xo[j].z, xn.z), fmaf(theta, xn.w - xo[j].w, xn.w));
    }
  }
}

hipError_t ulpda_finish_philox(float* x, float* xhat, const float* u, int64_t C, int H, int W, float s, float theta, uint32_t key0,
                               uint32_t key1, uint32_t iteration, uint32_t chain_offset, hipStream_t st) {
  if (W & 3) return hipErrorInvalidConfiguration;
  const size_t img = (size_t)H * W;
  const unsigned gx = (unsigned)(((size_t)(W >> 2) * ((H + 3) >> 2) + 255) / 256);
  for (int64_t c0 = 0; c0 < C; c0 += 65535) {
    const unsigned nc = (unsigned)((C - c0) < 65535 ? (C - c0) : 65535);
    hipLaunchKernelGGL(ulpda_finish_philox_kernel, dim3(gx, nc), dim3(256), 0, st, x + c0 * img, xhat + c0 * img, u + c0 * img, H, W, s, theta,
                       key0, key1, iteration, chain_offset + (uint32_t)c0);
  }
  return hipGetLastError();
}

// ---- fused finish + dual update (gfirst = false, algs.py:446-448):  x <- u + s xi ; xhat = x + theta (x - x_old) ;
//      y <- prox_{mu g*}(y + mu A xhat)   in ONE row-streaming pass.  xhat is consumed in registers and never stored: with gfirst = false
// nothing else reads it.  One wavefront owns the full width of a band of rows of one chain (lane = PXL consecutive pixels, W <= 64 PXL,
// W % 4 == 0), so the right-hand neighbour of the forward difference is in the same lane or one wave-shift DPP move away, and the row
// below is the next row the wave computes -- a band looks one row ahead (recomputed by the next band: its Philox quad included).
// HBM bytes per pixel: x_old 4 + u 4 + y 8 read, x 4 + y 8 written = 28 (the two separate passes: 16 + 20 = 36).  x_new goes to a second
// state buffer (ping-pong).
template <int PXL>
__global__ __launch_bounds__(256) void ulpda_finish_dual_kernel(const float* __restrict__ x, float* __restrict__ xnew, const float* __restrict__ u, float* __restrict__ y,
                                                                const float* __restrict__ xi, int H, int W, int C, int band_rows, int nbands,
                                                                float s, float theta, float mu, float radius, int iso, int philox,
                                                                uint32_t key0, uint32_t key1, uint32_t iteration, uint32_t chain_offset) {
  const int lane = threadIdx.x & 63;
  const int gw = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (gw >= C * nbands) return;
  const int chain = gw / nbands, band = gw - chain * nbands;
  const int r0 = band * band_rows, r1 = min(r0 + band_rows, H);      // band_rows % 4 == 0: bands start on Philox quad rows
  const int c0 = lane * PXL;
  const size_t img = (size_t)H * W;
  const float* __restrict__ xc = x + (size_t)chain * img;       // x_old: read-only here (the look-ahead row of a band is the first row
  float* __restrict__ xo_ = xnew + (size_t)chain * img;         // of the next one, which writes its x_new concurrently: separate buffers)
  const float* __restrict__ uc = u + (size_t)chain * img;
  const float* __restrict__ nc = xi ? xi + (size_t)chain * img : nullptr;
  float* __restrict__ yr = y + (size_t)chain * 2 * img;
  float* __restrict__ yc = yr + img;
  constexpr int NG = PXL / 4;
  bool gok[NG];
#pragma unroll
  for (int g = 0; g < NG; ++g) gok[g] = c0 + 4 * g < W;
  float nz[PXL][4];                    // normals of the current quad row-group: [pixel][row in quad]
  float xh_prev[PXL], xh_cur[PXL];
#pragma unroll
  for (int k = 0; k < PXL; ++k) { xh_prev[k] = 0.f; xh_cur[k] = 0.f; nz[k][0] = nz[k][1] = nz[k][2] = nz[k][3] = 0.f; }
  const int i_end = r1 < H ? r1 : H - 1;              // last row whose xhat is formed: the look-ahead row r1, or the last image row
  // software pipeline: the loads of row i + 1 (x_old, u, injected noise, and the dual rows of row i) are issued before the arithmetic of
  // row i; vector-memory operations complete in order, so they stay in flight behind nothing but the stores of row i - 1
  float4 xo4[2][NG], uu4[2][NG], nn4[2][NG], ya4[2][NG], yb4[2][NG];
  auto fetch = [&](const int slot, const int row) __attribute__((always_inline)) {
    const size_t go = (size_t)min(row, H - 1) * W;
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const int cc = gok[g] ? c0 + 4 * g : 0;             // lanes past the row read its start (valid memory), their results are never stored
      xo4[slot][g] = *reinterpret_cast<const float4*>(xc + go + cc);
      uu4[slot][g] = *reinterpret_cast<const float4*>(uc + go + cc);
      if (nc) nn4[slot][g] = *reinterpret_cast<const float4*>(nc + go + cc);
      if (row > r0) {                                      // dual rows of row - 1
        ya4[slot][g] = *reinterpret_cast<const float4*>(yr + go - W + cc);
        yb4[slot][g] = *reinterpret_cast<const float4*>(yc + go - W + cc);
      }
    }
  };
#pragma unroll
  for (int sl = 0; sl < 2; ++sl)
#pragma unroll
    for (int g = 0; g < NG; ++g) nn4[sl][g] = ya4[sl][g] = yb4[sl][g] = make_float4(0.f, 0.f, 0.f, 0.f);
  fetch(0, r0);
  auto row_step = [&](auto ss, const int i) __attribute__((always_inline)) {
    constexpr int SL = decltype(ss)::value;
    if (i < i_end) fetch(SL ^ 1, i + 1);
    if (philox && (i & 3) == 0) {
#pragma unroll
      for (int k = 0; k < PXL; ++k)
        if (gok[k >> 2]) quad_normals(key0, key1, iteration, chain_offset + (uint32_t)chain, (uint32_t)(i >> 2) * (uint32_t)W + (uint32_t)(c0 + k), nz[k]);
    }
    const size_t go = (size_t)i * W;
    // x_new and xhat of row i
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      const float xo[4] = {xo4[SL][g].x, xo4[SL][g].y, xo4[SL][g].z, xo4[SL][g].w};
      const float uu[4] = {uu4[SL][g].x, uu4[SL][g].y, uu4[SL][g].z, uu4[SL][g].w};
      const float nn[4] = {nn4[SL][g].x, nn4[SL][g].y, nn4[SL][g].z, nn4[SL][g].w};
      float xn[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float nv = philox ? nz[4 * g + q][i & 3] : nn[q];
        xn[q] = (philox || nc) ? fmaf(s, nv, uu[q]) : uu[q];
        xh_cur[4 * g + q] = fmaf(theta, xn[q] - xo[q], xn[q]);
      }
      if (gok[g] && i < r1) *reinterpret_cast<float4*>(xo_ + go + c0 + 4 * g) = make_float4(xn[0], xn[1], xn[2], xn[3]);
    }
    // dual update of row j = i - 1 (its vertical difference needs row i)
    if (i > r0) {
      const float right_edge = dpp_right0(xh_prev[0]);          // first pixel of the lane to the right (0 past the wave)
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const float av[4] = {ya4[SL][g].x, ya4[SL][g].y, ya4[SL][g].z, ya4[SL][g].w}, bv[4] = {yb4[SL][g].x, yb4[SL][g].y, yb4[SL][g].z, yb4[SL][g].w};
        float ao[4], bo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = 4 * g + q;
          const float here = xh_prev[k];
          const float rgt = k == PXL - 1 ? right_edge : xh_prev[k + 1];
          const float dx = xh_cur[k] - here;
          const float dy = (c0 + k + 1 < W) ? rgt - here : 0.f;
          float a = fmaf(mu, dx, av[q]), b = fmaf(mu, dy, bv[q]);
          if (iso) {
            const float sc = 1.f / fmaxf(1.f, sqrtf(fmaf(a, a, b * b)) / radius);
            a *= sc; b *= sc;
          } else {
            a = fminf(fmaxf(a, -radius), radius);
            b = fminf(fmaxf(b, -radius), radius);
          }
          ao[q] = a; bo[q] = b;
        }
        if (gok[g]) {
          *reinterpret_cast<float4*>(yr + go - W + c0 + 4 * g) = make_float4(ao[0], ao[1], ao[2], ao[3]);
          *reinterpret_cast<float4*>(yc + go - W + c0 + 4 * g) = make_float4(bo[0], bo[1], bo[2], bo[3]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < PXL; ++k) xh_prev[k] = xh_cur[k];
  };
  for (int i = r0; i <= i_end; i += 2) {
    row_step(std::integral_constant<int, 0>{}, i);
    if (i + 1 <= i_end) row_step(std::integral_constant<int, 1>{}, i + 1);
  }
  if (r1 == H) {      // last image row (this band owns it): no row below, dx = 0
    const size_t go = (size_t)(H - 1) * W;
    const float right_edge = dpp_right0(xh_prev[0]);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      if (!gok[g]) continue;
      const float4 a4 = *reinterpret_cast<const float4*>(yr + go + c0 + 4 * g);
      const float4 b4 = *reinterpret_cast<const float4*>(yc + go + c0 + 4 * g);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w}, bv[4] = {b4.x, b4.y, b4.z, b4.w};
      float ao[4], bo[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int k = 4 * g + q;
        const float here = xh_prev[k];
        const float rgt = k == PXL - 1 ? right_edge : xh_prev[k + 1];
        const float dy = (c0 + k + 1 < W) ? rgt - here : 0.f;
        float a = av[q], b = fmaf(mu, dy, bv[q]);
        if (iso) {
          const float sc = 1.f / fmaxf(1.f, sqrtf(fmaf(a, a, b * b)) / radius);
          a *= sc; b *= sc;
        } else {
          a = fminf(fmaxf(a, -radius), radius);
          b = fminf(fmaxf(b, -radius), radius);
        }
        ao[q] = a; bo[q] = b;
      }
      *reinterpret_cast<float4*>(yr + go + c0 + 4 * g) = make_float4(ao[0], ao[1], ao[2], ao[3]);
      *reinterpret_cast<float4*>(yc + go + c0 + 4 * g) = make_float4(bo[0], bo[1], bo[2], bo[3]);
    }
  }
}

bool ulpda_finish_dual_supported(int H, int W) { return (W & 3) == 0 && W >= 4 && W <= 512 && H >= 1; }

// xi: injected noise [C][H][W] or NULL; philox != 0 draws the field in place (xi must then be NULL)
hipError_t ulpda_finish_dual(const float* x, float* xnew, const float* u, float* y, const float* xi, int64_t C, int H, int W, float s, float theta, float mu,
                             float radius, int iso, int philox, uint32_t key0, uint32_t key1, uint32_t iteration, uint32_t chain_offset,
                             hipStream_t st) {
  if (!ulpda_finish_dual_supported(H, W) || C > (1 << 24) || x == xnew) return hipErrorInvalidConfiguration;
  const int want = (int)((4096 + C - 1) / C);                       // bands per chain for ~4 waves per SIMD
  int band = (H + want - 1) / want;
  if (band < 16) band = 16;
  band = (band + 3) & ~3;
  const int nbands = (H + band - 1) / band;
  const long long waves = (long long)C * nbands;
  const int nblk = (int)((waves + 3) / 4);
  if (W <= 256)
    hipLaunchKernelGGL(ulpda_finish_dual_kernel<4>, dim3(nblk), dim3(256), 0, st, x, xnew, u, y, xi, H, W, (int)C, band, nbands, s, theta, mu, radius,
                       iso, philox, key0, key1, iteration, chain_offset);
  else
    hipLaunchKernelGGL(ulpda_finish_dual_kernel<8>, dim3(nblk), dim3(256), 0, st, x, xnew, u, y, xi, H, W, (int)C, band, nbands, s, theta, mu, radius,
                       iso, philox, key0, key1, iteration, chain_offset);
  return hipGetLastError();
}

static inline bool vec4_ok(int H, int W) { return (W & 3) == 0 && (size_t)H * W < (1ull << 31); }

hipError_t ulpda_dual_update(const float* xhat, float* y, int64_t C, int H, int W, float mu, float radius, int iso,
                             hipStream_t st) {
  if (vec4_ok(H, W)) {
    const size_t img = (size_t)H * W;
    const unsigned gx = (unsigned)((img / 4 + 255) / 256);
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {
      const unsigned nc = (unsigned)((C - c0) < 65535 ? (C - c0) : 65535);
      hipLaunchKernelGGL(ulpda_dual4_kernel, dim3(gx, nc), dim3(256), 0, st, xhat + c0 * img, y + c0 * 2 * img, H, W, mu, radius, iso);
    }
    return hipGetLastError();
  }
  hipLaunchKernelGGL(ulpda_dual_kernel, dim3(grid1d((size_t)H * W * C, 256)), dim3(256), 0, st, xhat, y, H, W, C, mu, radius, iso);
  return hipGetLastError();
}

bool ulpda_dual_rhs_supported(int H, int W) { return vec4_ok(H, W); }
hipError_t ulpda_dual_rhs(const float* xhat, const float* y_in, float* y_out, const float* x, const float* z, const float* htb, float* rhs, int64_t C,
                          int H, int W, float mu, float radius, int iso, float tau, float ts, hipStream_t st) {
  if (!vec4_ok(H, W) || y_in == y_out) return hipErrorInvalidConfiguration;
  const size_t img = (size_t)H * W;
  const unsigned gx = (unsigned)((img / 4 + 255) / 256);
  for (int64_t c0 = 0; c0 < C; c0 += 65535) {
    const unsigned nc = (unsigned)((C - c0) < 65535 ? (C - c0) : 65535);
    hipLaunchKernelGGL(ulpda_dual_rhs4_kernel, dim3(gx, nc), dim3(256), 0, st, xhat + c0 * img, y_in + c0 * 2 * img, y_out + c0 * 2 * img, x + c0 * img, z,
                       htb, rhs + c0 * img, H, W, mu, radius, iso, tau, ts);
  }
  return hipGetLastError();
}

hipError_t ulpda_rhs(const float* x, const float* y, const float* z, const float* htb, float* rhs, int64_t C, int H, int W,
                     float tau, float ts, hipStream_t st) {
  if (vec4_ok(H, W)) {
    const size_t img = (size_t)H * W;
    const unsigned gx = (unsigned)((img / 4 + 255) / 256);
    for (int64_t c0 = 0; c0 < C; c0 += 65535) {
      const unsigned nc = (unsigned)((C - c0) < 65535 ? (C - c0) : 65535);
      hipLaunchKernelGGL(ulpda_rhs4_kernel, dim3(gx, nc), dim3(256), 0, st, x + c0 * img, y + c0 * 2 * img, z, htb, rhs + c0 * img, H, W, tau, ts);
    }
    return hipGetLastError();
  }
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
  if ((((size_t)H * W) & 3) == 0) {
    hipLaunchKernelGGL(ulpda_finish4_kernel, dim3(grid1d(total / 4, 256)), dim3(256), 0, st, reinterpret_cast<float4*>(x),
                       reinterpret_cast<float4*>(xhat), reinterpret_cast<const float4*>(u), reinterpret_cast<const float4*>(xi), total / 4, s,
                       theta);
    return hipGetLastError();
  }
  hipLaunchKernelGGL(ulpda_finish_kernel, dim3(grid1d(total, 256)), dim3(256), 0, st, x, xhat, u, xi, total, s, theta);
  return hipGetLastError();
}

}  // namespace lmc
