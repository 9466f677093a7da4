// Metropolis adjustment of the MYULA proposal at image scale (MYMALA; generalises prox_lmc.py:134-158 to [C][H][W] states):
//   x' = m(x) + sqrt(2 tau) xi,   m(x) = (1 - tau/gamma) x - tau grad f(x) + (tau/gamma) prox_{gamma g}(x)   (prox_lmc.py:150)
//   log alpha = [f(x) + eps g(x)] - [f(x') + eps g(x')] - ( ||x - m(x')||^2 - ||x' - m(x)||^2 ) / (4 tau)           (:139-143)
//   accept if u <= min(1, alpha), u ~ U(0,1)                                                               (:152-154)
// The two means come from the fused step kernel (s = 0); these kernels are the glue: proposal + its squared norm,
// the per-chain decision (one thread per chain, Philox uniform), and the conditional per-chain copy.
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

// xp = mx + s * xi ;  d1[c] += sum (s * xi)^2
__global__ __launch_bounds__(256) void mala_propose_kernel(const float* __restrict__ mx, const float* __restrict__ xi,
                                                           float* __restrict__ xp, size_t img, float s, double* __restrict__ d1) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.x;
  double acc = 0.0;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x) {
    const float m = mx[c * img + k];
    const float p = fmaf(s, xi[c * img + k], m);
    xp[c * img + k] = p;
    const double d = (double)p - (double)m;
    acc += d * d;
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&d1[c], t);
}

// The same with the Philox field drawn in place (no xi buffer, no separate noise pass): one thread = one quad (4 rows of a column),
// the counter layout of every step kernel (ctr = (quad, iteration, global chain, stream tag), key = seed).
__global__ __launch_bounds__(256) void mala_propose_philox_kernel(const float* __restrict__ mx, float* __restrict__ xp, int H, int W, float s,
                                                                  uint32_t key0, uint32_t key1, uint32_t iteration, uint32_t chain_offset,
                                                                  double* __restrict__ d1) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.x;
  const size_t img = (size_t)H * W;
  const int nq = (H + 3) >> 2;
  const size_t total = (size_t)nq * W;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.y * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.y * blockDim.x) {
    const int q = (int)(i / W), col = (int)(i - (size_t)q * W);
    float n[4];
    quad_normals(key0, key1, iteration, chain_offset + (uint32_t)c, (uint32_t)q * (uint32_t)W + (uint32_t)col, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * q + j;
      if (r < H) {
        const size_t k = c * img + (size_t)r * W + col;
        const float m = mx[k];
        const float p = fmaf(s, n[j], m);
        xp[k] = p;
        const double d = (double)p - (double)m;
        acc += d * d;
      }
    }
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&d1[c], t);
}

hipError_t mala_propose_philox(const float* mx, float* xp, int64_t C, int H, int W, float s, uint32_t key0, uint32_t key1,
                               uint32_t iteration, uint32_t chain_offset, double* d1, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d1, 0, sizeof(double) * C, st);
  if (e != hipSuccess) return e;
  const size_t total = (size_t)((H + 3) / 4) * W;
  int gx = (int)((total + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(mala_propose_philox_kernel, dim3((unsigned)C, gx), dim3(256), 0, st, mx, xp, H, W, s, key0, key1, iteration, chain_offset, d1);
  return hipGetLastError();
}

hipError_t mala_propose(const float* mx, const float* xi, float* xp, int64_t C, size_t img, float s, double* d1, hipStream_t st) {
  hipError_t e = hipMemsetAsync(d1, 0, sizeof(double) * C, st);
  if (e != hipSuccess) return e;
  int gx = (int)((img + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(mala_propose_kernel, dim3((unsigned)C, gx), dim3(256), 0, st, mx, xi, xp, img, s, d1);
  return hipGetLastError();
}

// One thread per chain.  u = u01(first word of Philox(ctr = (0, iteration, global chain, kPhiloxAccept), key = seed)).
__global__ void mala_accept_kernel(int C, double* __restrict__ U, const double* __restrict__ fp, const double* __restrict__ gp, float epsg,
                                   const double* __restrict__ d1, const double* __restrict__ d2, float tau, uint32_t key0,
                                   uint32_t key1, uint32_t iteration, uint32_t chain_offset, int* __restrict__ flag,
                                   unsigned long long* __restrict__ nacc, double* __restrict__ log_alpha) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const double Up = fp[c] + (double)epsg * gp[c];      // U = f + epsg g
  const double la = (U[c] - Up) - (d2[c] - d1[c]) / (4.0 * (double)tau);
  uint32_t o[4];
  philox4x32_10(0u, iteration, chain_offset + (uint32_t)c, kPhiloxAccept, key0, key1, o);
  const double u = (double)u01(o[0]);
  const int acc = (log(u) <= la) ? 1 : 0;      // NaN (non-finite energies) rejects
  flag[c] = acc;
  if (acc) { U[c] = Up; nacc[c] += 1ull; }
  log_alpha[c] = la;
}

hipError_t mala_accept(int C, double* U, const double* fp, const double* gp, float epsg, const double* d1, const double* d2, float tau,
                       uint32_t key0, uint32_t key1, uint32_t iteration, uint32_t chain_offset, int* flag,
                       unsigned long long* nacc, double* log_alpha, hipStream_t st) {
  hipLaunchKernelGGL(mala_accept_kernel, dim3((C + 127) / 128), dim3(128), 0, st, C, U, fp, gp, epsg, d1, d2, tau, key0, key1, iteration,
                     chain_offset, flag, nacc, log_alpha);
  return hipGetLastError();
}

// chains whose flag equals `when`: x <- xp, mx <- mxp   (when = 1: accepted chains take the proposal; when = 0, after the caller has swapped
// the roles of the buffer pairs: rejected chains get their old state back -- the cheaper direction once most proposals are accepted)
__global__ __launch_bounds__(256) void mala_select_kernel(const int* __restrict__ flag, float4* __restrict__ x, float4* __restrict__ mx,
                                                          const float4* __restrict__ xp, const float4* __restrict__ mxp, size_t img4, int when) {
  const size_t c = blockIdx.x;
  if ((flag[c] != 0) != (when != 0)) return;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img4; k += (size_t)gridDim.y * blockDim.x) {
    x[c * img4 + k] = xp[c * img4 + k];
    mx[c * img4 + k] = mxp[c * img4 + k];
  }
}

__global__ __launch_bounds__(256) void mala_select1_kernel(const int* __restrict__ flag, float* __restrict__ x, float* __restrict__ mx,
                                                           const float* __restrict__ xp, const float* __restrict__ mxp, size_t img, int when) {
  const size_t c = blockIdx.x;
  if ((flag[c] != 0) != (when != 0)) return;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x) {
    x[c * img + k] = xp[c * img + k];
    mx[c * img + k] = mxp[c * img + k];
  }
}

hipError_t mala_select(const int* flag, float* x, float* mx, const float* xp, const float* mxp, int64_t C, size_t img, int when, hipStream_t st) {
  if ((img & 3) == 0) {
    int gx = (int)((img / 4 + 255) / 256);
    if (gx > 64) gx = 64;
    hipLaunchKernelGGL(mala_select_kernel, dim3((unsigned)C, gx), dim3(256), 0, st, flag, reinterpret_cast<float4*>(x), reinterpret_cast<float4*>(mx),
                       reinterpret_cast<const float4*>(xp), reinterpret_cast<const float4*>(mxp), img / 4, when);
    return hipGetLastError();
  }
  int gx = (int)((img + 255) / 256);
  if (gx > 64) gx = 64;
  hipLaunchKernelGGL(mala_select1_kernel, dim3((unsigned)C, gx), dim3(256), 0, st, flag, x, mx, xp, mxp, img, when);
  return hipGetLastError();
}

}  // namespace lmc
