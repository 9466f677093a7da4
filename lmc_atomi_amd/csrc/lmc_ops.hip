// Stand-alone operator kernels of the plugin surface + diagnostics (moments, energies, noise dump).
// These run outside the per-iteration hot loop (or once per kept sample) and are plain
// coalesced global-memory kernels; the hot loop is lmc_step_*.hip.
#include "lmc_device.h"
#include "lmc_launch.h"

namespace lmc {

// ---- H x / H^T x  (Convolve2D.matvec / rmatvec) -------------------------------------------
__global__ __launch_bounds__(256) void blur_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                   int H, int W, BlurTaps T, int adjoint) {
  const int gc = blockIdx.x * 64 + (threadIdx.x & 63);
  const int gr = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (gc >= W || gr >= H) return;
  const size_t img = (size_t)H * W;
  const float* xi = x + (size_t)blockIdx.z * img;
  float acc = 0.f;
  for (int a = 0; a < T.kh; ++a)
    for (int b = 0; b < T.kw; ++b) {
      const int r = adjoint ? gr + a - T.oy : gr - a + T.oy;
      const int c = adjoint ? gc + b - T.ox : gc - b + T.ox;
      if (r >= 0 && r < H && c >= 0 && c < W) acc = fmaf(T.h[a * T.kw + b], xi[(size_t)r * W + c], acc);
    }
  out[(size_t)blockIdx.z * img + (size_t)gr * W + gc] = acc;
}

hipError_t launch_blur(const float* x, float* out, int64_t n_img, int H, int W, const BlurTaps& T, int adjoint,
                       hipStream_t st) {
  const int64_t zmax = 65535;
  for (int64_t z0 = 0; z0 < n_img; z0 += zmax) {
    const int nz = (int)((n_img - z0) < zmax ? (n_img - z0) : zmax);
    dim3 grid((W + 63) / 64, (H + 3) / 4, nz);
    hipLaunchKernelGGL(blur_kernel, grid, dim3(256), 0, st, x + z0 * (size_t)H * W, out + z0 * (size_t)H * W, H, W, T,
                       adjoint);
  }
  return hipGetLastError();
}

// ---- forward-difference gradient and its adjoint (pylops.Gradient, edge=False, forward) ----
__global__ __launch_bounds__(256) void gradient_kernel(const float* __restrict__ x, float* __restrict__ out, int H,
                                                       int W, int64_t n_img) {
  const size_t img = (size_t)H * W;
  const size_t total = img * n_img;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t im = i / img, p = i - im * img;
    const int r = (int)(p / W), c = (int)(p - (size_t)r * W);
    const float v = x[i];
    out[im * 2 * img + p] = (r + 1 < H) ? x[i + W] - v : 0.f;
    out[im * 2 * img + img + p] = (c + 1 < W) ? x[i + 1] - v : 0.f;
  }
}

__global__ __launch_bounds__(256) void gradient_adjoint_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                               int H, int W, int64_t n_img) {
  const size_t img = (size_t)H * W;
  const size_t total = img * n_img;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t im = i / img, p = i - im * img;
    const int r = (int)(p / W), c = (int)(p - (size_t)r * W);
    const float* yr = y + im * 2 * img;
    const float* yc = yr + img;
    // A^T y = -div(y): -(yr[r]-yr[r-1]) - (yc[c]-yc[c-1]) with the last row/col of y ignored
    float acc = 0.f;
    if (r + 1 < H) acc -= yr[p];
    if (r > 0) acc += yr[p - W];
    if (c + 1 < W) acc -= yc[p];
    if (c > 0) acc += yc[p - 1];
    out[i] = acc;
  }
}

static inline int grid_for(size_t total, int block) {
  size_t g = (total + block - 1) / block;
  return (int)(g < 8192 ? (g ? g : 1) : 8192);
}

hipError_t launch_gradient(const float* x, float* out, int64_t n_img, int H, int W, bool adjoint, hipStream_t st) {
  const size_t total = (size_t)H * W * n_img;
  if (adjoint)
    hipLaunchKernelGGL(gradient_adjoint_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, x, out, H, W, n_img);
  else
    hipLaunchKernelGGL(gradient_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, x, out, H, W, n_img);
  return hipGetLastError();
}

// ---- dual projections (L21.proxdual / L1.proxdual) -----------------------------------------
__global__ __launch_bounds__(256) void dual_project_kernel(const float* __restrict__ y, float* __restrict__ out,
                                                           size_t img, int64_t n_img, float radius, int iso) {
  const size_t total = img * n_img;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const size_t im = i / img, p = i - im * img;
    const float a = y[im * 2 * img + p], b = y[im * 2 * img + img + p];
    float oa, ob;
    if (iso) {
      const float nrm = sqrtf(fmaf(a, a, b * b));
      const float sc = 1.f / fmaxf(1.f, nrm / radius);
      oa = a * sc; ob = b * sc;
    } else {
      oa = fminf(fmaxf(a, -radius), radius);
      ob = fminf(fmaxf(b, -radius), radius);
    }
    out[im * 2 * img + p] = oa;
    out[im * 2 * img + img + p] = ob;
  }
}

hipError_t launch_dual_project(const float* y, float* out, int64_t n_img, int H, int W, float radius, int iso,
                               hipStream_t st) {
  const size_t img = (size_t)H * W;
  hipLaunchKernelGGL(dual_project_kernel, dim3(grid_for(img * n_img, 256)), dim3(256), 0, st, y, out, img, n_img,
                     radius, iso);
  return hipGetLastError();
}

// ---- closed-form elementwise proxes (prox.py:9-85) -----------------------------------------
__global__ __launch_bounds__(256) void eprox_kernel(int kind, const float* __restrict__ x, float* __restrict__ out,
                                                    size_t n, EproxParams q) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    out[i] = eprox(kind, x[i], q);
}

hipError_t launch_eprox(int kind, const float* x, float* out, int64_t n, float p0, float p1, hipStream_t st) {
  hipLaunchKernelGGL(eprox_kernel, dim3(grid_for((size_t)n, 256)), dim3(256), 0, st, kind, x, out, (size_t)n,
                     EproxParams{p0, p1});
  return hipGetLastError();
}

// Closed-form prior prox with an ARRAY-valued prox parameter (array epsg, algs.py:509,569): t(c, i) = pt * scale[c * cs + i * ps];
// prior 1 = l2: x / (1 + t sigma); 2 = l1: soft(x, t sigma); 6 = prox.py closed form with the parameters the mask names multiplied by t.
__global__ __launch_bounds__(256) void prior_prox_scaled_kernel(int prior, int kind, const float* __restrict__ x, float* __restrict__ out, size_t img, size_t n,
                                                               const float* __restrict__ scale, size_t cs, size_t ps, float pt, float sigma, float p0,
                                                               float p1, int mask) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t c = i / img, px = i - c * img;
    const float t = pt * scale[c * cs + px * ps];
    const float v = x[i];
    float o = v;
    if (prior == LMC_PRIOR_L2) o = v * (1.f / (1.f + t * sigma));
    else if (prior == LMC_PRIOR_L1) o = copysignf(fmaxf(fabsf(v) - t * sigma, 0.f), v);
    else if (prior == LMC_PRIOR_EPROX) o = eprox(kind, v, EproxParams{(mask & 1) ? t * p0 : p0, (mask & 2) ? t * p1 : p1});
    out[i] = o;
  }
}

hipError_t launch_prior_prox_scaled(int prior, int kind, const float* x, float* out, int64_t n_chains, int64_t img, const float* scale, int64_t cs, int64_t ps,
                                    float pt, float sigma, float p0, float p1, int mask, hipStream_t st) {
  const size_t n = (size_t)n_chains * (size_t)img;
  hipLaunchKernelGGL(prior_prox_scaled_kernel, dim3(grid_for(n, 256)), dim3(256), 0, st, prior, kind, x, out, (size_t)img, n, scale, (size_t)cs, (size_t)ps, pt,
                     sigma, p0, p1, mask);
  return hipGetLastError();
}

// ---- posterior moments: sum_c x, sum_c x^2 into fp64 accumulators --------------------------
// grid.x covers pixels (one per thread, coalesced over chains), grid.y = chain segments.
__global__ __launch_bounds__(256) void moments_kernel(const float* __restrict__ x, int C, size_t img, int seg_len,
                                                      double* __restrict__ s1, double* __restrict__ s2) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= img) return;
  const int c0 = blockIdx.y * seg_len;
  const int c1 = min(C, c0 + seg_len);
  double a = 0.0, b = 0.0;
  for (int c = c0; c < c1; ++c) {
    const double v = (double)x[(size_t)c * img + p];
    a += v;
    b = fma(v, v, b);
  }
  unsafeAtomicAdd(&s1[p], a);
  unsafeAtomicAdd(&s2[p], b);
}

// 4 pixels per thread (b128 loads), chains unrolled by 8: 8 independent 16-byte loads in flight per lane.
__global__ __launch_bounds__(256) void moments4_kernel(const float* __restrict__ x, int C, size_t img, int seg_len,
                                                       double* __restrict__ s1, double* __restrict__ s2) {
  const size_t p = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
  const int c0 = blockIdx.y * seg_len;
  const int c1 = p < img ? min(C, c0 + seg_len) : c0;      // threads past the image keep zeros (they still take part in the shuffles)
  double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
  const float* __restrict__ src = x + p;
  int c = c0;
  for (; c + 8 <= c1; c += 8) {
    float4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(c + u) * img);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double d0 = v[u].x, d1 = v[u].y, d2 = v[u].z, d3 = v[u].w;
      a[0] += d0; a[1] += d1; a[2] += d2; a[3] += d3;
      b[0] = fma(d0, d0, b[0]); b[1] = fma(d1, d1, b[1]); b[2] = fma(d2, d2, b[2]); b[3] = fma(d3, d3, b[3]);
    }
  }
  for (; c < c1; ++c) {
    const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * img);
    const double d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
    a[0] += d0; a[1] += d1; a[2] += d2; a[3] += d3;
    b[0] = fma(d0, d0, b[0]); b[1] = fma(d1, d1, b[1]); b[2] = fma(d2, d2, b[2]); b[3] = fma(d3, d3, b[3]);
  }
  // lane-contiguous atomics (a wave's 64 adds hit 512 contiguous bytes): the 4-pixel groups are transposed inside the wave with
  // ds_bpermute shuffles -- no LDS allocation, so the kernel can share a CU with the step kernel (whose LDS is full) when the two
  // run concurrently on different streams.  Atomic k of lane l covers pixel 64*k + l of the wave's 256: component l & 3 of lane 16*k + l/4.
  const int lane = threadIdx.x & 63;
  const size_t pw = ((size_t)blockIdx.x * blockDim.x + (threadIdx.x & ~63)) * 4;     // first pixel of this wave
  const int sel = lane & 3;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int srcl = 16 * k + (lane >> 2);
    double va = 0.0, vb = 0.0;
#pragma unroll
    for (int m = 0; m < 4; ++m) {            // the value of component m travels from lane srcl; each lane keeps the one it needs
      const double ta = __shfl(a[m], srcl, 64), tb = __shfl(b[m], srcl, 64);
      if (sel == m) { va = ta; vb = tb; }
    }
    const size_t q = pw + (size_t)k * 64 + lane;
    if (q < img) {
      unsafeAtomicAdd(&s1[q], va);
      unsafeAtomicAdd(&s2[q], vb);
    }
  }
}

// Background variant for the side stream: few workgroups, each walking several 1024-pixel groups, so that the reduction trickles
// along under the (VALU-bound, load-latency-sensitive) step kernel instead of saturating HBM for 0.2 ms.
__global__ __launch_bounds__(256) void moments4_bg_kernel(const float* __restrict__ x, int C, size_t img, int n_groups,
                                                          double* __restrict__ s1, double* __restrict__ s2) {
  const int lane = threadIdx.x & 63;
  const int sel = lane & 3;
  // gridDim.y > 1: the chains are split into that many segments (small configurations: more workgroups than 1024-pixel groups)
  const int seg_len = (C + (int)gridDim.y - 1) / (int)gridDim.y;
  const int cs = (int)blockIdx.y * seg_len, ce = min(C, cs + seg_len);
  for (int g = blockIdx.x; g < n_groups; g += gridDim.x) {
    const size_t p = ((size_t)g * 256 + threadIdx.x) * 4;
    const int c1 = p < img ? ce : cs;
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    const float* __restrict__ src = x + p;
    int c = cs;
    for (; c + 4 <= c1; c += 4) {
      float4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)(c + u) * img);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double d0 = v[u].x, d1 = v[u].y, d2 = v[u].z, d3 = v[u].w;
        a[0] += d0; a[1] += d1; a[2] += d2; a[3] += d3;
        b[0] = fma(d0, d0, b[0]); b[1] = fma(d1, d1, b[1]); b[2] = fma(d2, d2, b[2]); b[3] = fma(d3, d3, b[3]);
      }
    }
    for (; c < c1; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(src + (size_t)c * img);
      const double d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
      a[0] += d0; a[1] += d1; a[2] += d2; a[3] += d3;
      b[0] = fma(d0, d0, b[0]); b[1] = fma(d1, d1, b[1]); b[2] = fma(d2, d2, b[2]); b[3] = fma(d3, d3, b[3]);
    }
    const size_t pw = ((size_t)g * 256 + (threadIdx.x & ~63)) * 4;     // first pixel of this wave (same transpose as moments4_kernel)
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int srcl = 16 * k + (lane >> 2);
      double va = 0.0, vb = 0.0;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const double ta = __shfl(a[m], srcl, 64), tb = __shfl(b[m], srcl, 64);
        if (sel == m) { va = ta; vb = tb; }
      }
      const size_t q = pw + (size_t)k * 64 + lane;
      if (q < img) {
        unsafeAtomicAdd(&s1[q], va);
        unsafeAtomicAdd(&s2[q], vb);
      }
    }
  }
}

// same result as launch_moments, paced: `n_wg` workgroups in total (H*W % 4 == 0 required, else falls back to launch_moments)
hipError_t launch_moments_bg(const float* x, int C, int H, int W, double* s1, double* s2, int n_wg, hipStream_t st) {
  const size_t img = (size_t)H * W;
  if ((img & 3) || n_wg < 1) return launch_moments(x, C, H, W, s1, s2, st);
  const int n_groups = (int)((img / 4 + 255) / 256);
  int nseg = 1;
  if (n_wg > n_groups) {        // fewer pixel groups than workgroups asked for: split the chains as well (>= 16 chains per segment)
    while (n_groups * nseg * 2 <= n_wg && nseg * 32 <= C) nseg *= 2;
    n_wg = n_groups;
  }
  hipLaunchKernelGGL(moments4_bg_kernel, dim3(n_wg, nseg), dim3(256), 0, st, x, C, img, n_groups, s1, s2);
  return hipGetLastError();
}

hipError_t launch_moments(const float* x, int C, int H, int W, double* s1, double* s2, hipStream_t st) {
  const size_t img = (size_t)H * W;
  if ((img & 3) == 0) {
    const int gx4 = (int)((img / 4 + 255) / 256);
    int nseg = 1;
    while ((size_t)gx4 * nseg < 2048 && nseg * 16 <= C) nseg *= 2;
    const int seg_len = (C + nseg - 1) / nseg;
    hipLaunchKernelGGL(moments4_kernel, dim3(gx4, nseg), dim3(256), 0, st, x, C, img, seg_len, s1, s2);
    return hipGetLastError();
  }
  const int gx = (int)((img + 255) / 256);
  int nseg = 1;
  while ((size_t)gx * nseg < 2048 && nseg * 8 <= C) nseg *= 2;
  const int seg_len = (C + nseg - 1) / nseg;
  hipLaunchKernelGGL(moments_kernel, dim3(gx, nseg), dim3(256), 0, st, x, C, img, seg_len, s1, s2);
  return hipGetLastError();
}

// ---- per-image energies f(x), g(x): wave-shuffle + LDS block reduction, one atomic per block -

__global__ __launch_bounds__(256) void energy_kernel(const float* __restrict__ x, EnergyArgs E, double* __restrict__ f_out,
                                                     double* __restrict__ g_out) {
  __shared__ double scratch[4];
  const int H = E.H, W = E.W;
  const size_t img = (size_t)H * W;
  const float* xi = x + (size_t)blockIdx.y * img;
  double fa = 0.0, ga = 0.0, na = 0.0;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < img; p += (size_t)gridDim.x * blockDim.x) {
    const int r = (int)(p / W), c = (int)(p - (size_t)r * W);
    const float v = xi[p];
    float res = 0.f;
    if (E.data_kind == LMC_DATA_BLUR) {
      float acc = 0.f;
      for (int a = 0; a < E.blur.kh; ++a)
        for (int b = 0; b < E.blur.kw; ++b) {
          const int rr = r - a + E.blur.oy, cc = c - b + E.blur.ox;
          if (rr >= 0 && rr < H && cc >= 0 && cc < W) acc = fmaf(E.blur.h[a * E.blur.kw + b], xi[(size_t)rr * W + cc], acc);
        }
      res = acc - E.y[p];
    } else if (E.data_kind == LMC_DATA_IDENTITY) {
      res = v - E.y[p];
    } else if (E.data_kind == LMC_DATA_MASK) {
      res = E.mask[p] * v - E.y[p];
    }
    fa += (double)res * (double)res;
    if (E.ncvx_kind == LMC_NCVX_MC_TV) {   // Moreau envelope of |.| at |grad x|: Huber, subtracted with weight lambda
      const float dx = (r + 1 < H) ? xi[p + W] - v : 0.f;
      const float dy = (c + 1 < W) ? xi[p + 1] - v : 0.f;
      na += mc_tv_envelope(dx, dy, E.ncvx_gamma);
    }
    if (E.prior_kind == LMC_PRIOR_TV_ISO) {
      const float dx = (r + 1 < H) ? xi[p + W] - v : 0.f;
      const float dy = (c + 1 < W) ? xi[p + 1] - v : 0.f;
      ga += (double)sqrtf(fmaf(dx, dx, dy * dy));
    } else if (E.prior_kind == LMC_PRIOR_TV_ANISO) {
      const float dx = (r + 1 < H) ? xi[p + W] - v : 0.f;
      const float dy = (c + 1 < W) ? xi[p + 1] - v : 0.f;
      ga += (double)fabsf(dx) + (double)fabsf(dy);
    } else if (E.prior_kind == LMC_PRIOR_L1) {
      ga += (double)fabsf(v);
    } else if (E.prior_kind == LMC_PRIOR_L2) {
      ga += 0.5 * (double)v * (double)v;
    }
  }
  const double ft = block_sum(fa, scratch);
  const double gt = block_sum(ga, scratch);
  const double nt = E.ncvx_kind != LMC_NCVX_NONE ? block_sum(na, scratch) : 0.0;
  if (threadIdx.x == 0) {
    if (f_out) unsafeAtomicAdd(&f_out[blockIdx.y], 0.5 * (double)E.sigma_f * ft - (double)E.ncvx_lambda * nt);
    if (g_out) unsafeAtomicAdd(&g_out[blockIdx.y], (double)E.prior_sigma * gt);
  }
}

// Separable, centred blur (KT = 5 or 7 taps: every kernel of the reference's experiments): one workgroup = one 32 x 64 tile of
// one image; the tile plus a halo of HW (+1 row / column for the TV differences) is staged in LDS once, the blur runs as a
// horizontal and a vertical 1-D pass, the other terms read the same LDS tile.  ~6x faster than the direct 2-D loop above.
constexpr int kEnTH = 32, kEnTW = 64;
template <int KT>
__global__ __launch_bounds__(256) void energy_sep_kernel(const float* __restrict__ x, EnergyArgs E, int tiles_x, int tiles_y,
                                                         double* __restrict__ f_out, double* __restrict__ g_out) {
  constexpr int HW = (KT - 1) / 2, HL = HW > 1 ? HW : 1;
  constexpr int PW = kEnTW + 2 * HL, PH = kEnTH + 2 * HL;
  __shared__ float xs[PH * PW];
  __shared__ float hs[PH * kEnTW];
  __shared__ double scratch[4];
  const int H = E.H, W = E.W;
  const size_t img = (size_t)H * W;
  const int tiles = tiles_x * tiles_y;
  const int im = blockIdx.x / tiles, tile = blockIdx.x - im * tiles;
  const int ty0 = (tile / tiles_x) * kEnTH, tx0 = (tile % tiles_x) * kEnTW;
  const float* __restrict__ xi = x + (size_t)im * img;
  const int tid = threadIdx.x;
  for (int e = tid; e < PH * PW; e += 256) {
    const int r = e / PW, c = e - r * PW;
    const int gr = ty0 - HL + r, gc = tx0 - HL + c;
    xs[e] = (gr >= 0 && gr < H && gc >= 0 && gc < W) ? xi[(size_t)gr * W + gc] : 0.f;
  }
  __syncthreads();
  const float* __restrict__ uv = E.blur.h;   // centred: u[0..KT) then v[0..KT) at h[kMaxBlur..]
  for (int e = tid; e < PH * kEnTW; e += 256) {   // horizontal pass, all staged rows: hx[r][c] = sum_b v[b] x[r][c + HW - b]
    const int r = e / kEnTW, c = e - r * kEnTW;
    float acc = 0.f;
#pragma unroll
    for (int b = 0; b < KT; ++b) acc = fmaf(uv[kMaxBlur + b], xs[r * PW + HL + c + HW - b], acc);
    hs[e] = acc;
  }
  __syncthreads();
  double fa = 0.0, ga = 0.0, na = 0.0;
  const int tx = tid & (kEnTW - 1), tq = tid / kEnTW;      // column, row group (8 rows each)
  const int gc = tx0 + tx;
  if (gc < W) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int lr = tq * 8 + j, gr = ty0 + lr;
      if (gr >= H) break;
      float acc = 0.f;
#pragma unroll
      for (int a = 0; a < KT; ++a) acc = fmaf(uv[a], hs[(HL + lr + HW - a) * kEnTW + tx], acc);
      const float res = acc - E.y[(size_t)gr * W + gc];
      fa += (double)res * (double)res;
      const float v = xs[(HL + lr) * PW + HL + tx];
      const float dx = (gr + 1 < H) ? xs[(HL + lr + 1) * PW + HL + tx] - v : 0.f;
      const float dy = (gc + 1 < W) ? xs[(HL + lr) * PW + HL + tx + 1] - v : 0.f;
      if (E.ncvx_kind == LMC_NCVX_MC_TV) {
        na += mc_tv_envelope(dx, dy, E.ncvx_gamma);
      }
      if (E.prior_kind == LMC_PRIOR_TV_ISO) ga += (double)sqrtf(fmaf(dx, dx, dy * dy));
      else if (E.prior_kind == LMC_PRIOR_TV_ANISO) ga += (double)fabsf(dx) + (double)fabsf(dy);
      else if (E.prior_kind == LMC_PRIOR_L1) ga += (double)fabsf(v);
      else if (E.prior_kind == LMC_PRIOR_L2) ga += 0.5 * (double)v * (double)v;
    }
  }
  const double ft = block_sum(fa, scratch);
  __syncthreads();
  const double gt = block_sum(ga, scratch);
  __syncthreads();
  const double nt = E.ncvx_kind != LMC_NCVX_NONE ? block_sum(na, scratch) : 0.0;
  if (threadIdx.x == 0) {
    if (f_out) unsafeAtomicAdd(&f_out[im], 0.5 * (double)E.sigma_f * ft - (double)E.ncvx_lambda * nt);
    if (g_out) unsafeAtomicAdd(&g_out[im], (double)E.prior_sigma * gt);
  }
}

hipError_t launch_energies(const float* x, int64_t n_img, const EnergyArgs& E, double* f_out, double* g_out,
                           hipStream_t st) {
  hipError_t e;
  if (f_out && (e = hipMemsetAsync(f_out, 0, sizeof(double) * n_img, st)) != hipSuccess) return e;
  if (g_out && (e = hipMemsetAsync(g_out, 0, sizeof(double) * n_img, st)) != hipSuccess) return e;
  if (E.data_kind == LMC_DATA_BLUR) {
    float uc[kMaxBlur] = {0}, vc[kMaxBlur] = {0};
    const int KT = centred_blur_taps(E.blur, uc, vc);
    const int tiles_x = (E.W + kEnTW - 1) / kEnTW, tiles_y = (E.H + kEnTH - 1) / kEnTH;
    if (KT != 0 && (long long)tiles_x * tiles_y * n_img < 0x7fffffffLL) {
      EnergyArgs S = E;
      for (int i = 0; i < kMaxBlur; ++i) { S.blur.h[i] = i < KT ? uc[i] : 0.f; S.blur.h[kMaxBlur + i] = i < KT ? vc[i] : 0.f; }
      const dim3 grid((unsigned)((long long)tiles_x * tiles_y * n_img));
      if (KT == 5) hipLaunchKernelGGL(energy_sep_kernel<5>, grid, dim3(256), 0, st, x, S, tiles_x, tiles_y, f_out, g_out);
      else hipLaunchKernelGGL(energy_sep_kernel<7>, grid, dim3(256), 0, st, x, S, tiles_x, tiles_y, f_out, g_out);
      return hipGetLastError();
    }
  }
  const size_t img = (size_t)E.H * E.W;
  int gx = (int)((img + 255) / 256);
  if (gx > 64) gx = 64;
  const int64_t ymax = 65535;
  for (int64_t z0 = 0; z0 < n_img; z0 += ymax) {
    const int nz = (int)((n_img - z0) < ymax ? (n_img - z0) : ymax);
    hipLaunchKernelGGL(energy_kernel, dim3(gx, nz), dim3(256), 0, st, x + z0 * img, E, f_out ? f_out + z0 : nullptr,
                       g_out ? g_out + z0 : nullptr);
  }
  return hipGetLastError();
}

// ---- 3-level orthonormal Haar wavelet, l1 prox of the detail coefficients (BASELINE config 5's prior) -------------
// The 3-level transform acts on independent 8 x 8 blocks: one thread = one block, all 64 values in registers,
// in-place butterflies at strides 1, 2, 4; detail coefficients are soft-thresholded (MODE 0) or summed in absolute
// value (MODE 1: g(x) per image).  Adjacent lanes own adjacent blocks of a block row, so each of the 8 row loads of a
// wavefront is one contiguous 2 KiB segment (2 x float4 per lane).
__device__ __forceinline__ float soft_thr(float v, float t) { return copysignf(fmaxf(fabsf(v) - t, 0.f), v); }

template <int MODE>
__global__ __launch_bounds__(256) void haar_l1_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                      int64_t n_img, float thr, double* __restrict__ val, float sigma) {
  __shared__ double scratch[4];
  const int nbx = W >> 3, nby = H >> 3;
  const size_t blocks_per_img = (size_t)nbx * nby;
  const size_t img = (size_t)H * W;
  const size_t c = blockIdx.y;
  double acc = 0.0;
  for (size_t bi = (size_t)blockIdx.x * blockDim.x + threadIdx.x; bi < blocks_per_img; bi += (size_t)gridDim.x * blockDim.x) {
    const int by = (int)(bi / nbx), bx = (int)(bi - (size_t)by * nbx);
    const float* src = x + c * img + (size_t)(by * 8) * W + bx * 8;
    float v[8][8];
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const float4 lo = *reinterpret_cast<const float4*>(src + (size_t)r * W);
      const float4 hi = *reinterpret_cast<const float4*>(src + (size_t)r * W + 4);
      v[r][0] = lo.x; v[r][1] = lo.y; v[r][2] = lo.z; v[r][3] = lo.w; v[r][4] = hi.x; v[r][5] = hi.y; v[r][6] = hi.z; v[r][7] = hi.w;
    }
    float dsum = 0.f;
#pragma unroll
    for (int s = 1; s <= 4; s <<= 1) {      // forward: quads at stride s; (i, j) holds LL, (i, j+s) LH, (i+s, j) HL, (i+s, j+s) HH
#pragma unroll
      for (int i = 0; i < 8; i += 2 * s)
#pragma unroll
        for (int j = 0; j < 8; j += 2 * s) {
          const float a = v[i][j], b = v[i][j + s], cc = v[i + s][j], d = v[i + s][j + s];
          const float ll = 0.5f * (a + b + cc + d), lh = 0.5f * (a - b + cc - d), hl = 0.5f * (a + b - cc - d), hh = 0.5f * (a - b - cc + d);
          v[i][j] = ll;
          if (MODE == 0) { v[i][j + s] = soft_thr(lh, thr); v[i + s][j] = soft_thr(hl, thr); v[i + s][j + s] = soft_thr(hh, thr); }
          else dsum += fabsf(lh) + fabsf(hl) + fabsf(hh);
        }
    }
    if (MODE == 1) { acc += (double)dsum; continue; }
#pragma unroll
    for (int s = 4; s >= 1; s >>= 1) {      // inverse
#pragma unroll
      for (int i = 0; i < 8; i += 2 * s)
#pragma unroll
        for (int j = 0; j < 8; j += 2 * s) {
          const float ll = v[i][j], lh = v[i][j + s], hl = v[i + s][j], hh = v[i + s][j + s];
          v[i][j] = 0.5f * (ll + lh + hl + hh); v[i][j + s] = 0.5f * (ll - lh + hl - hh);
          v[i + s][j] = 0.5f * (ll + lh - hl - hh); v[i + s][j + s] = 0.5f * (ll - lh - hl + hh);
        }
    }
    float* dst = out + c * img + (size_t)(by * 8) * W + bx * 8;
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      *reinterpret_cast<float4*>(dst + (size_t)r * W) = make_float4(v[r][0], v[r][1], v[r][2], v[r][3]);
      *reinterpret_cast<float4*>(dst + (size_t)r * W + 4) = make_float4(v[r][4], v[r][5], v[r][6], v[r][7]);
    }
  }
  if (MODE == 1) {
    const double t = block_sum(acc, scratch);
    if (threadIdx.x == 0) unsafeAtomicAdd(&val[c], (double)sigma * t);
  }
}

// out = prox_{thr ||W_detail .||_1}(x)   (H, W multiples of 8)
hipError_t launch_haar_prox(const float* x, float* out, int64_t n_img, int H, int W, float thr, hipStream_t st) {
  const size_t nb = (size_t)(H >> 3) * (W >> 3);
  int gx = (int)((nb + 255) / 256);
  for (int64_t z0 = 0; z0 < n_img; z0 += 65535) {
    const int nz = (int)((n_img - z0) < 65535 ? (n_img - z0) : 65535);
    hipLaunchKernelGGL(haar_l1_kernel<0>, dim3(gx, nz), dim3(256), 0, st, x + z0 * (size_t)H * W, out + z0 * (size_t)H * W, H, W,
                       (int64_t)nz, thr, nullptr, 0.f);
  }
  return hipGetLastError();
}

// val[i] = sigma * sum |detail coefficients of Haar(x_i)|
hipError_t launch_haar_value(const float* x, int64_t n_img, int H, int W, float sigma, double* val, hipStream_t st) {
  hipError_t e = hipMemsetAsync(val, 0, sizeof(double) * n_img, st);
  if (e != hipSuccess) return e;
  const size_t nb = (size_t)(H >> 3) * (W >> 3);
  int gx = (int)((nb + 255) / 256);
  if (gx > 32) gx = 32;
  for (int64_t z0 = 0; z0 < n_img; z0 += 65535) {
    const int nz = (int)((n_img - z0) < 65535 ? (n_img - z0) : 65535);
    hipLaunchKernelGGL(haar_l1_kernel<1>, dim3(gx, nz), dim3(256), 0, st, x + z0 * (size_t)H * W, nullptr, H, W, (int64_t)nz, 0.f,
                       val + z0, sigma);
  }
  return hipGetLastError();
}

// ---- out += coef * A^T(A x / max(|A x|, gamma)): the MC-TV term of L2_ncvx_tv (algs.py:273-277) added to an update computed
// without it (register-block kernel + this pass instead of the row pipeline: inpainting + Haar-l1 + MC-TV) ----------
__global__ __launch_bounds__(256) void mc_tv_add_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W,
                                                        int tiles_x, int tiles_y, float coef, float gamma) {
  constexpr int TH = 32, TW = 64, PW = TW + 2, PH = TH + 2;
  __shared__ float xs[PH * PW];
  const size_t img = (size_t)H * W;
  const int tiles = tiles_x * tiles_y;
  const int im = blockIdx.x / tiles, tile = blockIdx.x - im * tiles;
  const int ty0 = (tile / tiles_x) * TH, tx0 = (tile % tiles_x) * TW;
  const float* __restrict__ xi = x + (size_t)im * img;
  float* __restrict__ oi = out + (size_t)im * img;
  for (int e = threadIdx.x; e < PH * PW; e += 256) {
    const int r = e / PW, c = e - r * PW;
    const int gr = ty0 - 1 + r, gc = tx0 - 1 + c;
    xs[e] = (gr >= 0 && gr < H && gc >= 0 && gc < W) ? xi[(size_t)gr * W + gc] : 0.f;
  }
  __syncthreads();
  const int tx = threadIdx.x & (TW - 1), tq = threadIdx.x / TW;
  const int gc = tx0 + tx;
  if (gc >= W) return;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int lr = tq * 8 + j, gr = ty0 + lr;
    if (gr >= H) break;
    const float* p = xs + (lr + 1) * PW + tx + 1;
    const float g = mc_tv_grad(p[-PW], p[-PW + 1], p[-1], p[0], p[1], p[PW - 1], p[PW], gr > 0, gr + 1 < H, gc > 0, gc + 1 < W, gamma);
    oi[(size_t)gr * W + gc] = fmaf(coef, g, oi[(size_t)gr * W + gc]);
  }
}

hipError_t launch_mc_tv_add(const float* x, float* out, int64_t n_img, int H, int W, float coef, float gamma, hipStream_t st) {
  const int tiles_x = (W + 63) / 64, tiles_y = (H + 31) / 32;
  const long long nb = (long long)tiles_x * tiles_y * n_img;
  if (nb > 0x7fffffffLL) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(mc_tv_add_kernel, dim3((unsigned)nb), dim3(256), 0, st, x, out, H, W, tiles_x, tiles_y, coef, gamma);
  return hipGetLastError();
}

// ---- noise dump: the field xi[C][H][W] the step kernels draw at `iteration` -----------------
__global__ __launch_bounds__(256) void noise_kernel(float* __restrict__ out, int C, int H, int W, uint32_t key0,
                                                    uint32_t key1, uint32_t iteration, uint32_t chain_offset) {
  const int nq = (H + 3) >> 2;
  const size_t total = (size_t)C * nq * W;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int col = (int)(i % W);
    const size_t t = i / W;
    const int q = (int)(t % nq), c = (int)(t / nq);
    float n[4];
    quad_normals(key0, key1, iteration, chain_offset + (uint32_t)c, (uint32_t)q * (uint32_t)W + (uint32_t)col, n);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * q + j;
      if (r < H) out[((size_t)c * H + r) * W + col] = n[j];
    }
  }
}

hipError_t launch_noise(float* out, int C, int H, int W, uint32_t key0, uint32_t key1, uint32_t iteration,
                        uint32_t chain_offset, hipStream_t st) {
  const size_t total = (size_t)C * ((H + 3) / 4) * W;
  hipLaunchKernelGGL(noise_kernel, dim3(grid_for(total, 256)), dim3(256), 0, st, out, C, H, W, key0, key1, iteration,
                     chain_offset);
  return hipGetLastError();
}

// ---- chain probes: every image reduced to a ph x pw grid of block means (convergence diagnostics across chains) ----
// One workgroup per (probe row, image): column sums of the band in double (coalesced row reads), then one thread per probe.
__global__ void __launch_bounds__(256) chain_probe_kernel(const float* __restrict__ x, float* __restrict__ out, int H, int W, int ph, int pw) {
  extern __shared__ double probe_colsum[];
  const int a = blockIdx.x;
  const size_t img = blockIdx.y;
  const int r0 = (int)((int64_t)a * H / ph), r1 = (int)((int64_t)(a + 1) * H / ph);
  const float* xi = x + img * (size_t)H * W;
  for (int col = threadIdx.x; col < W; col += 256) {
    double s = 0.0;
    for (int r = r0; r < r1; ++r) s += (double)xi[(size_t)r * W + col];
    probe_colsum[col] = s;
  }
  __syncthreads();
  for (int b = threadIdx.x; b < pw; b += 256) {
    const int c0 = (int)((int64_t)b * W / pw), c1 = (int)((int64_t)(b + 1) * W / pw);
    double s = 0.0;
    for (int c = c0; c < c1; ++c) s += probe_colsum[c];
    out[(img * ph + a) * pw + b] = (float)(s / ((double)(r1 - r0) * (double)(c1 - c0)));
  }
}

// out[i, a, b] = mean of x_i over rows [a H/ph, (a+1) H/ph) x columns [b W/pw, (b+1) W/pw)   (1 <= ph <= H, 1 <= pw <= W)
hipError_t launch_chain_probes(const float* x, float* out, int64_t n_img, int H, int W, int ph, int pw, hipStream_t st) {
  for (int64_t z0 = 0; z0 < n_img; z0 += 65535) {
    const int nz = (int)((n_img - z0) < 65535 ? (n_img - z0) : 65535);
    hipLaunchKernelGGL(chain_probe_kernel, dim3(ph, nz), dim3(256), (size_t)W * sizeof(double), st, x + z0 * (size_t)H * W,
                       out + z0 * (size_t)ph * pw, H, W, ph, pw);
  }
  return hipGetLastError();
}

// ---- HBM streaming probe: the measured bandwidth the roofline fractions are quoted beside ("confirm on the box with a
// copy probe and report both").  A state-shaped copy: read one buffer, write another, 16 bytes per lane.
__global__ __launch_bounds__(256) void hbm_copy_probe_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n4, float a) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = x[i];
    v.x *= a; v.y *= a; v.z *= a; v.w *= a;
    y[i] = v;
  }
}

// two independent 16-byte loads in flight per lane and iteration
__global__ __launch_bounds__(256) void hbm_copy_probe2_kernel(const float4* __restrict__ x, float4* __restrict__ y, size_t n4, float a) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + stride < n4; i += 2 * stride) {
    float4 v = x[i], w = x[i + stride];
    v.x *= a; v.y *= a; v.z *= a; v.w *= a;
    w.x *= a; w.y *= a; w.z *= a; w.w *= a;
    y[i] = v;
    y[i + stride] = w;
  }
  if (i < n4) { float4 v = x[i]; v.x *= a; v.y *= a; v.z *= a; v.w *= a; y[i] = v; }
}

// shape: 0..4 = grid-stride copy with 1024 << shape workgroups; 5..9 = the same grids with two loads in flight per lane
int hbm_copy_probe_shapes() { return 10; }
hipError_t launch_hbm_copy_probe(const float* x, float* y, size_t n_floats, int shape, hipStream_t st) {
  const int grid = 1024 << (shape % 5);
  if (shape < 5)
    hipLaunchKernelGGL(hbm_copy_probe_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y),
                       n_floats / 4, 1.0f);
  else
    hipLaunchKernelGGL(hbm_copy_probe2_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<const float4*>(x), reinterpret_cast<float4*>(y),
                       n_floats / 4, 1.0f);
  return hipGetLastError();
}

// ---- pyproximal.TV's early exit (lmc_problem.tv_rtol > 0): the pieces of the exact, pass-by-pass path (lmc_capi.hip: tv_prox_rtol) -------
// obj[c] += 1/2 ||x_c - sol_c||^2 + gam * TV_iso(sol_c) for the chains still iterating (flag[c] < 0): the primal objective upstream
// evaluates at the top of every loop pass.  fp32 terms, fp64 sums.
__global__ __launch_bounds__(256) void tv_objective_kernel(const float* __restrict__ x, const float* __restrict__ sol, int H, int W, float gam,
                                                           const int* __restrict__ flag, double* __restrict__ obj) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.x;
  if (flag[c] >= 0) return;                    // uniform over the block
  const size_t img = (size_t)H * W;
  const float* xc = x + c * img;
  const float* sc = sol + c * img;
  double acc = 0.0;
  for (size_t p = (size_t)blockIdx.y * blockDim.x + threadIdx.x; p < img; p += (size_t)gridDim.y * blockDim.x) {
    const int r = (int)(p / W), col = (int)(p - (size_t)r * W);
    const float v = sc[p];
    const float d = xc[p] - v;
    const float dx = (r + 1 < H) ? sc[p + W] - v : 0.f;
    const float dy = (col + 1 < W) ? sc[p + 1] - v : 0.f;
    acc += 0.5 * (double)d * (double)d + (double)gam * (double)sqrtf(fmaf(dx, dx, dy * dy));
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&obj[c], t);
}

// One thread per chain, pass j >= 1: a chain still iterating leaves when |obj_j - obj_{j-1}| / obj_j < rtol (the CPU checker's rule, tv_prox_fgp --
// upstream's loop); flag[c] = j marks it.  Otherwise obj_j becomes its obj_{j-1}.  n_active counts the rest.
__global__ void tv_rtol_decide_kernel(int64_t n, double* __restrict__ prev, double* __restrict__ cur, int* __restrict__ flag, int pass, double rtol,
                                      int* __restrict__ n_active) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n || flag[c] >= 0) return;
  const double o = cur[c], p = prev[c];
  const double rel = o > 0.0 ? fabs(o - p) / o : 2.0 * rtol;
  if (pass >= 1 && rel < rtol) { flag[c] = pass; }
  else { prev[c] = o; atomicAdd(n_active, 1); }
  cur[c] = 0.0;
}

// sol_c <- tmp_c for the chains whose flag equals `pass` (the pass they left in; -1: the ones that ran out of passes)
__global__ __launch_bounds__(256) void tv_rtol_select_kernel(const float4* __restrict__ tmp, float4* __restrict__ sol, const int* __restrict__ flag,
                                                             int pass, size_t img4) {
  const size_t c = blockIdx.x;
  if (flag[c] != pass) return;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img4; k += (size_t)gridDim.y * blockDim.x) sol[c * img4 + k] = tmp[c * img4 + k];
}
__global__ __launch_bounds__(256) void tv_rtol_select1_kernel(const float* __restrict__ tmp, float* __restrict__ sol, const int* __restrict__ flag,
                                                              int pass, size_t img) {
  const size_t c = blockIdx.x;
  if (flag[c] != pass) return;
  for (size_t k = (size_t)blockIdx.y * blockDim.x + threadIdx.x; k < img; k += (size_t)gridDim.y * blockDim.x) sol[c * img + k] = tmp[c * img + k];
}

hipError_t launch_tv_objective(const float* x, const float* sol, int64_t n, int H, int W, float gam, const int* flag, double* obj, hipStream_t st) {
  const size_t img = (size_t)H * W;
  int gy = (int)((img + 255) / 256);
  if (gy > 64) gy = 64;
  hipLaunchKernelGGL(tv_objective_kernel, dim3((unsigned)n, gy), dim3(256), 0, st, x, sol, H, W, gam, flag, obj);
  return hipGetLastError();
}
hipError_t launch_tv_rtol_decide(int64_t n, double* prev, double* cur, int* flag, int pass, double rtol, int* n_active, hipStream_t st) {
  hipLaunchKernelGGL(tv_rtol_decide_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, n, prev, cur, flag, pass, rtol, n_active);
  return hipGetLastError();
}
hipError_t launch_tv_rtol_select(const float* tmp, float* sol, const int* flag, int pass, int64_t n, size_t img, hipStream_t st) {
  int gy = (int)((img / 4 + 255) / 256);
  if (gy > 64) gy = 64;
  if (gy < 1) gy = 1;
  if ((img & 3) == 0)
    hipLaunchKernelGGL(tv_rtol_select_kernel, dim3((unsigned)n, gy), dim3(256), 0, st, reinterpret_cast<const float4*>(tmp), reinterpret_cast<float4*>(sol),
                       flag, pass, img / 4);
  else
    hipLaunchKernelGGL(tv_rtol_select1_kernel, dim3((unsigned)n, gy), dim3(256), 0, st, tmp, sol, flag, pass, img);
  return hipGetLastError();
}

// ---- 1-D TV over the flattened image: the inner prox of the ANISOTROPIC ME-TV term of algs.L2_ncvx_tv (algs.py:170: pyproximal.TV((prod(dims),), 1., niter,
// rtol)) ----  No model of the reference's driver uses it (prox_lmc_deconv.py:106-113 are the isotropic ones), so this is plain coverage: one pass over the
// images per dual iteration, dual and projected dual ping-ponged through memory, the same pass-by-pass early exit as tv_prox_rtol (lmc_capi.hip: tv1d_prox).
// One dual component r: sol = x - gam div(rr), div(r)[i] = r[i] - r[i-1] with the last entry of r taken as zero; r = rr - c (sol[i+1] - sol[i]) (0 at the
// end), p' = r / max(1, |r|), rr' = p' + beta (p' - p).  Images with flag[c] >= 0 (left already) are skipped.
__device__ __forceinline__ float tv1d_sol_at(const float* __restrict__ x, const float* __restrict__ rr, size_t i, size_t N, float gam) {
  const float ri = i + 1 < N ? rr[i] : 0.f, rm = i > 0 ? rr[i - 1] : 0.f;
  return x[i] - gam * (ri - rm);
}
__global__ __launch_bounds__(256) void tv1d_sol_kernel(const float* __restrict__ x, const float* __restrict__ rr, float* __restrict__ out, size_t N, float gam,
                                                       const int* __restrict__ flag) {
  const size_t c = blockIdx.y;
  if (flag && flag[c] >= 0) return;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x)
    out[c * N + i] = tv1d_sol_at(x + c * N, rr + c * N, i, N, gam);
}
__global__ __launch_bounds__(256) void tv1d_iter_kernel(const float* __restrict__ x, const float* __restrict__ rr_in, float* __restrict__ p, float* __restrict__ rr_out,
                                                        size_t N, float gam, float cstep, float beta, const int* __restrict__ flag) {
  const size_t c = blockIdx.y;
  if (flag && flag[c] >= 0) return;
  const float* xc = x + c * N;
  const float* rc = rr_in + c * N;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
    const float d = i + 1 < N ? tv1d_sol_at(xc, rc, i + 1, N, gam) - tv1d_sol_at(xc, rc, i, N, gam) : 0.f;
    const float r = rc[i] - cstep * d;
    const float pn = r / fmaxf(1.f, fabsf(r));
    rr_out[c * N + i] = pn + beta * (pn - p[c * N + i]);
    p[c * N + i] = pn;
  }
}
// obj[c] += 1/2 ||x_c - sol_c||^2 + gam TV_1D(sol_c)   (gam = 1 and x = sol: the 1-D TV value alone)
__global__ __launch_bounds__(256) void tv1d_objective_kernel(const float* __restrict__ x, const float* __restrict__ sol, size_t N, float gam, const int* __restrict__ flag,
                                                             double* __restrict__ obj) {
  __shared__ double scratch[4];
  const size_t c = blockIdx.y;
  if (flag && flag[c] >= 0) return;
  double acc = 0.0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < N; i += (size_t)gridDim.x * blockDim.x) {
    const float v = sol[c * N + i];
    const float d = x[c * N + i] - v;
    const float dn = i + 1 < N ? sol[c * N + i + 1] - v : 0.f;
    acc += 0.5 * (double)d * (double)d + (double)gam * (double)fabsf(dn);
  }
  const double t = block_sum(acc, scratch);
  if (threadIdx.x == 0) unsafeAtomicAdd(&obj[c], t);
}
static dim3 tv1d_grid(int64_t n, size_t N) {
  size_t gx = (N + 255) / 256;
  if (gx > 256) gx = 256;
  return dim3((unsigned)gx, (unsigned)n);
}
hipError_t launch_tv1d_sol(const float* x, const float* rr, float* out, int64_t n, size_t N, float gam, const int* flag, hipStream_t st) {
  if (n > 65535) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(tv1d_sol_kernel, tv1d_grid(n, N), dim3(256), 0, st, x, rr, out, N, gam, flag);
  return hipGetLastError();
}
hipError_t launch_tv1d_iter(const float* x, const float* rr_in, float* p, float* rr_out, int64_t n, size_t N, float gam, float cstep, float beta, const int* flag,
                            hipStream_t st) {
  if (n > 65535) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(tv1d_iter_kernel, tv1d_grid(n, N), dim3(256), 0, st, x, rr_in, p, rr_out, N, gam, cstep, beta, flag);
  return hipGetLastError();
}
hipError_t launch_tv1d_objective(const float* x, const float* sol, int64_t n, size_t N, float gam, const int* flag, double* obj, hipStream_t st) {
  if (n > 65535) return hipErrorInvalidConfiguration;
  hipLaunchKernelGGL(tv1d_objective_kernel, tv1d_grid(n, N), dim3(256), 0, st, x, sol, N, gam, flag, obj);
  return hipGetLastError();
}

// ---- the same early exit without leaving the device (lmc_capi.hip: tv_prox_rt): speculate, verify, re-run -------------------------------------
// A chain's prox runs with a PREDICTED number of dual updates k (the pass it left in at the previous call: the objective is a sum over the
// whole image and moves little from one MYULA iterate to the next), fused in the RT instantiations of the pipe kernel, which leave the primal
// objective of every iterate they form in obj[c][0 .. k].  tv_rt_decide replays upstream's test on them, one thread per chain:
//   the first pass j in 1 .. min(k, niter - 1) with |obj_j - obj_{j-1}| / obj_j < rtol (obj_j > 0) is the pass e the chain leaves in;
//   e == k, or no such pass and k == niter (the iterate after niter updates is returned untested): the run was the reference's -- settled;
//   e <  k: run again with e updates (exact, settled after it);
//   no such pass and k < niter: the chain leaves later -- run again with k + 1 updates (pass counts wander by one between MYULA iterates), and if that
//   is still not it, with all niter (whose objectives show e exactly), then with e.
// Four rounds settle every chain.  kc[c] = -1 marks a settled chain, whose workgroups return at once.
// A chained prox (more than 10 updates: links of 10 handing the dual state over in two ping-pong buffers) does not start again from its first
// link: the links below `start[c]` keep their result of the earlier round.  What survives a run whose last link was j_k (it wrote no state): the
// states after links j_k - 1 and j_k - 2 -- so a re-run can begin at link j_k (always: its input state is intact) or at j_k - 1 (when the run
// itself began at or before it), else at link 0.  The objectives below the first re-run link stay as they are, the others are cleared.
// (start[c] also carries, in bit 30, "the count of this run is known to be the exit pass": such a run settles the chain without a second look --
// the objective of the iterate a run returns is summed by the combine wave, that of the same iterate inside a longer run by a TV stage, and the two
// sums differ in their last bits: a chain sitting on the threshold could otherwise be sent back and forth.)
constexpr int kRtExact = 1 << 30;
__global__ void tv_rt_begin_kernel(int64_t n, const int* __restrict__ pred, int* __restrict__ kc, int* __restrict__ start, double* __restrict__ obj,
                                   int stride, int niter) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int k = pred[c];
  kc[c] = (k >= 1 && k <= niter) ? k : niter;
  start[c] = 0;
  for (int j = 0; j <= niter; ++j) obj[c * stride + j] = 0.0;
}
__global__ void tv_rt_decide_kernel(int64_t n, int* __restrict__ kc, int* __restrict__ start, int* __restrict__ pred, double* __restrict__ obj, int stride,
                                    int niter, double rtol, int round, unsigned long long* __restrict__ reruns) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= n) return;
  const int k = kc[c];
  if (k < 0) return;
  if (start[c] & kRtExact) { pred[c] = k; kc[c] = -1; return; }
  double* o = obj + c * stride;
  int e = 0;
  const int jmax = k < niter - 1 ? k : niter - 1;
  for (int j = 1; j <= jmax; ++j) {
    const double rel = o[j] > 0.0 ? fabs(o[j] - o[j - 1]) / o[j] : 2.0 * rtol;
    if (rel < rtol) { e = j; break; }
  }
  if (e == k || (e == 0 && k == niter)) { pred[c] = k; kc[c] = -1; return; }
  const int s = start[c], jk = (k - 1) / 10;
  int nk, ns;
  if (e > 0) {                    // left earlier than predicted: exact
    nk = e;
    const int je = (e - 1) / 10;
    ns = je == jk ? jk : ((je == jk - 1 && s <= jk - 1) ? jk - 1 : 0);
  } else {                        // leaves later
    nk = round == 0 ? (k + 1 < niter ? k + 1 : niter) : niter;
    ns = jk;
  }
  kc[c] = nk;
  start[c] = ns | (e > 0 ? kRtExact : 0);
  for (int j = 10 * ns; j <= niter; ++j) o[j] = 0.0;
  if (reruns) atomicAdd(reruns, 1ull);
}
hipError_t launch_tv_rt_begin(int64_t n, const int* pred, int* kc, int* start, double* obj, int stride, int niter, hipStream_t st) {
  hipLaunchKernelGGL(tv_rt_begin_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, n, pred, kc, start, obj, stride, niter);
  return hipGetLastError();
}
hipError_t launch_tv_rt_decide(int64_t n, int* kc, int* start, int* pred, double* obj, int stride, int niter, double rtol, int round,
                               unsigned long long* reruns, hipStream_t st) {
  hipLaunchKernelGGL(tv_rt_decide_kernel, dim3((unsigned)((n + 127) / 128)), dim3(128), 0, st, n, kc, start, pred, obj, stride, niter, rtol, round, reruns);
  return hipGetLastError();
}

// *p += by (one thread): the device-side iteration base of a replayed hipGraph of MYULA iterations
__global__ void bump_u32_kernel(uint32_t* p, uint32_t by) { *p += by; }
hipError_t launch_bump_u32(uint32_t* p, uint32_t by, hipStream_t st) {
  hipLaunchKernelGGL(bump_u32_kernel, dim3(1), dim3(1), 0, st, p, by);
  return hipGetLastError();
}

}  // namespace lmc
