// Fused MYULA update for a separable blur data term and a prior without a stencil (l2, l1, none, or a prox computed by a
// preceding launch):   out = a*x - t*sigma_f H^T(Hx - y) + b*prox(x) + s*xi        (algs.py:569, grad algs.py:283-284)
//
// Barrier-free row streaming.  ONE wave owns the full width of a band of rows of one chain: lane l holds PXL = 4 or 8
// consecutive pixels of the row (W <= 64*PXL), so the zero boundary of the "same" convolution is the wave's own edge and no
// wave ever talks to another one -- no LDS, no __syncthreads.  The wave walks down its band one input row per step:
//   x row i --h-blur--> scattered into KT residual accumulators (rows i-HW..i+HW)
//   residual row i-HW complete: R = Hx - y (zero outside the image) --h-adjoint--> scattered into KT gradient accumulators
//   gradient row o = i-(KT-1) complete: combine with x[o] (still in the register ring), prox, Philox noise, store.
// Horizontal neighbours: 2*HW wave-shift DPP moves per pass for PXL pixels.  All ring slots are (row & 7) with the row loop
// unrolled by 8, so every index is a compile-time constant and the rings live in VGPRs without rotation moves.
// Bands start KT-1 rows early (recompute instead of exchange); HBM traffic = x read once (+ band overlap) + x' written once.
#include "lmc_step_rows_kernel.h"

#include <cmath>
#include <cstdlib>

namespace lmc {

hipError_t launch_step_rows_uni(const StepArgs& a, int KT, int lo, int hi, bool al, int nblk, int band, int nbands, hipStream_t st, bool* handled);

// Rank-1 factorisation h = u v^T of the blur taps (u: kh, v: kw).  Returns false if h is not separable.
bool separate_blur_taps(const BlurTaps& T, float* u, float* v) {
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

// Centred taps: the window of a kh-tap kernel with offset oy is rows r+oy-kh+1 .. r+oy; with KT = 2*HW+1 taps centred
// on r that is u'[a + HW - oy] = u[a], which needs 0 <= HW - oy and kh + HW - oy <= KT.
static bool rows_centre(const float* u, int n, int off, int KT, float* out) {
  const int HW = (KT - 1) / 2, sh = HW - off;
  if (sh < 0 || n + sh > KT) return false;
  for (int i = 0; i < KT; ++i) out[i] = 0.f;
  for (int i = 0; i < n; ++i) out[i + sh] = u[i];
  return true;
}

int centred_blur_taps(const BlurTaps& T, float* uc, float* vc) {
  float u[kMaxBlur] = {0}, v[kMaxBlur] = {0};
  if (T.kh > 7 || T.kw > 7 || !separate_blur_taps(T, u, v)) return 0;
  for (int KT = 5; KT <= 7; KT += 2)
    if (rows_centre(u, T.kh, T.oy, KT, uc) && rows_centre(v, T.kw, T.ox, KT, vc)) return KT;
  return 0;
}
int centred_blur_taps(const StepArgs& a, float* uc, float* vc) { return centred_blur_taps(a.blur, uc, vc); }

bool rows_supported(const StepArgs& a) {
  if (a.data_kind != LMC_DATA_BLUR || a.ncvx_kind != LMC_NCVX_NONE) return false;
  if (a.prior_kind != LMC_PRIOR_NONE && a.prior_kind != LMC_PRIOR_L2 && a.prior_kind != LMC_PRIOR_L1 && a.prior_kind != LMC_PRIOR_EPROX) return false;
  if (a.prior_kind == LMC_PRIOR_EPROX && ((a.W & 3) || a.W > 512 || a.dot_out)) return false;      // EP instantiations: aligned rows, one strip
  if (a.tv_in || a.tv_out) return false;
  // any width: rows that are not 16-byte aligned (W % 4 != 0) go pixel by pixel, images wider than one wave (512 columns) as column strips
  if (a.W < 4 || a.W > 16384 || a.H < 1) return false;
  float uc[kMaxBlur], vc[kMaxBlur];
  const int KT = centred_blur_taps(a, uc, vc);
  if (KT == 0) return false;
  // 7 taps x 8 pixels per lane does not fit 256 VGPRs: that instantiation runs one wave per SIMD with the overflow in AGPRs --
  // slower per pixel than the 5-tap one, still several times faster than the general kernels (6x6 / 7x7 blurs of the reference at 512 wide)
  return true;
}

hipError_t launch_step_rows(StepArgs a, hipStream_t st) {
  if (!rows_supported(a)) return hipErrorInvalidConfiguration;
  float uc[kMaxBlur] = {0}, vc[kMaxBlur] = {0};
  const int KT = centred_blur_taps(a, uc, vc);
  for (int i = 0; i < kMaxBlur; ++i) { a.blur.h[i] = i < KT ? uc[i] : 0.f; a.blur.h[kMaxBlur + i] = i < KT ? vc[i] : 0.f; }
  // bands: enough waves to fill 1024 SIMDs a few times over, but >= 16 rows each (a band recomputes KT-1 rows at each end)
  static const int env_band = [] { const char* e = getenv("LMC_ROWS_BAND"); return e ? atoi(e) : 0; }();
  int band = env_band > 0 ? env_band : 0;
  if (band == 0) {
    const int want = (4096 + a.C - 1) / a.C;                         // bands per chain for ~4 waves per SIMD
    band = (a.H + want - 1) / want;
    if (band < 16) band = 16;
  }
  band = (band + 7) & ~7;
  const int nbands = (a.H + band - 1) / band;
  const bool al = (a.W & 3) == 0;
  const int pxl = (a.W <= 256 && al) ? 4 : 8;                        // the pixel-by-pixel instantiations are 8 per lane only
  const int ustrip = 64 * pxl - 16;
  const int nstrips = a.W <= 64 * pxl ? 1 : (a.W + ustrip - 1) / ustrip;     // as in the kernel
  const long long waves = (long long)a.C * nbands * nstrips;
  if (waves > (1ll << 31) - 8) return hipErrorInvalidConfiguration;
  const int nblk = (int)((waves + 3) / 4);
  if (a.dot_out) {        // CG operator apply with the p.Ap reduction fused (lmc_capi.hip: cg_solve_fused)
    if (!al) {
      if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<8, 5, true, -1, -1, false>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
      else hipLaunchKernelGGL((myula_step_rows_kernel<8, 7, true, -1, -1, false>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    } else if (a.W <= 256) {
      if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<4, 5, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
      else hipLaunchKernelGGL((myula_step_rows_kernel<4, 7, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    } else {
      if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<8, 5, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
      else hipLaunchKernelGGL((myula_step_rows_kernel<8, 7, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    }
    return hipGetLastError();
  }
  // uniform boxes (all the reference's blurs): the sliding-window form.  Taps constant on one window [lo, hi] -- the same for rows and columns --
  // and zero elsewhere; LMC_ROWS_UNI=0 keeps the general form (A/B runs).
  static const bool uni_on = [] { const char* e = getenv("LMC_ROWS_UNI"); return !e || atoi(e) != 0; }();
  int lo = -1, hi = -1;
  if (uni_on) {
    auto window = [&](const float* t, int& l, int& h) {
      l = -1; h = -1;
      for (int i = 0; i < KT; ++i) if (t[i] != 0.f) { if (l < 0) l = i; h = i; }
      if (l < 0) return false;
      for (int i = l; i <= h; ++i) if (std::fabs(t[i] - t[l]) > 1e-6f * std::fabs(t[l])) return false;
      return true;
    };
    int l2, h2;
    if (!(window(uc, lo, hi) && window(vc, l2, h2) && l2 == lo && h2 == hi)) lo = hi = -1;
  }
  {   // uniform boxes and the closed-form elementwise priors: instantiated in lmc_step_rows_uni.hip
    bool handled = false;
    const hipError_t e = launch_step_rows_uni(a, KT, lo, hi, al, nblk, band, nbands, st, &handled);
    if (handled) return e;
  }
  if (!al) {
    if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<8, 5, false, -1, -1, false>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    else hipLaunchKernelGGL((myula_step_rows_kernel<8, 7, false, -1, -1, false>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
  } else if (a.W <= 256) {
    if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<4, 5>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    else hipLaunchKernelGGL((myula_step_rows_kernel<4, 7>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
  } else {
    if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<8, 5>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    else hipLaunchKernelGGL((myula_step_rows_kernel<8, 7>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
  }
  return hipGetLastError();
}

}  // namespace lmc
