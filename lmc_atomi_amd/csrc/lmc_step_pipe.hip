// Host side of the "pipe" step kernel (lmc_step_pipe_kernel.h): coverage tests, dispatch, chained launches.  This translation unit
// holds the instantiations WITHOUT dual-state hand-over (one launch = the whole TV prox from the zero dual state); the ones that
// read / write the dual state in HBM (chained launches, warm-started prox) are in lmc_step_pipe_chain.hip.
#include "lmc_step_pipe_kernel.h"

namespace lmc {

int centred_blur_taps(const StepArgs& a, float* uc, float* vc);   // lmc_step_rows.hip

// image geometry and data term the kernel covers (whatever the number of dual iterations)
bool pipe_geometry_ok(const StepArgs& a) {
  if (a.prior_kind != LMC_PRIOR_TV_ISO || a.prox_ext) return false;
  // 8 (4) pixels per lane above (up to) 256 columns; narrower than 129 columns half the lanes idle and the split kernel wins.  Any width:
  // rows that are not 16-byte aligned (W % 4 != 0) use dword-aligned 16-byte accesses and per-pixel masks, images wider than 512 columns run as column
  // strips of 512 with recomputed halos (the reference's 667 x 877 image: two strips) -- without the energy by-products, whose sums would count the halos.
  if (a.W <= 128 || a.H < 1 || a.W > 16384) return false;
  if (a.W > 512 && (a.f_out || a.g_out)) return false;
  // widths that are not a multiple of the pixels per lane (8 above 256 columns, else 4): K = 10 (and its chains) only
  if ((a.W & (a.W > 256 ? 7 : 3)) && !(a.tv.niter == 10 || (a.tv.niter >= 20 && a.tv.niter % 10 == 0))) return false;
  if (a.data_kind == LMC_DATA_BLUR) {
    float uc[kMaxBlur], vc[kMaxBlur];
    if (centred_blur_taps(a, uc, vc) == 0) return false;
  } else if (a.data_kind != LMC_DATA_NONE && a.data_kind != LMC_DATA_IDENTITY && a.data_kind != LMC_DATA_MASK) {
    return false;
  }
  // without a blur (no data term, or a pointwise one formed in the load wave): no extra gradient terms, no energy by-products
  if (a.data_kind != LMC_DATA_BLUR && (a.ncvx_kind != LMC_NCVX_NONE || a.f_out)) return false;
  return true;
}

// Number of launches a configuration needs (0: not covered).  One launch: 2, 6, 8, 9 or 10 dual iterations (9 = the "lagged" reading of
// niter = 10, lmc_problem.tv_lagged_output).  More (19, 20, 29, 30, ... 60): a chain of launches of 10 (the last one 9 or 10) that hand the
// dual state (rr, ss, p, q) over in HBM -- exact, the same mechanism as the tile kernel's chunks.
// (measured at 512x512x1024: K = 2 / 6 / 8 / 10: 0.99 / 1.60 / 1.79 / 1.85 ms vs 1.25 / 1.78 / 2.10 / 2.36 split; 4: 1.53 vs 1.42, left to split)
int pipe_links(const StepArgs& a) {
  if (a.prior_kind != LMC_PRIOR_TV_ISO || a.tv.niter > kMaxTvIters) return 0;
  const int n = a.tv.niter;
  const bool single = n == 2 || n == 6 || n == 8 || n == 9 || n == 10;
  if (!single && (n < 19 || (n % 10 != 0 && n % 10 != 9))) return 0;
  if (a.tv_in || a.tv_out || a.tv_state_only || a.tv_warm) return 0;
  if (!pipe_geometry_ok(a)) return 0;
  return single ? 1 : (n + 9) / 10;
}

bool pipe_supported(const StepArgs& a) { return pipe_links(a) == 1; }

// warm-started prox: the projected dual (p, q) of the previous MYULA iteration comes in through a.tv_in and the new one leaves through
// a.tv_out ([C][2][H][W] each, never NULL), a.tv.niter in {1, 2, 3} dual iterations per MYULA iteration
bool pipe_warm_supported(const StepArgs& a) {
  const int n = a.tv.niter;
  if (!(n == 1 || n == 2 || n == 3)) return false;
  return pipe_geometry_ok(a);
}

int pipe_taps(StepArgs& a) {
  int KT = 0;
  if (a.data_kind == LMC_DATA_BLUR) {
    float uc[kMaxBlur] = {0}, vc[kMaxBlur] = {0};
    KT = centred_blur_taps(a, uc, vc);
    for (int i = 0; i < kMaxBlur; ++i) { a.blur.h[i] = i < KT ? uc[i] : 0.f; a.blur.h[kMaxBlur + i] = i < KT ? vc[i] : 0.f; }
  }
  return KT;
}

hipError_t launch_step_pipe_warm(StepArgs a, hipStream_t st) {
  if (!pipe_warm_supported(a) || !a.tv_in || !a.tv_out || a.tv_in == a.tv_out) return hipErrorInvalidConfiguration;
  a.tv_warm = 1;
  a.tv_state_only = 0;
  const int KT = pipe_taps(a);
  return pipe_dispatch_chain(a, a.tv.niter, KT, st);
}

// state0 / state1: [C][4][H][W] ping-pong buffers for the dual state between links (needed when a.tv.niter > 10)
hipError_t launch_step_pipe(StepArgs a, hipStream_t st, float* state0, float* state1) {
  const int links = pipe_links(a);
  if (links == 0 || (links > 1 && (!state0 || !state1))) return hipErrorInvalidConfiguration;
  const int KT = pipe_taps(a);
  if (links == 1) {
    switch (a.tv.niter) {
      case 2: return pipe_dispatch_k<2, false>(a, KT, st);
      case 6: return pipe_dispatch_k<6, false>(a, KT, st);
      case 8: return pipe_dispatch_k<8, false>(a, KT, st);
      case 9: return pipe_dispatch_k<9, false>(a, KT, st);
      default: return pipe_dispatch_k<10, false>(a, KT, st);
    }
  }
  float* st_buf[2] = {state0, state1};
  for (int j = 0; j < links; ++j) {
    StepArgs b = a;
    const int kl = (j == links - 1) ? a.tv.niter - 10 * j : 10;      // 10, or 9 in the last link
    b.tv.niter = kl;
    for (int i = 0; i < kl; ++i) b.tv.betas[i] = a.tv.betas[10 * j + i];
    b.tv_in = j > 0 ? st_buf[(j - 1) & 1] : nullptr;
    b.tv_out = j < links - 1 ? st_buf[j & 1] : nullptr;
    b.tv_state_only = j < links - 1;
    // the data term, the noise and the energies belong to the last link only; the earlier ones skip the blur pipeline
    hipError_t e = pipe_dispatch_chain(b, kl, b.tv_state_only ? 0 : KT, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace lmc
