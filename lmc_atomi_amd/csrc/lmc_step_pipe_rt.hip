// RT instantiations of the pipe step kernel (lmc_step_pipe_kernel.h): the TV prox with upstream's per-image early exit (pyproximal.TV's rtol, in force
// at prox_lmc_deconv.py:122 and algs.py:169) as a per-chain number of live pipeline stages, with the primal objectives of the iterates as by-products.
// The K = 10 pipeline serves every iteration cap up to 10 in one launch (stages past a chain's count pass the dual through) and up to 60 as a chain of
// links that hand the dual state over in HBM (the inner prox of the ME-TV term, niter_l2 = 50): a chain leaves in the link that holds its last update,
// the links before it only advance its dual state, the links after it return at once.  The host side that predicts, verifies and re-runs is
// lmc_capi.hip: tv_prox_rt.  A translation unit of its own so that the instantiations compile in parallel with the others.
#include "lmc_step_pipe_kernel.h"

namespace lmc {

bool pipe_rt_supported(const StepArgs& a) {
  if (a.prior_kind != LMC_PRIOR_TV_ISO) return false;
  const int n = a.tv.niter;
  if (n < 1 || n > 60 || n > kMaxTvIters) return false;
  if (a.tv_in || a.tv_out || a.tv_state_only || a.tv_warm) return false;
  if (!pipe_geometry_ok(a)) return false;
  // one strip (objective sums would count the recomputed halos of column strips twice), the last image column a lane's last pixel
  if (a.W > 512 || (a.W & (a.W > 256 ? 7 : 3))) return false;
  if (a.f_out || a.g_out) return false;
  // chained links: the pure prox only (every link may be the one some chain leaves in, so none could skip a data term)
  if (n > 10 && (a.data_kind != LMC_DATA_NONE || a.noise_mode != LMC_NOISE_NONE || a.ncvx_kind != LMC_NCVX_NONE || a.extra)) return false;
  return true;
}

template <bool CHAIN>
static hipError_t pipe_dispatch_rt(const StepArgs& a, int KT, hipStream_t st) {
  if (a.W > 256) {
    if constexpr (!CHAIN) {
      if (KT == 5) return pipe_launch_one<8, 5, false, 10, false, true, true>(a, st);
      if (KT == 7) return pipe_launch_one<8, 7, false, 10, false, true, true>(a, st);
    }
    return pipe_launch_one<8, 0, CHAIN, 10, false, true, true>(a, st);
  }
  if constexpr (!CHAIN) {
    if (KT == 5) return pipe_launch_one<4, 5, false, 10, false, true, true>(a, st);
    if (KT == 7) return pipe_launch_one<4, 7, false, 10, false, true, true>(a, st);
  }
  return pipe_launch_one<4, 0, CHAIN, 10, false, true, true>(a, st);
}

hipError_t launch_step_pipe_rt(StepArgs a, hipStream_t st, float* state0, float* state1) {
  if (!pipe_rt_supported(a) || !a.rt_kc || !a.rt_obj || a.rt_stride < a.tv.niter + 1) return hipErrorInvalidConfiguration;
  const int n = a.tv.niter, links = (n + 9) / 10;
  if (links > 1 && (!state0 || !state1)) return hipErrorInvalidConfiguration;
  const int KT = pipe_taps(a);
  a.rt_total = n;
  float* st_buf[2] = {state0, state1};
  for (int j = 0; j < links; ++j) {
    StepArgs b = a;
    b.tv.niter = 10;
    for (int i = 0; i < 10; ++i) b.tv.betas[i] = 10 * j + i < n ? a.tv.betas[10 * j + i] : 0.f;
    b.rt_base = 10 * j;
    if (links > 1) {
      b.tv_in = j > 0 ? st_buf[(j - 1) & 1] : nullptr;
      b.tv_out = j < links - 1 ? st_buf[j & 1] : nullptr;
    }
    const hipError_t e = links > 1 ? pipe_dispatch_rt<true>(b, 0, st) : pipe_dispatch_rt<false>(b, KT, st);
    if (e != hipSuccess) return e;
  }
  return hipSuccess;
}

}  // namespace lmc
