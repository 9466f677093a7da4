// Instantiations of the pipe step kernel (lmc_step_pipe_kernel.h) that read / write the TV dual state in HBM: the links of a chained
// launch (more than 10 dual iterations: 10 per link, the last one 9 or 10) and the warm-started prox (the projected dual carried
// from one MYULA iteration to the next, 1 / 2 / 3 dual iterations per MYULA iteration).  A translation unit of its own so that
// the two sets of instantiations compile in parallel.
#include "lmc_step_pipe_kernel.h"

namespace lmc {

hipError_t pipe_dispatch_chain(const StepArgs& a, int K, int KT, hipStream_t st) {
  if (a.tv_warm) {
    switch (K) {
      case 1: return pipe_dispatch_k<1, true, true>(a, KT, st);
      case 2: return pipe_dispatch_k<2, true, true>(a, KT, st);
      case 3: return pipe_dispatch_k<3, true, true>(a, KT, st);
      default: return hipErrorInvalidConfiguration;
    }
  }
  switch (K) {
    case 9: return pipe_dispatch_k<9, true>(a, KT, st);
    case 10: return pipe_dispatch_k<10, true>(a, KT, st);
    default: return hipErrorInvalidConfiguration;
  }
}

}  // namespace lmc
