// Row-streaming step kernel (lmc_step_rows_kernel.h): the uniform-box instantiations (the reference's blurs) and those with a closed-form elementwise prior.
#include "lmc_step_rows_kernel.h"

namespace lmc {

hipError_t launch_step_rows_uni(const StepArgs& a, int KT, int lo, int hi, bool al, int nblk, int band, int nbands, hipStream_t st, bool* handled) {
  *handled = true;
#define LMC_ROWS_UNI_LAUNCH(PX, KTT, LO, HI, ...)                                                                              \
  if (KT == KTT && lo == LO && hi == HI) {                                                                                   \
    hipLaunchKernelGGL((myula_step_rows_kernel<PX, KTT, false, LO, HI, ##__VA_ARGS__>), dim3(nblk), dim3(256), 0, st, a, band, nbands); \
    return hipGetLastError();                                                                                                \
  }
  if (a.prior_kind == LMC_PRIOR_EPROX) {     // closed-form elementwise priors: the general-taps form and the 5 x 5 uniform box, aligned rows
    if (a.W <= 256) {
      if (KT == 5 && lo == 0 && hi == 4) hipLaunchKernelGGL((myula_step_rows_kernel<4, 5, false, 0, 4, true, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
      else if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<4, 5, false, -1, -1, true, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
      else hipLaunchKernelGGL((myula_step_rows_kernel<4, 7, false, -1, -1, true, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    } else {
      if (KT == 5 && lo == 0 && hi == 4) hipLaunchKernelGGL((myula_step_rows_kernel<8, 5, false, 0, 4, true, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
      else if (KT == 5) hipLaunchKernelGGL((myula_step_rows_kernel<8, 5, false, -1, -1, true, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
      else hipLaunchKernelGGL((myula_step_rows_kernel<8, 7, false, -1, -1, true, true>), dim3(nblk), dim3(256), 0, st, a, band, nbands);
    }
    return hipGetLastError();
  }
  if (!al) {
    LMC_ROWS_UNI_LAUNCH(8, 5, 0, 4, false) LMC_ROWS_UNI_LAUNCH(8, 7, 0, 6, false) LMC_ROWS_UNI_LAUNCH(8, 7, 0, 5, false)
  } else if (a.W <= 256) {
    LMC_ROWS_UNI_LAUNCH(4, 5, 0, 4) LMC_ROWS_UNI_LAUNCH(4, 7, 0, 6) LMC_ROWS_UNI_LAUNCH(4, 7, 0, 5)
  } else {
    LMC_ROWS_UNI_LAUNCH(8, 5, 0, 4) LMC_ROWS_UNI_LAUNCH(8, 7, 0, 6) LMC_ROWS_UNI_LAUNCH(8, 7, 0, 5)
  }
#undef LMC_ROWS_UNI_LAUNCH
  *handled = false;
  return hipSuccess;
}

}  // namespace lmc
