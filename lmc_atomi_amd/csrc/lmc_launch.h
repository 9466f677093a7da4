// Host-side launcher declarations (defined in lmc_step_*.hip / lmc_ops.hip).
#pragma once
#include "lmc_common.h"

namespace lmc {

struct EnergyArgs {
  int H, W;
  int data_kind;
  float sigma_f;
  const float* y;
  const float* mask;
  BlurTaps blur;
  int prior_kind;
  float prior_sigma;
  int ncvx_kind;          // f -= ncvx_lambda * sum huber_gamma(|grad x|)  (algs.py:173-190, MC-TV isotropic)
  float ncvx_lambda, ncvx_gamma;
};

hipError_t launch_step_tile(StepArgs a, hipStream_t st);
bool tile_needs_chunks(const StepArgs& a);
hipError_t launch_step_tile_chunked(const StepArgs& a, float* state0, float* state1, hipStream_t st);
hipError_t launch_axpy_env(double* f, const double* tvv, const double* sq, int64_t n, float lambda, float gamma, hipStream_t st);
hipError_t launch_sqdiff(const float* a, const float* b, int64_t n_img, size_t img, double* out, hipStream_t st);
hipError_t launch_blur(const float* x, float* out, int64_t n_img, int H, int W, const BlurTaps& T, int adjoint,
                       hipStream_t st);
hipError_t launch_gradient(const float* x, float* out, int64_t n_img, int H, int W, bool adjoint, hipStream_t st);
hipError_t launch_dual_project(const float* y, float* out, int64_t n_img, int H, int W, float radius, int iso,
                               hipStream_t st);
hipError_t launch_eprox(int kind, const float* x, float* out, int64_t n, float p0, float p1, hipStream_t st);
hipError_t launch_prior_prox_scaled(int prior, int kind, const float* x, float* out, int64_t n_chains, int64_t img, const float* scale, int64_t cs, int64_t ps,
                                    float pt, float sigma, float p0, float p1, int mask, hipStream_t st);
hipError_t launch_haar_prox(const float* x, float* out, int64_t n_img, int H, int W, float thr, hipStream_t st);
hipError_t launch_chain_probes(const float* x, float* out, int64_t n_img, int H, int W, int ph, int pw, hipStream_t st);
hipError_t launch_haar_value(const float* x, int64_t n_img, int H, int W, float sigma, double* val, hipStream_t st);
hipError_t launch_mc_tv_add(const float* x, float* out, int64_t n_img, int H, int W, float coef, float gamma, hipStream_t st);
hipError_t launch_moments(const float* x, int C, int H, int W, double* s1, double* s2, hipStream_t st);
hipError_t launch_moments_bg(const float* x, int C, int H, int W, double* s1, double* s2, int n_wg, hipStream_t st);
hipError_t launch_energies(const float* x, int64_t n_img, const EnergyArgs& E, double* f_out, double* g_out,
                           hipStream_t st);
hipError_t launch_bump_u32(uint32_t* p, uint32_t by, hipStream_t st);
// pieces of the exact early-exit path of the TV prox (lmc_problem.tv_rtol > 0)
hipError_t launch_tv_objective(const float* x, const float* sol, int64_t n, int H, int W, float gam, const int* flag, double* obj, hipStream_t st);
hipError_t launch_tv_rtol_decide(int64_t n, double* prev, double* cur, int* flag, int pass, double rtol, int* n_active, hipStream_t st);
hipError_t launch_tv_rtol_select(const float* tmp, float* sol, const int* flag, int pass, int64_t n, size_t img, hipStream_t st);
// 1-D TV over the flattened image (inner prox of the anisotropic ME-TV term, algs.py:170): one dual iteration / the primal iterate / its objective
hipError_t launch_tv1d_sol(const float* x, const float* rr, float* out, int64_t n, size_t N, float gam, const int* flag, hipStream_t st);
hipError_t launch_tv1d_iter(const float* x, const float* rr_in, float* p, float* rr_out, int64_t n, size_t N, float gam, float cstep, float beta, const int* flag,
                            hipStream_t st);
hipError_t launch_tv1d_objective(const float* x, const float* sol, int64_t n, size_t N, float gam, const int* flag, double* obj, hipStream_t st);
// the early exit without leaving the device (speculate / verify / re-run; lmc_ops.hip, lmc_capi.hip: tv_prox_rt)
hipError_t launch_tv_rt_begin(int64_t n, const int* pred, int* kc, int* start, double* obj, int stride, int niter, hipStream_t st);
hipError_t launch_tv_rt_decide(int64_t n, int* kc, int* start, int* pred, double* obj, int stride, int niter, double rtol, int round,
                               unsigned long long* reruns, hipStream_t st);
int hbm_copy_probe_shapes();
hipError_t launch_hbm_copy_probe(const float* x, float* y, size_t n_floats, int shape, hipStream_t st);
hipError_t launch_noise(float* out, int C, int H, int W, uint32_t key0, uint32_t key1, uint32_t iteration,
                        uint32_t chain_offset, hipStream_t st);

}  // namespace lmc

namespace lmc {
// HBM-bound tiled kernel for closed-form priors (lmc_step_point.hip)
bool point_supported(const StepArgs& a);
hipError_t launch_step_point(StepArgs a, hipStream_t st);
// register-block kernel for stencil-free data terms and block-local proxes, Haar-l1 included (lmc_step_block.hip)
bool block_supported(const StepArgs& a);
bool block_pair_supported(const StepArgs& a);      // StepArgs::fused_iters = 2
hipError_t launch_step_block(const StepArgs& a, hipStream_t st);
// barrier-free row streaming for a separable blur + closed-form prior (lmc_step_rows.hip)
bool rows_supported(const StepArgs& a);
hipError_t launch_step_rows(StepArgs a, hipStream_t st);
// two MYULA iterations per launch (lmc_step_rows_pair.hip): x_out <- x_{k+2}, x_mid (may be NULL) <- x_{k+1}; three distinct arrays
bool rows_pair_supported(const StepArgs& a);
hipError_t launch_step_rows_pair(StepArgs a, float* x_mid, hipStream_t st);
// taps re-centred for the full-width kernels: returns KT (5 or 7) and fills uc / vc, or 0 (lmc_step_rows.hip)
int centred_blur_taps(const StepArgs& a, float* uc, float* vc);
int centred_blur_taps(const BlurTaps& T, float* uc, float* vc);
// TV stages spread over the waves of a workgroup, one wave = full image width (lmc_step_pipe.hip)
bool pipe_supported(const StepArgs& a);                  // one launch covers it (10 dual iterations)
int pipe_links(const StepArgs& a);                       // launches needed (20 .. 60 iterations: chained through HBM state), 0 = not covered
hipError_t launch_step_pipe(StepArgs a, hipStream_t st, float* state0 = nullptr, float* state1 = nullptr);
// warm-started TV prox: a.tv_in / a.tv_out = [C][2][H][W] projected dual of the previous / this MYULA iteration
bool pipe_warm_supported(const StepArgs& a);
hipError_t launch_step_pipe_warm(StepArgs a, hipStream_t st);
// per-chain early exit of the TV prox (lmc_step_pipe_rt.hip): a.tv.niter in 1 .. 60 dual updates at most, a.rt_kc / rt_obj / rt_stride set by the
// caller; more than 10: a chain of links through state0 / state1 ([C][4][H][W]), pure prox only (no data term, no noise)
bool pipe_rt_supported(const StepArgs& a);
hipError_t launch_step_pipe_rt(StepArgs a, hipStream_t st, float* state0 = nullptr, float* state1 = nullptr);
// split streaming variant: the same pipeline over two wave groups (lmc_step_split.hip)
bool split_supported(const StepArgs& a);
hipError_t launch_step_split(StepArgs a, hipStream_t st);
}  // namespace lmc

namespace lmc {
// ULPDA building blocks (lmc_ulpda.hip)
hipError_t ulpda_dual_update(const float* xhat, float* y, int64_t C, int H, int W, float mu, float radius, int iso, hipStream_t st);
hipError_t ulpda_rhs(const float* x, const float* y, const float* z, const float* htb, float* rhs, int64_t C, int H, int W,
                     float tau, float ts, hipStream_t st);
hipError_t ulpda_me_rhs(const float* v, const float* extra, const float* htb, float* rhs, int64_t C, int H, int W, float coef,
                        float ts, hipStream_t st);
hipError_t ulpda_ncvx_rhs(const float* v, const float* htb, float* rhs, int64_t C, int H, int W, float coef, float gamma, float ts,
                          hipStream_t st);
hipError_t ulpda_pointwise_prox(const float* v, float* u, const float* b, const float* m, int64_t C, int H, int W, float ts,
                                int kind, hipStream_t st);
hipError_t ulpda_finish_philox(float* x, float* xhat, const float* u, int64_t C, int H, int W, float s, float theta, uint32_t key0,
                               uint32_t key1, uint32_t iteration, uint32_t chain_offset, hipStream_t st);
hipError_t ulpda_finish(float* x, float* xhat, const float* u, const float* xi, int64_t C, int H, int W, float s, float theta,
                        hipStream_t st);
// finish + dual update in one row-streaming pass (gfirst = false; xhat stays in registers)
bool ulpda_finish_dual_supported(int H, int W);
hipError_t ulpda_finish_dual(const float* x, float* xnew, const float* u, float* y, const float* xi, int64_t C, int H, int W, float s, float theta, float mu,
                             float radius, int iso, int philox, uint32_t key0, uint32_t key1, uint32_t iteration, uint32_t chain_offset,
                             hipStream_t st);
hipError_t cg_dot(const float* p, const float* q, int64_t C, size_t img, double* pq, const int* done, hipStream_t st);
hipError_t cg_init(const float* rhs, const float* q, float* r, float* p, int64_t C, size_t img, double* rs, double* b2, hipStream_t st);
hipError_t cg_update(float* u, float* r, const float* p, const float* q, int64_t C, size_t img, const double* rs, const double* pq,
                     double* rs_new, const int* done, hipStream_t st);
hipError_t cg_dir(float* p, const float* r, int64_t C, size_t img, double* rs, const double* rs_new, const int* done, hipStream_t st);
bool ulpda_dual_rhs_supported(int H, int W);
hipError_t ulpda_dual_rhs(const float* xhat, const float* y_in, float* y_out, const float* x, const float* z, const float* htb, float* rhs, int64_t C,
                          int H, int W, float mu, float radius, int iso, float tau, float ts, hipStream_t st);
hipError_t cheb_count(int64_t C, const double* stat, double inv_alpha2, double tol, double inv_log_inv_c, int kmax, int* count,
                      hipStream_t st);
hipError_t cg_check(int64_t C, const double* rsv, const double* b2, double tol2, int* done, hipStream_t st);

// Two Chebyshev iterations per launch (lmc_cheb_pair.hip):  f1 = u_{k+1} = a0 cur - t0 sigma H^T H cur + b0 rhs + s0 prv,
// f2 (or f2_last) = u_{k+2} = a1 f1 - t1 sigma H^T H f1 + b1 rhs + s1 cur.  tg = t sigma_f c_u c_v (the launcher fills cbox and folds it in).
// The launch returns at once when *run_count <= run_index; it writes u_{k+2} to f2_last instead of f2 when force_last or *run_count <= last_index
// (no later pair of the solve will run).  cur, prv, rhs, f1, f2, f2_last: [C][H][W]; f1, f2, f2_last distinct from the inputs.
struct ChebPairArgs {
  int H, W, C;
  float cbox;
  const float* cur;
  const float* prv;
  const float* rhs;
  float* f1;
  float* f2;
  float* f2_last;
  float a0, tg0, b0, s0, a1, tg1, b1, s1;
  const int* run_count;
  int run_index, last_index, force_last;
  double* dot_out;      // non-NULL: dot_out[2c] += sum (f1 - cur)^2, dot_out[2c+1] += sum rhs^2 (the statistics the adaptive iteration count is made from)
};
bool cheb_pair_supported(int H, int W, const BlurTaps& taps);
bool cheb_pair_pays(int64_t C, int H);
hipError_t launch_cheb_pair(ChebPairArgs a, const BlurTaps& taps, hipStream_t st);
}  // namespace lmc

namespace lmc {
// Metropolis adjustment glue (lmc_mala.hip)
hipError_t mala_propose(const float* mx, const float* xi, float* xp, int64_t C, size_t img, float s, double* d1, hipStream_t st);
hipError_t mala_propose_philox(const float* mx, float* xp, int64_t C, int H, int W, float s, uint32_t key0, uint32_t key1,
                               uint32_t iteration, uint32_t chain_offset, double* d1, hipStream_t st);
hipError_t mala_accept(int C, double* U, const double* fp, const double* gp, float epsg, const double* d1, const double* d2, float tau,
                       uint32_t key0, uint32_t key1, uint32_t iteration, uint32_t chain_offset, int* flag,
                       unsigned long long* nacc, double* log_alpha, hipStream_t st);
hipError_t mala_select(const int* flag, float* x, float* mx, const float* xp, const float* mxp, int64_t C, size_t img, int when,
                       hipStream_t st);
}  // namespace lmc
