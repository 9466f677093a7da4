/* oracle/lmc_oracle_c.c -- plain C (float64) restatement of the MYULA step, second checker and all-cores CPU baseline.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under lmc_atomi_amd/ may link, load or call this file; it is used by tests/
 * (agreement with oracle/lmc_oracle.py, which is pinned by the reference's own outputs, tests/golden/) and by
 * bench.py's cpu_baseline leg.  It follows the same reference lines as the numpy restatement:
 *
 *   MYULA update            algs.py:564-570   x <- (1-tau/gamma) x - tau grad f(x) + (tau/gamma) prox_{eps gamma g}(x) + sqrt(2 tau) xi
 *   data gradient           algs.py:283-284   sigma_f H^T (H x - y)          (H = blur | diagonal mask | identity)
 *   blur / adjoint          prox_lmc_deconv.py:55-69   zero-padded "same" convolution with origin `offset`, adjoint = correlation
 *   TV prox                 prox_lmc_deconv.py:122     K fast-gradient-projection dual iterations (restated in lmc_oracle.py:tv_prox_fgp)
 *   closed-form priors      prox.py:18-27              soft threshold / x / (1 + t sigma)
 *
 * The arithmetic is written in the order numpy evaluates lmc_oracle.py (tap order, left-to-right sums, no FMA
 * contraction: build with -ffp-contract=off), so the two restatements agree to the last bit on the same noise.
 * The noise xi is an input (numpy's PCG64 standard_normal, the reference's stream).
 * Chains are independent: OpenMP over images.
 */
#include <math.h>
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { OC_PRIOR_NONE = 0, OC_PRIOR_L2 = 1, OC_PRIOR_L1 = 2, OC_PRIOR_TV = 3 };
enum { OC_DATA_NONE = 0, OC_DATA_IDENTITY = 1, OC_DATA_MASK = 2, OC_DATA_BLUR = 3 };

typedef struct {
    int H, W;
    int data_kind;            /* OC_DATA_* */
    const double* y;          /* [H,W] observation (shared by all chains) */
    const double* mask;       /* [H,W] when data_kind == MASK */
    const double* h;          /* [kh,kw] blur kernel when data_kind == BLUR */
    int kh, kw, oy, ox;
    double sigma_f, tau, gamma;
    int prior_kind;           /* OC_PRIOR_* */
    double prior_sigma, t;    /* prox parameter t = epsg * gamma */
    int tv_niter;
    double tv_step;
    const double* betas;      /* [tv_niter] momentum table (lmc_oracle.fgp_betas) */
    double tv_rtol;           /* > 0: pyproximal.TV's per-image early exit on the relative change of the primal objective (its default 1e-4, which the
                               * reference's calls at prox_lmc_deconv.py:122 and algs.py:169 leave in force); lmc_oracle.tv_prox_fgp(rtol=...) */
    int* tv_passes;           /* optional [n_img] output: the pass each image's prox left in (tv_niter = ran out of passes) */
} oc_step_config;

/* (Hx)[i,j] = sum_{a,b} h[a,b] x[i-a+oy, j-b+ox], zero outside (lmc_oracle.blur).  Per pixel the taps are added in (a,b)
 * order exactly as numpy's shifted-slice accumulation does; the j loop is innermost so that it vectorises. */
static void blur_img(const double* x, double* out, int H, int W, const double* h, int kh, int kw, int oy, int ox) {
    for (int i = 0; i < H; ++i) {
        double* o = out + (size_t)i * W;
        for (int j = 0; j < W; ++j) o[j] = 0.0;
        for (int a = 0; a < kh; ++a) {
            const int ii = i + oy - a;
            if (ii < 0 || ii >= H) continue;
            const double* xr = x + (size_t)ii * W;
            for (int b = 0; b < kw; ++b) {
                const int d = ox - b;                         /* o[j] += h * xr[j + d] */
                const int j0 = d < 0 ? -d : 0, j1 = d > 0 ? W - d : W;
                const double hv = h[a * kw + b];
                for (int j = j0; j < j1; ++j) o[j] += hv * xr[j + d];
            }
        }
    }
}

/* (H^T r)[m,n] = sum_{a,b} h[a,b] r[m+a-oy, n+b-ox] (lmc_oracle.blur_adjoint) */
static void blur_adj_img(const double* r, double* out, int H, int W, const double* h, int kh, int kw, int oy, int ox) {
    for (int i = 0; i < H; ++i) {
        double* o = out + (size_t)i * W;
        for (int j = 0; j < W; ++j) o[j] = 0.0;
        for (int a = 0; a < kh; ++a) {
            const int ii = i + a - oy;
            if (ii < 0 || ii >= H) continue;
            const double* rr = r + (size_t)ii * W;
            for (int b = 0; b < kw; ++b) {
                const int d = b - ox;
                const int j0 = d < 0 ? -d : 0, j1 = d > 0 ? W - d : W;
                const double hv = h[a * kw + b];
                for (int j = j0; j < j1; ++j) o[j] += hv * rr[j + d];
            }
        }
    }
}

/* One row of sol = x - gamma div(rr,ss); div(r,s)[i,j] = r[i,j]-r[i-1,j] + s[i,j]-s[i,j-1] with the last row of r / last column
 * of s taken as zero (lmc_oracle.div2d; the sum starts from 0.0 and runs left to right like its four in-place updates).
 * Row conditions are hoisted, the border columns are peeled off, so the interior loop vectorises. */
static void primal_row(const double* x, const double* rr, const double* ss, double* sol, int H, int W, int i, double gamma) {
    const double* xr = x + (size_t)i * W;
    const double* r0 = rr + (size_t)i * W;
    const double* rm = rr + (size_t)(i > 0 ? i - 1 : 0) * W;
    const double* s0 = ss + (size_t)i * W;
    double* o = sol + (size_t)i * W;
    const double dn = i < H - 1 ? 1.0 : 0.0, up = i > 0 ? 1.0 : 0.0;   /* exact: multiplying by 1.0 or skipping the term */
    for (int j = 0; j < W; ++j) {
        double d = 0.0;
        if (dn != 0.0) d += r0[j];
        if (up != 0.0) d -= rm[j];
        if (j < W - 1) d += s0[j];
        if (j > 0) d -= s0[j - 1];
        o[j] = xr[j] - gamma * d;
    }
}

/* numpy's pairwise summation of a contiguous float64 array (numpy/_core/src/umath/loops_utils.h.src: blocks of 128, eight partial sums),
 * so that the objective below is the number `float(np.sum(...))` gives in lmc_oracle.tv_prox_fgp -- the early-exit test compares it with rtol */
static double np_pairwise_sum(const double* a, size_t n) {
    if (n < 8) {
        double res = 0.0;
        for (size_t i = 0; i < n; ++i) res += a[i];
        return res;
    } else if (n <= 128) {
        double r[8];
        size_t i;
        for (int j = 0; j < 8; ++j) r[j] = a[j];
        for (i = 8; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] += a[i + j];
        double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
        for (; i < n; ++i) res += a[i];
        return res;
    } else {
        size_t n2 = n / 2;
        n2 -= n2 % 8;
        return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
    }
}

/* primal objective of the iterate `sol`: 0.5 ||x - sol||^2 + gamma TV_iso(sol)  (lmc_oracle.tv_prox_fgp, rtol branch); tmp = 1 image */
static double tv_objective_img(const double* x, const double* sol, int H, int W, double gamma, double* tmp) {
    const size_t n = (size_t)H * W;
    for (size_t e = 0; e < n; ++e) { const double d = x[e] - sol[e]; tmp[e] = d * d; }
    const double sq = np_pairwise_sum(tmp, n);
    for (int i = 0; i < H; ++i) {
        const double* so = sol + (size_t)i * W;
        const double* sd = sol + (size_t)(i < H - 1 ? i + 1 : i) * W;
        double* t = tmp + (size_t)i * W;
        for (int j = 0; j < W; ++j) {
            const double dr = sd[j] - so[j];
            const double dc = (j < W - 1) ? so[j + 1] - so[j] : 0.0;
            t[j] = sqrt(dr * dr + dc * dc);
        }
    }
    return 0.5 * sq + gamma * np_pairwise_sum(tmp, n);
}

/* prox_{gamma TV}(x) by niter FGP dual iterations (lmc_oracle.tv_prox_fgp); rtol > 0: with upstream's early exit -- the iterate formed at
 * the top of a pass is returned as soon as the relative change of its objective is below rtol (never in the first pass).  Returns the pass
 * the image left in (niter: ran out of passes).  work = 5 images (6 with rtol > 0) */
static int tv_prox_img(const double* x, double* out, int H, int W, double gamma, int niter, double step, const double* betas,
                       double rtol, double* work) {
    const size_t n = (size_t)H * W;
    double *rr = work, *ss = work + n, *p = work + 2 * n, *q = work + 3 * n, *sol = work + 4 * n;
    memset(work, 0, 4 * n * sizeof(double));
    const double c = step / gamma;
    double prev_obj = 0.0;
    for (int k = 0; k < niter; ++k) {
        for (int i = 0; i < H; ++i) primal_row(x, rr, ss, sol, H, W, i, gamma);
        if (rtol > 0.0) {
            const double obj = tv_objective_img(x, sol, H, W, gamma, work + 5 * n);
            const double rel = (k > 0 && obj > 0.0) ? fabs(obj - prev_obj) / obj : 2.0 * rtol;
            prev_obj = obj;
            if (rel < rtol) {
                memcpy(out, sol, n * sizeof(double));
                return k;
            }
        }
        const double beta = betas[k];
        for (int i = 0; i < H; ++i) {
            const double* so = sol + (size_t)i * W;
            const double* sd = sol + (size_t)(i < H - 1 ? i + 1 : i) * W;      /* last row: dr = so - so = 0 */
            double *rr_ = rr + (size_t)i * W, *ss_ = ss + (size_t)i * W, *p_ = p + (size_t)i * W, *q_ = q + (size_t)i * W;
            for (int j = 0; j < W; ++j) {
                const double dr = sd[j] - so[j];
                const double dc = (j < W - 1) ? so[j + 1] - so[j] : 0.0;
                const double r = rr_[j] - c * dr, s = ss_[j] - c * dc;
                const double nrm = sqrt(r * r + s * s);
                const double w = nrm > 1.0 ? nrm : 1.0;
                const double pn = r / w, qn = s / w;
                rr_[j] = pn + beta * (pn - p_[j]);
                ss_[j] = qn + beta * (qn - q_[j]);
                p_[j] = pn;
                q_[j] = qn;
            }
        }
    }
    for (int i = 0; i < H; ++i) primal_row(x, rr, ss, out, H, W, i, gamma);
    return niter;
}

static void step_img(const oc_step_config* c, const double* x, const double* xi, double* out, double* work, int* passes) {
    const int H = c->H, W = c->W;
    const size_t n = (size_t)H * W;
    double *g = work, *tmp = work + n, *px = work + 2 * n, *tvw = work + 3 * n;
    switch (c->data_kind) {
        case OC_DATA_BLUR:
            blur_img(x, tmp, H, W, c->h, c->kh, c->kw, c->oy, c->ox);
            for (size_t e = 0; e < n; ++e) tmp[e] = tmp[e] - c->y[e];
            blur_adj_img(tmp, g, H, W, c->h, c->kh, c->kw, c->oy, c->ox);
            for (size_t e = 0; e < n; ++e) g[e] = c->sigma_f * g[e];
            break;
        case OC_DATA_MASK:
            for (size_t e = 0; e < n; ++e) g[e] = c->sigma_f * (c->mask[e] * (c->mask[e] * x[e] - c->y[e]));
            break;
        case OC_DATA_IDENTITY:
            for (size_t e = 0; e < n; ++e) g[e] = c->sigma_f * (x[e] - c->y[e]);
            break;
        default:
            memset(g, 0, n * sizeof(double));
    }
    switch (c->prior_kind) {
        case OC_PRIOR_L2: {
            const double d = 1.0 + c->t * c->prior_sigma;
            for (size_t e = 0; e < n; ++e) px[e] = x[e] / d;
        } break;
        case OC_PRIOR_L1: {
            const double thr = c->t * c->prior_sigma;
            for (size_t e = 0; e < n; ++e) {
                const double m = fabs(x[e]) - thr;
                const double sg = (x[e] > 0.0) - (x[e] < 0.0);
                px[e] = sg * (m > 0.0 ? m : 0.0);
            }
        } break;
        case OC_PRIOR_TV: {
            const int left = tv_prox_img(x, px, H, W, c->t * c->prior_sigma, c->tv_niter, c->tv_step, c->betas, c->tv_rtol, tvw);
            if (passes) *passes = left;
        } break;
        default:
            memcpy(px, x, n * sizeof(double));
    }
    const double a = 1.0 - c->tau / c->gamma, b = c->tau / c->gamma, sn = sqrt(2.0 * c->tau);
    for (size_t e = 0; e < n; ++e) out[e] = a * x[e] - c->tau * g[e] + b * px[e] + sn * xi[e];
}

/* ---- exported ---- */

int lmc_oc_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* one MYULA step of n_img independent chains: x, xi, out are [n_img,H,W] float64 (out may not alias x) */
int lmc_oc_myula_step(const oc_step_config* c, const double* x, const double* xi, double* out, int n_img, int n_threads) {
    if (!c || !x || !xi || !out || c->H <= 0 || c->W <= 0 || n_img < 0) return -1;
    if (c->prior_kind == OC_PRIOR_TV && (c->tv_niter < 0 || (c->tv_niter > 0 && !c->betas))) return -1;
    const size_t n = (size_t)c->H * c->W;
    if (n_threads < 1) n_threads = 1;
    int fail = 0;
#pragma omp parallel num_threads(n_threads)
    {
        double* work = (double*)malloc(9 * n * sizeof(double));
        if (!work) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp barrier
        if (!fail) {
#pragma omp for schedule(dynamic, 1)
            for (int im = 0; im < n_img; ++im)
                step_img(c, x + (size_t)im * n, xi + (size_t)im * n, out + (size_t)im * n, work, c->tv_passes ? c->tv_passes + im : NULL);
        }
        free(work);
    }
    return fail ? -2 : 0;
}

/* rtol > 0: with the early exit; passes (optional, [n_img]): the pass each image left in */
int lmc_oc_tv_prox(const double* x, double* out, int n_img, int H, int W, double gamma, int niter, double step, const double* betas,
                   double rtol, int* passes, int n_threads) {
    if (!x || !out || H <= 0 || W <= 0 || niter < 0 || (niter > 0 && !betas)) return -1;
    const size_t n = (size_t)H * W;
    if (n_threads < 1) n_threads = 1;
    int fail = 0;
#pragma omp parallel num_threads(n_threads)
    {
        double* work = (double*)malloc(6 * n * sizeof(double));
        if (!work) {
#pragma omp atomic write
            fail = 1;
        }
#pragma omp barrier
        if (!fail) {
#pragma omp for schedule(dynamic, 1)
            for (int im = 0; im < n_img; ++im) {
                const int left = tv_prox_img(x + (size_t)im * n, out + (size_t)im * n, H, W, gamma, niter, step, betas, rtol, work);
                if (passes) passes[im] = left;
            }
        }
        free(work);
    }
    return fail ? -2 : 0;
}

int lmc_oc_blur(const double* x, double* out, int n_img, int H, int W, const double* h, int kh, int kw, int oy, int ox, int adjoint) {
    if (!x || !out || !h || H <= 0 || W <= 0) return -1;
    const size_t n = (size_t)H * W;
    for (int im = 0; im < n_img; ++im) {
        if (adjoint)
            blur_adj_img(x + (size_t)im * n, out + (size_t)im * n, H, W, h, kh, kw, oy, ox);
        else
            blur_img(x + (size_t)im * n, out + (size_t)im * n, H, W, h, kh, kw, oy, ox);
    }
    return 0;
}
