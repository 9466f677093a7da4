"""Closed-form proxes of the reference's prox.py (prox.py:18-85) as PRIORS of the fused MYULA step (LMC_PRIOR_EPROX, ABI 3): one MYULA step per
functor through the row-streaming, register-block and tiled step kernels against  a x - tau grad f(x) + b prox(x) + s xi  assembled from the
checker's data gradient and the prox values the reference's own functions returned (tests/golden/prox.npz)."""
import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu

SIGMA = 0.75
GAMMA, TAU = SIGMA ** 2, 0.2 * SIGMA ** 2

# (golden key, kind, params as the golden call had them, indices scaled by the prox parameter)
CASES = [("laplace_0.5", "laplace", (0.5,), (0,)), ("uncentered_laplace_0.7_1.5", "uncentered_laplace", (0.7, 1.5), (0,)),
         ("gaussian_0.3", "gaussian", (0.3,), (0,)), ("gen_gaussian_0.6_4_3", "gen_gaussian_4_3", (0.6,), (0,)),
         ("gen_gaussian_0.6_3_2", "gen_gaussian_3_2", (0.6,), (0,)), ("gen_gaussian_0.6_3", "gen_gaussian_3", (0.6,), (0,)),
         ("gen_gaussian_0.6_4", "gen_gaussian_4", (0.6,), (0,)), ("huber_0.5_0.4", "huber", (0.5, 0.4), (1,)),
         ("smoothed_laplace_0.9", "smoothed_laplace", (0.9,), (0,)), ("exp_0.5", "exp", (0.5,), (0,)), ("gamma_0.4_1.3", "gamma", (0.4, 1.3), (0, 1)),
         ("chi_0.7", "chi", (0.7,), ()), ("uniform_1.2", "uniform", (1.2,), ()), ("triangular_-0.5_0.8", "triangular", (-0.5, 0.8), ()),
         ("conjugate_laplace_0.8", "laplace_conj", (0.8,), ())]


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available()
    import lmc_atomi_amd as la
    return la


@pytest.mark.parametrize("key,kind,params,scaled", CASES)
@pytest.mark.parametrize("route", ["rows4", "rows8", "rows8k7", "block", "point"])
def test_one_myula_step_per_functor(la, golden, key, kind, params, scaled, route):
    g = golden("prox.npz")
    gx = g["xp"] if kind in ("exp", "gamma", "chi") and "xp" in g.files and False else g["x"]
    want_p = g[key]
    shape = {"rows4": (24, 136), "rows8": (16, 264), "rows8k7": (16, 264), "block": (16, 72), "point": (20, 50)}[route]
    C_ = 3
    rng = np.random.default_rng(len(key))
    idx = rng.integers(0, gx.size, size=(C_,) + shape)
    x = gx[idx]                                        # every pixel is one of the grid points the reference evaluated
    px = want_p[idx]
    noise = rng.standard_normal((1, C_) + shape)
    k = 7 if route == "rows8k7" else 5
    h = np.ones((k, k)) / (k * k)
    y = rng.normal(0, 1.0, shape)
    # the golden values are prox_X(x; params): with the prox parameter t = epsg * gamma folded into the scaled ones, give the class params / t there
    t = GAMMA
    cls_params = tuple(p / t if i in scaled else p for i, p in enumerate(params))
    pg = la.ElementwiseProx(kind, *cls_params, scaled=scaled)
    if route == "block":
        mask = (rng.uniform(size=shape) < 0.6).astype(np.float64)
        pf = la.L2(Op=la.Diagonal(mask, dims=shape), b=mask * y, sigma=1 / SIGMA ** 2, dims=shape)
        gf = (1 / SIGMA ** 2) * mask * (mask * x - mask * y)
    else:
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=(k // 2, k // 2)), b=y, sigma=1 / SIGMA ** 2)
        gf = np.stack([(1 / SIGMA ** 2) * O.blur_adjoint(O.blur(x[c], h, (k // 2, k // 2)) - y, h, (k // 2, k // 2)) for c in range(C_)])
    smp = la.MYULASampler(pf, pg, shape, n_chains=C_, tau=TAU, gamma=GAMMA, noise="injected", variant="point" if route == "point" else None)
    smp.set_state(x)
    smp.step(1, noise=noise)
    got = smp.get_state().cpu().numpy()
    name = smp.kernel_name
    smp.close()
    want = (1 - TAU / GAMMA) * x - TAU * gf + (TAU / GAMMA) * px + np.sqrt(2 * TAU) * noise[0]
    assert {"rows4": "rows", "rows8": "rows", "rows8k7": "rows", "block": "block", "point": "point"}[route] in name, name
    tol = 3e-5 if "gen_gaussian" in kind else 1e-5
    assert rel(got, want) < tol, (key, route, name, rel(got, want))
    # the stand-alone prox of the same object
    assert np.allclose(pg.prox(gx, t), want_p, rtol=1e-4 if "gen_gaussian" in kind else 2e-5, atol=2e-5)
