"""GPU parity of the "pipe" step kernel (TV stages spread over the waves of a workgroup, one wave = full image width)
against the oracle step and the other step kernels, through the C ABI."""
import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available()
    import lmc_atomi_amd as la
    yield la
    la.set_step_variant("auto")


def problem(shape, rng, k=5):
    img = np.zeros(shape)
    img[shape[0] // 5:shape[0] // 2 + 1, shape[1] // 4:shape[1] // 2 + 2] = 190.0
    img += np.linspace(0, 30, shape[1])[None, :]
    h = np.ones((k, k)) / (k * k)
    off = (k // 2, k // 2)
    y = O.blur(img, h, off) + rng.normal(0, 0.75, shape)
    return img, h, off, y


@pytest.mark.parametrize("shape,k", [((40, 512), 5), ((37, 264), 5), ((100, 320), 5), ((5, 512), 5), ((1, 512), 5), ((23, 504), 5),
                                     ((40, 512), 7), ((33, 400), 6), ((30, 512), 3),
                                     ((64, 256), 5), ((30, 132), 5), ((9, 200), 3), ((50, 252), 7), ((256, 256), 5)])
def test_pipe_kernel_matches_oracle_step(la, shape, k):
    sigma, tau_reg = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    rng = np.random.default_rng(21)
    C, nit = 2, 3
    img, h, off, y = problem(shape, rng, k)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    pg = la.TV(shape, sigma=tau_reg, niter=10)
    op = {"kind": "tv", "sigma": tau_reg, "niter": 10, "t": gamma}
    x0 = img[None] + rng.normal(0, 10, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    la.set_step_variant("pipe")
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
    smp.set_state(x0)
    x = x0.copy()
    for it in range(nit):
        smp.step(1, noise=noise[it:it + 1])
        x = O.myula_step(x, y, h, off, 1 / sigma ** 2, tau, gamma, op, noise[it])
        got = smp.get_state().cpu().numpy()
        assert rel(got, x) < 2e-6 * (it + 1), (it, rel(got, x))
    assert smp.kernel_name == "myula_step_pipe_kernel"
    smp.close()
    la.set_step_variant("auto")


def _gauss(k, s=1.0):
    t = np.exp(-0.5 * ((np.arange(k) - k // 2) / s) ** 2)
    return np.outer(t, t) / np.sum(np.outer(t, t))


@pytest.mark.parametrize("name,h,off,expect_pipe", [
    ("gaussian5", _gauss(5), (2, 2), True),                                  # BASELINE's "Gaussian-deblur": separable, symmetric
    ("gaussian7", _gauss(7, 1.5), (3, 3), True),
    ("asymmetric5", np.outer([1.0, 2.0, 4.0, 3.0, 0.5], [0.2, 1.0, 3.0, 0.7, 0.1]) / 66.0, (2, 2), True),   # conv vs corr flips
    ("asymmetric_offcentre", np.outer([1.0, 2.0, 4.0], [3.0, 1.0, 0.5, 0.25]) / 33.25, (0, 3), True),
    ("non_separable", np.array([[0.0, 1, 0], [1, 4, 1], [0, 1, 2]]) / 10.0, (1, 1), False),                  # rank 2: general kernels
])
def test_non_uniform_blur_kernels(la, name, h, off, expect_pipe):
    """Runtime blur taps: Gaussian, asymmetric separable (catches a convolution / correlation mix-up of H and H^T) and a
    non-separable kernel, which the pipe kernel must refuse and another kernel must get right."""
    sigma, tau_reg = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    rng = np.random.default_rng(77)
    shape, C, nit = (37, 264), 2, 3
    img = problem(shape, rng)[0]
    y = O.blur(img, h, off) + rng.normal(0, sigma, shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    op = {"kind": "tv", "sigma": tau_reg, "niter": 10, "t": gamma}
    x0 = img[None] + rng.normal(0, 10, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    la.set_step_variant("auto")
    smp = la.MYULASampler(pf, la.TV(shape, sigma=tau_reg, niter=10), shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
    smp.set_state(x0)
    x = x0.copy()
    for it in range(nit):
        smp.step(1, noise=noise[it:it + 1])
        x = O.myula_step(x, y, h, off, 1 / sigma ** 2, tau, gamma, op, noise[it])
        got = smp.get_state().cpu().numpy()
        assert rel(got, x) < 2e-6 * (it + 1), (name, it, rel(got, x))
    assert (smp.kernel_name == "myula_step_pipe_kernel") == expect_pipe, (name, smp.kernel_name)
    smp.close()


def test_pipe_kernel_philox_matches_split(la):
    rng = np.random.default_rng(4)
    for shape, C in [((128, 512), 5), ((64, 384), 3)]:
        img, h, off, y = problem(shape, rng)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / 0.75 ** 2)
        outs = {}
        for v in ("split", "pipe"):
            la.set_step_variant(v)
            smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=C, tau=0.1125, gamma=0.5625, seed=3,
                                  chain_offset=11)
            smp.set_state(img)
            smp.step(5)
            outs[v] = smp.get_state().cpu().numpy()
            smp.close()
        assert rel(outs["pipe"], outs["split"]) < 2e-6, (shape, rel(outs["pipe"], outs["split"]))
    la.set_step_variant("auto")


def test_pipe_kernel_with_me_tv_term_matches_tile(la):
    """The ME-TV term of L2_ncvx_tv (pointwise `extra` input of the combine) through the pipe kernel on a wide image."""
    rng = np.random.default_rng(9)
    shape = (48, 512)
    img, h, off, y = problem(shape, rng)
    outs = {}
    for v in ("tile", "auto"):
        la.set_step_variant(v)
        pf = la.L2_ncvx_tv(dims=shape, Op=la.Convolve2D(shape, h, offset=off), b=y.ravel(), sigma=1 / 0.75 ** 2, lamda=0.3, gamma=15.0,
                           isotropic=True, niter=10, rtol=0.0)
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=2, tau=0.1125, gamma=0.5625, seed=1)
        smp.set_state(img)
        smp.step(3)
        outs[v] = smp.get_state().cpu().numpy()
        if v == "auto":
            assert smp.kernel_name == "myula_step_pipe_kernel"
        smp.close()
    la.set_step_variant("auto")
    assert rel(outs["auto"], outs["tile"]) < 3e-6, rel(outs["auto"], outs["tile"])


@pytest.mark.parametrize("shape,K,k", [((40, 264), 20, 5), ((33, 160), 50, 5), ((21, 512), 30, 5), ((24, 512), 20, 7), ((19, 200), 20, 6)])
def test_pipe_chained_launches_long_tv_prox(la, shape, K, k):
    """More than 10 dual iterations: a chain of launches handing the dual state (rr, ss, p, q) over in HBM -- the TV prox alone
    (no data term) against the oracle, and the full update (blur + TV(K) + noise) against the tiled kernel's exact chunks."""
    rng = np.random.default_rng(31)
    img, h, off, y = problem(shape, rng, k)
    x = img[None] + rng.normal(0, 8, (2,) + shape)
    la.set_step_variant("auto")
    tv = la.TV(shape, sigma=0.3, niter=K)
    got = np.stack([tv.prox(x[i].ravel().copy(), 0.7).reshape(shape) for i in range(2)])
    want = np.stack([O.tv_prox_fgp(x[i], 0.3 * 0.7, K) for i in range(2)])
    assert rel(got, want) < 2e-6 * K, rel(got, want)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / 0.75 ** 2)
    outs = {}
    for v in ("tile", "pipe"):
        la.set_step_variant(v)
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=K), shape, n_chains=2, tau=0.1125, gamma=0.5625, seed=6)
        smp.set_state(x)
        smp.step(2)
        outs[v] = smp.get_state().cpu().numpy()
        assert v in smp.kernel_name
        smp.close()
    la.set_step_variant("auto")
    assert rel(outs["pipe"], outs["tile"]) < 2e-6 * K, rel(outs["pipe"], outs["tile"])


def test_pipe_me_tv_inner_prox_chained(la):
    """L2_ncvx_tv ME-TV with niter = 20 inner iterations on a wide image: the inner prox runs as two chained launches."""
    rng = np.random.default_rng(10)
    shape = (30, 264)
    img, h, off, y = problem(shape, rng)
    outs = {}
    for v in ("tile", "auto"):
        la.set_step_variant(v)
        pf = la.L2_ncvx_tv(dims=shape, Op=la.Convolve2D(shape, h, offset=off), b=y.ravel(), sigma=1 / 0.75 ** 2, lamda=0.3, gamma=15.0,
                           isotropic=True, niter=20, rtol=0.0)
        outs[("g", v)] = pf.grad((img + 3.0).ravel())
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=2, tau=0.1125, gamma=0.5625, seed=1)
        smp.set_state(img)
        smp.step(2)
        outs[v] = smp.get_state().cpu().numpy()
        smp.close()
    la.set_step_variant("auto")
    assert rel(outs[("g", "auto")], outs[("g", "tile")]) < 1e-5
    assert rel(outs["auto"], outs["tile"]) < 5e-6


@pytest.mark.parametrize("shape", [(40, 264), (33, 160), (64, 512)])
def test_pipe_pure_tv_prox_k10(la, shape):
    """prox_{gamma TV} alone (no data term, K = 10) on wide images: the KT = 0 instantiation, against the oracle."""
    rng = np.random.default_rng(8)
    x = rng.uniform(0, 255, (3,) + shape)
    la.set_step_variant("auto")
    tv = la.TV(shape, sigma=0.3, niter=10)
    got = np.stack([tv.prox(x[i].ravel().copy(), 2.5).reshape(shape) for i in range(3)])
    want = np.stack([O.tv_prox_fgp(x[i], 0.3 * 2.5, 10) for i in range(3)])
    assert rel(got, want) < 2e-6, rel(got, want)
    la.set_step_variant("tile")
    ref = np.stack([tv.prox(x[i].ravel().copy(), 2.5).reshape(shape) for i in range(3)])
    la.set_step_variant("auto")
    assert rel(got, ref) < 2e-6


@pytest.mark.parametrize("K", [2, 6, 8])
@pytest.mark.parametrize("shape", [(40, 512), (33, 160)])
def test_pipe_kernel_fewer_dual_iterations(la, shape, K):
    """2, 4, 6, 8 dual iterations: the same kernel with fewer TV waves, against the oracle step and the tiled kernel."""
    sigma, tau_reg = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    rng = np.random.default_rng(40 + K)
    img, h, off, y = problem(shape, rng)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    op = {"kind": "tv", "sigma": tau_reg, "niter": K, "t": gamma}
    x0 = img[None] + rng.normal(0, 10, (2,) + shape)
    noise = rng.standard_normal((2, 2) + shape)
    outs = {}
    for v in ("tile", "auto"):
        la.set_step_variant(v)
        smp = la.MYULASampler(pf, la.TV(shape, sigma=tau_reg, niter=K), shape, n_chains=2, tau=tau, gamma=gamma, noise="injected")
        smp.set_state(x0)
        x = x0.copy()
        for it in range(2):
            smp.step(1, noise=noise[it:it + 1])
            x = O.myula_step(x, y, h, off, 1 / sigma ** 2, tau, gamma, op, noise[it])
            assert rel(smp.get_state().cpu().numpy(), x) < 2e-6 * (it + 1)
        if v == "auto":
            assert smp.kernel_name == "myula_step_pipe_kernel"
        outs[v] = smp.get_state().cpu().numpy()
        smp.close()
    la.set_step_variant("auto")
    assert rel(outs["auto"], outs["tile"]) < 2e-6


@pytest.mark.parametrize("data", ["mask", "identity"])
@pytest.mark.parametrize("shape,K", [((40, 264), 10), ((33, 512), 10), ((21, 200), 6), ((5, 136), 2), ((1, 256), 10)])
def test_pipe_kernel_pointwise_data_terms(la, data, shape, K):
    """Inpainting (diagonal mask) and denoising (identity) data terms with the TV prior: their gradient is formed in the load wave of the
    instantiation without a blur; against the oracle step with injected noise."""
    sigma, tau_reg = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    rng = np.random.default_rng(61)
    C, nit = 2, 3
    img = problem(shape, rng)[0]
    mask = (rng.uniform(size=shape) < 0.6).astype(np.float64) if data == "mask" else None
    y = (mask * img if mask is not None else img) + rng.normal(0, sigma, shape) * (mask if mask is not None else 1.0)
    if mask is not None:
        pf = la.L2(Op=la.Diagonal(mask, dims=shape), b=y, sigma=1 / sigma ** 2, dims=shape)
    else:
        pf = la.L2(b=y, sigma=1 / sigma ** 2, dims=shape)
    op = {"kind": "tv", "sigma": tau_reg, "niter": K, "t": gamma}
    x0 = img[None] + rng.normal(0, 10, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    la.set_step_variant("auto")
    smp = la.MYULASampler(pf, la.TV(shape, sigma=tau_reg, niter=K), shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
    smp.set_state(x0)
    x = x0.copy()
    for it in range(nit):
        smp.step(1, noise=noise[it:it + 1])
        x = O.myula_step(x, y, None, None, 1 / sigma ** 2, tau, gamma, op, noise[it], mask=mask)
        got = smp.get_state().cpu().numpy()
        assert rel(got, x) < 2e-6 * (it + 1), (it, rel(got, x))
    assert smp.kernel_name == "myula_step_pipe_kernel", smp.kernel_name
    smp.close()
