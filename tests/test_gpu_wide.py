"""Any image width on the full-width pipeline (round 2): rows that are not 16-byte aligned (W % 4 != 0: pixel-by-pixel global accesses,
the pixel in the last image column anywhere inside a lane) and images wider than 512 columns (column strips of 512 with recomputed
halos) -- the reference's optional `einstein` image is 667 x 877 (prox_lmc_deconv.py:44-46).  Against the checker with injected noise,
through the C ABI; K = 10 (niter_tv) and the chained 10-iteration links of the ME-TV term."""
import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu
SIGMA = 0.75
GAMMA, TAU = SIGMA ** 2, 0.2 * SIGMA ** 2


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


def synth(shape, k=5, off=None, seed=0):
    rng = np.random.default_rng(seed)
    img = np.zeros(shape)
    img[shape[0] // 5:shape[0] // 2, shape[1] // 6:2 * shape[1] // 3] = 160.0
    img[shape[0] // 2:, shape[1] // 2:] = 70.0
    img[:, -3:] += 40.0                                   # structure right at the last columns
    img += np.linspace(0, 25, shape[1])[None, :]
    h = np.ones((k, k)) / (k * k)
    off = (k // 2, k // 2) if off is None else off
    y = O.blur(img, h, off) + rng.normal(0, SIGMA, shape)
    return img, h, off, y


@pytest.mark.parametrize("shape,k", [((40, 877), 5), ((24, 1000), 5), ((30, 516), 5), ((33, 301), 5), ((20, 203), 5), ((16, 1500), 5),
                                     ((28, 877), 6), ((22, 645), 7), ((18, 131), 5)])
def test_myula_tv10_any_width(la, shape, k):
    img, h, off, y = synth(shape, k)
    rng = np.random.default_rng(shape[1])
    C_, nit = 2, 3
    x0 = img[None] + rng.normal(0, 8, (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / SIGMA ** 2)
    smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=C_, tau=TAU, gamma=GAMMA, noise="injected")
    smp.set_state(x0)
    smp.step(nit, noise=noise)
    assert "pipe" in smp.kernel_name, smp.kernel_name
    got = smp.get_state().cpu().numpy()
    prior = {"kind": "tv", "sigma": 0.3, "niter": 10, "t": GAMMA}
    ref = O.myula_batched(x0, y, h, off, 1 / SIGMA ** 2, TAU, GAMMA, prior, nit, lambda i: noise[i])
    err = rel(got, ref)
    assert err < 2e-5, err
    # column-wise: nothing special at the strip seams or in the last columns
    colerr = np.abs(got - ref).max(axis=(0, 1))
    assert colerr.max() < 2e-3, (int(colerr.argmax()), float(colerr.max()))
    smp.close()


def test_einstein_size_single_chain_philox(la):
    """667 x 877, one chain, Philox noise keyed by the GLOBAL column: two strips, unaligned rows."""
    shape = (667, 877)
    img, h, off, y = synth(shape, 5)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / SIGMA ** 2)
    seed = 77
    smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=1, tau=TAU, gamma=GAMMA, seed=seed, chain_offset=3)
    smp.set_state(np.zeros(shape))
    smp.step(2)
    assert "pipe" in smp.kernel_name
    got = smp.get_state().cpu().numpy()
    prior = {"kind": "tv", "sigma": 0.3, "niter": 10, "t": GAMMA}
    ref = O.myula_batched(np.zeros((1,) + shape), y, h, off, 1 / SIGMA ** 2, TAU, GAMMA, prior, 2,
                          lambda i: O.philox_normals(seed, i, np.arange(3, 4), *shape).astype(np.float64))
    assert rel(got, ref) < 5e-5, rel(got, ref)
    smp.close()


@pytest.mark.parametrize("shape", [(24, 877), (20, 1032)])
def test_ncvx_terms_and_chained_prox_on_wide_images(la, shape):
    """MC-TV (evaluated in the combine wave) and ME-TV (50-iteration inner prox = chained 10-iteration links through HBM state) across strips."""
    img, h, off, y = synth(shape, 5)
    rng = np.random.default_rng(5)
    x0 = img[None] + rng.normal(0, 8, (2,) + shape)
    noise = rng.standard_normal((2, 2) + shape)
    Hop = la.Convolve2D(shape, h, offset=off)
    oH = O.Convolve2D(shape, h, off)
    for op2 in (la.Gradient(shape), None):
        pf = la.L2_ncvx_tv(dims=shape, Op=Hop, Op2=op2, b=y.ravel(), sigma=1 / SIGMA ** 2, lamda=0.3, gamma=15.0, isotropic=True, niter=50, rtol=0.0)
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=2, tau=TAU, gamma=GAMMA, noise="injected")
        smp.set_state(x0)
        smp.step(2, noise=noise)
        got = smp.get_state().cpu().numpy()
        opf = O.L2NcvxTV(dims=shape, Op=oH, Op2=O.Gradient(shape) if op2 is not None else None, b=y.ravel(), sigma=1 / SIGMA ** 2, lamda=0.3,
                         gamma=15.0, isotropic=True, niter=50)
        otv = O.TV(shape, sigma=0.3, niter=10)
        ref = np.stack([O.myula(opf, otv, x0[c].ravel(), TAU, GAMMA, niter=2, noise=[noise[0, c].ravel(), noise[1, c].ravel()])[-1].reshape(shape)
                        for c in range(2)])
        assert rel(got, ref) < 5e-5, (op2 is not None, rel(got, ref))
        smp.close()


@pytest.mark.parametrize("shape,k", [((40, 877), 5), ((24, 1000), 5), ((30, 516), 5), ((33, 301), 7), ((20, 203), 5), ((16, 1500), 6), ((21, 1021), 7)])
@pytest.mark.parametrize("prior", ["l2", "l1", "none"])
def test_row_streaming_kernel_any_width(la, shape, k, prior):
    """Closed-form priors + separable blur on the row-streaming kernel: W % 4 != 0 pixel by pixel, W > 512 as column strips of 496 + 2 x 8."""
    img, h, off, y = synth(shape, k)
    rng = np.random.default_rng(shape[1] + k)
    C_, nit = 2, 3
    x0 = img[None] + rng.normal(0, 8, (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / SIGMA ** 2)
    pg = {"l2": lambda: la.L2(sigma=0.02), "l1": lambda: la.L1(sigma=0.8), "none": lambda: None}[prior]()
    smp = la.MYULASampler(pf, pg, shape, n_chains=C_, tau=TAU, gamma=GAMMA, noise="injected")
    smp.set_state(x0)
    smp.step(nit, noise=noise)
    assert "rows" in smp.kernel_name, smp.kernel_name
    got = smp.get_state().cpu().numpy()
    op = {"l2": {"kind": "l2", "sigma": 0.02, "t": GAMMA}, "l1": {"kind": "l1", "sigma": 0.8, "t": GAMMA}, "none": {"kind": "none"}}[prior]
    ref = O.myula_batched(x0, y, h, off, 1 / SIGMA ** 2, TAU, GAMMA, op, nit, lambda i: noise[i])
    assert rel(got, ref) < 5e-6, rel(got, ref)
    colerr = np.abs(got - ref).max(axis=(0, 1))
    assert colerr.max() < 1e-3, (int(colerr.argmax()), float(colerr.max()))
    smp.close()


def test_row_streaming_philox_on_a_wide_unaligned_image(la):
    shape = (64, 877)
    img, h, off, y = synth(shape, 5)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / SIGMA ** 2)
    seed = 9
    smp = la.MYULASampler(pf, la.L2(sigma=0.02), shape, n_chains=3, tau=TAU, gamma=GAMMA, seed=seed, chain_offset=5, moments=True)
    smp.step(4)
    assert "rows" in smp.kernel_name
    got = smp.get_state().cpu().numpy()
    ref, s1, s2, cnt = O.myula_batched(np.zeros((3,) + shape), y, h, off, 1 / SIGMA ** 2, TAU, GAMMA, {"kind": "l2", "sigma": 0.02, "t": GAMMA}, 4,
                                       lambda i: O.philox_normals(seed, i, np.arange(5, 8), *shape).astype(np.float64), moments=True)
    assert rel(got, ref) < 1e-5
    m1, m2, n = smp.moments()
    assert n == cnt and rel(m1.cpu().numpy(), s1) < 1e-5 and rel(m2.cpu().numpy(), s2) < 1e-5
    smp.close()


@pytest.mark.parametrize("shape", [(20, 877), (18, 1000)])
def test_ulpda_implicit_step_on_wide_images(la, shape):
    """ULPDA (algs.py:425-449): the implicit data step (Chebyshev on the row-streaming operator, in-place three-term recurrence) across strips."""
    img, h, off, y = synth(shape, 5)
    C_, seed, nit, cho = 2, 4, 4, 2
    l2 = la.L2(Op=la.Convolve2D(shape, h), b=y.ravel(), sigma=1 / SIGMA ** 2, niter=50, warm=True)
    smp = la.ULPDASampler(l2, la.L21(sigma=0.3), la.Gradient(shape), shape, n_chains=C_, tau=0.95 * GAMMA, mu=1.0, theta=1.0, gfirst=False, seed=seed,
                          chain_offset=cho)
    smp.step(nit)
    got = smp.get_state().cpu().numpy()
    Gop = O.Gradient(shape)
    for c in range(C_):
        l2o = O.L2(Op=O.Convolve2D(shape, h), b=y.ravel(), sigma=1 / SIGMA ** 2, niter=50, warm=True)
        noise = np.stack([O.philox_normals(seed, k_, [cho + c], *shape)[0].ravel().astype(np.float64) for k_ in range(nit)])
        xs = O.ulpda(l2o, O.L21(sigma=0.3), Gop, np.zeros(shape[0] * shape[1]), 0.95 * GAMMA, 1.0, theta=1.0, niter=nit, gfirst=False, noise=noise)
        assert rel(got[c].ravel(), xs[-1]) < 2e-4, (c, rel(got[c].ravel(), xs[-1]))
    smp.close()


@pytest.mark.parametrize("K,warm", [(9, False), (2, False), (8, False), (1, True), (2, True), (3, True)])
def test_other_pipeline_instantiations_across_strips(la, K, warm):
    """Odd / short dual-iteration counts and the warm-dual prox on a 1000-column image (three strips; their halo follows K)."""
    shape = (24, 1000)
    img, h, off, y = synth(shape, 5)
    rng = np.random.default_rng(K)
    C_, nit = 2, 4
    x0 = img[None] + rng.normal(0, 8, (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / SIGMA ** 2)
    smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=K, warm=warm), shape, n_chains=C_, tau=TAU, gamma=GAMMA, noise="injected")
    smp.set_state(x0)
    smp.step(nit, noise=noise)
    assert "pipe" in smp.kernel_name, smp.kernel_name
    got = smp.get_state().cpu().numpy()
    prior = {"kind": "tv", "sigma": 0.3, "niter": K, "t": GAMMA}
    if warm:
        prior["warm"] = True
    ref = O.myula_batched(x0, y, h, off, 1 / SIGMA ** 2, TAU, GAMMA, prior, nit, lambda i: noise[i])
    assert rel(got, ref) < 3e-5, rel(got, ref)
    colerr = np.abs(got - ref).max(axis=(0, 1))
    assert colerr.max() < 3e-3, (int(colerr.argmax()), float(colerr.max()))
    smp.close()
