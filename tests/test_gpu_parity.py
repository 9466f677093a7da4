"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (ctypes), against the
oracle on the same seeded inputs and against the golden vectors of the reference's own loops.

Tolerances (fp32 device arithmetic vs float64 oracle):
  * one operator / one step:  rel-L2 <= 1e-5
  * trajectories of <= 30 steps: rel-L2 <= 5e-5
"""
import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu

STEP_TOL = 1e-5
TRAJ_TOL = 5e-5


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available(), "gpu tests need a GPU"
    import lmc_atomi_amd as la
    return la


@pytest.fixture(autouse=True, params=["tile", "auto"])
def variant(request, la):
    """Every test runs twice: with the LDS-tiled step kernel forced, and with the default choice
    (the split streaming register-pipeline kernel wherever it covers the configuration)."""
    prev = la.set_step_variant(request.param)
    yield request.param
    la.set_step_variant(prev)


def synth(ny, nx, seed=0, k=5, sigma=0.75):
    rng = np.random.default_rng(seed)
    img = np.zeros((ny, nx))
    for _ in range(5):
        i0, j0 = rng.integers(0, ny - 1), rng.integers(0, nx - 1)
        i1, j1 = rng.integers(i0 + 1, ny + 1), rng.integers(j0 + 1, nx + 1)
        img[i0:i1, j0:j1] = rng.uniform(20, 235)
    img += np.linspace(0, 20, nx)[None, :]
    h = np.ones((k, k)) / (k * k)
    y = O.blur(img, h, (k // 2, k // 2)) + rng.normal(0, sigma, (ny, nx))
    return img, h, y


# ------------------------------------------------------------------ operators
@pytest.mark.parametrize("k,shape", [(5, (16, 16)), (6, (20, 24)), (7, (33, 70)), (3, (5, 7)), (5, (130, 67))])
def test_blur_and_adjoint(la, k, shape):
    rng = np.random.default_rng(k)
    x = rng.normal(size=(3,) + shape)
    h = rng.uniform(size=(k, k))
    off = (k // 2, k // 2)
    Hop = la.Convolve2D(shape, h, offset=off)
    assert rel(Hop.matvec(x), O.blur(x, h, off)) < STEP_TOL
    assert rel(Hop.rmatvec(x), O.blur_adjoint(x, h, off)) < STEP_TOL
    assert rel(Hop.H * x[0].ravel(), O.blur_adjoint(x[0], h, off).ravel()) < STEP_TOL
    # flat vector in -> flat vector out, numpy float64 in -> numpy float64 out (reference calling convention)
    out = Hop * x[0].ravel()
    assert isinstance(out, np.ndarray) and out.shape == (shape[0] * shape[1],) and out.dtype == np.float64


def test_blur_nonsquare_kernel_and_offsets(la):
    rng = np.random.default_rng(0)
    x = rng.normal(size=(21, 19))
    h = rng.uniform(size=(3, 7))
    for off in [(0, 0), (2, 6), (1, 3)]:
        Hop = la.Convolve2D((21, 19), h, offset=off)
        assert rel(Hop.matvec(x), O.blur(x, h, off)) < STEP_TOL
        assert rel(Hop.rmatvec(x), O.blur_adjoint(x, h, off)) < STEP_TOL


@pytest.mark.parametrize("shape", [(16, 16), (7, 130), (65, 33)])
def test_gradient_and_adjoint(la, shape):
    rng = np.random.default_rng(1)
    G, Go = la.Gradient(shape), O.Gradient(shape)
    x = rng.normal(size=shape[0] * shape[1])
    y = rng.normal(size=2 * shape[0] * shape[1])
    assert rel(G.matvec(x), Go.matvec(x)) < STEP_TOL
    assert rel(G.rmatvec(y), Go.rmatvec(y)) < STEP_TOL
    xb = rng.normal(size=(4, shape[0] * shape[1]))
    assert rel(G.matvec(xb)[2], Go.matvec(xb[2])) < STEP_TOL
    # adjoint dot test on the device operators themselves
    assert abs(np.dot(G.matvec(x), y) - np.dot(x, G.rmatvec(y))) < 1e-3 * np.linalg.norm(x) * np.linalg.norm(y) * 1e-2


@pytest.mark.parametrize("k,shape", [(5, (16, 16)), (6, (20, 24)), (7, (24, 18)), (5, (100, 70)), (5, (64, 64))])
def test_l2_grad_and_value(la, k, shape):
    img, h, y = synth(*shape, seed=k, k=k)
    rng = np.random.default_rng(3)
    x = img + rng.normal(0, 5, shape)
    off = (k // 2, k // 2)
    l2 = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y.ravel(), sigma=1 / 0.75 ** 2)
    l2o = O.L2(Op=O.Convolve2D(shape, h, offset=off), b=y.ravel(), sigma=1 / 0.75 ** 2)
    # grad = sigma_f H^T(Hx - y) cancels Hx ~ 200 against y ~ 200: the fp32 error is relative to the
    # operands before the cancellation (it enters the MYULA update multiplied by tau, i.e. at 1e-7 of x)
    got, ref = l2.grad(x.ravel()), l2o.grad(x.ravel())
    scale = (1 / 0.75 ** 2) * (np.linalg.norm(l2o.Op.rmatvec(l2o.Op.matvec(x.ravel()))) + np.linalg.norm(l2o.Op.rmatvec(y.ravel())))
    assert np.linalg.norm(got - ref) < 2e-7 * scale
    assert rel(got, ref) < 1e-4
    assert abs(l2(x.ravel()) - l2o(x.ravel())) < 1e-5 * abs(l2o(x.ravel()))


@pytest.mark.parametrize("niter,shape,gamma", [(10, (16, 16), 0.16875), (10, (100, 70), 0.16875), (1, (8, 8), 2.0),
                                               (20, (64, 64), 2.0), (10, (65, 129), 15.0), (3, (4, 4), 0.5),
                                               (32, (40, 40), 1.0)])
def test_tv_prox_matches_oracle(la, niter, shape, gamma):
    rng = np.random.default_rng(niter)
    img, _, _ = synth(*shape, seed=1)
    x = img + rng.normal(0, 8, shape)
    for momentum in ("unlocbox", "fista"):
        tv = la.TV(shape, sigma=0.3, niter=niter, momentum=momentum)
        ref = O.tv_prox_fgp(x, 0.3 * gamma / 0.3, niter, momentum=momentum)   # prox parameter tau=gamma/0.3, sigma=0.3
        out = tv.prox(x.ravel(), gamma / 0.3)
        assert out.shape == (shape[0] * shape[1],)
        assert rel(out, ref.ravel()) < STEP_TOL, (momentum, rel(out, ref.ravel()))
    tvo = O.TV(shape, sigma=0.3)
    assert abs(tv(x.ravel()) - tvo(x.ravel())) < 1e-5 * tvo(x.ravel())


def test_tv_prox_batch_and_independence(la):
    rng = np.random.default_rng(5)
    x = rng.normal(size=(6, 30, 50)) * 30
    tv = la.TV((30, 50), sigma=1.0, niter=10)
    out = tv.prox(x, 0.7)
    assert out.shape == x.shape
    for c in (0, 5):
        assert rel(out[c], O.tv_prox_fgp(x[c], 0.7, 10)) < STEP_TOL


def test_dual_projections(la):
    rng = np.random.default_rng(6)
    v = rng.normal(size=2 * 300) * 2
    assert rel(la.L21(sigma=0.3).proxdual(v, 1.0), O.L21(sigma=0.3).proxdual(v, 1.0)) < STEP_TOL
    assert rel(la.L21(sigma=0.3).prox(v, 0.7), O.L21(sigma=0.3).prox(v, 0.7)) < STEP_TOL
    assert rel(la.L1(sigma=0.3).proxdual(v, 1.0), O.L1(sigma=0.3).proxdual(v, 1.0)) < STEP_TOL
    assert rel(la.L1(sigma=0.3).prox(v, 0.7), O.L1(sigma=0.3).prox(v, 0.7)) < STEP_TOL


# ------------------------------------------------------------------ one MYULA step, injected noise
def _oracle_prior(kind, tau_reg, niter, t):
    return {"kind": kind, "sigma": tau_reg, "niter": niter, "t": t}


@pytest.mark.parametrize("prior", ["tv", "l1", "l2", "none"])
@pytest.mark.parametrize("data,k,shape", [("blur", 5, (32, 32)), ("blur", 6, (20, 24)), ("blur", 7, (70, 100)),
                                          ("identity", 0, (16, 48)), ("mask", 0, (33, 65))])
def test_myula_steps_injected_noise(la, prior, data, k, shape):
    sigma, tau_reg = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    rng = np.random.default_rng(11)
    C, nit = 3, 4
    img, h, y = synth(*shape, seed=2, k=max(k, 3))
    mask = None
    if data == "blur":
        off = (k // 2, k // 2)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    elif data == "identity":
        h, off = None, None
        y = img + rng.normal(0, sigma, shape)
        pf = la.L2(b=y, sigma=1 / sigma ** 2, dims=shape)
    else:
        h, off = None, None
        mask = (rng.uniform(size=shape) < 0.5).astype(np.float64)
        y = mask * img + rng.normal(0, sigma, shape) * mask
        pf = la.L2(Op=la.Diagonal(mask, dims=shape), b=y, sigma=1 / sigma ** 2, dims=shape)
    pg = {"tv": la.TV(shape, sigma=tau_reg, niter=10), "l1": la.L1(sigma=tau_reg), "l2": la.L2(sigma=0.05),
          "none": None}[prior]
    x0 = img[None] + rng.normal(0, 10, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
    smp.set_state(x0)
    op = _oracle_prior(prior, 0.05 if prior == "l2" else tau_reg, 10, gamma)
    x = x0.copy()
    for it in range(nit):
        smp.step(1, noise=noise[it:it + 1])
        x = O.myula_step(x, y, h, off, 1 / sigma ** 2, tau, gamma, op, noise[it], mask=mask)
        got = smp.get_state().cpu().numpy()
        assert rel(got, x) < STEP_TOL * (it + 1), (it, rel(got, x))
    assert smp.iteration == nit
    smp.close()


def test_multi_step_call_equals_single_steps(la):
    shape = (24, 40)
    img, h, y = synth(*shape, seed=4)
    pf = la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1 / 0.75 ** 2)
    pg = la.TV(shape, sigma=0.3, niter=5)
    a = la.MYULASampler(pf, pg, shape, n_chains=4, tau=0.1, gamma=0.5625, seed=9)
    b = la.MYULASampler(pf, pg, shape, n_chains=4, tau=0.1, gamma=0.5625, seed=9)
    a.step(6)
    for _ in range(6):
        b.step(1)
    np.testing.assert_array_equal(a.get_state().cpu().numpy(), b.get_state().cpu().numpy())


def test_split_and_tile_variants_agree(la):
    """Same inputs through different step kernels (different on-chip schedules of the same arithmetic), each sampler carrying its
    OWN variant (lmc_problem.step_variant, ABI 2): handles with different variants live side by side, the library default untouched."""
    rng = np.random.default_rng(2)
    la.set_step_variant("auto")          # (the autouse fixture of this module restores its own setting afterwards)
    for shape, k, niter in [((64, 64), 5, 10), ((100, 200), 7, 4), ((37, 130), 6, 3), ((512, 512), 5, 10),
                            ((40, 256), 5, 12), ((9, 33), 3, 1), ((40, 96), 5, 9)]:
        img, h, y = synth(*shape, seed=1, k=k)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=(k // 2, k // 2)), b=y, sigma=1 / 0.75 ** 2)
        pg = la.TV(shape, sigma=0.3, niter=niter)
        x0 = img[None] + rng.normal(0, 10, (2,) + shape)
        smps = {v: la.MYULASampler(pf, pg, shape, n_chains=2, tau=0.1125, gamma=0.5625, seed=5, variant=v) for v in ("tile", "split")}
        outs = {}
        for v, smp in smps.items():          # interleaved: each handle keeps its own kernel
            smp.set_state(x0)
            smp.step(1)
        for v, smp in smps.items():
            smp.step(2)
            outs[v] = smp.get_state().cpu().numpy()
            assert v in smp.kernel_name
            smp.close()
        assert rel(outs["split"], outs["tile"]) < 2e-6, (shape, rel(outs["split"], outs["tile"]))
    # other data terms / priors through all kernels that cover them
    shape = (45, 150)
    img, h, y = synth(*shape, seed=3)
    mask = (rng.uniform(size=shape) < 0.5).astype(np.float64)
    cases = [(la.L2(b=y, sigma=1.7, dims=shape), la.L1(sigma=0.3)),
             (la.L2(Op=la.Diagonal(mask, dims=shape), b=mask * y, sigma=1.7, dims=shape), la.TV(shape, sigma=0.3, niter=5)),
             (la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1.7), la.L2(sigma=0.05)),
             (la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1.7), None)]
    for pf, pg in cases:
        outs = {}
        variants = ("tile", "split") if isinstance(pg, la.TV) else ("tile", "split", "point")
        for v in variants:
            smp = la.MYULASampler(pf, pg, shape, n_chains=3, tau=0.1125, gamma=0.5625, seed=8, chain_offset=5, variant=v)
            smp.set_state(img)
            smp.step(4)
            outs[v] = smp.get_state().cpu().numpy()
            smp.close()
        assert rel(outs["split"], outs["tile"]) < 2e-6
        if "point" in outs:
            assert rel(outs["point"], outs["tile"]) < 2e-6
    # a configuration the split kernel does not cover (K = 16): forcing it is an error, the default falls back
    shape = (40, 96)
    img, h, y = synth(*shape, seed=1)
    pf = la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1 / 0.75 ** 2)
    pg = la.TV(shape, sigma=0.3, niter=16)
    smp = la.MYULASampler(pf, pg, shape, n_chains=1, tau=0.1125, gamma=0.5625, variant="split")
    with pytest.raises(la.LMCError, match="no step-kernel variant"):
        smp.step(1)
    smp.close()
    smp = la.MYULASampler(pf, pg, shape, n_chains=1, tau=0.1125, gamma=0.5625)
    smp.step(1)
    assert "tile" in smp.kernel_name
    smp.close()
    assert la.set_step_variant("auto") == "auto"       # per-handle variants never touched the library default
    # the one-group "stream" kernel of ABI 1 is gone: asking for it is an error, not a silent substitution
    with pytest.raises(ValueError):
        la.set_step_variant("stream")
    assert la._dev.lib().lmc_set_step_variant(2) < 0


def test_wide_images_use_tiled_kernels(la, variant):
    """W > 512 and W % 4 != 0 (e.g. the reference's 667 x 877 einstein image, prox_lmc_deconv.py:46): closed-form priors run on the
    row-streaming kernel and TV K = 10 on the pipeline (column strips, dword-aligned accesses; tests/test_gpu_wide.py); the 'point' kernel and
    the LDS-tiled kernel (other dual-iteration counts) stay as the general fallbacks; all against the oracle with injected noise."""
    if variant != "auto":
        pytest.skip("dispatch test")
    rng = np.random.default_rng(9)
    shape = (40, 877)
    img, h, y = synth(*shape, seed=5)
    noise = rng.standard_normal((2, 2) + shape)
    x0 = img[None] + rng.normal(0, 10, (2,) + shape)
    for pg, kern, var, prior in [(la.L1(sigma=0.3), "rows", None, {"kind": "l1", "sigma": 0.3, "t": 0.5625}),
                                 (la.L1(sigma=0.3), "point", "point", {"kind": "l1", "sigma": 0.3, "t": 0.5625}),
                                 (la.TV(shape, sigma=0.3, niter=10), "pipe", None, {"kind": "tv", "sigma": 0.3, "niter": 10, "t": 0.5625}),
                                 (la.TV(shape, sigma=0.3, niter=4), "tile", None, {"kind": "tv", "sigma": 0.3, "niter": 4, "t": 0.5625})]:
        pf = la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1 / 0.5625)
        smp = la.MYULASampler(pf, pg, shape, n_chains=2, tau=0.1125, gamma=0.5625, noise="injected", variant=var)
        smp.set_state(x0)
        smp.step(2, noise=noise)
        assert kern in smp.kernel_name, smp.kernel_name
        x = x0.copy()
        for it in range(2):
            x = O.myula_step(x, y, h, (2, 2), 1 / 0.5625, 0.1125, 0.5625, prior, noise[it])
        assert rel(smp.get_state().cpu().numpy(), x) < 2e-5
        smp.close()


# ------------------------------------------------------------------ RNG rung (R3)
def test_philox_noise_field_matches_oracle(la):
    shape = (30, 64)
    pf = la.L2(b=np.zeros(shape), sigma=1.0, dims=shape)
    smp = la.MYULASampler(pf, None, shape, n_chains=5, tau=0.1, gamma=0.5, seed=0x1234567890ABCDEF, chain_offset=7)
    for it in (0, 3, 1000):
        got = smp.noise_field(it).cpu().numpy()
        ref = O.philox_normals(0x1234567890ABCDEF, it, np.arange(7, 12), *shape)
        assert np.abs(got - ref).max() < 2e-5, np.abs(got - ref).max()   # hardware log2/sqrt/sin/cos
    assert abs(got.mean()) < 0.03 and abs(got.std() - 1) < 0.03


def test_philox_trajectory_matches_oracle_with_same_counters(la):
    """HIP Philox path == oracle driven by the oracle's own Philox field: the kernels consume
    the counters (seed, iteration, global chain, pixel) exactly as specified."""
    shape = (20, 36)
    sigma = 0.75
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    img, h, y = synth(*shape, seed=8)
    pf = la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1 / sigma ** 2)
    pg = la.TV(shape, sigma=0.3, niter=10)
    C, off, seed, nit = 4, 100, 42, 12
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, seed=seed, chain_offset=off)
    smp.step(nit)
    got = smp.get_state().cpu().numpy()
    prior = _oracle_prior("tv", 0.3, 10, gamma)
    x = O.myula_batched(np.zeros((C,) + shape), y, h, (2, 2), 1 / sigma ** 2, tau, gamma, prior, nit,
                        lambda k: O.philox_normals(seed, k, np.arange(off, off + C), *shape).astype(np.float64))
    assert rel(got, x) < TRAJ_TOL, rel(got, x)
    # sharding invariance: chains 2..3 run alone (as another GPU would) give the same states bit for bit
    part = la.MYULASampler(pf, pg, shape, n_chains=2, tau=tau, gamma=gamma, seed=seed, chain_offset=off + 2)
    part.step(nit)
    np.testing.assert_array_equal(part.get_state().cpu().numpy(), got[2:])


# ------------------------------------------------------------------ golden vectors of the reference loops
@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_drop_in_myula_reproduces_reference_trajectories(la, golden, tag):
    """la.MoreauYosidaUnadjustedLangevin(..., rng='pcg64') against trajectories produced by the
    reference's own MoreauYosidaUnadjustedLangevin (tests/golden/algs.npz)."""
    g = golden("algs.npz")
    sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0 = [float(v) for v in g["params"]]
    ny, nx, k, seed = [int(v) for v in g[f"{tag}_meta"]]
    h, y = g[f"{tag}_h"], g[f"{tag}_y"]
    H = la.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    priors = {"tv": la.TV((ny, nx), sigma=tau_reg, niter=10), "l1": la.L1(sigma=tau_reg), "l2": la.L2(sigma=0.05)}
    for pname, pg in priors.items():
        key = f"{tag}_myula_{pname}"
        if key not in g.files:
            continue
        ref = g[key]
        l2 = la.L2(Op=H, b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
        seen = []
        out = la.MoreauYosidaUnadjustedLangevin(l2, pg, tau=tau_myula, gamma=gamma_myula, x0=np.zeros(ny * nx),
                                                niter=ref.shape[0], seed=seed, rng="pcg64",
                                                callback=lambda x: seen.append(x[0]))
        assert out.shape == ref.shape and out.dtype == np.float64 and len(seen) == ref.shape[0]
        assert rel(out, ref) < TRAJ_TOL, (key, rel(out, ref))
        assert rel(out[-1], ref[-1]) < TRAJ_TOL


def test_elementwise_prox_library_matches_reference(la, golden):
    from lmc_atomi_amd import prox as P
    g = golden("prox.npz")
    x = g["x"]
    tol = dict(rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(P.prox_laplace(x, 0.5), g["laplace_0.5"], **tol)
    np.testing.assert_allclose(P.prox_uncentered_laplace(x, 0.7, 1.5), g["uncentered_laplace_0.7_1.5"], **tol)
    np.testing.assert_allclose(P.prox_gaussian(x, 0.3), g["gaussian_0.3"], **tol)
    np.testing.assert_allclose(P.prox_conjugate(x, 0.8, P.prox_laplace), g["conjugate_laplace_0.8"], **tol)
    for name, p in (("4_3", 4 / 3), ("3_2", 3 / 2), ("3", 3), ("4", 4)):
        np.testing.assert_allclose(P.prox_gen_gaussian(x, 0.6, p), g["gen_gaussian_0.6_" + name], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(P.prox_huber(x, 0.5, 0.4), g["huber_0.5_0.4"], **tol)
    np.testing.assert_allclose(P.prox_smoothed_laplace(x, 0.9), g["smoothed_laplace_0.9"], **tol)
    np.testing.assert_allclose(P.prox_exp(x, 0.5), g["exp_0.5"], **tol)
    np.testing.assert_allclose(P.prox_gamma(x, 0.4, 1.3), g["gamma_0.4_1.3"], **tol)
    np.testing.assert_allclose(P.prox_chi(x, 0.7), g["chi_0.7"], **tol)
    np.testing.assert_allclose(P.prox_uniform(x, 1.2), g["uniform_1.2"], **tol)
    np.testing.assert_allclose(P.prox_triangular(x, -0.5, 0.8), g["triangular_-0.5_0.8"], **tol)


# ------------------------------------------------------------------ moments, energies
def test_moments_and_energies(la):
    shape = (24, 32)
    sigma = 0.75
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    img, h, y = synth(*shape, seed=12)
    pf = la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1 / sigma ** 2)
    pg = la.TV(shape, sigma=0.3, niter=10)
    C, nit, burn, thin = 20, 9, 2, 3
    rng = np.random.default_rng(0)
    noise = rng.standard_normal((nit, C) + shape)
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected", moments=True,
                          burn_in=burn, thin=thin)
    smp.step(nit, noise=noise)
    s1, s2, cnt = smp.moments()
    prior = _oracle_prior("tv", 0.3, 10, gamma)
    x, r1, r2, rc = O.myula_batched(np.zeros((C,) + shape), y, h, (2, 2), 1 / sigma ** 2, tau, gamma, prior, nit,
                                    lambda k: noise[k], moments=True, burn_in=burn, thin=thin)
    assert cnt == rc == C * 3          # iterations 2, 5, 8
    assert rel(s1.cpu().numpy(), r1) < TRAJ_TOL and rel(s2.cpu().numpy(), r2) < TRAJ_TOL
    f, g = smp.energies()
    l2o, tvo = O.L2(Op=O.Convolve2D(shape, h), b=y.ravel(), sigma=1 / sigma ** 2), O.TV(shape, sigma=0.3)
    for c in (0, C - 1):
        assert abs(float(f[c]) - l2o(x[c].ravel())) < 1e-4 * l2o(x[c].ravel())
        assert abs(float(g[c]) - tvo(x[c].ravel())) < 1e-4 * tvo(x[c].ravel())
    smp.reset_moments()
    assert smp.moments()[2] == 0 and float(smp.moments()[0].abs().sum()) == 0.0


def test_many_chain_entry_point_statistics(la):
    """R4 (statistical rung, small): posterior mean of many Philox chains on the GPU vs many
    PCG64 chains of the oracle; tolerance set by Monte-Carlo error."""
    shape = (16, 16)
    sigma = 0.75
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    img, h, y = synth(*shape, seed=20)
    pf = la.L2(Op=la.Convolve2D(shape, h), b=y, sigma=1 / sigma ** 2)
    pg = la.TV(shape, sigma=0.3, niter=10)
    res = la.MoreauYosidaUnadjustedLangevin(pf, pg, np.zeros(shape), tau=tau, gamma=gamma, niter=300, seed=1,
                                            n_chains=256, burn_in=150, thin=10)
    assert res.count == 256 * 15
    prior = _oracle_prior("tv", 0.3, 10, gamma)
    rng = np.random.default_rng(5)
    Co = 64
    _, r1, r2, rc = O.myula_batched(np.zeros((Co,) + shape), y, h, (2, 2), 1 / sigma ** 2, tau, gamma, prior, 300,
                                    lambda k: rng.standard_normal((Co,) + shape), moments=True, burn_in=150, thin=10)
    mo, vo = r1 / rc, r2 / rc - (r1 / rc) ** 2
    mean, var = res.mean.cpu().numpy(), res.var.cpu().numpy()
    assert rel(mean, mo) < 5e-3, rel(mean, mo)          # MC error of 64x15 correlated samples dominates
    assert rel(var, vo) < 0.15, rel(var, vo)
    assert np.all(var > 0)


# ------------------------------------------------------------------ error behaviour
def test_errors_are_loud(la):
    import torch
    with pytest.raises(la.LMCError):
        la.Convolve2D((8, 8), np.ones((5, 5)), offset=(7, 0)).matvec(np.zeros(64))
    with pytest.raises(ValueError):
        la.Convolve2D((8, 8), np.ones((11, 11)))
    with pytest.raises(la.LMCError):
        la.TV((8, 8), sigma=0.3, niter=100).prox(np.zeros(64), 1.0)
    with pytest.raises(NotImplementedError):
        class Foreign:
            def prox(self, x, t):
                return x
        la.MYULASampler(la.L2(b=np.zeros((8, 8)), sigma=1.0, dims=(8, 8)), Foreign(), (8, 8), tau=0.1, gamma=0.5)
    smp = la.MYULASampler(la.L2(b=np.zeros((8, 8)), sigma=1.0, dims=(8, 8)), None, (8, 8), tau=0.1, gamma=0.5,
                          noise="injected")
    with pytest.raises(la.LMCError):
        smp.step(1)                       # injected mode without noise
    with pytest.raises(la.LMCError):
        smp.moments()                     # moments were not enabled
