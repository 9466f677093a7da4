"""GPU parity of the Haar-l1 wavelet prior (BASELINE config "inpainting mask + l1-wavelet prox"): operator, energies and
MYULA steps against the oracle (itself pinned by PyWavelets, tests/golden/haar_pywt.npz), through the C ABI."""
import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu

TOL = 2e-6


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


def test_haar_prox_matches_pywavelets_fixture(la, golden):
    g = golden("haar_pywt.npz")
    x = g["x"]
    for thr in (0.1, 2.0):
        got = la.WaveletL1(x.shape, sigma=1.0).prox(x.ravel(), thr).reshape(x.shape)
        assert rel(got, g["thr_%g" % thr]) < TOL
    val = la.WaveletL1(x.shape, sigma=0.7)(x.ravel())
    assert abs(val - 0.7 * float(g["val"])) < 1e-5 * abs(val)


@pytest.mark.parametrize("shape", [(8, 8), (16, 64), (40, 24), (128, 520)])
def test_haar_prox_and_value_match_oracle(la, shape):
    rng = np.random.default_rng(3)
    x = rng.normal(0, 30, (3,) + shape)
    w = la.WaveletL1(shape, sigma=0.4)
    got = w.prox(x.reshape(3, -1), 5.0).reshape(x.shape)
    ref = O.haar_l1_prox(x, 0.4 * 5.0)
    assert rel(got, ref) < TOL
    assert rel(w.prox(x.reshape(3, -1), 0.0).reshape(x.shape), x) < TOL       # W^T W = I
    vals = np.asarray(w(x.reshape(3, -1)))
    ref_v = np.array([0.4 * O.haar_l1_value(x[i]) for i in range(3)])
    assert np.allclose(vals, ref_v, rtol=1e-5)


def test_haar_rejects_ragged_sizes(la):
    with pytest.raises(ValueError):
        la.WaveletL1((12, 16))
    import torch
    x = torch.zeros(12 * 16, device="cuda")
    rc = la._dev.lib().lmc_haar_l1_prox(la._dev.ptr(x), la._dev.ptr(x), 1, 12, 16, 1.0, None)
    assert rc == -2  # LMC_E_UNSUPPORTED


@pytest.mark.parametrize("data,shape", [("mask", (64, 64)), ("mask", (24, 520)), ("blur", (48, 128)), ("identity", (16, 16))])
def test_myula_steps_haar_prior(la, data, shape):
    """The inpainting configuration: f = sigma/2 ||M x - y||^2, g = lam ||W_detail x||_1; every step kernel."""
    sigma, lam = 0.75, 2.0
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    rng = np.random.default_rng(5)
    C, nit = 3, 3
    img = np.zeros(shape)
    img[shape[0] // 4:shape[0] // 2, shape[1] // 5:shape[1] // 2] = 180.0
    img += np.linspace(0, 40, shape[1])[None, :]
    mask, h, off = None, None, None
    if data == "mask":
        mask = (rng.uniform(size=shape) < 0.6).astype(np.float64)
        y = mask * (img + rng.normal(0, sigma, shape))
        pf = la.L2(Op=la.Diagonal(mask, dims=shape), b=y, sigma=1 / sigma ** 2, dims=shape)
    elif data == "blur":
        h, off = np.ones((5, 5)) / 25, (2, 2)
        y = O.blur(img, h, off) + rng.normal(0, sigma, shape)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    else:
        y = img + rng.normal(0, sigma, shape)
        pf = la.L2(b=y, sigma=1 / sigma ** 2, dims=shape)
    pg = la.WaveletL1(shape, sigma=lam)
    x0 = img[None] + rng.normal(0, 10, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    prior = {"kind": "haar", "sigma": lam, "t": gamma}
    outs = {}
    for variant in ("tile", "point", "auto") + (("block",) if data != "blur" else ()):
        la.set_step_variant(variant)
        smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
        smp.set_state(x0)
        x = x0.copy()
        for it in range(nit):
            smp.step(1, noise=noise[it:it + 1])
            x = O.myula_step(x, y, h, off, 1 / sigma ** 2, tau, gamma, prior, noise[it], mask=mask)
            got = smp.get_state().cpu().numpy()
            assert rel(got, x) < TOL * (it + 1), (variant, it, rel(got, x))
        f, g = smp.energies()
        gref = np.array([lam * O.haar_l1_value(x[i]) for i in range(C)])
        assert np.allclose(g.cpu().numpy(), gref, rtol=2e-5), variant
        assert smp.kernel_name == {"tile": "myula_step_tile_kernel", "point": "myula_step_point_kernel", "block": "myula_step_block_kernel",
                                   "auto": "myula_step_rows_kernel" if data == "blur" else "myula_step_block_kernel"}[variant]
        outs[variant] = got
        smp.close()
    la.set_step_variant("auto")
    assert rel(outs["tile"], outs["auto"]) < TOL


def test_haar_prior_many_iterations_philox_and_moments(la):
    """Philox noise, several iterations per call, moments: the two-launch (prox, step) sequence keeps the iteration
    counter, the ping-pong buffers and the accumulators consistent."""
    shape = (32, 64)
    rng = np.random.default_rng(8)
    mask = (rng.uniform(size=shape) < 0.5).astype(np.float64)
    y = mask * rng.uniform(0, 255, shape)
    pf = la.L2(Op=la.Diagonal(mask, dims=shape), b=y, sigma=1.5, dims=shape)
    pg = la.WaveletL1(shape, sigma=1.0)
    kw = dict(n_chains=4, tau=0.1, gamma=0.5, seed=21, moments=True)
    a = la.MYULASampler(pf, pg, shape, **kw)
    b = la.MYULASampler(pf, pg, shape, **kw)
    a.step(7)
    xs = []
    for _ in range(7):
        b.step(1)
        xs.append(b.get_state().cpu().numpy().astype(np.float64))
    np.testing.assert_array_equal(a.get_state().cpu().numpy(), b.get_state().cpu().numpy())
    s1, s2, n = a.moments()
    assert n == 7 * 4
    ref = np.sum(np.concatenate(xs, axis=0), axis=0)
    assert rel(s1.cpu().numpy(), ref) < 1e-6
    # one step against the oracle with the sampler's own noise field
    c = la.MYULASampler(pf, pg, shape, **kw)
    x0 = c.get_state().cpu().numpy().astype(np.float64)
    xi = c.noise_field(0).cpu().numpy().astype(np.float64)
    c.step(1)
    ref1 = O.myula_step(x0, y, None, None, 1.5, 0.1, 0.5, {"kind": "haar", "sigma": 1.0, "t": 0.5}, xi, mask=mask)
    assert rel(c.get_state().cpu().numpy(), ref1) < 5e-6
    for s in (a, b, c):
        s.close()


def test_config5_mc_term_at_512_columns_column_halo_from_the_neighbouring_lanes(la):
    """At W = 512 a wave is exactly one row of 8 x 8 blocks and the block kernel takes the column halo of the MC-TV window from the neighbouring lanes
    (wave shifts) instead of memory: the same step against the checker's class gradient, image edges and block seams included."""
    shape = (24, 512)
    sigma, lam = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    mask = (np.random.default_rng(7).uniform(size=shape) < 0.5).astype(np.float64)
    rng = np.random.default_rng(3)
    img = np.zeros(shape); img[4:18, 100:400] = 200.0
    img += np.linspace(0, 40, shape[1])[None, :]
    y = mask * (img + rng.normal(0, sigma, shape))
    C, nit = 3, 3
    x0 = img[None] + rng.normal(0, 12, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    kw = dict(dims=shape, b=y.ravel(), sigma=1 / sigma ** 2, lamda=0.3, gamma=15.0, isotropic=True, niter=1)
    pf = la.L2_ncvx_tv(Op=la.Diagonal(mask, dims=shape), Op2=la.Gradient(shape), **kw)
    of = O.L2NcvxTV(Op=O.Diagonal(mask), Op2=O.Gradient(shape), **kw)
    pg = la.WaveletL1(shape, sigma=lam)
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
    smp.set_state(x0)
    x = x0.copy()
    for it in range(nit):
        smp.step(1, noise=noise[it:it + 1])
        assert "block" in smp.kernel_name, smp.kernel_name
        g = np.stack([of.grad(x[c].ravel().copy()).reshape(shape) for c in range(C)])
        x = (1 - tau / gamma) * x - tau * g + (tau / gamma) * O.haar_l1_prox(x, gamma * lam) + np.sqrt(2 * tau) * noise[it]
        got = smp.get_state().cpu().numpy()
        err = np.abs(got - x).max(axis=(0, 1))          # per column: a wrong halo shows at the block seams
        assert rel(got, x) < 5e-6 * (it + 1) and err.max() < 2e-3, (it, rel(got, x), int(err.argmax()), err.max())
    smp.close()


@pytest.mark.parametrize("kind,niter_in", [("mc", 0), ("me", 8), ("me", 30)])
def test_config5_mask_haar_with_nonconvex_term(la, kind, niter_in):
    """Bernoulli(0.5) mask from default_rng(7), Haar-l1 prox (threshold 0.3*gamma) and the L2_ncvx_tv-style
    Moreau-difference term (lamda = 0.3, gamma = 15; algs.py:270-291) -- one MYULA step per iteration against
    x' = (1 - tau/gamma) x - tau grad f(x) + tau/gamma prox(x) + sqrt(2 tau) xi with the oracle's class gradient."""
    shape = (64, 128)
    sigma, lam = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    mask = (np.random.default_rng(7).uniform(size=shape) < 0.5).astype(np.float64)
    rng = np.random.default_rng(1)
    img = np.zeros(shape); img[10:40, 30:90] = 200.0
    y = mask * (img + rng.normal(0, sigma, shape))
    C, nit = 2, 3
    x0 = img[None] + rng.normal(0, 12, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    kw = dict(dims=shape, b=y.ravel(), sigma=1 / sigma ** 2, lamda=0.3, gamma=15.0, isotropic=True, niter=max(niter_in, 1))
    if kind == "mc":
        pf = la.L2_ncvx_tv(Op=la.Diagonal(mask, dims=shape), Op2=la.Gradient(shape), **kw)
        of = O.L2NcvxTV(Op=O.Diagonal(mask), Op2=O.Gradient(shape), **kw)
    else:
        pf = la.L2_ncvx_tv(Op=la.Diagonal(mask, dims=shape), rtol=0.0, **kw)      # fixed count: the checker's default (its rtol branch: test_gpu_rtol.py)
        of = O.L2NcvxTV(Op=O.Diagonal(mask), **kw)
    pg = la.WaveletL1(shape, sigma=lam)
    for variant in ("tile", "auto"):
        la.set_step_variant(variant)
        smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
        smp.set_state(x0)
        x = x0.copy()
        for it in range(nit):
            smp.step(1, noise=noise[it:it + 1])
            g = np.stack([of.grad(x[c].ravel().copy()).reshape(shape) for c in range(C)])
            x = (1 - tau / gamma) * x - tau * g + (tau / gamma) * O.haar_l1_prox(x, gamma * lam) + np.sqrt(2 * tau) * noise[it]
            got = smp.get_state().cpu().numpy()
            assert rel(got, x) < 5e-6 * (it + 1), (variant, it, rel(got, x))
        f, gval = smp.energies()
        fref = np.array([of(x[c].ravel()) for c in range(C)])
        assert np.allclose(f.cpu().numpy(), fref, rtol=5e-5), (variant, f.cpu().numpy(), fref)
        smp.close()
    la.set_step_variant("auto")
