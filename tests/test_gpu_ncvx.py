"""GPU parity of the non-log-concave data term algs.L2_ncvx_tv (MC-TV branch), against values produced by the
reference's own class (tests/golden/algs.npz) and against the oracle, through the C ABI."""
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
    return la


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_mc_tv_value_grad_and_myula_match_reference_class(la, golden, tag):
    g = golden("algs.npz")
    sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0 = [float(v) for v in g["params"]]
    ny, nx, k, seed = [int(v) for v in g[f"{tag}_meta"]]
    h, y = g[f"{tag}_h"], g[f"{tag}_y"]
    H = la.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    G = la.Gradient((ny, nx))
    xt = g[f"{tag}_ncvx_x"]
    for variant in ("tile", "auto"):
        la.set_step_variant(variant)
        mc = la.L2_ncvx_tv(dims=(ny, nx), Op=H, Op2=G, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=15.0,
                           isotropic=True, niter=50, warm=True)
        got = mc.grad(xt.copy())
        ref = g[f"{tag}_ncvx_mc_grad"]
        # the data part cancels Hx ~ 200 against y ~ 200 (see test_l2_grad_and_value): scale-aware bound + loose rel
        l2o = O.L2(Op=O.Convolve2D((ny, nx), h, offset=(k // 2, k // 2)), b=y.ravel(), sigma=1 / sigma ** 2)
        scale = (1 / sigma ** 2) * (np.linalg.norm(l2o.Op.rmatvec(l2o.Op.matvec(xt))) + np.linalg.norm(l2o.Op.rmatvec(y.ravel())))
        assert np.linalg.norm(got - ref) < 2e-7 * scale and rel(got, ref) < 1e-4, (variant, rel(got, ref))
        val = float(g[f"{tag}_ncvx_mc_val"])
        assert abs(mc(xt.copy()) - val) < 1e-5 * abs(val)
        gx = g[f"{tag}_myula_mc_tv"]
        out = la.MoreauYosidaUnadjustedLangevin(mc, la.TV((ny, nx), sigma=tau_reg, niter=10), np.zeros(ny * nx),
                                                tau=tau_myula, gamma=gamma_myula, niter=gx.shape[0], seed=seed, rng="pcg64")
        assert rel(out, gx) < 5e-5, (variant, rel(out, gx))
    la.set_step_variant("auto")


def test_mc_tv_large_image_all_kernels_agree(la):
    """The MC-TV term through the tile and the split kernels on a 512-wide image (row / column boundaries, both
    |grad x| regimes: below and above gamma)."""
    rng = np.random.default_rng(0)
    shape = (96, 512)
    img = np.zeros(shape); img[20:60, 100:400] = 200.0
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    x0 = img[None] + rng.normal(0, 20.0, (2,) + shape)      # differences well above and below gamma = 15
    outs = {}
    for v in ("tile", "split", "pipe"):
        la.set_step_variant(v)
        mc = la.L2_ncvx_tv(dims=shape, Op=la.Convolve2D(shape, h), Op2=la.Gradient(shape), b=y.ravel(), sigma=1 / 0.5625,
                           lamda=0.3, gamma=15.0, isotropic=True)
        smp = la.MYULASampler(mc, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=2, tau=0.1125, gamma=0.5625, seed=4)
        smp.set_state(x0)
        smp.step(3)
        outs[v] = smp.get_state().cpu().numpy()
        assert v in smp.kernel_name
        smp.close()
    la.set_step_variant("auto")
    assert rel(outs["split"], outs["tile"]) < 2e-6
    assert rel(outs["pipe"], outs["tile"]) < 2e-6
    # 4 pixels per lane (W <= 256), image rows not a multiple of anything
    shape2 = (37, 200)
    img2 = np.zeros(shape2); img2[8:25, 40:150] = 200.0
    y2 = O.blur(img2, h, (2, 2)) + rng.normal(0, 0.75, shape2)
    x02 = img2[None] + rng.normal(0, 20.0, (3,) + shape2)
    o2 = {}
    for v in ("tile", "pipe"):
        la.set_step_variant(v)
        mc = la.L2_ncvx_tv(dims=shape2, Op=la.Convolve2D(shape2, h), Op2=la.Gradient(shape2), b=y2.ravel(), sigma=1 / 0.5625,
                           lamda=0.3, gamma=15.0, isotropic=True)
        smp = la.MYULASampler(mc, la.TV(shape2, sigma=0.3, niter=10), shape2, n_chains=3, tau=0.1125, gamma=0.5625, seed=4)
        smp.set_state(x02)
        smp.step(3)
        o2[v] = smp.get_state().cpu().numpy()
        smp.close()
    la.set_step_variant("auto")
    assert rel(o2["pipe"], o2["tile"]) < 2e-6
    mco = O.L2NcvxTV(shape, Op=O.Convolve2D(shape, h), Op2=O.Gradient(shape), b=y.ravel(), sigma=1 / 0.5625, lamda=0.3, gamma=15.0)
    mc = la.L2_ncvx_tv(dims=shape, Op=la.Convolve2D(shape, h), Op2=la.Gradient(shape), b=y.ravel(), sigma=1 / 0.5625,
                       lamda=0.3, gamma=15.0, isotropic=True)
    gm = mc.grad(x0[0].ravel()) - la.L2(Op=la.Convolve2D(shape, h), b=y.ravel(), sigma=1 / 0.5625).grad(x0[0].ravel())
    assert rel(gm, -0.3 * mco.grad_moreau(x0[0].ravel())) < 1e-4


@pytest.mark.parametrize("tag", ["a", "b"])
def test_anisotropic_me_tv_matches_reference_class(la, golden, tag):
    """The anisotropic ME-TV branch (isotropic=False without Op2: a Moreau envelope of the 1-D TV of the FLATTENED image, algs.py:170) against
    outputs of the reference's own class (tests/golden/algs_aniso_me.npz): value, gradient, prox and the MYULA trajectory it drives, with the
    inner prox's exit as the reference configures it (rtol = 1e-4) and switched off (rtol = 0)."""
    g = golden("algs_aniso_me.npz")
    sigma, tau_reg, tau_myula, gamma_myula, tau0 = [float(v) for v in g["params"]]
    ny, nx, k, seed, gam, niter = [float(v) for v in g[f"{tag}_meta"]]
    ny, nx, k, seed, niter = int(ny), int(nx), int(k), int(seed), int(niter)
    h, y = g[f"{tag}_h"], g[f"{tag}_y"]
    H = la.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    xt = g[f"{tag}_x"]
    l2o = O.L2(Op=O.Convolve2D((ny, nx), h, offset=(k // 2, k // 2)), b=y.ravel(), sigma=1 / sigma ** 2)
    scale = (1 / sigma ** 2) * (np.linalg.norm(l2o.Op.rmatvec(l2o.Op.matvec(xt))) + np.linalg.norm(l2o.Op.rmatvec(y.ravel())))
    for rt, rtol in (("rtol1e-4", 1e-4), ("rtol0", 0.0)):
        mk = lambda: la.L2_ncvx_tv(dims=(ny, nx), Op=H, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=gam, isotropic=False, niter=niter, rtol=rtol)
        me = mk()
        got, ref = me.grad(xt.copy()), g[f"{tag}_grad_{rt}"]
        assert np.linalg.norm(got - ref) < 2e-7 * scale and rel(got, ref) < 1e-4, (rt, rel(got, ref))
        val = float(g[f"{tag}_val_{rt}"])
        assert abs(me(xt.copy()) - val) < 1e-5 * abs(val)
        vp = g[f"{tag}_prox_in"]
        assert rel(mk().prox(vp.copy(), tau0), g[f"{tag}_prox_out1_{rt}"]) < 1e-4
        gx = g[f"{tag}_myula_{rt}"]
        out = la.MoreauYosidaUnadjustedLangevin(mk(), la.TV((ny, nx), sigma=tau_reg, niter=10), np.zeros(ny * nx), tau=tau_myula, gamma=gamma_myula,
                                                niter=gx.shape[0], seed=seed, rng="pcg64")
        assert rel(out, gx) < 5e-5, (rt, rel(out, gx))


def test_anisotropic_me_tv_many_chains_against_the_checker(la):
    """... and over several chains at once with injected noise (each chain exits its 1-D prox in its own pass): against the CPU checker's class."""
    rng = np.random.default_rng(12)
    shape = (24, 72)
    img = np.zeros(shape); img[6:18, 10:50] = 150.0
    img += np.linspace(0, 20, shape[1])[None, :]
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    C_, nit = 3, 3
    x0 = img[None] + rng.normal(0, [[[2.0]], [[8.0]], [[25.0]]], (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    sig, gam_m, tau = 0.75, 0.5625, 0.1125
    for rtol in (1e-4, 0.0):
        pf = la.L2_ncvx_tv(dims=shape, Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y.ravel(), sigma=1 / sig ** 2, lamda=0.3, gamma=15.0, isotropic=False,
                           niter=40, rtol=rtol)
        of = O.L2NcvxTV(shape, Op=O.Convolve2D(shape, h, (2, 2)), b=y.ravel(), sigma=1 / sig ** 2, lamda=0.3, gamma=15.0, isotropic=False, niter=40,
                        tv_kwargs={"rtol": rtol})
        otv = O.TV(shape, sigma=0.3, niter=10)
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=C_, tau=tau, gamma=gam_m, noise="injected")
        smp.set_state(x0)
        smp.step(nit, noise=noise)
        got = smp.get_state().cpu().numpy()
        ref = np.stack([O.myula(of, otv, x0[c].ravel(), tau, gam_m, niter=nit, noise=[noise[i, c].ravel() for i in range(nit)])[-1].reshape(shape)
                        for c in range(C_)])
        assert rel(got, ref) < 2e-5, (rtol, rel(got, ref))
        f, _ = smp.energies()
        fref = np.array([of(got[c].ravel().astype(np.float64)) for c in range(C_)])
        assert np.allclose(f.cpu().numpy(), fref, rtol=5e-5)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_anisotropic_mc_tv_matches_reference_class(la, golden, tag):
    """The anisotropic MC-TV branches (isotropic=False, Op2 = Gradient; algs.py:173-190, 218-219, 278-279) against outputs of the reference's
    own class (tests/golden/algs_aniso.npz): value, gradient, prox (cold and warm-started), and the MYULA / ULPDA trajectories it drives --
    through every kernel that evaluates the term (tiled, split / pipe by auto)."""
    g = golden("algs_aniso.npz")
    sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0 = [float(v) for v in g["params"]]
    ny, nx, k, seed, gam = [float(v) for v in g[f"{tag}_meta"]]
    ny, nx, k, seed = int(ny), int(nx), int(k), int(seed)
    h, y = g[f"{tag}_h"], g[f"{tag}_y"]
    H = la.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    G = la.Gradient((ny, nx))
    mk = lambda: la.L2_ncvx_tv(dims=(ny, nx), Op=H, Op2=G, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=gam, isotropic=False,
                               niter=50, warm=True)
    xt = g[f"{tag}_x"]
    l2o = O.L2(Op=O.Convolve2D((ny, nx), h, offset=(k // 2, k // 2)), b=y.ravel(), sigma=1 / sigma ** 2)
    scale = (1 / sigma ** 2) * (np.linalg.norm(l2o.Op.rmatvec(l2o.Op.matvec(xt))) + np.linalg.norm(l2o.Op.rmatvec(y.ravel())))
    for variant in ("tile", "auto"):
        la.set_step_variant(variant)
        mc = mk()
        got, ref = mc.grad(xt.copy()), g[f"{tag}_grad"]
        assert np.linalg.norm(got - ref) < 2e-7 * scale and rel(got, ref) < 1e-4, (variant, rel(got, ref))
        val = float(g[f"{tag}_val"])
        assert abs(mc(xt.copy()) - val) < 1e-5 * abs(val)
        gx = g[f"{tag}_myula"]
        out = la.MoreauYosidaUnadjustedLangevin(mc, la.TV((ny, nx), sigma=tau_reg, niter=10), np.zeros(ny * nx), tau=tau_myula, gamma=gamma_myula,
                                                niter=gx.shape[0], seed=seed, rng="pcg64")
        assert rel(out, gx) < 5e-5, (variant, rel(out, gx))
    la.set_step_variant("auto")
    m = mk()
    vp = g[f"{tag}_prox_in"]
    assert rel(m.prox(vp.copy(), tau0), g[f"{tag}_prox_out1"]) < 1e-4
    assert rel(m.prox(vp + 1.0, tau0), g[f"{tag}_prox_out2"]) < 1e-4
    gx = g[f"{tag}_ulpda"]
    xs = la.UnadjustedLangevinPrimalDual(mk(), la.L21(ndim=2, sigma=tau_reg), G, tau=tau0, mu=mu0, theta=1.0, x0=np.zeros(ny * nx), gfirst=False,
                                         niter=gx.shape[0], seed=seed, rng="pcg64")
    assert rel(xs, gx) < 1e-4, rel(xs, gx)


def test_anisotropic_mc_tv_on_the_pipe_and_block_kernels(la):
    """... and where the term is evaluated inside the full-width pipeline (combine wave) and added after the block kernel (mask + Haar):
    against the CPU checker's class gradient with injected noise."""
    rng = np.random.default_rng(8)
    shape = (40, 264)
    img = np.zeros(shape); img[8:30, 40:200] = 180.0
    img += np.linspace(0, 30, shape[1])[None, :]
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    C_, nit = 2, 3
    x0 = img[None] + rng.normal(0, 8, (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    sig, gam_m, tau = 0.75, 0.5625, 0.1125
    for gam in (15.0, 3.0):
        pf = la.L2_ncvx_tv(dims=shape, Op=la.Convolve2D(shape, h, offset=(2, 2)), Op2=la.Gradient(shape), b=y.ravel(), sigma=1 / sig ** 2, lamda=0.3,
                           gamma=gam, isotropic=False)
        of = O.L2NcvxTV(shape, Op=O.Convolve2D(shape, h, (2, 2)), Op2=O.Gradient(shape), b=y.ravel(), sigma=1 / sig ** 2, lamda=0.3, gamma=gam,
                        isotropic=False)
        otv = O.TV(shape, sigma=0.3, niter=10)
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=C_, tau=tau, gamma=gam_m, noise="injected")
        smp.set_state(x0)
        smp.step(nit, noise=noise)
        assert "pipe" in smp.kernel_name, smp.kernel_name
        got = smp.get_state().cpu().numpy()
        ref = np.stack([O.myula(of, otv, x0[c].ravel(), tau, gam_m, niter=nit, noise=[noise[i, c].ravel() for i in range(nit)])[-1].reshape(shape)
                        for c in range(C_)])
        assert rel(got, ref) < 2e-5, (gam, rel(got, ref))
        f, _ = smp.energies()
        fref = np.array([of(got[c].ravel().astype(np.float64)) for c in range(C_)])
        assert np.allclose(f.cpu().numpy(), fref, rtol=5e-5)
        smp.close()
    # mask + Haar-l1 + anisotropic MC-TV (the block kernel, then the stencil pass that adds the term)
    shape = (64, 128)
    mask = (np.random.default_rng(7).uniform(size=shape) < 0.5).astype(np.float64)
    img = np.zeros(shape); img[10:40, 30:90] = 200.0
    yb = mask * (img + rng.normal(0, sig, shape))
    pf = la.L2_ncvx_tv(dims=shape, Op=la.Diagonal(mask, dims=shape), Op2=la.Gradient(shape), b=yb.ravel(), sigma=1 / sig ** 2, lamda=0.3, gamma=15.0,
                       isotropic=False)
    of = O.L2NcvxTV(shape, Op=O.Diagonal(mask), Op2=O.Gradient(shape), b=yb.ravel(), sigma=1 / sig ** 2, lamda=0.3, gamma=15.0, isotropic=False)
    x0 = img[None] + rng.normal(0, 12, (2,) + shape)
    noise = rng.standard_normal((2, 2) + shape)
    smp = la.MYULASampler(pf, la.WaveletL1(shape, sigma=0.3), shape, n_chains=2, tau=tau, gamma=gam_m, noise="injected")
    smp.set_state(x0)
    x = x0.copy()
    for it in range(2):
        smp.step(1, noise=noise[it:it + 1])
        gr = np.stack([of.grad(x[c].ravel().copy()).reshape(shape) for c in range(2)])
        x = (1 - tau / gam_m) * x - tau * gr + (tau / gam_m) * O.haar_l1_prox(x, gam_m * 0.3) + np.sqrt(2 * tau) * noise[it]
        assert rel(smp.get_state().cpu().numpy(), x) < 5e-6 * (it + 1)
    assert "block" in smp.kernel_name
    smp.close()


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_me_tv_value_grad_and_myula_match_reference_class(la, golden, tag):
    """ME-TV branch (Op2 = None; TV prox with niter_l2 = 50 iterations inside the gradient, algs.py:169,282) against the
    reference's own class; the 50 dual iterations are chained through HBM-resident state in chunks of 8."""
    g = golden("algs.npz")
    sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0 = [float(v) for v in g["params"]]
    ny, nx, k, seed = [int(v) for v in g[f"{tag}_meta"]]
    h, y = g[f"{tag}_h"], g[f"{tag}_y"]
    H = la.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    xt = g[f"{tag}_ncvx_x"]
    me = la.L2_ncvx_tv(dims=(ny, nx), Op=H, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=15.0, isotropic=True,
                       niter=50, warm=True, rtol=0.0)      # algs.npz pins the loops at rtol = 0; the class's own rtol = 1e-4: test_gpu_rtol.py
    got, ref = me.grad(xt.copy()), g[f"{tag}_ncvx_me_grad"]
    l2o = O.L2(Op=O.Convolve2D((ny, nx), h, offset=(k // 2, k // 2)), b=y.ravel(), sigma=1 / sigma ** 2)
    scale = (1 / sigma ** 2) * (np.linalg.norm(l2o.Op.rmatvec(l2o.Op.matvec(xt))) + np.linalg.norm(l2o.Op.rmatvec(y.ravel())))
    assert np.linalg.norm(got - ref) < 2e-7 * scale and rel(got, ref) < 1e-4, rel(got, ref)
    val = float(g[f"{tag}_ncvx_me_val"])
    assert abs(me(xt.copy()) - val) < 1e-5 * abs(val)
    gx = g[f"{tag}_myula_me_tv"]
    out = la.MoreauYosidaUnadjustedLangevin(me, la.TV((ny, nx), sigma=tau_reg, niter=10), np.zeros(ny * nx), tau=tau_myula,
                                            gamma=gamma_myula, niter=gx.shape[0], seed=seed, rng="pcg64")
    assert rel(out, gx) < 5e-5, rel(out, gx)


def test_chunked_tv_prox_is_exact(la):
    """TV prox with K > 12 runs as chained chunks of 8 iterations (dual state in HBM): identical to the oracle's K iterations."""
    rng = np.random.default_rng(1)
    shape = (70, 90)
    x = rng.normal(100, 15, (2,) + shape)
    for K in (13, 17, 18, 24, 50, 64):       # 17, 18: a last chunk of one / two iterations (round 3: the resumed chunk's halo was one pixel short)
        tv = la.TV(shape, sigma=1.0, niter=K)
        out = tv.prox(x, 2.5)
        for c in range(2):
            # fp32 rounding grows with the number of momentum iterations (not with the chunking): 2e-6 per iteration
            assert rel(out[c], O.tv_prox_fgp(x[c], 2.5, K)) < 2e-6 * K, (K, rel(out[c], O.tv_prox_fgp(x[c], 2.5, K)))


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_ncvx_prox_and_ulpda_match_reference(la, golden, tag):
    """L2_ncvx_tv.prox and ULPDA with the non-log-concave data term (prox_lmc_deconv.py:478-487) against outputs of the
    reference's own class incl. its scipy LSQR solve (which stops at ~1e-6 relative residual: tolerance 1e-4)."""
    g = golden("algs.npz")
    sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0 = [float(v) for v in g["params"]]
    ny, nx, k, seed = [int(v) for v in g[f"{tag}_meta"]]
    h, y = g[f"{tag}_h"], g[f"{tag}_y"]
    H = la.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    G = la.Gradient((ny, nx))
    mk = lambda: la.L2_ncvx_tv(dims=(ny, nx), Op=H, Op2=G, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=15.0,
                               isotropic=True, niter=50, warm=True)
    mc = mk()
    vp = g[f"{tag}_ncvx_prox_in"]
    keep = vp.copy()
    assert rel(mc.prox(vp, tau0), g[f"{tag}_ncvx_prox_out1"]) < 1e-4
    np.testing.assert_array_equal(vp, keep)                     # input untouched
    assert rel(mc.prox(vp + 1.0, tau0), g[f"{tag}_ncvx_prox_out2"]) < 1e-4
    gx = g[f"{tag}_ulpda_mc"]
    xs = la.UnadjustedLangevinPrimalDual(mk(), la.L21(ndim=2, sigma=tau_reg), G, tau=tau0, mu=mu0, theta=1.0,
                                         x0=np.zeros(ny * nx), gfirst=False, niter=gx.shape[0], seed=seed, rng="pcg64")
    assert rel(xs, gx) < 2e-4, rel(xs, gx)
    # ME-TV branch (algs.py:221-223)
    mke = lambda: la.L2_ncvx_tv(dims=(ny, nx), Op=H, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=15.0,
                                isotropic=True, niter=50, warm=True, rtol=0.0)
    assert rel(mke().prox(vp, tau0), g[f"{tag}_ncvx_me_prox_out"]) < 1e-4
    gx = g[f"{tag}_ulpda_me"]
    xs = la.UnadjustedLangevinPrimalDual(mke(), la.L21(ndim=2, sigma=tau_reg), G, tau=tau0, mu=mu0, theta=1.0,
                                         x0=np.zeros(ny * nx), gfirst=False, niter=gx.shape[0], seed=seed, rng="pcg64")
    assert rel(xs, gx) < 2e-4, rel(xs, gx)
