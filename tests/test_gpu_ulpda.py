"""GPU parity of the ULPDA path (algs.py:295-474) and of the implicit L2 data step (row a8), through the C ABI.

The inner solver is `cg_niter` conjugate-gradient iterations in fp32 on the device vs float64 in the oracle; after
50 iterations both are converged to the same solution, so the tolerance is that of the outer recursion."""
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


@pytest.mark.parametrize("k,shape", [(5, (16, 16)), (7, (24, 18)), (5, (40, 70))])
def test_l2_implicit_step_matches_oracle(la, k, shape):
    rng = np.random.default_rng(k)
    h = np.ones((k, k)) / k ** 2
    b = rng.normal(100, 20, shape)
    v = rng.normal(100, 20, (3,) + shape)
    l2 = la.L2(Op=la.Convolve2D(shape, h), b=b.ravel(), sigma=1 / 0.75 ** 2, niter=50, warm=True)
    out = l2.prox(v.reshape(3, -1), 0.53)
    for c in range(3):
        l2o = O.L2(Op=O.Convolve2D(shape, h), b=b.ravel(), sigma=1 / 0.75 ** 2, niter=200, warm=False)
        assert rel(out[c], l2o.prox(v[c].ravel(), 0.53)) < 2e-5
    # a second call is warm-started from the first solution: the residual of the normal equations stays tiny
    out2 = l2.prox(v.reshape(3, -1), 0.53)
    assert rel(out2, out) < 1e-5
    # truncated solve, cold start: same iterate as the oracle's CG after the same number of iterations
    l2c = la.L2(Op=la.Convolve2D(shape, h), b=b.ravel(), sigma=1 / 0.75 ** 2, niter=3, warm=False)
    l2oc = O.L2(Op=O.Convolve2D(shape, h), b=b.ravel(), sigma=1 / 0.75 ** 2, niter=3, warm=False)
    assert rel(l2c.prox(v[0].ravel(), 0.53), l2oc.prox(v[0].ravel(), 0.53)) < 1e-5


def test_l2_closed_form_steps(la):
    shape = (20, 30)
    rng = np.random.default_rng(0)
    b = rng.normal(100, 20, shape)
    v = rng.normal(100, 20, shape)
    mask = (rng.uniform(size=shape) < 0.5).astype(np.float64)
    out = la.L2(b=b, sigma=1.7, dims=shape).prox(v.ravel(), 0.4)
    assert rel(out, ((v + 0.4 * 1.7 * b) / (1 + 0.4 * 1.7)).ravel()) < 1e-6
    out = la.L2(Op=la.Diagonal(mask, dims=shape), b=b, sigma=1.7, dims=shape).prox(v.ravel(), 0.4)
    assert rel(out, ((v + 0.4 * 1.7 * mask * b) / (1 + 0.4 * 1.7 * mask ** 2)).ravel()) < 1e-6


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_drop_in_ulpda_reproduces_reference_trajectories(la, golden, tag):
    """la.UnadjustedLangevinPrimalDual(..., rng='pcg64') against trajectories of the reference's own
    UnadjustedLangevinPrimalDual (tests/golden/algs.npz; inner solver = the oracle's 50-iteration CG)."""
    g = golden("algs.npz")
    sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0 = [float(v) for v in g["params"]]
    ny, nx, k, seed = [int(v) for v in g[f"{tag}_meta"]]
    h, y = g[f"{tag}_h"], g[f"{tag}_y"]
    H = la.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    G = la.Gradient((ny, nx))
    for gfirst in (False, True):
        gx, gy = g[f"{tag}_ulpda_l21_gfirst{int(gfirst)}_x"], g[f"{tag}_ulpda_l21_gfirst{int(gfirst)}_y"]
        l2 = la.L2(Op=H, b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
        seen = []
        xs, ys = la.UnadjustedLangevinPrimalDual(l2, la.L21(ndim=2, sigma=tau_reg), G, tau=tau0, mu=mu0, theta=1.0,
                                                 x0=np.zeros(ny * nx), gfirst=gfirst, niter=gx.shape[0], seed=seed,
                                                 returny=True, rng="pcg64", callback=lambda x: seen.append(x[0]))
        assert xs.shape == gx.shape and ys.shape == gy.shape and len(seen) == gx.shape[0]
        assert rel(xs, gx) < 1e-4, (tag, gfirst, rel(xs, gx))
        assert rel(ys, gy) < 1e-3, (tag, gfirst, rel(ys, gy))
    if f"{tag}_ulpda_l1" in g.files:
        gx = g[f"{tag}_ulpda_l1"]
        l2 = la.L2(Op=H, b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
        xs = la.UnadjustedLangevinPrimalDual(l2, la.L1(sigma=tau_reg), G, tau=tau0, mu=mu0, theta=1.0, x0=np.zeros(ny * nx),
                                             gfirst=False, niter=gx.shape[0], seed=seed, rng="pcg64")
        assert rel(xs, gx) < 1e-4, rel(xs, gx)


def test_ulpda_many_chains_philox_vs_oracle(la):
    """Batched Philox ULPDA on the GPU == per-chain oracle loops driven by the oracle's Philox field; per-iteration
    step arrays; energies and moments."""
    shape = (18, 40)
    sigma, tau_reg = 0.75, 0.3
    rng = np.random.default_rng(3)
    img = np.zeros(shape); img[4:12, 10:30] = 150.0
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, sigma, shape)
    C, off, seed, nit = 3, 11, 5, 6
    taus = np.linspace(0.5, 0.3, nit); mus = np.linspace(1.0, 0.8, nit)
    l2 = la.L2(Op=la.Convolve2D(shape, h), b=y.ravel(), sigma=1 / sigma ** 2, niter=40, warm=True)
    smp = la.ULPDASampler(l2, la.L21(sigma=tau_reg), la.Gradient(shape), shape, n_chains=C, tau=taus[0], mu=mus[0], theta=1.0,
                          gfirst=False, seed=seed, chain_offset=off, moments=True)
    for it in range(nit):
        smp.set_steps(taus[it], mus[it])
        smp.step(1)
    got = smp.get_state().cpu().numpy()
    goty = smp.get_dual().cpu().numpy()
    Gop = O.Gradient(shape)
    s1 = np.zeros(shape)
    for c in range(C):
        l2o = O.L2(Op=O.Convolve2D(shape, h), b=y.ravel(), sigma=1 / sigma ** 2, niter=40, warm=True)
        noise = np.stack([O.philox_normals(seed, k, [off + c], *shape)[0].ravel().astype(np.float64) for k in range(nit)])
        xs, ys = O.ulpda(l2o, O.L21(sigma=tau_reg), Gop, np.zeros(shape[0] * shape[1]), taus, mus, theta=1.0, niter=nit,
                         gfirst=False, returny=True, noise=noise)
        assert rel(got[c].ravel(), xs[-1]) < 1e-4, (c, rel(got[c].ravel(), xs[-1]))
        assert rel(goty[c].ravel(), ys[-1]) < 1e-3
        s1 += xs.sum(axis=0).reshape(shape)
    m1, m2, cnt = smp.moments()
    assert cnt == C * nit and rel(m1.cpu().numpy(), s1) < 1e-4
    f, g = smp.energies()
    l2o = O.L2(Op=O.Convolve2D(shape, h), b=y.ravel(), sigma=1 / sigma ** 2)
    assert abs(float(f[0]) - l2o(got[0].ravel())) < 1e-4 * l2o(got[0].ravel())
    assert abs(float(g[0]) - O.L21(sigma=tau_reg)(Gop.matvec(got[0].ravel()))) < 1e-4 * float(g[0])
    smp.close()


def test_ulpda_identity_data_and_anisotropic_prior(la):
    shape = (12, 20)
    rng = np.random.default_rng(4)
    b = rng.normal(100, 20, shape)
    nit, seed = 5, 2
    lf = la.L2(b=b, sigma=1.3, dims=shape)
    smp = la.ULPDASampler(lf, la.L1(sigma=0.4), la.Gradient(shape), shape, n_chains=1, tau=0.4, mu=0.3, theta=0.5, gfirst=True,
                          seed=seed, noise="injected")
    noise = rng.standard_normal((nit, 1) + shape)
    smp.step(nit, noise=noise)
    xs = O.ulpda(O.L2(b=b.ravel(), sigma=1.3), O.L1(sigma=0.4), O.Gradient(shape), np.zeros(shape[0] * shape[1]), 0.4, 0.3,
                 theta=0.5, niter=nit, gfirst=True, noise=noise.reshape(nit, -1))
    assert rel(smp.get_state().cpu().numpy().ravel(), xs[-1]) < 2e-5
    smp.close()


def test_ulpda_errors(la):
    shape = (8, 8)
    lf = la.L2(b=np.zeros(shape), sigma=1.0, dims=shape)
    with pytest.raises(NotImplementedError):
        la.ULPDASampler(lf, la.TV(shape, 0.3), la.Gradient(shape), shape, tau=0.1, mu=0.1)
    with pytest.raises(NotImplementedError):
        la.ULPDASampler(lf, la.L21(sigma=0.3), la.Identity(64), shape, tau=0.1, mu=0.1)
    smp = la.MYULASampler(lf, None, shape, tau=0.1, gamma=0.5)
    from lmc_atomi_amd import _capi, _dev
    with pytest.raises(la.LMCError):
        _capi.check(_dev.lib().lmc_sampler_set_steps(smp._h, 0.1, 0.1))   # not a ULPDA handle


def test_cg_early_exit_matches_fixed_iterations(la):
    """The inner solver stops when every chain reaches |r| <= 1e-6 |b| (the reference's lsqr btol, algs.py:250): same solution
    as the fixed 50 iterations to fp32 accuracy, for the operator-level prox and for ULPDA trajectories; tol = 0 restores the
    fixed count."""
    rng = np.random.default_rng(12)
    shape = (40, 64)
    img = np.zeros(shape); img[8:30, 10:50] = 120.0
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    v = img[None] + rng.normal(0, 5, (3,) + shape)
    outs = {}
    prev = la.set_cg_tolerance(1e-6)
    assert abs(prev - 1e-6) < 1e-12
    for tol in (0.0, 1e-6):
        la.set_cg_tolerance(tol)
        l2 = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y.ravel(), sigma=1 / 0.75 ** 2, niter=50, warm=False)
        outs[tol] = np.stack([l2.prox(v[i].ravel().copy(), 0.53) for i in range(3)])
        smp = la.ULPDASampler(la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y.ravel(), sigma=1 / 0.75 ** 2, niter=50, warm=True),
                              la.L21(ndim=2, sigma=0.3), la.Gradient(shape), shape, n_chains=3, tau=0.95 * 0.5625, mu=1.0, theta=1.0,
                              gfirst=False, seed=5)
        smp.set_state(v)
        smp.step(4)
        outs[("u", tol)] = smp.get_state().cpu().numpy()
        smp.close()
    la.set_cg_tolerance(1e-6)
    ref = O.L2(Op=O.Convolve2D(shape, h, offset=(2, 2)), b=y.ravel(), sigma=1 / 0.75 ** 2, niter=50, warm=False)
    want = np.stack([ref.prox(v[i].ravel().copy(), 0.53) for i in range(3)])
    assert rel(outs[0.0], want) < 2e-6 and rel(outs[1e-6], want) < 5e-6
    assert rel(outs[("u", 1e-6)], outs[("u", 0.0)]) < 2e-5


@pytest.mark.parametrize("k,off,shape", [(5, (2, 2), (32, 64)), (5, (2, 2), (48, 264)), (7, (3, 3), (40, 128)), (6, (3, 3), (36, 200)),
                                          (5, (2, 2), (64, 512)), (3, (1, 1), (33, 12)), (7, (3, 3), (40, 512)), (6, (3, 3), (33, 384))])
def test_l2_implicit_step_reaches_the_reference_tolerance(la, k, off, shape):
    """Shapes the row-streaming kernel covers: the implicit step is the Chebyshev semi-iteration (one launch per iteration, iteration
    count from the spectral bound).  The residual of the normal equations must meet the reference solver's stopping rule
    |r| <= 1e-6 |b| (scipy lsqr btol, algs.py:250) up to fp32 rounding, cold and warm, and agree with a converged oracle solve."""
    rng = np.random.default_rng(k + shape[1])
    h = rng.uniform(0.5, 1.5, (k, 1)) * rng.uniform(0.5, 1.5, (1, k))
    h /= h.sum()
    b = rng.normal(100, 20, shape)
    v = rng.normal(100, 20, (4,) + shape)
    sig, tau = 1 / 0.75 ** 2, 0.53
    l2 = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=b.ravel(), sigma=sig, niter=50, warm=True)
    ts = tau * sig
    for rep in range(2):                       # second call: warm start from the first solution
        u = np.asarray(l2.prox(v.reshape(4, -1), tau), dtype=np.float64).reshape((4,) + shape)
        rhs = v + ts * O.blur_adjoint(b, h, off)[None]
        res = rhs - (u + ts * O.blur_adjoint(O.blur(u, h, off), h, off))
        for c in range(4):
            assert np.linalg.norm(res[c]) <= 3e-6 * np.linalg.norm(rhs[c]), (rep, c, np.linalg.norm(res[c]) / np.linalg.norm(rhs[c]))
    l2o = O.L2(Op=O.Convolve2D(shape, h, off), b=b.ravel(), sigma=sig, niter=300, warm=False)
    assert rel(u[1].ravel(), l2o.prox(v[1].ravel(), tau)) < 5e-6


@pytest.mark.parametrize("gfirst", [True, False])
@pytest.mark.parametrize("shape,iso", [((24, 40), True), ((17, 64), False), ((9, 10), True)])
def test_ulpda_batched_step_equals_single_steps(la, gfirst, shape, iso):
    """step(n) must leave exactly the primal and dual state of n calls of step(1), however the iterations are grouped into calls."""
    rng = np.random.default_rng(5)
    h = np.ones((5, 5)) / 25.0
    b = rng.normal(100, 20, shape)
    x0 = rng.normal(100, 20, (3,) + shape)
    y0 = rng.normal(0, 0.2, (3, 2) + shape)
    z = rng.normal(0, 0.1, shape) if not gfirst else None
    outs = []
    for chunks in ([7], [1] * 7, [3, 4]):
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=b.ravel(), sigma=1 / 0.75 ** 2, niter=50, warm=True)
        pg = la.L21(ndim=2, sigma=0.3) if iso else la.L1(sigma=0.3)
        smp = la.ULPDASampler(pf, pg, la.Gradient(shape), shape, n_chains=3, tau=0.5, mu=0.9, theta=1.0, gfirst=gfirst, seed=9, z=z)
        smp.set_state(x0)
        smp.set_dual(y0)
        for n in chunks:
            smp.step(n)
        outs.append((smp.get_state().cpu().numpy(), smp.get_dual().cpu().numpy()))
        smp.close()
    for xs, ys in outs[1:]:
        assert np.array_equal(xs, outs[0][0]) and np.array_equal(ys, outs[0][1])


@pytest.mark.parametrize("tau,scale,signed", [(5.0, 1.0, False), (0.53, 1.7, False), (0.53, 1.0, True), (40.0, 1.0, False)])
def test_l2_implicit_step_spectral_bound_holds(la, tau, scale, signed):
    """The Chebyshev iteration relies on spec(H^T H) <= (sum |h|)^2: large steps (condition number ~ 10 and, at tau = 40, ~ 70: too many
    iterations for the cap, so CG takes over), kernels that do not sum to one, kernels with negative taps."""
    rng = np.random.default_rng(11)
    shape = (40, 64)
    u1, u2 = rng.uniform(0.5, 1.5, 5), rng.uniform(0.5, 1.5, 5)
    if signed:
        u1[1] *= -1.0
        u2[3] *= -1.0
    h = np.outer(u1, u2)
    h *= scale / np.abs(h).sum()
    b = rng.normal(100, 20, shape)
    v = rng.normal(100, 20, (2,) + shape)
    sig = 1 / 0.75 ** 2
    l2 = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=b.ravel(), sigma=sig, niter=50, warm=False)
    ts = tau * sig
    u = np.asarray(l2.prox(v.reshape(2, -1), tau), dtype=np.float64).reshape((2,) + shape)
    rhs = v + ts * O.blur_adjoint(b, h, (2, 2))[None]
    res = rhs - (u + ts * O.blur_adjoint(O.blur(u, h, (2, 2)), h, (2, 2)))
    for c in range(2):
        assert np.linalg.norm(res[c]) <= 5e-6 * np.linalg.norm(rhs[c]), np.linalg.norm(res[c]) / np.linalg.norm(rhs[c])


@pytest.mark.parametrize("shape,C,band", [((72, 128), 3, 32), ((150, 512), 2, 0), ((100, 264), 40, 40), ((37, 36), 2, 0), ((300, 256), 1, 0),
                                          ((90, 512), 2, 32)])
def test_ulpda_two_chebyshev_iterations_per_launch(la, shape, C, band, monkeypatch):
    """The implicit step with two Chebyshev iterations per launch (lmc_cheb_pair.hip: wave pairs, LDS hand-off, solution delivered in the sampler's
    second array): several bands per chain, 4 and 8 pixels per lane, partially filled last lanes, warm starts over 5 iterations -- against the
    per-chain loops of the CPU checker with the same Philox field."""
    monkeypatch.setenv("LMC_CHEB_PAIR", "2")         # also where the launch is too small for the pairs to pay (read per solve)
    if band:
        monkeypatch.setenv("LMC_PAIR_BAND", str(band))       # rows per wave pair (default: >= 128)
    sigma, tau_reg = 0.75, 0.3
    rng = np.random.default_rng(shape[1])
    img = np.zeros(shape); img[shape[0] // 5:shape[0] // 2, shape[1] // 6:2 * shape[1] // 3] = 150.0
    img += np.linspace(0, 30, shape[1])[None, :]
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, sigma, shape)
    seed, nit, cho = 11, 5, 3
    l2 = la.L2(Op=la.Convolve2D(shape, h), b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
    smp = la.ULPDASampler(l2, la.L21(sigma=tau_reg), la.Gradient(shape), shape, n_chains=C, tau=0.95 * sigma ** 2, mu=1.0, theta=1.0,
                          gfirst=False, seed=seed, chain_offset=cho)
    smp.step(nit)
    assert "pairs" in smp.kernel_name, smp.kernel_name
    got = smp.get_state().cpu().numpy()
    goty = smp.get_dual().cpu().numpy()
    Gop = O.Gradient(shape)
    for c in sorted({0, C - 1, C // 2}):
        l2o = O.L2(Op=O.Convolve2D(shape, h), b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
        noise = np.stack([O.philox_normals(seed, k, [cho + c], *shape)[0].ravel().astype(np.float64) for k in range(nit)])
        xs, ys = O.ulpda(l2o, O.L21(sigma=tau_reg), Gop, np.zeros(shape[0] * shape[1]), 0.95 * sigma ** 2, 1.0, theta=1.0, niter=nit,
                         gfirst=False, returny=True, noise=noise)
        assert rel(got[c].ravel(), xs[-1]) < 2e-4, (c, rel(got[c].ravel(), xs[-1]))
        assert rel(goty[c].ravel(), ys[-1]) < 2e-3
        rowerr = np.abs(got[c] - xs[-1].reshape(shape)).max(axis=1)
        assert rowerr.max() < 0.05, (int(rowerr.argmax()), float(rowerr.max()))       # nothing special at band seams
    smp.close()


@pytest.mark.parametrize("shape,C,prior", [((40, 64), 3, "l21"), ((37, 36), 2, "l1"), ((150, 512), 2, "l21"), ((24, 8), 1, "l21")])
def test_ulpda_dual_update_fused_with_the_next_right_hand_side(la, shape, C, prior, monkeypatch):
    """gfirst = False: the dual update of iteration k and the right-hand side of iteration k + 1 in one pass (28 instead of 36 B per pixel; the
    dual projections of the upper and left neighbours are recomputed): bit-identical to the two passes, state and dual, inside one call of
    several iterations and across calls (where the two passes run)."""
    rng = np.random.default_rng(shape[1] + 1)
    img = np.zeros(shape); img[shape[0] // 5:shape[0] // 2, shape[1] // 6:2 * shape[1] // 3] = 150.0
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    outs = {}
    for mode in ("0", "1"):
        monkeypatch.setenv("LMC_ULPDA_DUAL_RHS", mode)
        l2 = la.L2(Op=la.Convolve2D(shape, h), b=y.ravel(), sigma=1 / 0.5625, niter=50, warm=True)
        pg = la.L21(sigma=0.3) if prior == "l21" else la.L1(sigma=0.3)
        smp = la.ULPDASampler(l2, pg, la.Gradient(shape), shape, n_chains=C, tau=0.95 * 0.5625, mu=1.0, theta=1.0, gfirst=False, seed=3, chain_offset=1)
        smp.step(4)
        smp.set_steps(0.8 * 0.5625, 0.9)            # between calls: the right-hand side is formed again with the new steps
        smp.step(3)
        outs[mode] = (smp.get_state().cpu().numpy(), smp.get_dual().cpu().numpy())
        smp.close()
    np.testing.assert_array_equal(outs["1"][0], outs["0"][0])
    np.testing.assert_array_equal(outs["1"][1], outs["0"][1])
