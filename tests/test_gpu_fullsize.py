"""Properties at the BASELINE sizes (512x512 x 1024 chains, 256x256 x 128 chains, 512x512 inpainting): too large for the
oracle to replay in full, so the checks are size-independent -- a chain of the big batch equals the same chain run alone
(sharding invariance: noise keyed by the global chain id), one chain / one step equals the oracle, moments equal the sum over
chains of the states, and every kernel variant that covers the configuration agrees."""
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


def scene(shape, seed=1234):
    rng = np.random.default_rng(seed)
    ny, nx = shape
    img = np.zeros(shape)
    for _ in range(12):
        i0, j0 = rng.integers(0, ny - 8), rng.integers(0, nx - 8)
        i1, j1 = rng.integers(i0 + 4, ny + 1), rng.integers(j0 + 4, nx + 1)
        img[i0:i1, j0:j1] = rng.uniform(20, 235)
    img += np.linspace(0, 20, nx)[None, :]
    return np.clip(img, 0, 255)


CASES = [
    ("config3_tv", (512, 512), 1024, "blur", "tv"),        # BASELINE config 3: deblur + isotropic TV, 1024 chains
    ("config2_l2", (256, 256), 128, "blur", "l2"),         # BASELINE config 2: deblur + l2 prior, 128 chains
    ("config5_haar", (512, 512), 512, "mask", "haar"),     # BASELINE config 5 shape: inpainting mask + Haar-l1 (chains of one GPU)
    ("config5_as_specified", (512, 512), 512, "mask_mc", "haar"),   # + the L2_ncvx_tv Moreau-difference (MC-TV) term, lamda = 0.3, gamma = 15
]


@pytest.mark.parametrize("name,shape,C,data,prior", CASES)
def test_fullsize_properties(la, name, shape, C, data, prior):
    import torch
    sigma, tau_reg = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    img = scene(shape)
    rng = np.random.default_rng(0)
    h, off, mask = None, None, None
    if data == "blur":
        h, off = np.ones((5, 5)) / 25, (2, 2)
        y = O.blur(img, h, off) + rng.normal(0, sigma, shape)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    else:
        mask = (np.random.default_rng(7).uniform(size=shape) < 0.5).astype(np.float64)
        y = mask * (img + rng.normal(0, sigma, shape))
        pf = la.L2(Op=la.Diagonal(mask, dims=shape), b=y, sigma=1 / sigma ** 2, dims=shape)
        if data == "mask_mc":        # non-log-concave: f(x) = sigma/2 ||M x - y||^2 - lamda * env_gamma(l1)(grad x)   (algs.py:270-291)
            pf = la.L2_ncvx_tv(dims=shape, Op=la.Diagonal(mask, dims=shape), Op2=la.Gradient(shape), b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=15.0,
                               isotropic=True)
            of = O.L2NcvxTV(dims=shape, Op=O.Diagonal(mask), Op2=O.Gradient(shape), b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=15.0, isotropic=True)
    if prior == "tv":
        pg, op = la.TV(shape, sigma=tau_reg, niter=10), {"kind": "tv", "sigma": tau_reg, "niter": 10, "t": gamma}
    elif prior == "l2":
        pg, op = la.L2(sigma=0.05), {"kind": "l2", "sigma": 0.05, "t": gamma}
    else:
        pg, op = la.WaveletL1(shape, sigma=tau_reg), {"kind": "haar", "sigma": tau_reg, "t": gamma}
    seed, nit, base = 11, 3, 4096
    big = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, seed=seed, chain_offset=base, moments=True)
    big.set_state(np.zeros(shape, dtype=np.float32))                    # x0 = 0 (prox_lmc_deconv.py:135)
    big.step(nit)
    xb = big.get_state()
    # (1) sharding invariance, exact: chains picked from the batch, run alone with their global ids
    for c in (0, C // 3, C - 1):
        one = la.MYULASampler(pf, pg, shape, n_chains=1, tau=tau, gamma=gamma, seed=seed, chain_offset=base + c)
        one.set_state(np.zeros(shape, dtype=np.float32))
        one.step(nit)
        assert torch.equal(one.get_state()[0], xb[c]), (name, c)
        one.close()
    # (2) the oracle replays one chain of the batch with the device's own Philox field
    c = C // 2
    x = np.zeros((1,) + shape)
    for k in range(nit):
        xi = O.philox_normals(seed, k, np.array([base + c]), *shape).astype(np.float64)
        if data == "mask_mc":        # the update of algs.py:569 assembled from the checker's class gradient and the Haar prox
            x = ((1 - tau / gamma) * x - tau * of.grad(x.ravel()).reshape(x.shape) + (tau / gamma) * O.haar_l1_prox(x, gamma * tau_reg) + np.sqrt(2 * tau) * xi)
        else:
            x = O.myula_step(x, y, h, off, 1 / sigma ** 2, tau, gamma, op, xi, mask=mask)
    assert rel(xb[c].cpu().numpy(), x[0]) < 5e-6 * nit, (name, rel(xb[c].cpu().numpy(), x[0]))
    # (3) moments = sums over chains and kept iterations; with x0 = 0 and nit steps, recompute the last term from the states
    s1, s2, n = big.moments()
    assert n == nit * C
    big.reset_moments()
    big.step(1)
    t1, t2, n1 = big.moments()
    xs = big.get_state().double()
    assert n1 == C
    assert rel(t1.cpu().numpy(), xs.sum(dim=0).cpu().numpy()) < 1e-9
    assert rel(t2.cpu().numpy(), (xs * xs).sum(dim=0).cpu().numpy()) < 1e-9
    # (4) per-chain energies are finite and chains differ (independent noise)
    f, g = big.energies()
    assert torch.isfinite(f).all() and torch.isfinite(g).all() and f.std() > 0
    big.close()
    torch.cuda.empty_cache()


def test_fullsize_variants_agree_config3(la):
    """512x512 TV K=10: the stage-parallel kernel (default), the split kernel and the LDS-tiled kernel on the same 6 chains."""
    shape = (512, 512)
    img = scene(shape)
    rng = np.random.default_rng(3)
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / 0.75 ** 2)
    outs = {}
    for v in ("tile", "split", "auto"):
        la.set_step_variant(v)
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=6, tau=0.1125, gamma=0.5625, seed=2)
        smp.set_state(img)
        smp.step(4)
        outs[v] = smp.get_state().cpu().numpy()
        if v == "auto":
            assert smp.kernel_name == "myula_step_pipe_kernel"
        smp.close()
    la.set_step_variant("auto")
    assert rel(outs["split"], outs["tile"]) < 2e-6
    assert rel(outs["auto"], outs["tile"]) < 2e-6


def test_fullsize_run_to_run_determinism(la):
    """No race in the wave-to-wave hand-offs of the stage-parallel kernel: the same 512x512 x 512-chain run twice, 40 iterations,
    bit-identical states and moments (any unsynchronised LDS read would show up as run-to-run noise)."""
    import torch
    shape = (512, 512)
    img = scene(shape)
    rng = np.random.default_rng(5)
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / 0.75 ** 2)
    outs = []
    for _ in range(2):
        smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=512, tau=0.1125, gamma=0.5625, seed=77, moments=True)
        smp.set_state(np.zeros(shape, dtype=np.float32))
        smp.step(40)
        assert smp.kernel_name == "myula_step_pipe_kernel"
        s1, s2, n = smp.moments()
        outs.append((smp.get_state().clone(), s1.clone(), n))
        smp.close()
    assert torch.equal(outs[0][0], outs[1][0])
    assert outs[0][2] == outs[1][2]
    # the moment images are sums of fp64 atomics (order varies): equal to rounding
    assert rel(outs[0][1].cpu().numpy(), outs[1][1].cpu().numpy()) < 1e-12
    assert torch.isfinite(outs[0][0]).all()


def test_fullsize_ulpda_pairs_agree_with_single_launches_and_one_chain_alone(la, monkeypatch):
    """ULPDA at 512 x 512 x 512 chains (where the two-iterations-per-launch Chebyshev solve is the default): the same trajectory as with
    single-iteration launches (both solve the implicit step to 1e-6), and chain 300 of the batch equals that chain run alone (one band per
    chain in the batch, four in the solo run: the result does not depend on the band layout)."""
    import torch
    shape = (512, 512)
    img = scene(shape)
    h = np.ones((5, 5)) / 25
    rng = np.random.default_rng(2)
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    C, nit, seed = 512, 3, 21

    def run(n_chains, offset):
        l2 = la.L2(Op=la.Convolve2D(shape, h), b=y.ravel(), sigma=1 / 0.5625, niter=50, warm=True)
        smp = la.ULPDASampler(l2, la.L21(sigma=0.3), la.Gradient(shape), shape, n_chains=n_chains, tau=0.95 * 0.5625, mu=1.0, theta=1.0,
                              gfirst=False, seed=seed, chain_offset=offset)
        smp.step(nit)
        name = smp.kernel_name
        x = smp.get_state()
        out = x[[0, 300 - offset if offset == 0 else 0, -1]].cpu().numpy() if n_chains > 1 else x.cpu().numpy()
        smp.close()
        torch.cuda.empty_cache()
        return out, name

    pairs, name = run(C, 0)
    assert "pairs" in name, name
    monkeypatch.setenv("LMC_CHEB_PAIR", "0")
    single, name0 = run(C, 0)
    assert "pairs" not in name0
    assert rel(pairs, single) < 2e-5, rel(pairs, single)
    monkeypatch.setenv("LMC_CHEB_PAIR", "2")
    solo, name1 = run(1, 300)
    assert "pairs" in name1
    assert rel(solo[0], pairs[1]) < 2e-5, rel(solo[0], pairs[1])
