"""Chain probes (lmc_chain_probes) against the oracle, and R-hat / ESS of real MYULA chains through the drop-in."""
import numpy as np
import pytest
import torch

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def la():
    import lmc_atomi_amd as la
    return la


@pytest.mark.parametrize("shape,grid,n", [((64, 64), (8, 8), 5), ((13, 22), (3, 4), 4), ((512, 512), (8, 8), 3), ((40, 264), (1, 1), 2),
                                          ((7, 300), (7, 300), 2), ((33, 1000), (4, 300), 2), ((5, 5), (2, 3), 70000)])
def test_chain_probes_match_oracle(la, shape, grid, n):
    rng = np.random.default_rng(shape[0] + grid[1])
    x = rng.uniform(0, 255, (n,) + shape).astype(np.float32)
    got = la.chain_probes(x, grid).cpu().numpy()
    want = O.chain_probes(x, *grid)
    assert got.shape == want.shape
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)


def test_chain_probes_rejects_bad_grids(la):
    x = torch.zeros((2, 8, 8), device="cuda")
    for grid in [(9, 1), (1, 9)]:
        with pytest.raises(la.LMCError):
            la.chain_probes(x, grid)
    for grid in [(0, 1), (-1, 2)]:
        with pytest.raises(ValueError):
            la.chain_probes(x, grid)
    with pytest.raises(ValueError):
        la.chain_probes(torch.zeros(10, device="cuda"), (1, 1))


def _problem(la, H, W, prior="tv"):
    rng = np.random.default_rng(2)
    u = np.kron(rng.uniform(20, 235, (H // 8, W // 8)), np.ones((8, 8)))
    h = np.ones((5, 5)) / 25.0
    sigma = 0.75
    y = O.blur(u, h, (2, 2)) + rng.normal(0, sigma, (H, W))
    pf = la.L2(Op=la.Convolve2D((H, W), h, offset=(2, 2)), b=y.ravel(), sigma=1 / sigma ** 2)
    pg = la.TV((H, W), sigma=0.3, niter=10) if prior == "tv" else la.L2(sigma=0.05)
    return pf, pg, sigma, y


def test_myula_dropin_reports_rhat_and_ess(la):
    H, W, C = 32, 136, 16
    _, pg, sigma, y = _problem(la, H, W, prior="l2")
    # denoising + l2 prior: a Gaussian target whose every mode contracts by 1 - tau (1/sigma^2 + ...) = 0.8 per iteration.
    # (With the blur instead, the modes at the zeros of the box blur's transfer function take ~200 iterations and leak into the
    # block means: R-hat 1.4 after 640 iterations -- the diagnostic at work, but not a unit test.)
    pf = la.L2(b=y.ravel(), sigma=1 / sigma ** 2, dims=(H, W))
    kw = dict(tau=0.2 * sigma ** 2, gamma=sigma ** 2, seed=1, n_chains=C, dims=(H, W))
    # stationary chains (started from a long run's state) mix: R-hat near 1
    warm = la.MoreauYosidaUnadjustedLangevin(pf, pg, np.zeros(H * W), niter=400, **kw)
    res = la.MoreauYosidaUnadjustedLangevin(pf, pg, warm.state, niter=239, burn_in=40, thin=2, diagnostics=(2, 4), **kw)
    d = res.diagnostics
    assert res.trace.shape == (100, C, 10) and d["n_kept"] == 100 and d["n_chains"] == C and len(d["names"]) == 10
    assert res.count == 100 * C                       # recorded at exactly the iterations that entered the moments
    assert d["rhat_max"] < 1.2, d
    assert d["ess_min"] > 100, d
    # the trace is what the oracle formulas see
    tr = res.trace.cpu().numpy()
    for q in (0, 7, 9):
        np.testing.assert_allclose(float(d["rhat"][q]), O.split_rhat(tr[:, :, q]), rtol=1e-9)
        np.testing.assert_allclose(float(d["ess"][q]), O.ess_geyer(tr[:, :, q]), rtol=1e-8)
    # probes of the final state = last trace row
    np.testing.assert_allclose(tr[-1, :, :8], O.chain_probes(res.state.cpu().numpy(), 2, 4), rtol=1e-6)
    # chains frozen at different states (tau -> tiny) do not agree: R-hat is large
    x0 = np.stack([np.full((H, W), 20.0 * c) for c in range(C)])
    cold = la.MoreauYosidaUnadjustedLangevin(pf, pg, x0, niter=40, diagnostics=True,
                                             **dict(kw, tau=1e-6 * sigma ** 2))
    assert cold.diagnostics["rhat_max"] > 5
    assert cold.trace.shape == (40, C, 66)


def test_chain_trace_as_callback_with_ulpda_sampler(la):
    H, W, C = 16, 16, 4
    pf, pg, sigma, y = _problem(la, H, W)
    smp = la.MYULASampler(pf, pg, (H, W), n_chains=C, tau=0.2 * sigma ** 2, gamma=sigma ** 2, seed=0)
    smp.set_state(np.zeros((H, W), dtype=np.float32))
    tr = la.ChainTrace(smp, grid=(2, 2), energies=False)
    for _ in range(6):
        smp.step(1)
        tr(None)
    s = tr.summary()
    assert tr.trace().shape == (6, C, 4) and s["rhat"].shape == (4,) and np.isfinite(s["rhat_max"])
    smp.close()


def test_ulpda_dropin_reports_diagnostics(la):
    H, W, C = 16, 24, 6
    pf, _, sigma, y = _problem(la, H, W)
    pf.niter = 20
    res = la.UnadjustedLangevinPrimalDual(pf, la.L21(ndim=2, sigma=0.3), la.Gradient((H, W)), np.zeros(H * W), tau=0.95 * sigma ** 2, mu=1.0,
                                          theta=1.0, niter=31, gfirst=False, seed=2, n_chains=C, dims=(H, W), burn_in=5, thin=3,
                                          diagnostics=(2, 2))
    d = res.diagnostics
    assert res.trace.shape == (9, C, 6) and d["n_kept"] == 9 and res.count == 9 * C     # iterations 5, 8, ..., 29
    assert d["rhat"].shape == (6,) and np.isfinite(d["rhat_max"]) and d["ess_min"] > 0
    np.testing.assert_allclose(float(d["rhat"][0]), O.split_rhat(res.trace.cpu().numpy()[:, :, 0]), rtol=1e-9)
