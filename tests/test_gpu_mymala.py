"""GPU parity of image-scale MYMALA (Metropolis-adjusted MYULA; generalises prox_lmc.py:134-158, whose toy form pins the
oracle's accept / reject rule in tests/test_oracle_golden.py) against the oracle, through the C ABI."""
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


def build(la, kind, shape, rng):
    sigma = 0.75
    img = np.zeros(shape)
    img[shape[0] // 4:shape[0] // 2, shape[1] // 4:3 * shape[1] // 4] = 150.0
    img += np.linspace(0, 30, shape[1])[None, :]
    mask, h, off = None, None, None
    if kind in ("tv", "tv10", "l2"):
        h, off = np.ones((5, 5)) / 25, (2, 2)
        y = O.blur(img, h, off) + rng.normal(0, sigma, shape)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    else:
        mask = (rng.uniform(size=shape) < 0.6).astype(np.float64)
        y = mask * (img + rng.normal(0, sigma, shape))
        pf = la.L2(Op=la.Diagonal(mask, dims=shape), b=y, sigma=1 / sigma ** 2, dims=shape)
    gamma = sigma ** 2
    if kind in ("tv", "tv10"):
        K = 10 if kind == "tv10" else 5
        pg, prior = la.TV(shape, sigma=0.3, niter=K), {"kind": "tv", "sigma": 0.3, "niter": K, "t": gamma}
    elif kind == "l2":
        pg, prior = la.L2(sigma=0.05), {"kind": "l2", "sigma": 0.05, "t": gamma}
    else:
        pg, prior = la.WaveletL1(shape, sigma=2.0), {"kind": "haar", "sigma": 2.0, "t": gamma}
    return img, y, h, off, mask, pf, pg, prior, sigma


@pytest.mark.parametrize("kind,tau_scale,shape,epsg", [("tv", 0.02, (32, 32), 1.0), ("l2", 0.02, (32, 32), 1.0), ("haar", 0.02, (32, 32), 1.0),
                                                        ("tv", 1.0, (32, 32), 1.0),
                                                        # wide images: the step kernel returns f(x'), g(x') as by-products
                                                        ("tv10", 0.02, (24, 160), 1.0), ("tv10", 1.0, (20, 264), 1.0),
                                                        # epsg != 1: the Metropolis target is exp(-f - epsg g), the potential of the MYULA drift
                                                        ("tv", 0.02, (32, 32), 0.5), ("tv10", 0.02, (24, 160), 2.5)])
def test_mymala_matches_oracle_with_injected_noise(la, kind, tau_scale, shape, epsg):
    rng = np.random.default_rng(17)
    img, y, h, off, mask, pf, pg, prior, sigma = build(la, kind, shape, rng)
    gamma, tau = sigma ** 2, tau_scale * sigma ** 2
    prior = dict(prior, t=epsg * gamma)
    C, nit, seed, off_c = 6, 6, 1234, 40
    x0 = img[None] + rng.normal(0, 3, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    smp = la.MYMALASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, epsg=epsg, noise="injected", seed=seed, chain_offset=off_c)
    smp.set_state(x0)
    us = np.stack([O.philox_uniforms(seed, k, off_c + np.arange(C)) for k in range(nit)])
    xo, acc_o, la_o = O.mymala_batched(x0, y, h, off, 1 / sigma ** 2, tau, gamma, prior, nit, lambda k: noise[k], lambda k: us[k],
                                       mask=mask, epsg=epsg)
    # the device follows the oracle decision unless log u is within the fp32 energy error of log alpha; replay the oracle
    # chain by chain with the device's own decisions to compare states exactly where a borderline decision differs
    x = x0.copy()
    las = []
    for k in range(nit):
        smp.step(1, noise=noise[k:k + 1])
        _, la_d = smp.acceptance()
        las.append(la_d.cpu().numpy())
    las = np.array(las)
    f0, g0 = O.energies(x0, y, h, off, 1 / sigma ** 2, prior, mask=mask)
    scale = np.abs(f0 + epsg * g0).max()
    err = np.abs(las - la_o)
    margin = np.abs(np.log(us) - la_o)
    safe = (margin > 10 * (1e-6 * scale + 1e-3)).all(axis=0)        # chains whose every decision is clear-cut
    assert safe.sum() >= C // 2, "test problem too borderline"
    assert (err[:, safe] < 2e-6 * scale + 2e-3).all(), (err.max(), scale)
    acc_d, _ = smp.acceptance()
    acc_d = acc_d.cpu().numpy()
    got = smp.get_state().cpu().numpy()
    assert (acc_d[safe] == acc_o[safe]).all(), (acc_d, acc_o)
    assert rel(got[safe], xo[safe]) < 2e-5, rel(got[safe], xo[safe])
    if tau_scale >= 1.0 and shape == (32, 32):
        assert (acc_o[safe] < nit).any(), "expected some rejections at this step size"
    else:
        assert (acc_o[safe] > 0).any(), "expected some acceptances at this step size"
    assert smp.iteration == nit
    if kind == "tv10":
        assert smp.kernel_name == "myula_step_pipe_kernel"
    smp.close()


def test_mymala_philox_mode_moments_and_set_state(la):
    shape = (24, 40)
    rng = np.random.default_rng(2)
    img, y, h, off, mask, pf, pg, prior, sigma = build(la, "tv", shape, rng)
    gamma, tau = sigma ** 2, 0.01 * sigma ** 2
    C = 8
    a = la.MYMALASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, seed=9, moments=True)
    b = la.MYMALASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, seed=9, moments=True)
    a.set_state(img)
    b.set_state(img)
    a.step(10)
    s = np.zeros(shape)
    for _ in range(10):
        b.step(1)
        s += b.get_state().cpu().numpy().astype(np.float64).sum(axis=0)
    np.testing.assert_array_equal(a.get_state().cpu().numpy(), b.get_state().cpu().numpy())
    s1, s2, n = a.moments()
    assert n == 10 * C
    assert rel(s1.cpu().numpy(), s) < 1e-6
    rate = a.acceptance_rate().cpu().numpy()
    assert ((rate >= 0) & (rate <= 1)).all() and 0 < rate.mean() < 1, rate
    # the oracle, driven by the device's Philox normals and uniforms, reproduces the accept counts
    x0 = np.broadcast_to(img, (C,) + shape).copy()
    xo, acc_o, la_o = O.mymala_batched(x0, y, h, off, 1 / sigma ** 2, tau, gamma, prior, 10,
                                       lambda k: O.philox_normals(9, k, np.arange(C), *shape).astype(np.float64),
                                       lambda k: O.philox_uniforms(9, k, np.arange(C)))
    acc_d, _ = a.acceptance()
    us = np.stack([O.philox_uniforms(9, k, np.arange(C)) for k in range(10)])
    safe = (np.abs(np.log(us) - la_o) > 0.5).all(axis=0)
    assert (acc_d.cpu().numpy()[safe] == acc_o[safe]).all()
    assert rel(a.get_state().cpu().numpy()[safe], xo[safe]) < 1e-4
    # unadjusted sampler handles refuse the acceptance query
    u = la.MYULASampler(pf, pg, shape, n_chains=2, tau=tau, gamma=gamma)
    import torch
    with pytest.raises(la.LMCError):
        la._capi.check(la._dev.lib().lmc_sampler_get_acceptance(u._h, None, None, None))
    for smp in (a, b, u):
        smp.close()
