"""Array-valued ``epsg`` of the reference's MYULA (algs.py:509 "float or np.ndarray", :539-542, :569: the prox parameter ``epsg * gamma`` handed to
``proxg.prox`` is an array that the closed-form proxes broadcast) through the HIP path (``lmc_problem.prox_scale``): per pixel of the flattened image
and per right-hand side (= per chain) against trajectories of the reference's own loop (tests/golden/epsg_array.npz), both at once against the checker."""
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


def _priors(la):
    return (("l1", la.L1(sigma=2.0)), ("l2", la.L2(sigma=0.05)), ("laplace", la.Laplace(1.5)))


def test_per_pixel_epsg_matches_the_reference_loop(la, golden):
    g = golden("epsg_array.npz")
    sigma, tau, gam = (float(v) for v in g["params"])
    ny, nx, k, seed, nit = (int(v) for v in g["meta"])
    f = la.L2(Op=la.Convolve2D((ny, nx), g["h"], offset=(k // 2, k // 2)), b=g["y"].ravel(), sigma=1 / sigma ** 2)
    for name, pr in _priors(la):
        for e in (g["epsg_pixel"], g["epsg_pixel"].reshape(ny, nx)):
            xs = la.MoreauYosidaUnadjustedLangevin(f, pr, np.zeros(ny * nx), tau=tau, gamma=gam, epsg=e, niter=nit, seed=seed, rng="pcg64")
            assert rel(xs, g[f"pixel_{name}"]) < 1e-5, (name, rel(xs, g[f"pixel_{name}"]))


def test_per_chain_epsg_matches_the_reference_loop_with_several_right_hand_sides(la, golden):
    """x of shape (n, nrhs) with epsg of shape (nrhs,) in the reference = one weight per chain here; the noise of the reference's loop injected column
    by column."""
    g = golden("epsg_array.npz")
    sigma, tau, gam = (float(v) for v in g["params"])
    ny, nx, k, seed, nit = (int(v) for v in g["meta"])
    e = g["epsg_rhs"]
    f = la.L2(b=g["y"].ravel(), sigma=1 / sigma ** 2)
    for name, pr in _priors(la)[:2]:
        rng = np.random.default_rng(seed)
        smp = la.MYULASampler(f, pr, (ny, nx), n_chains=e.size, tau=tau, gamma=gam, epsg=e, noise="injected")
        smp.set_state(np.zeros((e.size, ny, nx)))
        for it in range(nit):
            xi = rng.standard_normal((ny * nx, e.size))
            smp.step(1, noise=np.ascontiguousarray(xi.T).reshape(1, e.size, ny, nx))
            got = smp.get_state().cpu().numpy().reshape(e.size, -1).T
            assert rel(got, g[f"rhs_{name}"][it]) < 1e-5, (name, it)
        smp.close()


def test_per_chain_and_pixel_epsg_with_moments_against_the_checker(la):
    rng = np.random.default_rng(5)
    shape, C_, nit = (24, 136), 4, 4
    img = np.zeros(shape); img[5:17, 20:100] = 120.0
    h = np.ones((5, 5)) / 25
    sig = 0.75
    y = O.blur(img, h, (2, 2)) + rng.normal(0, sig, shape)
    e = rng.uniform(0.3, 2.5, (C_,) + shape)
    x0 = img[None] + rng.normal(0, 6, (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    tau, gam = 0.2 * sig ** 2, sig ** 2
    of = O.L2(Op=O.Convolve2D(shape, h, (2, 2)), b=y.ravel(), sigma=1 / sig ** 2)
    for pr, opr in ((la.L1(sigma=1.2), O.L1(sigma=1.2)), (la.Huber(0.5, 0.4), None)):
        if opr is None:
            class opr_cls:
                def prox(self, x, t): return O.prox_huber(x, 0.5, t * 0.4)
            opr = opr_cls()
        smp = la.MYULASampler(la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y.ravel(), sigma=1 / sig ** 2), pr, shape, n_chains=C_, tau=tau, gamma=gam,
                              epsg=e, noise="injected", moments=True)
        smp.set_state(x0)
        smp.step(nit, noise=noise)
        got = smp.get_state().cpu().numpy()
        traj = [O.myula(of, opr, x0[c].ravel(), tau, gam, epsg=e[c].ravel(), niter=nit, noise=[noise[i, c].ravel() for i in range(nit)]) for c in range(C_)]
        ref = np.stack([t[-1].reshape(shape) for t in traj])
        assert rel(got, ref) < 1e-5, rel(got, ref)
        s1, s2, cnt = smp.moments()
        assert cnt == nit * C_
        assert rel(s1.cpu().numpy().ravel(), np.sum([t.sum(0) for t in traj], axis=0)) < 1e-5
        smp.close()


def test_array_epsg_is_refused_where_it_is_not_built(la):
    shape = (16, 136)
    f = la.L2(b=np.zeros(shape).ravel(), sigma=1.0)
    with pytest.raises(Exception, match="closed-form priors"):
        la.MYULASampler(f, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=2, tau=0.1, gamma=0.5, epsg=np.array([1.0, 2.0]))
    with pytest.raises(ValueError, match="matches neither"):
        la.MYULASampler(f, la.L1(sigma=1.0), shape, n_chains=2, tau=0.1, gamma=0.5, epsg=np.ones(7))
    with pytest.raises(Exception, match="scalar epsg"):
        la.MYMALASampler(f, la.L1(sigma=1.0), shape, n_chains=2, tau=0.1, gamma=0.5, epsg=np.array([1.0, 2.0]))
