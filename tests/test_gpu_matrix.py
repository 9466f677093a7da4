"""A matrix of configurations -- image widths that select every step kernel (tiled / split / rows / block / point / pipe / strips), data terms (blur 5 and
7 taps, identity, mask), priors (TV fixed count and with upstream's early exit, l2, l1, Haar-l1, a closed form of prox.py) and the non-log-concave terms
of L2_ncvx_tv (MC-TV, ME-TV with its rtol) -- each run for three MYULA iterations on three chains with injected noise against the CPU checker's
class-based loop.  The single-feature suites test each kernel in depth; this one looks for holes BETWEEN features (round 3 found one by running the driver
end to end: the ULPDA sampler without the early-exit state of the ME-TV prox)."""
import itertools

import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu

SIG = 0.75
GAM, TAU = SIG ** 2, 0.2 * SIG ** 2


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


SHAPES = [(16, 64), (24, 136), (16, 264), (16, 512), (8, 528)]
DATA = ["blur5", "blur7", "identity", "mask"]
PRIORS = ["tv", "tv_rtol", "l2", "l1", "haar", "laplace"]
NCVX = ["none", "mc", "me"]


def build(la, shape, data, prior, ncvx, rng):
    n = shape[0] * shape[1]
    img = np.zeros(shape); img[3:shape[0] - 4, shape[1] // 8:shape[1] // 2] = 150.0
    img += np.linspace(0, 30, shape[1])[None, :]
    if data.startswith("blur"):
        k = int(data[4:])
        h = np.ones((k, k)) / k ** 2
        y = O.blur(img, np.ones((5, 5)) / 25, (2, 2)) + rng.normal(0, SIG, shape)
        Op, oOp = la.Convolve2D(shape, h, offset=(k // 2, k // 2)), O.Convolve2D(shape, h, (k // 2, k // 2))
    elif data == "identity":
        y = img + rng.normal(0, SIG, shape)
        Op, oOp = None, None
    else:
        m = (rng.uniform(size=shape) < 0.5).astype(np.float64)
        y = m * (img + rng.normal(0, SIG, shape))
        Op, oOp = la.Diagonal(m, dims=shape), O.Diagonal(m)
    if ncvx == "none":
        f, of = la.L2(Op=Op, b=y.ravel(), sigma=1 / SIG ** 2, dims=shape), O.L2(Op=oOp, b=y.ravel(), sigma=1 / SIG ** 2)
    else:
        kw = dict(dims=shape, b=y.ravel(), sigma=1 / SIG ** 2, lamda=0.3, gamma=15.0, isotropic=True, niter=20)
        if Op is None:
            Op, oOp = la.Identity(n), O.Identity(n)
        if ncvx == "mc":
            f, of = la.L2_ncvx_tv(Op=Op, Op2=la.Gradient(shape), **kw), O.L2NcvxTV(Op=oOp, Op2=O.Gradient(shape), **kw)
        else:
            f, of = la.L2_ncvx_tv(Op=Op, rtol=1e-4, **kw), O.L2NcvxTV(Op=oOp, tv_kwargs={"rtol": 1e-4}, **kw)
    if prior == "tv":
        g, og = la.TV(shape, sigma=0.3, niter=10), O.TV(shape, sigma=0.3, niter=10)
    elif prior == "tv_rtol":
        g, og = la.TV(shape, sigma=0.3, niter=10, rtol=1e-4), O.TV(shape, sigma=0.3, niter=10, rtol=1e-4)
    elif prior == "l2":
        g, og = la.L2(sigma=0.05, dims=shape), O.L2(sigma=0.05)
    elif prior == "l1":
        g, og = la.L1(sigma=0.8), O.L1(sigma=0.8)
    elif prior == "haar":
        g, og = la.WaveletL1(shape, sigma=0.3), O.WaveletL1(shape, sigma=0.3)
    else:
        class _Lap:
            def prox(self, x, t): return O.prox_laplace(x, t * 1.5)
        g, og = la.Laplace(1.5), _Lap()
    return img, f, of, g, og


@pytest.mark.parametrize("shape", SHAPES)
def test_every_combination_of_data_term_prior_and_nonconvex_term(la, shape):
    rng = np.random.default_rng(shape[1])
    C_, nit = 3, 3
    bad = []
    for data, prior, ncvx in itertools.product(DATA, PRIORS, NCVX):
        if prior == "haar" and (shape[0] % 8 or shape[1] % 8):
            continue
        img, f, of, g, og = build(la, shape, data, prior, ncvx, rng)
        x0 = img[None] + rng.normal(0, [[[3.0]], [[10.0]], [[25.0]]], (C_,) + shape)
        noise = rng.standard_normal((nit, C_) + shape)
        try:
            smp = la.MYULASampler(f, g, shape, n_chains=C_, tau=TAU, gamma=GAM, noise="injected")
        except NotImplementedError:
            continue
        smp.set_state(x0)
        smp.step(nit, noise=noise)
        got = smp.get_state().cpu().numpy()
        name = smp.kernel_name
        smp.close()
        ref = np.stack([O.myula(of, og, x0[c].ravel(), TAU, GAM, niter=nit, noise=[noise[i, c].ravel() for i in range(nit)])[-1].reshape(shape) for c in range(C_)])
        e = rel(got, ref)
        if not (e < 5e-5):
            bad.append((data, prior, ncvx, name, e))
        if prior in ("l2", "l1", "laplace"):          # ... and with an array-valued epsg (one weight per pixel: lmc_problem.prox_scale)
            eps = rng.uniform(0.3, 2.5, shape)
            smp = la.MYULASampler(f, g, shape, n_chains=C_, tau=TAU, gamma=GAM, epsg=eps, noise="injected")
            smp.set_state(x0)
            smp.step(nit, noise=noise)
            got = smp.get_state().cpu().numpy()
            name = smp.kernel_name
            smp.close()
            ref = np.stack([O.myula(of, og, x0[c].ravel(), TAU, GAM, epsg=eps.ravel(), niter=nit, noise=[noise[i, c].ravel() for i in range(nit)])[-1].reshape(shape)
                            for c in range(C_)])
            e = rel(got, ref)
            if not (e < 5e-5):
                bad.append((data, prior + "+epsg[]", ncvx, name, e))
    assert not bad, "\n".join(f"{d} {p} {n} {k} {e:.2e}" for d, p, n, k, e in bad)


@pytest.mark.parametrize("shape", [(16, 64), (24, 136), (16, 264), (8, 528)])
def test_ulpda_every_data_term_nonconvex_term_and_dual_prox(la, shape):
    """The same for ULPDA (algs.py:295-474): data term (blur 5 / 7 taps, identity, mask) x non-log-concave term (none, MC-TV, ME-TV with its rtol) x dual prox
    (L21 = isotropic TV, L1 = anisotropic), both orders of the primal and dual step, three chains with injected noise against the checker's loop (whose
    implicit step is 50 CG iterations: 2e-4)."""
    rng = np.random.default_rng(shape[1] + 1)
    C_, nit = 2, 3
    n = shape[0] * shape[1]
    tau0 = 0.95 * SIG ** 2
    mu0 = 0.99 / (tau0 * 8)
    bad = []
    for data, ncvx, iso, gfirst in itertools.product(DATA, NCVX, (True, False), (False, True)):
        if gfirst and (ncvx != "none" or not iso):
            continue                                   # the other order: once per data term is enough
        img, f, of, _, _ = build(la, shape, data, "l2", ncvx, rng)
        g, og = (la.L21(ndim=2, sigma=0.3), O.L21(ndim=2, sigma=0.3)) if iso else (la.L1(sigma=0.3), O.L1(sigma=0.3))
        x0 = img[None] + rng.normal(0, [[[3.0]], [[12.0]]], (C_,) + shape)
        noise = rng.standard_normal((nit, C_) + shape)
        try:
            smp = la.ULPDASampler(f, g, la.Gradient(shape), shape, n_chains=C_, tau=tau0, mu=mu0, theta=1.0, gfirst=gfirst, noise="injected")
        except NotImplementedError:
            continue
        smp.set_state(x0)
        smp.step(nit, noise=noise)
        got = smp.get_state().cpu().numpy()
        smp.close()
        ref = np.stack([O.ulpda(of, og, O.Gradient(shape), x0[c].ravel(), tau0, mu0, theta=1.0, niter=nit, gfirst=gfirst,
                                noise=[noise[i, c].ravel() for i in range(nit)])[-1].reshape(shape) for c in range(C_)])
        e = rel(got, ref)
        if not (e < 2e-4):
            bad.append((data, ncvx, "L21" if iso else "L1", f"gfirst={gfirst}", e))
    assert not bad, "\n".join(f"{d} {p} {n_} {k} {e:.2e}" for d, p, n_, k, e in bad)


@pytest.mark.parametrize("shape", [(16, 64), (24, 136), (16, 264), (8, 528)])
def test_class_methods_value_grad_prox_over_the_same_matrix(la, shape):
    """The plugin protocol itself (what reference-style code calls: proxf(x), proxf.grad(x), proxf.prox(x, tau), proxg.prox(x, tau), proxg(x)) over the data
    terms x non-log-concave terms and over the priors, batches of two images, against the checker's classes."""
    rng = np.random.default_rng(shape[1] + 2)
    n = shape[0] * shape[1]
    bad = []
    for data, ncvx in itertools.product(DATA, NCVX):
        img, f, of, _, _ = build(la, shape, data, "l2", ncvx, rng)
        x = np.stack([img + rng.normal(0, s, shape) for s in (4.0, 20.0)]).reshape(2, n)
        try:
            val = np.atleast_1d(np.asarray(f(x)))
            ref = np.array([of(x[i]) for i in range(2)])
            if not np.allclose(val, ref, rtol=5e-5):
                bad.append((data, ncvx, "value", float(np.max(np.abs(val / ref - 1)))))
            e = rel(f.grad(x), np.stack([of.grad(x[i].copy()) for i in range(2)]))
            if not e < 5e-5:
                bad.append((data, ncvx, "grad", e))
            e = rel(f.prox(x, 0.5), np.stack([type(of).prox(of, x[i].copy(), 0.5) for i in range(2)]))
            if not e < 2e-4:
                bad.append((data, ncvx, "prox", e))
        except NotImplementedError:
            continue
    for prior in PRIORS:
        if prior == "haar" and (shape[0] % 8 or shape[1] % 8):
            continue
        img, _, _, g, og = build(la, shape, "identity", prior, "none", rng)
        x = np.stack([img + rng.normal(0, s, shape) for s in (4.0, 20.0)]).reshape(2, n)
        e = rel(g.prox(x, 0.4), np.stack([og.prox(x[i].copy(), 0.4) for i in range(2)]))
        if not e < 1e-5:
            bad.append(("-", prior, "prox", e))
        if prior in ("tv", "l2", "l1", "haar"):
            if prior in ("l2", "l1"):       # pyproximal's L1 / L2 take whatever they are given as ONE vector
                val, ref = np.atleast_1d(np.asarray(g(x))), np.array([og(x.ravel())])
            else:
                val, ref = np.atleast_1d(np.asarray(g(x))), np.array([og(x[i]) for i in range(2)])
            if not np.allclose(val, ref, rtol=5e-5):
                bad.append(("-", prior, "value", float(np.max(np.abs(val / ref - 1)))))
    assert not bad, "\n".join(f"{a} {b} {c} {e:.2e}" for a, b, c, e in bad)


@pytest.mark.parametrize("shape", [(24, 136), (16, 264), (16, 512)])
def test_a_forced_kernel_variant_either_computes_the_update_or_refuses(la, shape):
    """`MYULASampler(..., variant=...)` pins the step kernel: a variant that does not cover the configuration must say so (LMC_E_UNSUPPORTED), never run
    something else or a different update -- every variant x data term x prior x non-convex term against the checker."""
    rng = np.random.default_rng(shape[1] + 3)
    C_, nit = 2, 2
    bad, ran = [], 0
    for variant in ("tile", "split", "point", "block", "rows", "pipe"):
        for data, prior, ncvx in itertools.product(["blur5", "blur7", "mask"], PRIORS, ["none", "mc"]):
            if prior == "haar" and (shape[0] % 8 or shape[1] % 8):
                continue
            img, f, of, g, og = build(la, shape, data, prior, ncvx, rng)
            x0 = img[None] + rng.normal(0, [[[3.0]], [[15.0]]], (C_,) + shape)
            noise = rng.standard_normal((nit, C_) + shape)
            smp = la.MYULASampler(f, g, shape, n_chains=C_, tau=TAU, gamma=GAM, noise="injected", variant=variant)
            smp.set_state(x0)
            try:
                smp.step(nit, noise=noise)
            except la.LMCError as err:
                assert err.code == -2, (variant, data, prior, ncvx, err)
                smp.close()
                continue
            got = smp.get_state().cpu().numpy()
            name = smp.kernel_name
            smp.close()
            ran += 1
            # (the per-chain early exit is a path of its own -- the RT pipeline, or the pass-by-pass prox in front of the pinned kernel: `variant` pins the fixed-count kernels)
            if prior != "tv_rtol" and variant not in name:
                bad.append((variant, data, prior, ncvx, name, -1.0))
            ref = np.stack([O.myula(of, og, x0[c].ravel(), TAU, GAM, niter=nit, noise=[noise[i, c].ravel() for i in range(nit)])[-1].reshape(shape) for c in range(C_)])
            e = rel(got, ref)
            if not (e < 5e-5):
                bad.append((variant, data, prior, ncvx, name, e))
    assert ran > 40 and not bad, "\n".join(f"{v} {d} {p} {n} {k} {e:.2e}" for v, d, p, n, k, e in bad)


@pytest.mark.parametrize("shape,C_", [((24, 136), 3), ((16, 512), 5), ((64, 512), 40)])
def test_posterior_moments_over_launch_groupings_and_kernels(la, shape, C_):
    """lmc_sampler_moments after ONE call of n iterations (which the library may run as pair / four-iteration launches, with the reductions on a side stream)
    against the sums formed from the states of n single-iteration calls with the same Philox seed: burn-in and thinning, every prior family (so every
    kernel family), both overlap policies."""
    rng = np.random.default_rng(shape[1] + 4 + C_)
    nit, burn, thin = 7, 2, 2
    bad = []
    for data, prior, ncvx in [("blur5", "tv", "none"), ("blur5", "tv_rtol", "none"), ("blur5", "l2", "none"), ("blur7", "l1", "none"), ("mask", "haar", "none"),
                              ("mask", "haar", "mc"), ("identity", "laplace", "none"), ("blur5", "tv", "mc")]:
        img, f, of, g, og = build(la, shape, data, prior, ncvx, rng)
        x0 = img[None] + rng.normal(0, 10.0, (C_,) + shape)
        ref_s1, ref_s2, cnt = np.zeros(shape), np.zeros(shape), 0
        one = la.MYULASampler(f, g, shape, n_chains=C_, tau=TAU, gamma=GAM, seed=11)
        one.set_state(x0)
        for it in range(nit):
            one.step(1)
            if it >= burn and (it - burn) % thin == 0:
                x = one.get_state().cpu().numpy().astype(np.float64)
                ref_s1 += x.sum(0); ref_s2 += (x * x).sum(0); cnt += C_
        final = one.get_state().cpu().numpy()
        one.close()
        for overlap, ipl in ((1, 0), (-1, 0), (1, 2), (-1, 2), (1, 1)):      # iterations_per_launch 2: pair / four-iteration launches wherever a kernel covers them
            smp = la.MYULASampler(f, g, shape, n_chains=C_, tau=TAU, gamma=GAM, seed=11, moments=True, burn_in=burn, thin=thin,
                                  policy={"moments_overlap": overlap, "iterations_per_launch": ipl})
            smp.set_state(x0)
            smp.step(nit)
            s1, s2, n = smp.moments()
            e1, e2 = rel(s1.cpu().numpy(), ref_s1), rel(s2.cpu().numpy(), ref_s2)
            ef = rel(smp.get_state().cpu().numpy(), final)
            name = smp.kernel_name
            smp.close()
            if n != cnt or not (e1 < 2e-6 and e2 < 4e-6 and ef < 1e-5):
                bad.append((data, prior, ncvx, overlap, ipl, name, n, cnt, e1, e2, ef))
    assert not bad, "\n".join(str(b) for b in bad)
