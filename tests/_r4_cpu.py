"""CPU side of the statistical parity rung R4 (tests/test_gpu_r4.py): many independent PCG64 chains of the CPU checker, each one
the reference's single-chain recursion with its own seed (``default_rng(seed + c)``, one ``standard_normal(n)`` per iteration:
algs.py:561,565 / :431,433).  No torch import here: the ULPDA workers are spawned processes."""
import os
import sys
from concurrent.futures import ThreadPoolExecutor

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def truth(ny, nx, seed=1234):
    """Piecewise-constant blocks + a smooth ramp in [0, 255] (SURVEY 8(d) synthetic inputs)."""
    rng = np.random.default_rng(seed)
    img = np.zeros((ny, nx))
    for _ in range(12):
        a, b = sorted(rng.integers(0, ny, 2))
        c, d = sorted(rng.integers(0, nx, 2))
        img[a:b + 1, c:d + 1] = rng.uniform(20, 235)
    img += np.linspace(0, 20, nx)[None, :]
    return np.clip(img, 0, 255)


def myula_tv_chains(y, h, offset, sigma, tau_reg, K, tau, gamma, n_chains, n_iters, seed0, threads, lagged=False, rtol=0.0):
    """``n_chains`` MYULA chains (x0 = 0, blur data term, TV prior with K dual iterations) by the C twin of the checker
    (oracle/lmc_oracle_c.c, bit-identical to oracle/lmc_oracle.py) -- chain c draws its noise from ``default_rng(seed0 + c)``.
    ``rtol > 0``: the TV prox with upstream's per-image early exit -- 1e-4 is the reference AS CONFIGURED (prox_lmc_deconv.py:122 leaves
    pyproximal.TV's default in force).  Returns (sum over chains and iterations of x, of x^2, count, final states)."""
    from oracle import lmc_oracle_c as OC
    H, W = y.shape
    K_eff = K - 1 if lagged else K
    prior = {"kind": "tv", "sigma": tau_reg, "niter": K_eff, "t": gamma, "rtol": rtol} if K_eff > 0 else {"kind": "none"}
    rngs = [np.random.default_rng(seed0 + c) for c in range(n_chains)]
    xi = np.empty((n_chains, H, W))
    x = np.zeros((n_chains, H, W))
    s1 = np.zeros((H, W))
    s2 = np.zeros((H, W))
    with ThreadPoolExecutor(threads) as pool:
        for _ in range(n_iters):
            list(pool.map(lambda c: rngs[c].standard_normal(out=xi[c]), range(n_chains)))     # numpy releases the GIL while filling
            x = OC.myula_step(x, y, h, offset, 1.0 / sigma ** 2, tau, gamma, prior, xi, threads=threads)
            s1 += x.sum(axis=0)
            s2 += np.einsum("chw,chw->hw", x, x)
    return s1, s2, n_chains * n_iters, x


def _ulpda_chain(args):
    (y, h, offset, sigma, tau_reg, tau, mu, theta, gfirst, cg_niter, n_iters, seed) = args
    from oracle import lmc_oracle as O
    shape = y.shape
    Hop = O.Convolve2D(shape, h, offset)
    pf = O.L2(Op=Hop, b=y.ravel(), sigma=1.0 / sigma ** 2, niter=cg_niter, warm=True)
    pg = O.L21(ndim=2, sigma=tau_reg)
    xs = O.ulpda(pf, pg, O.Gradient(shape), np.zeros(y.size), tau, mu, theta=theta, niter=n_iters, seed=seed, gfirst=gfirst)
    return xs.sum(axis=0), (xs * xs).sum(axis=0), xs[-1]


def ulpda_chains(y, h, offset, sigma, tau_reg, tau, mu, theta, gfirst, cg_niter, n_chains, n_iters, seed0, workers):
    """``n_chains`` ULPDA chains, each ONE call of the checker's single-chain ``ulpda`` (the function pinned by the reference's own
    trajectories, tests/golden/algs.npz) with seed ``seed0 + c``, spread over spawned worker processes."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor
    jobs = [(y, h, offset, sigma, tau_reg, tau, mu, theta, gfirst, cg_niter, n_iters, seed0 + c) for c in range(n_chains)]
    s1 = np.zeros(y.size)
    s2 = np.zeros(y.size)
    last = []
    with ProcessPoolExecutor(workers, mp_context=mp.get_context("spawn")) as ex:
        for a, b, xl in ex.map(_ulpda_chain, jobs, chunksize=max(1, n_chains // (4 * workers))):
            s1 += a
            s2 += b
            last.append(xl)
    return s1.reshape(y.shape), s2.reshape(y.shape), n_chains * n_iters, np.array(last).reshape((n_chains,) + y.shape)
