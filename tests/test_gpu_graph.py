"""hipGraph replay of MYULA iterations (small configurations: BASELINE config 2): blocks of 8 iterations captured once, the moment
reduction of iteration k on a side branch under the step kernel of iteration k + 1, the Philox iteration word read from device
memory.  Must reproduce the plain launch sequence bit for bit (states) / to accumulation order (fp64 moment sums)."""
import os

import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu

SIGMA = 0.75
GAMMA, TAU = SIGMA ** 2, 0.2 * SIGMA ** 2


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available()
    import lmc_atomi_amd as la
    return la


def run(la, mode, pf, pg, shape, C, nits, burn_in=0, moments=True):
    os.environ["LMC_GRAPH"] = mode
    try:
        smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=TAU, gamma=GAMMA, seed=7, chain_offset=3, moments=moments, burn_in=burn_in)
        for n in nits:                     # several calls: graph blocks, plain remainders, and back
            smp.step(n)
        st = smp.get_state().cpu().numpy()
        mom = smp.moments() if moments else None
        it, name = smp.iteration, smp.kernel_name
        smp.close()
    finally:
        os.environ.pop("LMC_GRAPH", None)
    return st, mom, it, name


@pytest.mark.parametrize("kind,shape", [("rows", (64, 136)), ("rows", (48, 256)), ("block", (64, 136)), ("pipe", (40, 264)), ("pipe20", (24, 136))])
def test_graph_replay_equals_plain_launches(la, kind, shape):
    rng = np.random.default_rng(4)
    img = np.zeros(shape)
    img[8:30, 20:100] = 150.0
    img += np.linspace(0, 20, shape[1])[None, :]
    h = np.ones((5, 5)) / 25.0
    if kind == "block":
        m = (rng.uniform(size=shape) < 0.6).astype(np.float64)
        pf = la.L2(Op=la.Diagonal(m, dims=shape), b=m * img, sigma=1 / SIGMA ** 2, dims=shape)
        pg = la.WaveletL1(shape, sigma=0.3)
    else:
        y = O.blur(img, h, (2, 2)) + rng.normal(0, SIGMA, shape)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
        pg = la.L2(sigma=0.05) if kind == "rows" else la.TV(shape, sigma=0.3, niter=20 if kind == "pipe20" else 10)
    nits = (21, 3, 17)
    for burn in (0, 5):
        a_st, a_m, a_it, a_name = run(la, "0", pf, pg, shape, 6, nits, burn_in=burn)
        b_st, b_m, b_it, b_name = run(la, "1", pf, pg, shape, 6, nits, burn_in=burn)
        assert a_it == b_it == sum(nits) and a_name == b_name and kind.rstrip("20") in a_name
        np.testing.assert_array_equal(a_st, b_st)                     # same kernels, same Philox words: bit-identical states
        assert a_m[2] == b_m[2] == 6 * (sum(nits) - burn)
        for u, v in zip(a_m[:2], b_m[:2]):
            np.testing.assert_allclose(u.cpu().numpy(), v.cpu().numpy(), rtol=1e-12, atol=0)
    # without moments the graph is a plain chain of step kernels
    a_st, _, _, _ = run(la, "0", pf, pg, shape, 4, (19,), moments=False)
    b_st, _, _, _ = run(la, "1", pf, pg, shape, 4, (19,), moments=False)
    np.testing.assert_array_equal(a_st, b_st)
