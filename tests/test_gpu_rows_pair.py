"""Two MYULA iterations per launch (csrc/lmc_step_rows_pair.hip: wave pairs with an LDS hand-off; blur + closed-form prior) against the
single-iteration launches of the row-streaming kernel -- states and moment accumulators, odd iteration counts, burn-in / thinning, several
bands per chain, both lane widths -- and against the CPU checker driven by its own Philox field.  Same arithmetic and the same noise; the
running window sums start at band boundaries, so results agree to fp32 rounding (a different band height changes the last bits of the
single-iteration kernel too), not bit for bit."""
import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu
SIGMA = 0.75
GAMMA, TAU = SIGMA ** 2, 0.2 * SIGMA ** 2


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


def problem(shape, seed=0):
    rng = np.random.default_rng(seed)
    img = np.zeros(shape)
    img[shape[0] // 5:shape[0] // 2, shape[1] // 6:2 * shape[1] // 3] = 170.0
    img += np.linspace(0, 25, shape[1])[None, :]
    h = np.ones((5, 5)) / 25
    y = O.blur(img, h, (2, 2)) + rng.normal(0, SIGMA, shape)
    return img, h, y


PRIORS = {"l2": (lambda la: la.L2(sigma=0.02), {"kind": "l2", "sigma": 0.02, "t": GAMMA}),
          "l1": (lambda la: la.L1(sigma=0.8), {"kind": "l1", "sigma": 0.8, "t": GAMMA}),
          "none": (lambda la: None, {"kind": "none"})}


@pytest.mark.parametrize("shape,C,band,nit", [((72, 128), 3, 32, 5), ((150, 512), 2, 0, 4), ((100, 264), 5, 40, 6), ((37, 36), 2, 0, 3),
                                              ((300, 256), 1, 0, 7), ((90, 512), 2, 32, 2)])
@pytest.mark.parametrize("prior", ["l2", "l1", "none"])
def test_pairs_equal_single_launches(la, shape, C, band, nit, prior, monkeypatch):
    img, h, y = problem(shape, shape[1])
    monkeypatch.setenv("LMC_MOMENTS_OVERLAP", "0")
    if band:
        monkeypatch.setenv("LMC_PAIR_BAND", str(band))
    outs = {}
    for mode in ("0", "2"):
        monkeypatch.setenv("LMC_ROWS_PAIR", mode)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
        smp = la.MYULASampler(pf, PRIORS[prior][0](la), shape, n_chains=C, tau=TAU, gamma=GAMMA, seed=13, chain_offset=4, moments=True, burn_in=1,
                              thin=2 if nit > 4 else 1)
        smp.set_state(img)
        smp.step(nit)
        name = smp.kernel_name
        m1, m2, cnt = smp.moments()
        outs[mode] = (smp.get_state().cpu().numpy(), m1.cpu().numpy(), m2.cpu().numpy(), cnt, name)
        smp.close()
    assert "pair" in outs["2"][4] or nit % 2 == 1, outs["2"][4]          # (an odd count ends with a single launch)
    assert "pair" not in outs["0"][4]
    assert rel(outs["2"][0], outs["0"][0]) < 1e-6 * nit, rel(outs["2"][0], outs["0"][0])
    assert np.abs(outs["2"][0] - outs["0"][0]).max() < 2e-4            # ulps of values ~ 200
    assert outs["2"][3] == outs["0"][3]                                 # the same iterates were kept
    assert rel(outs["2"][1], outs["0"][1]) < 1e-6 * nit and rel(outs["2"][2], outs["0"][2]) < 2e-6 * nit


def test_pairs_against_the_checker_with_its_philox_field(la, monkeypatch):
    monkeypatch.setenv("LMC_ROWS_PAIR", "2")
    shape, C, seed, off, nit = (130, 264), 3, 5, 9, 4
    img, h, y = problem(shape, 3)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
    smp = la.MYULASampler(pf, la.L2(sigma=0.02), shape, n_chains=C, tau=TAU, gamma=GAMMA, seed=seed, chain_offset=off)
    smp.step(nit)
    assert smp.kernel_name == "myula_step_rows_pair_kernel"
    got = smp.get_state().cpu().numpy()
    ref = O.myula_batched(np.zeros((C,) + shape), y, h, (2, 2), 1 / SIGMA ** 2, TAU, GAMMA, {"kind": "l2", "sigma": 0.02, "t": GAMMA}, nit,
                          lambda i: O.philox_normals(seed, i, np.arange(off, off + C), *shape).astype(np.float64))
    assert rel(got, ref) < 1e-5, rel(got, ref)
    smp.close()


def test_pairs_are_the_default_only_where_they_pay_and_never_with_injected_noise(la, monkeypatch):
    monkeypatch.delenv("LMC_ROWS_PAIR", raising=False)
    shape = (64, 128)
    img, h, y = problem(shape, 1)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
    smp = la.MYULASampler(pf, la.L2(sigma=0.02), shape, n_chains=4, tau=TAU, gamma=GAMMA)
    smp.step(4)
    assert smp.kernel_name == "myula_step_rows_kernel"            # 4 x 64 rows: far below the 2^17 rows a launch of pairs needs
    smp.close()
    monkeypatch.setenv("LMC_ROWS_PAIR", "2")
    smp = la.MYULASampler(pf, la.L2(sigma=0.02), shape, n_chains=2, tau=TAU, gamma=GAMMA, noise="injected")
    smp.step(2, noise=np.zeros((2, 2) + shape))
    assert smp.kernel_name == "myula_step_rows_kernel"
    smp.close()
