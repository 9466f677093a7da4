"""GPU parity of the barrier-free row-streaming step kernel (separable blur + closed-form prior; BASELINE config
"256x256 deblur + l2 prior") against the oracle step and against the LDS-tiled kernel, through the C ABI."""
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


def problem(shape, k, rng, gaussian=False):
    img = np.zeros(shape)
    img[shape[0] // 5:shape[0] // 2 + 1, shape[1] // 4:shape[1] // 2 + 2] = 190.0
    img += np.linspace(0, 30, shape[1])[None, :]
    if gaussian:
        t = np.exp(-0.5 * (np.arange(k) - k // 2) ** 2)
        h = np.outer(t, t) / np.sum(np.outer(t, t))
    else:
        h = np.ones((k, k)) / (k * k)
    off = (k // 2, k // 2)
    y = O.blur(img, h, off) + rng.normal(0, 0.75, shape)
    return img, h, off, y


# widths: 4 px/lane (<= 256) and 8 px/lane (<= 512), partially filled last lanes, heights below the pipeline depth,
# not multiples of 8, several bands per chain (H > 32)
@pytest.mark.parametrize("shape,k", [((32, 32), 5), ((3, 8), 5), ((20, 24), 6), ((70, 100), 7), ((100, 64), 5), ((256, 256), 5),
                                     ((45, 260), 5), ((130, 512), 5), ((9, 36), 3), ((41, 300), 3),
                                     ((37, 512), 7), ((50, 384), 6), ((24, 264), 7)])     # 7 taps at 8 pixels per lane
@pytest.mark.parametrize("prior", ["l2", "l1", "none"])
def test_rows_kernel_matches_oracle_step(la, shape, k, prior):
    sigma, tau_reg = 0.75, 0.3
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    rng = np.random.default_rng(13)
    C, nit = 2, 3
    img, h, off, y = problem(shape, k, rng, gaussian=(k == 5 and shape[0] % 2 == 0))
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / sigma ** 2)
    pg = {"l1": la.L1(sigma=tau_reg), "l2": la.L2(sigma=0.05), "none": None}[prior]
    op = {"kind": prior, "sigma": 0.05 if prior == "l2" else tau_reg, "t": gamma}
    x0 = img[None] + rng.normal(0, 10, (C,) + shape)
    noise = rng.standard_normal((nit, C) + shape)
    la.set_step_variant("rows")
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected")
    smp.set_state(x0)
    x = x0.copy()
    for it in range(nit):
        smp.step(1, noise=noise[it:it + 1])
        x = O.myula_step(x, y, h, off, 1 / sigma ** 2, tau, gamma, op, noise[it])
        got = smp.get_state().cpu().numpy()
        assert rel(got, x) < 2e-6 * (it + 1), (it, rel(got, x))
    assert smp.kernel_name == "myula_step_rows_kernel"
    smp.close()
    la.set_step_variant("auto")


def test_rows_kernel_philox_matches_tile_and_is_the_default(la):
    """Philox noise (quad rows aligned to the bands), many chains, moments -- same trajectory as the tiled kernel."""
    rng = np.random.default_rng(4)
    for shape, k, C in [((256, 256), 5, 6), ((96, 512), 5, 3), ((64, 128), 7, 40)]:
        img, h, off, y = problem(shape, k, rng)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1 / 0.75 ** 2)
        outs = {}
        for v in ("tile", "rows", "auto"):
            la.set_step_variant(v)
            smp = la.MYULASampler(pf, la.L2(sigma=0.05), shape, n_chains=C, tau=0.1125, gamma=0.5625, seed=3, chain_offset=11)
            smp.set_state(img)
            smp.step(5)
            outs[v] = smp.get_state().cpu().numpy()
            if v == "auto":
                assert smp.kernel_name == "myula_step_rows_kernel"
            smp.close()
        assert rel(outs["rows"], outs["tile"]) < 2e-6, (shape, rel(outs["rows"], outs["tile"]))
        np.testing.assert_array_equal(outs["rows"], outs["auto"])
    la.set_step_variant("auto")


def test_rows_kernel_not_used_outside_its_domain(la):
    """Non-separable taps (or a TV prior): the dispatcher must pick another kernel.  W % 4 != 0 is INSIDE the domain since round 2
    (dword-aligned 16-byte accesses; tests/test_gpu_wide.py)."""
    rng = np.random.default_rng(5)
    shape = (24, 30)
    img, h, off, y = problem(shape, 5, rng)
    hn = rng.uniform(0.5, 1.5, (5, 5)); hn /= hn.sum()              # rank > 1
    pf = la.L2(Op=la.Convolve2D(shape, hn, offset=off), b=y, sigma=1.0)
    la.set_step_variant("rows")
    smp = la.MYULASampler(pf, la.L2(sigma=0.05), shape, n_chains=1, tau=0.1, gamma=0.5)
    with pytest.raises(la.LMCError):
        smp.step(1)
    smp.close()
    la.set_step_variant("auto")
    smp = la.MYULASampler(pf, la.L2(sigma=0.05), shape, n_chains=1, tau=0.1, gamma=0.5)
    smp.step(1)
    assert smp.kernel_name != "myula_step_rows_kernel"
    smp.close()
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=off), b=y, sigma=1.0)
    smp = la.MYULASampler(pf, la.L2(sigma=0.05), shape, n_chains=1, tau=0.1, gamma=0.5)
    smp.step(1)
    assert smp.kernel_name == "myula_step_rows_kernel"
    smp.close()
