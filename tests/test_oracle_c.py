"""The C restatement of the oracle (oracle/lmc_oracle_c.c) against the numpy restatement that the reference's own
outputs pin (tests/golden/, test_oracle_golden.py): same float64 arithmetic in the same order => bit-identical."""
import numpy as np
import pytest

from oracle import lmc_oracle as O
from oracle import lmc_oracle_c as OC


def _img(rng, *shape):
    return rng.standard_normal(shape) * 20.0 + 100.0


@pytest.mark.parametrize("k,off", [(5, (2, 2)), (6, (3, 3)), (7, (3, 3)), (3, (0, 2))])
def test_blur_and_adjoint_bit_identical(k, off):
    rng = np.random.default_rng(k)
    x = _img(rng, 3, 19, 23)
    h = rng.random((k, k))
    assert np.array_equal(OC.blur(x, h, off), O.blur(x, h, off))
    assert np.array_equal(OC.blur(x, h, off, adjoint=True), O.blur_adjoint(x, h, off))


@pytest.mark.parametrize("K", [0, 1, 3, 10, 25])
def test_tv_prox_bit_identical(K):
    rng = np.random.default_rng(K)
    x = _img(rng, 2, 21, 18)
    a = OC.tv_prox_fgp(x, 4.0, K, threads=2)
    b = O.tv_prox_fgp(x, 4.0, K)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["none", "l2", "l1", "tv"])
@pytest.mark.parametrize("data", ["blur5", "blur6", "mask", "identity"])
def test_myula_step_bit_identical(kind, data):
    rng = np.random.default_rng(11)
    Hh, W, Cn = 24, 28, 5
    sigma = 0.75
    x = _img(rng, Cn, Hh, W)
    y = _img(rng, Hh, W)
    xi = rng.standard_normal(x.shape)
    h = off = mask = None
    if data == "blur5":
        h, off = np.ones((5, 5)) / 25.0, (2, 2)
    elif data == "blur6":
        h, off = np.ones((6, 6)) / 36.0, (3, 3)
    elif data == "mask":
        mask = (rng.random((Hh, W)) < 0.6).astype(np.float64)
    prior = {"kind": kind, "sigma": 0.3, "niter": 10, "t": sigma ** 2}
    args = (y, h, off, 1 / sigma ** 2, 0.2 * sigma ** 2, sigma ** 2, prior, xi)
    a = OC.myula_step(x, *args, mask=mask, threads=3)
    b = O.myula_step(x, *args, mask=mask)
    assert np.array_equal(a, b)


def test_trajectory_and_thread_count_independent():
    rng = np.random.default_rng(5)
    Hh, W, Cn = 16, 16, 6
    sigma = 0.75
    y = _img(rng, Hh, W)
    h = np.ones((5, 5)) / 25.0
    prior = {"kind": "tv", "sigma": 0.3, "niter": 10, "t": sigma ** 2}
    xa = np.zeros((Cn, Hh, W))
    xb = xa.copy()
    xc = xa.copy()
    for k in range(20):
        xi = rng.standard_normal(xa.shape)
        xa = OC.myula_step(xa, y, h, (2, 2), 1 / sigma ** 2, 0.2 * sigma ** 2, sigma ** 2, prior, xi, threads=1)
        xb = OC.myula_step(xb, y, h, (2, 2), 1 / sigma ** 2, 0.2 * sigma ** 2, sigma ** 2, prior, xi, threads=4)
        xc = O.myula_step(xc, y, h, (2, 2), 1 / sigma ** 2, 0.2 * sigma ** 2, sigma ** 2, prior, xi)
    assert np.array_equal(xa, xb)
    assert np.array_equal(xa, xc)


def test_rejects_bad_arguments():
    with pytest.raises(RuntimeError):
        OC.lib()                                    # loads
        x = np.zeros((1, 4, 4))
        cfg = OC._Cfg()
        cfg.H, cfg.W = 0, 4
        rc = OC.lib().lmc_oc_myula_step(OC.C.byref(cfg), OC._p(x), OC._p(x), OC._p(x.copy()), 1, 1)
        if rc != 0:
            raise RuntimeError(rc)


def test_tv_early_exit_is_bit_identical_and_leaves_in_the_same_pass():
    """The rtol branch (pyproximal.TV's per-image early exit, in force at prox_lmc_deconv.py:122 and algs.py:169): the C twin forms the objective
    with numpy's pairwise summation, so it leaves in the very pass lmc_oracle.tv_prox_fgp leaves in and returns the same bits."""
    rng = np.random.default_rng(0)
    seen = set()
    for trial in range(8):
        Hh, W = [(24, 136), (33, 47), (64, 64), (17, 200)][trial % 4]
        img = np.zeros((Hh, W))
        img[5:15, 10:30] = 100 + 50 * rng.random()
        img += rng.normal(0, [0.5, 3, 10][trial % 3], (Hh, W))
        for rtol in (1e-4, 1e-3, 1e-2):
            a = O.tv_prox_fgp(img, 0.17, 10, rtol=rtol)
            b, ps = OC.tv_prox_fgp(img, 0.17, 10, rtol=rtol, return_passes=True)
            assert np.array_equal(a, b), (trial, rtol)
            seen.add(int(ps[0]))
            # the pass it left in: the iterate after that many dual updates
            assert np.array_equal(a, O.tv_prox_fgp(img, 0.17, int(ps[0]))), (trial, rtol, ps)
    assert len(seen) >= 3, seen
    # inside the MYULA step, with the pass counts reported per chain
    y = _img(rng, 24, 40)
    x = np.stack([np.zeros((24, 40)), y + rng.normal(0, 3, y.shape), y + rng.normal(0, 30, y.shape)])
    xi = rng.standard_normal(x.shape)
    prior = {"kind": "tv", "sigma": 0.3, "niter": 10, "t": 0.5625, "rtol": 1e-4}
    ps = np.zeros(3, dtype=np.int32)
    got = OC.myula_step(x, y, np.ones((5, 5)) / 25, (2, 2), 1 / 0.5625, 0.1125, 0.5625, prior, xi, threads=2, passes=ps)
    want = O.myula_step(x, y, np.ones((5, 5)) / 25, (2, 2), 1 / 0.5625, 0.1125, 0.5625, prior, xi)
    assert np.array_equal(got, want)
    assert ps[0] == 10 and 1 <= ps[1] <= 10
