"""The re-written experiment driver (prox_lmc_deconv.py sampling branch) end to end on a small image: all nine models,
both samplers, single-chain reference form and many-chain form."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_nine_models_both_samplers_small():
    import torch
    assert torch.cuda.is_available()
    from lmc_atomi_amd.deconv import prox_lmc_deconv, synthetic_image
    img = synthetic_image(48, 64)
    for alg, N in (("MYULA", 60), ("ULPDA", 25)):
        res = prox_lmc_deconv(N=N, image=img, alg=alg, seed=0, n_chains=8, burn_in=N // 2, thin=1, verbose=False,
                              niter_l2=20)
        y = res["_observation"]
        snr_y = 20 * np.log10(np.linalg.norm(img) / np.linalg.norm(y - img))
        for m in ("M1", "M2", "M3", "M4", "M5", "M6", "M7", "M8", "M9"):
            r = res[m]
            assert r["mean"].shape == img.shape and np.all(np.isfinite(r["mean"]))
            assert r["mse"] > 0 and np.isfinite(r["snr"]) and np.isfinite(r["psnr"])
        # the matched model (5x5 blur, the one that generated y) must denoise/deblur: better SNR than the observation
        assert res["M1"]["snr"] > snr_y, (alg, res["M1"]["snr"], snr_y)
    # reference form: one chain, every iterate on the host, mean over iterates (prox_lmc_deconv.py:474)
    res1 = prox_lmc_deconv(N=12, image=img, alg="MYULA", seed=0, models=["M1", "M2"], verbose=False)
    assert set(k for k in res1 if not k.startswith("_")) == {"M1", "M2"}
    assert res1["M1"]["mean"].shape == img.shape


@pytest.mark.parametrize("alg", ["MYULA", "ULPDA", "MYMALA"])
def test_nine_models_at_a_width_the_fused_kernels_cover(alg):
    """The same driver at 264 columns, where the full-width pipeline and the device-side early exit of the TV proxes run (the 64-column case above takes the
    general kernels and the pass-by-pass exit): every model with the driver's defaults (rtol = 1e-4, niter_l2 = 50), as `python -m lmc_atomi_amd.deconv` runs them."""
    import torch
    assert torch.cuda.is_available()
    from lmc_atomi_amd.deconv import prox_lmc_deconv, synthetic_image
    img = synthetic_image(40, 264)
    res = prox_lmc_deconv(N=12, image=img, alg=alg, seed=0, n_chains=3, burn_in=4, thin=1, verbose=False)
    for m in ("M1", "M2", "M3", "M4", "M5", "M6", "M7", "M8", "M9"):
        assert res[m]["mean"].shape == img.shape and np.all(np.isfinite(res[m]["mean"])), (alg, m)


def test_driver_mymala_branch():
    """--alg MYMALA: the Metropolis-adjusted sampler through the same driver (three models incl. both L2_ncvx_tv terms)."""
    import torch
    assert torch.cuda.is_available()
    from lmc_atomi_amd.deconv import prox_lmc_deconv, synthetic_image
    img = synthetic_image(32, 48)
    res = prox_lmc_deconv(N=20, image=img, alg="MYMALA", seed=0, n_chains=4, models=["M1", "M2", "M3"], verbose=False, niter_l2=10)
    for m in ("M1", "M2", "M3"):
        assert np.all(np.isfinite(res[m]["mean"])) and res[m]["mean"].shape == img.shape


def test_reference_form_is_the_same_with_and_without_a_callback():
    """A callback must not change the iterates (one chain, every iterate returned, both noise sources)."""
    import lmc_atomi_amd as la
    rng = np.random.default_rng(4)
    shape = (16, 24)
    y = rng.uniform(50, 200, shape)
    pf = la.L2(Op=la.Convolve2D(shape, np.ones((5, 5)) / 25.0, offset=(2, 2)), b=y.ravel(), sigma=1 / 0.5625)
    pg = la.TV(shape, sigma=0.3, niter=10)
    seen = []
    for r in ("philox", "pcg64"):
        a = la.MoreauYosidaUnadjustedLangevin(pf, pg, np.zeros(shape[0] * shape[1]), tau=0.1125, gamma=0.5625, niter=9, seed=3, rng=r)
        b = la.MoreauYosidaUnadjustedLangevin(pf, pg, np.zeros(shape[0] * shape[1]), tau=0.1125, gamma=0.5625, niter=9, seed=3, rng=r,
                                              callback=lambda x: seen.append(float(x[0])))
        assert a.shape == (9, shape[0] * shape[1]) and np.array_equal(a, b)
    assert len(seen) == 18
