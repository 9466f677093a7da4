"""LMC_MOMENTS_OVERLAP=1 (moment reductions on a side stream under the next step kernel, paced background kernel): same
states bit for bit, same accumulators to fp64 rounding (the atomics commute, their order does not) as the in-order path."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import lmc_atomi_amd as la
out = {}
for tag, shape, C, prior in [("pipe", (48, 264), 5, "tv"), ("rows", (40, 64), 7, "l2"), ("odd", (9, 7), 3, "tv")]:
    rng = np.random.default_rng(3)
    H, W = shape
    y = rng.uniform(50, 200, shape)
    h = np.ones((5, 5)) / 25.0
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / 0.5625)
    pg = la.TV(shape, sigma=0.3, niter=10) if prior == "tv" else la.L2(sigma=0.05)
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=0.1125, gamma=0.5625, seed=5, moments=True, burn_in=2, thin=2)
    smp.set_state(y.astype(np.float32))
    smp.step(4)                 # crosses the burn-in inside one call
    smp.step(1)
    smp.step(9)
    s1, s2, cnt = smp.moments()
    out[tag + "_s1"], out[tag + "_s2"], out[tag + "_cnt"] = s1.cpu().numpy(), s2.cpu().numpy(), cnt
    out[tag + "_x"] = smp.get_state().cpu().numpy()
    smp.reset_moments()
    smp.step(3)
    s1, s2, cnt = smp.moments()
    out[tag + "_s1b"], out[tag + "_cntb"] = s1.cpu().numpy(), cnt
    smp.close()
np.savez(sys.argv[2], **out)
'''


def _run(tmp_path, overlap, wgs="128"):
    out = str(tmp_path / f"o{overlap}_{wgs}.npz")
    env = dict(os.environ, LMC_MOMENTS_OVERLAP=str(overlap), LMC_MOMENTS_BG_WGS=wgs)
    subprocess.run([sys.executable, "-c", WORKER, ROOT, out], env=env, check=True, timeout=600)
    return np.load(out)


def test_overlapped_moments_equal_in_order_moments(tmp_path):
    ref = _run(tmp_path, 0)
    for wgs in ("128", "3", "0"):
        got = _run(tmp_path, 1, wgs)
        for k in ref.files:
            if k.endswith("_x"):
                assert np.array_equal(got[k], ref[k]), k
            elif "cnt" in k:
                assert int(got[k]) == int(ref[k]) and int(ref[k]) > 0, k
            else:
                np.testing.assert_allclose(got[k], ref[k], rtol=1e-12, atol=0, err_msg=k)
