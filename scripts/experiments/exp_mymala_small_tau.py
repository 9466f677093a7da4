"""MYMALA log acceptance ratios, device vs oracle, for small step sizes (debugging aid)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from oracle import lmc_oracle as O
import lmc_atomi_amd as la
sys.path.insert(0, "tests")
from test_gpu_mymala import build

for shape in [(20, 264), (128, 512)]:
    for ts in [0.2, 0.02, 0.006, 0.002, 0.0006]:
        rng = np.random.default_rng(17)
        img, y, h, off, mask, pf, pg, prior, sigma = build(la, "tv10", shape, rng)
        gamma, tau = sigma ** 2, ts * sigma ** 2
        C, nit, seed, off_c = 4, 4, 1234, 40
        x0 = img[None] + rng.normal(0, 3, (C,) + shape)
        noise = rng.standard_normal((nit, C) + shape)
        smp = la.MYMALASampler(pf, pg, shape, n_chains=C, tau=tau, gamma=gamma, noise="injected", seed=seed, chain_offset=off_c)
        smp.set_state(x0)
        us = np.stack([O.philox_uniforms(seed, k, off_c + np.arange(C)) for k in range(nit)])
        xo, acc_o, la_o = O.mymala_batched(x0, y, h, off, 1 / sigma ** 2, tau, gamma, prior, nit, lambda k: noise[k], lambda k: us[k], mask=mask)
        las = []
        for k in range(nit):
            smp.step(1, noise=noise[k:k + 1])
            las.append(smp.acceptance()[1].cpu().numpy())
        las = np.array(las)
        print(shape, "tau/sigma^2", ts, "\n  oracle", np.round(la_o, 3).tolist(), "\n  device", np.round(las, 3).tolist(), flush=True)
        smp.close()
