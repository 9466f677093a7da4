#!/usr/bin/env python3
"""Is there a per-call cost in lmc_sampler_step for the ME-TV models?  ms per iteration for calls of 1, 3, 10 and 30 iterations."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lmc_atomi_amd as la
from bench import synth_problem
H = W = 512; C = 1024; sigma = 0.75
_, h, y = synth_problem(H, W, sigma, 0, "box", 5)
Hop = la.Convolve2D((H, W), h, offset=(2, 2))
for rtol in (1e-4, 0.0):
    f = la.L2_ncvx_tv(dims=(H, W), Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=0.3, gamma=15.0, isotropic=True, niter=50, rtol=rtol)
    g = la.TV((H, W), sigma=0.3, niter=10, rtol=rtol)
    smp = la.MYULASampler(f, g, (H, W), n_chains=C, tau=0.2 * sigma ** 2, gamma=sigma ** 2, seed=0)
    smp.set_state(np.zeros((H, W), dtype=np.float32))
    smp.step(40)
    for n in (1, 3, 10, 30, 3, 1):
        torch.cuda.synchronize(); t0 = time.perf_counter(); smp.step(n); torch.cuda.synchronize()
        print(f"rtol {rtol:g}: step({n:2d}) {1e3 * (time.perf_counter() - t0) / n:8.3f} ms per iteration")
    smp.close()
