#!/usr/bin/env python3
"""R4-style statistics of the TV-prox variants against the cold K = 10 chain (the reference's configuration, niter_tv = 10): posterior
mean over chains x iterations at 256 x 256, 4096 Philox chains x 60 iterations from x0 = 0, same model as tests/test_gpu_r4.py.
Variants: warm-dual K = 1, 2, 3, cold K = 2, 6, the lagged reading of K = 10, and K = 50 (a nearly converged prox).
Prints one JSON line; run on the GPU box."""
import json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
import lmc_atomi_amd as la
from tests import _r4_cpu as R
from oracle import lmc_oracle as O     # blur of the synthetic truth only (set-up)

SIGMA, TAU_REG = 0.75, 0.3
GAMMA, TAU = SIGMA ** 2, 0.2 * SIGMA ** 2
shape, C, T = (256, 256), 4096, 60
img = R.truth(*shape)
h = np.ones((5, 5)) / 25.0
y = O.blur(img, h, (2, 2)) + np.random.default_rng(0).normal(0, SIGMA, shape)
pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)


def run(K, seed, warm=False, lagged=False):
    pg = la.TV(shape, sigma=TAU_REG, niter=K, warm=warm, lagged_output=lagged)
    smp = la.MYULASampler(pf, pg, shape, n_chains=C, tau=TAU, gamma=GAMMA, seed=seed, moments=True)
    smp.set_state(np.zeros(shape))
    torch.cuda.synchronize(); t0 = time.perf_counter()
    smp.step(T)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
    s1, s2, n = smp.moments()
    m = (s1 / n).cpu().numpy(); v = (s2 / n).cpu().numpy() - m * m
    name = smp.kernel_name
    smp.close()
    return m, v, dt / T * 1e3, name


rel = lambda a, b: float(np.linalg.norm(a - b) / np.linalg.norm(b))
ref, vref, ms_ref, _ = run(10, 11)
ref2, _, _, _ = run(10, 12)
out = {"shape": shape, "chains": C, "iterations": T, "mc_noise_rel_l2 (cold K=10, two seeds)": rel(ref2, ref), "ms_per_iter cold K=10": ms_ref, "variants": {}}
for label, kw in (("warm K=1", dict(K=1, warm=True)), ("warm K=2", dict(K=2, warm=True)), ("warm K=3", dict(K=3, warm=True)),
                  ("cold K=2", dict(K=2)), ("cold K=6", dict(K=6)), ("cold K=10 lagged (9 updates)", dict(K=10, lagged=True)),
                  ("cold K=50", dict(K=50))):
    m, v, ms, name = run(seed=13, **kw)                     # independent noise: the difference contains the Monte-Carlo error
    mc, vc, _, _ = run(seed=11, **kw)                        # common random numbers (the reference run's Philox stream): the bias alone
    out["variants"][label] = {"rel_l2_mean_vs_cold_K10": rel(m, ref), "rel_l2_var_vs_cold_K10": rel(v, vref),
                              "rel_l2_mean_same_noise": rel(mc, ref), "rel_l2_var_same_noise": rel(vc, vref), "ms_per_iter": ms, "kernel": name}
print(json.dumps(out))
