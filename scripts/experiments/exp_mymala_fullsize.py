"""MYMALA at 512x512 from x0 = 0: log acceptance ratios and energies per iteration for small step sizes (debugging aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import lmc_atomi_amd as la
import bench

H = W = 512
sigma, tau_reg = 0.75, 0.3
u, h, y = bench.synth_problem(H, W, sigma)
for ts in [0.03, 0.01]:
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2 * ts
    pf = la.L2(Op=la.Convolve2D((H, W), h, offset=(2, 2)), b=y, sigma=1 / sigma ** 2)
    pg = la.TV((H, W), sigma=tau_reg, niter=10)
    smp = la.MYMALASampler(pf, pg, (H, W), n_chains=4, tau=tau, gamma=gamma, seed=0)
    smp.set_state(np.zeros((H, W), dtype=np.float32))
    print("tau_scale", ts, "tau", tau)
    for k in range(8):
        smp.step(1)
        acc, lal = smp.acceptance()
        f, g = smp.energies()
        print("  it", k, "log_alpha", np.round(lal.cpu().numpy(), 2).tolist(), "acc", acc.cpu().numpy().tolist(),
              "f", np.round(f.cpu().numpy()[:2], 1).tolist(), "g", np.round(g.cpu().numpy()[:2], 1).tolist(), flush=True)
    smp.close()
