"""ULPDA at the headline shape (fewer chains): how many Chebyshev iterations does the warm-started implicit step need?
Runs the same 30 ULPDA iterations (Philox noise) with LMC_CHEB_K = 10 (the a-priori count) and smaller counts in subprocesses and
compares the final states."""
import os, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
WORKER = r'''
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
import lmc_atomi_amd as la, bench
H = W = 512; sigma = 0.75
u, h, y = bench.synth_problem(H, W, sigma)
pf = la.L2(Op=la.Convolve2D((H, W), h, offset=(2, 2)), b=y, sigma=1 / sigma ** 2); pf.niter = 50
smp = la.ULPDASampler(pf, la.L21(ndim=2, sigma=0.3), la.Gradient((H, W)), (H, W), n_chains=16, tau=0.95 * sigma ** 2, mu=1.0, theta=1.0,
                      gfirst=False, seed=0)
smp.set_state(np.zeros((H, W), dtype=np.float32))
smp.step(30)
np.save(sys.argv[2], smp.get_state().cpu().numpy())
'''
outs = {}
for k in (10, 0, 6):      # 0 = adaptive (default)
    out = f"/tmp/cheb_k{k}.npy"
    subprocess.run([sys.executable, "-c", WORKER, ROOT, out], env=dict(os.environ, **({'LMC_CHEB_K': str(k)} if k else {})), check=True)
    outs[k] = np.load(out).astype(np.float64)
    if k != 10:
        print("K", k, "rel diff of the state after 30 ULPDA iterations vs K=10:", np.linalg.norm(outs[k] - outs[10]) / np.linalg.norm(outs[10]), flush=True)
