#!/usr/bin/env python3
"""Timing experiment: which roles share a SIMD in the pipe kernel.  Hardware waves w and w + 4 of the 8-wave workgroup run on the same SIMD; a build with
-DLMC_EXP_PERM takes the role of every hardware wave from LMC_EXP_PERM (hex, nibble w = role of hardware wave w; roles 0 L, 1..5 T1..T5, 6 C, 7 N).  All
105 pairings of the eight roles, the headline configuration (fixed K = 10) or the chain as the reference configures it (--rtol).  Results stay exact (roles
only move between waves); prints ms per launch, sorted."""
import argparse, itertools, os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import lmc_atomi_amd as la
ap = argparse.ArgumentParser(); ap.add_argument("--rtol", type=float, default=0.0); ap.add_argument("--steps", type=int, default=16); ap.add_argument("--mc", action="store_true"); ap.add_argument("--me", action="store_true"); ap.add_argument("--k", type=int, default=5)
args = ap.parse_args()
H = W = 512; C = 1024; sigma = 0.75
rng = np.random.default_rng(0)
k = args.k
from bench import synth_problem          # bench.py's own data: the exit pass of the TV prox (4 on it) decides which waves are live
_, h, img = synth_problem(H, W, sigma, 0, "box", k)
Hop = la.Convolve2D((H, W), h, offset=(k // 2, k // 2))
f = la.L2(Op=Hop, b=img.ravel(), sigma=1 / sigma ** 2)
if args.mc:
    f = la.L2_ncvx_tv(dims=(H, W), Op=Hop, Op2=la.Gradient((H, W)), b=img.ravel(), sigma=1 / sigma ** 2, lamda=0.3, gamma=15.0)
if args.me:
    f = la.L2_ncvx_tv(dims=(H, W), Op=Hop, b=img.ravel(), sigma=1 / sigma ** 2, lamda=0.3, gamma=15.0, isotropic=True, niter=50, rtol=args.rtol)
g = la.TV((H, W), sigma=0.3, niter=10, rtol=args.rtol)
smp = la.MYULASampler(f, g, (H, W), n_chains=C, tau=0.2 * sigma ** 2, gamma=sigma ** 2, seed=0)
smp.set_state(np.zeros((H, W), dtype=np.float32))
smp.step(40 if args.me else (70 if args.rtol else 10))
NAMES = ["L", "T1", "T2", "T3", "T4", "T5", "C", "N"]
def pairings(items):
    if not items: yield []; return
    a = items[0]
    for i in range(1, len(items)):
        b = items[i]; rest = items[1:i] + items[i + 1:]
        for p in pairings(rest): yield [(a, b)] + p
def timed():
    torch.cuda.synchronize(); t0 = time.perf_counter(); smp.step(args.steps); torch.cuda.synchronize(); return (time.perf_counter() - t0) / args.steps * 1e3
res = []
for pr in pairings(list(range(8))):
    perm = [0] * 8
    for i, (a, b) in enumerate(pr): perm[i], perm[i + 4] = a, b
    os.environ["LMC_EXP_PERM"] = "%x" % sum(r << (4 * w) for w, r in enumerate(perm))
    smp.step(2); ms = timed()
    res.append((ms, " | ".join(f"{NAMES[a]}+{NAMES[b]}" for a, b in pr), os.environ["LMC_EXP_PERM"]))
res.sort()
for ms, name, code in res: print(f"{ms:7.4f} ms  {name}   {code}")
os.environ["LMC_EXP_PERM"] = "0"
smp.step(2); print(f"{timed():7.4f} ms  (the library's own order)")
if args.rtol and not args.me:
    import ctypes as C
    ps = torch.zeros(1024, dtype=torch.int32, device="cuda"); rr = (C.c_uint64 * 4)()
    print("exit passes of the last iteration:", np.bincount(smp.tv_exit_stats()[0].cpu().numpy(), minlength=11).tolist() if hasattr(smp, "tv_exit_stats") else "?")
smp.close()
