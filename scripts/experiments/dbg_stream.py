import os, sys, ctypes, numpy as np
sys.path.insert(0, '.')
if os.environ.get("LMC_DBG"): os.environ["LMC_ATOMI_LIB"] = os.path.abspath("build/dbg/liblmc_atomi_dbg.so")
import torch
import lmc_atomi_amd as la
from lmc_atomi_amd import _capi
from oracle import lmc_oracle as O
lib = _capi.load()
la.set_step_variant("stream")
def dbg():
    out = (ctypes.c_longlong * 8)()
    if os.environ.get("LMC_DBG"): lib.lmc_debug_read(out)
    return list(out)[:4]
for niter, shape, gamma in [(10, (16, 16), 0.16875), (10, (100, 70), 0.16875), (10, (100, 128), 0.17), (1, (8, 8), 2.0), (12, (64, 64), 2.0), (10, (65, 129), 15.0), (3, (4, 4), 0.5), (10, (512, 512), 0.17)]:
    rng = np.random.default_rng(niter)
    x = rng.normal(0, 8, shape) + 100
    tv = la.TV(shape, sigma=0.3, niter=niter)
    out = tv.prox(x.ravel(), gamma / 0.3)
    torch.cuda.synchronize()
    ref = O.tv_prox_fgp(x, gamma, niter)
    print("run", niter, shape, "rel", np.linalg.norm(out - ref.ravel()) / np.linalg.norm(ref), "dbg[tag,idx,n,base]", dbg(), flush=True)
