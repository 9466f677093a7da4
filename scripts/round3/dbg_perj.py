import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import lmc_atomi_amd as la
from oracle import lmc_oracle_c as OC
import importlib.util
spec = importlib.util.spec_from_file_location('t', 'tests/test_gpu_rtol.py'); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
rel = t.rel
for shape in [(22, 96), (24, 136)]:
    x = t.images(shape, 6, 20)[:6] + 40.0
    gam = 15.0
    for j in range(1, 23):
        got = la.TV(shape, sigma=1.0, niter=j).prox(x.reshape(6, -1), gam).reshape(x.shape)
        ref = OC.tv_prox_fgp(x, gam, j)
        print(shape, j, [f"{rel(got[c], ref[c]):.1e}" for c in range(6)])
