import sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import lmc_atomi_amd as la
from oracle import lmc_oracle as O, lmc_oracle_c as OC
import importlib.util
spec = importlib.util.spec_from_file_location('t', 'tests/test_gpu_rtol.py'); t = importlib.util.module_from_spec(spec); spec.loader.exec_module(t)
rel = t.rel
for shape, niter in [((22, 96), 20), ((24, 136), 20)]:
    x = t.images(shape, 6, niter)[:6] + 40.0
    gam = 15.0
    ref, ps = OC.tv_prox_fgp(x, gam, niter, rtol=1e-4, return_passes=True)
    fixed = OC.tv_prox_fgp(x, gam, niter)
    for path in ("auto", "passes"):
        tv = la.TV(shape, sigma=1.0, niter=niter, rtol=1e-4, exit_path=path)
        got = tv.prox(x.reshape(6, -1), gam).reshape(x.shape)
        print(shape, niter, path, 'prox rtol :', [f"{rel(got[c], ref[c]):.1e}" for c in range(6)], ps)
    got = la.TV(shape, sigma=1.0, niter=niter).prox(x.reshape(6, -1), gam).reshape(x.shape)
    print(shape, niter, 'prox fixed:', [f"{rel(got[c], fixed[c]):.1e}" for c in range(6)])
    h = np.ones((5, 5)) / 25.0
    y = O.blur(x[1], h, (2, 2))
    for rt in (0.0, 1e-4):
        me = la.L2_ncvx_tv(dims=shape, Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y.ravel(), sigma=1 / 0.5625, lamda=0.3, gamma=gam, isotropic=True, niter=niter, rtol=rt)
        g = me.grad(x.reshape(6, -1)).reshape(x.shape)
        pr = ref if rt else fixed
        errs = []
        for c in range(6):
            gl2 = (1 / 0.5625) * O.blur_adjoint(O.blur(x[c], h, (2, 2)) - y, h, (2, 2))
            want = gl2 - 0.3 * (x[c] - pr[c]) / gam
            errs.append(f"{rel(g[c], want):.1e}")
        print(shape, niter, 'me grad rtol', rt, errs)
    l2 = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / 0.5625)
    g = l2.grad(x.reshape(6, -1)).reshape(x.shape)
    print('l2 grad', [f"{rel(g[c], (1 / 0.5625) * O.blur_adjoint(O.blur(x[c], h, (2, 2)) - y, h, (2, 2))):.1e}" for c in range(6)])
