#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ -- run in the BUILD CONTAINER only.

    python tests/golden/make_golden.py            # needs /root/reference (read-only)

What is executed
----------------
The reference's OWN code, loaded from /root/reference at run time (nothing from the
reference is copied into this repository):

* ``algs.MoreauYosidaUnadjustedLangevin`` / ``algs.UnadjustedLangevinPrimalDual`` /
  ``algs.L2_ncvx_tv`` (algs.py) -- the sampler loops and the non-log-concave data term;
* ``lmc.LangevinMonteCarlo.ula``, ``prox_lmc.ProximalLangevinMonteCarlo.{myula,pgld}``;
* the closed-form proxes of ``prox.py``.

The reference imports modules that are not installed in this image and cannot be fetched
(no network): pylops, pyproximal, fire, fastprogress, seaborn, scienceplots, ot.  They are
replaced by inert namespace stand-ins (no arithmetic; ``progress_bar`` is the identity
iterator).  The operator objects that pylops / pyproximal would supply (blur, gradient,
L2 / L1 / L21 / TV proxes) are the oracle's restatements (``oracle/lmc_oracle.py``), passed
into the reference loops through the duck-typed protocol the loops consume.  Hence the
vectors pin: recursion, step formulae, RNG consumption order, output layout, closed-form
proxes, L2_ncvx_tv's own formulae.  They do NOT pin pylops/pyproximal arithmetic
("parity unpinned", see oracle header).

A second, independent fixture (``tv_chambolle.npz``) is produced with
``/opt/conda/bin/python3.9`` + scikit-image 0.18.3 (``denoise_tv_chambolle``), a converged
ROF solver that shares nothing with this repository, to check the oracle's TV prox at
convergence.
"""
import importlib.util
import os
import subprocess
import sys
import types

os.environ.setdefault("MPLBACKEND", "Agg")
import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, ROOT)
from oracle import lmc_oracle as O  # noqa: E402


HONOUR_RTOL = [False]


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_standins():
    class ProxOperator:                      # namespace-only base, as algs.py:132 expects
        def __init__(self, Op=None, hasgrad=False):
            self.Op, self.hasgrad = Op, hasgrad

    # Identity is only ever combined as Identity(n) + c * (Op.H * Op) and handed to scipy's lsqr (algs.py:247-251):
    # the oracle's Identity operator carries that algebra (matvec / rmatvec / shape / dtype for scipy).
    _mod("pylops", MatrixMult=None, Identity=lambda n, dtype=None: O.Identity(n))
    _mod("pylops.optimization")
    _mod("pylops.optimization.basic", lsqr=None)
    _mod("pylops.utils")
    _mod("pylops.utils.backend", get_array_module=lambda x: np,
         get_module_name=lambda m: "numpy", to_numpy=lambda x: x)
    # L2_ncvx_tv.__init__ instantiates pyproximal.L1 / pyproximal.TV (algs.py:166,169):
    # hand it the oracle's restatements of those two operators.
    # TV: rtol forced to 0 for the main fixtures (the device's fixed-count semantics); gen_algs_rtol() flips HONOUR_RTOL to
    # generate the second set in which the inner prox keeps the early exit the reference asks for (algs.py:169: rtol = 1e-4)
    _mod("pyproximal", ProxOperator=ProxOperator, L1=O.L1,
         TV=lambda dims, sigma, niter, rtol: O.TV(dims, sigma, niter, rtol=(rtol if HONOUR_RTOL[0] else 0.0)))
    _mod("pyproximal.ProxOperator", _check_tau=lambda f: f)
    _mod("fastprogress", progress_bar=lambda it, *a, **k: it)
    _mod("fire", Fire=lambda *a, **k: None)
    _mod("seaborn")
    _mod("scienceplots")
    ot = _mod("ot")
    ot.plot = _mod("ot.plot")
    import matplotlib.style.core as msc
    for s in ("science", "grid"):
        msc.library.setdefault(s, {})


def load_ref(name):
    spec = importlib.util.spec_from_file_location("ref_" + name, os.path.join(REF, name + ".py"))
    m = importlib.util.module_from_spec(spec)
    if name == "prox_lmc":
        sys.modules["prox"] = load_ref("prox")
    spec.loader.exec_module(m)
    return m


def versions():
    return np.array(f"numpy {np.__version__}; scipy {scipy.__version__}; python {sys.version.split()[0]}")


def synth_image(ny, nx, seed):
    """Piecewise-constant blocks + ramp in [0,255] (no file I/O)."""
    rng = np.random.default_rng(seed)
    img = np.zeros((ny, nx))
    for _ in range(6):
        i0, j0 = rng.integers(0, ny - 2), rng.integers(0, nx - 2)
        i1, j1 = rng.integers(i0 + 1, ny + 1), rng.integers(j0 + 1, nx + 1)
        img[i0:i1, j0:j1] = rng.uniform(20, 235)
    img += np.linspace(0, 20, nx)[None, :]
    return np.clip(img, 0, 255)


def deconv_problem(ny, nx, k, sigma, seed):
    """Mirror of prox_lmc_deconv.py:52-59,88-94 on a synthetic image."""
    img = synth_image(ny, nx, 1234)
    rng = np.random.default_rng(seed)
    h = np.ones((k, k)) / (k * k)
    Hop = O.Convolve2D((ny, nx), h, offset=(k // 2, k // 2))
    y = (Hop * img.ravel()).reshape(ny, nx) + rng.normal(0, sigma, size=(ny, nx))
    return img, h, Hop, y


def gen_toy(out):
    lmc = load_ref("lmc")
    plmc = load_ref("prox_lmc")
    d = {}
    # BASELINE config 1: 1-D N(0,1), 1 chain, 1000 iterations, gamma = 5e-2, seed 0
    mus, Sig, om = [np.array([0.0])], [np.array([[1.0]])], [1.0]
    d["c1_ula"] = lmc.LangevinMonteCarlo(mus, Sig, om, K=1000, seed=0).ula(5e-2)
    p = plmc.ProximalLangevinMonteCarlo(mus, Sig, om, lamda=0.25, alpha=0.15, mu=0.0, K=1000, seed=0)
    d["c1_myula"] = p.myula(5e-2)
    d["c1_pgld"] = p.pgld(5e-2)
    # 2-D, 2-component mixture
    mus2 = [np.array([0.0, 0.0]), np.array([3.0, -1.0])]
    Sig2 = [np.array([[1.0, 0.3], [0.3, 0.8]]), np.array([[0.5, -0.1], [-0.1, 1.2]])]
    om2 = [0.4, 0.6]
    d["m2_ula"] = lmc.LangevinMonteCarlo(mus2, Sig2, om2, K=300, seed=3).ula(1e-1)
    p2 = plmc.ProximalLangevinMonteCarlo(mus2, Sig2, om2, lamda=0.25, alpha=0.15, mu=0.0, K=300, seed=3)
    d["m2_myula"] = p2.myula(5e-2)
    d["m2_pgld"] = p2.pgld(5e-2)
    # MYMALA (prox_lmc.py:134-158): accepted states only + their count
    pm = plmc.ProximalLangevinMonteCarlo(mus, Sig, om, lamda=0.25, alpha=0.15, mu=np.array([0.0]), K=400, seed=0)
    d["c1_mymala"], d["c1_mymala_n"] = pm.mymala(2e-1)
    pm2 = plmc.ProximalLangevinMonteCarlo(mus2, Sig2, om2, lamda=0.25, alpha=0.15, mu=np.array([0.5, -0.5]), K=300, seed=3)
    d["m2_mymala"], d["m2_mymala_n"] = pm2.mymala(3e-1)
    d["m2_mus"], d["m2_Sigmas"], d["m2_omegas"] = np.array(mus2), np.array(Sig2), np.array(om2)
    d["versions"] = versions()
    np.savez_compressed(out, **d)


def gen_prox(out):
    P = load_ref("prox")
    x = np.concatenate([np.linspace(-6, 6, 49), np.array([0.0, 1e-9, -1e-9, 0.3, -0.3, 2.5])])
    xp = np.abs(x) + 0.05
    d = {"x": x, "xp": xp}
    d["laplace_0.5"] = P.prox_laplace(x, 0.5)
    d["uncentered_laplace_0.7_1.5"] = P.prox_uncentered_laplace(x, 0.7, 1.5)
    d["gaussian_0.3"] = P.prox_gaussian(x, 0.3)
    d["conjugate_laplace_0.8"] = P.prox_conjugate(x, 0.8, P.prox_laplace)
    for name, p in (("4_3", 4 / 3), ("3_2", 3 / 2), ("3", 3), ("4", 4)):
        d["gen_gaussian_0.6_" + name] = P.prox_gen_gaussian(x, 0.6, p)
    d["huber_0.5_0.4"] = np.array([P.prox_huber(v, 0.5, 0.4) for v in x])      # scalar-only in the reference
    d["smoothed_laplace_0.9"] = P.prox_smoothed_laplace(x, 0.9)
    d["exp_0.5"] = np.array([P.prox_exp(v, 0.5) for v in x])
    d["gamma_0.4_1.3"] = P.prox_gamma(x, 0.4, 1.3)
    d["chi_0.7"] = P.prox_chi(x, 0.7)
    d["uniform_1.2"] = np.array([P.prox_uniform(v, 1.2) for v in x])
    d["triangular_-0.5_0.8"] = np.array([P.prox_triangular(v, -0.5, 0.8) for v in x])
    d["versions"] = versions()
    np.savez_compressed(out, **d)


def gen_algs(out):
    A = load_ref("algs")
    d = {}
    sigma = 0.75
    tau_reg = 0.3
    L = 1.0 / sigma ** 2
    gamma_myula = 1.0 / L
    tau_myula = 0.2 * gamma_myula                     # prox_lmc_deconv.py:92-94
    tau0, mu0 = 0.95 / L, 1.0                         # prox_lmc_deconv.py:88-90
    cases = [("a", 16, 16, 5, 0), ("b", 20, 24, 6, 1), ("c", 24, 18, 7, 2)]
    for tag, ny, nx, k, seed in cases:
        img, h, Hop, y = deconv_problem(ny, nx, k, sigma, seed)
        d[f"{tag}_img"], d[f"{tag}_h"], d[f"{tag}_y"] = img, h, y
        d[f"{tag}_meta"] = np.array([ny, nx, k, seed])
        x0 = np.zeros(ny * nx)
        Gop = O.Gradient((ny, nx))
        # --- MYULA with TV / L1 / L2 priors (prox_lmc_deconv.py:465)
        priors = {"tv": O.TV((ny, nx), sigma=tau_reg, niter=10),
                  "l1": O.L1(sigma=tau_reg),
                  "l2": O.L2(sigma=0.05)}
        for pname, pg in priors.items():
            if tag != "a" and pname != "tv":
                continue                              # keep the fixture small
            l2 = O.L2(Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
            cb = []
            s = A.MoreauYosidaUnadjustedLangevin(l2, pg, tau=tau_myula, gamma=gamma_myula, x0=x0,
                                                 niter=16, seed=seed, callback=lambda x: cb.append(x[0]))
            assert s.shape == (16, ny * nx) and len(cb) == 16
            d[f"{tag}_myula_{pname}"] = s
        # --- ULPDA (prox_lmc_deconv.py:455-457), both orders, with dual output
        for gfirst in (False, True):
            l2 = O.L2(Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
            xs, ys = A.UnadjustedLangevinPrimalDual(l2, O.L21(ndim=2, sigma=tau_reg), Gop, tau=tau0, mu=mu0,
                                                    theta=1.0, x0=x0, gfirst=gfirst, niter=8, seed=seed,
                                                    returny=True)
            d[f"{tag}_ulpda_l21_gfirst{int(gfirst)}_x"] = xs
            d[f"{tag}_ulpda_l21_gfirst{int(gfirst)}_y"] = ys
        if tag == "a":
            l2 = O.L2(Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
            d[f"{tag}_ulpda_l1"] = A.UnadjustedLangevinPrimalDual(l2, O.L1(sigma=tau_reg), Gop, tau=tau0, mu=mu0,
                                                                  theta=1.0, x0=x0, gfirst=False, niter=8, seed=seed)
        # --- L2_ncvx_tv (prox_lmc_deconv.py:106-113): value and gradient, MC-TV and ME-TV
        rng = np.random.default_rng(100 + seed)
        xt = (img + rng.normal(0, 5.0, img.shape)).ravel()
        mc = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, Op2=Gop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg,
                          gamma=15.0, isotropic=True, niter=50, warm=True)
        me = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg,
                          gamma=15.0, isotropic=True, niter=50, warm=True)
        d[f"{tag}_ncvx_x"] = xt
        d[f"{tag}_ncvx_mc_grad"], d[f"{tag}_ncvx_mc_val"] = mc.grad(xt.copy()), np.array(mc(xt.copy()))
        d[f"{tag}_ncvx_me_grad"], d[f"{tag}_ncvx_me_val"] = me.grad(xt.copy()), np.array(me(xt.copy()))
        # L2_ncvx_tv.prox (algs.py:201-267): the reference's OWN code incl. its scipy.sparse.linalg.lsqr solve
        # (iter_lim = niter = 50, warm start); prox mutates its argument in place (:217), hence the copies
        rngp = np.random.default_rng(200 + seed)
        vp = (img + rngp.normal(0, 5.0, img.shape)).ravel()
        mcp = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, Op2=Gop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg,
                           gamma=15.0, isotropic=True, niter=50, warm=True)
        d[f"{tag}_ncvx_prox_in"] = vp
        d[f"{tag}_ncvx_prox_out1"] = mcp.prox(vp.copy(), tau0)
        d[f"{tag}_ncvx_prox_out2"] = mcp.prox((vp + 1.0).copy(), tau0)          # second call: warm-started
        # ULPDA with the non-log-concave data term (prox_lmc_deconv.py:478-487 pattern: M2)
        mcu = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, Op2=Gop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg,
                           gamma=15.0, isotropic=True, niter=50, warm=True)
        d[f"{tag}_ulpda_mc"] = A.UnadjustedLangevinPrimalDual(mcu, O.L21(ndim=2, sigma=tau_reg), Gop, tau=tau0, mu=mu0,
                                                              theta=1.0, x0=x0, gfirst=False, niter=6, seed=seed)
        # ME-TV: prox (algs.py:221-223 + LSQR) and ULPDA driven by it (prox_lmc_deconv.py:506-515 pattern, ULPDA branch)
        mep = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg,
                           gamma=15.0, isotropic=True, niter=50, warm=True)
        d[f"{tag}_ncvx_me_prox_out"] = mep.prox(vp.copy(), tau0)
        meq = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg,
                           gamma=15.0, isotropic=True, niter=50, warm=True)
        d[f"{tag}_ulpda_me"] = A.UnadjustedLangevinPrimalDual(meq, O.L21(ndim=2, sigma=tau_reg), Gop, tau=tau0, mu=mu0,
                                                              theta=1.0, x0=x0, gfirst=False, niter=4, seed=seed)
        # MYULA with the ME-TV data term (prox_lmc_deconv.py:506-515 pattern: M3)
        meu = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg,
                           gamma=15.0, isotropic=True, niter=50, warm=True)
        d[f"{tag}_myula_me_tv"] = A.MoreauYosidaUnadjustedLangevin(
            meu, O.TV((ny, nx), sigma=tau_reg, niter=10), tau=tau_myula, gamma=gamma_myula, x0=x0, niter=4, seed=seed)
        # MYULA with the non-log-concave data term (prox_lmc_deconv.py:492-501 pattern)
        d[f"{tag}_myula_mc_tv"] = A.MoreauYosidaUnadjustedLangevin(
            mc, O.TV((ny, nx), sigma=tau_reg, niter=10), tau=tau_myula, gamma=gamma_myula, x0=x0, niter=6, seed=seed)
    d["params"] = np.array([sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0])
    d["versions"] = versions()
    np.savez_compressed(out, **d)


def gen_algs_rtol(out):
    """Second golden set: the reference's loops with the TV prox keeping upstream's early exit (``rtol = 1e-4``, pyproximal's
    default, which ``TV(dims, sigma, niter=niter_tv)`` at prox_lmc_deconv.py:122 does not override, and which algs.py:169 passes on
    explicitly) -- through the oracle's ``rtol`` path.  Stored next to the rtol = 0 run of the same seeds so that the divergence the
    device's fixed-count prox introduces is on record (tests/test_oracle_golden.py prints and bounds it)."""
    A = load_ref("algs")
    d = {}
    sigma, tau_reg = 0.75, 0.3
    gamma_myula = sigma ** 2
    tau_myula = 0.2 * gamma_myula
    ny, nx, k, seed = 24, 136, 5, 0         # 136 columns: the device's full-width pipeline covers the fixed-count launches
    img, h, Hop, y = deconv_problem(ny, nx, k, sigma, seed)
    d["img"], d["h"], d["y"], d["meta"] = img, h, y, np.array([ny, nx, k, seed])
    x0 = np.zeros(ny * nx)
    for tag, rtol in (("rtol0", 0.0), ("rtol1e-4", 1e-4)):
        HONOUR_RTOL[0] = rtol > 0
        l2 = O.L2(Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, niter=50, warm=True)
        d[f"myula_tv_{tag}"] = A.MoreauYosidaUnadjustedLangevin(l2, O.TV((ny, nx), sigma=tau_reg, niter=10, rtol=rtol), tau=tau_myula,
                                                                gamma=gamma_myula, x0=x0, niter=200, seed=seed)[::10]
        me = A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=15.0, isotropic=True,
                          niter=50, rtol=1e-4, warm=True)
        xt = (img + np.random.default_rng(100).normal(0, 5.0, img.shape)).ravel()
        d[f"ncvx_me_grad_{tag}"] = me.grad(xt.copy())
        d[f"myula_me_tv_{tag}"] = A.MoreauYosidaUnadjustedLangevin(
            me, O.TV((ny, nx), sigma=tau_reg, niter=10, rtol=rtol), tau=tau_myula, gamma=gamma_myula, x0=x0, niter=20, seed=seed)[::5]
    HONOUR_RTOL[0] = False
    d["ncvx_x"] = xt
    d["params"] = np.array([sigma, tau_reg, tau_myula, gamma_myula])
    d["versions"] = versions()
    np.savez_compressed(out, **d)


CHAMBOLLE = r'''
import sys, numpy as np
from skimage.restoration import denoise_tv_chambolle
import skimage
x = np.load(sys.argv[1])["x"]
res = {}
for w in (0.16875, 2.0, 15.0):
    res["w_%g" % w] = denoise_tv_chambolle(x, weight=w, eps=1e-12, n_iter_max=20000)
res["versions"] = np.array("skimage %s numpy %s" % (skimage.__version__, np.__version__))
np.savez_compressed(sys.argv[2], x=x, **res)
'''


PYWT = r'''
import sys, numpy as np, pywt
x = np.load(sys.argv[1])["x"]
res = {}
for thr in (0.1, 2.0):
    cf = pywt.wavedec2(x, "haar", level=3, mode="periodization")
    new = [cf[0]] + [tuple(pywt.threshold(d, thr, "soft") for d in lev) for lev in cf[1:]]
    res["thr_%g" % thr] = pywt.waverec2(new, "haar", mode="periodization")
    res["val"] = np.array(sum(np.abs(d).sum() for lev in cf[1:] for d in lev))
res["versions"] = np.array("pywt %s numpy %s" % (pywt.__version__, np.__version__))
np.savez_compressed(sys.argv[2], x=x, **res)
'''


def gen_algs_aniso(out):
    """Third golden set: the ANISOTROPIC MC-TV branches of the reference's ``L2_ncvx_tv`` (``isotropic=False`` with ``Op2 = Gradient``:
    value algs.py:173-190 without the pixel-norm reduction, prox pre-step :218-219, grad :278-279) -- the reference's own class code, with
    the oracle's L1 standing in for ``pyproximal.L1`` (a soft threshold).  Value, gradient, prox (incl. the warm-started second call) and
    short MYULA / ULPDA trajectories driven by the class."""
    A = load_ref("algs")
    d = {}
    sigma, tau_reg = 0.75, 0.3
    L = 1.0 / sigma ** 2
    gamma_myula = 1.0 / L
    tau_myula = 0.2 * gamma_myula
    tau0, mu0 = 0.95 / L, 1.0
    for tag, ny, nx, k, seed, gam in [("a", 16, 16, 5, 0, 15.0), ("b", 20, 24, 6, 1, 4.0)]:
        img, h, Hop, y = deconv_problem(ny, nx, k, sigma, seed)
        d[f"{tag}_img"], d[f"{tag}_h"], d[f"{tag}_y"] = img, h, y
        d[f"{tag}_meta"] = np.array([ny, nx, k, seed, gam])
        Gop = O.Gradient((ny, nx))
        x0 = np.zeros(ny * nx)
        mk = lambda: A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, Op2=Gop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=gam,
                                  isotropic=False, niter=50, warm=True)
        rng = np.random.default_rng(300 + seed)
        xt = (img + rng.normal(0, 5.0, img.shape)).ravel()
        mc = mk()
        d[f"{tag}_x"] = xt
        d[f"{tag}_grad"], d[f"{tag}_val"] = mc.grad(xt.copy()), np.array(mc(xt.copy()))
        vp = (img + np.random.default_rng(400 + seed).normal(0, 5.0, img.shape)).ravel()
        mcp = mk()
        d[f"{tag}_prox_in"] = vp
        d[f"{tag}_prox_out1"] = mcp.prox(vp.copy(), tau0)
        d[f"{tag}_prox_out2"] = mcp.prox((vp + 1.0).copy(), tau0)
        d[f"{tag}_myula"] = A.MoreauYosidaUnadjustedLangevin(mk(), O.TV((ny, nx), sigma=tau_reg, niter=10), tau=tau_myula, gamma=gamma_myula,
                                                             x0=x0, niter=6, seed=seed)
        d[f"{tag}_ulpda"] = A.UnadjustedLangevinPrimalDual(mk(), O.L21(ndim=2, sigma=tau_reg), Gop, tau=tau0, mu=mu0, theta=1.0, x0=x0,
                                                           gfirst=False, niter=6, seed=seed)
    d["params"] = np.array([sigma, tau_reg, tau_myula, gamma_myula, tau0, mu0])
    np.savez_compressed(out, **d)


def gen_algs_aniso_me(out):
    """Fourth golden set: the ANISOTROPIC ME-TV branch of the reference's ``L2_ncvx_tv`` (``isotropic=False``, ``Op2 = None``): its inner prox is a 1-D TV
    over the flattened image, ``pyproximal.TV((np.prod(dims),), 1., niter, rtol)`` (algs.py:170) -- the reference's own class code, with the oracle's 1-D
    restatement standing in for pyproximal's (``lmc_oracle.tv1d_prox_fgp``; upstream arithmetic unpinned like every pyproximal operator).  Value, gradient,
    prox (incl. the warm-started second call) and a short MYULA trajectory driven by the class, at the class's own ``rtol = 1e-4`` and at ``rtol = 0``."""
    A = load_ref("algs")
    d = {}
    sigma, tau_reg = 0.75, 0.3
    L = 1.0 / sigma ** 2
    gamma_myula = 1.0 / L
    tau_myula = 0.2 * gamma_myula
    tau0 = 0.95 / L
    for tag, ny, nx, k, seed, gam, niter in [("a", 16, 24, 5, 0, 15.0, 30), ("b", 24, 136, 5, 1, 4.0, 20)]:
        img, h, Hop, y = deconv_problem(ny, nx, k, sigma, seed)
        d[f"{tag}_img"], d[f"{tag}_h"], d[f"{tag}_y"] = img, h, y
        d[f"{tag}_meta"] = np.array([ny, nx, k, seed, gam, niter])
        x0 = np.zeros(ny * nx)
        for rt, rtol in (("rtol1e-4", 1e-4), ("rtol0", 0.0)):
            HONOUR_RTOL[0] = rtol > 0
            mk = lambda: A.L2_ncvx_tv(dims=(ny, nx), Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2, lamda=tau_reg, gamma=gam, isotropic=False, niter=niter,
                                      rtol=1e-4, warm=True)
            xt = (img + np.random.default_rng(500 + seed).normal(0, 5.0, img.shape)).ravel()
            me = mk()
            d[f"{tag}_x"] = xt
            d[f"{tag}_grad_{rt}"], d[f"{tag}_val_{rt}"] = me.grad(xt.copy()), np.array(me(xt.copy()))
            vp = (img + np.random.default_rng(600 + seed).normal(0, 5.0, img.shape)).ravel()
            mep = mk()
            d[f"{tag}_prox_in"] = vp
            d[f"{tag}_prox_out1_{rt}"] = mep.prox(vp.copy(), tau0)
            d[f"{tag}_prox_out2_{rt}"] = mep.prox((vp + 1.0).copy(), tau0)
            d[f"{tag}_myula_{rt}"] = A.MoreauYosidaUnadjustedLangevin(mk(), O.TV((ny, nx), sigma=tau_reg, niter=10), tau=tau_myula, gamma=gamma_myula,
                                                                      x0=x0, niter=6, seed=seed)
    HONOUR_RTOL[0] = False
    d["params"] = np.array([sigma, tau_reg, tau_myula, gamma_myula, tau0])
    d["versions"] = versions()
    np.savez_compressed(out, **d)


def gen_epsg_array(out):
    """Fifth golden set: the reference's MYULA loop (algs.py:559-570) with an ARRAY-valued ``epsg`` (algs.py:509,539-542) -- the prox parameter
    ``epsg * gamma`` it hands to ``proxg.prox`` is then an array, which a closed-form prox broadcasts: per pixel for the flattened image (``x`` of shape
    ``(n,)``), per right-hand side for ``x`` of shape ``(n, nrhs)`` with ``epsg`` of shape ``(nrhs,)``.  Priors: the oracle's l1 and l2 (pyproximal.L1 / L2
    restated) and the reference's own ``prox.prox_laplace`` wrapped as a prox operator."""
    A = load_ref("algs")
    P = load_ref("prox")
    d = {}
    sigma = 0.75
    L = 1.0 / sigma ** 2
    gamma_myula = 1.0 / L
    tau_myula = 0.2 * gamma_myula
    ny, nx, k, seed, nit = 16, 40, 5, 3, 5
    img, h, Hop, y = deconv_problem(ny, nx, k, sigma, seed)
    n = ny * nx
    f = O.L2(Op=Hop, b=y.ravel(), sigma=1 / sigma ** 2)
    rng = np.random.default_rng(77)
    e_px = rng.uniform(0.2, 3.0, n)                       # one weight per pixel
    d["img"], d["h"], d["y"], d["epsg_pixel"] = img, h, y, e_px
    d["meta"] = np.array([ny, nx, k, seed, nit])

    class Laplace:                                        # the reference's closed form as the prior's prox (as at prox_lmc.py:106,115)
        def __init__(self, lam): self.lam = lam
        def __call__(self, x): return self.lam * float(np.sum(np.abs(x)))
        def prox(self, x, tau): return P.prox_laplace(x, tau * self.lam)

    x0 = np.zeros(n)
    for name, g in (("l1", O.L1(sigma=2.0)), ("l2", O.L2(sigma=0.05)), ("laplace", Laplace(1.5))):
        d[f"pixel_{name}"] = A.MoreauYosidaUnadjustedLangevin(f, g, tau=tau_myula, gamma=gamma_myula, epsg=e_px, x0=x0, niter=nit, seed=seed)
    # per right-hand side: x of shape (n, nrhs), a pointwise (Diagonal-free) data term so that grad broadcasts over the columns
    nrhs = 3
    e_rhs = np.array([0.5, 1.0, 4.0])
    yv = y.ravel()[:, None]
    class L2Id:                                           # f = sigma/2 ||x - y||^2 column by column
        def __call__(self, x): return 0.5 / sigma ** 2 * float(np.sum((x - yv) ** 2))
        def grad(self, x): return (x - yv) / sigma ** 2
    X0 = np.zeros((n, nrhs))
    d["epsg_rhs"] = e_rhs
    for name, g in (("l1", O.L1(sigma=2.0)), ("l2", O.L2(sigma=0.05))):
        d[f"rhs_{name}"] = A.MoreauYosidaUnadjustedLangevin(L2Id(), g, tau=tau_myula, gamma=gamma_myula, epsg=e_rhs, x0=X0, niter=nit, seed=seed)
    d["params"] = np.array([sigma, tau_myula, gamma_myula])
    d["versions"] = versions()
    np.savez_compressed(out, **d)


def gen_pywt(out):
    """Independent check of the Haar-l1 prox (BASELINE config 5's prior) with PyWavelets from the conda interpreter."""
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39):
        print("skip haar_pywt.npz: no", py39)
        return
    img = synth_image(32, 48, 1234) + np.random.default_rng(6).normal(0, 4.0, (32, 48))
    tmp = "/tmp/_pywt_in.npz"
    np.savez(tmp, x=img)
    subprocess.run([py39, "-W", "ignore", "-c", PYWT, tmp, out], check=True)


def gen_chambolle(out):
    py39 = "/opt/conda/bin/python3.9"
    if not os.path.exists(py39):
        print("skip tv_chambolle.npz: no", py39)
        return
    img = synth_image(24, 20, 1234) + np.random.default_rng(5).normal(0, 8.0, (24, 20))
    tmp = "/tmp/_chamb_in.npz"
    np.savez(tmp, x=img)
    subprocess.run([py39, "-W", "ignore", "-c", CHAMBOLLE, tmp, out], check=True)


if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("needs the reference at /root/reference (build container only)")
    install_standins()
    gen_toy(os.path.join(HERE, "toy.npz"))
    gen_prox(os.path.join(HERE, "prox.npz"))
    gen_algs(os.path.join(HERE, "algs.npz"))
    gen_algs_rtol(os.path.join(HERE, "algs_rtol.npz"))
    if "--aniso-me" in sys.argv or not os.path.exists(os.path.join(HERE, "algs_aniso_me.npz")):
        gen_algs_aniso_me(os.path.join(HERE, "algs_aniso_me.npz"))
    gen_algs_aniso(os.path.join(HERE, "algs_aniso.npz"))
    gen_epsg_array(os.path.join(HERE, "epsg_array.npz"))
    gen_chambolle(os.path.join(HERE, "tv_chambolle.npz"))
    gen_pywt(os.path.join(HERE, "haar_pywt.npz"))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)), "bytes")
