"""Prox-operator plugin surface (pyproximal ``ProxOperator`` protocol as consumed by
algs.py: ``obj(x)`` :461,578 ; ``obj.prox(x, tau)`` :440,569 ; ``obj.proxdual(x, tau)`` :436,448 ;
``obj.grad(x)`` :569), computed by HIP kernels through the C ABI.

Every class also exposes ``descriptor()``: the plain-data description the fused sampler
kernels are configured from (``lmc_problem`` in include/lmc_atomi.h).  A device-side
function-pointer plugin ABI is deliberately not offered (it would defeat fusion); new proxes
are added as functors compiled into the library.
"""
from __future__ import annotations

import ctypes as C
import warnings

import numpy as np
import torch

from . import _capi, _dev
from .operators import Convolve2D, Diagonal, Identity, LinearOperator


def fgp_betas(niter, momentum="unlocbox"):
    """Momentum table beta_k = (t_{k-1}-1)/t_k of the TV dual iteration.  'unlocbox' is the
    sequence pyproximal.TV inherits from UNLocBoX [upstream]; 'fista' the textbook one."""
    t, out = 1.0, []
    for _ in range(int(niter)):
        if momentum == "unlocbox":
            tn = (1.0 + np.sqrt(4.0 * t * t)) / 2.0
        elif momentum == "fista":
            tn = (1.0 + np.sqrt(1.0 + 4.0 * t * t)) / 2.0
        elif momentum == "none":
            tn = 1.0
        else:
            raise ValueError(f"unknown momentum {momentum!r}")
        out.append((t - 1.0) / tn)
        t = tn
    return np.asarray(out, dtype=np.float32)


class ProxOperator:
    """Base class; subclassing hook mirrors ``super().__init__(Op, hasgrad)`` (algs.py:132)."""

    def __init__(self, Op=None, hasgrad=False):
        self.Op = Op
        self.hasgrad = hasgrad

    def proxdual(self, x, tau):
        """Moreau identity (prox.py:9-10): prox_{tau f*}(x) = x - tau prox_{f/tau}(x/tau)."""
        return x - tau * self.prox(x / tau, 1.0 / tau)

    def descriptor(self):
        raise NotImplementedError(f"{type(self).__name__} has no device functor")


class _Problem:
    """Assembles an ``lmc_problem`` from a data-term descriptor and a prior descriptor and keeps
    every buffer it points to alive."""

    def __init__(self, dims, data=None, prior=None, dev=None, options=None):
        self.dims = (int(dims[0]), int(dims[1]))
        self.device = _dev.device(dev)
        dev = self.device
        p = _capi.lmc_problem()
        p.struct_size = C.sizeof(_capi.lmc_problem)
        p.H, p.W = self.dims
        self._keep = []
        data = data or {"data_kind": _capi.DATA_NONE}
        prior = prior or {"prior_kind": _capi.PRIOR_NONE}
        p.data_kind = data["data_kind"]
        p.sigma_f = float(data.get("sigma_f", 0.0))
        if data.get("y") is not None:
            y = _dev.to_dev(data["y"], dev).reshape(self.dims)
            self._keep.append(y)
            p.y_dev = y.data_ptr()
        if data.get("mask") is not None:
            m = _dev.to_dev(data["mask"], dev).reshape(self.dims)
            self._keep.append(m)
            p.mask_dev = m.data_ptr()
        if p.data_kind == _capi.DATA_BLUR:
            h = np.ascontiguousarray(data["h"], dtype=np.float32)
            self._keep.append(h)
            p.kh, p.kw = h.shape
            p.oy, p.ox = data["offset"]
            p.h_host = _dev.fptr(h)
        p.prior_kind = prior["prior_kind"]
        p.prior_sigma = float(prior.get("prior_sigma", 0.0))
        if p.prior_kind == _capi.PRIOR_TV_ISO:
            p.tv_niter = int(prior["tv_niter"])
            p.tv_step = float(prior.get("tv_step", 0.125))
            b = np.ascontiguousarray(prior["tv_betas"], dtype=np.float32)
            if b.size != p.tv_niter:
                raise ValueError("tv_betas must have tv_niter entries")
            self._keep.append(b)
            p.tv_betas_host = _dev.fptr(b)
        p.ncvx_kind = int(data.get("ncvx_kind", _capi.NCVX_NONE))
        p.ncvx_lambda = float(data.get("ncvx_lambda", 0.0))
        p.ncvx_gamma = float(data.get("ncvx_gamma", 1.0))
        p.ncvx_niter = int(data.get("ncvx_niter", 0))
        # ABI 2: which truncated iterate the TV prox returns, warm-started dual, per-problem kernel variant / solver tolerance
        opt = dict(options or {})
        p.tv_lagged_output = 1 if (prior.get("tv_lagged_output") or data.get("tv_lagged_output") or opt.get("tv_lagged_output")) else 0
        p.tv_rtol = float(prior.get("tv_rtol", 0.0) or 0.0)      # > 0: the exact early-exit path of the TV prior's prox (slow; 0 = fixed count)
        p.tv_warm = 1 if (prior.get("tv_warm") or opt.get("tv_warm")) else 0
        v = opt.get("step_variant", 0) or 0
        p.step_variant = _capi.VARIANTS.index(v) if isinstance(v, str) else int(v)
        p.implicit_tol = float(opt.get("implicit_tol", data.get("implicit_tol", 0.0)) or 0.0)
        # ABI 3: early exit of the ME-TV inner prox, which of the two exit paths, launch policy (0 = the library decides)
        p.ncvx_rtol = float(data.get("ncvx_rtol", 0.0) or 0.0)
        p.tv_exit_path = int(prior.get("tv_exit_path", 0) or opt.get("tv_exit_path", 0) or 0)
        p.iterations_per_launch = int(opt.get("iterations_per_launch", 0) or 0)
        p.moments_overlap = int(opt.get("moments_overlap", 0) or 0)
        p.moments_bg_workgroups = int(opt.get("moments_bg_workgroups", 0) or 0)
        p.graph_replay = 1 if opt.get("graph_replay") else 0
        if p.prior_kind == _capi.PRIOR_EPROX:
            p.eprox_kind = int(prior["eprox_kind"])
            p.eprox_p0, p.eprox_p1 = float(prior["eprox_p0"]), float(prior["eprox_p1"])
            p.eprox_scale_mask = int(prior.get("eprox_scale_mask", 0))
        if opt.get("prox_scale") is not None:     # array-valued epsg (MYULA): device array + (chain, pixel) strides
            sc, cs, ps = opt["prox_scale"]
            self._keep.append(sc)
            p.prox_scale, p.prox_scale_chain_stride, p.prox_scale_pixel_stride = _dev.ptr(sc), int(cs), int(ps)
        self.c = p

    def eval(self, x, a, t, b, pt):
        """out = a*x - t*grad f(x) + b*prox_{pt*g}(x) for image-shaped / flat batches."""
        n = self.dims[0] * self.dims[1]
        xt = _dev.to_dev(x, self.device)
        if xt.numel() % n:
            raise ValueError(f"operand of shape {tuple(xt.shape)} is not a batch of {self.dims} images")
        out = torch.empty_like(xt)
        with torch.cuda.device(self.device):       # the problem's buffers (y, mask) live there
            _dev.run(xt, "lmc_fused_eval", C.byref(self.c), _dev.ptr(xt), _dev.ptr(out), xt.numel() // n,
                                                  a, t, b, pt)
        return _dev.like_input(out, x)

    def energies(self, x):
        n = self.dims[0] * self.dims[1]
        xt = _dev.to_dev(x, self.device)
        n_img = xt.numel() // n
        f = torch.empty(n_img, dtype=torch.float64, device=xt.device)
        g = torch.empty(n_img, dtype=torch.float64, device=xt.device)
        with torch.cuda.device(self.device):
            _dev.run(xt, "lmc_energies", C.byref(self.c), _dev.ptr(xt), n_img, _dev.ptr(f), _dev.ptr(g))
        return f, g


class L2(ProxOperator):
    """``f(x) = sigma/2 ||Op x - b||^2`` -- drop-in for ``pyproximal.L2(Op=H, b=y, sigma=1/sigma**2,
    niter=50, warm=True)`` (prox_lmc_deconv.py:101-103) and for the prior ``pyproximal.L2(sigma=lam)``.

    ``Op`` may be a :class:`Convolve2D`, :class:`Diagonal`, :class:`Identity` or ``None``.
    ``grad`` runs the fused blur-residual-adjoint kernel; value via the energy kernel.
    The implicit step ``prox`` with an operator (row a8 of SURVEY section 8, used by ULPDA only) is
    provided by :mod:`lmc_atomi_amd.algs` for ULPDA; for ``Op is None`` it is closed form.
    """

    def __init__(self, Op=None, b=None, sigma=1.0, niter=10, warm=True, dims=None):
        super().__init__(Op, True)
        self.b = b
        self.sigma = float(sigma)
        self.niter = niter
        self.warm = warm
        self.dims = dims if dims is not None else getattr(Op, "dims", None)
        self._prob = None
        self._x0 = None          # warm-start vector of the implicit step (stateful, as the reference's L2)

    # -- descriptors ---------------------------------------------------------------------
    def descriptor(self):
        """As the data term f of the sampler."""
        if self.Op is None or isinstance(self.Op, Identity):
            if self.b is None:
                raise NotImplementedError("L2 without b as a data term: use it as the prior instead")
            return {"data_kind": _capi.DATA_IDENTITY, "sigma_f": self.sigma, "y": self.b}
        if isinstance(self.Op, Convolve2D):
            return {"data_kind": _capi.DATA_BLUR, "sigma_f": self.sigma, "y": self.b, "h": self.Op.h,
                    "offset": self.Op.offset}
        if isinstance(self.Op, Diagonal):
            return {"data_kind": _capi.DATA_MASK, "sigma_f": self.sigma, "y": self.b, "mask": self.Op.d}
        raise NotImplementedError(f"no device functor for L2 with Op of type {type(self.Op).__name__}")

    def prior_descriptor(self):
        """As the prior g = sigma/2 ||x||^2 (closed-form prox)."""
        if self.Op is not None or self.b is not None:
            raise NotImplementedError("only L2(sigma=...) without Op/b can act as the prior")
        return {"prior_kind": _capi.PRIOR_L2, "prior_sigma": self.sigma}

    def _problem(self):
        if self._prob is None:
            if self.dims is None:
                raise ValueError("L2 needs `dims` (image shape) when Op does not carry it")
            self._prob = _Problem(self.dims, data=self.descriptor())
        return self._prob

    # -- protocol ------------------------------------------------------------------------
    def __call__(self, x):
        if self.Op is None and self.b is None:       # pure quadratic prior sigma/2 ||x||^2
            xt = _dev.to_dev(x).reshape(1, -1)
            _, g = _Problem((1, xt.shape[1]), prior=self.prior_descriptor()).energies(xt)
            return float(g[0])
        f, _ = self._problem().energies(x)
        return float(f[0]) if f.numel() == 1 else (f if isinstance(x, torch.Tensor) else f.cpu().numpy())

    def grad(self, x):
        return self._problem().eval(x, 0.0, -1.0, 0.0, 0.0)

    def prox(self, x, tau):
        """``(I + tau*sigma*Op^T Op)^{-1}(x + tau*sigma*Op^T b)`` (pyproximal.L2.prox; in-repo twin algs.py:224-256).
        With a blur operator: ``niter`` warm-started CG iterations on the GPU (lmc_l2_prox)."""
        if self.Op is None and self.b is None:
            return x / (1.0 + tau * self.sigma)
        prob = self._problem()
        n = self.dims[0] * self.dims[1]
        xt = _dev.to_dev(x, prob.device)
        n_img = xt.numel() // n
        lib = _dev.lib()
        if self.warm and self._x0 is not None and self._x0.numel() == xt.numel():
            out, warm = self._x0.clone(), 1
        else:
            out, warm = torch.empty_like(xt), 0
        nbytes = lib.lmc_l2_prox_workspace_bytes(n_img, self.dims[0], self.dims[1])
        ws = torch.empty(nbytes, dtype=torch.uint8, device=xt.device)
        with torch.cuda.device(prob.device):
            _capi.check(lib.lmc_l2_prox(C.byref(prob.c), _dev.ptr(xt), _dev.ptr(out), n_img, float(tau), int(self.niter),
                                        warm, _dev.ptr(ws), _dev.stream_ptr()))
            torch.cuda.current_stream().synchronize()
        if self.warm:
            self._x0 = out.clone()
        return _dev.like_input(out.reshape(xt.shape), x)


class L1(ProxOperator):
    """``sigma ||x||_1`` -- drop-in for ``pyproximal.L1(sigma=tau)`` (prox_lmc_deconv.py:119)."""

    def __init__(self, sigma=1.0, dims=None):
        super().__init__(None, False)
        self.sigma = float(sigma)
        self.dims = dims

    def prior_descriptor(self):
        return {"prior_kind": _capi.PRIOR_L1, "prior_sigma": self.sigma}

    def __call__(self, x):
        xt = _dev.to_dev(x).reshape(1, -1)
        p = _Problem((1, xt.shape[1]), prior=self.prior_descriptor())
        _, g = p.energies(xt)
        return float(g[0])

    def prox(self, x, tau):
        xt = _dev.to_dev(x)
        out = torch.empty_like(xt)
        par = np.asarray([self.sigma * float(tau)], dtype=np.float32)
        _dev.run(xt, "lmc_prox_elementwise", _capi.EPROX_LAPLACE, _dev.ptr(xt), _dev.ptr(out), xt.numel(),
                                                    _dev.fptr(par), 1)
        return _dev.like_input(out, x)

    def proxdual(self, x, tau):
        """Clip to [-sigma, sigma] (projection onto the dual ball)."""
        xt = _dev.to_dev(x)
        out = torch.empty_like(xt)
        par = np.asarray([self.sigma], dtype=np.float32)
        _dev.run(xt, "lmc_prox_elementwise", _capi.EPROX_UNIFORM, _dev.ptr(xt), _dev.ptr(out), xt.numel(),
                                                    _dev.fptr(par), 1)
        return _dev.like_input(out, x)


class L21(ProxOperator):
    """``sigma * sum_pixels ||(v_row, v_col)||_2`` on stacked fields of length 2n -- drop-in for
    ``pyproximal.L21(ndim=2, sigma=tau)`` (prox_lmc_deconv.py:116)."""

    def __init__(self, ndim=2, sigma=1.0):
        if ndim != 2:
            raise NotImplementedError("ndim=2 only (the reference's configuration)")
        super().__init__(None, False)
        self.ndim = ndim
        self.sigma = float(sigma)

    def __call__(self, x):
        xt = _dev.to_dev(x).reshape(2, -1)
        return self.sigma * float(torch.sqrt((xt.double() ** 2).sum(0)).sum())

    def proxdual(self, x, tau):
        xt = _dev.to_dev(x)
        n2 = xt.shape[-1]
        if n2 % 2:
            raise ValueError("stacked field must have even length")
        out = torch.empty_like(xt)
        _dev.run(xt, "lmc_dual_project", _dev.ptr(xt), _dev.ptr(out), xt.numel() // n2, 1, n2 // 2,
                                                self.sigma, 1)
        return _dev.like_input(out, x)

    def prox(self, x, tau):
        """Moreau identity with the dual projection: prox_{tau g}(x) = x - proj_{tau*sigma}(x)."""
        xt = _dev.to_dev(x)
        n2 = xt.shape[-1]
        out = torch.empty_like(xt)
        _dev.run(xt, "lmc_dual_project", _dev.ptr(xt), _dev.ptr(out), xt.numel() // n2, 1, n2 // 2,
                                                self.sigma * float(tau), 1)
        return _dev.like_input(xt - out, x)


class TV(ProxOperator):
    """``sigma * TV_iso(x)`` -- drop-in for ``pyproximal.TV(dims=img.shape, sigma=tau, niter=niter_tv)``
    (prox_lmc_deconv.py:122).  ``prox`` runs ``niter`` fast-gradient-projection dual iterations fully
    on chip.

    Deviations from upstream, both named in DESIGN section 4:

    * ``rtol``: pyproximal's per-image early exit on the relative change of the primal objective (its default 1e-4, which the
      reference's call does not override).  Default here: 0 = off, every image runs ``niter`` dual iterations in ONE fused launch (against
      the reference as configured the trajectory differs by 1.5e-4 rel-L2 and the posterior mean by 8e-5).
      ``rtol > 0`` reproduces the reference's configured behaviour -- every image leaves in the pass upstream's loop leaves it in:
      decided on the device without synchronisation where the full-width pipeline covers the image (128 < W <= 512; every chain runs
      with the pass count it left in at the previous call, the launch leaves the primal objectives of its iterates behind, chains whose
      prediction was wrong run again), pass by pass elsewhere (``exit_path='passes'`` forces that path: one launch per loop pass plus
      one for its objective and a host read after every pass).
    * ``lagged_output``: whether upstream's truncated iterate after ``niter`` loop passes reflects ``niter`` or ``niter - 1`` dual
      updates depends on the loop bound of the un-pinned upstream version; ``False`` (default) = ``niter`` updates,
      ``True`` = ``niter - 1`` (one pipeline stage fewer).
    * ``warm`` (build extension, MYULA samplers only): carry the projected dual from one MYULA iteration to the next, ``niter``
      in {1, 2, 3} updates per MYULA iteration ("K in {1,3} warm-dual")."""

    def __init__(self, dims, sigma=1.0, niter=10, rtol=0.0, step=0.125, momentum="unlocbox", lagged_output=False, warm=False, exit_path="auto"):
        super().__init__(None, False)
        self.dims = (int(dims[0]), int(dims[1]))
        self.sigma = float(sigma)
        self.niter = int(niter)
        self.rtol = float(rtol)
        self.exit_path = {"auto": 0, "device": 0, "passes": 1}[exit_path]
        self.step = float(step)
        self.momentum = momentum
        self.lagged_output = bool(lagged_output)
        self.warm = bool(warm)
        self._prob = None

    def prior_descriptor(self):
        return {"prior_kind": _capi.PRIOR_TV_ISO, "prior_sigma": self.sigma, "tv_niter": self.niter,
                "tv_step": self.step, "tv_betas": fgp_betas(self.niter, self.momentum),
                "tv_lagged_output": self.lagged_output, "tv_warm": self.warm, "tv_rtol": self.rtol, "tv_exit_path": self.exit_path}

    def _problem(self):
        if self._prob is None:
            self._prob = _Problem(self.dims, prior=self.prior_descriptor())
        return self._prob

    def __call__(self, x):
        _, g = self._problem().energies(x)
        return float(g[0]) if g.numel() == 1 else (g if isinstance(x, torch.Tensor) else g.cpu().numpy())

    def prox(self, x, tau):
        return self._problem().eval(x, 0.0, 0.0, 1.0, float(tau))


class ElementwiseProx(ProxOperator):
    """A separable prior given by one of the closed-form proxes of the reference's ``prox.py`` (prox.py:18-85; used inside its samplers at
    prox_lmc.py:106,115 as ``prox.prox_laplace(theta, lamda * alpha)``): ``prox(x, tau) = prox_<kind>(x, *params)`` pixel by pixel, with the
    parameters listed in ``scaled`` multiplied by ``tau`` first.  As ``proxg`` of the MYULA samplers it is evaluated INSIDE the fused step
    kernel (``LMC_PRIOR_EPROX``: row-streaming, register-block and tiled kernels); stand-alone it is ``lmc_prox_elementwise``.

    ``kind``: 'laplace' | 'uncentered_laplace' | 'gaussian' | 'gen_gaussian_4_3' | 'gen_gaussian_3_2' | 'gen_gaussian_3' | 'gen_gaussian_4' |
    'huber' | 'smoothed_laplace' | 'exp' | 'gamma' | 'chi' | 'uniform' | 'triangular' | 'laplace_conj' (parameter order as in prox.py).
    The value g(x) is not defined for every family (the reference only ever uses their proxes): ``__call__`` returns 0."""

    KINDS = {"laplace": (_capi.EPROX_LAPLACE, 1), "uncentered_laplace": (_capi.EPROX_UNCENTERED_LAPLACE, 2), "gaussian": (_capi.EPROX_GAUSSIAN, 1),
             "gen_gaussian_4_3": (_capi.EPROX_GEN_GAUSSIAN_4_3, 1), "gen_gaussian_3_2": (_capi.EPROX_GEN_GAUSSIAN_3_2, 1),
             "gen_gaussian_3": (_capi.EPROX_GEN_GAUSSIAN_3, 1), "gen_gaussian_4": (_capi.EPROX_GEN_GAUSSIAN_4, 1), "huber": (_capi.EPROX_HUBER, 2),
             "smoothed_laplace": (_capi.EPROX_SMOOTHED_LAPLACE, 1), "exp": (_capi.EPROX_EXP, 1), "gamma": (_capi.EPROX_GAMMA, 2),
             "chi": (_capi.EPROX_CHI, 1), "uniform": (_capi.EPROX_UNIFORM, 1), "triangular": (_capi.EPROX_TRIANGULAR, 2),
             "laplace_conj": (_capi.EPROX_LAPLACE_CONJ, 1)}

    def __init__(self, kind, *params, scaled=()):
        super().__init__(None, False)
        if kind not in self.KINDS:
            raise ValueError(f"unknown closed-form prox {kind!r}")
        self.kind = kind
        self.code, n = self.KINDS[kind]
        if len(params) != n:
            raise ValueError(f"prox_{kind} takes {n} parameter(s)")
        self.params = tuple(float(v) for v in params)
        self.scaled = tuple(int(i) for i in scaled)
        if any(i < 0 or i >= n for i in self.scaled):
            raise ValueError("scaled: indices of the parameters that are multiplied by tau")

    def _scaled_params(self, tau):
        return [v * float(tau) if i in self.scaled else v for i, v in enumerate(self.params)]

    def prior_descriptor(self):
        p = self.params + (0.0,) * (2 - len(self.params))
        return {"prior_kind": _capi.PRIOR_EPROX, "eprox_kind": self.code, "eprox_p0": p[0], "eprox_p1": p[1],
                "eprox_scale_mask": sum(1 << i for i in self.scaled)}

    def __call__(self, x):
        return 0.0

    def prox(self, x, tau):
        xt = _dev.to_dev(x)
        out = torch.empty_like(xt)
        par = np.asarray(self._scaled_params(tau), dtype=np.float32)
        _dev.run(xt, "lmc_prox_elementwise", self.code, _dev.ptr(xt), _dev.ptr(out), xt.numel(), _dev.fptr(par), par.size)
        return _dev.like_input(out, x)


def Laplace(lam):
    """``lam * ||x||_1`` through ``prox_laplace(x, tau * lam)`` (prox.py:18; prox_lmc.py:106)."""
    return ElementwiseProx("laplace", lam, scaled=(0,))


def UncenteredLaplace(lam, mu):
    return ElementwiseProx("uncentered_laplace", lam, mu, scaled=(0,))       # prox.py:22


def Gaussian(lam):
    return ElementwiseProx("gaussian", lam, scaled=(0,))                     # prox.py:26: x / (2 tau lam + 1)


def GenGaussian(p, lam):
    """``lam * |x|^p``, p in {4/3, 3/2, 3, 4} (prox.py:30-41)."""
    names = {4 / 3: "gen_gaussian_4_3", 3 / 2: "gen_gaussian_3_2", 3: "gen_gaussian_3", 4: "gen_gaussian_4"}
    if p not in names:
        raise ValueError("p must be one of 4/3, 3/2, 3, 4")
    return ElementwiseProx(names[p], lam, scaled=(0,))


def Huber(gamma, tau):
    return ElementwiseProx("huber", gamma, tau, scaled=(1,))                 # prox.py:44: tau is the prox weight


def SmoothedLaplace(lam):
    return ElementwiseProx("smoothed_laplace", lam, scaled=(0,))             # prox.py:52


class WaveletL1(ProxOperator):
    """``sigma * || detail coefficients of the 3-level orthonormal Haar transform of x ||_1`` -- the prior of BASELINE config 5
    (build-specified; the reference has no wavelet code).  ``prox`` = W^T soft(W x, tau*sigma) on independent 8 x 8 blocks."""

    def __init__(self, dims, sigma=1.0, levels=3):
        super().__init__(None, False)
        if levels != 3:
            raise NotImplementedError("levels=3 (8 x 8 blocks) only")
        self.dims = (int(dims[0]), int(dims[1]))
        if self.dims[0] % 8 or self.dims[1] % 8:
            raise ValueError("image sides must be multiples of 8")
        self.sigma = float(sigma)
        self.levels = levels
        self._prob = None

    def prior_descriptor(self):
        return {"prior_kind": _capi.PRIOR_HAAR_L1, "prior_sigma": self.sigma}

    def __call__(self, x):
        if self._prob is None:
            self._prob = _Problem(self.dims, prior=self.prior_descriptor())
        _, g = self._prob.energies(x)
        return float(g[0]) if g.numel() == 1 else (g if isinstance(x, torch.Tensor) else g.cpu().numpy())

    def prox(self, x, tau):
        xt = _dev.to_dev(x)
        out = torch.empty_like(xt)
        n = self.dims[0] * self.dims[1]
        _dev.run(xt, "lmc_haar_l1_prox", _dev.ptr(xt), _dev.ptr(out), xt.numel() // n, self.dims[0], self.dims[1],
                                                self.sigma * float(tau))
        return _dev.like_input(out, x)


class L2_ncvx_tv(ProxOperator):
    r"""Non-log-concave data term -- drop-in for the reference's own class ``algs.L2_ncvx_tv`` (algs.py:22-291):
    ``f(x) = sigma/2 ||Op x - b||^2 - lamda * env_gamma(g)(Op2 x)``, same constructor arguments.

    Built on the GPU, both isotropic branches used by prox_lmc_deconv.py:106-113:
    * MC-TV (``Op2 = Gradient``): value (algs.py:173-190), gradient (:270-291,
      ``sigma Op^T(Op x - b) - lamda * Op2^T( Op2 x / max(|Op2 x|, gamma) )``) fused into the sampler step, and the implicit
      ``prox`` (:201-267; also inside ULPDA, prox_lmc_deconv.py:478-487);
    * ME-TV (``Op2 = None``): value and gradient ``sigma Op^T(Op x - b) - lamda (x - prox_{gamma TV}(x))/gamma`` (:282), the inner
      TV prox with ``niter`` (= niter_l2 = 50) dual iterations chained exactly through HBM-resident dual state in chunks of 8.
    ``prox`` (:201-267) is built for both (ME-TV pre-step :221-223).  Round 2: the anisotropic MC-TV branches too (``isotropic=False``
    with ``Op2 = Gradient``: :218-219, :278-279 -- the weight of a difference is 1 / max(|that difference|, gamma)).  Round 3: anisotropic
    ME-TV (``isotropic=False``, ``Op2 = None``: a 1-D TV over the flattened image, :170) as plain coverage -- one pass over the images per
    dual iteration of its inner prox, not fused (no model of the reference's driver uses it).
    """

    def __init__(self, dims, Op=None, Op2=None, b=None, q=None, sigma=1., alpha=1., lamda=1., gamma=.5, qgrad=True,
                 isotropic=False, niter=10, rtol=1e-4, x0=None, warm=True, densesolver=None, kwargs_solver=None, lagged_output=False):
        super().__init__(Op, True)
        from .operators import Gradient
        if q is not None:
            raise NotImplementedError("q (linear term) has no device functor")
        if Op2 is not None and not isinstance(Op2, Gradient):
            raise NotImplementedError("Op2 must be a Gradient (MC-TV) or None (ME-TV)")
        from .operators import Identity
        if not isinstance(Op, (Convolve2D, Diagonal, Identity)) or b is None:
            raise NotImplementedError("Op must be a Convolve2D (prox_lmc_deconv.py:106), a Diagonal mask or an Identity, and b given")
        self.dims = (int(dims[0]), int(dims[1]))
        self.Op2 = Op2
        self.b = b
        self.sigma, self.lamda, self.gamma = float(sigma), float(lamda), float(gamma)
        self.isotropic = isotropic
        self.niter = niter
        self.warm = warm
        # the reference hands rtol to its inner TV(dims, 1., niter, rtol) (algs.py:169): that prox is only used by the ME-TV branch, where
        # every image leaves the inner prox in the pass upstream's loop leaves it in (lmc_problem.ncvx_rtol; rtol = 0: always niter updates)
        self.rtol = float(rtol)
        self.lagged_output = bool(lagged_output)
        self._prob = None

    def descriptor(self):
        if isinstance(self.Op, Diagonal):
            base = {"data_kind": _capi.DATA_MASK, "sigma_f": self.sigma, "y": self.b, "mask": self.Op.d}
        elif not isinstance(self.Op, Convolve2D):     # Identity (denoising)
            base = {"data_kind": _capi.DATA_IDENTITY, "sigma_f": self.sigma, "y": self.b}
        else:
            base = {"data_kind": _capi.DATA_BLUR, "sigma_f": self.sigma, "y": self.b, "h": self.Op.h, "offset": self.Op.offset}
        if self.Op2 is None:       # ME-TV: isotropic (2-D TV, algs.py:169) or the 1-D TV of the flattened image (algs.py:170)
            kind = _capi.NCVX_ME_TV if self.isotropic else _capi.NCVX_ME_TV_ANISO
        else:
            kind = _capi.NCVX_MC_TV if self.isotropic else _capi.NCVX_MC_TV_ANISO
        return {**base, "ncvx_kind": kind,
                "ncvx_lambda": self.lamda, "ncvx_gamma": self.gamma, "ncvx_niter": int(self.niter),
                "ncvx_rtol": self.rtol if self.Op2 is None else 0.0,
                "tv_lagged_output": self.lagged_output}

    def _problem(self):
        if self._prob is None:
            self._prob = _Problem(self.dims, data=self.descriptor())
        return self._prob

    def __call__(self, x):
        f, _ = self._problem().energies(x)
        return float(f[0]) if f.numel() == 1 else (f if isinstance(x, torch.Tensor) else f.cpu().numpy())

    def grad(self, x):
        return self._problem().eval(x, 0.0, -1.0, 0.0, 0.0)

    def prox(self, x, tau):
        """``L2_ncvx_tv.prox`` (algs.py:201-267), MC-TV branch: ``v <- x + tau*lamda*Op2^T(Op2 x / max(|Op2 x|, gamma))`` then
        ``(I + tau sigma Op^T Op)^{-1}(v + tau sigma Op^T b)`` by ``niter`` warm-started CG iterations (lmc_l2_prox).  Unlike
        the reference (which adds the first term into its argument in place, :217) the input is left untouched."""
        prob = self._problem()
        n = self.dims[0] * self.dims[1]
        xt = _dev.to_dev(x, prob.device)
        n_img = xt.numel() // n
        lib = _dev.lib()
        x0 = getattr(self, "_x0", None)
        if self.warm and x0 is not None and x0.numel() == xt.numel():
            out, warm = x0.clone(), 1
        else:
            out, warm = torch.empty_like(xt), 0
        ws = torch.empty(lib.lmc_l2_prox_workspace_bytes(n_img, self.dims[0], self.dims[1]), dtype=torch.uint8, device=xt.device)
        with torch.cuda.device(prob.device):
            _capi.check(lib.lmc_l2_prox(C.byref(prob.c), _dev.ptr(xt), _dev.ptr(out), n_img, float(tau), int(self.niter), warm,
                                        _dev.ptr(ws), _dev.stream_ptr()))
            torch.cuda.current_stream().synchronize()
        if self.warm:
            self._x0 = out.clone()
        return _dev.like_input(out.reshape(xt.shape), x)
