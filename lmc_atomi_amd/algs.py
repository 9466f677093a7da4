"""Sampler entry points -- drop-ins for the reference's ``algs.py``.

``MoreauYosidaUnadjustedLangevin`` keeps the reference signature (algs.py:477-478) and return
value (``ndarray (niter, n)`` of all iterates) for one chain, and adds keyword-only extensions
for what the reference lacks: many chains per GPU, sharding-invariant counter-based noise,
posterior moments with burn-in / thinning, per-chain energy diagnostics.

The per-iteration update runs entirely in hand-written HIP kernels behind the C ABI
(include/lmc_atomi.h); torch tensors are HBM containers only.
"""
from __future__ import annotations

import ctypes as C
import time

import numpy as np
import torch
from numpy.random import default_rng

from . import _capi, _dev
from .proximal import _Problem


def _prior_descriptor(proxg):
    if proxg is None:
        return {"prior_kind": _capi.PRIOR_NONE}
    fn = getattr(proxg, "prior_descriptor", None)
    if fn is None:
        raise NotImplementedError(
            f"{type(proxg).__name__} has no device functor (prior_descriptor()); new proxes are added as "
            "functors compiled into liblmc_atomi -- there is no CPU fallback")
    return fn()


def _data_descriptor(proxf):
    if proxf is None:
        return {"data_kind": _capi.DATA_NONE}
    fn = getattr(proxf, "descriptor", None)
    if fn is None:
        raise NotImplementedError(f"{type(proxf).__name__} has no device functor (descriptor())")
    return fn()


class MYULASampler:
    """Many-chain MYULA on one GPU: owns an ``lmc_sampler`` handle.

    State layout in HBM: ``[n_chains, H, W]`` fp32.  ``chain_offset`` is the global id of local
    chain 0; the noise of a chain depends only on (seed, iteration, global chain id, pixel), so
    any sharding of the chains over GPUs reproduces the same trajectories.
    """

    def __init__(self, proxf, proxg, dims, n_chains=1, tau=None, gamma=0.1, epsg=1.0, seed=0,
                 chain_offset=0, noise="philox", moments=False, burn_in=0, thin=1, device=None, variant=None, tv_warm=None, policy=None):
        """``variant``: step-kernel variant of THIS sampler ('auto' | 'tile' | 'split' | 'point' | 'block' | 'rows' | 'pipe'; None = the
        library default, :func:`set_step_variant`).  ``tv_warm``: carry the TV dual between iterations (see :class:`TV`; None = as
        ``proxg.warm`` says).  Every call on the sampler runs on ``device`` whatever the current device is."""
        if tau is None:
            raise NotImplementedError("tau=None (backtracking) is not implemented by the reference loop either")
        self.dims = (int(dims[0]), int(dims[1]))
        self.n_chains = int(n_chains)
        self.device = _dev.device(device)
        self.proxf, self.proxg = proxf, proxg
        opts = {"step_variant": variant or 0}
        if tv_warm is not None:
            opts["tv_warm"] = bool(tv_warm)
        # launch policy of this sampler (lmc_problem, ABI 3): dict with any of iterations_per_launch (0 auto / 1 / 2), moments_overlap (0 auto /
        # 1 / -1), moments_bg_workgroups, graph_replay, tv_exit_path (1 = the pass-by-pass early exit)
        opts.update(policy or {})
        self.epsg = epsg
        if np.asarray(epsg).size > 1:      # array-valued epsg (algs.py:509,539-542): the prox parameter epsg * gamma is an array that the prox broadcasts
            opts["prox_scale"] = self._epsg_array(epsg)
            epsg = 1.0
        self._problem = _Problem(self.dims, _data_descriptor(proxf), _prior_descriptor(proxg), self.device, options=opts)
        cfg = _capi.lmc_myula_config()
        cfg.struct_size = C.sizeof(_capi.lmc_myula_config)
        cfg.problem = self._problem.c
        cfg.n_chains = self.n_chains
        cfg.chain_offset = int(chain_offset)
        cfg.tau, cfg.gamma, cfg.epsg = float(tau), float(gamma), float(epsg)
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.noise_mode = {"philox": _capi.NOISE_PHILOX, "injected": _capi.NOISE_INJECTED, "none": _capi.NOISE_NONE}[noise]
        cfg.moments = 1 if moments else 0
        cfg.burn_in = int(burn_in)
        cfg.thin = int(thin)
        self.noise_mode = noise
        self.moments_on = bool(moments)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _capi.check(getattr(_dev.lib(), self._create_fn)(C.byref(cfg), C.byref(self._h)))

    _create_fn = "lmc_myula_create"

    def _epsg_array(self, epsg):
        """Device copy + (chain, pixel) strides of an array-valued ``epsg``.  The reference hands ``epsg * gamma`` to ``proxg.prox`` (algs.py:569), whose
        closed forms broadcast it against ``x``: one weight per pixel of a flattened image (``x`` of shape ``(n,)``), one per right-hand side = per chain
        (``x`` of shape ``(n, nrhs)``, ``epsg`` of shape ``(nrhs,)``), or both.  Here: ``(H*W,)`` / ``(H, W)`` per pixel, ``(n_chains,)`` per chain,
        ``(n_chains, H*W)`` / ``(n_chains, H, W)`` both."""
        e = np.asarray(epsg, dtype=np.float32)
        n = self.dims[0] * self.dims[1]
        if e.size == n and e.shape in ((n,), self.dims):
            cs, ps = 0, 1
        elif e.shape == (self.n_chains,):
            cs, ps = 1, 0
        elif e.size == self.n_chains * n and e.shape[0] == self.n_chains:
            cs, ps = n, 1
        else:
            raise ValueError(f"epsg of shape {e.shape} matches neither the image {self.dims}, nor the {self.n_chains} chains, nor both")
        return _dev.to_dev(np.ascontiguousarray(e.ravel()), self.device), cs, ps

    # -- lifetime ------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            _dev.lib().lmc_sampler_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- state ---------------------------------------------------------------------------
    @property
    def shape(self):
        return (self.n_chains,) + self.dims

    def set_state(self, x):
        xt = _dev.to_dev(x, self.device)
        n = self.dims[0] * self.dims[1]
        if xt.numel() == n:                       # one image: broadcast to every chain (x0 of algs.py:559)
            xt = xt.reshape(1, *self.dims).expand(self.shape).contiguous()
        if xt.numel() != self.n_chains * n:
            raise ValueError(f"state of shape {tuple(xt.shape)} does not match {self.shape}")
        _capi.check(_dev.lib().lmc_sampler_set_state(self._h, _dev.ptr(xt), _dev.stream_ptr(self.device)))
        torch.cuda.current_stream(self.device).synchronize()  # xt may be a temporary

    def get_state(self, out=None):
        if out is None:
            out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        _capi.check(_dev.lib().lmc_sampler_get_state(self._h, _dev.ptr(out), _dev.stream_ptr(self.device)))
        return out

    @property
    def iteration(self):
        return int(_dev.lib().lmc_sampler_iteration(self._h))

    @iteration.setter
    def iteration(self, it):
        _capi.check(_dev.lib().lmc_sampler_set_iteration(self._h, int(it)))

    # -- hot loop ------------------------------------------------------------------------
    def step(self, n_iters=1, noise=None):
        """Run ``n_iters`` iterations of algs.py:564-570 on every chain.  ``noise`` (only with
        noise='injected'): ``[n_iters, n_chains, H, W]``."""
        nt = None
        if noise is not None:
            nt = _dev.to_dev(noise, self.device)
            if nt.numel() != n_iters * self.n_chains * self.dims[0] * self.dims[1]:
                raise ValueError("noise must have shape [n_iters, n_chains, H, W]")
        _capi.check(_dev.lib().lmc_sampler_step(self._h, int(n_iters), _dev.ptr(nt), _dev.stream_ptr(self.device)))
        if nt is not None:
            torch.cuda.current_stream(self.device).synchronize()

    def enable_timing(self, on=True):
        """Bracket every step-kernel launch with a HIP event pair on the launch stream."""
        _capi.check(_dev.lib().lmc_sampler_enable_timing(self._h, 1 if on else 0))

    def last_step_timing(self):
        """(summed step-kernel milliseconds, launches) of the last :meth:`step` call."""
        ms, n = C.c_float(), C.c_int32()
        _capi.check(_dev.lib().lmc_sampler_last_step_timing(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @property
    def kernel_name(self):
        return _dev.lib().lmc_sampler_kernel_name(self._h).decode()

    def tv_exit_stats(self, which="prior"):
        """Early-exit statistics of the device path (``TV(rtol > 0)`` / ``L2_ncvx_tv(rtol > 0)``): ``(passes, reruns)`` -- the loop pass each
        chain's latest prox left in (int32 tensor, ``niter`` = it ran out of passes) and the number of chain runs that had to be repeated
        after rounds 1 / 2 / 3 / 4 since the sampler was created (``which``: 'prior' = the TV prior's prox, 'ncvx' = the ME-TV inner prox)."""
        passes = torch.empty(self.n_chains, dtype=torch.int32, device=self.device)
        rr = (C.c_uint64 * 4)()
        _capi.check(_dev.lib().lmc_sampler_tv_exit_stats(self._h, {"prior": 0, "ncvx": 1}[which], _dev.ptr(passes), rr, _dev.stream_ptr(self.device)))
        return passes, [int(v) for v in rr]

    # -- diagnostics ---------------------------------------------------------------------
    def energies(self):
        """Per-chain ``f(x_c)``, ``g(x_c)`` (float64 tensors in HBM) -- the energy log of algs.py:578-582."""
        f = torch.empty(self.n_chains, dtype=torch.float64, device=self.device)
        g = torch.empty(self.n_chains, dtype=torch.float64, device=self.device)
        _capi.check(_dev.lib().lmc_sampler_energies(self._h, _dev.ptr(f), _dev.ptr(g), _dev.stream_ptr(self.device)))
        return f, g

    def noise_field(self, iteration):
        out = torch.empty(self.shape, dtype=torch.float32, device=self.device)
        _capi.check(_dev.lib().lmc_sampler_noise(self._h, int(iteration), _dev.ptr(out), _dev.stream_ptr(self.device)))
        return out

    def moments(self):
        """(sum [H,W] f64, sumsq [H,W] f64, count) over chains and kept iterations."""
        s1 = torch.empty(self.dims, dtype=torch.float64, device=self.device)
        s2 = torch.empty(self.dims, dtype=torch.float64, device=self.device)
        cnt = C.c_uint64()
        _capi.check(_dev.lib().lmc_sampler_get_moments(self._h, _dev.ptr(s1), _dev.ptr(s2), C.byref(cnt),
                                                       _dev.stream_ptr(self.device)))
        return s1, s2, int(cnt.value)

    def reset_moments(self):
        _capi.check(_dev.lib().lmc_sampler_reset_moments(self._h, _dev.stream_ptr(self.device)))

    def allreduce_moments(self, rccl_comm):
        """Job-wide (sum, sumsq, count): ONE ``ncclAllReduce`` (RCCL over xGMI) of the packed accumulators through the C ABI
        (``lmc_allreduce_moments``).  ``rccl_comm``: an ``ncclComm_t`` as an integer / ``c_void_p`` (``None`` or 0 = a job of one rank)."""
        s1 = torch.empty(self.dims, dtype=torch.float64, device=self.device)
        s2 = torch.empty(self.dims, dtype=torch.float64, device=self.device)
        cnt = C.c_uint64()
        comm = rccl_comm if isinstance(rccl_comm, C.c_void_p) else C.c_void_p(int(rccl_comm or 0))
        _capi.check(_dev.lib().lmc_allreduce_moments(self._h, comm, _dev.ptr(s1), _dev.ptr(s2), C.byref(cnt),
                                                     _dev.stream_ptr(self.device)))
        return s1, s2, int(cnt.value)


class ULPDASampler(MYULASampler):
    """Many-chain ULPDA on one GPU (algs.py:425-449).  ``proxf`` = data term (L2 with Convolve2D / Diagonal / Identity /
    no operator), ``proxg`` = L21 (isotropic) or L1 (anisotropic) acting on ``A x`` with ``A`` the forward-difference
    gradient.  The implicit data step runs ``proxf.niter`` warm-started CG iterations per chain on the GPU."""

    def __init__(self, proxf, proxg, A, dims, n_chains=1, tau=None, mu=None, theta=1.0, gfirst=True, z=None, seed=0,
                 chain_offset=0, noise="philox", moments=False, burn_in=0, thin=1, device=None, variant=None, implicit_tol=None):
        from .operators import Gradient
        from .proximal import L1, L21
        if not isinstance(A, Gradient):
            raise NotImplementedError("ULPDA on the GPU supports A = Gradient (the reference's operator, prox_lmc_deconv.py:98)")
        if isinstance(proxg, L21):
            prior = {"prior_kind": _capi.PRIOR_TV_ISO, "prior_sigma": proxg.sigma, "tv_niter": 1, "tv_betas": [0.0]}
        elif isinstance(proxg, L1):
            prior = {"prior_kind": _capi.PRIOR_TV_ANISO, "prior_sigma": proxg.sigma}
        else:
            raise NotImplementedError(f"{type(proxg).__name__} has no dual-prox device functor (L21 or L1 expected)")
        self.dims = (int(dims[0]), int(dims[1]))
        self.n_chains = int(n_chains)
        self.device = _dev.device(device)
        self.proxf, self.proxg = proxf, proxg
        # implicit_tol: relative residual of THIS sampler's implicit data step (None = the library default, set_cg_tolerance;
        # 0 or negative = disabled: always all iterations)
        tol = 0.0 if implicit_tol is None else (float(implicit_tol) if implicit_tol > 0 else -1.0)
        self._problem = _Problem(self.dims, _data_descriptor(proxf), prior, self.device,
                                 options={"step_variant": variant or 0, "implicit_tol": tol})
        cfg = _capi.lmc_ulpda_config()
        cfg.struct_size = C.sizeof(_capi.lmc_ulpda_config)
        cfg.problem = self._problem.c
        cfg.n_chains = self.n_chains
        cfg.chain_offset = int(chain_offset)
        cfg.tau, cfg.mu, cfg.theta = float(tau), float(mu), float(theta)
        cfg.gfirst = 1 if gfirst else 0
        cfg.cg_niter = int(getattr(proxf, "niter", 10) or 10)
        cfg.warm = 1 if getattr(proxf, "warm", True) else 0
        self._z = None
        if z is not None:
            self._z = _dev.to_dev(z, self.device).reshape(self.dims)
            cfg.z_dev = self._z.data_ptr()
        cfg.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        cfg.noise_mode = {"philox": _capi.NOISE_PHILOX, "injected": _capi.NOISE_INJECTED, "none": _capi.NOISE_NONE}[noise]
        cfg.moments = 1 if moments else 0
        cfg.burn_in = int(burn_in)
        cfg.thin = int(thin)
        self.noise_mode = noise
        self.moments_on = bool(moments)
        self._h = C.c_void_p()
        with torch.cuda.device(self.device):
            _capi.check(_dev.lib().lmc_ulpda_create(C.byref(cfg), C.byref(self._h)))

    def set_steps(self, tau, mu):
        _capi.check(_dev.lib().lmc_sampler_set_steps(self._h, float(tau), float(mu)))

    def set_dual(self, y):
        yt = _dev.to_dev(y, self.device)
        n2 = 2 * self.dims[0] * self.dims[1]
        if yt.numel() == n2:
            yt = yt.reshape(1, n2).expand(self.n_chains, n2).contiguous()
        if yt.numel() != self.n_chains * n2:
            raise ValueError("dual state must have 2*H*W entries per chain")
        _capi.check(_dev.lib().lmc_sampler_set_dual(self._h, _dev.ptr(yt), _dev.stream_ptr(self.device)))
        torch.cuda.current_stream(self.device).synchronize()

    def get_dual(self):
        out = torch.empty((self.n_chains, 2) + self.dims, dtype=torch.float32, device=self.device)
        _capi.check(_dev.lib().lmc_sampler_get_dual(self._h, _dev.ptr(out), _dev.stream_ptr(self.device)))
        return out


def UnadjustedLangevinPrimalDual(proxf, proxg, A, x0, tau, mu, y0=None, z=None, theta=1., niter=10, seed=0, gfirst=True,
                                 callback=None, callbacky=False, returny=False, show=False, *, n_chains=None, dims=None,
                                 rng="philox", chain_offset=0, burn_in=0, thin=1, device=None, diagnostics=None):
    r"""Unadjusted Langevin Primal-Dual algorithm (ULPDA) -- drop-in for algs.py:295-474.

    Reference form (``n_chains is None``): one chain, returns ``np.ndarray (niter, n)`` (and the duals ``(niter, 2n)`` with
    ``returny``), ``callback(x)`` / ``callback(x, y)`` every iteration, ``tau`` / ``mu`` scalars or per-iteration arrays
    (algs.py:402-408).  ``rng='pcg64'`` injects the reference's noise stream.  Many-chain form: :class:`MYULAResult`
    (``diagnostics=(ph, pw)`` or ``True``: split R-hat / ESS across chains as in :func:`MoreauYosidaUnadjustedLangevin`).
    """
    if dims is None:
        dims = getattr(A, "dims", None) or getattr(proxf, "dims", None)
    if dims is None:
        raise ValueError("image shape unknown: pass dims=(ny, nx)")
    many = n_chains is not None
    C_ = int(n_chains) if many else 1
    n = int(dims[0]) * int(dims[1])
    if rng not in ("philox", "pcg64"):
        raise ValueError("rng must be 'philox' or 'pcg64'")
    if rng == "pcg64" and C_ != 1:
        raise ValueError("rng='pcg64' reproduces the reference's single chain; use n_chains=None")
    taus = np.full(niter, tau, dtype=np.float64) if np.isscalar(tau) else np.asarray(tau, dtype=np.float64)
    mus = np.full(niter, mu, dtype=np.float64) if np.isscalar(mu) else np.asarray(mu, dtype=np.float64)
    smp = ULPDASampler(proxf, proxg, A, dims, n_chains=C_, tau=taus[0], mu=mus[0], theta=theta, gfirst=gfirst, z=z,
                       seed=seed, chain_offset=chain_offset, noise="injected" if rng == "pcg64" else "philox",
                       moments=many, burn_in=burn_in, thin=thin, device=device)
    try:
        smp.set_state(x0)
        if y0 is not None:
            smp.set_dual(y0)
        tstart = time.time()
        if show:
            print('Unadjusted Langevin primal-dual (lmc_atomi_amd / HIP): U(x) = f(x) + x^T z + g(Ax)\n'
                  '---------------------------------------------------------\n'
                  'Proximal operator (f): %s\nProximal operator (g): %s\nLinear operator (A): %s\n'
                  'tau = %s\t\tmu = %s\ntheta = %.2f\t\tniter = %d\tchains = %d\n' %
                  (type(proxf), type(proxg), type(A), str(taus[0]), str(mus[0]), theta, niter, C_))
            print('   Itn       x[0]          f         g o A      U = f + g o A')
        host_rng = default_rng(seed) if rng == "pcg64" else None
        xs = np.empty((niter, n), dtype=np.float64) if not many else None
        ys = np.empty((niter, 2 * n), dtype=np.float64) if (returny and not many) else None
        tracer = None
        if diagnostics and many:
            from .diagnostics import ChainTrace
            tracer = ChainTrace(smp, (8, 8) if diagnostics is True else diagnostics)
        for it in range(niter):
            smp.set_steps(taus[it], mus[it])
            if host_rng is not None:
                xi = host_rng.standard_normal(n)                        # algs.py:433
                smp.step(1, noise=xi.reshape(1, 1, *smp.dims))
            else:
                smp.step(1)
            if not many:
                xs[it] = smp.get_state().reshape(-1).cpu().numpy()
                if returny or callbacky:
                    yk = smp.get_dual().reshape(-1).cpu().numpy()
                    if returny:
                        ys[it] = yk
                if callback is not None:
                    callback(xs[it], yk) if callbacky else callback(xs[it])
            elif callback is not None:
                callback(smp.get_state(), smp.get_dual()) if callbacky else callback(smp.get_state())
            if tracer is not None and it >= burn_in and (it - burn_in) % thin == 0:      # the iterations that enter the moments
                tracer.record()
            if show and (it < 10 or niter - it < 10 or it % max(niter // 10, 1) == 0):
                f, g = smp.energies()
                x00 = float(smp.get_state().reshape(-1)[0])
                print('%6g  %12.5e  %10.3e  %10.3e      %10.3e' % (it + 1, x00, float(f.mean()), float(g.mean()),
                                                                  float((f + g).mean())))
        if show:
            print('\nTotal time (s) = %.2f' % (time.time() - tstart))
            print('---------------------------------------------------------\n')
        if not many:
            return (xs, ys) if returny else xs
        s1, s2, cnt = smp.moments()
        f, g = smp.energies()
        state = smp.get_state()
        torch.cuda.current_stream().synchronize()
        mean, var = mean_var_from_moments(s1, s2, max(cnt, 1))
        diag = tracer.summary() if tracer is not None and len(tracer) else None
        return MYULAResult(state, mean, var, cnt, f, g, time.time() - tstart, diagnostics=diag,
                           trace=tracer.trace() if diag is not None else None)
    finally:
        smp.close()


def mean_var_from_moments(s1, s2, count):
    """Posterior mean and pixel-wise variance from accumulated sums (any array type)."""
    mean = s1 / count
    var = s2 / count - mean * mean
    return mean, var


class MYULAResult:
    """Return value of the many-chain form of :func:`MoreauYosidaUnadjustedLangevin`."""

    def __init__(self, state, mean, var, count, energy_f, energy_g, elapsed, diagnostics=None, trace=None):
        self.state, self.mean, self.var, self.count = state, mean, var, count
        self.energy_f, self.energy_g, self.elapsed = energy_f, energy_g, elapsed
        self.diagnostics, self.trace = diagnostics, trace      # split R-hat / ESS across chains (diagnostics.py), [T, C, Q] trace


def MoreauYosidaUnadjustedLangevin(proxf, proxg, x0, tau=None, gamma=.1, epsg=1., niter=10, seed=0,
                                   callback=None, show=False, *, n_chains=None, dims=None, rng="philox",
                                   chain_offset=0, burn_in=0, thin=1, device=None, diagnostics=None):
    r"""Moreau--Yosida Unadjusted Langevin algorithm (MYULA) -- drop-in for algs.py:477-587.

    .. math::
        x^{k+1} = (1-\tau/\gamma)x^k - \tau\nabla f(x^k) + (\tau/\gamma)\,prox_{\gamma\epsilon g}(x^k)
                  + \sqrt{2\tau}\,\xi^k

    Reference form (``n_chains is None``): one chain, returns ``np.ndarray (niter, n)`` holding
    every iterate, ``callback(x)`` after every iteration, ``show`` prints the reference's log.
    ``rng='pcg64'`` draws the noise exactly as the reference does (``default_rng(seed)``, one
    ``standard_normal(n)`` per iteration, algs.py:561,565) and injects it, so the trajectory
    equals the reference's to fp32 rounding; ``rng='philox'`` (default) draws on the GPU.

    Many-chain form (``n_chains=C``): runs C chains from ``x0`` (one image or ``[C,H,W]``), keeps
    no iterates, returns a :class:`MYULAResult` (final states, posterior mean / variance over
    chains and kept iterations, per-chain energies).  ``diagnostics=(ph, pw)`` (or ``True`` = (8, 8)) additionally records,
    at every kept iteration, a ph x pw grid of block means and the energies of every chain and returns split R-hat and
    effective sample size across chains in ``result.diagnostics`` (:mod:`lmc_atomi_amd.diagnostics`).
    """
    if dims is None:
        dims = getattr(proxf, "dims", None) or getattr(proxg, "dims", None)
    if dims is None:
        raise ValueError("image shape unknown: pass dims=(ny, nx)")
    many = n_chains is not None
    C_ = int(n_chains) if many else 1
    n = int(dims[0]) * int(dims[1])
    if rng not in ("philox", "pcg64"):
        raise ValueError("rng must be 'philox' or 'pcg64'")
    if rng == "pcg64" and C_ != 1:
        raise ValueError("rng='pcg64' reproduces the reference's single chain; use n_chains=None")
    smp = MYULASampler(proxf, proxg, dims, n_chains=C_, tau=tau, gamma=gamma, epsg=epsg, seed=seed,
                       chain_offset=chain_offset, noise="injected" if rng == "pcg64" else "philox",
                       moments=many, burn_in=burn_in, thin=thin, device=device)
    try:
        smp.set_state(x0)
        tstart = time.time()
        if show:
            print('Moreau--Yosida Unadjusted Langevin (lmc_atomi_amd / HIP)\n'
                  '---------------------------------------------------------\n'
                  'Proximal operator (f): %s\nProximal operator (g): %s\n'
                  'tau = %s\tgamma=%10e\nepsg = %s\tniter = %d\tchains = %d\n' %
                  (type(proxf), type(proxg), str(tau), gamma, str(epsg) if np.asarray(epsg).size == 1 else 'Multi', niter, C_))     # algs.py:539-542
            print('   Itn       x[0]          f           g     J = f + eps*g')
        if not many:
            samples = np.empty((niter, n), dtype=np.asarray(x0).dtype if not isinstance(x0, torch.Tensor) else np.float32)
            host_rng = default_rng(seed) if rng == "pcg64" else None
            buf = torch.empty(smp.shape, dtype=torch.float32, device=smp.device)
            for it in range(niter):
                if host_rng is not None:
                    xi = host_rng.standard_normal(n)                  # algs.py:565
                    smp.step(1, noise=xi.reshape(1, 1, *smp.dims))
                else:
                    smp.step(1)
                smp.get_state(buf)
                xk = buf.reshape(-1).cpu().numpy()
                samples[it] = xk
                if callback is not None:
                    callback(samples[it])
                if show and (it < 10 or niter - it < 10 or it % max(niter // 10, 1) == 0):
                    f, g = smp.energies()
                    pf, pg = float(f[0]), float(g[0])
                    print('%6g  %12.5e  %10.3e  %10.3e  %10.3e' % (it + 1, samples[it][0], pf, pg, pf + float(np.sum(epsg * pg))))   # algs.py:582
            if show:
                print('\nTotal time (s) = %.2f' % (time.time() - tstart))
                print('---------------------------------------------------------\n')
            return samples
        # many chains: no iterates kept
        tracer = None
        if diagnostics:
            from .diagnostics import ChainTrace
            tracer = ChainTrace(smp, (8, 8) if diagnostics is True else diagnostics)
        next_rec = burn_in + 1                       # iteration counts after which the sampler has accumulated moments
        done = 0
        while done < niter:
            chunk = niter - done if (callback is None and not show) else 1
            if tracer is not None and next_rec > done:
                chunk = min(chunk, next_rec - done)
            smp.step(chunk)
            done += chunk
            if tracer is not None and done == next_rec:
                tracer.record()
                next_rec += thin
            if callback is not None:
                callback(smp.get_state())
            if show and (done <= 10 or niter - done < 10 or (done - 1) % max(niter // 10, 1) == 0):
                f, g = smp.energies()
                x00 = float(smp.get_state()[0, 0, 0])
                print('%6g  %12.5e  %10.3e  %10.3e  %10.3e' % (done, x00, float(f.mean()), float(g.mean()),
                                                              float((f + (epsg if np.asarray(epsg).size == 1 else float(np.sum(epsg))) * g).mean())))
        s1, s2, cnt = smp.moments()
        f, g = smp.energies()
        state = smp.get_state()
        torch.cuda.current_stream().synchronize()
        mean, var = mean_var_from_moments(s1, s2, max(cnt, 1))
        diag = tracer.summary() if tracer is not None and len(tracer) else None
        return MYULAResult(state, mean, var, cnt, f, g, time.time() - tstart, diagnostics=diag,
                           trace=tracer.trace() if diag is not None else None)
    finally:
        smp.close()


class MYMALASampler(MYULASampler):
    """Metropolis-adjusted MYULA (MYMALA) for many chains at image scale: the accept / reject of the reference's toy
    ``ProximalLangevinMonteCarlo.mymala`` (prox_lmc.py:134-158) generalised to ``[C, H, W]`` states, everything on the device.
    Same constructor as :class:`MYULASampler`; a rejected chain keeps its state (and is counted again by the moments)."""

    _create_fn = "lmc_mymala_create"

    def acceptance(self):
        """(accepted proposals per chain [C] int64 tensor, log acceptance ratio of the last iteration [C] float64 tensor)."""
        acc = torch.empty(self.n_chains, dtype=torch.int64, device=self.device)
        la = torch.empty(self.n_chains, dtype=torch.float64, device=self.device)
        _capi.check(_dev.lib().lmc_sampler_get_acceptance(self._h, _dev.ptr(acc), _dev.ptr(la), _dev.stream_ptr(self.device)))
        return acc, la

    def acceptance_rate(self):
        acc, _ = self.acceptance()
        return acc.double() / max(self.iteration, 1)


def MoreauYosidaMetropolisAdjustedLangevin(proxf, proxg, x0, tau=None, gamma=.1, epsg=1., niter=10, seed=0, callback=None, *,
                                           n_chains=1, dims=None, chain_offset=0, burn_in=0, thin=1, device=None):
    """MYMALA at image scale for ``n_chains`` chains (the accept / reject of prox_lmc.py:134-158 around the MYULA move of
    algs.py:569): returns a :class:`MYULAResult` with two extra attributes, ``accepted`` (per-chain counts) and
    ``acceptance_rate``.  ``callback(state)`` after every iteration if given."""
    if dims is None:
        dims = getattr(proxf, "dims", None) or getattr(proxg, "dims", None)
    if dims is None:
        raise ValueError("image shape unknown: pass dims=(ny, nx)")
    smp = MYMALASampler(proxf, proxg, dims, n_chains=int(n_chains), tau=tau, gamma=gamma, epsg=epsg, seed=seed,
                        chain_offset=chain_offset, moments=True, burn_in=burn_in, thin=thin, device=device)
    try:
        smp.set_state(x0)
        tstart = time.time()
        if callback is None:
            smp.step(niter)
        else:
            for _ in range(niter):
                smp.step(1)
                callback(smp.get_state())
        s1, s2, cnt = smp.moments()
        f, g = smp.energies()
        state = smp.get_state()
        acc, _ = smp.acceptance()
        torch.cuda.current_stream().synchronize()
        mean, var = mean_var_from_moments(s1, s2, max(cnt, 1))
        res = MYULAResult(state, mean, var, cnt, f, g, time.time() - tstart)
        res.accepted = acc
        res.acceptance_rate = acc.double() / max(niter, 1)
        return res
    finally:
        smp.close()


def set_cg_tolerance(tol=1e-6):
    """Relative residual at which the inner CG solver of the implicit data step stops early (the reference's solver, scipy
    lsqr, stops at btol = 1e-6 by default, algs.py:250); 0 = always run ``niter`` iterations.  Returns the previous value."""
    return float(_dev.lib().lmc_set_cg_tolerance(float(tol)))


def set_step_variant(variant="auto"):
    """Library-wide DEFAULT of the step-kernel variant ('auto' | 'tile' | 'split' | 'point' | 'block' | 'rows' | 'pipe'); returns the previous
    one.  Process-global, for A/B tests and profiles; a sampler's own ``variant=`` argument takes precedence.  All compute the same update."""
    names = _capi.VARIANTS
    if variant not in names or variant.startswith("("):
        raise ValueError(f"unknown step-kernel variant {variant!r} (the one-group 'stream' kernel of ABI 1 was removed)")
    prev = _dev.lib().lmc_set_step_variant(names.index(variant))
    if prev < 0:
        _capi.check(prev)
    return names[prev]
