"""CPU oracle for the LMC hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg
may import this module.  The product path (``lmc_atomi_amd``) never imports it and
fails loudly when the HIP library is missing.

This file is a plain numpy restatement of the reference's Langevin-Monte-Carlo inner
loop (192459/lmc-atomi, snapshot 2024-08-07).  Every function cites the reference
file:line it follows.  Citations are relative to the reference root.

Parity status
-------------
* Sampler recursions, step formulae, RNG consumption order, output layout
  (``algs.py:425-449``, ``algs.py:559-570``, ``lmc.py:94-104``, ``prox_lmc.py:99-130``), the Metropolis
  accept / reject rule of ``prox_lmc.py:134-158`` (``toy_mymala``; its image-scale generalisation ``mymala_batched``
  uses the same rule in log form) and the closed-form proxes (``prox.py:9-65``) are PINNED: ``tests/golden/*.npz`` hold
  outputs of the reference's own code executed in the build container
  (``tests/golden/make_golden.py``) and ``tests/test_oracle_golden.py`` checks this
  restatement against them bit for bit / to 1e-12.
* The operator arithmetic the reference delegates to third-party ``pylops`` /
  ``pyproximal`` (no version pinned: only ``pip install -U ... pylops pyproximal``,
  ``README.md:5``; neither library is in /root/reference nor installable here) is
  restated from the libraries' documented semantics and is **parity unpinned**:
  ``Convolve2D``, ``Gradient``, ``L2``, ``L1``, ``L21``, ``TV``.  The reference's own
  tests pin nothing there (``test_pyprox.py`` has no assertions).  These are checked by
  adjoint dot-tests, against ``scipy.signal`` and against an independent converged TV
  solver (``skimage.restoration.denoise_tv_chambolle`` fixture in ``tests/golden``).
* The Haar-l1 wavelet prior (BASELINE config 5) has no counterpart in the reference; it is build-specified
  (3-level orthonormal Haar, soft threshold of the detail coefficients) and checked against PyWavelets
  (``haar_pywt.npz``, made with the conda interpreter).

dtype: every routine computes in the dtype of its input (float64 = the reference's
dtype; float32 = the device dtype, used for per-step parity with injected noise).
All image routines accept leading batch (chain) dimensions: ``x[..., H, W]``.
"""
from __future__ import annotations

import math
import numpy as np
from numpy.random import default_rng

__all__ = [
    "Convolve2D", "Gradient", "Identity", "Diagonal",
    "L2", "L1", "L21", "TV", "L2NcvxTV", "WaveletL1", "haar_fwd", "haar_inv", "haar_l1_prox", "haar_l1_value",
    "blur", "blur_adjoint", "grad2d", "div2d", "tv_value", "tv_prox_fgp", "fgp_betas",
    "myula", "ulpda", "myula_batched", "myula_step",
    "philox4x32_10", "philox_normals", "box_muller",
    "toy_ula", "toy_myula", "toy_pgld", "toy_mymala", "mymala_batched", "energies", "philox_uniforms",
]

# ----------------------------------------------------------------------------------
# Linear operators (pylops protocol: matvec / rmatvec / shape / H / dtype / explicit)
# ----------------------------------------------------------------------------------


def blur(x, h, offset):
    """Zero-padded "same" 2-D convolution with origin ``offset`` of kernel ``h``.

    ``(Hx)[i,j] = sum_{a,b} h[a,b] * x[i-a+oy, j-b+ox]`` with x = 0 outside the image.
    Follows ``pylops.signalprocessing.Convolve2D((ny,nx), h, offset)`` as constructed at
    ``prox_lmc_deconv.py:55-69`` [upstream semantics, parity unpinned]: the kernel is
    embedded so that ``offset`` is its centre and convolved with ``mode='same'``; even
    kernels (6x6, offset 3) are therefore off-centre.
    """
    x = np.asarray(x)
    h = np.asarray(h, dtype=x.dtype)
    kh, kw = h.shape
    oy, ox = offset
    H, W = x.shape[-2:]
    out = np.zeros_like(x)
    for a in range(kh):
        for b in range(kw):
            dy, dx = oy - a, ox - b          # out[i,j] += h[a,b] * x[i+dy, j+dx]
            i0, i1 = max(0, -dy), min(H, H - dy)
            j0, j1 = max(0, -dx), min(W, W - dx)
            if i0 >= i1 or j0 >= j1:
                continue
            out[..., i0:i1, j0:j1] += h[a, b] * x[..., i0 + dy:i1 + dy, j0 + dx:j1 + dx]
    return out


def blur_adjoint(r, h, offset):
    """Adjoint of :func:`blur`: ``(H^T r)[m,n] = sum_{a,b} h[a,b] * r[m+a-oy, n+b-ox]``
    (correlation; r = 0 outside the image).  Used by ``L2.grad`` (``algs.py:283-284``)."""
    r = np.asarray(r)
    h = np.asarray(h, dtype=r.dtype)
    kh, kw = h.shape
    oy, ox = offset
    H, W = r.shape[-2:]
    out = np.zeros_like(r)
    for a in range(kh):
        for b in range(kw):
            dy, dx = a - oy, b - ox
            i0, i1 = max(0, -dy), min(H, H - dy)
            j0, j1 = max(0, -dx), min(W, W - dx)
            if i0 >= i1 or j0 >= j1:
                continue
            out[..., i0:i1, j0:j1] += h[a, b] * r[..., i0 + dy:i1 + dy, j0 + dx:j1 + dx]
    return out


def grad2d(x):
    """Forward differences, zero in the last row / column.

    ``pylops.Gradient(dims=(ny,nx), sampling=1, edge=False, kind='forward')`` as built at
    ``prox_lmc_deconv.py:98`` [upstream semantics, parity unpinned].  Returns (d_row, d_col).
    """
    x = np.asarray(x)
    dr = np.zeros_like(x)
    dc = np.zeros_like(x)
    dr[..., :-1, :] = x[..., 1:, :] - x[..., :-1, :]
    dc[..., :, :-1] = x[..., :, 1:] - x[..., :, :-1]
    return dr, dc


def div2d(r, s):
    """Discrete divergence = minus the adjoint of :func:`grad2d`:
    ``div(r,s)[i,j] = r[i,j]-r[i-1,j] + s[i,j]-s[i,j-1]`` with r[-1]=s[:,-1]=0 and the
    last row of r / last column of s taken as zero (they are never non-zero because
    :func:`grad2d` is zero there).  ``A.rmatvec(y) == -div2d(y_row, y_col)`` (``algs.py:437,443``)."""
    r = np.asarray(r)
    s = np.asarray(s)
    out = np.zeros_like(r)
    out[..., :-1, :] += r[..., :-1, :]
    out[..., 1:, :] -= r[..., :-1, :]
    out[..., :, :-1] += s[..., :, :-1]
    out[..., :, 1:] -= s[..., :, :-1]
    return out


class _LinOp:
    explicit = False

    def __init__(self, shape, dtype=np.float64):
        self.shape = shape
        self.dtype = np.dtype(dtype)

    @property
    def H(self):
        return _Adjoint(self)

    def __mul__(self, x):
        if isinstance(x, _LinOp):                # Op1 * Op2 (algs.py:248)
            return _Product(self, x)
        return self.matvec(np.asarray(x).ravel())

    __matmul__ = __mul__

    def __rmul__(self, a):                       # scalar * Op  (algs.py:159, 248)
        return _Scaled(self, a)

    def __add__(self, other):                    # Op1 + Op2 (algs.py:247)
        return _Sum(self, other)


class _Product(_LinOp):
    def __init__(self, a, b):
        super().__init__((a.shape[0], b.shape[1]), a.dtype)
        self.a, self.b = a, b

    def matvec(self, x):
        return self.a.matvec(self.b.matvec(x))

    def rmatvec(self, y):
        return self.b.rmatvec(self.a.rmatvec(y))


class _Sum(_LinOp):
    def __init__(self, a, b):
        super().__init__(a.shape, a.dtype)
        self.a, self.b = a, b

    def matvec(self, x):
        return self.a.matvec(x) + self.b.matvec(x)

    def rmatvec(self, y):
        return self.a.rmatvec(y) + self.b.rmatvec(y)


class _Scaled(_LinOp):
    def __init__(self, op, a):
        super().__init__(op.shape, op.dtype)
        self.op, self.a = op, a

    def matvec(self, x):
        return self.a * self.op.matvec(x)

    def rmatvec(self, y):
        return self.a * self.op.rmatvec(y)


class _Adjoint(_LinOp):
    def __init__(self, op):
        super().__init__((op.shape[1], op.shape[0]), op.dtype)
        self.op = op

    def matvec(self, x):
        return self.op.rmatvec(x)

    def rmatvec(self, y):
        return self.op.matvec(y)


class Convolve2D(_LinOp):
    """Flat-vector wrapper of :func:`blur` (``prox_lmc_deconv.py:58,64,69``)."""

    def __init__(self, dims, h, offset=None, dtype=np.float64):
        self.dims = tuple(dims)
        n = int(np.prod(self.dims))
        super().__init__((n, n), dtype)
        self.h = np.asarray(h, dtype=dtype)
        self.offset = tuple(offset) if offset is not None else (self.h.shape[0] // 2, self.h.shape[1] // 2)

    def matvec(self, x):
        return blur(np.asarray(x).reshape(self.dims), self.h.astype(x.dtype), self.offset).ravel()

    def rmatvec(self, y):
        return blur_adjoint(np.asarray(y).reshape(self.dims), self.h.astype(y.dtype), self.offset).ravel()


class Gradient(_LinOp):
    """Stacked forward differences ``[d_row x; d_col x]`` of length 2n (``prox_lmc_deconv.py:98``)."""

    def __init__(self, dims, dtype=np.float64):
        self.dims = tuple(dims)
        n = int(np.prod(self.dims))
        super().__init__((2 * n, n), dtype)

    def matvec(self, x):
        dr, dc = grad2d(np.asarray(x).reshape(self.dims))
        return np.concatenate([dr.ravel(), dc.ravel()])

    def rmatvec(self, y):
        n = self.shape[1]
        y = np.asarray(y)
        return -div2d(y[:n].reshape(self.dims), y[n:].reshape(self.dims)).ravel()


class Identity(_LinOp):
    """``pylops.Identity(n)`` (``prox_lmc_deconv.py:125``)."""

    def __init__(self, n, dtype=np.float64):
        super().__init__((n, n), dtype)

    def matvec(self, x):
        return np.asarray(x).copy()

    rmatvec = matvec


class Diagonal(_LinOp):
    """Diagonal (inpainting-mask) forward operator; BASELINE config 5's data term."""

    def __init__(self, d, dtype=np.float64):
        self.d = np.asarray(d, dtype=dtype).ravel()
        super().__init__((self.d.size, self.d.size), dtype)

    def matvec(self, x):
        return self.d.astype(x.dtype) * np.asarray(x)

    rmatvec = matvec


# ----------------------------------------------------------------------------------
# Prox operators (pyproximal ProxOperator protocol consumed by algs.py:
# __call__, prox, proxdual, grad  --  algs.py:436,440,448,461,569,578)
# ----------------------------------------------------------------------------------


class _Prox:
    def __init__(self, Op=None, hasgrad=False):
        self.Op = Op
        self.hasgrad = hasgrad

    def proxdual(self, x, tau):
        """Moreau identity, as pyproximal's default ``proxdual`` and ``prox.py:9-10``:
        ``prox_{tau f*}(x) = x - tau * prox_{f/tau}(x/tau)``."""
        return x - tau * self.prox(x / tau, 1.0 / tau)


def cg_solve(apply_A, b, x0, niter, tol=0.0):
    """Plain conjugate gradients on an SPD operator, fixed ``niter`` iterations.

    Build-specified inner solver for the implicit data step (row a8 of SURVEY section 8):
    the reference calls ``scipy.sparse.linalg.lsqr(I + tau*sigma*H^T H, y, iter_lim=niter,
    x0=warm)`` (``algs.py:247-251``; pyproximal.L2.prox upstream).  Truncated LSQR and
    truncated CG iterates differ; both converge to the same solution -- parity unpinned.
    """
    x = x0.copy()
    r = b - apply_A(x)
    p = r.copy()
    rs = float(np.vdot(r, r).real)
    for _ in range(niter):
        if rs <= tol:
            break
        Ap = apply_A(p)
        alpha = rs / float(np.vdot(p, Ap).real)
        x = x + alpha * p
        r = r - alpha * Ap
        rs_new = float(np.vdot(r, r).real)
        p = r + (rs_new / rs) * p
        rs = rs_new
    return x


class L2(_Prox):
    """``f(x) = sigma/2 ||Op x - b||^2`` -- ``pyproximal.L2(Op=H, b=y, sigma=1/sigma^2,
    niter=50, warm=True)`` as constructed at ``prox_lmc_deconv.py:101-103``; the same
    formulae are in-repo at ``algs.py:182-187`` (value), ``:283-288`` (grad),
    ``:224-266`` (prox)."""

    def __init__(self, Op=None, b=None, sigma=1.0, niter=10, warm=True):
        super().__init__(Op, True)
        self.b = None if b is None else np.asarray(b)
        self.sigma = sigma
        self.niter = niter
        self.warm = warm
        self.x0 = None

    def __call__(self, x):
        if self.Op is not None and self.b is not None:
            r = self.Op.matvec(x) - self.b
        elif self.b is not None:
            r = x - self.b
        else:
            r = x
        return (self.sigma / 2.0) * float(np.linalg.norm(r) ** 2)

    def grad(self, x):
        if self.Op is not None and self.b is not None:
            return self.sigma * self.Op.rmatvec(self.Op.matvec(x) - self.b)
        if self.b is not None:
            return self.sigma * (x - self.b)
        return self.sigma * x

    def prox(self, x, tau):
        if self.Op is not None and self.b is not None:
            ts = float(tau * self.sigma)
            rhs = x + ts * self.Op.rmatvec(self.b)

            def apply_A(v):
                return v + ts * self.Op.rmatvec(self.Op.matvec(v))
            x0 = self.x0 if (self.warm and self.x0 is not None) else np.zeros_like(x)
            sol = cg_solve(apply_A, rhs, x0, self.niter)
            if self.warm:
                self.x0 = sol
            return sol
        if self.b is not None:
            return (x + tau * self.sigma * self.b) / (1.0 + tau * self.sigma)
        return x / (1.0 + tau * self.sigma)


class L1(_Prox):
    """``sigma*||x||_1`` -- ``pyproximal.L1(sigma=tau)`` (``prox_lmc_deconv.py:119``);
    soft threshold == ``prox.py:18-19``; dual prox = clip to [-sigma, sigma]."""

    def __init__(self, sigma=1.0):
        super().__init__(None, False)
        self.sigma = sigma

    def __call__(self, x):
        return self.sigma * float(np.sum(np.abs(x)))

    def prox(self, x, tau):
        t = self.sigma * tau
        return np.sign(x) * np.maximum(np.abs(x) - t, 0)

    def proxdual(self, x, tau):
        return np.clip(x, -self.sigma, self.sigma)


class L21(_Prox):
    """``sigma * sum_pixels ||(x_row, x_col)||_2`` -- ``pyproximal.L21(ndim=2, sigma=tau)``
    (``prox_lmc_deconv.py:116``).  Input is the stacked vector of length 2n."""

    def __init__(self, ndim=2, sigma=1.0):
        super().__init__(None, False)
        self.ndim = ndim
        self.sigma = sigma

    def __call__(self, x):
        x = np.asarray(x).reshape(self.ndim, -1)
        return self.sigma * float(np.sum(np.sqrt(np.sum(x * x, axis=0))))

    def prox(self, x, tau):
        shp = np.shape(x)
        x = np.asarray(x).reshape(self.ndim, -1)
        nrm = np.sqrt(np.sum(x * x, axis=0))
        t = self.sigma * tau
        scale = np.maximum(1.0 - t / np.maximum(nrm, 1e-300), 0.0)
        return (x * scale).reshape(shp)

    def proxdual(self, x, tau):
        """Per-pixel projection onto the l2 ball of radius sigma."""
        shp = np.shape(x)
        x = np.asarray(x).reshape(self.ndim, -1)
        nrm = np.sqrt(np.sum(x * x, axis=0))
        return (x / np.maximum(1.0, nrm / self.sigma)).reshape(shp)


def tv_value(x):
    """Isotropic TV, forward differences: ``sum sqrt(d_row^2 + d_col^2)`` per image."""
    dr, dc = grad2d(x)
    return np.sum(np.sqrt(dr * dr + dc * dc), axis=(-2, -1))


def fgp_betas(niter, momentum="unlocbox", dtype=np.float64):
    """Momentum coefficients beta_k = (t_{k-1}-1)/t_k, k = 1..niter, t_0 = 1.

    'unlocbox': t_k = (1 + sqrt(4 t_{k-1}^2))/2 (the update pyproximal.TV inherits from
    UNLocBoX's prox_tv [upstream, unverified here]); 'fista': t_k = (1+sqrt(1+4 t^2))/2;
    'none': plain projected gradient.  The device kernel takes this table from the host,
    so the choice costs nothing there."""
    t = 1.0
    out = []
    for _ in range(niter):
        if momentum == "unlocbox":
            tn = (1.0 + math.sqrt(4.0 * t * t)) / 2.0
        elif momentum == "fista":
            tn = (1.0 + math.sqrt(1.0 + 4.0 * t * t)) / 2.0
        elif momentum == "none":
            tn = 1.0
        else:
            raise ValueError(momentum)
        out.append((t - 1.0) / tn)
        t = tn
    return np.asarray(out, dtype=dtype)


def tv_prox_fgp(x, gamma, niter, step=0.125, betas=None, rtol=0.0, momentum="unlocbox", dual0=None, return_dual=False):
    """``prox_{gamma*TV}(x)`` by ``niter`` fast-gradient-projection dual iterations.

    Build-specified restatement of ``pyproximal.TV(dims, sigma, niter, rtol).prox``
    (constructed at ``prox_lmc_deconv.py:122`` and ``algs.py:169-170``) [upstream:
    Beck-Teboulle FGP as in UNLocBoX prox_tv; parity unpinned]:

        (rr,ss) = (p,q) = 0
        repeat niter times:
            sol = x - gamma*div(rr,ss)
            (r,s) = (rr,ss) - step/gamma * grad(sol);  w = max(1, |(r,s)|_2)
            (p',q') = (r,s)/w;  (rr,ss) = (p',q') + beta_k ((p',q') - (p,q));  (p,q) = (p',q')
        return x - gamma*div(rr,ss)

    ``rtol > 0`` adds the reference's per-image early exit on the relative change of the
    primal objective (upstream's default 1e-4 is what ``prox_lmc_deconv.py:122`` and
    ``algs.py:169`` leave in force); the device follows it chain by chain when asked to
    (``TV(rtol=...)`` / ``L2_ncvx_tv(rtol=...)``) and runs the fixed ``niter``
    at ``rtol = 0``.

    NAMED RISK (cannot be settled in this image: pyproximal is neither vendored nor installable).  Upstream forms ``sol`` at the
    TOP of each loop pass, tests the exit, updates the dual, and returns the ``sol`` of the pass it leaves in.  Whether ``niter``
    passes leave a ``sol`` that reflects ``niter`` dual updates (loop ``while iter <= niter``, as recalled for pyproximal) or
    ``niter - 1`` (loop ``while iter < niter``, as in UNLocBoX's prox_tv with ``iter`` starting at 1) depends on the upstream
    version.  This restatement returns the iterate after ``niter`` updates; the other reading is ``niter - 1`` here, and the
    device runs either (``lmc_problem.tv_lagged_output``).  Likewise upstream's default ``rtol = 1e-4`` (which the reference's
    call ``TV(dims, sigma, niter=niter_tv)`` does not override) stops a typical MYULA iterate's prox after about 3 passes
    (tests/test_oracle_operators.py::test_tv_rtol_exit_statistics); tests/golden/algs.npz uses ``rtol = 0``, algs_rtol.npz both.

    ``dual0 = (p, q)`` starts from that projected dual instead of zero with the momentum restarted (the build's warm-dual
    variant; NOT the reference's algorithm); ``return_dual`` also returns the final ``(p, q)``.
    """
    x = np.asarray(x)
    dt = x.dtype
    gamma = dt.type(gamma)
    c = dt.type(step) / gamma
    if betas is None:
        betas = fgp_betas(niter, momentum)
    betas = np.asarray(betas, dtype=dt)
    if dual0 is None:
        rr = np.zeros_like(x)
        ss = np.zeros_like(x)
        p = np.zeros_like(x)
        q = np.zeros_like(x)
    else:
        p = np.array(dual0[0], dtype=dt)
        q = np.array(dual0[1], dtype=dt)
        rr, ss = p.copy(), q.copy()
    prev_obj = None
    one = dt.type(1)
    for k in range(niter):
        sol = x - gamma * div2d(rr, ss)
        if rtol > 0.0:
            if x.ndim != 2:
                raise ValueError("rtol early exit is per image; call with a single image")
            obj = 0.5 * float(np.sum((x - sol) ** 2)) + float(gamma) * float(tv_value(sol))
            rel = abs(obj - prev_obj) / obj if (prev_obj is not None and obj > 0) else 2 * rtol
            prev_obj = obj
            if rel < rtol:
                return sol
        dr, dc = grad2d(sol)
        r = rr - c * dr
        s = ss - c * dc
        w = np.maximum(one, np.sqrt(r * r + s * s))
        pn = r / w
        qn = s / w
        rr = pn + betas[k] * (pn - p)
        ss = qn + betas[k] * (qn - q)
        p, q = pn, qn
    out = x - gamma * div2d(rr, ss)
    return (out, (p, q)) if return_dual else out


def tv1d_value(x):
    """1-D total variation of the vector(s) on the last axis: ``sum |x[i+1] - x[i]|``."""
    return np.sum(np.abs(np.diff(x, axis=-1)), axis=-1)


def tv1d_prox_fgp(x, gamma, niter, step=0.25, betas=None, rtol=0.0, momentum="unlocbox"):
    """``prox_{gamma*TV_1D}(x)`` of a VECTOR by ``niter`` fast-gradient-projection dual iterations -- what ``pyproximal.TV`` does for
    ``len(dims) == 1``, which is how ``algs.L2_ncvx_tv`` builds the inner prox of its ANISOTROPIC ME-TV branch: a 1-D TV over the flattened
    image, ``TV((np.prod(dims),), 1., niter, rtol)`` (``algs.py:170``) [upstream: UNLocBoX prox_tv1d; parity unpinned].  The same loop as
    :func:`tv_prox_fgp` with one dual component, forward differences ``d[i] = sol[i+1] - sol[i]`` (0 at the end), dual step ``step / gamma`` with
    ``step = 1/4`` (1 / (2 ndim) where the 2-D prox has 1/8), projection onto ``|r| <= 1``; same momentum table, same early exit."""
    x = np.asarray(x)
    if x.ndim != 1:
        raise ValueError("one vector at a time (the early exit is per image)")
    dt = x.dtype
    gamma = dt.type(gamma)
    c = dt.type(step) / gamma
    if betas is None:
        betas = fgp_betas(niter, momentum)
    betas = np.asarray(betas, dtype=dt)
    rr = np.zeros_like(x)
    p = np.zeros_like(x)
    one = dt.type(1)

    def div1(r):            # r[i] - r[i-1], the last entry of r taken as zero (no difference across the end)
        d = np.zeros_like(r)
        d[:-1] += r[:-1]
        d[1:] -= r[:-1]
        return d
    prev_obj = None
    for k in range(niter):
        sol = x - gamma * div1(rr)
        if rtol > 0.0:
            obj = 0.5 * float(np.sum((x - sol) ** 2)) + float(gamma) * float(tv1d_value(sol))
            rel = abs(obj - prev_obj) / obj if (prev_obj is not None and obj > 0) else 2 * rtol
            prev_obj = obj
            if rel < rtol:
                return sol
        d = np.zeros_like(sol)
        d[:-1] = sol[1:] - sol[:-1]
        r = rr - c * d
        pn = r / np.maximum(one, np.abs(r))
        rr = pn + betas[k] * (pn - p)
        p = pn
    return x - gamma * div1(rr)


class TV(_Prox):
    """``sigma * TV_iso(x)`` -- ``pyproximal.TV(dims=img.shape, sigma=tau, niter=niter_tv)``
    (``prox_lmc_deconv.py:122``).  One-element ``dims``: the 1-D TV of a vector (``algs.py:170``)."""

    def __init__(self, dims, sigma=1.0, niter=10, rtol=0.0, step=0.125, momentum="unlocbox"):
        super().__init__(None, False)
        self.dims = tuple(dims)
        self.sigma = sigma
        self.niter = niter
        self.rtol = rtol
        self.step = step
        self.momentum = momentum

    def __call__(self, x):
        if len(self.dims) == 1:
            return self.sigma * float(tv1d_value(np.asarray(x).ravel()))
        return self.sigma * float(tv_value(np.asarray(x).reshape(self.dims)))

    def prox(self, x, tau):
        if len(self.dims) == 1:          # step 1 / (2 ndim) / 2 as upstream: 1/4 in 1-D where the 2-D default is 1/8
            return tv1d_prox_fgp(np.asarray(x).ravel(), self.sigma * tau, self.niter, step=2 * self.step, rtol=self.rtol, momentum=self.momentum)
        out = tv_prox_fgp(np.asarray(x).reshape(self.dims), self.sigma * tau, self.niter,
                          step=self.step, rtol=self.rtol, momentum=self.momentum)
        return out.ravel()


def haar_fwd(x, levels=3):
    """Orthonormal 2-D Haar transform, `levels` levels, in place layout (Mallat): after level l the approximation
    occupies the top-left (H/2^l, W/2^l) corner.  H and W must be multiples of 2^levels.  Build-specified prior of
    BASELINE config 5 (the reference has no wavelet code); checked against PyWavelets (tests/golden/haar_pywt.npz)."""
    x = np.array(x, copy=True)
    H, W = x.shape[-2:]
    assert H % (1 << levels) == 0 and W % (1 << levels) == 0
    h, w = H, W
    for _ in range(levels):
        a = x[..., 0:h:2, 0:w:2]; b = x[..., 0:h:2, 1:w:2]; c = x[..., 1:h:2, 0:w:2]; d = x[..., 1:h:2, 1:w:2]
        ll, lh, hl, hh = (a + b + c + d) / 2, (a - b + c - d) / 2, (a + b - c - d) / 2, (a - b - c + d) / 2
        x[..., :h // 2, :w // 2] = ll; x[..., :h // 2, w // 2:w] = lh
        x[..., h // 2:h, :w // 2] = hl; x[..., h // 2:h, w // 2:w] = hh
        h, w = h // 2, w // 2
    return x


def haar_inv(cf, levels=3):
    cf = np.array(cf, copy=True)
    H, W = cf.shape[-2:]
    h, w = H >> levels, W >> levels
    for _ in range(levels):
        ll = cf[..., :h, :w].copy(); lh = cf[..., :h, w:2 * w].copy(); hl = cf[..., h:2 * h, :w].copy(); hh = cf[..., h:2 * h, w:2 * w].copy()
        cf[..., 0:2 * h:2, 0:2 * w:2] = (ll + lh + hl + hh) / 2
        cf[..., 0:2 * h:2, 1:2 * w:2] = (ll - lh + hl - hh) / 2
        cf[..., 1:2 * h:2, 0:2 * w:2] = (ll + lh - hl - hh) / 2
        cf[..., 1:2 * h:2, 1:2 * w:2] = (ll - lh - hl + hh) / 2
        h, w = 2 * h, 2 * w
    return cf


def haar_l1_prox(x, thr, levels=3):
    """prox of thr * ||detail coefficients of Haar(x)||_1: soft-threshold every detail coefficient, keep the coarsest
    approximation (orthonormal transform => exact prox)."""
    cf = haar_fwd(x, levels)
    H, W = cf.shape[-2:]
    keep = cf[..., :H >> levels, :W >> levels].copy()
    cf = np.sign(cf) * np.maximum(np.abs(cf) - thr, 0)
    cf[..., :H >> levels, :W >> levels] = keep
    return haar_inv(cf, levels)


def haar_l1_value(x, levels=3):
    cf = haar_fwd(x, levels)
    H, W = cf.shape[-2:]
    tot = np.sum(np.abs(cf), axis=(-2, -1))
    return tot - np.sum(np.abs(cf[..., :H >> levels, :W >> levels]), axis=(-2, -1))


class WaveletL1(_Prox):
    """``sigma * ||W_detail x||_1`` with W the 3-level orthonormal Haar transform (BASELINE config 5's prior)."""

    def __init__(self, dims, sigma=1.0, levels=3):
        super().__init__(None, False)
        self.dims, self.sigma, self.levels = tuple(dims), sigma, levels

    def __call__(self, x):
        return self.sigma * float(haar_l1_value(np.asarray(x).reshape(self.dims), self.levels))

    def prox(self, x, tau):
        return haar_l1_prox(np.asarray(x).reshape(self.dims), self.sigma * tau, self.levels).ravel()


class L2NcvxTV(_Prox):
    """Restatement of the in-repo class ``L2_ncvx_tv`` (``algs.py:22-291``):
    ``f(x) = sigma/2||Op x - b||^2 - lamda * env_gamma(g)(Op2 x)``.

    Restated: the branches exercised by ``prox_lmc_deconv.py:106-113`` -- MC-TV isotropic (``Op2 = Gradient``,
    ``isotropic=True``; value ``algs.py:173-190``, grad ``:273-277``) and ME-TV (``Op2 = None``; grad ``:282``) -- and the
    anisotropic MC-TV branches (``isotropic=False``: value without the pixel-norm reduction, prox ``:218-219``, grad
    ``:278-279``), with ``q = None``.  (Anisotropic ME-TV is a 1-D TV over the flattened image, ``:170``: not restated.)
    """

    def __init__(self, dims, Op=None, Op2=None, b=None, sigma=1.0, lamda=1.0, gamma=0.5,
                 isotropic=True, niter=10, tv_kwargs=None):
        super().__init__(Op, True)
        self.dims = tuple(dims)
        self.ndim = len(self.dims)
        self.Op2 = Op2
        self.b = None if b is None else np.asarray(b)
        self.sigma = sigma
        self.lamda = lamda
        self.gamma = gamma
        self.isotropic = isotropic
        self.niter = niter
        if Op2 is not None:
            self.g_gamma = L1(1.0)                       # algs.py:166
        elif isotropic:
            self.g_gamma = TV(self.dims, 1.0, niter, **(tv_kwargs or {}))   # algs.py:169
        else:                                            # algs.py:170: a 1-D TV over the flattened image
            self.g_gamma = TV((int(np.prod(self.dims)),), 1.0, niter, **(tv_kwargs or {}))

    def __call__(self, x):                               # algs.py:173-190
        Op2x = self.Op2.matvec(x) if self.Op2 is not None else x
        if self.Op2 is not None and self.isotropic:
            Op2x = Op2x.reshape(self.ndim, len(Op2x) // self.ndim)
            Op2x = np.sqrt(np.sum(Op2x ** 2, axis=0))
        mp = self.g_gamma.prox(Op2x, self.gamma)
        env = self.g_gamma(mp) + np.linalg.norm(Op2x - mp) ** 2 / (2 * self.gamma)
        f = (self.sigma / 2.0) * (np.linalg.norm(self.Op.matvec(x) - self.b) ** 2)
        return f - self.lamda * env

    def grad_moreau(self, x):                            # algs.py:271-282
        if self.Op2 is not None:
            Op2x = self.Op2.matvec(x)
            if not self.isotropic:                       # algs.py:278-279 (and the prox pre-step :218-219): component-wise
                return self.Op2.rmatvec(Op2x - self.g_gamma.prox(Op2x, self.gamma)) / self.gamma
            e = np.linalg.norm(Op2x.reshape(self.ndim, len(Op2x) // self.ndim), axis=0)
            e = np.where(e != 0, e, 1e-9)
            return self.Op2.rmatvec(np.minimum(1 / self.gamma, np.tile(1 / e, 2)) * Op2x)
        return (x - self.g_gamma.prox(x, self.gamma)) / self.gamma

    def grad(self, x):                                   # algs.py:283-291
        g = self.sigma * self.Op.rmatvec(self.Op.matvec(x) - self.b)
        return g - self.lamda * self.grad_moreau(x)

    def prox(self, x, tau):
        """Restatement of ``L2_ncvx_tv.prox`` (algs.py:201-267), MC-TV isotropic branch (:213-217) followed by the
        linear solve (:224-256).  The reference solves ``(I + tau sigma Op^T Op) u = y`` with
        ``scipy.sparse.linalg.lsqr(Op1, y, iter_lim=niter, x0=warm)``; here the same system goes through
        :func:`cg_solve` (SPD system, condition number <= 1 + tau*sigma: both are converged to round-off after the
        reference's 50 iterations -- checked against the reference's own output in tests/test_oracle_golden.py)."""
        x = np.array(x, dtype=np.float64, copy=True)
        x = x + tau * self.lamda * self.grad_moreau(x)                       # algs.py:213-217 (MC-TV) / :221-223 (ME-TV)
        y = x + tau * self.sigma * self.Op.rmatvec(self.b)                   # algs.py:225 (OpTb = sigma Op^T b, :159)
        ts = float(tau * self.sigma)

        def apply_A(v):
            return v + ts * self.Op.rmatvec(self.Op.matvec(v))
        x0 = self._x0 if getattr(self, "_x0", None) is not None else np.zeros_like(y)
        sol = cg_solve(apply_A, y, x0, self.niter)
        self._x0 = sol                                                       # warm start (algs.py:255-256)
        return sol


# ----------------------------------------------------------------------------------
# Samplers
# ----------------------------------------------------------------------------------


def myula(proxf, proxg, x0, tau, gamma, epsg=1.0, niter=10, seed=0, callback=None, noise=None):
    """MYULA, one chain -- restatement of ``algs.py:559-570,587``.

    ``x <- (1-tau/gamma) x - tau grad f(x) + tau/gamma prox_{epsg*gamma*g}(x) + sqrt(2 tau) xi``.
    Noise: one ``rng.standard_normal(n)`` per iteration from ``default_rng(seed)`` -- bit
    identical to ``scipy.stats.multivariate_normal.rvs(size=x.shape, random_state=rng)``
    at ``algs.py:565``.  ``noise[k]`` overrides it (injected-noise parity).
    Returns all iterates ``(niter, n)``.
    """
    x = np.array(x0, copy=True)
    rng = default_rng(seed)
    out = []
    for k in range(niter):
        xi = rng.standard_normal(x.shape) if noise is None else noise[k]
        x = (1 - tau / gamma) * x - tau * proxf.grad(x) + tau / gamma * proxg.prox(x, epsg * gamma) \
            + np.sqrt(2 * tau) * xi
        out.append(x)
        if callback is not None:
            callback(x)
    return np.array(out)


def ulpda(proxf, proxg, A, x0, tau, mu, y0=None, z=None, theta=1.0, niter=10, seed=0,
          gfirst=True, callback=None, callbacky=False, returny=False, noise=None):
    """ULPDA, one chain -- restatement of ``algs.py:402-408,425-458,471-474``."""
    tau = np.full(niter, tau, dtype=x0.dtype) if np.isscalar(tau) else np.asarray(tau)
    mu = np.full(niter, mu, dtype=x0.dtype) if np.isscalar(mu) else np.asarray(mu)
    x = np.array(x0, copy=True)
    xhat = x.copy()
    y = np.array(y0, copy=True) if y0 is not None else np.zeros(A.shape[0], dtype=x.dtype)
    xs, ys = [], []
    rng = default_rng(seed)
    for k in range(niter):
        xi = rng.standard_normal(x.shape) if noise is None else noise[k]
        xold = x.copy()
        if gfirst:
            y = proxg.proxdual(y + mu[k] * A.matvec(xhat), mu[k])
            ATy = A.rmatvec(y)
            if z is not None:
                ATy = ATy + z
            x = proxf.prox(x - tau[k] * ATy, tau[k]) + np.sqrt(2 * tau[k]) * xi
            xhat = x + theta * (x - xold)
        else:
            ATy = A.rmatvec(y)
            if z is not None:
                ATy = ATy + z
            x = proxf.prox(x - tau[k] * ATy, tau[k]) + np.sqrt(2 * tau[k]) * xi
            xhat = x + theta * (x - xold)
            y = proxg.proxdual(y + mu[k] * A.matvec(xhat), mu[k])
        xs.append(x)
        ys.append(y)
        if callback is not None:
            callback(x, y) if callbacky else callback(x)
    if returny:
        return np.array(xs), np.array(ys)
    return np.array(xs)


def myula_step(x, y, h, offset, sigma_f, tau, gamma, prior, xi, mask=None):
    """One batched MYULA step on image-shaped states ``x[..., H, W]`` (same formula as
    :func:`myula`, data term = blur (``h``) or diagonal ``mask``), used for device parity.

    ``prior`` is a dict: {'kind': 'l2'|'l1'|'tv'|'none', 'sigma':, 'niter':, 'step':, 'betas':}
    with prox parameter ``epsg*gamma`` already folded in as ``prior['t']``.
    """
    dt = x.dtype
    if mask is not None:
        g = dt.type(sigma_f) * (mask * (mask * x - y))
    elif h is not None:
        g = dt.type(sigma_f) * blur_adjoint(blur(x, h, offset) - y, h, offset)
    else:
        g = dt.type(sigma_f) * (x - y)
    kind = prior["kind"]
    t = dt.type(prior.get("t", 0.0))
    if kind == "l2":
        px = x / (dt.type(1) + t * dt.type(prior["sigma"]))
    elif kind == "l1":
        thr = t * dt.type(prior["sigma"])
        px = np.sign(x) * np.maximum(np.abs(x) - thr, 0)
    elif kind == "tv":
        if prior.get("warm"):     # warm-dual variant: prior["dual"] carries (p, q) between calls (updated in place in the dict)
            px, prior["dual"] = tv_prox_fgp(x, float(t) * prior["sigma"], prior["niter"], step=prior.get("step", 0.125),
                                            betas=prior.get("betas"), momentum=prior.get("momentum", "unlocbox"),
                                            dual0=prior.get("dual"), return_dual=True)
        elif prior.get("rtol", 0.0) > 0.0:     # upstream's early exit is per image: one call per chain
            flat = x.reshape((-1,) + x.shape[-2:])
            px = np.stack([tv_prox_fgp(xc, float(t) * prior["sigma"], prior["niter"], step=prior.get("step", 0.125), betas=prior.get("betas"),
                                       momentum=prior.get("momentum", "unlocbox"), rtol=prior["rtol"]) for xc in flat]).reshape(x.shape)
        else:
            px = tv_prox_fgp(x, float(t) * prior["sigma"], prior["niter"], step=prior.get("step", 0.125),
                             betas=prior.get("betas"), momentum=prior.get("momentum", "unlocbox"))
    elif kind == "haar":
        px = haar_l1_prox(x, float(t) * prior["sigma"], prior.get("levels", 3))
    elif kind == "none":
        px = x
    else:
        raise ValueError(kind)
    a = dt.type(1 - tau / gamma)
    b = dt.type(tau / gamma)
    return a * x - dt.type(tau) * g + b * px + dt.type(np.sqrt(2 * tau)) * xi


def myula_batched(x0, y, h, offset, sigma_f, tau, gamma, prior, niter, noise_fn, mask=None,
                  moments=False, burn_in=0, thin=1):
    """``niter`` batched MYULA steps; ``noise_fn(k)`` returns the noise of iteration k with
    the shape of the state.  Optionally accumulates sum / sum-of-squares over chains and
    kept iterations in float64 (posterior mean ``prox_lmc_deconv.py:474`` generalised to
    many chains, burn-in and thinning)."""
    x = np.array(x0, copy=True)
    s1 = np.zeros(x.shape[-2:], dtype=np.float64)
    s2 = np.zeros(x.shape[-2:], dtype=np.float64)
    cnt = 0
    for k in range(niter):
        x = myula_step(x, y, h, offset, sigma_f, tau, gamma, prior, noise_fn(k), mask=mask)
        if moments and k >= burn_in and (k - burn_in) % thin == 0:
            xr = x.reshape(-1, *x.shape[-2:]).astype(np.float64)
            s1 += xr.sum(axis=0)
            s2 += (xr * xr).sum(axis=0)
            cnt += xr.shape[0]
    if moments:
        return x, s1, s2, cnt
    return x


# ----------------------------------------------------------------------------------
# Convergence diagnostics across chains (ABSENT in the reference, whose diagnostics are the per-iterate
# scalars of prox_lmc_deconv.py:128-133) -- build-specified, restated from the published definitions:
# split R-hat of Gelman et al. (BDA3, sec. 11.4) and the multi-chain effective sample size with Geyer's initial
# monotone sequence truncation as described in the Stan reference manual ("Effective sample size").  Plain loops on purpose:
# the device-side version (lmc_atomi_amd/diagnostics.py) is vectorised, this is its independent check.
# ----------------------------------------------------------------------------------

def chain_probes(x, ph, pw):
    """Block means of every image over a ph x pw grid: rows [a*H//ph, (a+1)*H//ph) x columns [b*W//pw, (b+1)*W//pw)."""
    x = np.asarray(x, dtype=np.float64)
    H, W = x.shape[-2:]
    xf = x.reshape(-1, H, W)
    out = np.empty((xf.shape[0], ph, pw))
    for a in range(ph):
        r0, r1 = a * H // ph, (a + 1) * H // ph
        for b in range(pw):
            c0, c1 = b * W // pw, (b + 1) * W // pw
            out[:, a, b] = xf[:, r0:r1, c0:c1].mean(axis=(1, 2))
    return out.reshape(xf.shape[0], ph * pw)


def split_rhat(tr):
    """Split R-hat of one scalar quantity; ``tr[T, M]`` = T kept iterations of M chains."""
    tr = np.asarray(tr, dtype=np.float64)
    T, M = tr.shape
    n = T // 2
    if n < 2:
        return float("nan")
    halves = [tr[:n, j] for j in range(M)] + [tr[T - n:, j] for j in range(M)]
    means = np.array([h.mean() for h in halves])
    variances = np.array([h.var(ddof=1) for h in halves])
    m = len(halves)
    B = n * means.var(ddof=1)
    Wn = variances.mean()
    var_plus = (n - 1) / n * Wn + B / n
    return float(np.sqrt(var_plus / Wn))


def ess_geyer(tr, max_lag=None):
    """Effective sample size of one scalar quantity over all chains; ``tr[T, M]``."""
    tr = np.asarray(tr, dtype=np.float64)
    T, M = tr.shape
    if T < 4:
        return float("nan")
    L = T - 1 if max_lag is None else min(T - 1, int(max_lag))
    mean_m = tr.mean(axis=0)
    acov = np.empty((L + 1, M))
    for m in range(M):
        d = tr[:, m] - mean_m[m]
        for t in range(L + 1):
            acov[t, m] = np.dot(d[:T - t], d[t:]) / T
    chain_var = acov[0] * T / (T - 1)
    Wn = chain_var.mean()
    var_plus = Wn * (T - 1) / T
    if M > 1:
        var_plus += mean_m.var(ddof=1)
    rho = np.empty(L + 1)
    for t in range(L + 1):
        rho[t] = 1.0 - (Wn - acov[t].mean()) / var_plus
    rho[0] = 1.0
    tau = -1.0
    prev = None
    t = 0
    while t + 1 <= L:
        P = rho[t] + rho[t + 1]
        if P <= 0:
            break
        if prev is not None and P > prev:
            P = prev
        tau += 2.0 * P
        prev = P
        t += 2
    N = T * M
    tau = max(tau, 1.0 / math.log10(N))
    return float(N / tau)


# ----------------------------------------------------------------------------------
# Counter-based RNG (device noise): Philox4x32-10 + Box-Muller, float32
# ----------------------------------------------------------------------------------

_PHILOX_M0 = np.uint64(0xD2511F53)
_PHILOX_M1 = np.uint64(0xCD9E8D57)
_PHILOX_W0 = np.uint32(0x9E3779B9)
_PHILOX_W1 = np.uint32(0xBB67AE85)
LMC_PHILOX_STREAM = 0x4C4D4301      # counter word 3: "LMC" + noise stream 1


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Philox4x32 with 10 rounds (Salmon et al. 2011), vectorised over uint32 arrays."""
    c0 = np.asarray(c0, dtype=np.uint32)
    c1 = np.broadcast_to(np.asarray(c1, dtype=np.uint32), c0.shape)
    c2 = np.broadcast_to(np.asarray(c2, dtype=np.uint32), c0.shape)
    c3 = np.broadcast_to(np.asarray(c3, dtype=np.uint32), c0.shape)
    k0 = np.uint32(k0)
    k1 = np.uint32(k1)
    mask = np.uint64(0xFFFFFFFF)
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = _PHILOX_M0 * c0.astype(np.uint64)
            p1 = _PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & mask).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & mask).astype(np.uint32)
            c0, c1, c2, c3 = hi1 ^ c1 ^ k0, lo1, hi0 ^ c3 ^ k1, lo0
            k0 = np.uint32((int(k0) + int(_PHILOX_W0)) & 0xFFFFFFFF)
            k1 = np.uint32((int(k1) + int(_PHILOX_W1)) & 0xFFFFFFFF)
    return c0, c1, c2, c3


def _u01(u):
    """uint32 -> float32 in (0, 1]: ``u * 2^-32 + 2^-33`` evaluated as one float32 fma
    (uint32->float32 conversion rounds to nearest even, as ``v_cvt_f32_u32`` does)."""
    uf = np.asarray(u, dtype=np.uint32).astype(np.float32)
    # fma(uf, 2^-32, 2^-33): the product is exact (power of two), so one rounding = fma.
    return (uf.astype(np.float64) * 2.0 ** -32 + 2.0 ** -33).astype(np.float32)


def box_muller(ua, ub):
    """Two N(0,1) float32 per (ua, ub) uint32 pair: r = sqrt(-2 ln u_a), n = r*(sin, cos)(2 pi u_b)."""
    u1 = _u01(ua).astype(np.float64)
    u2 = _u01(ub).astype(np.float64)
    r = np.sqrt(-2.0 * np.log(u1))
    th = 2.0 * np.pi * u2
    return (r * np.sin(th)).astype(np.float32), (r * np.cos(th)).astype(np.float32)


def philox_normals(seed, iteration, chain_ids, H, W):
    """The device noise field ``xi[c, i, j]`` of one iteration, float32.

    Counter layout (one Philox call serves the 4 vertically adjacent pixels of a "quad"):
    ``ctr = (q, iteration, chain_id, LMC_PHILOX_STREAM)``, ``q = (i >> 2) * W + j``,
    ``key = (seed & 0xffffffff, seed >> 32)``; outputs (o0,o1,o2,o3) ->
    Box-Muller(o0,o1) = normals of rows 4*(i>>2)+0, +1 ; Box-Muller(o2,o3) = rows +2, +3.
    Keyed by the GLOBAL chain id, so any sharding of chains over GPUs draws the same noise.
    """
    chain_ids = np.asarray(chain_ids, dtype=np.uint32).reshape(-1)
    nq = (H + 3) // 4
    q = (np.arange(nq, dtype=np.uint32)[:, None] * np.uint32(W) + np.arange(W, dtype=np.uint32)[None, :])
    c0 = np.broadcast_to(q[None], (chain_ids.size, nq, W))
    c2 = np.broadcast_to(chain_ids[:, None, None], c0.shape)
    o0, o1, o2, o3 = philox4x32_10(c0, np.uint32(iteration), c2, np.uint32(LMC_PHILOX_STREAM),
                                   seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    n0, n1 = box_muller(o0, o1)
    n2, n3 = box_muller(o2, o3)
    out = np.stack([n0, n1, n2, n3], axis=2).reshape(chain_ids.size, nq * 4, W)
    return np.ascontiguousarray(out[:, :H, :])


# ----------------------------------------------------------------------------------
# Low-dimensional plumbing samplers (BASELINE config 1)
# ----------------------------------------------------------------------------------


def _mixture_grad_potential(theta, mus, Sigmas, omegas):
    """``grad U`` of a Gaussian mixture -- ``lmc.py:39-61`` / ``prox_lmc.py:42-78``."""
    d = mus[0].shape[0]
    den = 0.0
    gden = 0.0
    for mu, S, om in zip(mus, Sigmas, omegas):
        S = np.atleast_2d(S)
        Sinv = np.linalg.inv(S)
        N = np.sqrt((2 * np.pi) ** d * np.abs(np.linalg.det(S)))
        fac = np.einsum('...k,kl,...l->...', theta - mu, Sinv, theta - mu)
        pdf = np.exp(-fac / 2) / N
        den = den + om * pdf
        gden = gden + om * (pdf * Sinv @ (mu - theta))
    return -gden / den


def toy_ula(mus, Sigmas, omegas, gamma, K=1000, seed=0):
    """ULA on a Gaussian mixture -- ``lmc.py:94-104``: ``theta0 = rng.standard_normal(d)``
    then per iteration ``xi = rng.multivariate_normal(0, I)`` (== ``standard_normal(d)``,
    SURVEY A.2) and ``theta <- theta - gamma grad U(theta) + sqrt(2 gamma) xi``."""
    d = mus[0].shape[0]
    rng = default_rng(seed)
    th = rng.standard_normal(d)
    out = []
    for _ in range(K):
        xi = rng.standard_normal(d)
        th = th - gamma * _mixture_grad_potential(th, mus, Sigmas, omegas) + np.sqrt(2 * gamma) * xi
        out.append(th)
    return np.array(out)


def _soft(x, t):
    """``prox.py:18-19``."""
    return np.sign(x) * np.maximum(np.abs(x) - t, 0)


def toy_myula(mus, Sigmas, omegas, lamda, alpha, gamma, K=1000, seed=0):
    """``prox_lmc.py:114-130``: gd_update + prox_update + noise."""
    d = mus[0].shape[0]
    rng = default_rng(seed)
    th = rng.standard_normal(d)
    out = []
    for _ in range(K):
        xi = rng.standard_normal(d)
        gd = th - gamma * _mixture_grad_potential(th, mus, Sigmas, omegas)
        pu = -gamma * (th - _soft(th, lamda * alpha)) / lamda
        th = gd + pu + np.sqrt(2 * gamma) * xi
        out.append(th)
    return np.array(out)


def toy_pgld(mus, Sigmas, omegas, lamda, alpha, gamma, K=1000, seed=0):
    """``prox_lmc.py:99-110``: prox first, then gradient step + noise."""
    d = mus[0].shape[0]
    rng = default_rng(seed)
    th = rng.standard_normal(d)
    out = []
    for _ in range(K):
        xi = rng.standard_normal(d)
        th = _soft(th, lamda * alpha)
        th = th - gamma * _mixture_grad_potential(th, mus, Sigmas, omegas) + np.sqrt(2 * gamma) * xi
        out.append(th)
    return np.array(out)


def _mixture_density(theta, mus, Sigmas, omegas):
    """``prox_lmc.py:42-51``."""
    d = mus[0].shape[0]
    den = 0.0
    for mu, S, om in zip(mus, Sigmas, omegas):
        S = np.atleast_2d(S)
        fac = np.einsum('...k,kl,...l->...', theta - mu, np.linalg.inv(S), theta - mu)
        den = den + om * np.exp(-fac / 2) / np.sqrt((2 * np.pi) ** d * np.abs(np.linalg.det(S)))
    return den


def toy_mymala(mus, Sigmas, omegas, lamda, alpha, mu, gamma, K=1000, seed=0):
    """MYMALA on the mixture x Laplace target -- ``prox_lmc.py:134-158``: MYULA proposal (:150), acceptance
    ``min(1, pi(new) q(old|new) / (pi(old) q(new|old)))`` with ``pi = mixture * Laplace`` (:139-143) and
    ``q(a|b) = N(a; gd_update(b) + prox_update(b), 2 gamma I)`` (:135-136); ``rng.random()`` is drawn after the normal
    draw of every iteration (:153); only accepted states are appended (:154-155).  Returns ``(states, n_accepted)``."""
    d = mus[0].shape[0]
    rng = default_rng(seed)
    th = rng.standard_normal(d)

    def mean(t):
        return t - gamma * _mixture_grad_potential(t, mus, Sigmas, omegas) - gamma * (t - _soft(t, lamda * alpha)) / lamda

    def target(t):
        return _mixture_density(t, mus, Sigmas, omegas) * (alpha / 2) ** d * np.exp(-alpha * np.linalg.norm(t - mu, ord=1, axis=-1))

    def q(a, b):                                       # scipy multivariate_normal(mean, cov=2 gamma).pdf
        return np.exp(-np.sum((a - mean(b)) ** 2) / (4 * gamma)) / (4 * np.pi * gamma) ** (d / 2)

    out = []
    for _ in range(K):
        xi = rng.standard_normal(d)
        new = mean(th) + np.sqrt(2 * gamma) * xi
        p = (target(new) / target(th)) * (q(th, new) / q(new, th))
        if rng.random() <= min(1, p):
            out.append(new)
            th = new
    return np.array(out), len(out)


def energies(x, y, h, offset, sigma_f, prior, mask=None):
    """Per-chain ``f(x_c) = sigma_f/2 ||A x_c - y||^2`` and ``g(x_c)`` of image-shaped states ``x[C, H, W]`` (the two
    numbers of the energy log, algs.py:578-582), ``prior`` as in :func:`myula_step`."""
    if mask is not None:
        r = mask * x - y
    elif h is not None:
        r = blur(x, h, offset) - y
    else:
        r = x - y
    f = 0.5 * sigma_f * np.sum(r * r, axis=(-2, -1))
    kind = prior["kind"]
    if kind == "tv":
        g = prior["sigma"] * np.array([tv_value(xc) for xc in x])
    elif kind == "l1":
        g = prior["sigma"] * np.sum(np.abs(x), axis=(-2, -1))
    elif kind == "l2":
        g = 0.5 * prior["sigma"] * np.sum(x * x, axis=(-2, -1))
    elif kind == "haar":
        g = prior["sigma"] * np.array([haar_l1_value(xc, prior.get("levels", 3)) for xc in x])
    else:
        g = np.zeros(x.shape[0])
    return f, g


def mymala_batched(x0, y, h, offset, sigma_f, tau, gamma, prior, niter, noise_fn, uniform_fn, mask=None, epsg=1.0):
    """MYMALA for ``C`` image-shaped chains: the accept / reject of ``prox_lmc.py:134-158`` with the MYULA proposal of
    ``algs.py:569`` and target ``exp(-f - epsg*g)`` (the potential ``f + eps*g`` of algs.py:582 whose gradient the MYULA drift
    approximates; ``prior['t']`` must hold ``epsg*gamma``), in log form
    ``log alpha = U(x) - U(x') - (||x - m(x')||^2 - ||x' - m(x)||^2) / (4 tau)``; ``noise_fn(k) -> [C,H,W]``,
    ``uniform_fn(k) -> [C]``.  A rejected chain keeps its state.  Returns ``(x, accepted[C], log_alpha[niter, C])``."""
    x = np.array(x0, dtype=np.float64)
    zero = np.zeros_like(x)

    def mean(v):
        return myula_step(v, y, h, offset, sigma_f, tau, gamma, prior, zero, mask=mask)

    def U(v):
        f, g = energies(v, y, h, offset, sigma_f, prior, mask=mask)
        return f + epsg * g

    mx, Ux = mean(x), U(x)
    acc = np.zeros(x.shape[0], dtype=np.int64)
    las = []
    for k in range(niter):
        xp = mx + np.sqrt(2 * tau) * noise_fn(k)
        mxp, Uxp = mean(xp), U(xp)
        d1 = np.sum((xp - mx) ** 2, axis=(-2, -1))
        d2 = np.sum((x - mxp) ** 2, axis=(-2, -1))
        la = (Ux - Uxp) - (d2 - d1) / (4 * tau)
        ok = np.log(uniform_fn(k)) <= la
        x[ok], mx[ok], Ux[ok] = xp[ok], mxp[ok], Uxp[ok]
        acc += ok
        las.append(la)
    return x, acc, np.array(las)


LMC_PHILOX_ACCEPT = 0x4C4D4302


def philox_uniforms(seed, iteration, chain_ids):
    """The Metropolis uniforms of the device: ``u01(first word of Philox(ctr=(0, iteration, chain, LMC_PHILOX_ACCEPT)))``."""
    chain_ids = np.asarray(chain_ids, dtype=np.uint32).reshape(-1)
    o0, _, _, _ = philox4x32_10(np.zeros_like(chain_ids), np.uint32(iteration), chain_ids, np.uint32(LMC_PHILOX_ACCEPT),
                                seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    return _u01(o0).astype(np.float64)


# ----------------------------------------------------------------------------------
# Closed-form prox library (prox.py:9-65), elementwise restatements
# ----------------------------------------------------------------------------------


def prox_laplace(x, gamma):                      # prox.py:18-19
    return np.sign(x) * np.maximum(np.abs(x) - gamma, 0)


def prox_uncentered_laplace(x, gamma, mu):       # prox.py:22-23
    return mu + prox_laplace(x - mu, gamma)


def prox_gaussian(x, gamma):                     # prox.py:26-27
    return x / (2 * gamma + 1)


def prox_conjugate(x, gamma, prox):              # prox.py:9-10
    return x - gamma * prox(x / gamma, 1 / gamma)


def prox_gen_gaussian(x, gamma, p):              # prox.py:30-41 (p matched by value)
    x = np.asarray(x, dtype=np.float64)
    if p == 4 / 3:
        xi = np.sqrt(x ** 2 + 256 * gamma ** 3 / 729)
        return x + 4 * gamma / (3 * 2 ** (1 / 3)) * ((xi - x) ** (1 / 3) - (xi + x) ** (1 / 3))
    if p == 3 / 2:
        return x + 9 * gamma ** 2 * np.sign(x) * (1 - np.sqrt(1 + 16 * np.abs(x) / (9 * gamma ** 2))) / 8
    if p == 3:
        return np.sign(x) * (np.sqrt(1 + 12 * gamma * np.abs(x)) - 1) / (6 * gamma)
    if p == 4:
        xi = np.sqrt(x ** 2 + 1 / (27 * gamma))
        return ((xi + x) / (8 * gamma)) ** (1 / 3) - ((xi - x) / (8 * gamma)) ** (1 / 3)
    raise ValueError("p must be one of 4/3, 3/2, 3, 4 (prox.py:30-41 returns an unbound name otherwise)")


def prox_huber(x, gamma, tau):                   # prox.py:44-45, vectorised with np.where
    x = np.asarray(x, dtype=np.float64)
    return np.where(np.abs(x) <= gamma * (2 * tau + 1) / np.sqrt(2 * tau),
                    x / (2 * tau + 1), x - gamma * np.sqrt(2 * tau) * np.sign(x))


def prox_smoothed_laplace(x, gamma):             # prox.py:52-53
    ax = np.abs(x)
    return np.sign(x) * (gamma * ax - gamma ** 2 - 1
                         + np.sqrt(np.abs(gamma * ax - gamma ** 2 - 1) ** 2 + 4 * gamma * ax)) / (2 * gamma)


def prox_exp(x, gamma):                          # prox.py:56-57, vectorised
    x = np.asarray(x, dtype=np.float64)
    return np.where(x >= gamma, x - gamma, 0.0)


def prox_gamma(x, omega, kappa):                 # prox.py:60-61
    return (x - omega + np.sqrt((x - omega) ** 2 + 4 * kappa)) / 2


def prox_chi(x, kappa):                          # prox.py:64-65
    return (x + np.sqrt(x ** 2 + 8 * kappa)) / 4


def prox_uniform(x, omega):                      # prox.py:68-75, vectorised
    return np.clip(x, -omega, omega)


def prox_triangular(x, omega1, omega2):          # prox.py:78-85, vectorised
    x = np.asarray(x, dtype=np.float64)
    lo = (x + omega1 + np.sqrt((x - omega1) ** 2 + 4)) / 2
    hi = (x + omega2 + np.sqrt((x - omega2) ** 2 + 4)) / 2
    return np.where(x < 1 / omega1, lo, np.where(x > 1 / omega2, hi, 0.0))
