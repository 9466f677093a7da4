"""Linear operators with the pylops protocol the reference's samplers consume
(``.matvec`` / ``.rmatvec`` / ``.shape`` / ``.H`` / ``*`` / ``@`` / ``.dtype`` / ``.explicit``:
algs.py:159-162,174,213,427,436-437), computed by HIP kernels through the C ABI.

Inputs may be numpy arrays (uploaded, result returned as numpy) or torch tensors in HBM
(result stays in HBM).  Flat vectors of length n (or batches ``[..., n]``) as in the reference.
"""
from __future__ import annotations

import numpy as np

from . import _capi, _dev


class LinearOperator:
    explicit = False
    dtype = np.dtype(np.float32)

    def __init__(self, shape):
        self.shape = tuple(int(s) for s in shape)

    @property
    def H(self):
        return _Adjoint(self)

    adjoint = lambda self: self.H  # noqa: E731  (pylops spelling used at prox.py:15)

    def __mul__(self, x):
        return self.matvec(x)

    __matmul__ = __mul__

    def __rmul__(self, a):
        return _Scaled(self, float(a))

    def _batch(self, x, n_in):
        """-> (fp32 HBM tensor [n_img, n_in], n_img, leading batch shape).  Accepts a flat vector
        ``(n_in,)``, a batch ``(..., n_in)`` or image-shaped ``(..., H, W)`` operands."""
        t = _dev.to_dev(x)
        if t.shape[-1] == n_in:
            lead = tuple(t.shape[:-1])
        elif t.ndim >= 2 and hasattr(self, "dims") and tuple(t.shape[-2:]) == self.dims and n_in == self.dims[0] * self.dims[1]:
            lead = tuple(t.shape[:-2])
        else:
            raise ValueError(f"operand of shape {tuple(t.shape)} does not match operator input size {n_in}")
        return t.reshape(-1, n_in), int(t.numel() // n_in), lead


class _Adjoint(LinearOperator):
    def __init__(self, op):
        super().__init__((op.shape[1], op.shape[0]))
        self.op = op

    def matvec(self, x):
        return self.op.rmatvec(x)

    def rmatvec(self, y):
        return self.op.matvec(y)


class _Scaled(LinearOperator):
    def __init__(self, op, a):
        super().__init__(op.shape)
        self.op, self.a = op, a

    def matvec(self, x):
        return self.a * self.op.matvec(x)

    def rmatvec(self, y):
        return self.a * self.op.rmatvec(y)


class Convolve2D(LinearOperator):
    """Zero-padded "same" convolution with origin ``offset`` -- drop-in for
    ``pylops.signalprocessing.Convolve2D((ny, nx), h=h, offset=(oy, ox))`` as constructed at
    prox_lmc_deconv.py:55-69.  ``matvec`` = H x, ``rmatvec`` = H^T x (lmc_blur)."""

    def __init__(self, dims, h, offset=None, dtype=None):
        self.dims = (int(dims[0]), int(dims[1]))
        n = self.dims[0] * self.dims[1]
        super().__init__((n, n))
        self.h = np.ascontiguousarray(np.asarray(h, dtype=np.float32))
        if self.h.ndim != 2:
            raise ValueError("h must be 2-D")
        kh, kw = self.h.shape
        if kh > _capi.MAX_BLUR or kw > _capi.MAX_BLUR:
            raise ValueError(f"kernels larger than {_capi.MAX_BLUR}x{_capi.MAX_BLUR} are not supported")
        self.offset = (kh // 2, kw // 2) if offset is None else (int(offset[0]), int(offset[1]))

    def _apply(self, x, adjoint):
        import torch
        t, n_img, _ = self._batch(x, self.shape[1])
        out = torch.empty_like(t)
        kh, kw = self.h.shape
        _dev.run(t, "lmc_blur", _dev.ptr(t), _dev.ptr(out), n_img, self.dims[0], self.dims[1],
                                        _dev.fptr(self.h), kh, kw, self.offset[0], self.offset[1],
                                        1 if adjoint else 0)
        return _dev.like_input(out.reshape(tuple(np.shape(x))), x)

    def matvec(self, x):
        return self._apply(x, False)

    def rmatvec(self, y):
        return self._apply(y, True)


class Gradient(LinearOperator):
    """Stacked forward differences ``[d_row x; d_col x]`` (zero in the last row / column) --
    drop-in for ``pylops.Gradient(dims, sampling=1, edge=False, kind='forward')``
    (prox_lmc_deconv.py:98).  ``rmatvec`` = -div."""

    def __init__(self, dims, sampling=1.0, edge=False, kind="forward", dtype=None):
        if sampling != 1 or edge or kind != "forward":
            raise NotImplementedError("only sampling=1, edge=False, kind='forward' (the reference's configuration)")
        self.dims = (int(dims[0]), int(dims[1]))
        n = self.dims[0] * self.dims[1]
        super().__init__((2 * n, n))

    def matvec(self, x):
        import torch
        t, n_img, lead = self._batch(x, self.shape[1])
        out = torch.empty((n_img, 2 * self.shape[1]), dtype=torch.float32, device=t.device)
        _dev.run(t, "lmc_gradient", _dev.ptr(t), _dev.ptr(out), n_img, self.dims[0], self.dims[1])
        return _dev.like_input(out.reshape(lead + (2 * self.shape[1],)), x)

    def rmatvec(self, y):
        import torch
        t = _dev.to_dev(y)
        if t.shape[-1] != self.shape[0]:
            raise ValueError(f"operand of shape {tuple(t.shape)} does not match the stacked field size {self.shape[0]}")
        lead = tuple(t.shape[:-1])
        t = t.reshape(-1, self.shape[0])
        n_img = t.shape[0]
        out = torch.empty((n_img, self.shape[1]), dtype=torch.float32, device=t.device)
        _dev.run(t, "lmc_gradient_adjoint", _dev.ptr(t), _dev.ptr(out), n_img, self.dims[0], self.dims[1])
        return _dev.like_input(out.reshape(lead + (self.shape[1],)), y)


class Identity(LinearOperator):
    """``pylops.Identity(n)`` (prox_lmc_deconv.py:125)."""

    def __init__(self, n, dtype=None):
        super().__init__((int(n), int(n)))

    def matvec(self, x):
        return x.clone() if hasattr(x, "clone") else np.array(x, copy=True)

    rmatvec = matvec


class Diagonal(LinearOperator):
    """Diagonal (inpainting mask) operator; the data term of BASELINE config 5."""

    def __init__(self, d, dims=None):
        self.d = np.ascontiguousarray(np.asarray(d, dtype=np.float32)).ravel()
        self.dims = None if dims is None else (int(dims[0]), int(dims[1]))
        super().__init__((self.d.size, self.d.size))
