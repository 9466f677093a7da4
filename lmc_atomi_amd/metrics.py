"""Per-iterate quality metrics of the reference driver, on the device and per chain:
``signal_noise_ratio`` (prox_lmc_deconv.py:35-36), ``peak_signal_noise_ratio`` / ``mean_squared_error``
(skimage.metrics as imported at prox_lmc_deconv.py:26-27), and the metrics callback of
prox_lmc_deconv.py:128-133 (cost, error norm, SNR, PSNR, MSE lists).

The squared error per image is one launch of the energy kernel (``lmc_energies`` with an identity data term whose
observation is the ground truth); nothing is copied to the host except the per-chain scalars.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _capi, _dev
from .proximal import _Problem


class _SqErr:
    def __init__(self, image_true, dims):
        self.dims = (int(dims[0]), int(dims[1]))
        self.true = np.asarray(image_true, dtype=np.float64).reshape(self.dims)
        self.norm_true = float(np.linalg.norm(self.true))
        # f = sigma_f/2 ||x - y||^2 with sigma_f = 2 and y = ground truth
        self._prob = _Problem(self.dims, data={"data_kind": _capi.DATA_IDENTITY, "sigma_f": 2.0, "y": self.true})

    def __call__(self, x):
        f, _ = self._prob.energies(x)
        return f            # float64 tensor [n_img] in HBM


def _out(t, x):
    if t.numel() == 1 and not isinstance(x, torch.Tensor):
        return float(t[0])
    return t if isinstance(x, torch.Tensor) else t.cpu().numpy()


def mean_squared_error(image_true, image_test, dims=None):
    """``skimage.metrics.mean_squared_error`` for one image or a batch ``[..., H*W]`` / ``[..., H, W]`` of test images."""
    dims = dims or np.shape(image_true)[-2:]
    e2 = _SqErr(image_true, dims)(image_test)
    return _out(e2 / (dims[0] * dims[1]), image_test)


def peak_signal_noise_ratio(image_true, image_test, data_range=255.0, dims=None):
    """``skimage.metrics.peak_signal_noise_ratio``; the reference calls it with a uint8 ground truth, for which
    skimage takes ``data_range = 255`` (prox_lmc_deconv.py:132)."""
    dims = dims or np.shape(image_true)[-2:]
    mse = _SqErr(image_true, dims)(image_test) / (dims[0] * dims[1])
    return _out(10.0 * torch.log10(data_range ** 2 / mse), image_test)


def signal_noise_ratio(image_true, image_test, dims=None):
    """``20 log10(||true|| / ||test - true||)`` (prox_lmc_deconv.py:35-36)."""
    dims = dims or np.shape(image_true)[-2:]
    se = _SqErr(image_true, dims)
    return _out(20.0 * torch.log10(se.norm_true / torch.sqrt(se(image_test))), image_test)


class MetricsCallback:
    """The callback of prox_lmc_deconv.py:128-133 for batched states: appends, per call, the per-chain cost
    ``f(x) + g(x)``, ``||x - x*||``, SNR, PSNR and MSE (numpy arrays of length n_chains).  ``sampler`` supplies the
    energies (``MYULASampler.energies`` / ``ULPDASampler.energies``)."""

    def __init__(self, image_true, dims, sampler=None, data_range=255.0):
        self.dims = (int(dims[0]), int(dims[1]))
        self._se = _SqErr(image_true, self.dims)
        self.sampler = sampler
        self.data_range = float(data_range)
        self.cost, self.err, self.snr, self.psnr, self.mse = [], [], [], [], []

    def __call__(self, x, y=None):
        e2 = self._se(x).cpu().numpy()
        n = self.dims[0] * self.dims[1]
        if self.sampler is not None:
            f, g = self.sampler.energies()
            self.cost.append((f + g).cpu().numpy())
        self.err.append(np.sqrt(e2))
        self.snr.append(20.0 * np.log10(self._se.norm_true / np.sqrt(e2)))
        self.mse.append(e2 / n)
        self.psnr.append(10.0 * np.log10(self.data_range ** 2 / (e2 / n)))
