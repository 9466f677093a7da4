"""Convergence diagnostics across chains, on the device: split R-hat and effective sample size of per-chain summaries.

SURVEY 8(f).3.  The reference runs ONE chain and has no such diagnostic (its diagnostics are the per-iterate scalars of
prox_lmc_deconv.py:128-133, see :mod:`lmc_atomi_amd.metrics`); with thousands of chains in HBM the natural check is across
chains.  Every kept iteration each chain is reduced to a few scalars -- a ``ph x pw`` grid of block means of the image
(``lmc_chain_probes``, one HIP pass over the states) and, optionally, its energies f(x), g(x) -- and the trace
``[T, C, Q]`` stays in HBM (T kept iterations, C chains, Q scalars; 66 floats per chain and iteration by default).

* :func:`split_rhat` -- Gelman et al. (BDA3 sec. 11.4): every chain split in halves, ``sqrt(var+ / W)``.
* :func:`ess` -- multi-chain effective sample size with Geyer's initial monotone sequence truncation (Stan reference manual,
  "Effective sample size"), capped at ``N log10 N``.

Both are a handful of reductions over the small trace tensor (torch ops on whatever device the trace lives on -- plumbing, not
the hot path); the tests compare them with a loop-by-loop restatement of the published definitions.
"""
from __future__ import annotations

import math

import numpy as np
import torch

from . import _capi, _dev


def chain_probes(x, grid=(8, 8), dims=None, out=None):
    """Block means ``[n_img, ph*pw]`` of the images ``x`` (``[..., H, W]`` or ``[..., H*W]`` with ``dims``) over a
    ``ph x pw`` grid: rows ``[a*H//ph, (a+1)*H//ph)`` x columns ``[b*W//pw, (b+1)*W//pw)``."""
    ph, pw = int(grid[0]), int(grid[1])
    if ph < 1 or pw < 1:
        raise ValueError(f"probe grid must be positive, got {ph}x{pw}")
    xt = _dev.to_dev(x)
    if dims is None:
        if xt.dim() < 2:
            raise ValueError("pass dims=(ny, nx) for flat images")
        dims = xt.shape[-2:]
    H, W = int(dims[0]), int(dims[1])
    if xt.numel() == 0 or xt.numel() % (H * W):
        raise ValueError(f"input of {xt.numel()} values is not a batch of {H}x{W} images")
    n_img = xt.numel() // (H * W)
    if out is None:
        out = torch.empty((n_img, ph * pw), dtype=torch.float32, device=xt.device)
    elif out.numel() != n_img * ph * pw or out.dtype != torch.float32 or not out.is_contiguous():
        raise ValueError("out must be a contiguous float32 tensor of n_img*ph*pw values")
    _dev.run(xt, "lmc_chain_probes", _dev.ptr(xt), _dev.ptr(out), n_img, H, W, ph, pw)
    return out


def _as_trace(tr):
    t = tr if isinstance(tr, torch.Tensor) else torch.as_tensor(np.asarray(tr))
    t = t.to(torch.float64)
    if t.dim() == 2:
        t = t[:, :, None]
    if t.dim() != 3:
        raise ValueError("trace must be [T, C] or [T, C, Q]")
    return t


def split_rhat(tr):
    """Split R-hat per quantity: ``tr[T, C, Q]`` (or ``[T, C]``) -> float64 tensor ``[Q]``; NaN if fewer than 4 iterations."""
    t = _as_trace(tr)
    T, M, Q = t.shape
    n = T // 2
    if n < 2:
        return torch.full((Q,), float("nan"), dtype=torch.float64, device=t.device)
    halves = torch.cat([t[:n], t[T - n:]], dim=1)                    # [n, 2M, Q]
    means = halves.mean(dim=0)
    variances = halves.var(dim=0, unbiased=True)
    B = n * means.var(dim=0, unbiased=True)
    Wn = variances.mean(dim=0)
    var_plus = (n - 1) / n * Wn + B / n
    return torch.sqrt(var_plus / Wn)


def ess(tr, max_lag=None):
    """Effective sample size per quantity over all chains: ``tr[T, C, Q]`` (or ``[T, C]``) -> float64 tensor ``[Q]``."""
    t = _as_trace(tr)
    T, M, Q = t.shape
    if T < 4:
        return torch.full((Q,), float("nan"), dtype=torch.float64, device=t.device)
    L = T - 1 if max_lag is None else min(T - 1, int(max_lag))
    mean_m = t.mean(dim=0)                                            # [M, Q]
    d = t - mean_m
    acov_mean = torch.empty((L + 1, Q), dtype=torch.float64, device=t.device)      # autocovariance averaged over chains
    for lag in range(L + 1):
        acov_mean[lag] = (d[:T - lag] * d[lag:]).sum(dim=0).mean(dim=0) / T
    Wn = acov_mean[0] * T / (T - 1)
    var_plus = Wn * (T - 1) / T
    if M > 1:
        var_plus = var_plus + mean_m.var(dim=0, unbiased=True)
    rho = 1.0 - (Wn - acov_mean) / var_plus
    rho[0] = 1.0
    npair = (L + 1) // 2
    P = rho[0:2 * npair:2] + rho[1:2 * npair:2]                       # [npair, Q]
    alive = torch.cumprod((P > 0).to(torch.float64), dim=0)           # Geyer: stop at the first non-positive pair sum
    Pm = torch.cummin(torch.where(alive > 0, P, torch.full_like(P, float("inf"))), dim=0).values   # ... and keep it monotone
    tau = -1.0 + 2.0 * torch.where(alive > 0, Pm, torch.zeros_like(P)).sum(dim=0)
    N = T * M
    tau = torch.clamp(tau, min=1.0 / math.log10(N))
    return N / tau


class ChainTrace:
    """Records the per-chain summaries of a sampler's current state and turns them into R-hat / ESS.

    ``tr = ChainTrace(sampler, grid=(8, 8))``; call ``tr.record()`` (or pass ``tr`` as a ``callback``) after every kept
    iteration; ``tr.summary()`` returns ``{'rhat': [Q], 'ess': [Q], 'rhat_max', 'ess_min', 'names', 'n_kept', 'n_chains'}``.
    """

    def __init__(self, sampler, grid=(8, 8), energies=True):
        self.sampler = sampler
        self.grid = (int(grid[0]), int(grid[1]))
        self.energies = bool(energies)
        self._rows = []
        self.names = [f"probe[{a},{b}]" for a in range(self.grid[0]) for b in range(self.grid[1])]
        if self.energies:
            self.names += ["f", "g"]

    def record(self):
        x = self.sampler.get_state()
        row = chain_probes(x, self.grid)
        if self.energies:
            f, g = self.sampler.energies()
            row = torch.cat([row, f.to(torch.float32)[:, None], g.to(torch.float32)[:, None]], dim=1)
        self._rows.append(row)

    def __call__(self, x=None, y=None):
        self.record()

    def __len__(self):
        return len(self._rows)

    def trace(self):
        """``[T, C, Q]`` float32 tensor in HBM."""
        if not self._rows:
            raise ValueError("nothing recorded")
        return torch.stack(self._rows, dim=0)

    def summary(self, trace=None, max_lag=None):
        t = self.trace() if trace is None else trace
        r, e = split_rhat(t), ess(t, max_lag)
        return {"rhat": r, "ess": e, "rhat_max": float(torch.nan_to_num(r, nan=float("inf")).max()),
                "ess_min": float(torch.nan_to_num(e, nan=0.0).min()), "names": list(self.names),
                "n_kept": int(t.shape[0]), "n_chains": int(t.shape[1])}
