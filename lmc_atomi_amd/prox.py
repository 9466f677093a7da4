"""Closed-form proxes of the reference's ``prox.py`` (functional plugin surface, called from
prox_lmc.py:60,106,115 and lmc_laplace.py:54), elementwise on the GPU via lmc_prox_elementwise.

Same names and argument order as the reference.  Scalar-only reference functions
(prox_huber, prox_exp, prox_uniform, prox_triangular use Python ``if``) are vectorised here.
The three ``minimize_scalar``-based proxes (prox_weibull, prox_gen_inv_gaussian, prox_pearson_I,
prox.py:88-104) and prox_square_loss (pylops solve) have no device functor and raise.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _capi, _dev


def _run(kind, x, *params):
    xt = _dev.to_dev(x)
    out = torch.empty_like(xt)
    par = np.asarray(params, dtype=np.float32)
    _dev.run(xt, "lmc_prox_elementwise", kind, _dev.ptr(xt), _dev.ptr(out), xt.numel(), _dev.fptr(par),
                                                par.size)
    return _dev.like_input(out, x)


def prox_laplace(x, gamma):                      # prox.py:18
    return _run(_capi.EPROX_LAPLACE, x, gamma)


def prox_uncentered_laplace(x, gamma, mu):       # prox.py:22
    return _run(_capi.EPROX_UNCENTERED_LAPLACE, x, gamma, mu)


def prox_gaussian(x, gamma):                     # prox.py:26
    return _run(_capi.EPROX_GAUSSIAN, x, gamma)


def prox_conjugate(x, gamma, prox):              # prox.py:9
    if prox is prox_laplace:
        return _run(_capi.EPROX_LAPLACE_CONJ, x, gamma)
    return x - gamma * prox(x / gamma, 1 / gamma)


def prox_gen_gaussian(x, gamma, p):              # prox.py:30
    kinds = {4 / 3: _capi.EPROX_GEN_GAUSSIAN_4_3, 3 / 2: _capi.EPROX_GEN_GAUSSIAN_3_2,
             3: _capi.EPROX_GEN_GAUSSIAN_3, 4: _capi.EPROX_GEN_GAUSSIAN_4}
    if p not in kinds:
        raise ValueError("p must be one of 4/3, 3/2, 3, 4")
    return _run(kinds[p], x, gamma)


def prox_huber(x, gamma, tau):                   # prox.py:44
    return _run(_capi.EPROX_HUBER, x, gamma, tau)


def prox_smoothed_laplace(x, gamma):             # prox.py:52
    return _run(_capi.EPROX_SMOOTHED_LAPLACE, x, gamma)


def prox_exp(x, gamma):                          # prox.py:56
    return _run(_capi.EPROX_EXP, x, gamma)


def prox_gamma(x, omega, kappa):                 # prox.py:60
    return _run(_capi.EPROX_GAMMA, x, omega, kappa)


def prox_chi(x, kappa):                          # prox.py:64
    return _run(_capi.EPROX_CHI, x, kappa)


def prox_uniform(x, omega):                      # prox.py:68
    return _run(_capi.EPROX_UNIFORM, x, omega)


def prox_triangular(x, omega1, omega2):          # prox.py:78
    return _run(_capi.EPROX_TRIANGULAR, x, omega1, omega2)


def _no_functor(name):
    def f(*a, **k):
        raise NotImplementedError(f"{name} has no device functor (scalar root-finding in the reference, prox.py:88-104)")
    f.__name__ = name
    return f


prox_weibull = _no_functor("prox_weibull")
prox_gen_inv_gaussian = _no_functor("prox_gen_inv_gaussian")
prox_pearson_I = _no_functor("prox_pearson_I")
prox_square_loss = _no_functor("prox_square_loss")
