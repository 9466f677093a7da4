"""ctypes loader of oracle/lmc_oracle_c.c (plain C float64 restatement of the MYULA step, OpenMP over chains).

TEST INFRASTRUCTURE ONLY -- imported by tests/ and by bench.py's cpu_baseline leg, never by lmc_atomi_amd/.
The numpy restatement (lmc_oracle.py) is the one pinned by the reference's own outputs (tests/golden/);
tests/test_oracle_c.py checks that this C twin agrees with it bit for bit on the same noise.
Build: ``make -C oracle`` (done by ``__graft_entry__.build()``)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "liblmc_oracle_c.so")

PRIOR = {"none": 0, "l2": 1, "l1": 2, "tv": 3}
DATA_NONE, DATA_IDENTITY, DATA_MASK, DATA_BLUR = 0, 1, 2, 3

_dp = C.POINTER(C.c_double)


class _Cfg(C.Structure):
    _fields_ = [("H", C.c_int), ("W", C.c_int), ("data_kind", C.c_int), ("y", _dp), ("mask", _dp), ("h", _dp),
                ("kh", C.c_int), ("kw", C.c_int), ("oy", C.c_int), ("ox", C.c_int),
                ("sigma_f", C.c_double), ("tau", C.c_double), ("gamma", C.c_double),
                ("prior_kind", C.c_int), ("prior_sigma", C.c_double), ("t", C.c_double),
                ("tv_niter", C.c_int), ("tv_step", C.c_double), ("betas", _dp),
                ("tv_rtol", C.c_double), ("tv_passes", C.POINTER(C.c_int))]


_lib = None


def build():
    subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.lmc_oc_max_threads.restype = C.c_int
        _lib.lmc_oc_myula_step.argtypes = [C.POINTER(_Cfg), _dp, _dp, _dp, C.c_int, C.c_int]
        _lib.lmc_oc_tv_prox.argtypes = [_dp, _dp, C.c_int, C.c_int, C.c_int, C.c_double, C.c_int, C.c_double, _dp, C.c_double,
                                        C.POINTER(C.c_int), C.c_int]
        _lib.lmc_oc_blur.argtypes = [_dp, _dp, C.c_int] + [C.c_int] * 2 + [_dp] + [C.c_int] * 5
    return _lib


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _p(a):
    return a.ctypes.data_as(_dp) if a is not None else None


def max_threads():
    return int(lib().lmc_oc_max_threads())


def myula_step(x, y, h, offset, sigma_f, tau, gamma, prior, xi, mask=None, threads=1, passes=None):
    """Same arguments and meaning as ``lmc_oracle.myula_step`` (priors none / l2 / l1 / tv), float64.  ``prior["rtol"] > 0``: the TV prox
    with upstream's early exit (``lmc_oracle.tv_prox_fgp(rtol=...)``, image by image); ``passes``: optional int32 array [n_img] that
    receives the pass each image's prox left in."""
    from . import lmc_oracle as O
    x = _f64(x)
    xi = _f64(xi)
    Hh, W = x.shape[-2:]
    n_img = x.size // (Hh * W)
    keep = []
    cfg = _Cfg()
    cfg.H, cfg.W = Hh, W
    y = _f64(y)
    keep.append(y)
    cfg.y = _p(y)
    if mask is not None:
        m = _f64(mask)
        keep.append(m)
        cfg.data_kind, cfg.mask = DATA_MASK, _p(m)
    elif h is not None:
        hh = _f64(h)
        keep.append(hh)
        cfg.data_kind, cfg.h = DATA_BLUR, _p(hh)
        cfg.kh, cfg.kw = hh.shape
        cfg.oy, cfg.ox = offset
    else:
        cfg.data_kind = DATA_IDENTITY
    cfg.sigma_f, cfg.tau, cfg.gamma = float(sigma_f), float(tau), float(gamma)
    kind = prior["kind"]
    cfg.prior_kind = PRIOR[kind]
    cfg.prior_sigma = float(prior.get("sigma", 0.0))
    cfg.t = float(prior.get("t", 0.0))
    if kind == "tv":
        K = int(prior["niter"])
        b = prior.get("betas")
        b = _f64(O.fgp_betas(K, prior.get("momentum", "unlocbox")) if b is None else b)
        keep.append(b)
        cfg.tv_niter, cfg.tv_step, cfg.betas = K, float(prior.get("step", 0.125)), _p(b)
        cfg.tv_rtol = float(prior.get("rtol", 0.0))
        if passes is not None:
            assert passes.dtype == np.int32 and passes.size == n_img and passes.flags.c_contiguous
            cfg.tv_passes = passes.ctypes.data_as(C.POINTER(C.c_int))
    out = np.empty_like(x)
    rc = lib().lmc_oc_myula_step(C.byref(cfg), _p(x), _p(xi), _p(out), n_img, int(threads))
    if rc != 0:
        raise RuntimeError(f"lmc_oc_myula_step failed ({rc})")
    return out


def tv_prox_fgp(x, gamma, niter, step=0.125, betas=None, momentum="unlocbox", threads=1, rtol=0.0, return_passes=False):
    from . import lmc_oracle as O
    x = _f64(x)
    Hh, W = x.shape[-2:]
    n_img = x.size // (Hh * W)
    b = _f64(O.fgp_betas(niter, momentum) if betas is None else betas)
    out = np.empty_like(x)
    passes = np.zeros(n_img, dtype=np.int32)
    rc = lib().lmc_oc_tv_prox(_p(x), _p(out), n_img, Hh, W, float(gamma), int(niter), float(step), _p(b), float(rtol),
                              passes.ctypes.data_as(C.POINTER(C.c_int)), int(threads))
    if rc != 0:
        raise RuntimeError(f"lmc_oc_tv_prox failed ({rc})")
    return (out, passes) if return_passes else out


def blur(x, h, offset, adjoint=False):
    x = _f64(x)
    hh = _f64(h)
    Hh, W = x.shape[-2:]
    out = np.empty_like(x)
    rc = lib().lmc_oc_blur(_p(x), _p(out), x.size // (Hh * W), Hh, W, _p(hh), hh.shape[0], hh.shape[1], offset[0], offset[1], int(adjoint))
    if rc != 0:
        raise RuntimeError(f"lmc_oc_blur failed ({rc})")
    return out
