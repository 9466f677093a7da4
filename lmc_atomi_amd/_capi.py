"""ctypes binding of liblmc_atomi.so (include/lmc_atomi.h).

This is the only place the package touches native code.  There is NO fallback: if the
library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "liblmc_atomi.so")
ABI_VERSION = 3

# enums (include/lmc_atomi.h)
DATA_NONE, DATA_IDENTITY, DATA_BLUR, DATA_MASK = 0, 1, 2, 3
PRIOR_NONE, PRIOR_L2, PRIOR_L1, PRIOR_TV_ISO, PRIOR_TV_ANISO, PRIOR_HAAR_L1, PRIOR_EPROX = 0, 1, 2, 3, 4, 5, 6
NOISE_PHILOX, NOISE_INJECTED, NOISE_NONE = 0, 1, 2
NCVX_NONE, NCVX_MC_TV, NCVX_ME_TV, NCVX_MC_TV_ANISO, NCVX_ME_TV_ANISO = 0, 1, 2, 3, 4
MAX_BLUR = 9
MAX_TV_ITERS = 64
(EPROX_LAPLACE, EPROX_UNCENTERED_LAPLACE, EPROX_GAUSSIAN, EPROX_GEN_GAUSSIAN_4_3, EPROX_GEN_GAUSSIAN_3_2,
 EPROX_GEN_GAUSSIAN_3, EPROX_GEN_GAUSSIAN_4, EPROX_HUBER, EPROX_SMOOTHED_LAPLACE, EPROX_EXP, EPROX_GAMMA,
 EPROX_CHI, EPROX_UNIFORM, EPROX_TRIANGULAR, EPROX_LAPLACE_CONJ) = range(15)


class LMCError(RuntimeError):
    """A liblmc_atomi call returned a negative lmc_status."""

    def __init__(self, code, msg):
        super().__init__(f"liblmc_atomi error {code}: {msg}")
        self.code = code


class lmc_problem(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("H", C.c_int32), ("W", C.c_int32),
        ("data_kind", C.c_int32),
        ("sigma_f", C.c_float),
        ("y_dev", C.c_void_p),
        ("mask_dev", C.c_void_p),
        ("kh", C.c_int32), ("kw", C.c_int32), ("oy", C.c_int32), ("ox", C.c_int32),
        ("h_host", C.POINTER(C.c_float)),
        ("prior_kind", C.c_int32),
        ("prior_sigma", C.c_float),
        ("tv_niter", C.c_int32),
        ("tv_step", C.c_float),
        ("tv_betas_host", C.POINTER(C.c_float)),
        ("ncvx_kind", C.c_int32),
        ("ncvx_lambda", C.c_float),
        ("ncvx_gamma", C.c_float),
        ("ncvx_niter", C.c_int32),
        # ABI 2
        ("tv_lagged_output", C.c_int32),
        ("tv_rtol", C.c_float),
        ("step_variant", C.c_int32),
        ("implicit_tol", C.c_float),
        ("tv_warm", C.c_int32),
        # ABI 3
        ("ncvx_rtol", C.c_float),
        ("tv_exit_path", C.c_int32),
        ("iterations_per_launch", C.c_int32),
        ("moments_overlap", C.c_int32),
        ("moments_bg_workgroups", C.c_int32),
        ("graph_replay", C.c_int32),
        ("eprox_kind", C.c_int32), ("eprox_p0", C.c_float), ("eprox_p1", C.c_float), ("eprox_scale_mask", C.c_int32),
        ("prox_scale", C.c_void_p), ("prox_scale_chain_stride", C.c_int64), ("prox_scale_pixel_stride", C.c_int32),
    ]


class lmc_myula_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("problem", lmc_problem),
        ("n_chains", C.c_int32),
        ("chain_offset", C.c_int64),
        ("tau", C.c_float), ("gamma", C.c_float), ("epsg", C.c_float),
        ("seed", C.c_uint64),
        ("noise_mode", C.c_int32),
        ("moments", C.c_int32),
        ("burn_in", C.c_int32),
        ("thin", C.c_int32),
    ]


class lmc_ulpda_config(C.Structure):
    _fields_ = [
        ("struct_size", C.c_uint32),
        ("problem", lmc_problem),
        ("n_chains", C.c_int32),
        ("chain_offset", C.c_int64),
        ("tau", C.c_float), ("mu", C.c_float), ("theta", C.c_float),
        ("gfirst", C.c_int32),
        ("cg_niter", C.c_int32),
        ("warm", C.c_int32),
        ("z_dev", C.c_void_p),
        ("seed", C.c_uint64),
        ("noise_mode", C.c_int32),
        ("moments", C.c_int32), ("burn_in", C.c_int32), ("thin", C.c_int32),
    ]


_P = C.c_void_p
_F = C.POINTER(C.c_float)
_SIGNATURES = {
    "lmc_version": (C.c_int, []),
    "lmc_last_error": (C.c_char_p, []),
    "lmc_device_info": (C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "lmc_hbm_copy_probe": (C.c_int, [C.c_size_t, C.c_int32, C.POINTER(C.c_float), _P]),
    "lmc_blur": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _F, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "lmc_gradient": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "lmc_gradient_adjoint": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, _P]),
    "lmc_fused_eval": (C.c_int, [C.POINTER(lmc_problem), _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, _P]),
    "lmc_l2_prox_workspace_bytes": (C.c_size_t, [C.c_int64, C.c_int32, C.c_int32]),
    "lmc_l2_prox": (C.c_int, [C.POINTER(lmc_problem), _P, _P, C.c_int64, C.c_float, C.c_int32, C.c_int32, _P, _P]),
    "lmc_energies": (C.c_int, [C.POINTER(lmc_problem), _P, C.c_int64, _P, _P, _P]),
    "lmc_mymala_create": (C.c_int, [C.POINTER(lmc_myula_config), C.POINTER(_P)]),
    "lmc_sampler_get_acceptance": (C.c_int, [_P, _P, _P, _P]),
    "lmc_set_cg_tolerance": (C.c_float, [C.c_float]),
    "lmc_chain_probes": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "lmc_haar_l1_prox": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, _P]),
    "lmc_dual_project": (C.c_int, [_P, _P, C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, _P]),
    "lmc_prox_elementwise": (C.c_int, [C.c_int32, _P, _P, C.c_int64, _F, C.c_int32, _P]),
    "lmc_myula_create": (C.c_int, [C.POINTER(lmc_myula_config), C.POINTER(_P)]),
    "lmc_sampler_destroy": (None, [_P]),
    "lmc_sampler_set_state": (C.c_int, [_P, _P, _P]),
    "lmc_sampler_get_state": (C.c_int, [_P, _P, _P]),
    "lmc_sampler_step": (C.c_int, [_P, C.c_int32, _P, _P]),
    "lmc_sampler_iteration": (C.c_int64, [_P]),
    "lmc_sampler_set_iteration": (C.c_int, [_P, C.c_int64]),
    "lmc_sampler_get_moments": (C.c_int, [_P, _P, _P, C.POINTER(C.c_uint64), _P]),
    "lmc_sampler_reset_moments": (C.c_int, [_P, _P]),
    "lmc_sampler_energies": (C.c_int, [_P, _P, _P, _P]),
    "lmc_sampler_noise": (C.c_int, [_P, C.c_int64, _P, _P]),
    "lmc_sampler_enable_timing": (C.c_int, [_P, C.c_int32]),
    "lmc_sampler_last_step_timing": (C.c_int, [_P, C.POINTER(C.c_float), C.POINTER(C.c_int32)]),
    "lmc_sampler_kernel_name": (C.c_char_p, [_P]),
    "lmc_sampler_tv_exit_stats": (C.c_int, [_P, C.c_int32, _P, C.POINTER(C.c_uint64), _P]),
    "lmc_set_step_variant": (C.c_int, [C.c_int32]),
    "lmc_ulpda_create": (C.c_int, [C.POINTER(lmc_ulpda_config), C.POINTER(_P)]),
    "lmc_sampler_set_dual": (C.c_int, [_P, _P, _P]),
    "lmc_sampler_get_dual": (C.c_int, [_P, _P, _P]),
    "lmc_sampler_set_steps": (C.c_int, [_P, C.c_float, C.c_float]),
    "lmc_rccl_available": (C.c_int, []),
    "lmc_rccl_unique_id": (C.c_int, [_P]),
    "lmc_rccl_comm_create": (C.c_int, [C.POINTER(_P), C.c_int32, C.c_int32, _P]),
    "lmc_rccl_comm_destroy": (C.c_int, [_P]),
    "lmc_allreduce_moments": (C.c_int, [_P, _P, _P, _P, C.POINTER(C.c_uint64), _P]),
}
RCCL_UNIQUE_ID_BYTES = 128
VARIANTS = ["auto", "tile", "(removed)", "split", "point", "block", "rows", "pipe"]

_lib = None
_lock = threading.Lock()


def load(path: str | None = None):
    """Load liblmc_atomi.so (once).  torch is imported first so that the HIP runtime that
    PyTorch-ROCm ships (same soname, libamdhip64.so.7) is the one the library binds to --
    device pointers of torch tensors are then valid in our launches."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        import torch  # noqa: F401  (HIP runtime first)
        p = path or os.environ.get("LMC_ATOMI_LIB", LIB_PATH)
        if not os.path.exists(p):
            raise ImportError(
                f"{p} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(or `make -C lmc_atomi_amd/csrc`).  lmc_atomi_amd has no CPU fallback.")
        lib = C.CDLL(p, mode=C.RTLD_GLOBAL)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(lib, name)          # AttributeError if the symbol is missing: fail loudly
            fn.restype = res
            fn.argtypes = args
        v = lib.lmc_version()
        if v != ABI_VERSION:
            raise ImportError(f"liblmc_atomi ABI {v} != binding ABI {ABI_VERSION}")
        _lib = lib
        return lib


def check(rc: int):
    if rc != 0:
        raise LMCError(rc, load().lmc_last_error().decode("utf-8", "replace"))


def exported_symbols():
    return sorted(_SIGNATURES)
