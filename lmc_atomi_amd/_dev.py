"""Tensor plumbing: PyTorch-ROCm tensors are used purely as HBM containers (allocation,
pointers, streams).  No torch op runs on the hot path."""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from . import _capi


def device(dev=None) -> torch.device:
    if not torch.cuda.is_available():
        raise RuntimeError("lmc_atomi_amd needs a ROCm GPU (torch.cuda.is_available() is False); "
                           "there is no CPU fallback")
    if dev is None:
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device(dev)


def stream_ptr(dev=None) -> C.c_void_p:
    """Current stream of ``dev`` (default: the current device) as a ``hipStream_t``."""
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def to_dev(x, dev=None) -> torch.Tensor:
    """numpy / torch (any dtype, any device) -> contiguous fp32 tensor in HBM: on ``dev`` if given, else where a GPU tensor already
    lives, else on the current device."""
    if dev is None and isinstance(x, torch.Tensor) and x.is_cuda:
        dev = x.device
    d = device(dev)
    if isinstance(x, torch.Tensor):
        return x.to(device=d, dtype=torch.float32).contiguous()
    return torch.from_numpy(np.ascontiguousarray(np.asarray(x), dtype=np.float32)).to(d)


def like_input(t: torch.Tensor, proto):
    """Return `t` in the container type of `proto`: numpy in -> numpy out (dtype of proto if
    floating), torch in -> torch out."""
    if isinstance(proto, torch.Tensor):
        return t
    a = t.detach().cpu().numpy()
    pd = np.asarray(proto).dtype
    return a.astype(pd) if np.issubdtype(pd, np.floating) else a


def ptr(t) -> C.c_void_p:
    return C.c_void_p(0 if t is None else t.data_ptr())


def fptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def lib():
    return _capi.load()


def run(t: torch.Tensor, name: str, *args):
    """Call the stateless entry point ``name(*args, stream)`` on the device ``t`` lives on, on that device's current stream."""
    with torch.cuda.device(t.device):
        _capi.check(getattr(lib(), name)(*args, stream_ptr()))
