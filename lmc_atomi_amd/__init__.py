"""lmc_atomi_amd -- MI355X-native Langevin-Monte-Carlo inner loop (drop-in for the hot path of
192459/lmc-atomi): hand-written HIP kernels for gfx950 behind a C ABI (include/lmc_atomi.h),
reached from Python through ctypes.  No CPU fallback: the package raises if liblmc_atomi.so
is missing or no GPU is present."""
from . import _capi
from ._capi import LMCError
from .operators import Convolve2D, Diagonal, Gradient, Identity, LinearOperator
from .proximal import (L1, L2, L21, TV, L2_ncvx_tv, WaveletL1, ProxOperator, fgp_betas, ElementwiseProx, Laplace, UncenteredLaplace, Gaussian,
                       GenGaussian, Huber, SmoothedLaplace)
from .algs import (MYULAResult, MYULASampler, MYMALASampler, MoreauYosidaUnadjustedLangevin, MoreauYosidaMetropolisAdjustedLangevin, ULPDASampler,
                   UnadjustedLangevinPrimalDual, mean_var_from_moments,
                   set_step_variant, set_cg_tolerance)

from . import diagnostics, metrics
from .diagnostics import ChainTrace, chain_probes, ess, split_rhat
from .metrics import MetricsCallback, mean_squared_error, peak_signal_noise_ratio, signal_noise_ratio
from .sharding import (allgather_chains, allreduce_moments, allreduce_sampler_moments, chain_shard, posterior_mean_var, rccl_comm,
                       sharded_myula)

__all__ = [
    "diagnostics", "ChainTrace", "chain_probes", "ess", "split_rhat", "allgather_chains",
    "metrics", "MetricsCallback", "mean_squared_error", "peak_signal_noise_ratio", "signal_noise_ratio",
    "allreduce_moments", "allreduce_sampler_moments", "rccl_comm", "chain_shard", "posterior_mean_var", "sharded_myula",
    "LMCError", "Convolve2D", "Diagonal", "Gradient", "Identity", "LinearOperator",
    "L1", "L2", "L21", "TV", "L2_ncvx_tv", "WaveletL1", "ProxOperator", "fgp_betas",
    "MYULASampler", "MYMALASampler", "MoreauYosidaMetropolisAdjustedLangevin", "MYULAResult", "MoreauYosidaUnadjustedLangevin", "ULPDASampler", "UnadjustedLangevinPrimalDual", "mean_var_from_moments", "set_step_variant", "set_cg_tolerance",
]
__version__ = "0.2.0"
