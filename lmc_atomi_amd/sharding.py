"""Chains across GPUs: one process per GPU, chains sharded by contiguous ranges of GLOBAL chain ids,
no data-path collective; the single collective is one all-reduce (RCCL over xGMI through
``torch.distributed``, backend "nccl"; "gloo" in the CPU tests) of the posterior moment images
{sum x, sum x^2, count} (2*H*W float64 + 1 scalar: 4 MiB at 512x512).

The reference runs exactly one chain in one process (algs.py:564); this module is what lets the
drop-in scale it out.  Because the device noise is keyed by (seed, iteration, global chain id, pixel)
every chain's trajectory is independent of the number of ranks.
"""
from __future__ import annotations

import torch


def chain_shard(n_total: int, world: int, rank: int):
    """(offset, count) of the contiguous block of global chain ids owned by ``rank``."""
    if not (0 <= rank < world):
        raise ValueError("rank outside world")
    base, rem = divmod(int(n_total), int(world))
    count = base + (1 if rank < rem else 0)
    offset = rank * base + min(rank, rem)
    return offset, count


def allreduce_moments(s1: torch.Tensor, s2: torch.Tensor, count: int, group=None):
    """Sum the per-rank accumulators over all ranks.  Returns (sum, sumsq, count) of the whole job.
    One collective call: the two images and the count travel in one packed float64 buffer."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return s1, s2, int(count)
    n = s1.numel()
    # "gloo" (CPU tests, and the two-ranks-on-one-GPU rehearsal of bench.py) reduces host tensors; "nccl" = RCCL reduces in HBM
    dev = torch.device("cpu") if (dist.get_backend(group) == "gloo" and s1.is_cuda) else s1.device
    packed = torch.empty(2 * n + 1, dtype=torch.float64, device=dev)
    packed[:n] = s1.reshape(-1).to(dev)
    packed[n:2 * n] = s2.reshape(-1).to(dev)
    packed[2 * n] = float(count)
    dist.all_reduce(packed, op=dist.ReduceOp.SUM, group=group)
    packed = packed.to(s1.device)
    return packed[:n].reshape(s1.shape), packed[n:2 * n].reshape(s2.shape), int(round(float(packed[2 * n])))


_own_comms = {}      # process group -> ncclComm_t created through the C ABI (lmc_rccl_comm_create)


def rccl_comm(group=None, device=None):
    """An ``ncclComm_t`` (as int) over the ranks of ``group`` with this process's GPU as its rank, for the C-ABI collective
    ``lmc_allreduce_moments``.  Preferred: the communicator ``torch.distributed``'s "nccl" (= RCCL) backend already holds for
    ``device``; if this PyTorch does not expose it, one is created through the C ABI (``lmc_rccl_unique_id`` on rank 0, the 128 bytes
    broadcast with ``torch.distributed``, ``lmc_rccl_comm_create`` on every rank) and cached.  Collective: call on every rank."""
    import ctypes as C
    import torch.distributed as dist
    from . import _capi, _dev
    import os
    pg = group if group is not None else dist.group.WORLD
    dev = _dev.device(device)
    if os.environ.get("LMC_RCCL_COMM", "torch") != "own":
        try:
            with torch.cuda.device(dev):           # the accessor answers for the CURRENT device
                ptr = int(pg._get_backend(dev)._comm_ptr())
            if ptr:
                return ptr
        except Exception:   # older / different PyTorch: no accessor -- fall through to a communicator of our own
            pass
    key = id(pg)
    if key in _own_comms:
        return _own_comms[key]
    lib = _dev.lib()
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    buf = C.create_string_buffer(_capi.RCCL_UNIQUE_ID_BYTES)
    if rank == 0:
        _capi.check(lib.lmc_rccl_unique_id(buf))
    box = [bytes(buf.raw)]
    dist.broadcast_object_list(box, src=dist.get_global_rank(pg, 0) if group is not None else 0, group=group)
    comm = C.c_void_p()
    with torch.cuda.device(dev):
        _capi.check(lib.lmc_rccl_comm_create(C.byref(comm), world, rank, C.create_string_buffer(box[0], _capi.RCCL_UNIQUE_ID_BYTES)))
    _own_comms[key] = int(comm.value)
    return _own_comms[key]


def allreduce_sampler_moments(smp, group=None):
    """Job-wide (sum, sumsq, count) of a sampler's posterior-moment accumulators.  Backend "nccl": ONE ``ncclAllReduce`` issued by
    the library itself (``lmc_allreduce_moments``, C ABI) on the RCCL communicator of the group -- no torch op in between;
    "gloo" (CPU tests, several ranks rehearsed on one GPU): the packed host all-reduce of :func:`allreduce_moments`."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return smp.moments()
    if dist.get_backend(group) == "nccl":
        return smp.allreduce_moments(rccl_comm(group, smp.device))
    s1, s2, cnt = smp.moments()
    return allreduce_moments(s1, s2, cnt, group)


def allgather_chains(t: torch.Tensor, dim: int = 1, group=None):
    """Concatenate per-rank tensors along their chain dimension in rank order (= global chain order under
    :func:`chain_shard`); ranks may own different numbers of chains.  Used for the diagnostics trace ``[T, C_rank, Q]``
    (a few MB), so that R-hat / ESS run over the chains of the whole job.  Two small collectives (counts, padded payload)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return t
    world = dist.get_world_size(group)
    t = t.movedim(dim, 0).contiguous()
    cnt = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
    counts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(counts, cnt, group=group)
    counts = [int(c.item()) for c in counts]
    cmax = max(counts)
    pad = torch.zeros((cmax,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[:t.shape[0]] = t
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)], dim=0).movedim(0, dim)


def posterior_mean_var(s1, s2, count):
    mean = s1 / count
    return mean, s2 / count - mean * mean


def sharded_myula(proxf, proxg, dims, n_chains_total, x0, tau, gamma, epsg=1.0, niter=10, seed=0,
                  burn_in=0, thin=1, group=None, device=None):
    """Run ``n_chains_total`` MYULA chains split over the ranks of the default process group (or run
    them all here when torch.distributed is not initialised) and return the job-wide posterior
    (mean, var, count) plus this rank's final states."""
    import torch.distributed as dist
    from .algs import MYULASampler
    world = dist.get_world_size(group) if (dist.is_available() and dist.is_initialized()) else 1
    rank = dist.get_rank(group) if world > 1 else 0
    if int(n_chains_total) < world:      # every rank sees the same numbers: all of them raise, none is left waiting in the collective
        raise ValueError(f"{n_chains_total} chains cannot be sharded over {world} ranks (every rank needs at least one)")
    offset, count = chain_shard(n_chains_total, world, rank)
    smp = MYULASampler(proxf, proxg, dims, n_chains=count, tau=tau, gamma=gamma, epsg=epsg, seed=seed,
                       chain_offset=offset, moments=True, burn_in=burn_in, thin=thin, device=device)
    try:
        smp.set_state(x0)
        smp.step(niter)
        s1, s2, cnt = allreduce_sampler_moments(smp, group)
        state = smp.get_state()
    finally:
        smp.close()
    mean, var = posterior_mean_var(s1, s2, cnt)
    return mean, var, cnt, state
