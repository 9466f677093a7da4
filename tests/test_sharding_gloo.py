"""N > 1 path on CPU: world_size-2 gloo.  The chain partition and the moment all-reduce are exercised
for real (two processes, torch.distributed); the per-rank accumulators come from the oracle driven by
the counter-based noise keyed by GLOBAL chain id, so the 2-rank result must equal the 1-rank result."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from oracle import lmc_oracle as O
from lmc_atomi_amd.sharding import chain_shard, allreduce_moments, posterior_mean_var
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
H, W, C, nit, seed = 12, 16, 7, 6, 11          # 7 chains over 2 ranks: ragged split 4 + 3
off, cnt = chain_shard(C, world, rank)
rng = np.random.default_rng(0)
y = rng.normal(100, 20, (H, W)); h = np.ones((5, 5)) / 25
prior = {"kind": "tv", "sigma": 0.3, "niter": 4, "t": 0.5625}
x, s1, s2, n = O.myula_batched(np.zeros((cnt, H, W)), y, h, (2, 2), 1 / 0.5625, 0.1125, 0.5625, prior, nit,
                               lambda k: O.philox_normals(seed, k, np.arange(off, off + cnt), H, W).astype(np.float64),
                               moments=True, burn_in=2, thin=2)
g1, g2, gn = allreduce_moments(torch.from_numpy(s1), torch.from_numpy(s2), n)
mean, var = posterior_mean_var(g1, g2, gn)
if rank == 0:
    np.savez(sys.argv[2], mean=mean.numpy(), var=var.numpy(), count=gn)
dist.destroy_process_group()
'''


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_chain_shard_partitions_exactly():
    from lmc_atomi_amd.sharding import chain_shard
    for total in (1, 7, 8, 1024, 8192, 13):
        for world in (1, 2, 3, 8):
            seen = []
            for r in range(world):
                off, cnt = chain_shard(total, world, r)
                seen.extend(range(off, off + cnt))
            assert seen == list(range(total))
    with pytest.raises(ValueError):
        chain_shard(8, 2, 2)


def test_two_rank_gloo_moments_equal_single_rank(tmp_path):
    from oracle import lmc_oracle as O
    out = str(tmp_path / "res.npz")
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ROOT, out], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    got = np.load(out)
    H, W, C, nit, seed = 12, 16, 7, 6, 11
    rng = np.random.default_rng(0)
    y = rng.normal(100, 20, (H, W)); h = np.ones((5, 5)) / 25
    prior = {"kind": "tv", "sigma": 0.3, "niter": 4, "t": 0.5625}
    _, s1, s2, n = O.myula_batched(np.zeros((C, H, W)), y, h, (2, 2), 1 / 0.5625, 0.1125, 0.5625, prior, nit,
                                   lambda k: O.philox_normals(seed, k, np.arange(C), H, W).astype(np.float64),
                                   moments=True, burn_in=2, thin=2)
    assert int(got["count"]) == n == C * 2
    np.testing.assert_allclose(got["mean"], s1 / n, rtol=1e-13, atol=1e-12)
    np.testing.assert_allclose(got["var"], s2 / n - (s1 / n) ** 2, rtol=1e-9, atol=1e-9)


def test_allreduce_is_identity_without_process_group():
    import torch
    from lmc_atomi_amd.sharding import allreduce_moments
    a, b = torch.ones(3, 4, dtype=torch.float64), torch.full((3, 4), 2.0, dtype=torch.float64)
    g1, g2, n = allreduce_moments(a, b, 5)
    assert n == 5 and torch.equal(g1, a) and torch.equal(g2, b)
