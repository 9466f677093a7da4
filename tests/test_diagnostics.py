"""Convergence diagnostics across chains: the vectorised device-side split R-hat / ESS (run here on CPU
tensors) against the oracle's loop-by-loop restatement and against known answers; the chain all-gather over gloo."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import lmc_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ar1(rng, T, M, rho):
    y = np.zeros((T, M))
    e = rng.standard_normal((T, M))
    y[0] = e[0]
    for t in range(1, T):
        y[t] = rho * y[t - 1] + np.sqrt(1 - rho ** 2) * e[t]
    return y


@pytest.mark.parametrize("T,M", [(4, 2), (5, 3), (50, 4), (201, 7), (300, 1)])
def test_rhat_and_ess_match_oracle(T, M):
    from lmc_atomi_amd.diagnostics import ess, split_rhat
    rng = np.random.default_rng(T * 10 + M)
    Q = 5
    tr = np.stack([_ar1(rng, T, M, rho) + shift * np.arange(M)[None, :] for rho, shift in
                   [(0.0, 0.0), (0.5, 0.0), (0.9, 0.0), (0.3, 0.5), (-0.4, 0.0)]], axis=2).astype(np.float32)
    r = split_rhat(torch.from_numpy(tr)).numpy()
    e = ess(torch.from_numpy(tr)).numpy()
    e20 = ess(torch.from_numpy(tr), max_lag=20).numpy()
    for q in range(Q):
        np.testing.assert_allclose(r[q], O.split_rhat(tr[:, :, q]), rtol=1e-10)
        np.testing.assert_allclose(e[q], O.ess_geyer(tr[:, :, q]), rtol=1e-9)
        np.testing.assert_allclose(e20[q], O.ess_geyer(tr[:, :, q], max_lag=20), rtol=1e-9)


def test_known_answers():
    from lmc_atomi_amd.diagnostics import ess, split_rhat
    rng = np.random.default_rng(3)
    T, M = 4000, 8
    iid = rng.standard_normal((T, M))
    ar = _ar1(rng, T, M, 0.9)
    stuck = iid + 3.0 * np.arange(M)[None, :]
    tr = torch.from_numpy(np.stack([iid, ar, stuck], axis=2))
    r, e = split_rhat(tr).numpy(), ess(tr).numpy()
    assert abs(r[0] - 1) < 0.01 and abs(r[1] - 1) < 0.05 and r[2] > 2
    assert 0.8 * T * M < e[0] < 1.3 * T * M
    theory = T * M * (1 - 0.9) / (1 + 0.9)
    assert 0.7 * theory < e[1] < 1.4 * theory
    assert e[2] < 3 * M


def test_short_traces_give_nan_and_shapes_are_checked():
    from lmc_atomi_amd.diagnostics import ess, split_rhat
    t = torch.zeros((3, 4, 2)) + torch.arange(3)[:, None, None]
    assert torch.isnan(split_rhat(t)).all() and torch.isnan(ess(t)).all()
    assert split_rhat(torch.randn(10, 4)).shape == (1,)
    with pytest.raises(ValueError):
        split_rhat(torch.zeros(5))


def test_oracle_chain_probes_partition():
    rng = np.random.default_rng(0)
    x = rng.uniform(0, 255, (3, 13, 22))
    p = O.chain_probes(x, 3, 4)
    assert p.shape == (3, 12)
    assert np.allclose(O.chain_probes(x, 1, 1)[:, 0], x.mean(axis=(1, 2)))
    assert np.allclose(O.chain_probes(x, 13, 22), x.reshape(3, -1))
    # block means weighted by block sizes give back the image mean
    w = np.array([[(((a + 1) * 13 // 3) - a * 13 // 3) * (((b + 1) * 22 // 4) - b * 22 // 4) for b in range(4)] for a in range(3)]).ravel()
    assert np.allclose((p * w).sum(axis=1) / w.sum(), x.mean(axis=(1, 2)))


WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from lmc_atomi_amd.sharding import chain_shard, allgather_chains
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
full = torch.from_numpy(np.random.default_rng(5).standard_normal((9, 7, 3)).astype(np.float32))   # [T, C, Q], 7 chains: 4 + 3
off, cnt = chain_shard(7, world, rank)
got = allgather_chains(full[:, off:off + cnt].contiguous(), dim=1)
assert got.shape == full.shape and torch.equal(got, full), (rank, got.shape)
dist.destroy_process_group()
'''


def test_allgather_chains_two_rank_gloo_ragged():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ROOT], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
