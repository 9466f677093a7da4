"""`sharded_myula` itself with more than one rank: two processes on the ONE GPU of the box (gloo process group: RCCL refuses two ranks on a
device), ragged shard 4 + 3 chains, against the same job run in one process -- the chains are keyed by global id, so the job-wide posterior
moments must agree to summation order, and every rank's final states must be the corresponding slice.  Also: fewer chains than ranks is
refused on every rank (nobody is left waiting in the collective)."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys, numpy as np, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
import lmc_atomi_amd as la
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=rank, world_size=world)
d = np.load(sys.argv[2])
shape = tuple(int(v) for v in d["y"].shape)
pf = la.L2(Op=la.Convolve2D(shape, d["h"], offset=(2, 2)), b=d["y"], sigma=1 / 0.5625)
pg = la.TV(shape, sigma=0.3, niter=10)
try:
    la.sharded_myula(pf, pg, shape, 1, np.zeros(shape), 0.1125, 0.5625, niter=1)
    refused = False
except ValueError:
    refused = True
mean, var, cnt, state = la.sharded_myula(pf, pg, shape, 7, np.zeros(shape), 0.1125, 0.5625, niter=9, seed=5, burn_in=2, thin=2)
np.savez(sys.argv[3] + f".{rank}.npz", mean=mean.cpu().numpy(), var=var.cpu().numpy(), count=cnt, state=state.cpu().numpy(), refused=refused)
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_myula_two_ranks_on_one_gpu(tmp_path):
    import torch
    assert torch.cuda.is_available()
    import lmc_atomi_amd as la
    from oracle import lmc_oracle as O
    shape = (24, 136)
    rng = np.random.default_rng(3)
    img = np.zeros(shape)
    img[5:18, 30:100] = 160.0
    h = np.ones((5, 5)) / 25.0
    y = O.blur(img, h, (2, 2)) + rng.normal(0, 0.75, shape)
    inp = str(tmp_path / "in.npz")
    np.savez(inp, y=y, h=h)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, "-c", WORKER, ROOT, inp, str(tmp_path / "out")], env=env))
    for p in procs:
        assert p.wait(timeout=300) == 0
    r0, r1 = np.load(str(tmp_path / "out.0.npz")), np.load(str(tmp_path / "out.1.npz"))
    assert bool(r0["refused"]) and bool(r1["refused"])
    # the same job in this process
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / 0.5625)
    pg = la.TV(shape, sigma=0.3, niter=10)
    mean, var, cnt, state = la.sharded_myula(pf, pg, shape, 7, np.zeros(shape), 0.1125, 0.5625, niter=9, seed=5, burn_in=2, thin=2)
    assert int(r0["count"]) == int(r1["count"]) == cnt == 7 * 4
    np.testing.assert_allclose(r0["mean"], mean.cpu().numpy(), rtol=1e-12)
    np.testing.assert_allclose(r1["mean"], r0["mean"], rtol=0, atol=0)
    np.testing.assert_allclose(r0["var"], var.cpu().numpy(), rtol=1e-9, atol=1e-9)
    st = state.cpu().numpy()
    np.testing.assert_array_equal(r0["state"], st[:4])          # rank 0 owns global chains 0..3, rank 1 chains 4..6
    np.testing.assert_array_equal(r1["state"], st[4:])
