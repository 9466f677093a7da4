"""ABI 2 additions through the C ABI on the GPU: odd dual-iteration counts and the lagged-output switch of the TV prox, the warm-dual
variant, the library's own RCCL collective (lmc_allreduce_moments), per-handle solver tolerance."""
import ctypes as C

import numpy as np
import pytest

from oracle import lmc_oracle as O

pytestmark = pytest.mark.gpu

SIGMA = 0.75
GAMMA, TAU = SIGMA ** 2, 0.2 * SIGMA ** 2


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)


@pytest.fixture(scope="module")
def la():
    import torch
    assert torch.cuda.is_available()
    import lmc_atomi_amd as la
    return la


def synth(shape, k=5, seed=0):
    rng = np.random.default_rng(seed)
    img = np.zeros(shape)
    img[shape[0] // 5:shape[0] // 2, shape[1] // 6:2 * shape[1] // 3] = 160.0
    img[shape[0] // 2:, shape[1] // 2:] = 70.0
    img += np.linspace(0, 25, shape[1])[None, :]
    h = np.ones((k, k)) / (k * k)
    y = O.blur(img, h, (k // 2, k // 2)) + rng.normal(0, SIGMA, shape)
    return img, h, y


def oracle_steps(x0, y, h, k, prior, noise):
    return O.myula_batched(x0, y, h, (k // 2, k // 2), 1 / SIGMA ** 2, TAU, GAMMA, prior, noise.shape[0], lambda i: noise[i])


@pytest.mark.parametrize("shape,k,K", [((40, 264), 5, 9), ((33, 136), 5, 9), ((24, 512), 7, 9), ((40, 96), 5, 9),
                                       ((30, 264), 5, 19), ((26, 200), 5, 29)])
def test_odd_dual_iteration_counts(la, shape, k, K):
    """K = 9 in one launch (pipe: the last TV wave runs one stage; split below 132 columns), 19 / 29 as chained launches whose last
    link has 9 -- against the checker with the same K, injected noise."""
    img, h, y = synth(shape, k)
    rng = np.random.default_rng(K)
    C_, nit = 3, 3
    x0 = img[None] + rng.normal(0, 8, (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(k // 2, k // 2)), b=y, sigma=1 / SIGMA ** 2)
    smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=K), shape, n_chains=C_, tau=TAU, gamma=GAMMA, noise="injected")
    smp.set_state(x0)
    smp.step(nit, noise=noise)
    got = smp.get_state().cpu().numpy()
    ref = oracle_steps(x0, y, h, k, {"kind": "tv", "sigma": 0.3, "niter": K, "t": GAMMA}, noise)
    assert rel(got, ref) < 2e-5, rel(got, ref)
    if shape[1] > 128:
        assert "pipe" in smp.kernel_name, smp.kernel_name
    smp.close()


@pytest.mark.parametrize("shape,K", [((36, 264), 10), ((28, 144), 10), ((20, 64), 10), ((20, 264), 20), ((16, 48), 1), ((24, 136), 3)])
def test_lagged_output_is_one_update_fewer(la, shape, K):
    """tv_lagged_output: the prox after K - 1 dual updates (the other reading of pyproximal.TV's loop) -- equal to the
    checker's K - 1 prox and different from its K prox; the stateless ``TV.prox`` honours it too."""
    img, h, y = synth(shape)
    rng = np.random.default_rng(3)
    x0 = img[None] + rng.normal(0, 8, (2,) + shape)
    noise = rng.standard_normal((2, 2) + shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
    tv = la.TV(shape, sigma=0.3, niter=K, lagged_output=True)
    smp = la.MYULASampler(pf, tv, shape, n_chains=2, tau=TAU, gamma=GAMMA, noise="injected")
    smp.set_state(x0)
    smp.step(2, noise=noise)
    got = smp.get_state().cpu().numpy()
    pri = {"kind": "tv", "sigma": 0.3, "niter": K - 1, "t": GAMMA} if K > 1 else {"kind": "none"}
    ref = oracle_steps(x0, y, h, 5, pri, noise)
    full = oracle_steps(x0, y, h, 5, {"kind": "tv", "sigma": 0.3, "niter": K, "t": GAMMA}, noise)
    assert rel(got, ref) < 2e-5, rel(got, ref)
    # (measured: after 10 updates one more or less moves the state by 4e-7 rel-L2 -- which reading upstream takes is immaterial at the
    # 1e-3 north-star tolerance; for short proxes it is not)
    if K <= 3:
        assert rel(got, full) > 20 * rel(got, ref)
    else:
        assert rel(got, ref) <= rel(got, full)
    smp.close()
    px = tv.prox(x0[0].ravel(), GAMMA)
    want = O.tv_prox_fgp(x0[0], 0.3 * GAMMA, K - 1) if K > 1 else x0[0]
    assert rel(px.reshape(shape), want) < 5e-6


@pytest.mark.parametrize("shape,k,K", [((40, 264), 5, 1), ((40, 264), 5, 2), ((36, 512), 5, 3), ((30, 136), 7, 2), ((24, 200), 5, 3),
                                       ((20, 264), 7, 1)])
def test_warm_dual_tv_matches_the_checker(la, shape, k, K):
    """Warm-dual TV (K in {1, 3}; build extension): the projected dual is carried between MYULA iterations in HBM, the
    momentum restarts.  Against the checker's warm-dual prox with the same injected noise, over 6 iterations (so that the carried
    dual matters); set_state resets the dual."""
    img, h, y = synth(shape, k)
    rng = np.random.default_rng(40 + K)
    C_, nit = 3, 6
    x0 = img[None] + rng.normal(0, 8, (C_,) + shape)
    noise = rng.standard_normal((nit, C_) + shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(k // 2, k // 2)), b=y, sigma=1 / SIGMA ** 2)
    smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=K, warm=True), shape, n_chains=C_, tau=TAU, gamma=GAMMA, noise="injected")
    for _ in range(2):                        # the second pass checks that set_state starts again from a zero dual
        smp.set_state(x0)
        smp.iteration = 0
        smp.step(nit, noise=noise)
        got = smp.get_state().cpu().numpy()
        ref = oracle_steps(x0, y, h, k, {"kind": "tv", "sigma": 0.3, "niter": K, "t": GAMMA, "warm": True}, noise)
        cold = oracle_steps(x0, y, h, k, {"kind": "tv", "sigma": 0.3, "niter": K, "t": GAMMA}, noise)
        assert rel(got, ref) < 3e-5, rel(got, ref)
        assert rel(got, cold) > 10 * rel(got, ref)       # the carried dual is not a no-op
    assert "warm" in smp.kernel_name
    smp.close()
    # Philox mode and the moments go through the same launches
    a = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=K, warm=True), shape, n_chains=2, tau=TAU, gamma=GAMMA, seed=4, moments=True)
    a.step(3)
    assert a.moments()[2] == 6 and np.isfinite(a.get_state().cpu().numpy()).all()
    a.close()


def test_warm_dual_refusals(la):
    shape = (24, 64)          # too narrow for the full-width pipeline
    img, h, y = synth(shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
    with pytest.raises(la.LMCError, match="tv_warm"):
        la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=2, warm=True), shape, n_chains=1, tau=TAU, gamma=GAMMA)
    shape = (24, 264)
    img, h, y = synth(shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
    with pytest.raises(la.LMCError, match="tv_warm"):
        la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10, warm=True), shape, n_chains=1, tau=TAU, gamma=GAMMA)
    with pytest.raises(la.LMCError, match="tv_warm"):
        la.MYMALASampler(pf, la.TV(shape, sigma=0.3, niter=2, warm=True), shape, n_chains=1, tau=TAU, gamma=GAMMA)
    # tv_rtol > 0 excludes the warm dual and MYMALA
    with pytest.raises(la.LMCError, match="tv_rtol"):
        la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=2, warm=True, rtol=1e-4), shape, n_chains=1, tau=TAU, gamma=GAMMA)
    with pytest.raises(la.LMCError, match="tv_rtol"):
        la.MYMALASampler(pf, la.TV(shape, sigma=0.3, niter=10, rtol=1e-4), shape, n_chains=1, tau=TAU, gamma=GAMMA)


@pytest.mark.parametrize("shape,K", [((24, 136), 10), ((40, 264), 10), ((20, 64), 10), ((24, 136), 4)])
def test_tv_rtol_early_exit_matches_the_checker(la, shape, K):
    """pyproximal.TV's per-image early exit (rtol = 1e-4, upstream's default, which the reference's call leaves in force): the exact
    pass-by-pass device path against the checker's rtol branch, chain by chain, injected noise -- the chains leave their proxes after
    different numbers of passes (x0 differs per chain: one starts from zero, where the first proxes run to the end)."""
    img, h, y = synth(shape)
    rng = np.random.default_rng(11)
    C_, nit = 4, 8
    x0 = img[None] + rng.normal(0, 8, (C_,) + shape)
    x0[0] = 0.0
    noise = rng.standard_normal((nit, C_) + shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
    pg = la.TV(shape, sigma=0.3, niter=K, rtol=1e-4)
    smp = la.MYULASampler(pf, pg, shape, n_chains=C_, tau=TAU, gamma=GAMMA, noise="injected")
    smp.set_state(x0)
    smp.step(nit, noise=noise)
    got = smp.get_state().cpu().numpy()
    pri = {"kind": "tv", "sigma": 0.3, "niter": K, "t": GAMMA}
    ref = oracle_steps(x0, y, h, 5, dict(pri, rtol=1e-4), noise)
    fixed = oracle_steps(x0, y, h, 5, pri, noise)
    assert rel(got, ref) < 3e-5, rel(got, ref)
    assert rel(got, fixed) > 3 * rel(got, ref)          # the early exit is visible, and it is the checker's
    smp.close()
    # the stateless prox honours it too (lmc_fused_eval)
    px = pg.prox(x0[1].ravel(), GAMMA).reshape(shape)
    assert rel(px, O.tv_prox_fgp(x0[1], 0.3 * GAMMA, K, rtol=1e-4)) < 5e-6
    assert rel(px, O.tv_prox_fgp(x0[1], 0.3 * GAMMA, K)) > 1e-5


def test_drop_in_reproduces_the_reference_at_its_configured_rtol(la, golden):
    """tests/golden/algs_rtol.npz: the reference's OWN MYULA loop run with the TV prox keeping upstream's default rtol = 1e-4 (through the
    checker's rtol branch).  The drop-in with ``TV(rtol=1e-4)`` and the reference's PCG64 noise reproduces that trajectory; the
    fixed-count default reproduces the rtol = 0 one -- and the two stored trajectories differ by 1.5e-4."""
    g = golden("algs_rtol.npz")
    ny, nx, k, seed = (int(v) for v in g["meta"])
    sigma, tau_reg, tau, gamma = (float(v) for v in g["params"])
    shape = (ny, nx)
    pf = la.L2(Op=la.Convolve2D(shape, g["h"], offset=(k // 2, k // 2)), b=g["y"], sigma=1 / sigma ** 2)
    for tag, rtol in (("rtol1e-4", 1e-4), ("rtol0", 0.0)):
        xs = la.MoreauYosidaUnadjustedLangevin(pf, la.TV(shape, sigma=tau_reg, niter=10, rtol=rtol), np.zeros(ny * nx), tau=tau, gamma=gamma,
                                               niter=41, seed=seed, rng="pcg64")
        want = g[f"myula_tv_{tag}"][:5]                     # iterates 0, 10, 20, 30, 40
        assert rel(xs[::10], want) < 5e-5, (tag, rel(xs[::10], want))
    assert rel(g["myula_tv_rtol1e-4"][:5], g["myula_tv_rtol0"][:5]) > 5e-5


def test_allreduce_moments_through_the_c_abi(la):
    """lmc_allreduce_moments on real RCCL communicators of one rank: (i) one created by the library's own helpers
    (lmc_rccl_unique_id / lmc_rccl_comm_create), (ii) NULL = a job of one rank.  The all-reduce of a one-rank job must return the
    sampler's own accumulators; the N > 1 arithmetic is the same ncclAllReduce(sum) call (covered on CPU by the gloo tests)."""
    import torch
    lib = la._dev.lib()
    assert lib.lmc_rccl_available() == 1, la._capi.load().lmc_last_error()
    shape = (32, 136)
    img, h, y = synth(shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
    smp = la.MYULASampler(pf, la.TV(shape, sigma=0.3, niter=10), shape, n_chains=5, tau=TAU, gamma=GAMMA, seed=2, moments=True)
    smp.step(4)
    s1, s2, n = smp.moments()
    idb = C.create_string_buffer(la._capi.RCCL_UNIQUE_ID_BYTES)
    la._capi.check(lib.lmc_rccl_unique_id(idb))
    comm = C.c_void_p()
    la._capi.check(lib.lmc_rccl_comm_create(C.byref(comm), 1, 0, idb))
    assert comm.value
    for c in (comm, None):
        r1, r2, rn = smp.allreduce_moments(c)
        assert rn == n == 20
        assert torch.equal(r1, s1) and torch.equal(r2, s2)
    # the sampler's own accumulators are untouched: accumulate more, reduce again
    smp.step(1)
    r1, r2, rn = smp.allreduce_moments(comm)
    t1, t2, tn = smp.moments()
    assert rn == tn == 25 and torch.equal(r1, t1) and torch.equal(r2, t2)
    la._capi.check(lib.lmc_rccl_comm_destroy(comm))
    with pytest.raises(la.LMCError):
        la._capi.check(lib.lmc_rccl_comm_create(C.byref(comm), 2, 5, idb))
    smp.close()


def test_sharded_myula_over_rccl_world_of_one(la, tmp_path):
    """sharding.allreduce_sampler_moments on torch.distributed's "nccl" (= RCCL) backend at world size 1: the communicator comes from
    the process group (or is created through the C ABI) and the reduction is the library's own ncclAllReduce."""
    import torch
    import torch.distributed as dist
    if dist.is_initialized():
        pytest.skip("a process group already exists in this process")
    dist.init_process_group("nccl", init_method=f"file://{tmp_path}/rdzv", world_size=1, rank=0, device_id=torch.device("cuda", 0))
    try:
        shape = (24, 136)
        img, h, y = synth(shape)
        pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2)
        pg = la.TV(shape, sigma=0.3, niter=10)
        dist.barrier(device_ids=[0])
        comm = la.rccl_comm(None, torch.device("cuda", 0))
        assert comm
        smp = la.MYULASampler(pf, pg, shape, n_chains=4, tau=TAU, gamma=GAMMA, seed=6, moments=True)
        smp.step(3)
        s1, s2, n = smp.moments()
        r1, r2, rn = smp.allreduce_moments(comm)
        assert rn == n and torch.equal(r1, s1) and torch.equal(r2, s2)
        smp.close()
        mean, var, cnt, state = la.sharded_myula(pf, pg, shape, 4, np.zeros(shape), TAU, GAMMA, niter=3, seed=6)
        assert cnt == 12 and rel(mean.cpu().numpy(), (s1 / n).cpu().numpy()) == 0.0
        with pytest.raises(ValueError, match="cannot be sharded"):
            la.sharding.chain_shard(4, 1, 0) and la.sharded_myula(pf, pg, shape, 0, np.zeros(shape), TAU, GAMMA, niter=1)
    finally:
        dist.destroy_process_group()


def test_per_handle_implicit_tolerance(la):
    """implicit_tol of one ULPDA sampler (lmc_problem.implicit_tol) leaves the library default and other handles alone."""
    shape = (32, 136)
    img, h, y = synth(shape)
    pf = la.L2(Op=la.Convolve2D(shape, h, offset=(2, 2)), b=y, sigma=1 / SIGMA ** 2, niter=50, warm=True)
    pg = la.L21(ndim=2, sigma=0.3)
    assert la.set_cg_tolerance(1e-6) == pytest.approx(1e-6)
    outs = {}
    for name, tol in (("default", None), ("loose", 1e-2), ("tight", 1e-7)):
        smp = la.ULPDASampler(pf, pg, la.Gradient(shape), shape, n_chains=2, tau=0.95 * GAMMA, mu=1.0, gfirst=False, noise="none",
                              implicit_tol=tol)
        smp.set_state(img)
        smp.step(3)
        outs[name] = smp.get_state().cpu().numpy()
        smp.close()
    assert la.set_cg_tolerance(1e-6) == pytest.approx(1e-6)             # untouched
    assert rel(outs["tight"], outs["default"]) < 1e-5
    assert 1e-6 < rel(outs["loose"], outs["tight"]) < 5e-2              # a looser solve is visibly different, and only there
