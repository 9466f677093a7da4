#!/usr/bin/env python3
"""bench.py -- LMC chain-iterations/s on MI355X (BASELINE.json metric), with the HBM roofline of the
fused step kernel and the CPU oracle timed beside it.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--repeats R] [--tv-warm]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one MYULA iteration (algs.py:569) of every chain resident on the GPU.  Workload at
N = 1: BASELINE.json configs[2] -- 512x512 deblurring (5x5 uniform box blur as prox_lmc_deconv.py:55-58,
Gaussian noise sigma = 0.75) + isotropic TV prox (niter_tv = 10 dual iterations), 1024 chains.  For N > 1
every rank runs 1024 chains of its own (weak scaling, configs[3]: 8192 chains on 8 GPUs), chains keyed by
global id, and the only collective is one RCCL all-reduce of the posterior moment images at the end of the
timed region.  Inputs are synthetic and resident in HBM before the timed region starts.

Protocol: W warm-up steps, then the timed region of EXACTLY K steps (barrier + synchronize on both sides, max over
ranks) is run R = 3 times back to back and the MEDIAN region is reported (`repeats`, `ms_per_step_all`); the step kernel's launch time
is the HIP-event average over the launches of that median region, on the launch stream.  `roofline` carries the spec peak, the
bandwidth a state-shaped copy measures on this device in this process (`peak_measured`), and a `valu` block (vector-ALU busy
fraction of the same kernel from the committed rocprofv3 counters): at K = 10 the update is bound by VALU issue, not by HBM.
Where a launch advances every chain by TWO iterations (closed-form priors and the Haar prior: `--prior l2|l1|haar`; not the headline),
`roofline.iterations_per_launch` = 2 and `algorithmic_bytes_per_launch` counts both (8 B per pixel per iteration).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)
BYTES_PER_PIXEL_STEP = 8       # algorithmic: one fp32 read of x_k + one fp32 write of x_{k+1}


def kernel_source_hash():
    """sha256 (12 hex digits) over the HIP sources of the library: profiles/r03_counters.json entries carry the hash of the sources they were
    measured on, and the bench line quotes `roofline.traffic` / `roofline.valu` from them only while the sources are still the same."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "lmc_atomi_amd", "csrc", "*.h*"))):
        h.update(os.path.basename(f).encode())
        h.update(open(f, "rb").read())
    return h.hexdigest()[:12]


def synth_problem(H, W, sigma, seed=0, blur="box", blur_k=5):
    """Piecewise-constant + ramp ground truth in [0,255] from default_rng(1234); y = H u + N(0, sigma^2)
    with noise from default_rng(seed) (mirrors prox_lmc_deconv.py:53-59).  Host-side setup, not timed."""
    rng = np.random.default_rng(1234)
    u = np.zeros((H, W))
    for _ in range(12):
        i0, j0 = rng.integers(0, H - 8), rng.integers(0, W - 8)
        i1, j1 = rng.integers(i0 + 4, H + 1), rng.integers(j0 + 4, W + 1)
        u[i0:i1, j0:j1] = rng.uniform(20, 235)
    u += np.linspace(0, 20, W)[None, :]
    u = np.clip(u, 0, 255)
    h = np.ones((5, 5)) / 25.0            # the reference's uniform box (prox_lmc_deconv.py:55-59)
    if blur == "gaussian":                 # BASELINE.json says "Gaussian-deblur": 5x5, s = 1; same kernels, runtime taps
        t = np.exp(-0.5 * (np.arange(5) - 2.0) ** 2)
        h = np.outer(t, t) / np.sum(np.outer(t, t))
    # blur with scipy (setup only): zero-padded 'same' convolution, centred 5x5
    import scipy.signal
    y = scipy.signal.convolve2d(u, h, mode="same") + np.random.default_rng(seed).normal(0, sigma, (H, W))
    if blur_k != 5:     # the reference's mismatched models: the observation is always made with the 5x5 box, the MODEL blurs with k x k
        h = np.ones((blur_k, blur_k)) / float(blur_k * blur_k)     # (prox_lmc_deconv.py:61-69, 102-103)
    return u, h, y


def cpu_baseline(H, W, h, y, sigma, prior, label, mask=None, chains=4, iters=20, offset=(2, 2)):
    """The oracle on a bounded sample of the same workload, timed on this host: the C restatement
    (oracle/lmc_oracle_c.c, float64, OpenMP over chains, PCG64 noise from numpy) on up to 16 cores -- the GPU box's CPU
    share for one GPU -- is the reported value; the numpy restatement (the one pinned by the reference's own outputs;
    the two agree bit for bit, tests/test_oracle_c.py) and the C one on a single core are reported beside it.
    The Haar prior exists in the numpy restatement only.  Checker / baseline only -- never on the product path."""
    from oracle import lmc_oracle as O
    from oracle import lmc_oracle_c as OC
    OC.build()
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2
    hh, off = (None, None) if mask is not None else (h, offset)

    def run(step, n_chains, n_it, **kw):
        rng = np.random.default_rng(0)
        x = np.zeros((n_chains, H, W))
        t0 = time.perf_counter()
        for _ in range(n_it):
            x = step(x, y, hh, off, 1 / sigma ** 2, tau, gamma, prior, rng.standard_normal(x.shape), mask=mask, **kw)
        dt = time.perf_counter() - t0
        return n_chains * n_it / dt, dt

    v_np, t_np = run(O.myula_step, chains, max(1, iters // 2))
    if prior["kind"] == "haar":
        return {"value": v_np, "unit": "chain-it/s", "cores": 1, "kind": "port",
                "sample": f"{chains} chains x {max(1, iters // 2)} iterations of the same {H}x{W} {label} workload, oracle/lmc_oracle.py float64 numpy on one core, {t_np:.1f} s"}
    v_c1, t_c1 = run(OC.myula_step, chains, iters, threads=1)
    T = max(1, min(16, os.cpu_count() or 1, OC.max_threads()))
    mc, mi = 4 * T, 2 * iters
    v_mt, t_mt = run(OC.myula_step, mc, mi, threads=T)
    return {"value": v_mt, "unit": "chain-it/s", "cores": T, "kind": "port",
            "sample": f"{mc} chains x {mi} iterations of the same {H}x{W} {label} workload, "
                      f"oracle/lmc_oracle_c.c float64 with {T} OpenMP threads, {t_mt:.1f} s on {T} of {os.cpu_count()} host cores",
            "one_core": {"c": v_c1, "numpy": v_np, "unit": "chain-it/s",
                         "sample": f"{chains} chains x {iters} (C, {t_c1:.1f} s) / {max(1, iters // 2)} (numpy, {t_np:.1f} s) iterations"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--width", type=int, default=0, help="image width if not square (experiments)")
    ap.add_argument("--chains", type=int, default=1024, help="chains per GPU")
    ap.add_argument("--tv-iters", type=int, default=10)
    ap.add_argument("--prior", default="tv", choices=["tv", "l2", "l1", "haar"])
    ap.add_argument("--thin", type=int, default=1, help="accumulate posterior moments every thin-th iteration")
    ap.add_argument("--alg", default="myula", choices=["myula", "ulpda", "mymala"], help="sampler: MYULA (headline) or ULPDA (algs.py:295-474)")
    ap.add_argument("--cg-iters", type=int, default=50, help="ULPDA: inner CG iterations of the implicit data step")
    ap.add_argument("--no-moments", action="store_true")
    ap.add_argument("--data", default="blur", choices=["blur", "identity", "mask"], help="data term (experiments)")
    ap.add_argument("--blur", default="box", choices=["box", "gaussian"], help="5x5 blur: the reference's uniform box or a Gaussian (s=1)")
    ap.add_argument("--blur-k", type=int, default=5, choices=[5, 6, 7],
                    help="size of the model's uniform box blur, offset (k//2, k//2): the reference's H5 / H6 / H7 (prox_lmc_deconv.py:55-69)")
    ap.add_argument("--mask-p", type=float, default=0.5, help="--data mask: fraction of observed pixels (SURVEY 8(d) C5: Bernoulli(0.5), default_rng(7))")
    ap.add_argument("--ncvx", default="none", choices=["none", "mc", "me"],
                    help="add the L2_ncvx_tv Moreau-difference term (lamda=0.3, gamma=15; SURVEY 8(d) C5)")
    ap.add_argument("--ncvx-iters", type=int, default=None, help="inner TV-prox iterations of the ME-TV term (default: --tv-iters; the reference uses niter_l2 = 50)")
    ap.add_argument("--noise", default="philox", choices=["philox", "none"], help="noise source (experiments)")
    ap.add_argument("--tau-scale", type=float, default=1.0, help="multiplies the step size tau = 0.2 sigma^2 (MYMALA acceptance experiments)")
    ap.add_argument("--repeats", type=int, default=3, help="timed regions of --steps steps; the median is reported (SURVEY 8(d))")
    ap.add_argument("--tv-warm", action="store_true",
                    help="warm-dual TV (SURVEY 8(d) C3 variant; build extension): carry the TV dual between MYULA iterations, "
                         "--tv-iters in {1, 2, 3} per iteration (+16 B/px of HBM traffic, quoted as `actual_bytes_per_launch`)")
    ap.add_argument("--tv-rtol", type=float, default=0.0,
                    help="pyproximal.TV's per-image early exit (1e-4 = the reference as configured, prox_lmc_deconv.py:122; 0 = always --tv-iters passes, the headline)")
    ap.add_argument("--ncvx-rtol", type=float, default=0.0, help="--ncvx me: early exit of the inner TV prox (1e-4 = the class's own default, algs.py:130,169)")
    ap.add_argument("--tv-lagged", action="store_true", help="TV prox after tv_iters - 1 dual updates (lmc_problem.tv_lagged_output)")
    ap.add_argument("--no-hbm-probe", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--config", type=int, default=0, choices=[0, 2, 3, 4, 5],
                    help="BASELINE.json configuration by number (per-GPU shard): 2 = 256x256 deblur + l2 prior, 128 chains; 3 / 4 = 512x512 deblur + TV K=10, "
                         "1024 chains per GPU (the default; 4 = the same over 8 GPUs); 5 = 512x512 inpainting mask + Haar-l1 prior, 512 chains per GPU")
    ap.add_argument("--cpu-chains", type=int, default=4)
    ap.add_argument("--cpu-iters", type=int, default=20)
    args = ap.parse_args()
    if args.config == 2:
        args.size, args.chains, args.prior = 256, 128, "l2"
    elif args.config == 5:
        # Bernoulli(0.5) mask + Haar-l1 prox + the L2_ncvx_tv Moreau-difference (MC-TV) term that makes it non-log-concave
        args.size, args.chains, args.prior, args.data, args.ncvx = 512, 512, "haar", "mask", "mc"

    # stdout carries ONE JSON line and nothing else: libraries that print banners on fd 1 (RCCL's version block at communicator set-up)
    # are sent to stderr for the whole run; the line itself goes to the saved descriptor
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("for --gpus N > 1 launch with torch.distributed.run --nproc-per-node N (one rank per GPU)")
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # rehearsal of the N > 1 code on a one-GPU box: LMC_BENCH_DEVICE=0 puts every rank on GPU 0, LMC_BENCH_BACKEND=gloo replaces RCCL
    # (which refuses two ranks on one GPU); the numbers of such a run mean nothing
    backend = os.environ.get("LMC_BENCH_BACKEND", "nccl")
    if os.environ.get("LMC_BENCH_DEVICE"):
        local_rank = int(os.environ["LMC_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    # LMC_BENCH_DIST=1 forces the RCCL path (process group, barrier, moment all-reduce, max-over-ranks) at world size 1:
    # the rehearsal of the N > 1 code on a one-GPU box
    use_dist = world > 1 or os.environ.get("LMC_BENCH_DIST") == "1"
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29517")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    import lmc_atomi_amd as la
    if os.environ.get("LMC_VARIANT"):                # A/B runs of the step-kernel variants (scripts/bench_variants.py)
        la.set_step_variant(os.environ["LMC_VARIANT"])

    H = W = args.size
    if args.width:
        W = args.width
    C = args.chains
    sigma, tau_reg = 0.75, 0.3                       # prox_lmc_deconv.py:40 defaults
    gamma, tau = sigma ** 2, 0.2 * sigma ** 2 * args.tau_scale        # prox_lmc_deconv.py:92-94
    u, h, y = synth_problem(H, W, sigma, blur=args.blur, blur_k=args.blur_k)
    boff = (args.blur_k // 2, args.blur_k // 2)
    if args.data == "blur":
        pf = la.L2(Op=la.Convolve2D((H, W), h, offset=boff), b=y, sigma=1 / sigma ** 2)
    elif args.data == "mask":                        # inpainting: a Bernoulli(mask_p) mask from default_rng(7)
        m = (np.random.default_rng(7).uniform(size=(H, W)) < args.mask_p).astype(np.float32)
        pf = la.L2(Op=la.Diagonal(m, dims=(H, W)), b=m * u, sigma=1 / sigma ** 2, dims=(H, W))
    else:
        pf = la.L2(b=y, sigma=1 / sigma ** 2, dims=(H, W))
    if args.ncvx != "none":                          # prox_lmc_deconv.py:106-113 (niter of the inner TV prox = --tv-iters)
        pf = la.L2_ncvx_tv(dims=(H, W), Op=pf.Op, Op2=la.Gradient((H, W)) if args.ncvx == "mc" else None, b=pf.b, sigma=1 / sigma ** 2,
                           lamda=tau_reg, gamma=15.0, isotropic=True, niter=args.ncvx_iters or args.tv_iters, rtol=args.ncvx_rtol)
    pg = {"tv": lambda: la.TV((H, W), sigma=tau_reg, niter=args.tv_iters, warm=args.tv_warm, lagged_output=args.tv_lagged, rtol=args.tv_rtol), "l2": lambda: la.L2(sigma=0.05),
          "l1": lambda: la.L1(sigma=tau_reg), "haar": lambda: la.WaveletL1((H, W), sigma=tau_reg)}[args.prior]()
    if args.alg == "ulpda":      # prox_lmc_deconv.py:88-90,455-457: tau0 = 0.95 sigma^2, mu0 = 1, theta = 1, gfirst = False
        pf.niter = args.cg_iters
        smp = la.ULPDASampler(pf, la.L21(ndim=2, sigma=tau_reg), la.Gradient((H, W)), (H, W), n_chains=C, tau=0.95 * sigma ** 2,
                              mu=1.0, theta=1.0, gfirst=False, seed=0, chain_offset=rank * C, moments=not args.no_moments,
                              burn_in=0, thin=args.thin, noise=args.noise)
    else:
        cls = la.MYMALASampler if args.alg == "mymala" else la.MYULASampler
        smp = cls(pf, pg, (H, W), n_chains=C, tau=tau, gamma=gamma, seed=0, chain_offset=rank * C,
                  moments=not args.no_moments, burn_in=0, thin=args.thin, noise=args.noise)
    smp.set_state(np.zeros((H, W), dtype=np.float32))        # x0 = 0 (prox_lmc_deconv.py:135)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
            torch.cuda.synchronize()

    from lmc_atomi_amd.sharding import allreduce_sampler_moments
    collective = "none"
    if use_dist and not args.no_moments:
        collective = ("lmc_allreduce_moments (C ABI): one ncclAllReduce of {sum x, sum x^2, count} on the process group's RCCL communicator"
                      if backend == "nccl" else "torch.distributed gloo all-reduce of the packed accumulators (rehearsal)")
    smp.step(args.warmup)
    reduce_fn = allreduce_sampler_moments
    if not args.no_moments:
        if use_dist:                                 # warm the collective too (communicator set-up, buffers of this size)
            try:
                allreduce_sampler_moments(smp)
            except Exception as exc:                 # plumbing, not the compute path: keep the job alive and say so in the JSON line
                from lmc_atomi_amd.sharding import allreduce_moments

                def reduce_fn(sm):
                    return allreduce_moments(*sm.moments())
                reduce_fn(smp)
                collective = f"torch.distributed all-reduce of the packed accumulators (C-ABI collective unavailable: {exc})"
        smp.reset_moments()
    # one HIP-event pair per step-kernel launch (the roofline leg) -- except on small configurations, where the two event records cost
    # 6 us of a 40 us iteration (measured at 256 x 256 x 128) and would be what the bench measures: there the launch time is wall / steps
    timed_launches = args.alg == "myula" and os.environ.get("LMC_BENCH_NO_TIMING") != "1" and H * W * C > (1 << 25)
    # Round 3: `value` is measured as the sampler runs by default -- posterior-moment reductions on a side stream under the next step kernel --
    # and the step kernel's own launch time (the roofline leg) in ONE MORE region of K steps after the timed ones, with per-launch HIP events
    # switched on, which also takes the reductions back in line so that the kernel is timed alone.
    regions = []                                     # (elapsed s of K steps, step-kernel ms summed over its launches, launches)
    for _ in range(max(1, args.repeats)):
        if not args.no_moments:
            smp.reset_moments()
        sync_all()
        t0 = time.perf_counter()
        smp.step(args.steps)
        if use_dist and not args.no_moments:
            s1, s2, cnt = reduce_fn(smp)                     # RCCL over xGMI: posterior mean/var accumulators of the whole job
        sync_all()
        el = time.perf_counter() - t0
        if use_dist:
            te = torch.tensor([el], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
            dist.all_reduce(te, op=dist.ReduceOp.MAX)
            el = float(te.item())
        regions.append((el, el * 1e3, args.steps))   # ULPDA / MYMALA are sequences of launches per iteration, small configurations are not event-timed: the whole iteration
    order = sorted(range(len(regions)), key=lambda i: regions[i][0])
    elapsed, kern_ms, launches = regions[order[len(order) // 2]]          # the median region
    if timed_launches:           # the roofline leg: the same K steps once more, every step-kernel launch between its own pair of HIP events
        smp.enable_timing(True)
        sync_all()
        smp.step(args.steps)
        sync_all()
        kern_ms, launches = smp.last_step_timing()
        smp.enable_timing(False)

    if rank == 0:
        # HBM bytes per launch from the committed rocprofv3 PMC passes (they cannot be collected inside this process);
        # only quoted when the profile is of this kernel on this workload
        traffic, valu = None, None
        want = {"H": H, "W": W, "C": C, "prior": args.prior, "data": args.data, "tv_iters": args.tv_iters, "ncvx": args.ncvx}
        if args.blur_k != 5:
            want["blur_k"] = args.blur_k
        if args.tv_rtol:
            want["tv_rtol"] = args.tv_rtol
        if args.ncvx_rtol:
            want["ncvx_rtol"] = args.ncvx_rtol
        if args.tv_warm:
            want["tv_warm"] = True
        if args.tv_lagged:
            want["tv_lagged"] = True
        counters_from = None
        src_hash = kernel_source_hash()
        try:             # the newest committed counters of this kernel on this workload -- only while the kernel sources are the ones they were taken on
            for ent in json.load(open(os.path.join(ROOT, "profiles", "r03_counters.json")))["entries"]:
                if ent["workload"] == want and ent["kernel"] == smp.kernel_name and traffic is None and ent.get("source_hash") == src_hash:
                    traffic = ent["traffic_bytes_per_launch"]
                    counters_from = {"file": ent.get("source"), "source_hash": ent.get("source_hash"), "commit": ent.get("commit")}
                    if "valu" in ent:
                        valu = dict(ent["valu"], source=ent.get("source"))
        except Exception:
            pass
        peak_measured = None
        if not args.no_hbm_probe:
            import ctypes
            g = ctypes.c_float()
            la._capi.check(la._dev.lib().lmc_hbm_copy_probe(1 << 30, 3, ctypes.byref(g), None))     # 1 GiB read + 1 GiB written, as a state pass
            peak_measured = float(g.value)
        value = C * world * args.steps / elapsed
        per_launch_ms = kern_ms / launches
        its_per_launch = args.steps / launches if timed_launches else 1.0      # 2 where a launch advances every chain by two iterations
        achieved = BYTES_PER_PIXEL_STEP * H * W * C * its_per_launch / (per_launch_ms * 1e-3) / 1e9
        prior_desc = (f"isotropic TV prox K={args.tv_iters}{' warm-dual' if args.tv_warm else ''}{' lagged output' if args.tv_lagged else ''}{f' early exit rtol={args.tv_rtol:g}' if args.tv_rtol else ''} (tau_reg={tau_reg})" if args.prior == "tv"
                      else f"{args.prior} prior")
        out = {
            "metric": "lmc_chain_iterations_per_s",
            "value": value,
            "unit": "chain-it/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "repeats": len(regions), "statistic": "median of the timed regions",
            "ms_per_step_all": [r[0] / args.steps * 1e3 for r in regions],
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{H}x{W} MYULA " + {"blur": f"deblur ({args.blur_k}x{args.blur_k} {'uniform box' if args.blur == 'box' else 'Gaussian s=1'} blur, sigma={sigma})", "mask": f"inpainting (Bernoulli({args.mask_p}) mask)",
                                                   "identity": "denoise"}[args.data] + f" + {prior_desc}" + ({"none": "", "mc": " - MC-TV term (lamda=0.3, gamma=15)", "me": f" - ME-TV term (lamda=0.3, gamma=15, {args.ncvx_iters or args.tv_iters} inner its)"}[args.ncvx]) + ", "
                            f"{C} chains/GPU x {world} GPU, Philox noise, x0=0, "
                            + ("no moments" if args.no_moments else f"posterior moments every {args.thin} it"),
                "image": [H, W], "chains_per_gpu": C, "chains_total": C * world, "tv_iters": args.tv_iters,
                "sampler": "MYULA (algs.py:477-587)" if args.alg == "myula" else "MYMALA (Metropolis-adjusted MYULA, generalises prox_lmc.py:134-158)" if args.alg == "mymala" else f"ULPDA (algs.py:295-474), implicit step by CG: at most {args.cg_iters} iterations, stops at |r| <= 1e-6 |b| for every chain (the reference solver's rule, algs.py:250)", "parallelism": f"chains sharded x{world}",
                "moment_reductions": ("none" if args.no_moments else "every kept iterate, on a side stream under the next step kernel (lmc_sampler_step default; "
                                      "roofline.launch_ms is measured in a separate event-timed region with the reductions in line, i.e. the kernel alone)"),
                "iterations_per_s": args.steps / elapsed,
                "collective": collective,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": smp.kernel_name,
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "peak_measured": peak_measured,
                "frac_of_measured": (achieved / peak_measured) if peak_measured else None,
                "peak_measured_how": "lmc_hbm_copy_probe: 1 GiB read + 1 GiB written, float4 per lane, best of 10 launch shapes x 3 passes, this process",
                "traffic": traffic,
                "counters_from": counters_from,      # null (and traffic / valu null) when the HIP sources have changed since the committed PMC passes
                "kernel_source_hash": src_hash,
                "launch_ms": per_launch_ms,
                "launches": launches,
                "algorithmic_bytes_per_launch": int(BYTES_PER_PIXEL_STEP * H * W * C * its_per_launch),
                "iterations_per_launch": its_per_launch,
                # what actually limits the kernel when it is not HBM: vector-ALU busy fraction from the committed counters
                # (SQ_ACTIVE_INST_VALU quad-cycles x 4 / (SIMDs x kernel cycles), profiles/): ~1 means VALU-issue bound
                "valu": valu,
                "limiter": ("valu" if (valu and valu.get("busy_frac", 0) > 0.6 and achieved / HBM_PEAK_GBS < 0.3) else "hbm"),
            },
        }
        if args.tv_warm:      # the un-fused variant reports its ACTUAL bytes beside the algorithmic 8 B/px
            out["roofline"]["actual_bytes_per_launch"] = (BYTES_PER_PIXEL_STEP + 16) * H * W * C
            out["roofline"]["actual_gbs"] = (BYTES_PER_PIXEL_STEP + 16) * H * W * C / (per_launch_ms * 1e-3) / 1e9
            out["config"]["tv_warm"] = True
        if args.tv_rtol and args.prior == "tv" and args.alg == "myula":
            try:
                ps, rr = smp.tv_exit_stats("prior")
                pc = np.bincount(ps.cpu().numpy(), minlength=args.tv_iters + 1)
                out["config"]["tv_exit"] = {"passes_histogram_last_iteration": pc.tolist(), "reruns_after_round_1_2_3_4": rr,
                                            "chain_iterations": int(C * (args.warmup + args.steps * (len(regions) + (1 if timed_launches else 0))))}
            except Exception as exc:
                out["config"]["tv_exit"] = f"pass-by-pass path ({exc})"
        if args.ncvx_rtol and args.ncvx == "me" and args.alg == "myula":
            ps, rr = smp.tv_exit_stats("ncvx")
            pc = np.bincount(ps.cpu().numpy(), minlength=(args.ncvx_iters or args.tv_iters) + 1)
            out["config"]["ncvx_exit"] = {"passes_histogram_last_iteration": pc.tolist(), "reruns_after_round_1_2_3_4": rr}
        headline = (args.prior == "tv" and args.alg == "myula" and args.ncvx == "none" and args.data == "blur" and args.tv_iters == 10 and not args.tv_warm
                    and not args.tv_lagged and H == 512 and W == 512)
        if headline and not args.tv_rtol:
            # what binds the K = 10 kernel is the vector ALU, not HBM: the bound of ITS instruction stream (VALU issue slots from the counters)
            out["roofline"]["valu_bound_ms"] = {
                "perfectly_balanced": 1.16 * C / 1024.0, "whole_waves_on_simds": 1.29 * C / 1024.0,
                "how": "SQ_ACTIVE_INST_VALU (4-cycle issue slots) of the shipped kernel and of free-running single-role builds (profiles/r03_pipe_role_counters.txt): "
                       "per tick L 165, T1 159, T2..T5 190.5 each, C 52, N 163 = 1281 slots per CU; 320 per SIMD (541 ns at 2.37 GHz) if it could be split evenly; with "
                       "whole waves on SIMDs the busiest pair (L beside a two-stage TV wave, whatever the pairing) carries 355.5 (600 ns); x 536 ticks x 4 rounds of 256 workgroups"}
        if headline and not args.tv_rtol and world == 1 and os.environ.get("LMC_BENCH_AS_CONFIGURED", "1") != "0":
            # beside the headline (fixed K = 10 passes): the same chain AS THE REFERENCE IS CONFIGURED -- pyproximal.TV's default rtol = 1e-4, which
            # prox_lmc_deconv.py:122 leaves in force; the exit is decided on the device, chain by chain.  60 warm-up iterations: the pass counts
            # fall 10 -> 4 over the first ~30 iterations from x0 = 0, every change a re-run of the chains concerned.
            smp.close()
            pg2 = la.TV((H, W), sigma=tau_reg, niter=10, rtol=1e-4)
            smp2 = la.MYULASampler(pf, pg2, (H, W), n_chains=C, tau=tau, gamma=gamma, seed=0, chain_offset=rank * C, moments=not args.no_moments)
            smp2.set_state(np.zeros((H, W), dtype=np.float32))
            smp2.step(60)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            smp2.step(args.steps)
            torch.cuda.synchronize()
            el2 = time.perf_counter() - t0
            ps, rr = smp2.tv_exit_stats("prior")
            out["config"]["reference_as_configured"] = {
                "what": "TV(niter=10, rtol=1e-4): every chain leaves its prox in the pass upstream's loop leaves it in; same kernels, per-chain live stage counts",
                "ms_per_step": el2 / args.steps * 1e3, "value": C * args.steps / el2, "unit": "chain-it/s", "steps": args.steps, "warmup": 60,
                "passes_histogram_last_iteration": np.bincount(ps.cpu().numpy(), minlength=11).tolist(), "reruns_after_round_1_2_3_4": rr}
            smp2.close()
        if args.alg == "mymala":
            out["config"]["acceptance_rate_mean"] = float(smp.acceptance_rate().mean())
            out["config"]["tau_scale"] = args.tau_scale
        if world == 1 and not args.no_cpu_baseline and args.alg == "myula" and args.ncvx == "none" and not args.tv_warm:
            oprior = {"tv": {"kind": "tv", "sigma": tau_reg, "niter": args.tv_iters - (1 if args.tv_lagged else 0), "t": gamma, "rtol": args.tv_rtol},
                      "l2": {"kind": "l2", "sigma": 0.05, "t": gamma}, "l1": {"kind": "l1", "sigma": tau_reg, "t": gamma},
                      "haar": {"kind": "haar", "sigma": tau_reg, "t": gamma}}[args.prior]
            if args.data == "blur":
                oy, omask, oh = y, None, h
            elif args.data == "mask":
                omask = (np.random.default_rng(7).uniform(size=(H, W)) < args.mask_p).astype(np.float64)
                oy, oh = omask * u, None
            else:
                oy, omask, oh = y, None, None
            label = f"MYULA {args.data} + {args.prior}" + (f"(K={args.tv_iters})" if args.prior == "tv" else "")
            out["cpu_baseline"] = cpu_baseline(H, W, oh, oy, sigma, oprior, label, mask=omask, chains=args.cpu_chains, iters=args.cpu_iters, offset=boff)
        sys.stdout.flush()
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    smp.close()
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
