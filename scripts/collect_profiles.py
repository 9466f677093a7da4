#!/usr/bin/env python3
"""Copy the summaries of gpurun_out/prof_<round>_*/ (scripts/profile.sh) into profiles/ and assemble profiles/<round>_counters.json, the
file bench.py reads the per-launch HBM traffic and the VALU-busy fraction of the step kernel from.  usage: collect_profiles.py r02"""
import json, os, re, shutil, subprocess, sys
rnd = sys.argv[1] if len(sys.argv) > 1 else "r03"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import kernel_source_hash      # the hash bench.py compares before it quotes these counters
SRC_HASH = kernel_source_hash()
COMMIT = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip()
DIRTY = bool(subprocess.run(["git", "-C", ROOT, "status", "--porcelain", "--", "lmc_atomi_amd/csrc"], capture_output=True, text=True).stdout.strip())
T512 = {"H": 512, "W": 512, "C": 1024, "data": "blur", "tv_iters": 10, "ncvx": "none"}
WORK = {   # tag -> (bench.py workload key, sampler kernel name)
    "pipe": ({"H": 512, "W": 512, "C": 1024, "prior": "tv", "data": "blur", "tv_iters": 10, "ncvx": "none"}, "myula_step_pipe_kernel"),
    "rows": ({"H": 512, "W": 512, "C": 1024, "prior": "l2", "data": "blur", "tv_iters": 10, "ncvx": "none"}, "myula_step_rows_kernel"),
    "rowspair": ({"H": 512, "W": 512, "C": 1024, "prior": "l2", "data": "blur", "tv_iters": 10, "ncvx": "none"}, "myula_step_rows_pair_kernel"),
    "c2": ({"H": 256, "W": 256, "C": 128, "prior": "l2", "data": "blur", "tv_iters": 10, "ncvx": "none"}, "myula_step_rows_kernel"),
    "block": ({"H": 512, "W": 512, "C": 1024, "prior": "haar", "data": "mask", "tv_iters": 10, "ncvx": "none"}, "myula_step_block_kernel"),
    "blockpair": ({"H": 512, "W": 512, "C": 1024, "prior": "haar", "data": "mask", "tv_iters": 10, "ncvx": "none"}, "myula_step_block_kernel(2 iterations)"),
    "warm1": ({"H": 512, "W": 512, "C": 1024, "prior": "tv", "data": "blur", "tv_iters": 1, "ncvx": "none", "tv_warm": True}, "myula_step_pipe_kernel(warm)"),
    "warm2": ({"H": 512, "W": 512, "C": 1024, "prior": "tv", "data": "blur", "tv_iters": 2, "ncvx": "none", "tv_warm": True}, "myula_step_pipe_kernel(warm)"),
    "warm3": ({"H": 512, "W": 512, "C": 1024, "prior": "tv", "data": "blur", "tv_iters": 3, "ncvx": "none", "tv_warm": True}, "myula_step_pipe_kernel(warm)"),
    "wide877tv": ({"H": 667, "W": 877, "C": 512, "prior": "tv", "data": "blur", "tv_iters": 10, "ncvx": "none"}, "myula_step_pipe_kernel"),
    "wide877l2": ({"H": 667, "W": 877, "C": 512, "prior": "l2", "data": "blur", "tv_iters": 10, "ncvx": "none"}, "myula_step_rows_kernel"),
    # round 3
    "pipert": (dict(T512, prior="tv", tv_rtol=1e-4), "myula_step_pipe_kernel(per-chain exit)"),
    "pipe7": (dict(T512, prior="tv", blur_k=7), "myula_step_pipe_kernel"),
    "pipemc": (dict(T512, prior="tv", ncvx="mc"), "myula_step_pipe_kernel"),
    "rows7": (dict(T512, prior="l2", blur_k=7), "myula_step_rows_kernel"),
    "c5": ({"H": 512, "W": 512, "C": 512, "prior": "haar", "data": "mask", "tv_iters": 10, "ncvx": "mc"}, "myula_step_block_kernel"),
}
entries = []
for d in sorted(os.listdir(os.path.join(ROOT, "gpurun_out"))):
    m = re.match(r"prof_%s_(\w+)$" % rnd, d)
    if not m:
        continue
    tag = m.group(1)
    src = os.path.join(ROOT, "gpurun_out", d)
    for f, dst in (("kernel_stats.csv", f"{rnd}_{tag}_kernel_stats.csv"), ("summary.txt", f"{rnd}_{tag}_rocprofv3_summary.txt")):
        if os.path.exists(os.path.join(src, f)):
            shutil.copy(os.path.join(src, f), os.path.join(ROOT, "profiles", dst))
    if tag not in WORK or not os.path.exists(os.path.join(src, "counters.json")):
        continue
    work, kname = WORK[tag]
    cj = json.load(open(os.path.join(src, "counters.json")))
    base = kname.split("(")[0]
    for k, ent in cj["kernels"].items():
        if base in k and "traffic_bytes_per_launch" in ent:
            e = {"kernel": kname, "kernel_instantiation": k, "workload": work, "source": f"profiles/{rnd}_{tag}_rocprofv3_summary.txt",
                 "source_hash": SRC_HASH, "commit": COMMIT + ("+dirty" if DIRTY else ""),
                 "FETCH_SIZE_KiB": ent["counters"]["FETCH_SIZE"], "WRITE_SIZE_KiB": ent["counters"]["WRITE_SIZE"],
                 "traffic_bytes_per_launch": ent["traffic_bytes_per_launch"],
                 "algorithmic_bytes_per_launch": 8 * work["H"] * work["W"] * work["C"] * (2 if ("pair" in kname or "2 iterations" in kname) else 1),   # a pair launch = two iterations
                 "trace_avg_ns": ent.get("avg_ns"), "trace_calls": ent.get("calls"), "valu": ent.get("valu")}
            entries.append(e)
json.dump({"note": "per-launch counters of the step kernels from rocprofv3 (scripts/profile.sh: 60-step kernel trace; PMC in separate passes). "
                   "FETCH_SIZE / WRITE_SIZE are KiB; on gfx950 FETCH_SIZE reports 1/2 of the bytes of a 16-B-per-lane streaming read "
                   "(MI355X_MICROARCH.md) -> doubled in traffic_bytes_per_launch.  valu.busy_frac = 4 x SQ_ACTIVE_INST_VALU (quad-cycles) / "
                   "(1024 SIMDs x GRBM_GUI_ACTIVE / 8).",
           "entries": entries}, open(os.path.join(ROOT, "profiles", f"{rnd}_counters.json"), "w"), indent=1)
for e in entries:
    print(f"{e['kernel_instantiation'][:58]:58s} {e['trace_avg_ns'] / 1e6:7.4f} ms  frac {e['algorithmic_bytes_per_launch'] / (e['trace_avg_ns'] * 1e-9) / 8e12:5.3f}"
          f"  traffic x{e['traffic_bytes_per_launch'] / e['algorithmic_bytes_per_launch']:.3f}  VALU busy {e['valu']['busy_frac']:.3f}")
