#!/bin/bash
# per-chain exit with one live stage per wave (spread): parity suites of the RT paths, then timings of the chains as the reference configures them
set -o pipefail
out=gpurun_out/r3_spread; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_rtol.py tests/test_gpu_pipe.py tests/test_gpu_ncvx.py tests/test_gpu_r4.py -x -q -m gpu -k "not north_star and not ulpda" > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
[ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 60 --warmup 60 --no-hbm-probe --no-cpu-baseline"
LMC_BENCH_AS_CONFIGURED=0 $B --tv-rtol 1e-4 > $out/tv_rtol.json 2> $out/tv_rtol.err || exit 1
LMC_BENCH_AS_CONFIGURED=0 $B --tv-rtol 1e-4 --ncvx mc > $out/mc_rtol.json 2> $out/mc_rtol.err || exit 1
LMC_BENCH_AS_CONFIGURED=0 $B --tv-rtol 1e-4 --blur-k 7 > $out/k7_rtol.json 2> $out/k7_rtol.err || exit 1
LMC_BENCH_AS_CONFIGURED=0 $B --ncvx me --ncvx-iters 50 --ncvx-rtol 1e-4 --tv-rtol 1e-4 --steps 10 --warmup 30 > $out/me_rtol.json 2> $out/me_rtol.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_spread/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    print(f"{f.split('/')[-1]:18s} {d['ms_per_step']:8.4f} ms/step  launch {r['launch_ms']:.4f} ms  {r['kernel']}  {d['config'].get('tv_exit',{}).get('passes_histogram_last_iteration')}")
PY
