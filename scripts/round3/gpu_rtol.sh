#!/bin/bash
# round 3: the device-side early exit (per-chain stage counts, objectives as by-products) -- parity tests, then timings at the headline size
set -o pipefail
out=gpurun_out/r3_rtol; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_rtol.py tests/test_gpu_abi2.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -15 $out/tests.log
[ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 40 --warmup 20 --no-cpu-baseline"
$B > $out/tv_fixed.json 2> $out/tv_fixed.err || exit 1
$B --tv-rtol 1e-4 > $out/tv_rtol.json 2> $out/tv_rtol.err || exit 1
$B --blur-k 7 --tv-rtol 1e-4 > $out/tv_rtol_k7.json 2> $out/tv_rtol_k7.err || exit 1
$B --ncvx me --ncvx-iters 50 --steps 10 --warmup 5 > $out/me_fixed.json 2> $out/me_fixed.err || exit 1
$B --ncvx me --ncvx-iters 50 --ncvx-rtol 1e-4 --steps 10 --warmup 5 > $out/me_rtol.json 2> $out/me_rtol.err || exit 1
$B --ncvx me --ncvx-iters 50 --ncvx-rtol 1e-4 --tv-rtol 1e-4 --steps 10 --warmup 5 > $out/me_rtol_both.json 2> $out/me_rtol_both.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_rtol/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f.split('/')[-1]:22s} {d['ms_per_step']:8.3f} ms/step  launch {d['roofline']['launch_ms']:.3f} ms  {d['roofline']['kernel']}  {d['config'].get('tv_exit', '')} {d['config'].get('ncvx_exit', '')}")
    except Exception as e:
        print(f, 'ERR', e)
PY
