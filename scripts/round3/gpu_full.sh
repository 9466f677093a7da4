#!/bin/bash
# the whole GPU suite, then the timings that changed in round 3
set -o pipefail
out=gpurun_out/r3_full; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu --durations=15 > $out/tests.log 2>&1; rc=$?
tail -30 $out/tests.log
[ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 40 --warmup 60 --no-cpu-baseline --no-hbm-probe"
$B > $out/tv_fixed.json 2> $out/tv_fixed.err || exit 1
$B --tv-rtol 1e-4 > $out/tv_rtol.json 2> $out/tv_rtol.err || exit 1
$B --ncvx mc > $out/mc.json 2> $out/mc.err || exit 1
$B --ncvx mc --tv-rtol 1e-4 > $out/mc_rtol.json 2> $out/mc_rtol.err || exit 1
$B --ncvx me --ncvx-iters 50 --ncvx-rtol 1e-4 --tv-rtol 1e-4 --steps 10 --warmup 30 > $out/me_rtol_both.json 2> $out/me_rtol_both.err || exit 1
$B --blur-k 7 --prior l2 > $out/l2_k7.json 2> $out/l2_k7.err || exit 1
$B --config 2 --steps 400 --warmup 100 > $out/c2.json 2> $out/c2.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_full/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f.split('/')[-1]:22s} {d['ms_per_step']:8.4f} ms/step  launch {d['roofline']['launch_ms']:.4f} ms  {d['roofline']['kernel']}  {d['config'].get('tv_exit', '')} {d['config'].get('ncvx_exit', '')}")
    except Exception as e:
        print(f, 'ERR', e)
PY
