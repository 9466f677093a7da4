#!/bin/bash
# whole GPU suite (moment reductions now on the side stream by default, pair launches included), then the closed-form / config-5 / headline timings
set -o pipefail
out=gpurun_out/r3_full2; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -8 $out/tests.log
[ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 60 --warmup 20 --no-cpu-baseline --no-hbm-probe"
$B > $out/tv.json 2> $out/tv.err || exit 1
LMC_MOMENTS_OVERLAP=0 $B > $out/tv_inline.json 2> $out/tv_inline.err || exit 1
$B --prior l2 > $out/l2.json 2> $out/l2.err || exit 1
LMC_MOMENTS_OVERLAP=0 $B --prior l2 > $out/l2_inline.json 2> $out/l2_inline.err || exit 1
$B --prior haar --data mask > $out/haar.json 2> $out/haar.err || exit 1
LMC_MOMENTS_OVERLAP=0 $B --prior haar --data mask > $out/haar_inline.json 2> $out/haar_inline.err || exit 1
$B --config 5 > $out/c5.json 2> $out/c5.err || exit 1
$B --config 5 --no-moments > $out/c5_nomom.json 2> $out/c5_nomom.err || exit 1
$B --config 2 --steps 400 --warmup 100 > $out/c2.json 2> $out/c2.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_full2/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        r = d['roofline']
        print(f"{f.split('/')[-1]:22s} {d['ms_per_step']:8.4f} ms/step  launch {r['launch_ms']:.4f} ms x {r['iterations_per_launch']:.0f} it  {r['kernel']}  value {d['value']:.0f}")
    except Exception as e:
        print(f, 'ERR', e)
PY
