#!/bin/bash
# round 3, first GPU call: the reference's 6x6 / 7x7 model blurs and config 5 as specified, before any kernel change
set -o pipefail
out=gpurun_out/r3_blurk; mkdir -p $out
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline"
for k in 5 6 7; do
  $B --blur-k $k > $out/tv_k$k.json 2> $out/tv_k$k.err || exit 1
  $B --blur-k $k --prior l2 > $out/l2_k$k.json 2> $out/l2_k$k.err || exit 1
  $B --blur-k $k --ncvx mc > $out/mc_k$k.json 2> $out/mc_k$k.err || exit 1
  $B --blur-k $k --ncvx me --ncvx-iters 50 --steps 10 --warmup 3 > $out/me_k$k.json 2> $out/me_k$k.err || exit 1
  $B --blur-k $k --alg ulpda --steps 20 --warmup 5 > $out/ulpda_k$k.json 2> $out/ulpda_k$k.err || exit 1
  echo "blur-k $k done"
done
$B --config 5 > $out/c5.json 2> $out/c5.err || exit 1
$B --config 5 --chains 1024 > $out/c5_1024.json 2> $out/c5_1024.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_blurk/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f.split('/')[-1]:18s} {d['ms_per_step']:8.3f} ms/step  launch {d['roofline']['launch_ms']:.3f} ms  {d['roofline']['kernel']}  frac {d['roofline']['frac']:.3f}")
    except Exception as e:
        print(f, 'ERR', e)
PY
