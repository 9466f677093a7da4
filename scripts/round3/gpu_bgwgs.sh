#!/bin/bash
# moment reduction beside an HBM-heavy step kernel (single rows launches): in line, and on the side stream with 256 .. 2048 / full-speed workgroups
set -o pipefail
out=gpurun_out/r3_bgwgs; mkdir -p $out
B="python bench.py --steps 60 --warmup 20 --no-hbm-probe --no-cpu-baseline --prior l2"
for k in 5 7; do
  P=""; [ $k = 5 ] && P="LMC_ROWS_PAIR=0"
  env $P LMC_MOMENTS_OVERLAP=0 $B --blur-k $k > $out/k${k}_inline.json 2> $out/k${k}_inline.err || exit 1
  for w in 256 512 1024 2048 0; do
    env $P LMC_MOMENTS_BG_WGS=$w $B --blur-k $k > $out/k${k}_bg$w.json 2> $out/k${k}_bg$w.err || exit 1
  done
done
for w in 256 1024 0; do LMC_MOMENTS_BG_WGS=$w $B > $out/pair_bg$w.json 2> $out/pair_bg$w.err || exit 1; done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_bgwgs/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    print(f"{f.split('/')[-1]:18s} {d['ms_per_step']:8.4f} ms/step  launch {r['launch_ms']:.4f} ms  {r['kernel']}")
PY
