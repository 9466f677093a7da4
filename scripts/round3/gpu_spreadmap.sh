#!/bin/bash
set -o pipefail
out=gpurun_out/r3_spreadmap; mkdir -p $out
B="python bench.py --steps 60 --warmup 60 --no-hbm-probe --no-cpu-baseline --tv-rtol 1e-4"
export LMC_BENCH_AS_CONFIGURED=0
for rep in 1 2; do
for v in S3 S4 S5; do
  if [ $v = S3 ]; then unset LMC_ATOMI_LIB; else export LMC_ATOMI_LIB=$PWD/build/var/liblmc_atomi_$v.so; fi
  $B > $out/${v}_$rep.json 2> $out/${v}_$rep.err || exit 1
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_spreadmap/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:14s} launch {d['roofline']['launch_ms']:.3f} ms   step {d['ms_per_step']:.3f}")
PY
