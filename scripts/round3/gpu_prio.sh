#!/bin/bash
# s_setprio of the pipe kernel's roles (L C T N) after the new role pairing: A = 3 3 1 0 (shipped), B 3 3 2 0, C 3 3 1 1, D 2 3 1 0, E 3 2 1 0, F 0 0 0 0, G 3 3 0 0, H 1 1 3 0
set -o pipefail
out=gpurun_out/r3_prio; mkdir -p $out
B="python bench.py --steps 40 --warmup 10 --no-hbm-probe --no-cpu-baseline"
export LMC_BENCH_AS_CONFIGURED=0 LMC_MOMENTS_OVERLAP=0
for rep in 1 2; do
for v in A B C D E F G H; do
  if [ $v = A ]; then unset LMC_ATOMI_LIB; else export LMC_ATOMI_LIB=$PWD/build/var/liblmc_atomi_$v.so; fi
  $B > $out/${v}_$rep.json 2> $out/${v}_$rep.err || exit 1
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_prio/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:10s} launch {d['roofline']['launch_ms']:.4f} ms")
PY
