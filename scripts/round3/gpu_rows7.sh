#!/bin/bash
# rows kernel at 7 taps x 8 pixels per lane: prefetch depths that fit the register file (A: PF 1 / y 1 ahead, B: PF 1 / y 0, C: PF 2 / y 0)
set -o pipefail
out=gpurun_out/r3_rows7; mkdir -p $out
B="python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe"
for v in A B C; do
  export LMC_ATOMI_LIB=$PWD/build/var/liblmc_atomi_$v.so
  for k in 6 7; do
    LMC_ROWS_PAIR=0 $B --blur-k $k --prior l2 > $out/l2_k${k}_$v.json 2> $out/l2_k${k}_$v.err || exit 1
  done
  $B --blur-k 7 --alg ulpda --steps 20 --warmup 5 > $out/ulpda_k7_$v.json 2> $out/ulpda_k7_$v.err || exit 1
done
unset LMC_ATOMI_LIB
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_rows7/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:22s} {d['ms_per_step']:8.3f} ms/step  launch {d['roofline']['launch_ms']:.3f} ms  {d['roofline']['kernel']}")
PY
