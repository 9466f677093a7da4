#!/bin/bash
# role -> SIMD mapping of the per-chain-exit kernel for chains with <= 4 live stages: M0 (shipped) against two others (build/var, -DLMC_RT_MAP)
set -o pipefail
out=gpurun_out/r3_rtmap; mkdir -p $out
B="python bench.py --steps 60 --warmup 60 --no-cpu-baseline --no-hbm-probe --tv-rtol 1e-4"
export LMC_BENCH_AS_CONFIGURED=0
for rep in 1 2; do
for v in M0 M1 M2; do
  if [ $v = M0 ]; then unset LMC_ATOMI_LIB; else export LMC_ATOMI_LIB=$PWD/build/var/liblmc_atomi_$v.so; fi
  $B > $out/${v}_$rep.json 2> $out/${v}_$rep.err || exit 1
  $B --ncvx mc > $out/${v}_mc_$rep.json 2> $out/${v}_mc_$rep.err || exit 1
done
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_rtmap/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:14s} launch {d['roofline']['launch_ms']:.3f} ms   step {d['ms_per_step']:.3f}  {d['config'].get('tv_exit',{}).get('passes_histogram_last_iteration')}")
PY
