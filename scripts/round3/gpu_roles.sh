#!/bin/bash
# Free-running role timings of the headline kernel (no barrier, some waves leaving at once: results are wrong, only the launch time means anything):
# which role, alone or with the wave it shares a SIMD with, needs how long for its 4 x 536 ticks.  Variants are built on the host into build/var/.
set -o pipefail
out=gpurun_out/r3_roles; mkdir -p $out
B="python bench.py --steps 30 --warmup 8 --no-cpu-baseline --no-hbm-probe"
export LMC_BENCH_AS_CONFIGURED=0 LMC_MOMENTS_OVERLAP=0
$B > $out/base.json 2> $out/base.err || exit 1
for v in NB LCN TV L C N T1 P04 P15 P26 P37; do
  LMC_ATOMI_LIB=$PWD/build/var/liblmc_atomi_$v.so $B > $out/$v.json 2> $out/$v.err || exit 1
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_roles/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f"{f.split('/')[-1]:12s} launch {d['roofline']['launch_ms']:.3f} ms   step {d['ms_per_step']:.3f}")
PY
