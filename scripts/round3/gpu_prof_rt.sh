#!/bin/bash
# kernel traces of the device-side early exit (TV prior, ME-TV inner prox), the new tests, and the config-2 prefetch question
set -o pipefail
out=gpurun_out/r3_prof_rt; mkdir -p $out
export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests/test_gpu_eprox_prior.py tests/test_gpu_rtol.py tests/test_gpu_ncvx.py tests/test_gpu_pipe.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -12 $out/tests.log
[ $rc -eq 0 ] || exit $rc
A="--repeats 1 --no-cpu-baseline --no-hbm-probe"
rocprofv3 --kernel-trace --stats -d $out/tv_rtol --output-format csv -- python3 bench.py --steps 40 --warmup 60 $A --tv-rtol 1e-4 > $out/tv_rtol.json 2> $out/tv_rtol.log || exit 1
rocprofv3 --kernel-trace --stats -d $out/me_rtol --output-format csv -- python3 bench.py --steps 10 --warmup 30 $A --ncvx me --ncvx-iters 50 --ncvx-rtol 1e-4 --tv-rtol 1e-4 > $out/me_rtol.json 2> $out/me_rtol.log || exit 1
rocprofv3 --kernel-trace --stats -d $out/mc --output-format csv -- python3 bench.py --steps 40 --warmup 10 $A --ncvx mc > $out/mc.json 2> $out/mc.log || exit 1
for d in tv_rtol me_rtol mc; do
  f=$(find $out/$d -name '*kernel_stats.csv' | head -1); cp $f $out/${d}_kernel_stats.csv; echo "== $d"; head -8 $f | cut -c1-200
done
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_prof_rt/*.json')):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f"{f.split('/')[-1]:22s} {d['ms_per_step']:8.4f} ms/step  launch {d['roofline']['launch_ms']:.4f} ms  {d['roofline']['kernel']}  {d['config'].get('tv_exit', '')} {d['config'].get('ncvx_exit', '')}")
    except Exception as e:
        print(f, 'ERR', e)
PY
