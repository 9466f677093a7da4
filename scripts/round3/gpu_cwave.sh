#!/bin/bash
# after a change to the pipe kernel: its parity suites, then the headline / as-configured / MC-TV / 7-tap timings
set -o pipefail
out=gpurun_out/r3_cwave; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_parity.py tests/test_gpu_rtol.py tests/test_gpu_ncvx.py tests/test_gpu_mymala.py tests/test_gpu_wide.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -5 $out/tests.log
[ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 60 --warmup 20 --no-hbm-probe --no-cpu-baseline"
LMC_MOMENTS_OVERLAP=0 $B > $out/tv_inline.json 2> $out/tv_inline.err || exit 1
$B > $out/tv.json 2> $out/tv.err || exit 1
LMC_BENCH_AS_CONFIGURED=0 $B --ncvx mc > $out/mc.json 2> $out/mc.err || exit 1
LMC_BENCH_AS_CONFIGURED=0 $B --ncvx mc --tv-rtol 1e-4 --warmup 60 > $out/mc_rtol.json 2> $out/mc_rtol.err || exit 1
LMC_BENCH_AS_CONFIGURED=0 $B --blur-k 7 > $out/k7.json 2> $out/k7.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_cwave/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    ac = d['config'].get('reference_as_configured') or {}
    print(f"{f.split('/')[-1]:18s} {d['ms_per_step']:8.4f} ms/step  launch {r['launch_ms']:.4f} ms  {r['kernel']}  value {d['value']:.0f}", {k: ac[k] for k in ('ms_per_step', 'value') if k in ac})
PY
