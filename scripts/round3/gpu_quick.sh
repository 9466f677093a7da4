#!/bin/bash
set -o pipefail
out=gpurun_out/r3_quick; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_parity.py tests/test_gpu_rtol.py tests/test_gpu_deconv_driver.py tests/test_gpu_mymala.py tests/test_gpu_fullsize.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -5 $out/tests.log
[ $rc -eq 0 ] || exit $rc
timeout -k 10 600 python -m pytest tests/test_gpu_r4.py -x -q -m gpu -s -k "configured or north_star" 2>&1 | grep -E "R4|passed|failed" | tee $out/r4.log
B="python bench.py --steps 60 --warmup 20 --no-hbm-probe"
LMC_MOMENTS_OVERLAP=0 $B --no-cpu-baseline > $out/tv_inline.json 2> $out/tv_inline.err || exit 1
$B > $out/tv.json 2> $out/tv.err || exit 1
$B --no-cpu-baseline --ncvx mc --tv-rtol 1e-4 --warmup 60 > $out/mc_rtol.json 2> $out/mc_rtol.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_quick/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    print(f"{f.split('/')[-1]:22s} {d['ms_per_step']:8.4f} ms/step  launch {r['launch_ms']:.4f} ms  {r['kernel']}  value {d['value']:.0f}", d['config'].get('reference_as_configured', ''))
PY
