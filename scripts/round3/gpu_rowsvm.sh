#!/bin/bash
# rows kernel after the optional-load copies: parity suites that reach it, then its timings (single launches, 5 / 7 taps, config 2, ULPDA through it)
set -o pipefail
out=gpurun_out/r3_rowsvm; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_ulpda.py tests/test_gpu_wide.py tests/test_gpu_eprox_prior.py tests/test_gpu_haar.py tests/test_gpu_epsg_array.py tests/test_gpu_abi2.py -x -q -m gpu > $out/tests.log 2>&1; rc=$?
tail -4 $out/tests.log
[ $rc -eq 0 ] || exit $rc
B="python bench.py --steps 60 --warmup 20 --no-hbm-probe --no-cpu-baseline"
LMC_ROWS_PAIR=0 $B --prior l2 > $out/l2_single.json 2> $out/l2_single.err || exit 1
$B --prior l2 > $out/l2_pair.json 2> $out/l2_pair.err || exit 1
$B --prior l2 --blur-k 7 > $out/l2_k7.json 2> $out/l2_k7.err || exit 1
$B --prior l2 --blur-k 6 > $out/l2_k6.json 2> $out/l2_k6.err || exit 1
$B --config 2 > $out/c2.json 2> $out/c2.err || exit 1
$B --alg ulpda --blur-k 7 --steps 20 --warmup 5 > $out/ulpda_k7.json 2> $out/ulpda_k7.err || exit 1
$B --alg ulpda --steps 20 --warmup 5 > $out/ulpda.json 2> $out/ulpda.err || exit 1
python - <<'PY'
import json, glob
for f in sorted(glob.glob('gpurun_out/r3_rowsvm/*.json')):
    d = json.loads(open(f).read().strip().splitlines()[-1]); r = d['roofline']
    print(f"{f.split('/')[-1]:18s} {d['ms_per_step']:8.4f} ms/step  launch {r['launch_ms']:.4f} ms  {r['kernel']}")
PY
