#!/bin/bash
run() {
  v=$1; band=$2; shift; shift
  LMC_ROWS_BAND=$band LMC_VARIANT=$v timeout -k 10 120 python bench.py "$@" --no-cpu-baseline 2>/dev/null > gpurun_out/_b.json
  python - "$v band=$band $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f step_ms=%.4f frac=%.3f" % (r["launch_ms"], j["ms_per_step"], r["frac"]))
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_rows.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
run rows 0 --prior l2 --data blur --size 256 --chains 128 --steps 200 --warmup 20
run rows 0 --prior l2 --data blur --steps 50 --warmup 5
run rows 0 --prior l2 --data blur --steps 50 --warmup 5 --noise none
run auto 0 --steps 50 --warmup 5
run auto 0 --prior haar --data mask --steps 50 --warmup 5
