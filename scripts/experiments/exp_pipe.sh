#!/bin/bash
run() {
  v=$1; shift
  LMC_VARIANT=$v timeout -k 10 120 python bench.py "$@" --no-cpu-baseline 2>/dev/null > gpurun_out/_b.json
  python - "$v $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f step_ms=%.4f frac=%.3f" % (r["launch_ms"], j["ms_per_step"], r["frac"]))
PY
}
run pipe --steps 30 --warmup 5
run split --steps 30 --warmup 5
run pipe --steps 30 --warmup 5 --chains 256
run pipe --steps 30 --warmup 5 --chains 512
run pipe --steps 30 --warmup 5 --noise none
