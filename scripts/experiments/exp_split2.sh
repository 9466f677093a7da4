#!/bin/bash
run() {
  v=$1; shift
  LMC_VARIANT=$v timeout -k 10 120 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$v $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f" % r["launch_ms"])
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
run split
run split --tv-iters 5
run auto --size 256 --chains 1024
