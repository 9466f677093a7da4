#!/bin/bash
run() {
  v=$1; shift
  LMC_VARIANT=$v timeout -k 10 120 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$v $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f frac=%.3f" % (r["launch_ms"], r["frac"]))
PY
}
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
run pipe --size 256 --chains 1024
run split --size 256 --chains 1024
run pipe --size 256 --chains 4096
run split --size 256 --chains 4096
run pipe --size 128 --chains 4096
run split --size 128 --chains 4096
