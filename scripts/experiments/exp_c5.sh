#!/bin/bash
# SURVEY 8(d) C5 shapes: mask + Haar-l1 (+ non-convex term) at 512x512x1024 per GPU
run() {
  timeout -k 10 120 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/_b.json
  python - "$*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.3f step_ms=%.3f frac=%.3f" % (r["launch_ms"], j["ms_per_step"], r["frac"]))
PY
}
run --prior haar --data mask
run --prior haar --data mask --ncvx mc
run --prior haar --data mask --ncvx me --tv-iters 10
run --prior haar --data blur
run --prior tv --data blur --ncvx mc
run --prior tv --data blur --ncvx me --tv-iters 10
run --prior l2 --data blur --size 256 --chains 128
