#!/bin/bash
# K sweep on the headline shape (512x512 x 1024 chains, 5x5 blur): launch ms of the step kernel per TV iteration count
for k in 2 4 5 6 8 10 20 30 50; do
  timeout -k 10 200 python bench.py --tv-iters $k --steps 20 --warmup 3 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - $k <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print("K=%s" % sys.argv[1], j["roofline"]["kernel"], "step_ms=%.3f" % j["ms_per_step"])
PY
done
