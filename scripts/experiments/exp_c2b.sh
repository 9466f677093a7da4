#!/bin/bash
for b in 8 16 24 32; do
  LMC_ROWS_BAND=$b timeout -k 10 120 python bench.py --prior l2 --size 256 --chains 128 --steps 300 --warmup 30 --no-cpu-baseline 2>/dev/null > gpurun_out/_b.json
  python - $b <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print("band", sys.argv[1], "launch_us=%.1f step_us=%.1f" % (1e3*j["roofline"]["launch_ms"], 1e3*j["ms_per_step"]))
PY
done
