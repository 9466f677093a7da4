#!/bin/bash
# block-kernel sweep: prior x data at 512x512x1024 (kernel name, launch ms, whole-step ms, roofline frac)
for p in haar l1 l2; do for d in mask identity; do
  timeout -k 10 120 python bench.py --prior $p --data $d --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null > gpurun_out/_b.json
  python - "$p" "$d" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], sys.argv[2], r["kernel"], "launch_ms=%.3f step_ms=%.3f frac=%.3f" % (r["launch_ms"], j["ms_per_step"], r["frac"]))
PY
done; done
