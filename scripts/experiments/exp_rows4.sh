#!/bin/bash
run() {
  timeout -k 10 120 python bench.py "$@" --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f frac=%.3f" % (r["launch_ms"], r["frac"]))
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_rows.py tests/test_gpu_ulpda.py -m gpu -x -q 2>&1 | tail -2
run --prior l2 --steps 50 --warmup 5
run --prior l2 --steps 50 --warmup 5 --noise none
run --prior l2 --size 256 --chains 128 --steps 200 --warmup 20
run --prior l2 --size 256 --chains 1024 --steps 100 --warmup 10
