#!/bin/bash
run() {
  timeout -k 10 120 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f" % r["launch_ms"])
PY
}
timeout -k 10 200 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_haar.py -m gpu -x -q 2>&1 | tail -2
run
run --noise none
run --chains 256
run --prior haar --data mask
