#!/bin/bash
run() {
  timeout -k 10 200 python bench.py "$@" --steps 20 --warmup 3 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", j["roofline"]["kernel"], "step_ms=%.3f launch_ms=%.3f" % (j["ms_per_step"], j["roofline"]["launch_ms"]))
PY
}
timeout -k 10 300 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_ncvx.py -m gpu -x -q 2>&1 | tail -2
run --ncvx me --ncvx-iters 10
run --ncvx me --ncvx-iters 50
run
