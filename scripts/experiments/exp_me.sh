#!/bin/bash
# ME-TV term (inner TV prox of K iterations inside the gradient): chained pipe launches vs the tiled chunks
run() {
  v=$1; shift
  LMC_VARIANT=$v timeout -k 10 200 python bench.py "$@" --steps 10 --warmup 2 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$v $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", j["roofline"]["kernel"], "step_ms=%.3f" % j["ms_per_step"])
PY
}
run auto --ncvx me --ncvx-iters 10
run auto --ncvx me --ncvx-iters 50
run tile --ncvx me --ncvx-iters 50 --chains 128
run auto --ncvx me --ncvx-iters 50 --chains 128
run auto --tv-iters 50
