#!/bin/bash
# 256x256, 128 chains, 5x5 blur + l2 prior, every step-kernel variant
run() {
  v=$1; shift
  LMC_VARIANT=$v timeout -k 10 120 python bench.py "$@" --steps 200 --warmup 20 --no-cpu-baseline 2>/dev/null > gpurun_out/_b.json
  python - "$v $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f step_ms=%.4f frac=%.3f" % (r["launch_ms"], j["ms_per_step"], r["frac"]))
PY
}
for v in split point tile stream; do run $v --prior l2 --data blur --size 256 --chains 128; done
for v in split point; do run $v --prior l2 --data blur --size 256 --chains 128 --no-moments; done
for v in split point; do run $v --prior l2 --data blur --size 256 --chains 1024; done
