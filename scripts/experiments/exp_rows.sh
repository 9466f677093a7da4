#!/bin/bash
# rows kernel: band-size sweep at C2 (256x256x128) and 512x512x1024 (blur + l2 prior), against split / point
run() {
  v=$1; band=$2; shift; shift
  LMC_ROWS_BAND=$band LMC_VARIANT=$v timeout -k 10 120 python bench.py "$@" --no-cpu-baseline 2>/dev/null > gpurun_out/_b.json
  python - "$v band=$band $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f step_ms=%.4f frac=%.3f" % (r["launch_ms"], j["ms_per_step"], r["frac"]))
PY
}
for b in 0 16 32 64 128 256; do run rows $b --prior l2 --data blur --size 256 --chains 128 --steps 200 --warmup 20; done
run point 0 --prior l2 --data blur --size 256 --chains 128 --steps 200 --warmup 20
for b in 0 64 128 256 512; do run rows $b --prior l2 --data blur --steps 50 --warmup 5; done
run split 0 --prior l2 --data blur --steps 50 --warmup 5
run rows 0 --prior l2 --data blur --steps 50 --warmup 5 --no-moments
run rows 0 --prior l2 --data blur --steps 50 --warmup 5 --noise none
run rows 0 --prior haar --data blur --steps 50 --warmup 5
