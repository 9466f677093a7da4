#!/bin/bash
# timing of alternative builds (LMC_ATOMI_LIB override): usage exp_pipe2.sh <variant> <expname>...
v=$1; shift
run() {
  lib=$1; shift
  LMC_ATOMI_LIB=$lib LMC_VARIANT=$v timeout -k 10 120 python bench.py "$@" --steps 30 --warmup 5 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$lib $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f" % r["launch_ms"])
PY
}
for l in "$@"; do run build/exp_$l/liblmc.so; done
