#!/bin/bash
v=pipe
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
run build/exp_pslp/liblmc.so
run build/exp_pslp/liblmc.so --noise none
run build/exp_pnb/liblmc.so
run build/exp_pnb/liblmc.so --noise none
run build/exp_pslp/liblmc.so --chains 256
