#!/bin/bash
run() {
  lib=$1; shift
  LMC_ATOMI_LIB=$lib timeout -k 10 120 python bench.py "$@" --steps 40 --warmup 5 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$lib $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
print(sys.argv[1], "|", "launch_ms=%.4f" % j["roofline"]["launch_ms"])
PY
}
run lmc_atomi_amd/lib/liblmc_atomi.so
for l in "$@"; do run build/exp_$l/liblmc.so; done
run lmc_atomi_amd/lib/liblmc_atomi.so
