#!/bin/bash
run() {
  lib=$1; shift
  LMC_ATOMI_LIB=$lib timeout -k 10 120 python bench.py "$@" --steps 50 --warmup 5 --no-cpu-baseline --no-moments 2>/dev/null > gpurun_out/_b.json
  python - "$lib $*" <<'PY'
import sys, json
j = json.loads(open("gpurun_out/_b.json").read().strip().splitlines()[-1])
r = j["roofline"]
print(sys.argv[1], "|", r["kernel"], "launch_ms=%.4f" % r["launch_ms"])
PY
}
run lmc_atomi_amd/lib/liblmc_atomi.so --prior l2
run build/exp_rslp/liblmc.so --prior l2
run lmc_atomi_amd/lib/liblmc_atomi.so --prior l2 --size 256 --chains 128
run build/exp_rslp/liblmc.so --prior l2 --size 256 --chains 128
run lmc_atomi_amd/lib/liblmc_atomi.so --prior haar --data mask
run build/exp_bslp/liblmc.so --prior haar --data mask
