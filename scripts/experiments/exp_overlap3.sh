#!/bin/bash
# overlap + wave priorities of the pipe kernel shifted up by one (background reduction waves stay at 0)
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'ms/step', round(d['ms_per_step'],4), 'launch', round(d['roofline']['launch_ms'],4), 'value', round(d['value']))"; }
for lib in "" build/exp_prioA/liblmc.so; do
  for o in 0 1; do LMC_ATOMI_LIB=${lib:-lmc_atomi_amd/lib/liblmc_atomi.so} LMC_MOMENTS_OVERLAP=$o LMC_MOMENTS_BG_WGS=256 run "lib=${lib:-default} overlap=$o"; done
done
