#!/bin/bash
for n in "$@"; do
  LMC_ATOMI_LIB=build/exp_$n/liblmc.so LMC_VARIANT=split timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-moments | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', 'ms/launch', round(d['roofline']['launch_ms'],3))"
done
