#!/bin/bash
run() { lib=$1; shift; LMC_ATOMI_LIB=$lib LMC_VARIANT=point timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-moments "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib $*', '-> ms/launch', round(d['roofline']['launch_ms'],3), 'GB/s', round(d['roofline']['achieved']))"; }
for lib in lmc_atomi_amd/lib/liblmc_atomi.so build/exp_pt128/liblmc.so build/exp_pt256/liblmc.so; do
  run $lib --prior l2 --data identity
  run $lib --prior l2 --data identity --noise none
  run $lib --prior l2
done
