#!/bin/bash
o=gpurun_out/r02dr; mkdir -p $o
for m in 1 0 1 0; do
  LMC_ULPDA_DUAL_RHS=$m timeout -k 10 200 python bench.py --alg ulpda --steps 20 --warmup 5 --no-hbm-probe --no-cpu-baseline --repeats 1 > $o/b$m.json 2> $o/b$m.err || exit 1
  python -c "import json;d=json.load(open('$o/b$m.json'));print('dual_rhs',$m,'ms/it %.3f'%d['ms_per_step'])"
done
