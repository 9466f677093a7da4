#!/bin/bash
# MYULA blur + l2 prior at 512 x 512 x 1024: two iterations per launch against single launches (with and without the moment reductions)
o=gpurun_out/r02rp; mkdir -p $o
run() { tag=$1; shift; env "$@" > /dev/null 2>&1; }
for mode in 1 0; do
  for mom in "" "--no-moments"; do
    t=m${mode}$(echo $mom | tr -d ' -')
    LMC_ROWS_PAIR=$mode timeout -k 10 200 python bench.py --prior l2 --no-hbm-probe --no-cpu-baseline --repeats 2 $mom > $o/$t.json 2> $o/$t.err || exit 1
    python -c "import json;d=json.load(open('$o/$t.json'));r=d['roofline'];print('pair=$mode','$mom','ms/it %.4f'%d['ms_per_step'],'launch_ms',r.get('launch_ms'),'frac %.3f'%r['frac'], r.get('kernel'), r.get('iterations_per_launch'))"
  done
done
