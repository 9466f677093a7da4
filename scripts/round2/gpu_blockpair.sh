#!/bin/bash
# mask + Haar-l1 (BASELINE config 5 prior) at 512 x 512 x 1024: two iterations per launch on the block kernel against single launches
o=gpurun_out/r02bp; mkdir -p $o
for mode in 1; do
  for mom in "" "--no-moments"; do
    t=m${mode}$(echo $mom | tr -d ' -')
    LMC_BLOCK_PAIR=$mode timeout -k 10 200 python bench.py --prior haar --data mask --no-hbm-probe --no-cpu-baseline --repeats 2 $mom > $o/$t.json 2> $o/$t.err || exit 1
    python -c "import json;d=json.load(open('$o/$t.json'));r=d['roofline'];print('pair=$mode','$mom','ms/it %.4f'%d['ms_per_step'],'launch_ms',r.get('launch_ms'),'frac %.3f'%r['frac'], r.get('kernel'), r.get('iterations_per_launch'))"
  done
done
