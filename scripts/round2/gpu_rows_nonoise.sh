#!/bin/bash
o=gpurun_out/r02rowsx; mkdir -p $o
for nz in philox none; do
  timeout -k 10 200 python bench.py --prior l2 --noise $nz --no-hbm-probe --no-cpu-baseline --repeats 1 --no-moments > $o/$nz.json 2> $o/$nz.err || exit 1
  python -c "import json;d=json.load(open('$o/$nz.json'));print('$nz','ms/it %.4f'%d['ms_per_step'],'launch_ms',d['roofline'].get('launch_ms'))"
done
