#!/bin/bash
# headline: the moment reduction under the next step kernel (side stream), number of background workgroups
o=gpurun_out/r02momov; mkdir -p $o
run() { tag=$1; shift; env "$@" timeout -k 10 200 python bench.py --no-hbm-probe --no-cpu-baseline --repeats 2 > $o/$tag.json 2> $o/$tag.err || exit 1
  python -c "import json;d=json.load(open('$o/$tag.json'));print('$tag','ms/it %.4f'%d['ms_per_step'],'launch_ms',d['roofline'].get('launch_ms'))"; }
for w in 256 384 512 768 1024 2048; do run ov$w LMC_MOMENTS_OVERLAP=1 LMC_MOMENTS_BG_WGS=$w; done
