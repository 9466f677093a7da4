#!/bin/bash
for w in 0 64 128 512; do
  LMC_GRAPH=0 LMC_BENCH_NO_TIMING=1 LMC_MOMENTS_OVERLAP=1 LMC_MOMENTS_BG_WGS=$w timeout -k 10 200 python bench.py --size 256 --chains 128 --prior l2 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('c2 overlap bg_wgs=$w us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']])"
done
