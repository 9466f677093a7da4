#!/bin/bash
timeout -k 10 300 python -m pytest tests/test_gpu_overlap.py tests/test_gpu_graph.py -m gpu -q -x 2>&1 | tail -2
for w in 64 128 256 512; do
  LMC_MOMENTS_BG_WGS=$w timeout -k 10 200 python bench.py --size 256 --chains 128 --prior l2 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('c2 bg_wgs=$w us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']], round(d['value']))"
done
