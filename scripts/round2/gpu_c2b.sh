#!/bin/bash
out=gpurun_out/r02c; mkdir -p $out
for g in 0 1; do
  LMC_BENCH_NO_TIMING=1 LMC_GRAPH=$g timeout -k 10 200 python bench.py --size 256 --chains 128 --prior l2 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe > $out/c2_graph$g.json 2> $out/c2_graph$g.err
  python -c "import json;d=json.load(open('$out/c2_graph$g.json'));print('c2 graph=$g us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']], 'value', round(d['value']))"
  LMC_BENCH_NO_TIMING=1 LMC_GRAPH=$g timeout -k 10 200 python bench.py --size 256 --chains 128 --prior l2 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe --no-moments | python -c "import json,sys;d=json.load(sys.stdin);print('c2 no-moments graph=$g us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']])"
  LMC_BENCH_NO_TIMING=1 LMC_GRAPH=$g timeout -k 10 200 python bench.py --size 256 --chains 128 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('256x128 TV10 graph=$g us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']])"
done
