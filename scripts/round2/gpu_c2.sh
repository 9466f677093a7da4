#!/bin/bash
out=gpurun_out/r02c; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_graph.py tests/test_gpu_abi2.py tests/test_gpu_rows.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -n 6 $out/pytest.log; echo "pytest rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
for g in 0 1; do
  LMC_GRAPH=$g timeout -k 10 200 python bench.py --size 256 --chains 128 --prior l2 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe > $out/c2_graph$g.json 2> $out/c2_graph$g.err
  python -c "import json;d=json.load(open('$out/c2_graph$g.json'));print('c2 graph=$g us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']], 'value', round(d['value']))"
done
LMC_GRAPH=1 timeout -k 10 200 python bench.py --size 256 --chains 128 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('256x128 TV10 graph=1 us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']])"
LMC_GRAPH=0 timeout -k 10 200 python bench.py --size 256 --chains 128 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('256x128 TV10 graph=0 us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']])"
