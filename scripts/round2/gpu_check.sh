#!/bin/bash
out=gpurun_out/r02e; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q > $out/pytest.log 2>&1; rc=$?
tail -n 3 $out/pytest.log; echo "pytest rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --alg ulpda --steps 30 --warmup 5 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('ulpda ms/iter', [round(x,3) for x in d['ms_per_step_all']])"
LMC_ROWS_UNI=0 timeout -k 10 300 python bench.py --alg ulpda --steps 30 --warmup 5 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('ulpda uni=0 ms/iter', [round(x,3) for x in d['ms_per_step_all']])"
timeout -k 10 300 python bench.py --alg mymala --steps 30 --warmup 5 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('mymala ms/iter', [round(x,3) for x in d['ms_per_step_all']])"
