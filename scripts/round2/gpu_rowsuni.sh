#!/bin/bash
out=gpurun_out/r02v; mkdir -p $out
timeout -k 10 600 python -m pytest tests/test_gpu_rows.py tests/test_gpu_parity.py tests/test_gpu_ulpda.py tests/test_gpu_graph.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -n 4 $out/pytest.log; echo "pytest rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
for u in 0 1; do
LMC_ROWS_UNI=$u timeout -k 10 200 python bench.py --prior l2 --steps 60 --warmup 10 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('rows 512x1024 uni=$u launch_ms', round(d['roofline']['launch_ms'],4), [round(x,4) for x in d['ms_per_step_all']])"
LMC_ROWS_UNI=$u timeout -k 10 200 python bench.py --prior l2 --size 256 --chains 128 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe | python -c "import json,sys;d=json.load(sys.stdin);print('c2 uni=$u us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']], round(d['value']))"
LMC_ROWS_UNI=$u timeout -k 10 200 python bench.py --prior l2 --size 256 --chains 128 --steps 400 --warmup 40 --no-cpu-baseline --no-hbm-probe --no-moments | python -c "import json,sys;d=json.load(sys.stdin);print('c2 no-moments uni=$u us/iter', [round(1e3*x,2) for x in d['ms_per_step_all']])"
done
