#!/bin/bash
set -e
mkdir -p gpurun_out/r02pair
timeout -k 10 600 python -m pytest tests/test_gpu_ulpda.py tests/test_gpu_ncvx.py tests/test_gpu_abi2.py -x -q > gpurun_out/r02pair/tests.log 2>&1 || { tail -30 gpurun_out/r02pair/tests.log; exit 1; }
tail -3 gpurun_out/r02pair/tests.log
for pair in 1 0; do
  LMC_CHEB_PAIR=$pair timeout -k 10 200 python bench.py --alg ulpda --steps 20 --warmup 5 --no-hbm-probe --no-cpu-baseline --repeats 1 > gpurun_out/r02pair/b$pair.json 2> gpurun_out/r02pair/b$pair.err
  python -c "import json;d=json.load(open('gpurun_out/r02pair/b$pair.json'));print('pair',$pair,'ms/it %.3f'%d['ms_per_step'])"
done
