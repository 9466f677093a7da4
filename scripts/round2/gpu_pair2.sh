#!/bin/bash
set -e
mkdir -p gpurun_out/r02pair
timeout -k 10 600 python -m pytest tests/test_gpu_ulpda.py tests/test_gpu_ncvx.py tests/test_gpu_wide.py tests/test_gpu_r4.py -x -q > gpurun_out/r02pair/tests.log 2>&1 || { tail -30 gpurun_out/r02pair/tests.log; exit 1; }
tail -1 gpurun_out/r02pair/tests.log
for band in 0; do
  LMC_PAIR_BAND=$band timeout -k 10 200 python bench.py --alg ulpda --steps 20 --warmup 5 --no-hbm-probe --no-cpu-baseline --repeats 1 > gpurun_out/r02pair/bb$band.json 2> gpurun_out/r02pair/bb$band.err
  python -c "import json;d=json.load(open('gpurun_out/r02pair/bb$band.json'));print('band',$band,'ms/it %.3f'%d['ms_per_step'])"
done
