#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_wide.py tests/test_gpu_rows.py tests/test_gpu_pipe.py tests/test_gpu_ncvx.py tests/test_gpu_haar.py -x -q > gpurun_out/wide_tests.log 2>&1 || { tail -40 gpurun_out/wide_tests.log; exit 1; }
tail -2 gpurun_out/wide_tests.log
bash scripts/round2/gpu_widebench.sh
