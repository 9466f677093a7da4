#!/bin/bash
# wide / unaligned widths on the pipe kernel: new tests, then the suites that share the kernel, then the two bench lines to compare with r02 numbers
set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_wide.py -x -q > gpurun_out/wide_tests.log 2>&1 || { tail -40 gpurun_out/wide_tests.log; exit 1; }
tail -3 gpurun_out/wide_tests.log
timeout -k 10 400 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_ncvx.py tests/test_gpu_abi2.py tests/test_gpu_parity.py tests/test_gpu_rows.py tests/test_gpu_ulpda.py -x -q > gpurun_out/wide_tests2.log 2>&1 || { tail -40 gpurun_out/wide_tests2.log; exit 1; }
tail -3 gpurun_out/wide_tests2.log
timeout -k 10 200 python bench.py > gpurun_out/wide_bench.json 2> gpurun_out/wide_bench.err && cat gpurun_out/wide_bench.json
timeout -k 10 200 python bench.py --ncvx me --ncvx-iters 50 --steps 20 --warmup 5 --no-hbm-probe > gpurun_out/wide_bench_me.json 2> gpurun_out/wide_bench_me.err && cat gpurun_out/wide_bench_me.json
timeout -k 10 200 python bench.py --prior l2 --no-hbm-probe > gpurun_out/wide_bench_l2.json 2> gpurun_out/wide_bench_l2.err && cat gpurun_out/wide_bench_l2.json
