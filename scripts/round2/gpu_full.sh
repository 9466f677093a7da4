#!/bin/bash
# full GPU suite, then the bench lines whose kernels changed (headline, rows, ME-TV, ULPDA), each only if the step before passed
set -e
bash scripts/round2/gpu_tests.sh r02full
mkdir -p gpurun_out/r02full
timeout -k 10 200 python bench.py > gpurun_out/r02full/bench.json 2> gpurun_out/r02full/bench.err && cat gpurun_out/r02full/bench.json
timeout -k 10 200 python bench.py --prior l2 --no-hbm-probe > gpurun_out/r02full/bench_l2.json 2> gpurun_out/r02full/bench_l2.err && cut -c1-400 gpurun_out/r02full/bench_l2.json
timeout -k 10 200 python bench.py --ncvx me --ncvx-iters 50 --steps 20 --warmup 5 --no-hbm-probe > gpurun_out/r02full/bench_me.json 2> gpurun_out/r02full/bench_me.err && cut -c1-400 gpurun_out/r02full/bench_me.json
timeout -k 10 200 python bench.py --alg ulpda --steps 20 --warmup 5 --no-hbm-probe > gpurun_out/r02full/bench_ulpda.json 2> gpurun_out/r02full/bench_ulpda.err && cut -c1-400 gpurun_out/r02full/bench_ulpda.json
