#!/bin/bash
# kernel trace of one bench configuration: gpu_trace.sh <tag> [bench args]
tag=$1; shift
out=$(pwd)/gpurun_out/trace_$tag; mkdir -p $out; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $out/t --output-format csv -- python3 bench.py --steps 30 --warmup 5 --repeats 1 --no-cpu-baseline --no-hbm-probe "$@" > $out/bench.json 2> $out/log
f=$(find $out/t -name "*kernel_stats.csv" | head -1); cut -c1-150 $f | head -12
