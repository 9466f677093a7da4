#!/bin/bash
# kernel trace of the ULPDA bench (no counters)
o=gpurun_out/r02pair_trace; mkdir -p $o; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $o/trace --output-format csv -- python3 bench.py --alg ulpda --steps 20 --warmup 5 --repeats 1 --no-cpu-baseline --no-hbm-probe > $o/bench.json 2> $o/trace.log || exit 1
f=$(find $o/trace -name '*kernel_stats.csv' | head -1); cp $f $o/kernel_stats.csv; rm -rf $o/trace
head -12 $o/kernel_stats.csv | cut -c1-160
