#!/bin/bash
# ULPDA at 512 x 512 x 1024: Chebyshev launches over chain chunks (working set of a chunk inside the memory-side cache)
o=gpurun_out/r02cheb; mkdir -p $o
for ch in 0 16 32 64 128 256; do
  LMC_CHEB_CHUNK=$ch timeout -k 10 200 python bench.py --alg ulpda --steps 20 --warmup 5 --no-hbm-probe --no-cpu-baseline --repeats 1 > $o/c$ch.json 2> $o/c$ch.err || exit 1
  python -c "import json;d=json.load(open('$o/c$ch.json'));print('chunk',$ch,'ms/it %.3f'%d['ms_per_step'])"
done
