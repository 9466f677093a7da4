#!/bin/bash
run() { v=$1; shift; LMC_VARIANT=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-moments "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $*', '-> ms/launch', round(d['roofline']['launch_ms'],3), 'GB/s', round(d['roofline']['achieved']))"; }
for v in tile stream split; do run $v --prior l2; done
for v in tile stream split; do run $v --prior l2 --size 256 --chains 128; done
run split --tv-iters 10 --size 256 --chains 4096
