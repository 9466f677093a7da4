#!/bin/bash
run() { v=$1; shift; LMC_VARIANT=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-moments "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $*', '-> ms/launch', round(d['roofline']['launch_ms'],3), 'GB/s', round(d['roofline']['achieved']))"; }
for v in split point; do run $v --prior l2; done
for v in split point; do run $v --prior l2 --size 256 --chains 128; done
run point --prior l1
run point --prior l2 --data identity
