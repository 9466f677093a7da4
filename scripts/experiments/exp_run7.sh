#!/bin/bash
run() { v=$1; shift; LMC_VARIANT=$v timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-moments "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v $*', '-> ms/launch', round(d['roofline']['launch_ms'],3), 'GB/s', round(d['roofline']['achieved']))"; }
run stream --prior l2 --size 512 --width 64 --chains 8192
run split --prior l2 --size 512 --width 64 --chains 8192
run stream --prior l2 --size 512 --width 128 --chains 4096
run stream --prior l2 --size 128 --width 64 --chains 32768
