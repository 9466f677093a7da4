#!/bin/bash
run() { LMC_VARIANT=split timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-moments "$@" | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$*', '-> ms/launch', round(d['roofline']['launch_ms'],3))"; }
run --prior l2
run --tv-iters 2
run --tv-iters 5
run --tv-iters 10
