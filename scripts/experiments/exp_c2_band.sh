#!/bin/bash
# BASELINE config 2 (256x256, 128 chains, blur + l2 prior): rows-kernel band height sweep
for b in 0 8 16 24 32 64 128; do
  LMC_ROWS_BAND=$b python bench.py --size 256 --chains 128 --prior l2 --steps 400 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('band', $b, 'launch_us', round(d['roofline']['launch_ms']*1e3,2), 'ms/step', round(d['ms_per_step'],4), 'frac', round(d['roofline']['frac'],3), d['roofline']['kernel'])"
done
