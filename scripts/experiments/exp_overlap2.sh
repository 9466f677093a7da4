#!/bin/bash
run() { python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', 'ms/step', round(d['ms_per_step'],4), 'launch', round(d['roofline']['launch_ms'],4), 'value', round(d['value']))"; }
LMC_MOMENTS_OVERLAP=0 run "overlap=0"
for p in 0 1; do for w in 0 128 256; do LMC_MOMENTS_OVERLAP=1 LMC_MOMENTS_SIDE_PRIO=$p LMC_MOMENTS_BG_WGS=$w run "overlap=1 lowprio=$p bg_wgs=$w"; done; done
