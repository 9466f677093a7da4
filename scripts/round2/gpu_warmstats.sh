#!/bin/bash
mkdir -p gpurun_out/r02w
timeout -k 10 500 python scripts/round2/warm_stats.py > gpurun_out/r02w/warm_stats.json 2> gpurun_out/r02w/warm_stats.err; echo rc=$?
python -c "
import json; d=json.load(open('gpurun_out/r02w/warm_stats.json'))
print('MC noise', d['mc_noise_rel_l2 (cold K=10, two seeds)'])
for k,v in d['variants'].items(): print(k, {a: (round(b,6) if isinstance(b,float) else b) for a,b in v.items()})
"
