#!/bin/bash
# MYMALA at the headline shape: acceptance rate and time per iteration against the step size
for ts in 1.0 0.3 0.1 0.03 0.01; do
  python bench.py --alg mymala --tau-scale $ts --steps 40 --warmup 100 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('tau_scale', d['config']['tau_scale'], 'acc', round(d['config']['acceptance_rate_mean'],4), 'ms/it', round(d['ms_per_step'],3))"
done
