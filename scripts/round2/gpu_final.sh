#!/bin/bash
# last check of the round: smoke, the default bench line, the other BASELINE configurations
set -e
o=gpurun_out/r02final; mkdir -p $o
python -c "import __graft_entry__ as g; g.smoke()"
timeout -k 10 300 python bench.py > $o/bench.json 2> $o/bench.err; python -c "import json;d=json.load(open('$o/bench.json'));print('default', round(d['value']), d['ms_per_step'], d['roofline']['frac'], d['roofline']['launch_ms'])"
for c in 2 5; do
  timeout -k 10 300 python bench.py --config $c --no-hbm-probe > $o/c$c.json 2> $o/c$c.err; python -c "import json;d=json.load(open('$o/c$c.json'));r=d['roofline'];print('config $c', round(d['value']), d['ms_per_step'], r['kernel'], r['frac'], r.get('iterations_per_launch'))"
done
