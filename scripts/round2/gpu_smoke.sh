#!/bin/bash
python -c "import __graft_entry__ as g; g.smoke(); print('__SMOKE_OK__')" 2>&1 | tail -3
python bench.py --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['frac'], d['roofline']['limiter'], d['cpu_baseline']['value'])"
python bench.py --config 2 --steps 200 --warmup 20 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config2', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['cpu_baseline']['value'])"
python bench.py --config 5 --steps 20 --warmup 5 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5', d['value'], d['ms_per_step'], d['roofline']['kernel'], d['roofline']['frac'], d['cpu_baseline']['value'])"
