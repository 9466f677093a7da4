#!/bin/bash
# round 2, first GPU pass: the whole GPU suite, the headline bench line, the warm-dual variants.  Output under gpurun_out/r02a/.
out=gpurun_out/r02a; mkdir -p $out
timeout -k 10 1000 python -m pytest tests -m gpu -q -x --durations=15 > $out/pytest.log 2>&1; rc=$?
tail -n 25 $out/pytest.log
echo "pytest rc=$rc"
if [ $rc -ge 124 ]; then exit $rc; fi
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err || { echo bench failed; tail -5 $out/bench.err; exit 1; }
cat $out/bench.json
for k in 1 2 3; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --tv-warm --tv-iters $k --no-cpu-baseline > $out/bench_warm$k.json 2> $out/bench_warm$k.err || { echo warm $k failed; tail -5 $out/bench_warm$k.err; exit 1; }
  python -c "import json;d=json.load(open('$out/bench_warm$k.json'));print('warm K=$k', d['ms_per_step'], d['roofline']['launch_ms'], d['roofline'].get('actual_gbs'))"
done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --tv-lagged --no-cpu-baseline > $out/bench_lag.json 2> $out/bench_lag.err && python -c "import json;d=json.load(open('$out/bench_lag.json'));print('lagged K=10->9', d['ms_per_step'], d['roofline']['launch_ms'])"
