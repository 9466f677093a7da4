#!/bin/bash
out=gpurun_out/r02t2; mkdir -p $out
LMC_PIPE_TEAMS=1 timeout -k 10 600 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_abi2.py tests/test_gpu_fullsize.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -n 4 $out/pytest.log; echo "pytest rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
for tm in 0 1; do
LMC_PIPE_TEAMS=$tm timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe > $out/bench_t$tm.json 2> $out/bench_t$tm.err
python -c "import json;d=json.load(open('$out/bench_t$tm.json'));print('teams=$tm launch_ms', round(d['roofline']['launch_ms'],4), 'ms/step', [round(x,4) for x in d['ms_per_step_all']])"
done
