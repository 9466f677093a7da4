#!/bin/bash
out=gpurun_out/r02u; mkdir -p $out
timeout -k 10 900 python -m pytest tests/test_gpu_ulpda.py tests/test_gpu_r4.py::test_r4_ulpda_64_posterior_mean tests/test_gpu_ncvx.py -m gpu -q -x > $out/pytest.log 2>&1; rc=$?
tail -n 5 $out/pytest.log; echo "pytest rc=$rc"; if [ $rc -ge 124 ]; then exit $rc; fi
for f in 1 0; do
LMC_ULPDA_FUSE=$f timeout -k 10 300 python bench.py --alg ulpda --steps 30 --warmup 5 --no-cpu-baseline --no-hbm-probe > $out/bench_fuse$f.json 2> $out/bench_fuse$f.err
python -c "import json;d=json.load(open('$out/bench_fuse$f.json'));print('ulpda fuse=$f ms/iter', [round(x,3) for x in d['ms_per_step_all']])"
done
