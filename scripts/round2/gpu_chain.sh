#!/bin/bash
# chained / warm pipe kernels after the raw state loads: parity suites, then ME-TV, TV(niter=50) and warm-dual K = 1, 2, 3
set -e
o=gpurun_out/r02chain; mkdir -p $o
timeout -k 10 600 python -m pytest tests/test_gpu_pipe.py tests/test_gpu_ncvx.py tests/test_gpu_abi2.py tests/test_gpu_wide.py tests/test_gpu_haar.py -x -q > $o/tests.log 2>&1 || { tail -30 $o/tests.log; exit 1; }
tail -1 $o/tests.log
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-hbm-probe --no-cpu-baseline --steps 30 --warmup 5 --repeats 1 "$@" > $o/$tag.json 2> $o/$tag.err || exit 1; python - $o/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "launch_ms", r.get("launch_ms"))
PY
}
run metv --ncvx me --ncvx-iters 50 --steps 15
run tv50 --tv-iters 50 --steps 15
run warm1 --tv-warm --tv-iters 1
run warm2 --tv-warm --tv-iters 2
run warm3 --tv-warm --tv-iters 3
