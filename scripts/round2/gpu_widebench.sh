#!/bin/bash
# einstein-size (667 x 877, prox_lmc_deconv.py:44-46) and 1024 x 1024 images: fast kernels against the previous fallbacks
set -e
o=gpurun_out/r02wide; mkdir -p $o
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-hbm-probe --no-cpu-baseline --steps 30 --warmup 5 "$@" > $o/$tag.json 2> $o/$tag.err; python - $o/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "launch_ms", r.get("launch_ms"), r.get("kernel"), "frac %.3f" % (r.get("frac") or 0))
PY
}
run e_tv_pipe --size 667 --width 877 --chains 512
LMC_VARIANT=tile run e_tv_tile --size 667 --width 877 --chains 512
run e_l2_rows --size 667 --width 877 --chains 512 --prior l2
LMC_VARIANT=point run e_l2_point --size 667 --width 877 --chains 512 --prior l2
run k_tv_pipe --size 1024 --chains 256
LMC_VARIANT=tile run k_tv_tile --size 1024 --chains 256
run k_l2_rows --size 1024 --chains 256 --prior l2
LMC_VARIANT=point run k_l2_point --size 1024 --chains 256 --prior l2
run e_ulpda --size 667 --width 877 --chains 256 --alg ulpda --steps 10 --warmup 3
