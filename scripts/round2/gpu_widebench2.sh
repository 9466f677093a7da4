#!/bin/bash
# aligned vs unaligned width at the same size (667 x 880 vs 667 x 877), pipe and rows
o=gpurun_out/r02wide; mkdir -p $o
run() { tag=$1; shift; timeout -k 10 300 python bench.py --no-hbm-probe --no-cpu-baseline --steps 30 --warmup 5 --repeats 1 "$@" > $o/$tag.json 2> $o/$tag.err || exit 1; python - $o/$tag.json $tag <<'PY'
import json, sys
d = json.load(open(sys.argv[1])); r = d["roofline"]
print(sys.argv[2], "ms/step %.3f" % d["ms_per_step"], "launch_ms", r.get("launch_ms"), r.get("kernel"), "frac %.3f" % (r.get("frac") or 0))
PY
}
run a_tv_880 --size 667 --width 880 --chains 512
run a_tv_877 --size 667 --width 877 --chains 512
run a_l2_880 --size 667 --width 880 --chains 512 --prior l2
run a_l2_877 --size 667 --width 877 --chains 512 --prior l2
