#!/bin/bash
# A/B of one translation unit: exp_obj.sh <tag> <unit> "<bench args>" [hipcc flags...] -- builds lmc_atomi_amd/csrc/<unit>.hip with the flags
# given (base flags without -fno-slp-vectorize unless passed), links a scratch library, runs the bench line.
tag=$1; unit=$2; bargs=$3; shift 3
out=gpurun_out/exp_$tag; mkdir -p $out
d=/tmp/exp_$tag; mkdir -p $d
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Iinclude -Ilmc_atomi_amd/csrc -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c lmc_atomi_amd/csrc/$unit.hip -o $d/u.o || exit 1
objs=$(ls build/obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblmc_atomi.so $objs $d/u.o -ldl || exit 1
export LMC_ATOMI_LIB=$d/liblmc_atomi.so
timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe $bargs > $out/bench.json 2> $out/bench.err
python -c "import json;d=json.load(open('$out/bench.json'));print('$tag', d['roofline']['kernel'], 'launch_ms', round(d['roofline']['launch_ms'],4), 'ms/step', [round(x,4) for x in d['ms_per_step_all']])"
