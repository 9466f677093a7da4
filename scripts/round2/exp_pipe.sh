#!/bin/bash
# A/B of the headline kernel: builds lmc_step_pipe.o with the given -D flags into a scratch library, runs the pipe parity tests and the
# bench (usage on the GPU box: exp_pipe.sh <tag> [-DFLAG ...]).  The committed library is untouched.
tag=$1; shift
out=gpurun_out/exp_$tag; mkdir -p $out
d=/tmp/exp_$tag; mkdir -p $d
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Iinclude -Ilmc_atomi_amd/csrc -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c lmc_atomi_amd/csrc/lmc_step_pipe.hip -o $d/pipe.o || exit 1
objs=$(ls build/obj/*.o | grep -v lmc_step_pipe.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblmc_atomi.so $objs $d/pipe.o -ldl || exit 1
export LMC_ATOMI_LIB=$d/liblmc_atomi.so
timeout -k 10 300 python -m pytest tests/test_gpu_pipe.py -m gpu -q -x > $out/pytest.log 2>&1; echo "$tag pytest rc=$? $(tail -1 $out/pytest.log)"
timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe > $out/bench.json 2> $out/bench.err
python -c "import json;d=json.load(open('$out/bench.json'));print('$tag', 'launch_ms', round(d['roofline']['launch_ms'],4), 'ms/step', [round(x,4) for x in d['ms_per_step_all']])"
