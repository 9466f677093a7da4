#!/bin/bash
# A/B of the pipe kernel built from another source directory: exp_src.sh <tag> <srcdir> [-D flags]; with PROFILE=1 also collects the SQ counters
tag=$1; src=$2; shift 2
out=$(pwd)/gpurun_out/exp_$tag; mkdir -p $out
d=/tmp/exp_$tag; mkdir -p $d
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -Iinclude -I$src -Ilmc_atomi_amd/csrc -Wno-unused-function"
/opt/rocm/bin/hipcc $F "$@" -c $src/lmc_step_pipe.hip -o $d/pipe.o || exit 1
objs=$(ls build/obj/*.o | grep -v lmc_step_pipe.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $d/liblmc_atomi.so $objs $d/pipe.o -ldl || exit 1
export LMC_ATOMI_LIB=$d/liblmc_atomi.so
timeout -k 10 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-hbm-probe > $out/bench.json 2> $out/bench.err
python -c "import json;d=json.load(open('$out/bench.json'));print('$tag', 'launch_ms', round(d['roofline']['launch_ms'],4))"
if [ "${PROFILE:-0}" = "1" ]; then
  export TMPDIR=/tmp
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT"; do
    name=$(echo $pass | tr ' ' '_' | cut -c1-30)
    rocprofv3 --pmc $pass -d $out/pmc_$name --output-format csv -- python3 bench.py --steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-hbm-probe > /dev/null 2> $out/pmc_$name.log
  done
  python3 - $out <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'myula_step_pipe' in r.get('Kernel_Name', ''): agg[r['Counter_Name']].append(float(r['Counter_Value']))
for c, v in sorted(agg.items()): print(f'   {c:24s} {sum(v)/len(v):.5g}')
PY
fi
