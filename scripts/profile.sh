#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats + PMC passes.  Outputs under gpurun_out/prof_<tag>/.
# usage: scripts/profile.sh <tag> [bench args...]
set -u
tag=${1:-r1}; shift || true
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
args="--steps 10 --warmup 2 --no-cpu-baseline $*"
cd $root
targs="--steps 60 --warmup 10 --no-cpu-baseline $*"   # long enough for clocks to settle: the average must agree with bench.py
rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- python3 bench.py $targs > $out/bench_trace.json 2> $out/trace.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass -d $out/pmc_$name --output-format csv -- python3 bench.py $args > /dev/null 2> $out/pmc_$name.log
done
python3 - $out <<'PY'
import csv, glob, sys, os, collections
out = sys.argv[1]
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats', f)
    for row in list(csv.reader(open(f)))[:8]: print(','.join(row)[:240])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'myula_step' not in k: continue
        agg[k.split('(')[0][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in agg.items():
    print('== pmc (mean per dispatch)', k)
    for c, v in sorted(d.items()): print(f'   {c:28s} {sum(v)/len(v):.6g}   (n={len(v)})')
PY
