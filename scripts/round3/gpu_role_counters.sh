#!/bin/bash
# Per-role counters of the headline kernel: the free-running single-role / role-pair builds of gpu_roles.sh under rocprofv3 --pmc (three passes each).
set -u
out=$PWD/gpurun_out/r3_rolepmc; mkdir -p $out
export TMPDIR=/tmp LMC_BENCH_AS_CONFIGURED=0 LMC_MOMENTS_OVERLAP=0
args="--steps 6 --warmup 2 --repeats 1 --no-cpu-baseline --no-hbm-probe"
for v in base NB L C N T1 T2 T5 P04 P15 P26 P37; do
  if [ $v = base ]; then unset LMC_ATOMI_LIB; else export LMC_ATOMI_LIB=$PWD/build/var/liblmc_atomi_$v.so; fi
  i=0
  for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
              "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
              "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    rocprofv3 --pmc $pass -d $out/${v}_p$i --output-format csv -- python3 bench.py $args > /dev/null 2> $out/${v}_p$i.log || { echo "fail $v $i"; exit 1; }
  done
  echo "$v done"
done
python3 - $out <<'PY' | tee $out/summary.txt
import csv, glob, sys, collections
out = sys.argv[1]
names = ['GRBM_GUI_ACTIVE','SQ_WAVE_CYCLES','SQ_BUSY_CYCLES','SQ_INSTS_VALU','SQ_INSTS_SALU','SQ_INSTS_LDS','SQ_ACTIVE_INST_VALU','SQ_ACTIVE_INST_LDS','SQ_ACTIVE_INST_ANY','SQ_WAIT_ANY','SQ_WAIT_INST_ANY','SQ_WAIT_INST_LDS','SQ_INST_CYCLES_SALU','SQ_LDS_BANK_CONFLICT']
print('variant   ' + ' '.join(f'{n[-14:]:>14s}' for n in names))
for v in "base NB L C N T1 T2 T5 P04 P15 P26 P37".split():
    agg = collections.defaultdict(list)
    for f in glob.glob(f'{out}/{v}_p*/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if 'myula_step_pipe_kernel' in r.get('Kernel_Name', ''):
                agg[r['Counter_Name']].append(float(r['Counter_Value']))
    m = {c: sum(x) / len(x) for c, x in agg.items()}
    print(f'{v:9s} ' + ' '.join(f'{m.get(n, float("nan")):14.5g}' for n in names))
PY
