#!/bin/bash
# Profiles bench.py on the GPU box: kernel-trace stats (60 timed steps, the same protocol as the bench line) + PMC passes in
# their own runs (never combined with a trace: gpurun refuses that).  Outputs under gpurun_out/prof_<tag>/:
#   kernel_stats.csv  summary.txt  counters.json (per-kernel mean per dispatch; HBM traffic corrected per MI355X_MICROARCH.md)
# usage: scripts/profile.sh <tag> [bench args...]
set -u
tag=${1:-r2}; shift || true
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# the step kernel ALONE, as bench.py's event-timed region measures it (roofline.launch_ms): moment reductions in line, not on the side stream
# (bench.py's `value` is measured with them on the side stream: the default)
export LMC_MOMENTS_OVERLAP=${LMC_MOMENTS_OVERLAP:-0}
args="--steps 10 --warmup 2 --repeats 1 --no-cpu-baseline --no-hbm-probe $*"
targs="--steps 60 --warmup 10 --repeats 1 --no-cpu-baseline --no-hbm-probe $*"   # long enough for clocks to settle: the average must agree with bench.py
cd $root
rocprofv3 --kernel-trace --stats -d $out/trace --output-format csv -- python3 bench.py $targs > $out/bench_trace.json 2> $out/trace.log
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
            "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_LDS_BANK_CONFLICT" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" "WRITE_SIZE GRBM_COUNT"; do
  name=$(echo $pass | tr ' ' '_' | cut -c1-40)
  rocprofv3 --pmc $pass -d $out/pmc_$name --output-format csv -- python3 bench.py $args > /dev/null 2> $out/pmc_$name.log
done
python3 - $out "$tag" "$*" <<'PY' | tee $out/summary.txt
import csv, glob, sys, os, collections, json
out, tag, bargs = sys.argv[1], sys.argv[2], sys.argv[3]
stats = {}
for f in glob.glob(out + '/trace/**/*kernel_stats.csv', recursive=True):
    print('== kernel stats (rocprofv3 --kernel-trace --stats, bench.py --steps 60 --warmup 10', bargs, ')')
    rows = list(csv.reader(open(f)))
    open(out + '/kernel_stats.csv', 'w').write(open(f).read())
    for row in rows[:9]: print(','.join(row)[:230])
    for row in rows[1:]:
        stats[row[0].split('(')[0].replace('void ', '')] = {'calls': int(row[1]), 'avg_ns': float(row[3])}
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + '/pmc_*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r.get('Kernel_Name', '')
        if 'lmc::' not in k: continue
        agg[k.split('(')[0].replace('void ', '')][r['Counter_Name']].append(float(r['Counter_Value']))
res = {}
for k, d in agg.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    print('== pmc (mean per dispatch)', k)
    for c, v in sorted(m.items()): print(f'   {c:28s} {v:.6g}   (n={len(d[c])})')
    ent = {'counters': m}
    if k in stats: ent.update(stats[k])
    if 'FETCH_SIZE' in m and 'WRITE_SIZE' in m:     # KiB; FETCH_SIZE reports half the bytes of 16-B streaming reads on gfx950
        ent['traffic_bytes_per_launch'] = int((2 * m['FETCH_SIZE'] + m['WRITE_SIZE']) * 1024)
    if 'SQ_ACTIVE_INST_VALU' in m and 'GRBM_GUI_ACTIVE' in m:
        cyc = m['GRBM_GUI_ACTIVE'] / 8.0                     # kernel cycles (the counter sums the 8 XCDs)
        ent['valu'] = {'busy_frac': 4.0 * m['SQ_ACTIVE_INST_VALU'] / (1024.0 * cyc),   # quad-cycles x 4 / (SIMDs x kernel cycles)
                       'active_inst_valu_quadcycles': m['SQ_ACTIVE_INST_VALU'], 'insts_valu': m.get('SQ_INSTS_VALU'),
                       'per_wave_frac': m['SQ_ACTIVE_INST_VALU'] / m['SQ_WAVE_CYCLES'] if m.get('SQ_WAVE_CYCLES') else None,
                       'wait_any_frac': m.get('SQ_WAIT_ANY', 0) / m['SQ_WAVE_CYCLES'] if m.get('SQ_WAVE_CYCLES') else None,
                       'kernel_cycles': cyc}
        print(f"   -> VALU busy {ent['valu']['busy_frac']:.3f} of the SIMD-cycles of the launch; per wave {ent['valu']['per_wave_frac']:.3f}")
    res[k] = ent
json.dump({'tag': tag, 'bench_args': bargs, 'kernels': res}, open(out + '/counters.json', 'w'), indent=1)
PY
