#!/bin/bash
# usage: scripts/kres.sh <file.hip> [extra hipcc flags]  -> per-kernel VGPR / SGPR / LDS / scratch of every kernel in the translation unit
# (compiles with -save-temps under /tmp/kres/<name>; the .s is kept there for instruction counts)
src=$1; shift
name=$(basename $src .hip)
d=/tmp/kres/$name; mkdir -p $d; cd $d
flags="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -I/root/repo/include -I/root/repo/lmc_atomi_amd/csrc"
case $name in lmc_step_pipe*|lmc_step_block) ;; *) flags="$flags -fno-slp-vectorize";; esac
/opt/rocm/bin/hipcc $flags "$@" -c /root/repo/lmc_atomi_amd/csrc/$name.hip -o s.o -save-temps 2>/dev/null
python3 - <<'PY'
import re, glob, subprocess
s = open(glob.glob('*gfx950.s')[0]).read()
md = s[s.rindex('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    g = lambda k: re.search(r'\.%s:\s+(\d+)' % k, blk)
    name = re.search(r'\.name:\s+(\S+)', blk).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dem = re.sub(r'\(.*', '', dem).replace('lmc::', '').replace('void ', '')
    print(f"{dem:60s} vgpr {g('vgpr_count').group(1):>4s} sgpr {g('sgpr_count').group(1):>4s} lds {g('group_segment_fixed_size').group(1):>6s} "
          f"scratch {g('private_segment_fixed_size').group(1):>5s} spill {g('vgpr_spill_count').group(1) if g('vgpr_spill_count') else '-':>3s}")
PY
