#!/bin/bash
# usage: scripts/kstat.sh <tag> [extra hipcc flags...]   -> compiles lmc_step_stream.hip (K=10 only) in /tmp/asm/<tag>, prints metadata
tag=$1; shift
d=/tmp/asm/$tag; mkdir -p $d; cd $d
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -DLMC_ONLY_K10 "$@" -I/root/repo/include -I/root/repo/lmc_atomi_amd/csrc -c /root/repo/lmc_atomi_amd/csrc/lmc_step_stream.hip -o s.o -save-temps 2>/dev/null
python3 - "$tag" <<'PY'
import re,sys
from collections import Counter
s=open('lmc_step_stream-hip-amdgcn-amd-amdhsa-gfx950.s').read()
md=s[s.rindex('amdhsa.kernels:'):]
for blk in md.split('  - .agpr_count:')[1:]:
    name=re.search(r'\.name:\s+(\S+)',blk).group(1)
    m=re.match(r'_ZN3lmc24myula_step_stream_kernelILi(\d+)ELi(\d+)ELi(\d+)E',name)
    if not m: continue
    K,NW,KT=map(int,m.groups())
    if KT!=5 or NW not in (2,8): continue
    g=lambda k: re.search(r'\.%s:\s+(\d+)'%k,blk).group(1)
    mm=re.search(r'^%s:(.*?)\.Lfunc_end\d+:'%re.escape(name), s, re.S|re.M)
    ins=[l.strip() for l in mm.group(1).split('\n') if l.strip() and not l.strip().startswith((';','.'))]
    c=Counter(x.split()[0] for x in ins)
    valu=sum(v for k,v in c.items() if k.startswith('v_')); salu=sum(v for k,v in c.items() if k.startswith('s_') and k!='s_waitcnt'); lds=sum(v for k,v in c.items() if k.startswith('ds_'))
    print(f"[{sys.argv[1]}] K={K} NW={NW} vgpr={g('vgpr_count')} agpr={blk.split()[0]} sgpr_spill={g('sgpr_spill_count')} vgpr_spill={g('vgpr_spill_count')} scratch={g('private_segment_fixed_size')} | instr={len(ins)} valu={valu} salu={salu} lds={lds} wait={c.get('s_waitcnt',0)} scratch_ops={sum(v for k,v in c.items() if k.startswith('scratch'))} readlane={c.get('v_readlane_b32',0)}")
PY
