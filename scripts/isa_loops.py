#!/usr/bin/env python3
"""Instruction mix of the loops of one kernel in a gfx950 .s file (hipcc -save-temps): for every backward branch target .. branch range,
counts by class (valu packed / plain / trans / dpp-mov, salu, lds, vmem, waitcnt, barrier, branch).  usage: isa_loops.py file.s 'kernel substring'"""
import re, sys, subprocess
from collections import Counter
path, want = sys.argv[1], sys.argv[2]
s = open(path).read()
names = re.findall(r'^(_Z\S+):\s*; @', s, re.M)
sel = None
for n in names:
    d = subprocess.run(['c++filt', n], capture_output=True, text=True).stdout
    if want in d:
        sel = n; break
assert sel, names[:5]
body = re.search(r'^%s:(.*?)^\.Lfunc_end\d+:' % re.escape(sel), s, re.S | re.M).group(1)
lines = [l.strip() for l in body.split('\n')]
ins = []      # (index, label or None, text)
labels = {}
for l in lines:
    if not l or l.startswith(';'): continue
    m = re.match(r'^(\.LBB\d+_\d+):', l)
    if m: labels[m.group(1)] = len(ins); continue
    if l.startswith('.'): continue
    ins.append(l.split(';')[0].strip())
TRANS = ('v_rsq', 'v_rcp', 'v_sqrt', 'v_log', 'v_exp', 'v_sin', 'v_cos')
def cls(t):
    op = t.split()[0]
    if op.startswith('v_pk_'): return 'valu_pk'
    if op.startswith(TRANS): return 'valu_trans'
    if op.startswith('v_') and ('dpp' in t or 'row_' in t or 'wave_' in t): return 'valu_dpp'
    if op.startswith('v_mad_u64') or op.startswith('v_mul_lo') or op.startswith('v_mul_hi'): return 'valu_int64'
    if op.startswith('v_mov') or op.startswith('v_accvgpr'): return 'valu_mov'
    if op.startswith('v_'): return 'valu_plain'
    if op.startswith('ds_'): return 'lds'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'vmem'
    if op == 's_waitcnt': return 'waitcnt'
    if op == 's_barrier': return 'barrier'
    if op.startswith(('s_cbranch', 's_branch')): return 'branch'
    if op.startswith('s_'): return 'salu'
    return 'other'
loops = []
for i, t in enumerate(ins):
    m = re.match(r'^s_cbranch\S*\s+(\.LBB\d+_\d+)', t) or re.match(r'^s_branch\s+(\.LBB\d+_\d+)', t)
    if m and m.group(1) in labels and labels[m.group(1)] <= i and i - labels[m.group(1)] > 200:
        loops.append((labels[m.group(1)], i))
print(sel[:60], 'instructions', len(ins), 'loops', len(loops))
for a, b in loops:
    c = Counter(cls(t) for t in ins[a:b + 1])
    nb = c['barrier']
    tot = sum(c.values())
    per = {k: round(v / max(nb, 1), 1) for k, v in sorted(c.items())}
    print(f'loop [{a},{b}] {tot} instr, {nb} barriers -> per tick:', per)
