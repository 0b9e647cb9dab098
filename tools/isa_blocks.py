#!/usr/bin/env python3
"""Developer tool: per-basic-block instruction mix of one kernel in a hipcc -save-temps .s file.
usage: isa_blocks.py file.s <substring of mangled kernel name> [block-to-dump]"""
import re, sys, collections
s = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(s) if re.match(r'^_Z\S*:', l) and key in l.split(':')[0])
end = next(i for i in range(start, len(s)) if 's_endpgm' in s[i] and not any('.LBB' in s[j] for j in range(i+1, min(i+3, len(s)))) ) if False else next(i for i in range(start, len(s)) if s[i].startswith('.Lfunc_end'))
blocks = []; name = 'entry'; cur = []
for l in s[start + 1:end]:
    if re.match(r'^\.LBB\d+_\d+:', l):
        blocks.append((name, cur)); name = l.split(':')[0]; cur = []
    else:
        t = l.strip()
        if t and not t.startswith(';') and not t.startswith('.'): cur.append(t)
blocks.append((name, cur))
def cls(i):
    op = i.split()[0]
    if op.startswith('v_mfma'): return 'mfma'
    if op.startswith('ds_read') or op.startswith('ds_load'): return 'ds_read'
    if op.startswith('ds_write') or op.startswith('ds_store'): return 'ds_write'
    if op.startswith('global_load') or op.startswith('buffer_load'): return 'vmem_ld'
    if op.startswith('global_store') or op.startswith('buffer_store'): return 'vmem_st'
    if op.startswith('scratch_'): return 'scratch'
    if op.startswith('s_waitcnt'): return 'waitcnt'
    if op.startswith('s_barrier'): return 'barrier'
    if op.startswith('s_nop'): return 'nop'
    if op.startswith('v_accvgpr'): return 'accvgpr'
    if op.startswith('v_'): return 'valu'
    if op.startswith('s_'): return 'salu'
    return 'other'
for n, b in blocks:
    c = collections.Counter(cls(i) for i in b)
    if len(b) > 40: print('%-10s %5d instrs  %s' % (n, len(b), dict(c)))
if len(sys.argv) > 3:
    for n, b in blocks:
        if n == sys.argv[3]: print('\n'.join(b))
