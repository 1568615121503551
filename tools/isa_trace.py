#!/usr/bin/env python
"""Compact op/waitcnt trace of the largest basic block of a kernel in a hipcc -S dump.
M mfma, G global load, L ds_read, W ds_write, v VALU, s SALU, n s_nop, [..] s_waitcnt, # sched/other barrier
usage: python tools/isa_trace.py file.s <kernel-name substring> [max chars]"""
import re
import sys

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
blocks, cur = [], []
for l in lines[start:end]:
    if re.match(r'^\.LBB', l):
        blocks.append(cur)
        cur = []
    cur.append(l)
blocks.append(cur)
big = max(blocks, key=len)
out = []
for o in (l.strip() for l in big):
    if not o or o.startswith(('.',)):
        continue
    if o.startswith(';'):
        if 'sched_barrier' in o:
            out.append('#\n')
        continue
    t = o.split(' ')[0]
    if t.startswith('v_mfma'):
        out.append('M')
    elif t == 's_waitcnt':
        out.append('[' + o.split(' ', 1)[1].strip().replace('vmcnt', 'vm').replace('lgkmcnt', 'lgkm') + ']')
    elif t.startswith(('global_load', 'buffer_load')):
        out.append('G')
    elif t.startswith(('global_store', 'buffer_store')):
        out.append('S')
    elif t.startswith('ds_read'):
        out.append('L')
    elif t.startswith('ds_write'):
        out.append('W')
    elif t.startswith('s_barrier'):
        out.append('|BAR|')
    elif t.startswith('v_'):
        out.append('v')
    elif t.startswith('s_nop'):
        out.append('n')
    else:
        out.append('s')
s = ''.join(out)
s = re.sub(r'(v{4,})', lambda m: f'v{len(m.group(1))}', s)
s = re.sub(r'(s{3,})', lambda m: f's{len(m.group(1))}', s)
print(s[:int(sys.argv[3]) if len(sys.argv) > 3 else 100000])
