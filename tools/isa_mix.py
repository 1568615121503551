#!/usr/bin/env python
"""Instruction mix of the largest basic block (the main loop) of one kernel in a hipcc -S dump.
usage: python tools/isa_mix.py file.s <substring of the mangled kernel name>"""
import re
import sys
from collections import Counter

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
ops = [l.strip().split(' ')[0] for l in big if l.strip() and not l.strip().startswith((';', '.'))]
c = Counter()
for t in ops:
    if t.startswith('v_mfma'):
        c['MFMA'] += 1
    elif t.startswith('v_'):
        c['VALU'] += 1
    elif t.startswith('ds_'):
        c['LDS'] += 1
    elif t.startswith(('global_', 'buffer_', 'flat_', 'scratch_')):
        c['VMEM'] += 1
    elif t.startswith('s_waitcnt'):
        c['waitcnt'] += 1
    elif t.startswith('s_'):
        c['SALU'] += 1
print(f'{len(blocks)} blocks; largest has {len(ops)} instructions:', dict(c))
print(Counter(t for t in ops if t.startswith('v_') and not t.startswith('v_mfma')).most_common(25))
print(Counter(t for t in ops if not t.startswith('v_')).most_common(15))
