#!/usr/bin/env python
"""Per-basic-block instruction mix of one kernel in a hipcc -S dump (blocks with MFMAs or >= 40 instructions).
usage: python tools/isa_blocks.py file.s <substring of the mangled kernel name> [min instructions]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
minn = int(sys.argv[3]) if len(sys.argv) > 3 else 40
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
end = next(i for i in range(start, len(lines)) if 's_endpgm' in lines[i])
blocks, cur, name = [], [], 'entry'
for l in lines[start:end]:
    m = re.match(r'^(\.LBB\S+):', l)
    if m:
        blocks.append((name, cur))
        cur, name = [], m.group(1)
    cur.append(l)
blocks.append((name, cur))
for name, b in blocks:
    ops = [l.strip().split(' ')[0] for l in b if l.strip() and not l.strip().startswith((';', '.'))]
    c = Counter()
    for t in ops:
        c['MFMA' if t.startswith('v_mfma') else 'VALU' if t.startswith('v_') else 'LDS' if t.startswith('ds_') else
          'VMEM' if t.startswith(('global_', 'buffer_', 'flat_', 'scratch_')) else 'wait' if t.startswith('s_waitcnt') else 'SALU'] += 1
    if c['MFMA'] or len(ops) >= minn:
        print(f'{name:12s} n={len(ops):4d} {dict(c)}')
        print('     ', Counter(t for t in ops if t.startswith('v_') and not t.startswith('v_mfma')).most_common(14))
