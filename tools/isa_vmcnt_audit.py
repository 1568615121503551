#!/usr/bin/env python
"""Which `s_waitcnt vmcnt(N)` inside LOOPS of a kernel did the COMPILER insert (as opposed to the kernel's own inline-asm waits)?
hipcc's waitcnt pass conservatively drains vmcnt before an LDS read it cannot prove independent of a pending LDS-DMA
(buffer_load ... lds / global_load_lds): inside a DMA ring that turns "three tiles in flight" into "wait for the newest tile every step".
usage: python tools/isa_vmcnt_audit.py file.s [kernel-name substring]   (file.s from hipcc -S --cuda-device-only)"""
import re
import sys


def audit(path, key=''):
    """-> {mangled kernel name: [(line number, block label, wait instruction, next instruction)]} for every s_waitcnt with a vmcnt field that sits
    in a loop block and is NOT inside an inline-asm region (;;#ASMSTART ... ;;#ASMEND)"""
    lines = open(path).read().split('\n')
    kern, in_app, in_loop, label = None, False, False, ''
    out = {}
    for i, l in enumerate(lines):
        m = re.match(r'^(_Z\S+):', l)
        if m:
            kern, in_loop = m.group(1), False
            continue
        if kern is None:
            continue
        if 's_endpgm' in l:
            kern = None
            continue
        s = l.strip()
        if s.startswith((';APP', ';;#ASMSTART')):
            in_app = True
        elif s.startswith((';NO_APP', ';;#ASMEND')):
            in_app = False
        m = re.match(r'^(\.LBB\S+):(.*)', l)
        if m:
            label = m.group(1)
            in_loop = 'Loop' in m.group(2)
        if in_loop and not in_app and re.search(r's_waitcnt.*vmcnt\(\d+\)', s) and key in kern:
            nxt = next((x.strip() for x in lines[i + 1:i + 6] if x.strip() and not x.strip().startswith(';')), '')
            out.setdefault(kern, []).append((i + 1, label, s, nxt))
    return out


if __name__ == '__main__':
    res = audit(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else '')
    for k, v in res.items():
        print(k[:110])
        for ln, lab, s, nxt in v:
            print(f'    line {ln:6d} {lab:12s} {s:34s} -> {nxt[:70]}')
    if not res:
        print('no compiler-inserted vmcnt waits inside loops')
