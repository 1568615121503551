#!/bin/bash
# per-kernel register / spill / LDS summary of one csrc file: bash tools/kres.sh <stem> [extra hipcc flags]
STEM=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c viewformer_amd/csrc/$STEM.hip -o /tmp/kres_$STEM.o -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys,re
cur=None; d={}
for line in sys.stdin:
    m=re.search(r'remark: Function Name: (\S+)',line)
    if m: cur=m.group(1); d={}; continue
    m=re.search(r'remark:\s+([A-Za-z][A-Za-z ]*?)(?: \[[^\]]*\])?: (\w+)',line)
    if m and cur is not None:
        d[m.group(1).strip()]=m.group(2)
        if m.group(1).strip().startswith('LDS Size'):
            print(cur[:72].ljust(72), 'VGPR',d.get('VGPRs'),'AGPR',d.get('AGPRs'),'SGPR',d.get('TotalSGPRs'),'spill',d.get('VGPRs Spill'),'scratch',d.get('ScratchSize'),'occ',d.get('Occupancy'))
"
