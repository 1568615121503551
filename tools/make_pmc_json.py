#!/usr/bin/env python
"""merge the per-launch-shape PMC summaries of tools/prof_bench_pmc.sh / prof_train_pmc.sh (gpurun_out/pmc_<tag>/pmc_by_shape.json, written by
tools/summarize_prof.py) into the file bench.py reads for roofline.traffic:  python tools/make_pmc_json.py views=<dir> train=<dir> > profiles/rN_pmc_traffic.json"""
import json
import os
import sys

out = {}
for arg in sys.argv[1:]:
    workload, d = arg.split('=')
    by = json.load(open(os.path.join(d, 'pmc_by_shape.json')))
    w = out.setdefault(workload, {})
    for tag, entries in by.items():
        for key, e in entries.items():
            for counter in ('FETCH_SIZE', 'WRITE_SIZE'):
                if counter in e:
                    w.setdefault(counter, {})[key] = {'calls': e['calls'], 'avg_us': round(e['avg_us'], 2), counter: e[counter]}
out['_note'] = ('rocprofv3 --kernel-trace --pmc FETCH_SIZE (pass 1) / WRITE_SIZE (pass 2) over one bench step; key = <kernel>|<grid_size_x>; values = mean over the '
                'dispatches of that launch shape, summed over counter instances; KB.  HBM read bytes = 2 x FETCH_SIZE x 1024 (gfx950: MI355X_MICROARCH.md), '
                'written bytes = WRITE_SIZE x 1024')
json.dump(out, sys.stdout, indent=1)
