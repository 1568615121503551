#!/usr/bin/env python
"""Summarise the rocprofv3 (rocpd sqlite) outputs written by tools/prof.sh / prof_bench.sh:
per-kernel average duration from the kernel trace, and for PMC passes the per-dispatch TOTAL of each
counter (summed over all XCD/SE/channel instances of the dispatch, then averaged over dispatches).
usage: python tools/summarize_prof.py <prof dir> [kernel-name substrings...]"""
import glob
import os
import sqlite3
import sys
from collections import defaultdict


def tables(c):
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    return lambda p: [t for t in tabs if t.startswith(p)][0]


def short(n):
    n = n.replace('_ZN12_GLOBAL__N_1', '').replace('.kd', '')
    return n[:78]


def main():
    out = sys.argv[1]
    filt = sys.argv[2:] or ['igemm', 'conv3_halo', 'gemm_direct', 'vq_argmin', 'attn_', 'gn_partial', 'layernorm',
                            'conv_in', 'softmax']
    lines = []
    by_shape = {}
    for db in sorted(glob.glob(os.path.join(out, '*', '*.db'))):
        c = sqlite3.connect(db)
        T = tables(c)
        tag = os.path.basename(os.path.dirname(db))
        kd, ks = T('rocpd_kernel_dispatch'), T('rocpd_info_kernel_symbol')
        rows = c.execute(f"select s.kernel_name, count(*), avg(d.end-d.start), sum(d.end-d.start), max(s.arch_vgpr_count), "
                         f"max(d.group_segment_size) from {kd} d join {ks} s on d.kernel_id=s.id group by s.kernel_name "
                         f"order by 4 desc").fetchall()
        tot = sum(r[3] for r in rows) or 1
        if tag.startswith('trace'):
            lines.append(f'== {tag}: rocprofv3 --kernel-trace, all dispatches of the run (incl. warm-up)')
            for n, cnt, avg, s, vg, lds in rows[:16]:
                lines.append(f'  {short(n):78s} calls={cnt:5d} avg_us={avg / 1e3:10.2f} total_ms={s / 1e6:9.3f} '
                             f'pct={100 * s / tot:5.1f} vgpr={vg} lds={lds}')
            # the top kernel broken down by launch shape (grid size): the average of the dominant shape is what bench.py's
            # roofline.avg_launch_ms (HIP events) must agree with
            try:
                top = rows[0][0]
                per = c.execute(f"select d.grid_size_x, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on "
                                f"d.kernel_id=s.id where s.kernel_name=? group by d.grid_size_x order by 4 desc", (top,)).fetchall()
                lines.append(f'  -- {short(top)} by launch shape:')
                for g, cnt, avg, ssum in per[:6]:
                    lines.append(f'       grid_x={g:>9} (workgroups={g // 256:>6})  calls={cnt:4d} avg_us={avg / 1e3:10.2f} total_ms={ssum / 1e6:9.3f}')
            except Exception as e:           # column names differ between rocprofv3 versions: the per-kernel table above stands
                lines.append(f'  -- (per-shape breakdown unavailable: {e})')
            if os.environ.get('PROF_ALL'):   # every kernel of the run, each by launch shape -> summary_full.txt (the hunt for small kernels)
                full = []
                for n, cnt, avg, s, vg, lds in rows:
                    full.append(f'{short(n):78s} calls={cnt:5d} avg_us={avg / 1e3:10.2f} total_ms={s / 1e6:9.3f} pct={100 * s / tot:5.2f} vgpr={vg} lds={lds}')
                    per = c.execute(f"select d.grid_size_x, count(*), avg(d.end-d.start), sum(d.end-d.start) from {kd} d join {ks} s on "
                                    f"d.kernel_id=s.id where s.kernel_name=? group by d.grid_size_x order by 4 desc", (n,)).fetchall()
                    if len(per) > 1:
                        for g, pc, pa, ps in per[:10]:
                            full.append(f'       grid_x={g:>10}  calls={pc:5d} avg_us={pa / 1e3:10.2f} total_ms={ps / 1e6:9.3f}')
                open(os.path.join(out, 'summary_full.txt'), 'w').write('\n'.join(full) + '\n')
            continue
        pe, ip = T('rocpd_pmc_event'), T('rocpd_info_pmc')
        q = (f"select s.kernel_name, p.name, d.id, sum(e.value), count(*) from {pe} e join {ip} p on e.pmc_id=p.id "
             f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, p.name, d.id")
        acc = defaultdict(lambda: defaultdict(list))
        ninst = {}
        for n, pn, did, v, cnt in c.execute(q):
            acc[n][pn].append(v)
            ninst[pn] = cnt
        dur = {r[0]: r[2] for r in rows}
        # per LAUNCH SHAPE (kernel, grid size): a kernel's shapes differ by orders of magnitude in traffic — the mean over all of a kernel's dispatches
        # (below) says nothing about one launch.  Written as JSON too (pmc_by_shape.json): bench.py reads the committed copy for roofline.traffic.
        qs = (f"select s.kernel_name, d.grid_size_x, p.name, d.id, sum(e.value), (d.end - d.start) from {pe} e join {ip} p on e.pmc_id=p.id "
              f"join {kd} d on e.event_id=d.event_id join {ks} s on d.kernel_id=s.id group by s.kernel_name, d.grid_size_x, p.name, d.id")
        shp = defaultdict(lambda: defaultdict(list))
        sdur = defaultdict(list)
        for n, gx, pn, did, v, dt in c.execute(qs):
            if any(f in n for f in filt):
                shp[(n, gx)][pn].append(v)
                sdur[(n, gx)].append(dt)
        by_shape.setdefault(tag, {})
        lines.append(f'== {tag}: rocprofv3 --pmc, per LAUNCH SHAPE (kernel, grid_size_x): mean over the dispatches of that shape')
        for (n, gx), d in sorted(shp.items(), key=lambda kv: -sum(sdur[kv[0]])):
            cnt = len(next(iter(d.values())))
            lines.append(f'  {short(n)[:60]:60s} grid_x={gx:>9} calls={cnt:4d} avg_us={sum(sdur[(n, gx)]) / len(sdur[(n, gx)]) / 1e3:9.2f}  '
                         + ', '.join(f'{k}={sum(v) / len(v):.6g}' for k, v in sorted(d.items())))
            by_shape[tag][f'{short(n)}|{gx}'] = dict(calls=cnt, avg_us=sum(sdur[(n, gx)]) / len(sdur[(n, gx)]) / 1e3,
                                                    **{k: sum(v) / len(v) for k, v in d.items()})
        lines.append(f'== {tag}: rocprofv3 --pmc, per-dispatch totals (sum over instances), mean over dispatches')
        for n, d in acc.items():
            if any(f in n for f in filt):
                lines.append(f'  {short(n)}  avg_us={dur.get(n, 0) / 1e3:.2f}')
                lines.append('      ' + ', '.join(f'{k}={sum(v) / len(v):.5g}' for k, v in sorted(d.items())))
                if 'GRBM_GUI_ACTIVE' in d and dur.get(n):
                    # effective shader clock of the dispatch: busy cycles (per counter instance) / wall time of the dispatch
                    g = sum(d['GRBM_GUI_ACTIVE']) / len(d['GRBM_GUI_ACTIVE']) / max(ninst.get('GRBM_GUI_ACTIVE', 1), 1)
                    lines.append(f'      GRBM_GUI_ACTIVE per instance ({ninst.get("GRBM_GUI_ACTIVE", 1)} instances) = {g:.5g} cycles -> '
                                 f'{g / (dur[n] / 1e3) / 1e3:.3f} GHz effective clock over the dispatch')
    open(os.path.join(out, 'summary.txt'), 'w').write('\n'.join(lines) + '\n')
    if by_shape:
        import json
        json.dump(by_shape, open(os.path.join(out, 'pmc_by_shape.json'), 'w'), indent=1)
    print('\n'.join(lines))


if __name__ == '__main__':
    main()
