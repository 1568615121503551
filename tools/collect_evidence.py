#!/usr/bin/env python3
"""Copy what one `tools/gpu/evidence.sh` run left under gpurun_out/ into the tracked, per-round files under profiles/
(gpurun_out/ is scratch).  Usage: python tools/collect_evidence.py r6

Nothing is computed here that the run did not print, except the GEMM-family sum of the training trace (recomputed from the trace's own
rows, shown with its terms) and the dominant launch's line of the inference trace."""
import json
import os
import re
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(REPO, 'gpurun_out')
E = os.path.join(OUT, 'evidence')
TRAIN_GFLOP = 9861.0                 # GEMM family of one ALL-ROWS training step; the run's own figure (bench line: summed over the launches made) is used when present
GEMM_FAMILY = ('gemm_bf16_g256_kernel', 'gemm_tn_bf16_kernel', 'sum_slabs_kernel', 'igemm_f32_kernel', 'gemm_bf16_direct_kernel',
               'dense_small_n')


def last_json_line(path):
    for line in reversed(open(path).read().strip().splitlines()):
        if line.startswith('{'):
            return json.loads(line)
    raise SystemExit(f'no JSON line in {path}')


def trace_rows(summary):
    rows = []
    for line in open(summary):
        m = re.match(r'\s+(\S+)\s+calls=\s*(\d+)\s+avg_us=\s*([\d.]+)\s+total_ms=\s*([\d.]+)', line)
        if m and not line.lstrip().startswith('grid_x'):
            rows.append((m.group(1), int(m.group(2)), float(m.group(3)), float(m.group(4))))
    return rows


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r6'
    P = lambda name: os.path.join(REPO, 'profiles', f'{tag}_{name}')
    for src, dst in (('bench_views.json', 'bench_mixed_batch128.json'), ('bench_train.json', 'bench_train.json'),
                     ('bench_allimg.json', 'bench_allimg.json'), ('bench_s20.json', 'bench_s20.json'),
                     ('bench_sweep.json', 'bench_batch_sweep.json'), ('bench_bare_gpus2_gloo.json', 'bench_bare_gpus2_gloo.json'),
                     ('pmc_traffic.json', 'pmc_traffic.json'), ('parity_report.jsonl', 'parity_report.jsonl')):
        shutil.copy(os.path.join(E, src), P(dst))
    views, train = last_json_line(os.path.join(E, 'bench_views.json')), last_json_line(os.path.join(E, 'bench_train.json'))

    # ---- the suite's and the smoke's last lines
    gpu = [l for l in open(os.path.join(E, 'gpu.log')).read().splitlines() if re.search(r'\d+ passed|failed|error', l)]
    smoke = open(os.path.join(E, 'smoke.log')).read().strip().splitlines()[-1]
    refused = open(os.path.join(E, 'bench_bare_gpus2_refused.err')).read().strip().splitlines()[-1]
    with open(P('gpu_tests.txt'), 'w') as f:
        f.write(f'# round {tag[1:]} evidence run (tools/gpu/evidence.sh, one gpurun call, final code): the -m gpu suite\'s and the smoke\'s last lines\n')
        f.write('\n'.join(gpu[-3:]) + '\n' + smoke + '\n')
        f.write('bare --gpus 2 without VF_DIST_BACKEND=gloo on the 1-GPU box:\n' + refused + '\n')

    # ---- pre-flight lines
    with open(P('preflight.jsonl'), 'w') as f:
        f.write(f'# bench.py --gpus N --preflight on a 1-GPU MI355X box (round {tag[1:]} evidence run): world 1 over RCCL, world 2 over gloo (ranks '
                'share cuda:0), and the line the bare \'bench.py --gpus 2\' (gloo) printed on stderr before its run\n')
        for name in ('preflight_1rank_rccl.json', 'preflight_2ranks_gloo.json'):
            f.write(open(os.path.join(E, name)).read().strip() + '\n')
        for line in open(os.path.join(E, 'bench_bare_gpus2_gloo.err')):
            if '"preflight"' in line:
                f.write(line.strip() + '\n')

    # ---- kernel traces
    src = os.path.join(OUT, 'prof_evidence', 'summary.txt')
    body = open(src).read()
    dom = re.search(r'grid_x=\s*29360128 \(workgroups=114688\)\s+calls=\s*\d+\s+avg_us=\s*([\d.]+)', body)
    with open(P('bench_mixed_kernel_trace.txt'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats of: python bench.py --no-cpu-baseline --no-f32-arm --steps 3 --warmup 1   (final code, '
                f'tools/prof_bench.sh evidence; one box, same gpurun call as {tag}_bench_mixed_batch128.json)\n')
        f.write(f'# dominant launch shape: conv3_halo_x3h16_kernel grid 114688 workgroups — {float(dom.group(1)) / 1000:.2f} ms here; '
                f'roofline.avg_launch_ms of the bench line (HIP events, unprofiled run, same call): {views["roofline"]["avg_launch_ms"]:.2f} ms\n')
        f.write(body)
    src = os.path.join(OUT, 'prof_evidence_train', 'summary.txt')
    body = open(src).read()
    rows = [r for r in trace_rows(src) if any(k in r[0] for k in GEMM_FAMILY)]
    line = last_json_line(src) if '{"metric"' in body else None
    steps = (line['steps'] + line['warmup']) if line else 4
    calls_adamw = [r[1] for r in trace_rows(src) if 'adamw_pack_tiles' in r[0] or 'adamw_flat' in r[0]]      # one optimizer launch per step
    n_steps = calls_adamw[0] if calls_adamw else steps
    tot = sum(r[3] for r in rows) / n_steps
    terms = ' + '.join(f'{r[3]:.3f} [{r[0][2:22]}]' for r in rows)
    gflop = float(train['roofline'].get('algorithmic_gflop_per_step') or TRAIN_GFLOP)      # (the same code ran both: the launches' own flops)
    with open(P('train_step_kernel_trace.txt'), 'w') as f:
        f.write('# rocprofv3 --kernel-trace --stats of: python bench.py --no-cpu-baseline --no-f32-arm --workload train --serial-wgrad --steps 3 '
                '--warmup 1   (final code, tools/prof_bench.sh evidence_train)\n')
        f.write(f'# --serial-wgrad keeps the weight-gradient GEMMs on the main stream in EVERY step, so no duration below is stretched by a '
                f'concurrent kernel ({n_steps} steps in the run).\n')
        f.write(f'# GEMM family per step, recomputed from this table: ({terms}) / {n_steps} = {tot:.2f} ms -> {gflop:.0f} GFLOP (summed over the launches made: figure of the bench line) / {tot:.2f} ms = '
                f'{gflop / tot:.0f} TF = {gflop / tot / 2500:.3f} of the bf16 peak;\n')
        r = train['roofline']
        f.write(f'#   bench.py --workload train (HIP events, one serialised step, same gpurun call): {r.get("kernel_ms_per_step")} ms, '
                f'{r["achieved"]} TF, {r["frac"]}\n')
        f.write(body)

    # ---- counter summaries
    with open(P('bench_pmc_summary.txt'), 'w') as f:
        f.write(f'# Round {tag[1:]}: HBM-side counters per LAUNCH SHAPE, inference step (bash tools/prof_bench_pmc.sh evidence_views = rocprofv3 '
                '--kernel-trace --pmc FETCH_SIZE | WRITE_SIZE, one pass each, over\n# python bench.py --no-cpu-baseline --no-f32-arm --steps 1 '
                '--warmup 1) and training step (tools/prof_train_pmc.sh).  Values: KB, summed over counter instances, mean over the dispatches of '
                'the shape;\n# HBM read bytes = 2 x FETCH_SIZE x 1024 (the guide\'s gfx950 correction), written = WRITE_SIZE x 1024.  '
                f'Machine-readable: profiles/{tag}_pmc_traffic.json (what bench.py reads).\n\n## inference step\n')
        f.write(open(os.path.join(OUT, 'pmc_evidence_views', 'summary.txt')).read())
        f.write('\n## training step\n')
        f.write(open(os.path.join(OUT, 'pmc_evidence_train', 'summary.txt')).read())
    print('views', views['value'], views['ms_per_step'], 'frac', views['roofline']['frac'])
    print('train', train['value'], train['ms_per_step'], 'family', train['roofline']['frac'], 'trace family ms', round(tot, 2))


if __name__ == '__main__':
    main()
