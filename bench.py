#!/usr/bin/env python
"""bench.py — novel views/sec of the ViewFormer hot path on MI355X (BASELINE.json metric).

One "step" = one pass of ``generate_batch_predictions`` (evaluate_transformer.py:97-146) over one batch of synthetic scenes already
resident in HBM: uint8 frames [B,7,128,128,3] + cameras [B,7,7] -> encode all 7 views (target included, as the reference does) ->
MIGT pass with the MASK view -> argmax -> decode -> uint8 novel view (+ the localization pass the SM7 model runs).
Workload = BASELINE.json configs[1]: SM7 codebook + transformer, 6 context views -> 1 novel view, bf16: the encoder and the codebook
lookup stay exact fp32 (bit-exact token indices), the transformer's dense layers and the decoder's convolutions run on bf16 MFMA with
fp32 accumulation (--precision f32 runs the all-fp32 parity arm).  Inputs are resident in HBM when the timed region starts and the
generated images stay in HBM (``host_io`` in the line reports the rate with the host round trip the reference's loop makes).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Other workloads through the same contract (one JSON line, barrier + max-over-ranks timing):
  --workload train    BASELINE configs[3]: one data-parallel MIGT training step (CO3D 10-cat finetune: seq 10, 3 streams, 10 scenes
                      per GPU, RCCL gradient all-reduce SUM overlapped with the backward pass); value = scenes/s, whole job
  --workload allimg   BASELINE configs[4]: the all-images evaluator loop (transformer batch 128 / decode batch 64,
                      evaluate_transformer_multictx_allimg.py:173,177); bf16 attention unless --attention fp8; value = generated views/s
  --views 20              BASELINE configs[2]: 19-view context, image + localization heads (default 45 scenes = 900 images per step)

N>1: one process per GPU, scenes sharded (weak scaling: --batch scenes per GPU per step), weights replicated, no data-path
collective on the inference workloads; rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

import torch  # noqa: E402

F32_MFMA_PEAK_TFLOPS = 157.3     # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak (dense, f32 in)
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 / f16 / non-scaled fp8 MFMA peak
HBM_PEAK_GBS = 8000.0
# measured, not a contract peak: what a loop of nothing but 16-bit MFMAs sustains on full-entropy operands before the package power limit
# (tools/mfma16_peak.hip, profiles/r4_power_ceiling_probe.txt; 2440 on operands that do not toggle)
F16_MFMA_POWER_LIMITED_TFLOPS = 1722.0
BF16_MFMA_POWER_LIMITED_TFLOPS = 1839.0

# HBM traffic (roofline.traffic): PMC counters cannot be read from inside this process (rocprofv3 wraps the run), so the line quotes the
# committed counter passes of THIS round's code — tools/prof_bench_pmc.sh = two `rocprofv3 --kernel-trace --pmc` passes (FETCH_SIZE, WRITE_SIZE: they do
# not fit one pass) over `python bench.py --steps 1 --warmup 1`, summarised per launch shape by tools/summarize_prof.py — and says so in
# traffic_source.  Units as MI355X_MICROARCH.md prescribes: FETCH_SIZE / WRITE_SIZE in KB, read bytes = 2 x FETCH_SIZE (gfx950 correction).
PMC_FILE = 'profiles/r6_pmc_traffic.json'


def pmc_traffic(kernel_substr, workgroups, threads_per_wg=256, workload='views'):
    """bytes per launch of the launch shape (kernel name contains `kernel_substr`, grid = workgroups x threads) from the committed PMC passes, or None"""
    try:
        with open(os.path.join(REPO, PMC_FILE)) as f:
            d = json.load(f)[workload]
    except (OSError, KeyError, ValueError):
        return None
    grid = str(workgroups * threads_per_wg)
    rd = [v for k, v in d.get('FETCH_SIZE', {}).items() if kernel_substr in k and k.endswith('|' + grid)]
    wr = [v for k, v in d.get('WRITE_SIZE', {}).items() if kernel_substr in k and k.endswith('|' + grid)]
    if not rd or not wr:
        return None
    n = sum(e['calls'] for e in rd)
    fetch_kb = sum(e['FETCH_SIZE'] * e['calls'] for e in rd) / n
    write_kb = sum(e['WRITE_SIZE'] * e['calls'] for e in wr) / sum(e['calls'] for e in wr)
    return {'bytes': int((2 * fetch_kb + write_kb) * 1024), 'read_bytes': int(2 * fetch_kb * 1024), 'written_bytes': int(write_kb * 1024), 'dispatches': n,
            'source': f'{PMC_FILE}: rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE passes of this round\'s code over one bench step, mean of the {n} dispatches of '
                      f'this launch shape (grid {grid}); read bytes = 2 x FETCH_SIZE x 1024 (gfx950), written = WRITE_SIZE x 1024; not measured in THIS run '
                      f'(PMC counters need rocprofv3 around the process)'}


def build_models(dev, localization: bool, precision: str = 'f32', conv_arith: str = 'x3h', bf16_activations: bool = True,
                 encoder_chunk: int = 1024, attention=None, sequence_size: int = 6, decoder_act16=None):
    from viewformer_amd.config import VQGANConfig, MIGTConfig
    from viewformer_amd.weights import make_vqgan_weights, make_migt_weights
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.migt import MIGT
    vcfg = VQGANConfig()
    # SM7 transformer (README.md:348-360): seq 6 (+ the generated view), pose-multiplier 0.2,
    # localization schedule cosine(0,1,120000) => the localization head is on.
    mcfg = MIGTConfig(sequence_size=sequence_size, n_loss_skip=1, pose_multiplier=0.2,
                      localization_weight='cosine(0,1,120000)' if localization else '0')
    vsd = make_vqgan_weights(vcfg, seed=0, codebook_scale=0.05)
    msd = make_migt_weights(mcfg, seed=0)
    # precision 'mixed': encoder + codebook lookup exact fp32 (bit-exact tokens), transformer dense layers and decoder
    # convolutions on bf16 MFMA (tolerance-bounded logits / pixels) — the split the north star specifies
    arm = 'bf16' if precision == 'mixed' else 'f32'
    vq = VQGAN(vcfg, data_format='NHWC', decoder_precision=arm, conv_arith=conv_arith,
               max_images_per_call=encoder_chunk).load_state_dict(vsd).to(dev)
    if decoder_act16 is not None:
        vq.decoder_act16 = int(decoder_act16)
    # the transformer's fp32 dense layers follow the convolutions' arithmetic (x3h: LayerNorm / GELU / attention outputs are O(1))
    tr = MIGT(mcfg, precision=arm, dense_arith=conv_arith, bf16_activations=bf16_activations,
              attention=attention if arm == 'bf16' else None).load_state_dict(msd).to(dev)
    return vq, tr, (vcfg, vsd, mcfg, msd)


def flops_per_view(S: int, localization: bool):
    """algorithmic GFLOP per novel view of the REFERENCE's loop (two transformer passes when localizing), SURVEY.md §8(d)"""
    enc, dec = 34.507, 63.053
    gemm = 0.906 * S * 12
    attn = 12.58e-3 * S * (S + 1) / 2 * 12
    head = 0.101
    tr = gemm + attn + head
    return S * enc + dec + tr + ((gemm + attn) if localization else 0.0)


class OpTimer:
    """HIP-event timing (on the launch stream) of the C-ABI launches that matter for the roofline section: every implicit-GEMM launch,
    the codebook lookup and the transformer's attention."""

    def __init__(self, workload='views'):
        self.gemm, self.vq, self.attn = [], [], []
        self.workload = workload

    def install(self):
        from viewformer_amd import ops
        self._orig = (ops.igemm, ops.vq_argmin_filtered, ops.vq_argmin, ops.attn_blockcausal, ops.gemm_tn_bf16)
        t = self

        def ev():
            return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def igemm(x, w_packed, M, Cin, Cout, out, *a, **kw):
            mode = kw.get('mode', ops.MODE_GEMM)
            batch = kw.get('batch', 1)
            taps = 1 if mode == ops.MODE_GEMM else 9
            e0, e1 = ev()
            e0.record()
            r = t._orig[0](x, w_packed, M, Cin, Cout, out, *a, **kw)
            e1.record()
            t.gemm.append((e0, e1, 2.0 * M * Cin * Cout * taps * batch, (mode, M, Cin, Cout, batch)))
            return r

        def vqf(z, blob, D, Kc, *a, **kw):
            e0, e1 = ev()
            e0.record()
            r = t._orig[1](z, blob, D, Kc, *a, **kw)
            e1.record()
            t.vq.append((e0, e1, z.numel() // D, D, Kc, 'vq_filter_kernel (fp16 filter + exact fp32 re-rank)'))
            return r

        def vqe(z, Ep, esq, D, Kc):
            e0, e1 = ev()
            e0.record()
            r = t._orig[2](z, Ep, esq, D, Kc)
            e1.record()
            t.vq.append((e0, e1, z.numel() // D, D, Kc, 'vq_argmin_kernel (f32 MFMA)'))
            return r

        def attn(q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale=1.0, skip_masked=True, twin_view=-1, **kw):
            e0, e1 = ev()
            e0.record()
            r = t._orig[3](q, k, v, out, B, H, T, L, ldq, ldk, ldv, ldo, scale, skip_masked, twin_view, **kw)
            e1.record()
            S = T // max(L, 1)
            if twin_view >= 0:
                pairs = twin_view * (twin_view + 1) // 2 + (S - twin_view) * (twin_view + 1)
            elif twin_view <= -2:
                sv = -twin_view
                pairs = sv * (sv + 1) // 2 + (S // sv - 1) * sv * (sv + 1) // 2        # branch position i: i main views + itself
            else:
                pairs = S * (S + 1) // 2
            arm = 'fp8' if kw.get('fp8') else 'bf16' if kw.get('bf16') else 'x6' if kw.get('x6') else 'f32'
            io16 = q.dtype == torch.bfloat16 and out.dtype == torch.bfloat16
            from viewformer_amd import _lib as _vl
            dma = arm == 'bf16' and io16 and L == 64 and T % 64 == 0 and bool(_vl.load().vf_selected(_vl.SEL_ATTN_DMA))
            by = B * T * H * 64 * (3 * q.element_size() + out.element_size())             # q, k, v read once, o written once
            t.attn.append((e0, e1, 4.0 * H * 64 * L * L * pairs * B, arm, (B, H, T, L, twin_view), dma, by))
            return r
        def gemm_tn(x16, dy, M, K, N, dw, db=None, accumulate=True, **kw):    # the training step's weight-gradient GEMM (+ its slab sums)
            e0, e1 = ev()
            e0.record()
            r = t._orig[4](x16, dy, M, K, N, dw, db, accumulate, **kw)
            e1.record()
            t.gemm.append((e0, e1, 2.0 * M * K * N, ('tn', M, K, N, 1)))
            return r
        ops.igemm, ops.vq_argmin_filtered, ops.vq_argmin, ops.attn_blockcausal, ops.gemm_tn_bf16 = igemm, vqf, vqe, attn, gemm_tn

    def uninstall(self):
        from viewformer_amd import ops
        ops.igemm, ops.vq_argmin_filtered, ops.vq_argmin, ops.attn_blockcausal, ops.gemm_tn_bf16 = self._orig

    def gemm_summary(self):
        torch.cuda.synchronize()
        tot_ms, tot_fl, by = 0.0, 0.0, {}
        for e0, e1, fl, key in self.gemm:
            ms = e0.elapsed_time(e1)
            tot_ms += ms
            tot_fl += fl
            k = by.setdefault(key, [0.0, 0.0, 0])
            k[0] += ms; k[1] += fl; k[2] += 1
        top = sorted(by.items(), key=lambda kv: -kv[1][0])[:int(os.environ.get('VF_BENCH_TOP_SHAPES', '5'))]
        return tot_ms, tot_fl, len(self.gemm), top

    def vq_entry(self):
        if not self.vq:
            return None
        e0, e1, M, D, Kc, name = max(self.vq, key=lambda r: r[2])
        ms = e0.elapsed_time(e1)
        by = M * D * 4 + D * Kc * 4 + M * 8                                      # 66 048 B / image + the codebook once
        fl = 2.0 * M * D * Kc
        mfma_peak = BF16_MFMA_PEAK_TFLOPS if 'filter' in name else F32_MFMA_PEAK_TFLOPS
        return {'kernel': name, 'bound': 'hbm (north-star accounting; the arithmetic intensity, 508 FLOP/B, makes the matrix pipe the real roof)',
                'rows': M, 'avg_launch_us': round(ms * 1e3, 1), 'achieved': round(by / ms / 1e6, 1), 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                'frac': round(by / ms / 1e6 / HBM_PEAK_GBS, 4), 'algorithmic_bytes_per_launch': by,
                'mfma_tflops': round(fl / ms / 1e9, 1), 'mfma_frac': round(fl / ms / 1e9 / mfma_peak, 4), 'mfma_peak': mfma_peak}

    def attn_entry(self):
        if not self.attn:
            return None
        ms = sum(e0.elapsed_time(e1) for e0, e1, *_ in self.attn)
        fl = sum(r[2] for r in self.attn)
        arm, shape, dma = self.attn[0][3], self.attn[0][4], self.attn[0][5]
        by = sum(r[6] for r in self.attn)
        peak = {'fp8': BF16_MFMA_PEAK_TFLOPS, 'bf16': BF16_MFMA_PEAK_TFLOPS, 'x6': BF16_MFMA_PEAK_TFLOPS / 6, 'f32': F32_MFMA_PEAK_TFLOPS}[arm]
        name = {'fp8': 'attn_lp_kernel<fp8 e4m3>', 'bf16': 'attn_lp_kernel<bf16>', 'x6': 'attn_blockcausal_x6_kernel',
                'f32': 'attn_blockcausal_kernel'}[arm]
        if dma:
            name = 'attn_dma_kernel (bf16 q/k/v tiles by LDS-DMA, 3 in flight)'
        return {'kernel': name, 'bound': 'mfma', 'launches': len(self.attn),
                'B_H_T_L_twin': list(shape), 'avg_launch_us': round(ms / len(self.attn) * 1e3, 1),
                'achieved': round(fl / ms / 1e9, 1), 'peak': round(peak, 1), 'unit': 'TFLOP/s (useful: visible tile pairs only)',
                'frac': round(fl / ms / 1e9 / peak, 4),
                # the other roof: q, k, v read once + o written once (this shape: 140 FLOP/B -> the HBM roof sits at 45 % of the bf16 peak)
                'hbm': {'algorithmic_bytes_per_launch': by // len(self.attn), 'achieved': round(by / ms / 1e6, 1), 'peak': HBM_PEAK_GBS,
                        'unit': 'GB/s', 'frac': round(by / ms / 1e6 / HBM_PEAK_GBS, 4),
                        'traffic': (lambda t: None if t is None else {'bytes': t['bytes'], 'over_algorithmic': round(t['bytes'] / (by // len(self.attn)), 3),
                                                                      'source': t['source']})(
                            # (the counter passes ran over the default workload's launch — 128 scenes x 12 heads x 512 tokens; the trace's grid_x only carries the
                            # head count, so any other shape would match it by accident: no figure for those)
                            pmc_traffic('attn_dma_kernel', shape[1], 512, self.workload) if (dma and tuple(shape[:3]) == (128, 12, 512)) else None)},
                'binding_roof_note': 'the north star prices this kernel against the MFMA peak; at 64 features per head and 64-token views its arithmetic intensity '
                                     '(useful FLOP per byte of q, k, v, o) puts the HBM roof BELOW the matrix roof — see hbm.frac (algorithmic bytes) and '
                                     'hbm.traffic (counter bytes incl. the re-read of key tiles by the second query block)',
                'peak_note': ('non-scaled fp8 MFMA runs at the bf16 rate' if arm == 'fp8' else
                              'x6 executes 6 bf16 MFMA flops per fp32 flop: peak = 2500 / 6' if arm == 'x6' else '')}


def cpu_baseline(models_cfg, S, n_scenes, seed=123):
    """the oracle (torch-CPU restatement of the reference path) timed on the host cores"""
    from oracle import pipeline_oracle as po
    from viewformer_amd.weights import synthetic_scene_batch
    vcfg, vsd, mcfg, msd = models_cfg
    # torch's CPU conv/matmul stop scaling (and regress badly) past a few dozen threads at these sizes:
    # a 256-thread run of this sample took 14 s/scene on the GPU box's host, so cap the pool.
    host_cores = os.cpu_count() or 1
    cores = min(host_cores, 32)
    torch.set_num_threads(cores)
    frames, cams = synthetic_scene_batch(max(n_scenes, 1), S, 128, seed=seed)
    t0 = time.time()
    po.generate_batch_predictions(msd, mcfg, vsd, vcfg, frames[:1], cams[:1])      # warm (also sizes the sample)
    warm = time.time() - t0
    n = max(1, min(n_scenes, int(20.0 / max(warm, 1e-3))))                         # bound the sample to ~20 s
    stages = {}
    t0 = time.time()
    po.generate_batch_predictions(msd, mcfg, vsd, vcfg, frames[:n], cams[:n], timings=stages)
    dt = time.time() - t0
    return dict(value=round(n / dt, 4), unit='novel views/s', cores=cores, host_cores=host_cores, kind='port',
                stage_ms_per_view={k: round(v / n * 1e3, 1) for k, v in stages.items()},
                stage_note=f'encode = {S} frames per view (target included), transformer = generation + localization passes, '
                           'decode = 1 frame per view',
                sample=f'{n} scenes x {S} views (same synthetic workload), fp32 torch-CPU restatement of the '
                       f'reference path (oracle/), {cores} of the host\'s {host_cores} logical cores as torch threads '
                       f'(the full pool measured 14x slower: torch CPU conv / matmul regress past a few dozen threads), {dt:.1f} s')


def _tail_on():
    from viewformer_amd import _lib
    return bool(_lib.load().vf_selected(_lib.SEL_GEMM_TAIL))


def timed(step, steps, warmup, dev):
    """the contract's timed region: W warm-up steps, then exactly K steps between barrier + synchronize, max over ranks"""
    from viewformer_amd import sharding
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = step()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    return sharding.max_over_ranks(time.perf_counter() - t0, dev), out


# ------------------------------------------------------------------------------------------------ --workload train (configs[3])
def run_train(args, rank, local, world, dev):
    import numpy as np
    from viewformer_amd import geometry, sharding
    from viewformer_amd.config import MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.train import MIGTTrainer
    from viewformer_amd.weights import make_migt_weights, synthetic_scene_batch
    B, S = (args.batch if args.batch_set else 10), (args.views if args.views_set else 10)
    # CO3D 10-category finetune (README.md:250-264): seq 10, n_loss_skip 1, global batch 80 = 10 scenes per GPU on 8 GPUs,
    # localization weight 5, pose multiplier 0.05, lr 1e-4, weight decay 0.05, 40 k steps
    cfg = MIGTConfig(sequence_size=S, n_loss_skip=1, localization_weight='5', pose_multiplier=0.05, dropout=args.dropout,
                     learning_rate=1e-4, weight_decay=0.05, total_steps=40000, batch_size=B * world)
    arm = 'bf16' if args.precision == 'mixed' else 'f32'
    model = MIGT(cfg, precision=arm).load_state_dict(make_migt_weights(cfg, seed=0)).to(dev)
    tr = MIGTTrainer(model)
    g = np.random.Generator(np.random.PCG64(rank))
    tokens = torch.from_numpy(g.integers(0, 1024, size=(B, S, 8, 8))).to(dev)
    _, cams = synthetic_scene_batch(B, S, 8, seed=rank)
    poses = geometry.normalize_cameras(geometry.to_relative_cameras(torch.from_numpy(cams))[0]).to(dev)
    tr.grad_allreduce_dtype = args.grad_dtype
    if args.serial_wgrad:
        tr.overlap_weight_gradients = False
    tr.train_step(poses, tokens)                                       # setup (allocator, code objects), untimed
    dt, met = timed(lambda: tr.train_step(poses, tokens), args.steps, args.warmup, dev)
    scenes = sharding.sum_over_ranks(B * args.steps, dev)
    comm = None
    if world > 1:
        # what the collective costs: the same step without it, and the compute stream's wait for the per-layer all-reduces after the
        # last backward kernel (the part the overlap with the backward pass did not hide), averaged over the timed step count
        dt_off, _ = timed(lambda: tr.train_step(poses, tokens, reduce_gradients=False), args.steps, 1, dev)
        tr.time_allreduce = True
        waits = []
        for _ in range(args.steps):
            tr.train_step(poses, tokens)
            waits.append(tr.exposed_allreduce_ms())
        tr.time_allreduce = False
        exposed = sharding.max_over_ranks(sum(waits) / len(waits), dev)
        comm = {'ms_per_step_without_allreduce': round(dt_off / args.steps * 1e3, 3), 'exposed_allreduce_wait_ms': round(exposed, 3),
                'gradient_dtype_on_the_links': args.grad_dtype, 'backend': torch.distributed.get_backend(),
                'note': 'per-layer SUM all-reduce issued as each layer\'s backward completes; exposed = compute-stream wait after the '
                        'last backward kernel (HIP events), max over ranks'}
    if rank != 0:
        return
    # GEMM-family roofline: every launch timed by HIP events on the stream it is issued to, with the weight-gradient GEMMs back on the
    # MAIN stream for this one instrumented step (tr.overlap_weight_gradients off).  In the timed steps gemm_tn_bf16 runs on a second
    # stream beside the dX GEMM: two concurrent kernels share the CUs, each one's begin-to-end time stretches, and a sum of such
    # durations counts the overlap twice (round 4's line: 20.5 "kernel ms" inside a 21.2 ms step).  Serialised, a launch's duration is its
    # own, the sum is the family's time, and it is comparable with a rocprofv3 kernel trace of `--serial-wgrad` (profiles/r5_*)
    overlap = tr.overlap_weight_gradients
    tr.overlap_weight_gradients = False
    tr.train_step(poses, tokens, reduce_gradients=False)              # rank 0 alone from here on: no collective (the other ranks have returned)
    best = None
    for _ in range(2):                                                # two instrumented steps, the smaller total kept (clock drift after the timed loop)
        prof = OpTimer()
        prof.install()
        tr.train_step(poses, tokens, reduce_gradients=False)
        res = prof.gemm_summary()
        prof.uninstall()
        if best is None or res[0] < best[0]:
            best = res
    ms, fl, n, top = best
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3):
        tr.train_step(poses, tokens, reduce_gradients=False)
    torch.cuda.synchronize()
    serial_step_ms = (time.perf_counter() - t0) / 3 * 1e3
    tr.overlap_weight_gradients = overlap
    grad_mb = sum(int(t.numel()) for t in [tr.flat_g]) * 4 / 2 ** 20
    peak = BF16_MFMA_PEAK_TFLOPS if arm == 'bf16' else BF16_MFMA_PEAK_TFLOPS / 6
    line = {'metric': 'MIGT training scenes/sec (3-stream forward, losses, backward, AdamWeightDecay), CO3D 10-cat finetune config',
            'value': round(scenes / dt, 3), 'unit': 'scenes/s', 'n_gpus': world, 'shared_device': args.dist['shared_device'], 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'bf16' if arm == 'bf16' else 'f32', 'data': 'synthetic',
            'config': {'workload': 'CO3D 10-cat training step, DP scene-batch shard + RCCL grad all-reduce (BASELINE.json configs[3])',
                       'scenes_per_gpu_per_step': B, 'views_per_scene': S, 'streams': 3, 'tokens_per_scene': 3 * S * 64,
                       'parallelism': f'dp{world}: per-replica mean loss, gradients SUMmed (migt.py:471-476,488), all-reduce per layer range '
                                      'overlapped with the backward pass', 'ranks': args.dist['ranks'], 'backend': args.dist['backend'],
                       'communicator_world_size': args.dist['communicator_world_size'], 'dist': args.dist, 'gradient_mib': round(grad_mb, 1), 'dropout': args.dropout,
                       'dropout_note': 'the reference trains with MIGTConfig.dropout = 0.1 (models/config.py:66) at four sites (migt.py:72,216,403, '
                                       'branching_attention.py:15-17); counter-based masks recomputed in the backward pass, inside the GEMM '
                                       'epilogues / LayerNorm backward / flash attention kernels of the bf16 arm',
                       'precision': ('fp32 master weights, bf16-MFMA dense GEMMs (the reference trains with --fp16)' if arm == 'bf16' else
                                     'fp32-equivalent: x3h forward GEMMs, x6 backward GEMMs, x6 / f32 attention'),
                       'weights': 'random-init MIGT 88.4M', 'loss': float(met['loss']), 'collective': comm},
            'roofline': {'bound': 'mfma', 'kernel': 'dense GEMM family of the step (gemm_bf16 / gemm_bf16_g256 / gemm_tn_bf16 or gemm_x3h / gemm_x6 launches: forward, dX, dW incl. its slab sums)',
                         'achieved': round(fl / (ms * 1e-3) / 1e12, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                         'frac': round(fl / (ms * 1e-3) / 1e12 / peak, 4),
                         # HBM bytes of the family's largest launch shape (c_fc forward and its dX twin, 900 tiles: the same algorithmic bytes) from
                         # this round's committed counter passes
                         'traffic': (lambda t, alg: None if t is None else {
                             'launch': '19 200 x 768 x 3072 with bf16 in / out (c_fc forward with its GELU-dual epilogue; mlp.c_proj dX with its GELU-backward epilogue)',
                             'bytes': t['bytes'], 'algorithmic_bytes': alg, 'over_algorithmic': round(t['bytes'] / alg, 3), 'source': t['source']})(
                             pmc_traffic('gemm_bf16_g256_kernelILb1ELb0E', (64 + 15) * 12 if _tail_on() else 900, 512, 'train')
                             if (arm == 'bf16' and (B, S) == (10, 10)) else None, 19200 * 768 * 2 + 768 * 3072 * 2 + 2 * 19200 * 3072 * 2),
                         'launches_per_step': n,
                         'kernel_ms_per_step': round(ms, 3), 'algorithmic_gflop_per_step': round(fl / 1e9, 1),
                         'timing': 'HIP events on the launch stream around every GEMM-family launch of ONE step run with the weight-gradient '
                                   'GEMMs serialised on the main stream (no concurrent kernel stretches a duration); the timed steps overlap '
                                   'them on a second stream',
                         'ms_per_step_serialised': round(serial_step_ms, 3),
                         'weight_gradient_stream_overlap_in_timed_steps': bool(overlap),
                         'top_shapes_mode_M_K_N_batch': [{'shape': list(k), 'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                                          'launches': v[2]} for k, v in top],
                         'peak_note': 'fp32-equivalent GEMMs execute 3 (x3h) or 6 (x6) 16-bit MFMA flops per fp32 flop; peak quoted = 2500 / 6'
                                      if arm != 'bf16' else 'dense bf16 MFMA peak'}}
    print(json.dumps(line), flush=True)


# ------------------------------------------------------------------------------------------------ --workload allimg (configs[4])
def run_allimg(args, rank, local, world, dev):
    import numpy as np
    from viewformer_amd import evaluate_allimg as ea
    from viewformer_amd import sharding
    from viewformer_amd.weights import synthetic_scene_batch
    S = args.views if args.views_set else 10                           # CO3D: 9 context views + target (README.md:250-264)
    F = args.batch if args.batch_set else 128                          # frames per sequence = scenes per transformer batch (:173)
    # BASELINE configs[4] names "fp8 MFMA attention"; the line reports the arm that is FASTER and tighter on MI355X, which is bf16: at 64
    # features per head the attention is bound by the softmax's vector instructions (17.6 VALU per MFMA, profiles/r2_new_kernels_pmc.txt;
    # DESIGN.md §5), not by the matrix pipe or by K / V bytes, so e4m3 operands buy no time (r3: 1880 us per launch register-staged fp8
    # vs 669 us LDS-DMA bf16) and cost accuracy (5e-2 .. 1.1e-1 of max |out| vs 5e-3).  --attention fp8 keeps the fp8 arm measurable.
    vq, tr, _ = build_models(dev, True, 'mixed', args.conv_arith, True, args.encoder_chunk, attention=args.attention or 'bf16',
                             sequence_size=S)
    frames, cams = synthetic_scene_batch(1, F, 128, seed=rank)
    fr, cm = torch.from_numpy(frames[0]).to(dev), cams[0]
    ctx = list(np.random.default_rng(42).choice(F, (S - 1,), replace=False))      # :131-132

    def step():
        return ea.evaluate_sequence(tr, vq, fr, cm, ctx)
    step()
    dt, out = timed(step, args.steps, args.warmup, dev)
    n_img = sharding.sum_over_ranks(F * S * args.steps, dev)
    if rank != 0:
        return
    prof = OpTimer()
    prof.install()
    step()
    torch.cuda.synchronize()
    att = prof.attn_entry()
    prof.uninstall()
    line = {'metric': 'generated views/sec, all-images evaluator loop (encode sequence -> multi-context transformer -> decode), 128px',
            'value': round(n_img / dt, 3), 'unit': 'generated views/s', 'n_gpus': world, 'shared_device': args.dist['shared_device'], 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'fp8 attention / bf16 dense' if tr.attention == 'fp8' else 'bf16', 'data': 'synthetic',
            'config': {'workload': f'CO3D-all 128px inference, {tr.attention} MFMA attention (the arm this line ran), large-batch decode '
                                   f'(BASELINE.json configs[4], which names fp8: --attention fp8)',
                       'attention_arm_note': 'configs[4] names fp8; reported arm = bf16 LDS-DMA attention, the faster and tighter one on MI355X '
                                             '(the kernel is softmax-VALU-bound at 64 features per head: e4m3 operands buy no time); '
                                             '--attention fp8 runs the e4m3 tolerance arm',
                       'frames_per_sequence': F, 'views_per_scene': S, 'transformer_batch': ea.TRANSFORMER_BATCH,
                       'decode_batch_scenes': ea.DECODE_BATCH, 'images_decoded_per_step': F * S, 'attention': tr.attention,
                       'parallelism': f'sequence-shard x{world}, no collective', 'ranks': args.dist['ranks'],
                       'backend': args.dist['backend'], 'dist': args.dist},
            'roofline': att or {}}
    print(json.dumps(line), flush=True)


def self_launch(args):
    """``python bench.py --gpus N`` with no launcher (the driver's command form): start the N ranks here — re-run this command line under
    ``torch.distributed.run --nnodes=1 --nproc-per-node N`` on 127.0.0.1, rank 0's JSON line passes through on stdout, the exit status is the
    launcher's.  With fewer visible GPUs than ranks the run is REFUSED (non-zero exit, one line on stderr) instead of measuring one GPU and
    printing it as N — unless VF_DIST_BACKEND=gloo says that ranks sharing a device is what the caller wants (the 1-GPU test boxes)."""
    if args.gpus < 1:
        raise SystemExit(f'bench.py: --gpus {args.gpus}: need at least one rank')
    if args.gpus == 1 or 'WORLD_SIZE' in os.environ or 'RANK' in os.environ:
        return                                                  # one rank, or already inside a launcher's worker
    ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
    backend = os.environ.get('VF_DIST_BACKEND')
    if ndev < args.gpus and backend != 'gloo':
        raise SystemExit(f'bench.py: --gpus {args.gpus} but {ndev} GPU(s) visible: refusing to run {args.gpus} ranks (RCCL needs one device per '
                         f'rank; VF_DIST_BACKEND=gloo lets ranks share a device for plumbing tests only)')
    import subprocess
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))   # dmabuf IPC for RCCL
    # --standalone: the launcher's own c10d rendezvous picks a free port itself (no bind-close-reuse race under parallel runs), on 127.0.0.1
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--standalone', '--local-addr', '127.0.0.1', '--nnodes=1',
           f'--nproc-per-node={args.gpus}', os.path.abspath(__file__)] + sys.argv[1:]
    sys.stderr.write(f'bench.py: no launcher in the environment, starting {args.gpus} ranks: {" ".join(cmd[1:8])} ...\n')
    sys.stderr.flush()
    raise SystemExit(subprocess.call(cmd, env=env))


def dist_info(world):
    """what actually ran, for the line's ``config``: ranks, transport, the communicator's OWN world size (queried from the process group, not
    copied from the flags) and whether ranks shared a device (the gloo plumbing configuration of the 1-GPU test boxes)"""
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    if world > 1 and dist.is_initialized():
        return {'ranks': world, 'communicator_world_size': dist.get_world_size(), 'backend': dist.get_backend(),
                'launcher': 'torch.distributed.run, one process per rank', 'devices_visible': ndev, 'shared_device': ndev < world}
    return {'ranks': 1, 'communicator_world_size': 1, 'backend': None, 'launcher': 'single process', 'devices_visible': ndev,
            'shared_device': False}


LAYER_GRADIENT_FLOATS = 12 * 768 * 768 + 13 * 768                # one transformer layer's range of the flat gradient buffer: 28.4 MB fp32
ENCODER_LIVE_BYTES_PER_IMAGE = 50.3e6                            # 896 images = one encoder chunk ~ 45 GB of live fp32 activations


def preflight(args, rank, local, world, dev, iters=None):
    """Multi-GPU pre-flight — everything a first run on an N-GPU node depends on that a 1-GPU box never exercised, in a few
    seconds and BEFORE the timed run: this rank's device binding, the communicator (RCCL unless VF_DIST_BACKEND=gloo; created here also at world
    1), ``iters`` SUM all-reduces of one layer's gradient range (28 MB fp32 — the training step's collective unit, train.py) timed with HIP
    events, and the HBM headroom for one encoder chunk.  Returns the dict rank 0 prints (one JSON line); fields: DESIGN §8."""
    import torch.distributed as dist
    made_group = False
    if not dist.is_initialized():                                 # world 1: still build a communicator — RCCL must load and reduce on this box
        import tempfile
        backend = os.environ.get('VF_DIST_BACKEND') or 'nccl'
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        store = tempfile.NamedTemporaryFile(prefix='vf_preflight_', delete=False)
        store.close()
        os.unlink(store.name)
        dist.init_process_group(backend, init_method=f'file://{store.name}', rank=0, world_size=1)
        made_group = True
    backend = dist.get_backend()
    n = dist.get_world_size()
    iters = iters or (100 if backend == 'nccl' else 10)
    buf = torch.ones(LAYER_GRADIENT_FLOATS, dtype=torch.float32, device=dev)
    for _ in range(3):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
        buf.fill_(1.0)
    torch.cuda.synchronize()
    dist.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        dist.all_reduce(buf, op=dist.ReduceOp.SUM)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    # correctness of the reduction itself: iters SUMs of a buffer of ones = n ** iters (exact in fp32 while it stays below 2 ** 24 — checked on a
    # fresh buffer with one reduction, which every world size passes exactly)
    chk = torch.full((1024,), float(rank + 1), dtype=torch.float32, device=dev)
    dist.all_reduce(chk, op=dist.ReduceOp.SUM)
    sum_ok = bool((chk == n * (n + 1) / 2).all().item())
    free_b, total_b = torch.cuda.mem_get_info(dev)
    S = args.views or 7
    B = args.batch or (128 if S == 7 else -(-900 // S))
    chunk_images = min(B * S, args.encoder_chunk)
    need = chunk_images * ENCODER_LIVE_BYTES_PER_IMAGE + 1.3e9          # + replicated weights and packed copies
    mine = {'rank': rank, 'device': torch.cuda.current_device(), 'device_name': torch.cuda.get_device_name(dev),
            'hbm_free_gb': round(free_b / 1e9, 1), 'hbm_total_gb': round(total_b / 1e9, 1),
            'hbm_headroom_gb_after_encoder_chunk': round((free_b - need) / 1e9, 1)}
    allr = [None] * n
    dist.all_gather_object(allr, mine)
    t = torch.tensor([ms], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    worst_ms = float(t.item())
    nbytes = LAYER_GRADIENT_FLOATS * 4
    algbw = nbytes / (worst_ms * 1e-3) / 1e9
    out = {'preflight': 'ok' if sum_ok and all(r['hbm_headroom_gb_after_encoder_chunk'] > 0 for r in allr) else 'FAILED',
           'ranks': n, 'backend': backend, 'devices_visible': torch.cuda.device_count(), 'shared_device': torch.cuda.device_count() < n,
           'allreduce': {'bytes': nbytes, 'iters': iters, 'ms_max_over_ranks': round(worst_ms, 4), 'algbw_gbs': round(algbw, 2),
                         'busbw_gbs': round(algbw * 2 * (n - 1) / n, 2) if n > 1 else None, 'sum_exact': sum_ok,
                         'note': 'one transformer layer range of the gradient buffer, SUM, fp32; busbw = algbw x 2(n-1)/n (ring accounting); '
                                 'DESIGN §8: the per-layer all-reduce hides behind the next layer\'s backward above ~50 GB/s busbw'},
           'encoder_chunk': {'images': chunk_images, 'estimated_live_gb': round(need / 1e9, 1)},
           'per_rank': allr}
    if made_group:
        dist.destroy_process_group()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', choices=['views', 'train', 'allimg'], default='views')
    ap.add_argument('--batch', type=int, default=None,
                    help='scenes per GPU per step (views: default 128 — measured 602 / 628 / 641 views/s at 32 / 64 / 128 in round 1; '
                         'train: 10; allimg: frames per sequence, 128)')
    ap.add_argument('--views', type=int, default=None, help='views per scene (views: 7 = 6 context + 1 novel; train / allimg: 10)')
    ap.add_argument('--no-localization', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-f32-arm', action='store_true', help='skip the short timing of the all-fp32 parity arm reported beside the mixed arm')
    ap.add_argument('--precision', choices=['f32', 'mixed'], default='mixed',
                    help="mixed (default; BASELINE configs[1] names bf16): exact-fp32 encoder + codebook lookup (token indices "
                         "bit-exact) with the transformer's dense layers and the decoder's convolutions on bf16 MFMA, fp32 "
                         "accumulate (logits / pixels within the tolerances stated in tests/test_hip_bf16.py); "
                         "f32: everything exact fp32 (full fp32 parity arm)")
    ap.add_argument('--conv-arith', choices=['x3h', 'x6', 'f32'], default='x3h',
                    help="how the fp32 3x3 convolutions are evaluated: x3h = fp32-equivalent three-term split-fp16 products (low piece "
                         "carried at 2^11, cross terms in their own accumulator; error vs fp64 <= the f32 MFMA for activations in "
                         "fp16's range, tests/test_hip_x3h.py) for the stride-1 / upsample convs, x6 elsewhere; x6 = six-term "
                         "split-bf16 products everywhere (no range condition, tests/test_hip_x6.py); f32 = native f32 MFMA")
    ap.add_argument('--attention', choices=['bf16', 'fp8'], default=None,
                    help='mixed arm: attention operand format (default bf16 everywhere; fp8 = the OCP e4m3 tolerance arm configs[4] names — slower and looser)')
    ap.add_argument('--fp32-activations', action='store_true',
                    help='mixed arm: keep LayerNorm / GELU / attention outputs fp32 in HBM (A/B of the bf16 activation chain; same results)')
    ap.add_argument('--decoder-act16', type=int, choices=[0, 32, 64, 128], default=None,
                    help='mixed arm: bf16 activations between the decoder layers from this resolution up (0 = fp32 activations; default: VQGAN.decoder_act16)')
    ap.add_argument('--encoder-chunk', type=int, default=1024, help='images per encoder / decoder launch chunk (VQGAN max_images_per_call)')
    ap.add_argument('--cpu-scenes', type=int, default=0, help='scenes in the CPU-baseline sample (0 = auto)')
    ap.add_argument('--dropout', type=float, default=0.1,
                    help='train: dropout rate — default 0.1 = MIGTConfig.dropout of the reference (viewformer/models/config.py:66), which the CO3D '
                         'finetune command does not override (README.md:250-264): embedding, attention-weight, residual and MLP sites')
    ap.add_argument('--grad-dtype', choices=['f32', 'bf16'], default='f32',
                    help='train: dtype of the gradient buckets on the links (bf16 halves the 354 MB all-reduce; default f32 like the reference)')
    ap.add_argument('--serial-wgrad', action='store_true',
                    help='train: keep the weight-gradient GEMMs on the main stream in every step (for kernel traces whose durations are not '
                         'stretched by a concurrent kernel; the default overlaps them with the dX GEMMs on a second stream)')
    ap.add_argument('--preflight', action='store_true',
                    help='multi-GPU pre-flight only: bind devices, create the communicator, time 100 all-reduces of one layer\'s 28 MB gradient '
                         'range, report bus GB/s and per-rank HBM headroom for one encoder chunk, print ONE JSON line and exit (with N > 1 ranks the '
                         'same line goes to stderr before every timed run)')
    ap.add_argument('--batch-sweep', default=None,
                    help='views workload: comma-separated scenes-per-step list (SURVEY 8d: 1,8,64,256,1024); prints ONE JSON line with the '
                         'views/s of every batch size (same models, inputs resident in HBM, --steps timed steps each)')
    args = ap.parse_args()
    args.batch_set, args.views_set = args.batch is not None, args.views is not None

    from viewformer_amd import sharding
    from viewformer_amd.evaluate import generate_batch_predictions
    from viewformer_amd.weights import synthetic_scene_batch

    self_launch(args)                  # bare `python bench.py --gpus N`: N ranks under torch.distributed.run, or a loud refusal
    rank, local, world = sharding.init_from_env()
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: the launcher and the flag disagree, refusing to print a line '
                         f'whose n_gpus is not the number of ranks that ran')
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU fallback for the hot path)'
    args.dist = dist_info(world)
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if args.preflight or world > 1:
        try:
            pf = preflight(args, rank, local, world, dev)
        except Exception as e:      # noqa: BLE001  (inside a timed run the pre-flight is a courtesy: it must never cost the line; --preflight alone fails loudly)
            if args.preflight:
                raise
            pf = {'preflight': 'ERROR', 'error': repr(e)[:400]}
        if rank == 0:
            (sys.stdout if args.preflight else sys.stderr).write(json.dumps(pf) + '\n')
            (sys.stdout if args.preflight else sys.stderr).flush()
        if args.preflight:
            sharding.barrier()
            if world > 1:
                torch.distributed.destroy_process_group()
            if pf['preflight'] != 'ok':
                raise SystemExit(1)
            return
    if args.workload != 'views':
        (run_train if args.workload == 'train' else run_allimg)(args, rank, local, world, dev)
        sharding.barrier()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    localization = not args.no_localization
    S = args.views or 7
    # scenes per step: 128 x 7 = 896 images at configs[1]; for other view counts the same saturating image count (>= 900 images: 45 scenes
    # x 20 views at configs[2] — the round-4 line ran 12 scenes = 240 images, a sub-saturating batch; the sweep saturates at >= 64 x 7)
    B = args.batch or (128 if S == 7 else -(-900 // S))

    vq, tr, models_cfg = build_models(dev, localization, args.precision, args.conv_arith, not args.fp32_activations, args.encoder_chunk,
                                      attention=args.attention, decoder_act16=args.decoder_act16)
    if args.batch_sweep:
        sweep = []
        for b in [int(x) for x in args.batch_sweep.split(',')]:
            fr, cm = synthetic_scene_batch(min(b, 256), S, 128, seed=rank)     # (beyond 256 scenes the batch repeats: host-side generation
            fr_d, cm_d = torch.from_numpy(fr).to(dev), torch.from_numpy(cm).to(dev)   # of 7168 frames takes longer than the measurement)
            if b > 256:
                fr_d, cm_d = fr_d.repeat(-(-b // 256), 1, 1, 1, 1)[:b].contiguous(), cm_d.repeat(-(-b // 256), 1, 1)[:b].contiguous()

            def step_b():
                return generate_batch_predictions(tr, vq, fr_d, cm_d)
            step_b()
            torch.cuda.synchronize()
            dt_b, _ = timed(step_b, args.steps, args.warmup, dev)
            n_b = sharding.sum_over_ranks(b * args.steps, dev)
            sweep.append({'scenes_per_gpu_per_step': b, 'images_encoded_per_step': b * S, 'value': round(n_b / dt_b, 2),
                          'ms_per_step': round(dt_b / args.steps * 1e3, 3)})
            del fr_d, cm_d
            torch.cuda.empty_cache()
        if rank == 0:
            from viewformer_amd.evaluate import MAX_SCENES_PER_CALL
            print(json.dumps({'metric': 'novel views/sec (encode->AR transformer->decode), 128px 6-ctx', 'unit': 'novel views/s',
                              'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'higher_is_better': True, 'scaling': 'weak',
                              'dtype': 'bf16' if args.precision == 'mixed' else 'f32', 'data': 'synthetic',
                              'config': {'workload': 'batch sweep of BASELINE.json configs[1] (SURVEY 8d)', 'views_per_scene': S,
                                         'localization_pass': localization, 'max_scenes_per_transformer_call': MAX_SCENES_PER_CALL,
                                         'max_images_per_encoder_call': args.encoder_chunk},
                              'batch_sweep': sweep}), flush=True)
        sharding.barrier()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    frames, cams = synthetic_scene_batch(B, S, 128, seed=rank)             # this rank's shard of the global batch
    frames_d = torch.from_numpy(frames).to(dev)
    cams_d = torch.from_numpy(cams).to(dev)

    def step():
        return generate_batch_predictions(tr, vq, frames_d, cams_d)

    step()                             # setup, untimed and not counted as warm-up: code objects loaded, caching allocator grown to
    torch.cuda.synchronize()           # its steady-state footprint, clocks off idle (a cold first process once read 4 % low)
    dt, out = timed(step, args.steps, args.warmup, dev)
    views = sharding.sum_over_ranks(B * args.steps, dev)
    assert out['generated_images'].dtype == torch.uint8

    if rank != 0:
        sharding.barrier()
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    value = views / dt
    gf = flops_per_view(S, localization)
    line = {
        'metric': f'novel views/sec (encode->AR transformer->decode), 128px {S - 1}-ctx',
        'value': round(value, 3), 'unit': 'novel views/s', 'n_gpus': world, 'shared_device': args.dist['shared_device'], 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': ('f32' if args.precision == 'f32' else 'bf16'), 'data': 'synthetic',
        'config': {'workload': ('SM7 codebook+transformer, 6 context views -> 1 novel view, 128x128 '
                                '(BASELINE.json configs[1])' if S == 7 else
                                f'InteriorNet-style {S - 1}-view context -> 1 novel view, image + localization heads, 128x128 '
                                '(BASELINE.json configs[2])'),
                   'scenes_per_gpu_per_step': B, 'views_per_scene': S, 'localization_pass': localization,
                   'target_view_encoded': True, 'parallelism': f'scene-shard x{world}, no collective',
                   'ranks': args.dist['ranks'], 'backend': args.dist['backend'], 'dist': args.dist,
                   'io': 'inputs (uint8 frames, cameras) resident in HBM when the timed region starts; generated uint8 images stay in HBM '
                         '(see host_io for the rate with the host round trip)',
                   'precision': ('fp32 everywhere' if args.precision == 'f32' else
                                 'mixed: fp32 encoder + codebook lookup (bit-exact tokens), bf16-MFMA transformer dense '
                                 f'layers + decoder convs (fp32 accumulate; tolerances in tests/test_hip_bf16.py), {tr.attention} attention'),
                   'fp32_conv_arithmetic': {
                       'x3h': 'x3h: every fp32 product = 3 exact fp16 partial products (operands split h + l*2^-11 with l carried at '
                              '2^11, cross terms in their own fp32 accumulator, power-of-two pre-scaled weights) on the fp16 MFMA '
                              'pipe; error vs fp64 <= native f32 MFMA (tests/test_hip_x3h.py); stride-1, stride-2, upsample and 1x1 convolutions and the '
                              'AttnBlock core all run it (quant_conv too); x6 only for shapes it does not tile',
                       'x6': 'x6: every fp32 product = 6 exact bf16 partial products (operands split h+m+l) accumulated in fp32 on '
                             'the bf16 MFMA pipe; error vs fp64 <= native f32 MFMA (tests/test_hip_x6.py)',
                       'f32': 'native f32 MFMA'}[args.conv_arith],
                   'weights': 'random-init (deterministic generator), full-size VQGAN 67.9M + MIGT 88.4M',
                   'algorithmic_gflop_per_view': round(gf, 1),
                   'reference_equivalent_tflops': round(value * gf / 1e3, 2),
                   'reference_equivalent_note': 'views/s x the FLOPs of the REFERENCE loop per view (two transformer passes); the fused twin-view '
                                                'pass executes 8/14 of the transformer part, so this is equivalent work, not executed FLOPs'},
    }
    # ---- host round trip (the reference moves frames to the device and images back every batch, evaluate_transformer.py:18-19,222)
    from viewformer_amd.evaluate import stream_batch_predictions
    fr_h = torch.from_numpy(frames).pin_memory()
    cm_h = torch.from_numpy(cams).pin_memory()
    n_io = max(args.steps, 4)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for o in stream_batch_predictions(tr, vq, ((fr_h, cm_h) for _ in range(n_io))):
        img_h = o['generated_images']                  # host tensors (pinned ring)
    host_dt = (time.perf_counter() - t0) / n_io
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n_io):                              # the resident loop again, right beside it (clocks drift between the phases of a run)
        step()
    torch.cuda.synchronize()
    resident_dt = (time.perf_counter() - t0) / n_io
    t0 = time.perf_counter()
    for _ in range(2):
        o = generate_batch_predictions(tr, vq, fr_h.to(dev, non_blocking=True), cm_h.to(dev, non_blocking=True))
        img_s, cam_s = o['generated_images'].cpu(), o['generated_cameras'].cpu()
    serial_dt = (time.perf_counter() - t0) / 2
    line['host_io'] = {'value': round(B / host_dt, 2), 'unit': 'novel views/s', 'steps': n_io,
                       'serial_value': round(B / serial_dt, 2), 'resident_value_same_phase': round(B / resident_dt, 2),
                       'note': f'the evaluator\'s outer loop with host tensors in and out (evaluate.stream_batch_predictions): {fr_h.numel() >> 20} MiB of frames up '
                               f'and {img_h.numel() >> 20} MiB of generated images down per step on a second HIP stream, overlapped with the neighbouring '
                               'steps\' kernels; serial_value = the same copies issued in line with the step (the reference\'s loop); never the headline value'}
    del img_s, cam_s
    # ---- roofline section (rank 0): HIP events on the launch stream around every GEMM-family launch, the lookup and the attention
    # two instrumented passes, the one with the smaller total kept: the first launches after the timed loop occasionally run
    # at a lower clock (seen once: 5.5 ms instead of 3.7 ms for the dominant launch while the rocprofv3 trace said 3.75 ms)
    best, prof_best = None, None
    for _ in range(2):
        prof = OpTimer()
        prof.install()
        step()
        res = prof.gemm_summary()
        prof.uninstall()
        if best is None or res[0] < best[0]:
            best, prof_best = res, prof
    ms, fl, n, top = best
    fam = fl / (ms * 1e-3) / 1e12
    # dominant kernel = the halo-tile 3x3 conv at its dominant launch shape: 128->128 @128x128 (mode 1,
    # M = images*128*128): SURVEY §8(d) per-unit figure 2*9*128*128 FLOP per output pixel x M pixels.
    dom_key, dom = max(((k, v) for k, v in top if k[0] == 1), key=lambda kv: kv[1][0], default=(None, None))
    if dom is None:
        dom_key, dom = top[0]
    d_ms = dom[0] / dom[2]                               # average launch duration (HIP events, launch stream)
    d_fl = dom[1] / dom[2]                               # algorithmic (fp32 conv) FLOP per launch
    ach = d_fl / (d_ms * 1e-3) / 1e12
    x6 = args.conv_arith in ('x6', 'x3h')
    nprod = {'x3h': 3, 'x6': 6, 'f32': 1}[args.conv_arith]
    from viewformer_amd import _lib as _vflib
    k32 = bool(_vflib.load().vf_selected(_vflib.SEL_CONV_X3H_K32))
    is_dom_shape = dom_key[0] == 1 and tuple(dom_key[2:4]) == (128, 128)
    # algorithmic bytes of a launch: x read once + out written once (+ the residual read once: ResnetBlock's conv2 — half of this shape's launches)
    alg_nores, alg_res = dom_key[1] * (dom_key[2] + dom_key[3]) * 4, dom_key[1] * (dom_key[2] + 2 * dom_key[3]) * 4
    alg_bytes, pmc, pmc_detail = alg_res, None, None
    if is_dom_shape and args.conv_arith == 'x3h' and k32:
        wgs = dom_key[1] // 128 * (dom_key[3] // 128)
        # (one kernel instantiation serves conv1 — no residual — and conv2 of a ResnetBlock: the PMC mean is over both kinds, and so is the
        # algorithmic figure it is compared with: half of this shape's launches read a residual)
        pmc = pmc_traffic('conv3_halo_x3h16_kernelILb1ELb1E', wgs)
        alg_bytes = (alg_res + alg_nores) // 2
        if pmc:
            pmc_detail = {'read': pmc['read_bytes'], 'written': pmc['written_bytes'], 'dispatches': pmc['dispatches'],
                          'algorithmic_bytes_with_residual': alg_res, 'algorithmic_bytes_without_residual': alg_nores}
    # x6 executes 6 bf16 MFMA flops per algorithmic fp32 flop, so its ceiling in algorithmic terms is bf16_peak / 6
    peak = BF16_MFMA_PEAK_TFLOPS / nprod if x6 else F32_MFMA_PEAK_TFLOPS      # dense f16 peak == dense bf16 peak (2.5 PF)
    line['roofline'] = {'bound': 'mfma',
                        'kernel': (('conv3_halo_x3h16_kernel<GN+swish> (3x3 conv 128->128 @128x128, 3x v_mfma_f32_16x16x32_f16 per fp32 product)'
                                    if k32 else 'conv3_halo_x3h_kernel<s1,GN+swish> (3x3 conv 128->128 @128x128, 3x v_mfma_f32_32x32x16_f16 per fp32 '
                                    'product; vf_select(VF_SEL_CONV_X3H_K32, 0))') if nprod == 3 else
                                   'conv3_halo_x6_kernel<s1,GN+swish> (3x3 conv 128->128 @128x128, 6x v_mfma_f32_32x32x16_bf16 '
                                   'per fp32 product)' if x6 else
                                   'conv3_halo_kernel<s1,GN+swish> (3x3 conv 128->128 @128x128, v_mfma_f32_32x32x2_f32)'),
                        'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                        'frac': round(ach / peak, 4),
                        'peak_note': (f'algorithmic fp32 TFLOP/s; peak = dense 16-bit MFMA peak 2500 / {nprod} partial products. Executed '
                                      f'16-bit rate {round(nprod * ach, 1)} TFLOP/s; the native f32 MFMA peak is {F32_MFMA_PEAK_TFLOPS}.  On full-entropy '
                                      f'operands the matrix pipe alone draws the package power limit at {F16_MFMA_POWER_LIMITED_TFLOPS} TFLOP/s (f16; '
                                      f'bf16 {BF16_MFMA_POWER_LIMITED_TFLOPS}; profiles/r4_power_ceiling_probe.txt): the executed rate is '
                                      f'{round(nprod * ach / (F16_MFMA_POWER_LIMITED_TFLOPS if nprod == 3 else BF16_MFMA_POWER_LIMITED_TFLOPS), 3)} of that'
                                      if x6 else 'dense f32 MFMA peak'),
                        'traffic': (pmc['bytes'] if pmc else None),
                        'traffic_unit': 'bytes/launch',
                        'traffic_over_algorithmic_bytes': (round(pmc['bytes'] / alg_bytes, 4) if pmc else None),
                        'traffic_by_launch_kind': pmc_detail,
                        'traffic_source': (pmc['source'] if pmc else 'no committed PMC pass for this launch shape'),
                        'launch_shape_mode_M_Cin_Cout_batch': list(dom_key), 'avg_launch_ms': round(d_ms, 4),
                        'algorithmic_gflop_per_launch': round(d_fl / 1e9, 1),
                        'algorithmic_bytes_per_launch': alg_bytes,
                        'family': {'kernels': 'every conv/dense launch of the step (conv3_halo_x3h/_x6/_bf16/_f32, igemm_f32, gemm_bf16, gemm_x6)',
                                   'achieved': round(fam, 2),
                                   'launches_per_step': n, 'kernel_ms_per_step': round(ms, 3),
                                   'algorithmic_gflop_per_step': round(fl / 1e9, 1)},
                        'top_shapes_mode_M_Cin_Cout_batch': [
                            {'shape': list(k), 'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                             'launches': v[2]} for k, v in top],
                        # the two kernels the north star names beside the dominant one, measured in the same instrumented step
                        'codebook_lookup': prof_best.vq_entry(), 'attention': prof_best.attn_entry()}
    # ---- the all-fp32 parity arm, timed briefly beside the headline (mixed) arm
    if args.precision == 'mixed' and not args.no_f32_arm and world == 1:
        del out, o
        vq32, tr32, _ = build_models(dev, localization, 'f32', args.conv_arith, True, args.encoder_chunk)

        def step32():
            return generate_batch_predictions(tr32, vq32, frames_d, cams_d)
        step32()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            step32()
        torch.cuda.synchronize()
        dt32 = (time.perf_counter() - t0) / 2
        line['f32_arm'] = {'value': round(B / dt32, 2), 'unit': 'novel views/s', 'ms_per_step': round(dt32 * 1e3, 2), 'steps': 2,
                           'note': 'everything fp32-equivalent (x3h / x6 convolutions and dense layers, x6 attention): the full-parity arm, '
                                   'same workload, this GPU only'}
        del vq32, tr32
    if world == 1 and not args.no_cpu_baseline:
        n_cpu = args.cpu_scenes or 8
        line['cpu_baseline'] = cpu_baseline(models_cfg, S, n_cpu)
    print(json.dumps(line), flush=True)
    sharding.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
