#!/usr/bin/env python
"""bench.py — novel views/sec of the ViewFormer hot path on MI355X (BASELINE.json metric).

One "step" = one pass of ``generate_batch_predictions`` (evaluate_transformer.py:97-146) over one
batch of synthetic scenes already resident in HBM: uint8 frames [B,7,128,128,3] + cameras [B,7,7]
-> encode all 7 views (target included, as the reference does) -> MIGT pass with the MASK view ->
argmax -> decode -> uint8 novel view (+ the localization pass the SM7 model runs).
Workload = BASELINE.json configs[1]: SM7 codebook + transformer, 6 context views -> 1 novel view, bf16: the encoder and
the codebook lookup stay exact fp32 (bit-exact token indices), the transformer's dense layers and the decoder's
convolutions run on bf16 MFMA with fp32 accumulation (--precision f32 runs the all-fp32 parity arm).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

N>1: one process per GPU, scenes sharded (weak scaling: --batch scenes per GPU per step), weights
replicated, no data-path collective; barrier + max-over-ranks time; rank 0 prints ONE JSON line.
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
BF16_MFMA_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense bf16 MFMA peak (measured ceiling 2.1-2.2 PF, profiles/r1_split_bf16_probe.txt)
HBM_PEAK_GBS = 8000.0


X3H_PMC_KB = (532920e3, 501760e3)   # FETCH_SIZE, WRITE_SIZE of the x3h conv (profiles/r1_conv_x3h_pmc.txt), 56-image launch


def build_models(dev, localization: bool, precision: str = 'f32', conv_arith: str = 'x3h', bf16_activations: bool = True,
                 encoder_chunk: int = 1024):
    from viewformer_amd.config import VQGANConfig, MIGTConfig
    from viewformer_amd.weights import make_vqgan_weights, make_migt_weights
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.migt import MIGT
    vcfg = VQGANConfig()
    # SM7 transformer (README.md:348-360): seq 6 (+ the generated view), pose-multiplier 0.2,
    # localization schedule cosine(0,1,120000) => the localization head is on.
    mcfg = MIGTConfig(sequence_size=6, n_loss_skip=1, pose_multiplier=0.2,
                      localization_weight='cosine(0,1,120000)' if localization else '0')
    vsd = make_vqgan_weights(vcfg, seed=0, codebook_scale=0.05)
    msd = make_migt_weights(mcfg, seed=0)
    # precision 'mixed': encoder + codebook lookup exact fp32 (bit-exact tokens), transformer dense layers and decoder
    # convolutions on bf16 MFMA (tolerance-bounded logits / pixels) — the split the north star specifies
    arm = 'bf16' if precision == 'mixed' else 'f32'
    vq = VQGAN(vcfg, data_format='NHWC', decoder_precision=arm, conv_arith=conv_arith,
               max_images_per_call=encoder_chunk).load_state_dict(vsd).to(dev)
    # the transformer's fp32 dense layers follow the convolutions' arithmetic (x3h: LayerNorm / GELU / attention outputs are O(1))
    tr = MIGT(mcfg, precision=arm, dense_arith=conv_arith, bf16_activations=bf16_activations).load_state_dict(msd).to(dev)
    return vq, tr, (vcfg, vsd, mcfg, msd)


def flops_per_view(S: int, localization: bool):
    """algorithmic GFLOP per novel view, SURVEY.md §8(d)"""
    enc, dec = 34.507, 63.053
    gemm = 0.906 * S * 12
    attn = 12.58e-3 * S * (S + 1) / 2 * 12
    head = 0.101
    tr = gemm + attn + head
    return S * enc + dec + tr + ((gemm + attn) if localization else 0.0)


class IgemmProfiler:
    """HIP-event timing of every igemm launch (the dominant kernel) on the launch stream."""

    def __init__(self):
        self.records = []

    def install(self):
        from viewformer_amd import ops
        self._orig = ops.igemm
        prof = self

        def timed(x, w_packed, M, Cin, Cout, out, *a, **kw):
            mode = kw.get('mode', ops.MODE_GEMM)
            batch = kw.get('batch', 1)
            taps = 1 if mode == ops.MODE_GEMM else 9
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = prof._orig(x, w_packed, M, Cin, Cout, out, *a, **kw)
            e1.record()
            prof.records.append((e0, e1, 2.0 * M * Cin * Cout * taps * batch, (mode, M, Cin, Cout, batch)))
            return r
        ops.igemm = timed
        import viewformer_amd.vqgan as v
        import viewformer_amd.migt as m
        v.ops.igemm = timed
        m.ops.igemm = timed

    def uninstall(self):
        from viewformer_amd import ops
        ops.igemm = self._orig

    def summary(self):
        torch.cuda.synchronize()
        tot_ms, tot_fl, by = 0.0, 0.0, {}
        for e0, e1, fl, key in self.records:
            ms = e0.elapsed_time(e1)
            tot_ms += ms
            tot_fl += fl
            k = by.setdefault(key, [0.0, 0.0, 0])
            k[0] += ms; k[1] += fl; k[2] += 1
        top = sorted(by.items(), key=lambda kv: -kv[1][0])[:5]
        return tot_ms, tot_fl, len(self.records), top


def cpu_baseline(models_cfg, S, n_scenes, seed=123):
    """the oracle (torch-CPU restatement of the reference path) timed on the host cores"""
    from oracle import pipeline_oracle as po
    from viewformer_amd.weights import synthetic_scene_batch
    vcfg, vsd, mcfg, msd = models_cfg
    # torch's CPU conv/matmul stop scaling (and regress badly) past a few dozen threads at these sizes:
    # a 256-thread run of this sample took 14 s/scene on the GPU box's host, so cap the pool.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    frames, cams = synthetic_scene_batch(max(n_scenes, 1), S, 128, seed=seed)
    t0 = time.time()
    po.generate_batch_predictions(msd, mcfg, vsd, vcfg, frames[:1], cams[:1])      # warm (also sizes the sample)
    warm = time.time() - t0
    n = max(1, min(n_scenes, int(20.0 / max(warm, 1e-3))))                         # bound the sample to ~20 s
    t0 = time.time()
    po.generate_batch_predictions(msd, mcfg, vsd, vcfg, frames[:n], cams[:n])
    dt = time.time() - t0
    return dict(value=round(n / dt, 4), unit='novel views/s', cores=cores, kind='port',
                sample=f'{n} scenes x {S} views (same synthetic workload), fp32 torch-CPU restatement of the '
                       f'reference path (oracle/), {cores} threads, {dt:.1f} s')


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=4)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--batch', type=int, default=128,
                    help='scenes per GPU per step (measured 602 / 628 / 641 views/s at 32 / 64 / 128; the encoder runs in chunks of 256 images)')
    ap.add_argument('--views', type=int, default=7, help='views per scene (6 context + 1 novel)')
    ap.add_argument('--no-localization', action='store_true')
    ap.add_argument('--no-cpu-baseline', action='store_true')
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
    ap.add_argument('--fp32-activations', action='store_true',
                    help='mixed arm: keep LayerNorm / GELU / attention outputs fp32 in HBM (A/B of the bf16 activation chain; same results)')
    ap.add_argument('--encoder-chunk', type=int, default=1024, help='images per encoder / decoder launch chunk (VQGAN max_images_per_call)')
    ap.add_argument('--cpu-scenes', type=int, default=0, help='scenes in the CPU-baseline sample (0 = auto)')
    args = ap.parse_args()

    from viewformer_amd import sharding
    from viewformer_amd.evaluate import generate_batch_predictions
    from viewformer_amd.weights import synthetic_scene_batch

    rank, local, world = sharding.init_from_env()
    if world != args.gpus and world > 1:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    assert torch.cuda.is_available(), 'bench.py needs the MI355X (no CPU fallback for the hot path)'
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    localization = not args.no_localization
    S, B = args.views, args.batch

    vq, tr, models_cfg = build_models(dev, localization, args.precision, args.conv_arith, not args.fp32_activations, args.encoder_chunk)
    frames, cams = synthetic_scene_batch(B, S, 128, seed=rank)             # this rank's shard of the global batch
    frames_d = torch.from_numpy(frames).to(dev)
    cams_d = torch.from_numpy(cams).to(dev)

    def step():
        return generate_batch_predictions(tr, vq, frames_d, cams_d)

    step()                             # setup, untimed and not counted as warm-up: code objects loaded, caching allocator grown to
    torch.cuda.synchronize()           # its steady-state footprint, clocks off idle (a cold first process once read 4 % low)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    sharding.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    dt = sharding.max_over_ranks(dt, dev)
    views = sharding.sum_over_ranks(B * args.steps, dev)
    assert out['generated_images'].dtype == torch.uint8

    line = None
    if rank == 0:
        value = views / dt
        gf = flops_per_view(S, localization)
        line = {
            'metric': f'novel views/sec (encode->AR transformer->decode), 128px {S - 1}-ctx',
            'value': round(value, 3), 'unit': 'novel views/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': round(dt / args.steps * 1e3, 3), 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'dtype': ('f32' if args.precision == 'f32' else 'bf16'), 'data': 'synthetic',
            'config': {'workload': ('SM7 codebook+transformer, 6 context views -> 1 novel view, 128x128 '
                                    '(BASELINE.json configs[1])' if S == 7 else
                                    f'InteriorNet-style {S - 1}-view context -> 1 novel view, image + localization heads, 128x128 '
                                    '(BASELINE.json configs[2])'),
                       'scenes_per_gpu_per_step': B, 'views_per_scene': S, 'localization_pass': localization,
                       'target_view_encoded': True, 'parallelism': f'scene-shard x{world}, no collective',
                       'precision': ('fp32 everywhere' if args.precision == 'f32' else
                                     'mixed: fp32 encoder + codebook lookup (bit-exact tokens), bf16-MFMA transformer dense '
                                     'layers + decoder convs (fp32 accumulate; tolerances in tests/test_hip_bf16.py)'),
                       'fp32_conv_arithmetic': {
                           'x3h': 'x3h: every fp32 product = 3 exact fp16 partial products (operands split h + l*2^-11 with l carried at '
                                  '2^11, cross terms in their own fp32 accumulator, power-of-two pre-scaled weights) on the fp16 MFMA '
                                  'pipe; error vs fp64 <= native f32 MFMA (tests/test_hip_x3h.py); stride-2 and 1x1 convs: x6',
                           'x6': 'x6: every fp32 product = 6 exact bf16 partial products (operands split h+m+l) accumulated in fp32 on '
                                 'the bf16 MFMA pipe; error vs fp64 <= native f32 MFMA (tests/test_hip_x6.py)',
                           'f32': 'native f32 MFMA'}[args.conv_arith],
                       'weights': 'random-init (deterministic generator), full-size VQGAN 67.9M + MIGT 88.4M',
                       'algorithmic_gflop_per_view': round(gf, 1),
                       'whole_path_tflops': round(value * gf / 1e3, 2)},
        }
    # ---- roofline of the dominant kernel (igemm_f32, all conv + dense layers), rank 0 only -----------
    if rank == 0:
        # two instrumented passes, the one with the smaller total kept: the first launches after the timed loop occasionally run
        # at a lower clock (seen once: 5.5 ms instead of 3.7 ms for the dominant launch while the rocprofv3 trace of the same
        # box said 3.75 ms)
        best = None
        for _ in range(2):
            prof = IgemmProfiler()
            prof.install()
            step()
            res = prof.summary()
            prof.uninstall()
            if best is None or res[0] < best[0]:
                best = res
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
        # HBM bytes per launch from the PMC passes (profiles/r1_conv_x6_pmc.txt / r1_conv_halo_pmc.txt: 2*FETCH_SIZE +
        # WRITE_SIZE with the gfx950 unit correction), measured on a 56-image launch of this shape, scaled by pixels
        fetch_kb, write_kb = {'x3h': X3H_PMC_KB, 'x6': (516830e3, 458750e3), 'f32': (497520e3, 458750e3)}[args.conv_arith]
        is_dom_shape = fetch_kb is not None and dom_key[0] == 1 and tuple(dom_key[2:4]) == (128, 128)
        pmc_bytes_per_pixel = (2 * fetch_kb + write_kb) * 1.024 / (56 * 128 * 128) if is_dom_shape else None
        # x6 executes 6 bf16 MFMA flops per algorithmic fp32 flop, so its ceiling in algorithmic terms is bf16_peak / 6
        peak = BF16_MFMA_PEAK_TFLOPS / nprod if x6 else F32_MFMA_PEAK_TFLOPS      # dense f16 peak == dense bf16 peak (2.5 PF)
        line['roofline'] = {'bound': 'mfma',
                            'kernel': ('conv3_halo_x3h_kernel<s1,GN+swish> (3x3 conv 128->128 @128x128, 3x v_mfma_f32_32x32x16_f16 '
                                       'per fp32 product)' if nprod == 3 else
                                       'conv3_halo_x6_kernel<s1,GN+swish> (3x3 conv 128->128 @128x128, 6x v_mfma_f32_32x32x16_bf16 '
                                       'per fp32 product)' if x6 else
                                       'conv3_halo_kernel<s1,GN+swish> (3x3 conv 128->128 @128x128, v_mfma_f32_32x32x2_f32)'),
                            'achieved': round(ach, 2), 'peak': round(peak, 1), 'unit': 'TFLOP/s',
                            'frac': round(ach / peak, 4),
                            'peak_note': (f'algorithmic fp32 TFLOP/s; peak = dense 16-bit MFMA peak 2500 / {nprod} partial products. Executed '
                                          f'16-bit rate {round(nprod * ach, 1)} TFLOP/s; the native f32 MFMA peak is {F32_MFMA_PEAK_TFLOPS}'
                                          if x6 else 'dense f32 MFMA peak'),
                            'traffic': (round(pmc_bytes_per_pixel * dom_key[1]) if pmc_bytes_per_pixel else None),
                            'traffic_unit': 'bytes/launch (PMC, scaled from the 56-image profile)',
                            'launch_shape_mode_M_Cin_Cout_batch': list(dom_key), 'avg_launch_ms': round(d_ms, 4),
                            'algorithmic_gflop_per_launch': round(d_fl / 1e9, 1),
                            'algorithmic_bytes_per_launch': dom_key[1] * (dom_key[2] + 2 * dom_key[3]) * 4,
                            'family': {'kernels': 'every conv/dense launch of the step (conv3_halo_x6/_bf16/_f32, igemm_f32, gemm_bf16)',
                                       'achieved': round(fam, 2),
                                       'launches_per_step': n, 'kernel_ms_per_step': round(ms, 3),
                                       'algorithmic_gflop_per_step': round(fl / 1e9, 1)},
                            'top_shapes_mode_M_Cin_Cout_batch': [
                                {'shape': list(k), 'ms': round(v[0], 3), 'tflops': round(v[1] / (v[0] * 1e-3) / 1e12, 1),
                                 'launches': v[2]} for k, v in top]}
        if world == 1 and not args.no_cpu_baseline:
            n_cpu = args.cpu_scenes or 8
            line['cpu_baseline'] = cpu_baseline(models_cfg, S, n_cpu)
        print(json.dumps(line), flush=True)
    sharding.barrier()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
