"""is the inference path (encode -> MIGT -> decode, full-size models, mixed arm) bit-reproducible while ANOTHER process shares the GPU?
Two processes repeat the same 16-scene batch and compare codes, generated codes, logits and images with their first result."""
import os
import sys
import torch
import torch.multiprocessing as mp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def worker(rank, n_iter):
    from viewformer_amd.config import VQGANConfig, MIGTConfig
    from viewformer_amd.migt import MIGT
    from viewformer_amd.vqgan import VQGAN
    from viewformer_amd.weights import make_migt_weights, make_vqgan_weights, synthetic_scene_batch
    from viewformer_amd.evaluate import generate_batch_predictions
    dev = torch.device('cuda:0')
    vcfg, mcfg = VQGANConfig(), MIGTConfig()
    vq = VQGAN(vcfg, data_format='NHWC').load_state_dict(make_vqgan_weights(vcfg, seed=1, codebook_scale=0.05)).to(dev)
    tr = MIGT(mcfg, precision='bf16').load_state_dict(make_migt_weights(mcfg, seed=1, std=0.05)).to(dev)
    frames, cams = synthetic_scene_batch(16, 7, 128, seed=5 + rank)
    keys = ('codes', 'generated_codes', 'logits_last', 'generated_images', 'generated_cameras')
    ref = generate_batch_predictions(tr, vq, frames, cams, return_codes=True)
    ref = {k: ref[k].clone() for k in keys}
    bad = 0
    for it in range(n_iter):
        out = generate_batch_predictions(tr, vq, frames, cams, return_codes=True)
        d = [k for k in keys if not torch.equal(out[k], ref[k])]
        if d:
            bad += 1
            if bad <= 5:
                print(f'rank {rank} iter {it}: differs in {d}', flush=True)
    print(f'rank {rank}: {bad} of {n_iter} repeats differ', flush=True)


if __name__ == '__main__':
    n_iter = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    mp.set_start_method('spawn')
    ps = [mp.Process(target=worker, args=(r, n_iter)) for r in range(2)]
    for p in ps:
        p.start()
    for p in ps:
        p.join()
