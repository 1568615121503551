"""which decoder layer breaks batch invariance? (debug helper: decode 10 images and images [7:9], compare every conv output)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from viewformer_amd.config import VQGANConfig
from viewformer_amd.vqgan import VQGAN
from viewformer_amd.weights import make_vqgan_weights
dev = torch.device('cuda:0')
cfg = VQGANConfig()
sd = make_vqgan_weights(cfg, seed=0, codebook_scale=0.05)
arith = sys.argv[1] if len(sys.argv) > 1 else 'x3h'
m = VQGAN(cfg, data_format='NHWC', conv_arith=arith).load_state_dict(sd).to(dev)
g = np.random.Generator(np.random.PCG64(5))
from viewformer_amd.weights import synthetic_scene_batch
frames, _ = synthetic_scene_batch(2, 5, 128, seed=9)
codes = m.encode(torch.from_numpy(frames.reshape(10, 128, 128, 3)).to(dev))[-1]
rec = []
for name in ('_conv3', '_conv1', '_attn', '_gn'):
    orig = getattr(m, name)
    def wrap(*a, _o=orig, _n=name, **k):
        r = _o(*a, **k)
        t = r[0] if isinstance(r, tuple) else r
        rec.append((_n, a[1] if len(a) > 1 and isinstance(a[1], str) else '', t.clone() if torch.is_tensor(t) else t))
        return r
    setattr(m, name, wrap)
rec.clear(); full = m.decode_code(codes); rec_full = list(rec)
rec.clear(); part = m.decode_code(codes[7:9]); rec_part = list(rec)
print('final equal', torch.equal(full[7:9], part), len(rec_full), len(rec_part))
for (n1, l1, a), (n2, l2, b) in zip(rec_full, rec_part):
    if not torch.is_tensor(a):
        continue
    if n1 == '_gn':
        a2, b2 = a, b
        # (mean_c, scale_c, beta): compare the mean rows of images 7, 8
        rows = a.shape[0] // 10
        eq = torch.equal(a[7 * rows:9 * rows], b)
    else:
        rows = a.shape[0] // 10
        eq = torch.equal(a[7 * rows:9 * rows], b)
    if not eq:
        d = (a[7 * rows:9 * rows] - b).abs().max().item()
        print('FIRST DIFFERENCE at', n1, l1, tuple(a.shape), 'max abs diff', d)
        break
else:
    print('all recorded layer outputs equal')
