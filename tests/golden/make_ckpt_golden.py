#!/usr/bin/env python
"""Write a tiny codebook MODEL DIRECTORY exactly as the reference produces one, FROM THE REFERENCE ITSELF:
``tests/golden/vqgan_tiny_model/{config.json, model.ckpt}``.  Run in the build container only
(needs /root/reference):   python tests/golden/make_ckpt_golden.py

config.json is ``reference_config.asdict()`` (train/utils.py:63-69); model.ckpt is a Lightning-style checkpoint whose
``state_dict`` is the reference model's own ``state_dict()`` (vqgan_th.py key names, including the loss sub-module keys a
real checkpoint carries).  The weights are the same deterministic tensors as vqgan_tiny.npz (seed 3), so the codes the
checkpoint must reproduce are already recorded there.  Only data is written."""
import json
import os
import sys

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_golden import TINY, build_reference, import_reference      # noqa: E402


def main():
    AutoModelTH, RefCfg = import_reference()
    ref, cfg, _ = build_reference(AutoModelTH, RefCfg, TINY, seed=3)
    out = os.path.join(HERE, 'vqgan_tiny_model')
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, 'config.json'), 'w') as f:
        json.dump(ref.config.asdict(), f)
    ckpt = {'epoch': 0, 'global_step': 0, 'pytorch-lightning_version': '1.2.4', 'state_dict': ref.state_dict()}
    torch.save(ckpt, os.path.join(out, 'model.ckpt'))
    print(sorted(ref.state_dict().keys())[:5], '...', len(ref.state_dict()), 'keys;',
          os.path.getsize(os.path.join(out, 'model.ckpt')), 'bytes')


if __name__ == '__main__':
    main()
