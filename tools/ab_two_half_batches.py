#!/usr/bin/env python
"""Does the inference step gain from running its batch as two half batches on two HIP streams (kernel tails of one half filled by the other)?
One box, alternating blocks: (a) generate_batch_predictions on 128 scenes, (b) two calls on 64 scenes each, issued to two streams.
usage: python tools/ab_two_half_batches.py [batch]"""
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import bench  # noqa: E402
from viewformer_amd.evaluate import generate_batch_predictions  # noqa: E402
from viewformer_amd.weights import synthetic_scene_batch  # noqa: E402

dev = torch.device('cuda:0')
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
vq, tr, _ = bench.build_models(dev, True, 'mixed')
frames, cams = synthetic_scene_batch(B, 7, 128, seed=0)
fr, cm = torch.from_numpy(frames).to(dev), torch.from_numpy(cams).to(dev)
h = B // 2
halves = [(fr[:h].contiguous(), cm[:h].contiguous()), (fr[h:].contiguous(), cm[h:].contiguous())]
streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
main = torch.cuda.current_stream(dev)


def whole():
    return generate_batch_predictions(tr, vq, fr, cm)


def split():
    outs = []
    for s, (f, c) in zip(streams, halves):
        s.wait_stream(main)
        with torch.cuda.stream(s):
            outs.append(generate_batch_predictions(tr, vq, f, c))
    for s in streams:
        main.wait_stream(s)
    return {k: torch.cat([o[k] for o in outs]) for k in ('generated_images', 'generated_cameras')}


ref = whole()
got = split()
torch.cuda.synchronize()
same = {k: bool(torch.equal(ref[k], got[k])) for k in got}
ms = {'whole': [], 'two_streams': []}
for r in range(6):
    for name, fn in ((('whole', whole), ('two_streams', split)) if r % 2 == 0 else (('two_streams', split), ('whole', whole))):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(6):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms[name].append(e0.elapsed_time(e1) / 6)
print(json.dumps({'batch': B, 'ms_per_batch_median': {k: round(statistics.median(v), 3) for k, v in ms.items()},
                  'all': {k: [round(x, 2) for x in v] for k, v in ms.items()}, 'same_results': same}))
