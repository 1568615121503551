import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from viewformer_amd.evaluate import generate_batch_predictions
from viewformer_amd.weights import synthetic_scene_batch
dev = torch.device('cuda:0')
B, S = int(os.environ.get('B', 128)), 7
vq, tr, _ = bench.build_models(dev, True, 'mixed', 'x3h', True, 1024)
frames, cams = synthetic_scene_batch(B, S, 128, seed=0)
fd, cd = torch.from_numpy(frames).to(dev), torch.from_numpy(cams).to(dev)
def step(): return generate_batch_predictions(tr, vq, fd, cd)
for _ in range(3): out = step()
torch.cuda.synchronize()
def timeit(fn, n=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
print('eager ms/step', timeit(step))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(2): step()
torch.cuda.current_stream().wait_stream(s)
with torch.cuda.graph(g):
    gout = step()
torch.cuda.synchronize()
print('graph ms/step', timeit(g.replay))
ref = step()
g.replay(); torch.cuda.synchronize()
print('same images', torch.equal(ref['generated_images'], gout['generated_images']), 'same cams', torch.equal(ref['generated_cameras'], gout['generated_cameras']))
