"""How long the host needs to ENQUEUE one pipelined step (no synchronisation): the margin before the path becomes
launch-bound.  Prints ms of host time per step next to the GPU step time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural

dev = torch.device("cuda:0")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
pts, pad = synthetic.frames(64, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
for _ in range(4):
    hot.submit(pts, pad, pcd)
hot.flush(); torch.cuda.synchronize()
host = []
t0 = time.perf_counter()
for _ in range(20):
    t = time.perf_counter()
    hot.submit(pts, pad, pcd)
    host.append(time.perf_counter() - t)
hot.flush()
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 20
host.sort()
print(f"host enqueue per step: median {host[10] * 1e3:.2f} ms, min {host[0] * 1e3:.2f} ms; GPU step {wall * 1e3:.2f} ms")
# pure host cost: one submit at a time into an empty queue (no back-pressure from a full HIP queue)
pure = []
for _ in range(10):
    torch.cuda.synchronize()
    t = time.perf_counter()
    hot.submit(pts, pad, pcd)
    pure.append(time.perf_counter() - t)
hot.flush(); torch.cuda.synchronize()
pure.sort()
print(f"host cost of one submit into an empty queue: median {pure[5] * 1e3:.2f} ms, min {pure[0] * 1e3:.2f} ms")
