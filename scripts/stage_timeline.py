"""Per-stage durations inside the running pipeline (HIP events on each stage's own stream): how long a batch spends
in the geometry, feature and registration stages while all of them share the chip, next to the step time."""
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
ev = {"G": [], "F": [], "R": []}


def bracket(tag, fn):
    def wrapped(*a, **k):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        out = fn(*a, **k)
        e1.record()
        ev[tag].append((e0, e1))
        return out
    return wrapped


enc, dec = init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev)
hot = HotPath(enc, dec)
enc.presample = bracket("G", enc.presample)
hot.extract = bracket("F", hot.extract)
hot.register = bracket("R", hot.register)
pts, pad = synthetic.frames(64, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
for _ in range(4):
    hot.submit(pts, pad, pcd)
hot.flush(); torch.cuda.synchronize()
for v in ev.values():
    v.clear()
t = time.perf_counter()
for _ in range(20):
    hot.submit(pts, pad, pcd)
hot.flush(); torch.cuda.synchronize()
dt = (time.perf_counter() - t) / 20
print(f"step {dt * 1e3:.2f} ms")
for tag, v in ev.items():
    d = sorted(a.elapsed_time(b) for a, b in v)
    print(f"stage {tag}: median {d[len(d) // 2]:.2f} ms  min {d[0]:.2f}  max {d[-1]:.2f}  (n={len(d)})")
