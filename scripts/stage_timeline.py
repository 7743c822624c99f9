"""Coarse timeline of the pipelined bench WITHOUT a profiler (rocprofv3's per-launch host cost makes the host the
bottleneck and the stages alternate): HIP events at the start and end of every stage of every batch, on the stage's
own stream.  Prints, for a few steady-state batches, when G / F / R ran (ms from a common origin) and how long the
feature and registration stages were running at the same time."""
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
hot.inputs_on_caller_stream = os.environ.get("INPUTS_RESIDENT", "0") != "1"   # INPUTS_RESIDENT=1: the geometry stage does not wait for the caller's stream
pts, pad = synthetic.frames(64, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
for _ in range(4):
    hot.submit(pts, pad, pcd)
hot.flush(); torch.cuda.synchronize()

ev = {"G": [], "F": [], "R": []}
def wrap(obj, name, key, stream_of):
    orig = getattr(obj, name)
    def f(*a, **k):
        s = stream_of()
        e0 = torch.cuda.Event(enable_timing=True); e0.record(s)
        out = orig(*a, **k)
        e1 = torch.cuda.Event(enable_timing=True); e1.record(stream_of())
        ev[key].append((e0, e1))
        return out
    setattr(obj, name, f)
cur = lambda: torch.cuda.current_stream(dev)
wrap(hot.encoder, "presample", "G", cur)           # called inside `with torch.cuda.stream(sa)`
orig_extract = hot.extract
def extract(points, padding, presampled=None):
    e0 = torch.cuda.Event(enable_timing=True); e0.record(cur())
    out = orig_extract(points, padding, presampled=presampled)
    e1 = torch.cuda.Event(enable_timing=True); e1.record(cur())
    ev["F"].append((e0, e1))
    return out
hot.extract = extract
wrap(hot, "register", "R", cur)                    # called inside `with torch.cuda.stream(sb)`
base = torch.cuda.Event(enable_timing=True)
torch.cuda.synchronize(); base.record()
host = []
t0 = time.perf_counter()
N = 24
for _ in range(N):
    t = time.perf_counter()
    hot.submit(pts, pad, pcd)
    host.append((t - t0, time.perf_counter() - t0))
hot.flush(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / N * 1e3
tl = {k: [(base.elapsed_time(a), base.elapsed_time(b)) for a, b in v] for k, v in ev.items()}
print(f"{wall:.2f} ms per step; host submit calls take {sum(b - a for a, b in host) / N * 1e3:.2f} ms each")
print("batch |   G start -> end   |   F start -> end   |   R start -> end   | host submit start -> end")
for i in range(10, 16):
    g, f, r = tl["G"][i], tl["F"][i], tl["R"][i]
    print(f"{i:5d} | {g[0]:7.2f} -> {g[1]:7.2f} | {f[0]:7.2f} -> {f[1]:7.2f} | {r[0]:7.2f} -> {r[1]:7.2f} | {host[i][0] * 1e3:7.2f} -> {host[i][1] * 1e3:7.2f}")
# overlap of F and R intervals over the steady state
def total_overlap(A, B):
    s = 0.0
    for a0, a1 in A:
        for b0, b1 in B:
            s += max(0.0, min(a1, b1) - max(a0, b0))
    return s
F, R = tl["F"][8:20], tl["R"][8:20]
print(f"per step: F busy {sum(b - a for a, b in F) / len(F):.2f} ms, R busy {sum(b - a for a, b in R) / len(R):.2f} ms, "
      f"both at once {total_overlap(F, R) / len(F):.2f} ms")
