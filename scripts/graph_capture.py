"""Stage F (features of one batch, geometry given) and stage R (registration + information matrices of one batch) captured
as HIP graphs (shapes are static per batch size) and replayed, next to the eager enqueue of the same work: what the
launch path costs.  Prints ms per stage for eager / graph and whether the replayed results equal the eager ones."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import ops, synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural

dev = torch.device("cuda:0")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
F = int(sys.argv[1]) if len(sys.argv) > 1 else 64  # 1 or 2: the latency-mode shapes, where launch gaps dominate
pts, pad = synthetic.frames(F, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
pre = hot.encoder.presample(pts, pad)
pairs, index = hot._ring_pairs(F, dev)
if F == 2:
    pairs, index = [(0, 1)], (torch.tensor([0], dtype=torch.int32, device=dev), torch.tensor([1], dtype=torch.int32, device=dev))
desc = hot.extract(pts, pad, presampled=pre)
_, table = hot.register(desc, pcd, pairs, materialize=False, pair_index=index)
torch.cuda.synchronize()


def tm(fn, n=20):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(n):
        fn()
    host = (time.perf_counter() - t) / n * 1e3
    torch.cuda.synchronize()
    return host, (time.perf_counter() - t) / n * 1e3


def try_graph(name, fn):
    eager_host, eager = tm(fn)
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream(device=dev)
    try:
        s.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(s):
            fn()  # warm-up on the capture stream
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            out = fn()
        torch.cuda.synchronize()
    except Exception as e:  # noqa: BLE001
        print(f"{name}: eager {eager:.3f} ms (host enqueue {eager_host:.3f} ms); capture FAILED: {type(e).__name__}: {str(e)[:200]}")
        return None
    ghost, gtime = tm(g.replay)
    print(f"{name}: eager {eager:.3f} ms (host enqueue {eager_host:.3f} ms) | graph replay {gtime:.3f} ms (host {ghost:.3f} ms)")
    return out


d2 = try_graph("stage F (kNN + grouped MLPs + GEMMs -> descriptors)", lambda: hot.extract(pts, pad, presampled=pre))
if d2 is not None:
    print("  graph output equals eager:", bool(torch.equal(d2, desc)))
t2 = try_graph("stage R (decoder + Kabsch + information matrices)", lambda: hot.register(desc, pcd, pairs, materialize=False, pair_index=index)[1])
if t2 is not None:
    print("  graph output equals eager:", bool(torch.equal(t2, table)))
