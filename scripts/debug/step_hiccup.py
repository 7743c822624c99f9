"""debug: per-step host times of the pipelined bench loop, to find intermittent stalls (a 20-step run that takes 20 ms more).
Prints, per run, the host time of every submit() and the allocator's segment counters before / after."""
import gc, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural
dev = torch.device("cuda")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
pts, pad = synthetic.frames(64, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * synthetic.COOR_SCALE).contiguous()
runs = int(sys.argv[1]) if len(sys.argv) > 1 else 8
for run in range(runs):
    for _ in range(5):
        hot.submit(pts, pad, pcd)
    hot.flush(); torch.cuda.synchronize()
    st0 = torch.cuda.memory_stats()
    g0 = gc.get_count()
    ts = []
    t0 = time.perf_counter()
    for _ in range(20):
        a = time.perf_counter(); hot.submit(pts, pad, pcd); ts.append((time.perf_counter() - a) * 1e3)
    hot.flush(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) * 1e3
    st1 = torch.cuda.memory_stats()
    print(f"run {run}: {dt / 20:.3f} ms/step  host submit ms: " + " ".join(f"{x:.1f}" for x in ts) +
          f" | segments +{st1['segment.all.allocated'] - st0['segment.all.allocated']} retries +{st1['num_alloc_retries'] - st0['num_alloc_retries']}"
          f" reserved {st1['reserved_bytes.all.current'] >> 20} MiB gc {g0}->{gc.get_count()}")
