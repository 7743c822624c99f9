"""How much do the feature + registration stages slow down while first-level sampling launches run next to them?
Background: `NBG` streams each re-launching dpm_fps_ex (algo, frames per launch) back to back; foreground: the two-stream
feature + registration loop of scripts/stage_isolation.py.  Separates kernel interference from pipeline dependencies."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import ops, synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural

dev = torch.device("cuda:0")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
F = 64
pts, pad = synthetic.frames(F, 65536)
pts, pad = pts.to(dev), pad.to(dev)
pcd = (pts * 60).contiguous()
pairs = [((f - 1) % F, f) for f in range(F)]
pre = hot.encoder.presample(pts, pad)
torch.cuda.synchronize()
sb = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream(dev)


def fr_steps(n):
    prev = None
    for _ in range(n):
        desc = hot.extract(pts, pad, presampled=pre)
        ev = main.record_event()
        if prev is not None:
            d, e = prev
            with torch.cuda.stream(sb):
                sb.wait_event(e)
                hot.register(d, pcd, pairs, materialize=False)
        prev = (desc, ev)


fr_steps(3); torch.cuda.synchronize()
t = time.perf_counter(); fr_steps(20); torch.cuda.synchronize(); base = (time.perf_counter() - t) / 20
print(f"feature + registration alone: {base * 1e3:.2f} ms per step")
xyz = pts.transpose(1, 2).contiguous()
for algo, frames, nbg in ((2, 64, 2), (4, 64, 2), (4, 128, 2), (4, 192, 2), (4, 64, 1)):
    x = xyz.repeat(frames // F, 1, 1) if frames > F else xyz
    lens = torch.full((frames,), 65536, dtype=torch.int32, device=dev)
    bg = [torch.cuda.Stream(device=dev) for _ in range(nbg)]
    n_launch = 12 if algo == 2 else 6
    torch.cuda.synchronize()
    evs = []
    for s in bg:
        with torch.cuda.stream(s):
            e0 = s.record_event() if False else torch.cuda.Event(enable_timing=True); e0.record(s)
            for _ in range(n_launch):
                ops.fps(x, lens, 4096, algo=algo)
            e1 = torch.cuda.Event(enable_timing=True); e1.record(s)
            evs.append((e0, e1))
    t = time.perf_counter(); fr_steps(16); main.synchronize(); sb.synchronize(); dt = (time.perf_counter() - t) / 16
    torch.cuda.synchronize()
    per = sum(a.elapsed_time(b) for a, b in evs) / len(evs) / n_launch
    print(f"algo {algo}, {frames} frames per launch, {nbg} background stream(s): feature + registration {dt * 1e3:.2f} ms per step "
          f"(+{(dt - base) * 1e3:.2f}); a sampling launch takes {per:.2f} ms -> {per / nbg / (frames / F):.2f} ms per 64-frame batch")
