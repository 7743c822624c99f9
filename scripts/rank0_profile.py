"""Where the rank-0 consumer's time per gathered step goes (bench.py's `rank0_serial_ms`): cProfile of Rank0Consumer.consume on the rows of
one 128-frame gathered step (two ranks' worth), the same preparation as bench.py's extra (3)."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.consumer import Rank0Consumer
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.pipeline import HotPath
from deeppointmap_amd.weights import init_procedural
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = default_args()
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
F = 64
descs, tables = [], []
for r in range(2):
    pts, pad = synthetic.frames(F, 65536, start=r * F)
    d, _, t = hot.step(pts.to(dev), pad.to(dev), (pts * 60).to(dev), materialize=False)
    descs.append(d), tables.append(t)
gd, gt = torch.cat(descs), torch.cat(tables).clone()
truth = torch.stack([synthetic.relative_pose(g - 1, g) for g in range(gt.shape[0])]).to(gt)
gt[:, 0:9], gt[:, 9:12] = truth[:, :3, :3].reshape(-1, 9), truth[:, :3, 3]
gt[:, 12], gt[:, 16] = 0.15, 0.9
cons = Rank0Consumer(hot.decoder, dev, slam_args=dict(enable_loop_closure=True), optimize_every=16)
cons.consume(gd, gt), cons.consume(gd, gt)
torch.cuda.synchronize()
ms = [cons.consume(gd, gt) for _ in range(3)]
print("consume:", [round(m, 2) for m in ms], "ms per 128 gathered frames; stats", dict(cons.stats))
pr = cProfile.Profile()
pr.enable()
cons.consume(gd, gt)
pr.disable()
out = io.StringIO()
pstats.Stats(pr, stream=out).sort_stats("cumulative").print_stats(28)
print(out.getvalue()[:6000])
