"""debug: where does Rank0Consumer.consume spend its time on one gathered step (2 x 64 frames)?"""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args
from deeppointmap_amd.consumer import Rank0Consumer
from deeppointmap_amd.pipeline import HotPath

dev = torch.device("cuda:0")
torch.set_grad_enabled(False)
cfg = default_args()
from deeppointmap_amd.decoder import Decoder
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
hot = HotPath(init_procedural(Encoder(cfg)).to(dev), init_procedural(Decoder(cfg)).to(dev))
F, N = 64, 65536
descs, tabs = [], []
for b in range(2):
    pts, pad = synthetic.frames(F, N, start=b * F)
    pts, pad = pts.to(dev), pad.to(dev)
    d = hot.extract(pts, pad)
    pairs = [((f - 1) % F, f) for f in range(F)]
    _, t = hot.register(d, pts * synthetic.COOR_SCALE, pairs, materialize=False)
    descs.append(d), tabs.append(t)
gd, gt = torch.cat(descs), torch.cat(tabs)
print("gathered", tuple(gd.shape), tuple(gt.shape))
loop = len(sys.argv) > 1 and sys.argv[1] == "loop"
slam = dict(enable_loop_closure=loop)
gt = gt.clone()
truth = torch.stack([synthetic.relative_pose(g - 1, g) for g in range(gt.shape[0])]).to(gt)
gt[:, 0:9], gt[:, 9:12] = truth[:, :3, :3].reshape(-1, 9), truth[:, :3, 3]
gt[:, 12], gt[:, 16] = 0.15, 0.9
cons = Rank0Consumer(hot.decoder, dev, slam_args=slam, optimize_every=16)
cons.consume(gd, gt), cons.consume(gd, gt)
before = dict(cons.stats)
pr = cProfile.Profile()
pr.enable()
ms = [cons.consume(gd, gt) for _ in range(2)]
pr.disable()
print("ms per step", ms, {k: cons.stats[k] - before[k] for k in before})
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
