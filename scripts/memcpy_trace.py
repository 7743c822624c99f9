import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from torch.profiler import profile, ProfilerActivity
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
for _ in range(2): hot.step(pts, pad, pcd, materialize=False)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    hot.step(pts, pad, pcd, materialize=False)
    torch.cuda.synchronize()
import collections
cnt = collections.Counter()
for e in prof.events():
    n = e.name
    if 'emcpy' in n or 'copyBuffer' in n or 'emset' in n or 'fillBuffer' in n:
        cnt[n] += 1
print(cnt)
# which aten ops call hipMemcpyAsync: print stacks of cpu events named hipMemcpyAsync
seen = collections.Counter()
for e in prof.events():
    if e.name in ('hipMemcpyAsync', 'hipMemsetAsync', 'hipMemcpyWithStream'):
        st = [s for s in (e.stack or []) if 'deeppointmap_amd' in s][:2]
        seen[(e.name, tuple(st))] += 1
for k, v in seen.most_common(20): print(v, k)
