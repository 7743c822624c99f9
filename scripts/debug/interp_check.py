"""debug: three_interp_cat against the oracle's formula with coarse points that ARE fine points (nested FPS picks), the
distance taken from torch.matmul on this host and from the explicit fma chain."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import ops, synthetic
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
dev = "cuda:0"
pts, pad = synthetic.frames(2, 5119, start=500)
xyz = pts.transpose(1, 2).contiguous()
lens = torch.tensor([5119, 5119], dtype=torch.int32)
idx = ops.fps(xyz.to(dev), lens.to(dev), 64)[0].cpu().long()
x1 = torch.stack([xyz[b, idx[b]] for b in range(2)])          # 64 fine points
for S in (16, 64):
    x2 = x1[:, :S].contiguous()                                 # coarse = a prefix of the fine ones
    g = torch.Generator().manual_seed(1)
    f1, f2 = torch.randn(2, 64, 8, generator=g), torch.randn(2, S, 32, generator=g) * 5
    got = ops.three_interp_cat(x1.to(dev), x2.to(dev), torch.tensor([S, S], dtype=torch.int32, device=dev), f1.to(dev), f2.to(dev)).cpu()[..., 8:]
    def chain(a, b):
        a64, b64 = a.double(), b.double()
        t = (a[..., :, None, 0] * b[..., None, :, 0])
        t = (a64[..., :, None, 1] * b64[..., None, :, 1] + t.double()).float()
        t = (a64[..., :, None, 2] * b64[..., None, :, 2] + t.double()).float()
        d = -2 * t
        d += (a ** 2).sum(-1).unsqueeze(2)
        d += (b ** 2).sum(-1).unsqueeze(1)
        return d
    for name, fn in (("matmul", O.expanded_sqdist), ("fma chain", chain)):
        d, i = torch.topk(fn(x1, x2), k=3, dim=-1, largest=False)
        w = 1.0 / d.clamp(min=1e-8)
        w = w / w.sum(dim=2, keepdim=True)
        want = (f2[torch.arange(2).view(2, 1, 1), i] * w.unsqueeze(-1)).sum(dim=2)
        err = (got - want).abs().amax(-1)
        b, n = divmod(int(err.argmax()), 64)
        print(f"S {S} {name:9s}: max err {float(err.max()):.2e} at row {(b, n)}: d {d[b, n].tolist()} idx {i[b, n].tolist()}")
    dm, dc = O.expanded_sqdist(x1, x2), chain(x1, x2)
    print(f"   matmul vs chain on this host: {int((dm != dc).sum())} of {dm.numel()} differ, max {float((dm - dc).abs().max()):.2e}")
# which rows fail at S = 16, and what did the kernel pick?  (solve for the weights it must have used)
S = 16
x2 = x1[:, :S].contiguous()
f1, f2 = torch.zeros(2, 64, 4), torch.eye(S).unsqueeze(0).repeat(2, 1, 1)   # features = one-hot: the output row IS the weight vector
got = ops.three_interp_cat(x1.to(dev), x2.to(dev), torch.tensor([S, S], dtype=torch.int32, device=dev), f1.to(dev), f2.to(dev)).cpu()[..., 4:]
d, i = torch.topk(O.expanded_sqdist(x1, x2), k=3, dim=-1, largest=False)
w = 1.0 / d.clamp(min=1e-8); w = w / w.sum(dim=2, keepdim=True)
want = torch.zeros(2, 64, S).scatter_(2, i, w)
bad = ((got - want).abs().amax(-1) > 1e-5).nonzero().tolist()
print("failing rows:", bad[:20], "of", 128)
for b, n in bad[:4]:
    print(f" row {(b, n)}: kernel weights {[(j, round(float(v), 4)) for j, v in enumerate(got[b, n]) if v != 0]}  oracle {[(int(j), round(float(v), 4)) for j, v in zip(i[b, n], w[b, n])]}")
    dd = O.expanded_sqdist(x1, x2)[b, n]
    print("   all distances:", [round(float(v), 5) for v in dd])
