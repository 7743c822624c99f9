import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deeppointmap_amd import ops
torch.manual_seed(1)
x = torch.randn(4096, 256, device="cuda")
W = torch.randn(768, 256, device="cuda") / 16
b = torch.randn(768, device="cuda")
big = ops.linear_bf16x3(x, W, b)              # 64x64 tiles
small = ops.linear_bf16x3(x[:256].contiguous(), W, b)   # 32x32 tiles
print("rows 0..255 equal across tile variants:", torch.equal(big[:256], small), float((big[:256] - small).abs().max()))
mid = ops.linear_bf16x3(x[:1280].contiguous(), W, b)
print("1280-row call vs 4096-row call:", torch.equal(big[:1280], mid))
v = ops.linear_bf16x3(x[:256].contiguous(), W[256:], b[256:])
print("row-block view (small):", torch.equal(v, small[:, 256:]))
v2 = ops.linear_bf16x3(x, W[256:], b[256:])
print("row-block view (big):", torch.equal(v2, big[:, 256:]))
