"""One GEMM shape, a few launches (for rocprofv3 counter passes): python scripts/gemm_one.py R K N"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops
R, K, N = (int(v) for v in sys.argv[1:4])
x = torch.randn(R, K, device='cuda'); W = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
out = torch.empty(R, N, device='cuda')
for _ in range(5): ops.linear(x, W, b, out=out)
torch.cuda.synchronize()
