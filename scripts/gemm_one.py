"""One GEMM shape in a loop (for rocprofv3 --pmc passes): python scripts/gemm_one.py R K N [reps]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops
R, K, N = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
torch.manual_seed(0)
x = torch.randn(R, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
out = torch.empty(R, N, device='cuda')
for _ in range(reps):
    ops.linear(x, W, b, out=out)
torch.cuda.synchronize()
