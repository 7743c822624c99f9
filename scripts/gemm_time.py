"""One GEMM shape alone on the chip, HIP-event timed: python scripts/gemm_time.py R K N [reps]  (DPM_LIB selects a build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops, _lib
R, K, N = (int(a) for a in sys.argv[1:4])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 50
torch.manual_seed(0)
x = torch.randn(R, K, device='cuda'); W = torch.randn(N, K, device='cuda') / K ** 0.5; b = torch.randn(N, device='cuda')
out = torch.empty(R, N, device='cuda')
for _ in range(5):
    ops.linear(x, W, b, out=out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.linear(x, W, b, out=out)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / reps * 1e3
print(f"{os.path.basename(_lib.LIB_PATH)}: {R} x {K} -> {N}: {us:.1f} us, {2 * R * K * N / us / 1e6:.1f} TFLOP/s fp32-equivalent, {12 * R * K * N / us / 1e6:.0f} bf16")
