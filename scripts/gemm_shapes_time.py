"""The encoder tail's GEMM shapes alone on the chip: python scripts/gemm_shapes_time.py  (DPM_LIB / DPM_* knobs select a build / a dispatch)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops
torch.manual_seed(0)
tot = 0.0
for R, K, N in ((4096, 256, 1024), (4096, 1024, 256), (1024, 512, 2048), (1024, 2048, 512), (4096, 768, 256), (4096, 256, 256), (16384, 384, 128), (16384, 128, 128), (4096, 128, 256), (1024, 256, 512)):
    x = torch.randn(R, K, device="cuda"); W = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
    out = torch.empty(R, N, device="cuda")
    for _ in range(5):
        ops.linear(x, W, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        ops.linear(x, W, b, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    tot += us
    print(f"{R} x {K} -> {N}: {us:.1f} us  checksum {out.view(torch.int32).sum().item()}")
print(f"sum {tot:.1f} us  ({ {k: v for k, v in os.environ.items() if k.startswith('DPM_B3')} })")
