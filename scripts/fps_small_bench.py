"""Timing of the lower farthest-point-sampling levels (register-resident kernels), 64 frames."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import ops
torch.manual_seed(0)
dev = torch.device("cuda:0")
for N, K in [(4096, 1024), (1024, 256), (256, 64), (64, 16)]:
    xyz = torch.rand(64, N, 3, device=dev)
    lengths = torch.full((64,), N, dtype=torch.int32, device=dev)
    for _ in range(2):
        out = ops.fps(xyz, lengths, K)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        out = ops.fps(xyz, lengths, K)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    print(f"N={N} K={K}: {us:.1f} us  {us / (K - 1):.3f} us/round  checksum {int(out[0].sum())}")
