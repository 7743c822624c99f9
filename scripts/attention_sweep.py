"""Attention kernel time against the number of sequences (256 tokens, 8 heads of 32): does the grid's fit into whole
rounds of resident workgroups matter?  (2 x 8 x B workgroups of 128 queries.)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops
dev = torch.device("cuda")
for B in (32, 64, 80, 96, 112, 128, 160, 192, 256):
    qkv = torch.randn(B * 256, 768, device=dev)
    q, k, v = qkv[:, :256], qkv[:, 256:512], qkv[:, 512:]
    out = torch.empty(B * 256, 256, device=dev)
    for _ in range(3):
        ops.attention(q, k, v, B, 256, 256, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(q, k, v, B, 256, 256, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    fl = B * 8 * (256 * 256 * 32 * 2 * 2)
    print(f"B={B:4d}  {2 * 8 * B:5d} workgroups  {us:7.1f} us  {fl / us / 1e6:6.1f} TFLOP/s  {us / B:6.3f} us per sequence")
