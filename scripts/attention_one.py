"""One attention shape, n launches (for rocprofv3 passes): scripts/attention_one.py [B] [n]  (256 tokens, 8 heads of 32)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
dev = torch.device("cuda")
qkv = torch.randn(B * 256, 768, device=dev)
out = torch.empty(B * 256, 256, device=dev)
for _ in range(n):
    ops.attention(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], B, 256, 256, out=out, kv_shift=B // 2)
torch.cuda.synchronize()
