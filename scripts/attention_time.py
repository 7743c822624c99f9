"""The attention operator alone on the chip, HIP-event timed, with its error against an fp64 softmax(Q K^T / sqrt(d)) V:
python scripts/attention_time.py  (DPM_LIB selects a build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops, _lib
dev = torch.device("cuda")
torch.manual_seed(0)
for B, M, N in ((128, 256, 256), (64, 256, 256), (2, 4096, 4096), (1, 4096, 4096), (1, 256, 4096), (2, 1000, 777)):
    q, k, v = (torch.randn(B * L, 256, device=dev) for L in (M, N, N))
    out = torch.empty(B * M, 256, device=dev)
    for _ in range(3):
        ops.attention(q, k, v, B, M, N, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.attention(q, k, v, B, M, N, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    b = 0   # fp64 reference of the first sequence
    qh, kh, vh = (t[:L].double().view(L, 8, 32).transpose(0, 1) for t, L in ((q, M), (k, N), (v, N)))
    want = (torch.softmax(qh @ kh.transpose(1, 2) / 32 ** 0.5, -1) @ vh).transpose(0, 1).reshape(M, 256)
    err = float((out[:M].double() - want).abs().max())
    fl = 4 * B * M * N * 256
    print(f"{os.path.basename(_lib.LIB_PATH)}: {B} x ({M} queries, {N} keys): {us:.1f} us, {fl / us / 1e6:.1f} TFLOP/s fp32-equivalent, max err vs fp64 {err:.2e}")
