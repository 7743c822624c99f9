"""GEMM shapes of the path (R rows, K = Cin, N = Cout) through dpm_linear: time, TFLOP/s, max error vs fp64."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops
torch.manual_seed(0)
shapes = [(32768, 256, 768), (32768, 256, 256), (32768, 256, 512), (32768, 512, 256), (16384, 256, 256), (262144, 32, 128),
          (262144, 128, 32), (262144, 32, 32), (65536, 64, 256), (65536, 256, 64), (16384, 128, 512), (16384, 512, 128),
          (4096, 1024, 256), (1024, 2048, 512), (4096, 4096, 4096)]
for R, K, N in shapes:
    x = torch.randn(R, K, device='cuda'); W = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    out = torch.empty(R, N, device='cuda')
    ops.linear(x, W, b, out=out)
    err = float((out[:512] - (x[:512].double() @ W.double().t() + b.double()).float()).abs().max())
    for _ in range(2): ops.linear(x, W, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): ops.linear(x, W, b, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 10 * 1e3
    print(f'R={R:6d} K={K:4d} N={N:4d}  {us:8.1f} us  {2*R*K*N/us/1e6:7.1f} TFLOP/s  maxerr {err:.2e}')
