import os, sys, time, torch
sys.path.insert(0, '.')
tile = os.environ.get('DPM_GEMM_V', '0')
from deeppointmap_amd import ops
torch.manual_seed(0)
shapes = [(16384, 256, 256), (32768, 256, 256), (65536, 256, 256), (262144, 256, 256), (16384, 256, 768), (16384, 256, 512), (262144, 32, 128), (262144, 128, 32), (65536, 64, 256), (65536, 256, 64), (16384, 128, 512), (16384, 512, 128), (4096, 1024, 256), (1024, 2048, 512), (32768, 512, 256)]
for R, K, N in shapes:
    x = torch.randn(R, K, device='cuda'); W = torch.randn(N, K, device='cuda'); b = torch.randn(N, device='cuda')
    out = torch.empty(R, N, device='cuda')
    ref = (x.double() @ W.double().t() + b.double()).float(); ops.linear(x, W, b, out=out); err = float((out - ref).abs().max())
    for _ in range(3): ops.linear(x, W, b, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): ops.linear(x, W, b, out=out)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    print(f'tile {tile}: R={R:6d} K={K:4d} N={N:4d}  {us:8.1f} us  {2*R*K*N/us/1e6:7.1f} TFLOP/s  maxerr {err:.2e}')
