"""debug: does the GEMM time depend on the DATA (power / clocks)?  fp32-MFMA kernel and bf16x3 kernel on zeros, on small
integers and on random normal inputs, 32 768 x 256 -> 768."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import ops, _lib
lib = _lib.load()
dev = 'cuda:0'
R, K, N = 32768, 256, 768
def split(W):
    planes = torch.empty(3, *W.shape, device=dev, dtype=torch.int16)
    _lib.check(lib.dpm_split_bf16x3(ops._ptr(W), W.numel(), ops._ptr(planes), ops._stream(W)), "split")
    return planes
def lin3(x, Wp, out):
    _lib.check(lib.dpm_linear_bf16x3(ops._ptr(x), K, ops._ptr(Wp), K, N * K, None, None, 0, ops._ptr(out), N, R, K, N, 0, ops._stream(x)), "lin3")
def tm(f, n=200):
    for _ in range(20): f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
out = torch.empty(R, N, device=dev)
for name, mk in (("zeros", lambda *s: torch.zeros(*s, device=dev)), ("ones", lambda *s: torch.ones(*s, device=dev)),
                 ("small ints", lambda *s: torch.randint(-3, 4, s, device=dev).float()), ("normal", lambda *s: torch.randn(*s, device=dev))):
    x, W = mk(R, K), mk(N, K)
    Wp = split(W)
    t32 = tm(lambda: ops.linear(x, W, None, out=out))
    t3 = tm(lambda: lin3(x, Wp, out))
    print(f"{name:11s}: fp32 MFMA {t32:6.1f} us   bf16x3 {t3:6.1f} us")
