"""The bf16x3 GEMM (csrc/gemm_b3.hip) against fp64 and against the exact-fp32 MFMA kernel: error level and time."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops
torch.manual_seed(0)
dev = "cuda"


def timed(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


print("| R | K | N | fp32 MFMA us | TFLOP/s | bf16x3 us | fp32-equivalent TFLOP/s | max err fp32 MFMA vs fp64 | max err bf16x3 vs fp64 | rms ratio |\n|---|---|---|---|---|---|---|---|---|---|")
for R, K, N in [(32768, 256, 768), (16384, 256, 768), (32768, 256, 256), (32768, 256, 512), (32768, 512, 256), (16384, 256, 256),
                (65536, 64, 256), (262144, 32, 128), (262144, 32, 32), (16384, 128, 512), (4096, 1024, 256), (1024, 2048, 512),
                (1000, 96, 132)]:
    x = torch.randn(R, K, device=dev) * 3
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev)
    res = torch.randn(R, N, device=dev)
    o32 = ops.linear(x, W, b, act=ops.ACT_RELU, residual=res, exact=True)
    o3 = ops.linear_bf16x3(x, W, b, act=ops.ACT_RELU, residual=res)
    assert o3 is not None, (R, K, N)
    n = min(R, 512)
    ref = torch.relu(x[:n].double() @ W.double().t() + b.double() + res[:n].double())
    e32, e3 = (o32[:n].double() - ref).abs(), (o3[:n].double() - ref).abs()
    out = torch.empty(R, N, device=dev)
    t32 = timed(lambda: ops.linear(x, W, b, out=out, exact=True))
    t3 = timed(lambda: ops.linear_bf16x3(x, W, b, out=out))
    print(f"| {R} | {K} | {N} | {t32:.1f} | {2 * R * K * N / t32 / 1e6:.1f} | {t3:.1f} | {2 * R * K * N / t3 / 1e6:.1f} | {float(e32.max()):.2e} | "
          f"{float(e3.max()):.2e} | {float((e3.pow(2).mean() / e32.pow(2).mean()).sqrt()):.2f} |")
# a weight view (rows of a bigger parameter, as the attention in-projection is used) hits the same planes
Wb = torch.randn(768, 256, device=dev) / 16
x = torch.randn(4096, 256, device=dev)
a = ops.linear_bf16x3(x, Wb[256:], None)
b_ = ops.linear_bf16x3(x, Wb[256:].clone(), None)
print("row-block view of a parameter == its copy:", torch.equal(a, b_))
