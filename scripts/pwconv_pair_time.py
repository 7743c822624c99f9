import sys, torch
sys.path.insert(0, '.')
from deeppointmap_amd import ops
dev = "cuda"; torch.manual_seed(0)
C, H, R = 32, 128, 262144
W1 = torch.randn(H, C, 1, device=dev) / C ** 0.5; W2 = torch.randn(C, H, 1, device=dev) / H ** 0.5
b1, g1, be1 = (torch.randn(H, device=dev) for _ in range(3)); b2, g2, be2 = (torch.randn(C, device=dev) for _ in range(3))
x = torch.randn(R, C, device=dev); post = torch.randn(R, C, device=dev)
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
one = timed(lambda: ops.pwconv_pair(x, W1, b1, g1, be1, W2, b2, g2, be2, post))
def two():
    u = ops.linear_layernorm(x, W1, b1, g1, be1, act=ops.ACT_RELU)
    return ops.linear_layernorm(u, W2, b2, g2, be2, act=ops.ACT_RELU, post=post)
print(f"pw_conv pair, {R} rows: one kernel {one:.1f} us | two fused GEMM + LayerNorm kernels {timed(two):.1f} us")
