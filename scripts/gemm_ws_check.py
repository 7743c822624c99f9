"""The wave-specialised GEMM (gemm_ws_kernel) against the 64 x 64 kernel: bit-equal outputs on full, ragged, biased,
residual and activated shapes, then the timing of both on the decoder's shapes.  Needs an experimental library
(DPM_GEMM_WS switches the dispatch): python deeppointmap_amd/csrc/build.py --out deeppointmap_amd/libdpm_exp.so -DDPM_EXPERIMENT;
DPM_LIB=$PWD/deeppointmap_amd/libdpm_exp.so python scripts/gemm_ws_check.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import _lib, ops
assert _lib.experimental(), __doc__
torch.manual_seed(0)
dev = "cuda"


def run(mode, x, W, b, res, act):
    os.environ["DPM_GEMM_WS"] = str(mode)
    return ops.linear(x, W, b, act=act, residual=res)


bad = 0
for R, K, N, use_b, use_r, act in [(32768, 256, 768, 1, 0, 0), (32768, 256, 256, 1, 1, 1), (32768 + 77, 256, 768, 1, 0, 0),
                                   (16384, 512, 256, 0, 1, 0), (33000, 64, 132, 1, 1, 2), (65536, 32, 128, 1, 0, 1),
                                   (32768, 1024, 256, 1, 0, 0), (40000, 96, 260, 0, 0, 0)]:
    x = torch.randn(R, K, device=dev)
    W = torch.randn(N, K, device=dev) / K ** 0.5
    b = torch.randn(N, device=dev) if use_b else None
    res = torch.randn(R, N, device=dev) if use_r else None
    o0 = run(0, x, W, b, res, act)
    o1 = run(1, x, W, b, res, act)
    o2 = run(2, x, W, b, res, act)
    same1, same2 = torch.equal(o0, o1), torch.equal(o0, o2)
    ref = (x[:256].double() @ W.double().t())
    if b is not None:
        ref += b.double()
    if res is not None:
        ref += res[:256].double()
    ref = torch.relu(ref) if act == 1 else torch.sigmoid(ref) if act == 2 else ref
    err = float((o1[:256].double() - ref).abs().max())
    print(f"R={R} K={K} N={N} bias={use_b} res={use_r} act={act}: ws == 64x64: {same1}, paced == 64x64: {same2}, max err vs fp64 {err:.2e}")
    bad += (not same1) + (not same2)


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


print("\n| R | K | N | 64 x 64 kernel us | TFLOP/s | wave-specialised us | TFLOP/s | paced (s_nop 1) us | TFLOP/s |\n|---|---|---|---|---|---|---|---|---|")
for R, K, N in [(32768, 256, 768), (16384, 256, 768), (32768, 256, 256), (32768, 256, 512), (32768, 512, 256), (16384, 256, 256),
                (65536, 64, 256), (262144, 32, 128), (32768, 1024, 256), (4096, 4096, 4096)]:
    x = torch.randn(R, K, device=dev); W = torch.randn(N, K, device=dev) / K ** 0.5; b = torch.randn(N, device=dev)
    out = torch.empty(R, N, device=dev)
    row = f"| {R} | {K} | {N} |"
    for mode in (0, 1, 2):
        os.environ["DPM_GEMM_WS"] = str(mode)
        us = timed(lambda: ops.linear(x, W, b, out=out))
        row += f" {us:.1f} | {2 * R * K * N / us / 1e6:.1f} |"
    print(row)


print("\nablations of the wave-specialised kernel (wrong results, right timing) on 32768 x 256 -> 768:")
x = torch.randn(32768, 256, device=dev); W = torch.randn(768, 256, device=dev) / 16; b = torch.randn(768, device=dev)
out = torch.empty(32768, 768, device=dev)
for mode, what in ((1, "as shipped"), (11, "no output stores"), (12, "no global loads"), (13, "neither"), (17, "neither, no LDS stores")):
    os.environ["DPM_GEMM_WS"] = str(mode)
    print(f"  {what}: {timed(lambda: ops.linear(x, W, b, out=out)):.1f} us")

import ctypes
lib = ctypes.CDLL(_lib.LIB_PATH)
for mode, what in ((1, "as shipped"), (17, "no memory traffic")):
    os.environ["DPM_GEMM_WS"] = str(mode)
    ops.linear(x, W, b, out=out); torch.cuda.synchronize()
    buf = (ctypes.c_longlong * 200)()
    lib.dpm_debug_ws_trace(buf, 200)
    tr = list(buf)
    t0 = tr[0]
    print(f"\n{what}: block 0 / matrix wave 0, shader cycles per K-tile: MFMA block | barrier wait | epilogue")
    for n in range(48):
        prev = tr[3 * (n - 1) + 3] if n else t0
        print(f"  K-tile {n:2d}: {tr[1 + 3 * n] - prev:6d} | {tr[2 + 3 * n] - tr[1 + 3 * n]:6d} | {tr[3 + 3 * n] - tr[2 + 3 * n]:6d}")
    print("  total", tr[3 * 47 + 3] - t0)
sys.exit(1 if bad else 0)
