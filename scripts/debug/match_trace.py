"""debug: cycle stamps inside the two match kernels for ONE pair (experimental build: csrc/build.py --out <lib> -DDPM_EXPERIMENT,
DPM_LIB=<lib>).  Prints the phases of match_stats_kernel (strip 0) and match_topk_kernel (strip 0 and the merging strip)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from deeppointmap_amd import _lib, ops
lib = _lib.load()
torch.manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
a = torch.nn.functional.normalize(torch.randn(B, 256, 256, device="cuda"), dim=-1)
b = torch.nn.functional.normalize(torch.randn(B, 256, 256, device="cuda"), dim=-1)
for _ in range(3):
    ops.match_topk(a, b, 0.1, 1024)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); ops.match_topk(a, b, 0.1, 1024); e1.record(); torch.cuda.synchronize()
buf = (ctypes.c_longlong * 64)()
lib.dpm_debug_match_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
lib.dpm_debug_match_trace(buf, 64)
t = list(buf)
clk = 100e6   # s_memtime / readcyclecounter ticks at 100 MHz on this part
us = lambda x, y: (t[y] - t[x]) / clk * 1e6
print(f"pairs {B}: both launches {e0.elapsed_time(e1) * 1e3:.1f} us")
print(f"stats kernel, strip 0: strip GEMM {us(16, 17):.1f} us | statistics {us(17, 18):.1f} us")
print(f"topk kernel, strip 0: strip GEMM {us(0, 1):.1f} | column fold + P into LDS {us(1, 2):.1f} | strip selection {us(2, 3):.1f}")
print(f"merging strip: ticket -> candidates in LDS {us(4, 5):.1f} | select {us(5, 8):.1f} | collect {us(8, 9):.1f} | sort {us(9, 6):.1f} | store {us(6, 7):.1f}")
print(f"topk kernel start -> end of merge: {us(0, 7):.1f} us (strip 0 start to the merging strip's last store)")
