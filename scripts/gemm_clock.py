"""Shader clock / socket power (rocm-smi) while one GEMM variant runs in a loop: is a dense fp32-MFMA kernel power-limited?
DPM_LIB=<experimental library> python scripts/gemm_clock.py"""
import json, os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import _lib, ops
assert _lib.experimental()
dev = "cuda"
x = torch.randn(32768, 256, device=dev); W = torch.randn(768, 256, device=dev) / 16; b = torch.randn(768, device=dev)
out = torch.empty(32768, 768, device=dev)


def smi():
    o = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=20).stdout
    card = next(iter(json.loads(o).values()))
    return {k: v for k, v in card.items() if "sclk" in k.lower() or "power" in k.lower() or "mclk" in k.lower()}


for mode, what in ((0, "64 x 64 kernel"), (1, "wave-specialised"), (2, "wave-specialised, paced"), (17, "wave-specialised, no memory traffic")):
    os.environ["DPM_GEMM_WS"] = str(mode)
    got = []
    th = threading.Thread(target=lambda: (time.sleep(1.2), got.append(smi()), time.sleep(0.8), got.append(smi())))
    th.start()
    t0 = time.perf_counter(); n = 0
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    while time.perf_counter() - t0 < 3.0:
        for _ in range(50):
            ops.linear(x, W, b, out=out)
        n += 50
        torch.cuda.synchronize()
    e1.record(); torch.cuda.synchronize()
    th.join()
    print(f"{what}: {e0.elapsed_time(e1) / n * 1e3:.1f} us per launch over {n} launches; rocm-smi mid-run: {got}")
