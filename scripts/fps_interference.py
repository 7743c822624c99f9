"""Which shared resource does the first-level sampling kernel take from the kernels next to it?  Four chip-filling victim kernels with
one bound each (vector ALU, HBM stream, L2-resident gather, LDS + barriers; scripts/micro/victims.hip) are timed alone and next to the
real first-level sampling kept in flight on two streams (64 frames each, back to back), and next to the sampling of fewer frames.
Needs deeppointmap_amd/csrc/build/libvictims.so (hipcc --offload-arch=gfx950 -O3 -shared -fPIC scripts/micro/victims.hip).
Run with GPU_MAX_HW_QUEUES=16 so that no two streams share a hardware queue."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from deeppointmap_amd import ops, synthetic

dev = torch.device("cuda:0")
V = ctypes.CDLL(os.path.join(ROOT, "deeppointmap_amd", "csrc", "build", "libvictims.so"))
vp, ci, cz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
V.run_alu.argtypes = [ci, vp, ci, vp]
V.run_stream.argtypes = [ci, vp, vp, cz, vp]
V.run_gather.argtypes = [ci, vp, vp, vp, ci, ci, vp]
V.run_lds.argtypes = [ci, vp, ci, vp]
out = torch.zeros(16, device=dev)
big_in, big_out = torch.zeros(1 << 28, device=dev), torch.empty(1 << 28, device=dev)          # 1 GiB each
tab = torch.rand(1 << 16, 4, device=dev)                                                          # 1 MB: L2-resident
idx = torch.randint(0, 1 << 16, (1 << 20,), device=dev, dtype=torch.int32)
main = torch.cuda.current_stream(dev)
victims = {
    "vector ALU (fma chains, no memory)": lambda: V.run_alu(8192, out.data_ptr(), 4000, main.cuda_stream),
    "HBM stream (1 GiB in, 1 GiB out)": lambda: V.run_stream(8192, big_in.data_ptr(), big_out.data_ptr(), (1 << 28) // 4, main.cuda_stream),
    "L2-resident gather (16 B from a 1 MB table)": lambda: V.run_gather(8192, tab.data_ptr(), idx.data_ptr(), out.data_ptr(), 1 << 20, 400, main.cuda_stream),
    "LDS + workgroup barriers": lambda: V.run_lds(8192, out.data_ptr(), 3000, main.cuda_stream),
}


def time_victims(n=6):
    res = {}
    for name, fn in victims.items():
        fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        e1.synchronize()
        res[name] = e0.elapsed_time(e1) / n * 1e3
    return res


F = 64
pts, _ = synthetic.frames(F, 65536)
xyz = pts.transpose(1, 2).contiguous().to(dev)
lens = torch.full((F,), 65536, dtype=torch.int32, device=dev)
ops.fps(xyz, lens, 4096)
torch.cuda.synchronize()
base = time_victims()
table = [("alone", base)]
sa, sb = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
for nf in (64, 16, 4):
    torch.cuda.synchronize()
    x, l = xyz[:nf].contiguous(), lens[:nf].contiguous()
    ev = []
    for i in range(2 * 40):   # ~40 launches of >= 3.5 ms per stream: longer than the measurement
        with torch.cuda.stream(sa if i % 2 == 0 else sb):
            e0 = torch.cuda.Event(enable_timing=True); e0.record()
            ops.fps(x, l, 4096)
            e1 = torch.cuda.Event(enable_timing=True); e1.record()
            ev.append((e0, e1))
    r = time_victims()
    busy = sum(1 for a, b in ev if not b.query())
    torch.cuda.synchronize()
    fps_ms = sum(a.elapsed_time(b) for a, b in ev[4:20]) / 16
    table.append((f"next to the sampling of 2 x {nf} frames in flight ({fps_ms:.2f} ms per launch; {busy} of {len(ev)} launches still pending when the victims had finished)", r))
print("| configuration | " + " | ".join(victims) + " |")
print("|---|" + "---|" * len(victims))
for label, r in table:
    print(f"| {label} | " + " | ".join(f"{r[k]:.0f} us ({r[k] / base[k]:.2f} x)" for k in victims) + " |")
