"""First-stage FPS (64 frames x 65 536 points -> 4096) per algorithm: 2 = bucket kernel over grid cells, 5 = over the STR packing
(16 384 < N <= 65 536)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import synthetic, ops
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64  # 1: the latency of a single frame
pts, _ = synthetic.frames(B, 65536)
xyz = pts.transpose(1, 2).contiguous().cuda()
lens = torch.full((B,), 65536, dtype=torch.int32, device="cuda")
ref = None
for algo in ([int(a) for a in sys.argv[2:]] or (2, 5)):
    for _ in range(2):
        out = ops.fps(xyz, lens, 4096, algo=algo)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        out = ops.fps(xyz, lens, 4096, algo=algo)
    e1.record(); torch.cuda.synchronize()
    same = "" if ref is None else f"  identical to algo 2: {torch.equal(out[0], ref)}"
    ref = out[0] if ref is None else ref
    ms = e0.elapsed_time(e1) / 3
    print(f"algo {algo}: {ms:.3f} ms for {B} frames, {ms * 1e3 / 4095:.3f} us per pick{same}")
