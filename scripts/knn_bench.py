"""Isolated timing of the two grid-search neighbour queries of the first encoder stage (64 frames).
The A/B half (DPM_KNN_FAST flipped inside the process) needs an experimental library:
  python deeppointmap_amd/csrc/build.py --out /tmp/libdpm_exp.so -DDPM_EXPERIMENT; DPM_LIB=/tmp/libdpm_exp.so python scripts/knn_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import ops, synthetic

F, N = 64, 65536
dev = torch.device("cuda:0")
pts, pad = synthetic.frames(F, N)
xyz, lengths = ops.prepare_points(pts.to(dev), pad.to(dev))
fidx, new_xyz, new_len = ops.fps(xyz, lengths, 4096)
torch.cuda.synchronize()


def timed(name, fn, reps=5):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / reps * 1e3:.1f} us")


timed("SA0  N=65536 S=4096 r=0.05 K=32", lambda: ops.knn_hybrid(xyz, lengths, new_xyz, 32, 0.05))
timed("LA0  N=4096  S=4096 r=0.10 K=32", lambda: ops.knn_hybrid(new_xyz, new_len, new_xyz, 32, 0.1))

# the quarter-wave search against the one-wave-per-centre search: identical sets on every row
from deeppointmap_amd import _lib
if not _lib.experimental():
    sys.exit("A/B half skipped: the shipped library has no DPM_KNN_FAST switch (see the docstring)")
def sets(a):
    return torch.sort(a, dim=-1).values
for name, args in (("SA0", (xyz, lengths, new_xyz, 32, 0.05)), ("LA0", (new_xyz, new_len, new_xyz, 32, 0.1))):
    os.environ["DPM_KNN_FAST"] = "1"
    a = ops.knn_hybrid(*args)
    os.environ["DPM_KNN_FAST"] = "0"
    b = ops.knn_hybrid(*args)
    timed(name + " one wave per centre", lambda: ops.knn_hybrid(*args))
    os.environ["DPM_KNN_FAST"] = "1"
    bad = (sets(a) != sets(b)).any(-1)
    print(name, "rows differing:", int(bad.sum()), "of", bad.numel(), "slot0 differing:", int((a[..., 0] != b[..., 0]).sum()))
