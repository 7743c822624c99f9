"""One neighbour query of the first encoder stage, a few times (for profilers): python scripts/knn_one.py SA0|LA0 [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import ops, synthetic

which, reps = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
pts, pad = synthetic.frames(64, 65536)
xyz, lengths = ops.prepare_points(pts.to(dev), pad.to(dev))
fidx, new_xyz, new_len = ops.fps(xyz, lengths, 4096)
args = (xyz, lengths, new_xyz, 32, 0.05) if which == "SA0" else (new_xyz, new_len, new_xyz, 32, 0.1)
for _ in range(reps):
    ops.knn_hybrid(*args)
torch.cuda.synchronize()
