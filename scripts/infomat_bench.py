"""Isolated timing of the batched information-matrix call (64 pairs x 65536 points), for rocprofv3:
    rocprofv3 --kernel-trace --stats -d out -o t -- python scripts/infomat_bench.py
"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import ops, synthetic

F, N = 64, 65536
dev = torch.device("cuda:0")
pts, _ = synthetic.frames(F, N)
pcd = (pts * 60).to(dev)
table = torch.zeros(F, 56, device=dev)
for f in range(F):
    SE3 = synthetic.relative_pose((f - 1) % F, f).float()
    table[f, :9] = SE3[:3, :3].reshape(9)
    table[f, 9:12] = SE3[:3, 3]
src = torch.tensor([(f - 1) % F for f in range(F)], dtype=torch.int32, device=dev)
dst = torch.arange(F, dtype=torch.int32, device=dev)
reps = int(os.environ.get("REPS", "5"))
for _ in range(2):
    ops.information_matrix_batched(pcd, src, dst, table[:, :12], table[:, 20:])
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    ops.information_matrix_batched(pcd, src, dst, table[:, :12], table[:, 20:])
e1.record()
torch.cuda.synchronize()
print(f"information_matrix_batched {F}x{N}: {e0.elapsed_time(e1) / reps:.3f} ms/call; matched[1]={float(table[1, 20 + 21]):.0f}")
