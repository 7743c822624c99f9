"""debug: how many rows does the grid search queue for the exact tie replay, per level of the benchmark batch?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import ops, synthetic, _lib
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
dev = "cuda:0"
cfg = default_args()
enc = init_procedural(Encoder(cfg)).to(dev)
pts, pad = synthetic.frames(64, 65536)
pre = enc.presample(pts.to(dev), pad.to(dev))
GDIM = 128
def tie_count(ws, B, N):
    base = ws.data_ptr()
    p = (base + 255) & ~255
    p = (p + 16 * B + 255) & ~255
    p = (p + 4 * B * (GDIM * GDIM + 1) + 255) & ~255
    p = (p + 16 * B * N + 255) & ~255
    off = p - base
    return int(ws[off:off + 4].view(torch.int32)[0])
xyz, lens = pre["xyz"], pre["lengths"]
levels = [(xyz, lens)] + [(pre[f"xyz{i}"], pre[f"len{i}"]) for i in range(5)]
for i in range(2):
    (p0, l0), (p1, l1) = levels[i], levels[i + 1]
    r, K = cfg.encoder.radius_list[i], cfg.encoder.nsample_list[i]
    for name, P, L, C, rad, k in (("SA", p0, l0, p1, r[0], K[0]), ("LA", p1, l1, p1, r[1], K[1])):
        B, N, _ = P.shape
        if N < 1024: continue
        ws = ops.knn_grid(P, L, rad)
        ops.knn_hybrid(P, L, C, k, rad, grid=ws)
        torch.cuda.synchronize()
        print(f"level {i} {name}: N={N} S={C.shape[1]} K={k} r={rad}: queued tie rows {tie_count(ws, B, N)} of {B * C.shape[1]}")
