"""The first level's two gather kernels (SetAbstraction with the affine point map, LocalAggregation) and a deeper one alone on
the chip, 64 frames: python scripts/gather_time.py  (DPM_LIB selects a build)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops, synthetic, _lib
from deeppointmap_amd.config import default_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
dev = torch.device("cuda")
enc = init_procedural(Encoder(default_args())).to(dev)
pts, pad = synthetic.frames(64, 65536)
xyz, lengths = ops.prepare_points(pts.to(dev).contiguous(), pad.to(dev).contiguous())
_, cen, clen = ops.fps(xyz, lengths, 4096)
idx0 = ops.knn_hybrid(xyz, lengths, cen, 32, 0.05)
idx1 = ops.knn_hybrid(cen, clen, cen, 32, 0.1)
m, q = "downsampler.0.sa.mlp", "downsampler.0.irm.0.la.mlp"
fea = torch.randn(64, 4096, 32, device=dev)


def t(fn, n=20):
    for _ in range(3):
        out = fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        out = fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3, out


a, oa = t(lambda: ops.group_mlp_max_from_xyz(xyz, enc.p("point_mlp0.weight"), enc.p("point_mlp0.bias"), cen, idx0, enc.p(m + ".0.weight"),
                                             enc.p(m + ".0.bias"), enc.p(m + ".1.ln.weight"), enc.p(m + ".1.ln.bias"), 0.05))
b, ob = t(lambda: ops.group_mlp_max(cen, fea, cen, idx1, enc.p(q + ".0.weight"), enc.p(q + ".0.bias"), enc.p(q + ".1.ln.weight"),
                                    enc.p(q + ".1.ln.bias"), 0.1))
print(f"{os.path.basename(_lib.LIB_PATH)}: affine gather <32> {a:.1f} us (sum {oa.double().sum().item():.6f}), projection + gather <32> {b:.1f} us (sum {ob.double().sum().item():.6f})")
