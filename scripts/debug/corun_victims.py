"""debug: which kernels return different results when a bf16x3 GEMM loop runs on another stream of the same process?"""
import os, sys, threading
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from deeppointmap_amd import knobs, ops, synthetic
from deeppointmap_amd.config import reduced_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
knobs.GEMM_BF16X3 = False
enc = init_procedural(Encoder(reduced_args())).to(dev)
pts, pad = synthetic.frames(1, 8192, start=40)
xyz, lengths = ops.prepare_points(pts.to(dev).contiguous(), pad.to(dev).contiguous())
_, cen, clen = ops.fps(xyz, lengths, 512)
idx = ops.knn_hybrid(xyz, lengths, cen, 32, 0.05)
m = "downsampler.0.sa.mlp"
fea = torch.randn(1, 512, 32, device=dev)
idx2 = ops.knn_hybrid(cen, clen, cen, 32, 0.1)
q = "downsampler.0.irm.0.la.mlp"
g = torch.randn(4096, 256, device=dev); gm, bt = torch.randn(256, device=dev), torch.randn(256, device=dev)
X = torch.randn(4096, 256, device=dev); Wd = torch.randn(256, 256, device=dev) / 16; bd = torch.randn(256, device=dev)
qkv = torch.randn(8 * 256, 768, device=dev)
VICTIMS = {
    "gather affine <32> (stage-0 SetAbstraction)": lambda: ops.group_mlp_max_from_xyz(xyz, enc.p("point_mlp0.weight"), enc.p("point_mlp0.bias"), cen, idx,
                                                            enc.p(m + ".0.weight"), enc.p(m + ".0.bias"), enc.p(m + ".1.ln.weight"), enc.p(m + ".1.ln.bias"), 0.05),
    "project + gather <32> (LocalAggregation)": lambda: ops.group_mlp_max(cen, fea, cen, idx2, enc.p(q + ".0.weight"), enc.p(q + ".0.bias"), enc.p(q + ".1.ln.weight"),
                                                                       enc.p(q + ".1.ln.bias"), 0.1),
    "layernorm 4096 x 256": lambda: ops.layernorm(g, gm, bt),
    "fp32 GEMM 4096 x 256 -> 256": lambda: ops.linear(X, Wd, bd, exact=True),
    "fp32 GEMM + LayerNorm (two kernels)": lambda: ops.linear_layernorm(X, Wd, bd, gm, bt),
    "attention 8 x 256": lambda: ops.attention(qkv[:, :256], qkv[:, 256:512], qkv[:, 512:], 8, 256, 256, 8),
    "neighbour search (sorted rows)": lambda: torch.sort(ops.knn_hybrid(xyz, lengths, cen, 32, 0.05), dim=-1).values,
    "farthest point sampling": lambda: ops.fps(xyz, lengths, 512)[0],
}
x = torch.randn(4096, 256, device=dev); W = torch.randn(768, 256, device=dev) / 16; b = torch.randn(768, device=dev)
xs = torch.randn(512, 32, device=dev); Ws = torch.randn(32, 32, device=dev); bs = torch.randn(32, device=dev)
NOISE = {
    "bf16x3 GEMM 64 x 64 tiles": lambda: ops.linear_bf16x3(x, W, b),
    "bf16x3 GEMM 32 x 32 tiles": lambda: ops.linear_bf16x3(xs, Ws, bs),
    "fp32 MFMA GEMM": lambda: ops.linear(x, W, b, exact=True),
}
for nname, nfn in NOISE.items():
    stop = False

    def noise():
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            while not stop:
                for _ in range(30):
                    nfn()
                s.synchronize()
    th = threading.Thread(target=noise); th.start()
    print(f"noise: {nname}")
    for vname, vfn in VICTIMS.items():
        ref = vfn().clone()
        bad = 0
        for _ in range(300):
            if not torch.equal(vfn(), ref):
                bad += 1
        print(f"    {vname}: {bad} of 300 calls differ")
    stop = True; th.join()
