"""debug: the stage-0 (affine) gather kernel next to a bf16x3 GEMM loop on another stream -- what exactly changes?"""
import os, sys, threading
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import torch
from deeppointmap_amd import _lib, knobs, ops, synthetic
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
W0, b0 = enc.p("point_mlp0.weight"), enc.p("point_mlp0.bias")
W, bias, gm, bt = enc.p(m + ".0.weight"), enc.p(m + ".0.bias"), enc.p(m + ".1.ln.weight"), enc.p(m + ".1.ln.bias")
vfn = lambda: ops.group_mlp_max_from_xyz(xyz, W0, b0, cen, idx, W, bias, gm, bt, 0.05)
quiet = vfn().clone()
torch.cuda.synchronize()
key = [k for k in ops._DERIVED if k[0] == "affine-stage0"][0]
A, cvec = ops._DERIVED[key][2]
A0, c0 = A.clone(), cvec.clone()
x = torch.randn(4096, 256, device=dev); Wn = torch.randn(768, 256, device=dev) / 16; bn = torch.randn(768, device=dev)
pre = torch.empty(4096, 768, device=dev)
lib = _lib.load()


def private():   # the kernel through the C ABI on private copies of A / cvec, output preallocated
    out = torch.empty(1, 512, 32, device=dev)
    W2 = W.reshape(32, 19)
    _lib.check(lib.dpm_group_affine_ln_max(A0.data_ptr(), c0.data_ptr(), xyz.data_ptr(), cen.data_ptr(), idx.data_ptr(), W2.data_ptr() + 4 * 16, 19,
                                           gm.data_ptr(), bt.data_ptr(), 1, 8192, 512, 32, 32, 0.05, out.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream), "x")
    return out


for nname, nfn in (("bf16x3 GEMM, output allocated per call", lambda: ops.linear_bf16x3(x, Wn, bn)),
                   ("bf16x3 GEMM, preallocated output", lambda: ops.linear_bf16x3(x, Wn, bn, out=pre)),
                   ("fp32 GEMM, output allocated per call", lambda: ops.linear(x, Wn, bn, exact=True))):
    stop = False

    def noise():
        s = torch.cuda.Stream(device=dev)
        with torch.cuda.stream(s):
            while not stop:
                for _ in range(30):
                    nfn()
                s.synchronize()
    th = threading.Thread(target=noise); th.start()
    bad = badp = 0; worst = 0.0
    for _ in range(300):
        o = vfn()
        if not torch.equal(o, quiet):
            bad += 1; worst = max(worst, float((o - quiet).abs().max()))
        if not torch.equal(private(), quiet):
            badp += 1
    torch.cuda.synchronize()
    print(f"noise {nname}: wrapper {bad}/300 differ from the quiet result (worst {worst:.3e}), C-ABI call on private copies {badp}/300; "
          f"A unchanged {torch.equal(A, A0)}, cvec unchanged {torch.equal(cvec, c0)}")
    stop = True; th.join()
