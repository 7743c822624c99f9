"""debug: when the stage-0 gather output of an encoder pass differs (bf16x3 GEMM loop on another stream), which neighbour
produced the differing maxima?"""
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
pts, pad = synthetic.frames(4, 8192, start=40)
x = torch.randn(4096, 256, device=dev); W = torch.randn(768, 256, device=dev) / 16; b = torch.randn(768, device=dev)
pre = torch.empty(4096, 768, device=dev)
stop = False


def noise():
    s = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(s):
        while not stop:
            for _ in range(20):
                ops.linear_bf16x3(x, W, b, out=pre)
            s.synchronize()


m = "downsampler.0.sa.mlp"
W0, b0 = enc.p("point_mlp0.weight").reshape(16, 3).double().cpu(), enc.p("point_mlp0.bias").double().cpu()
Wsa = enc.p(m + ".0.weight").reshape(32, 19).double().cpu(); bsa = enc.p(m + ".0.bias").double().cpu()
gm, bt = enc.p(m + ".1.ln.weight").double().cpu(), enc.p(m + ".1.ln.bias").double().cpu()


def per_neighbour(xyz, cen, idxrow, row):
    """(K,32) LayerNorm outputs of centre `row` for the neighbours idxrow, in double"""
    p = xyz[idxrow.long()]                                   # (K,3)
    fea = p @ W0.t() + b0                                    # (K,16)
    rel = (p - cen[row]) / 0.05
    y = torch.cat([fea, rel], 1) @ Wsa.t() + bsa             # (K,32)
    y = (y - y.mean(1, keepdim=True)) / torch.sqrt(y.var(1, unbiased=False, keepdim=True) + 1e-5) * gm + bt
    return y


# quiet references
ref = {}
for f in range(4):
    tr = {}
    enc(pts[f:f + 1], pad[f:f + 1], trace=tr, descriptor_scale=60.0)
    ref[f] = {k: v.clone() for k, v in tr.items() if isinstance(v, torch.Tensor)}
torch.cuda.synchronize()
th = threading.Thread(target=noise); th.start()
shown = 0
for it in range(400):
    f = it % 4
    tr = {}
    samp = enc.presample(pts[f:f + 1], pad[f:f + 1])
    enc(pts[f:f + 1], pad[f:f + 1], trace=tr, presampled=samp, descriptor_scale=60.0)
    out, want = tr["downsampler.0.sa.out"], ref[f]["downsampler.0.sa.out"]
    if not torch.equal(out, want) and shown < 6:
        shown += 1
        d = (out != want)[0]
        rows = d.any(-1).nonzero().flatten().tolist()
        idx_now = tr["downsampler.0.sa.idx"][0].cpu(); idx_ref = ref[f]["downsampler.0.sa.idx"][0].cpu()
        xyz = samp["xyz"][0].double().cpu(); cen = tr["downsampler.0.fps.new"][0].double().cpu()
        same_xyz = torch.equal(samp["xyz"], ops.prepare_points(pts[f:f + 1].to(dev).contiguous(), pad[f:f + 1].to(dev).contiguous())[0])
        print(f"pass {it} frame {f}: rows {rows[:8]} ({len(rows)}), xyz equals a fresh staging: {same_xyz}, "
              f"centres equal: {torch.equal(tr['downsampler.0.fps.new'], ref[f]['downsampler.0.fps.new'])}")
        for r in rows[:3]:
            cols = d[r].nonzero().flatten().tolist()
            set_now, set_ref = set(idx_now[r].tolist()), set(idx_ref[r].tolist())
            y = per_neighbour(xyz, cen, idx_now[r], r)
            mx = torch.clamp(y.max(0).values, min=0)
            print(f"   row {r}: idx set equal to quiet run: {set_now == set_ref}; cols {cols[:6]}; noisy {[round(float(out[0, r, c]), 4) for c in cols[:6]]} "
                  f"quiet {[round(float(want[0, r, c]), 4) for c in cols[:6]]} fp64 from the row's neighbours {[round(float(mx[c]), 4) for c in cols[:6]]}")
            # is the noisy value the maximum over a SUBSET of the neighbours?  which neighbour attains the quiet max?
            for c in cols[:3]:
                k_star = int(y[:, c].argmax())
                without = torch.clamp(torch.cat([y[:k_star, c], y[k_star + 1:, c]]).max(), min=0)
                print(f"      col {c}: neighbour slot {k_star} (point {int(idx_now[r, k_star])}) attains the maximum; without it the maximum is {float(without):.4f}")
stop = True; th.join()
