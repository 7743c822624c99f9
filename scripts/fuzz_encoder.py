"""One-off fuzz of Encoder.forward against the oracle (reduced configuration) on random frame sizes and paddings: key-point
coordinates must be bit-equal, features within 5e-4."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.config import default_args, reduced_args
from deeppointmap_amd.encoder import Encoder
from deeppointmap_amd.weights import init_procedural
from oracle import dpm_oracle as O

torch.set_grad_enabled(False)
if os.environ.get("PINNED_DIST") == "1":
    # The reference takes -2ab from torch.matmul, i.e. from the HOST's sgemm, whose last bit differs between CPUs.  The
    # kernels are pinned to the build container's result (the golden fixtures): a k-ordered fma chain.  This switch makes
    # the oracle use that chain explicitly, so that a fuzz on another host measures the kernels and not the host's BLAS.
    def pinned(a, b):
        a64, b64 = a.double(), b.double()
        t = (a[..., :, None, 0] * b[..., None, :, 0])
        t = (a64[..., :, None, 1] * b64[..., None, :, 1] + t.double()).float()
        t = (a64[..., :, None, 2] * b64[..., None, :, 2] + t.double()).float()
        d = -2 * t
        d += (a ** 2).sum(-1).unsqueeze(2)
        d += (b ** 2).sum(-1).unsqueeze(1)
        return d
    O.expanded_sqdist = pinned
from deeppointmap_amd import knobs
knobs.apply_env()   # DPM_FOLD_GATHER=0 / DPM_CENTRED_GATHER=0: the same cases through the other grouping-layer forms
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 120.0
rng = random.Random(seed)
FULL = os.environ.get("FULL_CONFIG") == "1"   # the shipped configuration on 16 384 - 65 536-point frames (slow oracle)
cfg = default_args() if FULL else reduced_args()
enc = init_procedural(Encoder(cfg)).to("cuda:0")
sd = {k: v.detach().cpu() for k, v in enc.flat().items()}
t0, n, bad, worst = time.time(), 0, 0, 0.0
errs = []
while time.time() - t0 < budget:
    B, N = rng.randint(1, 3), rng.choice([rng.randint(600, 3000), rng.randint(3000, 9000), 4096, 8192])
    if FULL:
        B, N = rng.randint(1, 2), rng.choice([65536, rng.randint(20000, 65536), 16384, rng.randint(5000, 16384)])
    start = rng.randint(0, 10_000)
    pts, pad = synthetic.frames(B, N, start=start)
    for b in range(B):
        if rng.random() < 0.5:
            L = rng.randint(max(N // 3, 600 if not FULL else 4500), N)
            pad[b, L:] = True
            pts[b, :, L:] = 0.0
    coor, fea, mask = enc(pts, pad)
    oc, of, _ = O.encoder_forward(sd, cfg, pts, pad, fast_fps=FULL)   # the C restatement of the sampling loop for big frames
    want = torch.cat([of, oc * 60.0], dim=1)            # (B, C + 3, S): features, then xyz * 60
    got = torch.cat([fea, coor * 60.0], 1).cpu()
    C = fea.shape[1]
    ok_xyz = torch.equal(got[:, C:], want[:, C:])
    err = float((got[:, :C] - want[:, :C]).abs().max())
    worst = max(worst, err)
    errs.append(err)
    n += 1
    if not ok_xyz or err > 5e-4:
        bad += 1
        print(f"MISMATCH seed {seed}: B {B} N {N} start {start} lengths {(~pad).sum(1).tolist()}: key points equal {ok_xyz}, feature err {err:.2e}")
errs.sort()
print(f"seed {seed}: {n} encoder passes, {bad} mismatches, feature err median {errs[len(errs) // 2]:.2e} / 90 % {errs[int(len(errs) * 0.9)]:.2e} / worst {worst:.2e}, {time.time() - t0:.0f} s")
