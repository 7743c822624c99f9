"""debug: first batched-preprocess mismatch of scripts/fuzz_preprocess.py (same random stream), dissected"""
import os, random, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from deeppointmap_amd.preprocess import preprocess_scan, preprocess_scans
from oracle import dpm_oracle as O

torch.set_grad_enabled(False)
seed = 0
rng = random.Random(seed)
g = torch.Generator().manual_seed(seed)
found = 0
while found < 2:
    scans = []
    for _ in range(rng.randint(1, 4)):
        N = rng.choice([rng.randint(1, 50), rng.randint(50, 5000), rng.randint(5000, 130000)])
        kind = rng.randint(0, 2)
        if kind == 0:
            xyz = torch.randn(N, 3, generator=g) * torch.tensor([rng.choice([5.0, 30.0, 80.0]), rng.choice([5.0, 30.0]), 2.0])
        elif kind == 1:
            xyz = torch.randint(-200, 201, (N, 3), generator=g).float() * 0.15
        else:
            xyz = (torch.rand(N, 3, generator=g) - 0.5) * 150
        scans.append((kind, xyz + torch.randn(3, generator=g) * rng.choice([0.0, 10.0])))
    vs, lo, hi = rng.choice([0.3, 0.3, 0.1, 1.0]), rng.choice([1.0, 0.0, 3.0]), rng.choice([60.0, 20.0, 200.0])
    wants = [O.preprocess_scan(x, vs, lo, hi) for _, x in scans]
    try:
        bp, bpad, blen = preprocess_scans([x for _, x in scans], vs, lo, hi)
    except ValueError:
        bp = None
    if bp is not None:
        for f, (wp, wi) in enumerate(wants):
            L = int(blen[f])
            if L != wp.shape[0] or not np.array_equal(bp[f, :, :L].t().cpu().numpy(), wp.numpy()):
                found += 1
                kind, x = scans[f]
                print(f"scan {f} of {len(scans)}: kind {kind} N {x.shape[0]} voxel {vs} crop [{lo}, {hi}]: batched {L} vs oracle {wp.shape[0]}")
                pts, pad, idx = preprocess_scan(x, vs, lo, hi, return_index=True)
                print("   single path:", pts.shape[2], "equal to oracle:", np.array_equal(pts[0].t().cpu().numpy(), wp.numpy()))
                got = bp[f, :, :L].t().cpu()
                a = {tuple(r) for r in got.numpy().round(7).tolist()}
                b = {tuple(r) for r in wp.numpy().round(7).tolist()}
                extra, missing = list(a - b)[:4], list(b - a)[:4]
                print("   in batched only:", extra, " in oracle only:", missing)
                for e in extra:
                    p = np.array(e, dtype=np.float32) * np.float32(60.0)
                    print("      metres", p, "range", float(np.linalg.norm(p)), "voxel coords", (p / np.float32(vs)).tolist())
    # keep the random stream in step with the fuzzer's map-tile part
    Kf, S = rng.randint(1, 16), rng.choice([256, 64, 100])
    kps = torch.randn(Kf, 131, S, generator=g)
    for i in range(Kf):
        torch.randn(1, generator=g), torch.randn(3, generator=g)
    torch.randperm(Kf, generator=g), rng.randint(1, Kf)
