"""One-off fuzz: scan pre-processing head (voxel sample -> distance crop -> normalisation; bit-exact against the oracle,
single and batched) and the map tile assembly."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from deeppointmap_amd.maptile import assemble_map_tile
from deeppointmap_amd.preprocess import preprocess_scan, preprocess_scans
from oracle import dpm_oracle as O

torch.set_grad_enabled(False)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 100.0
rng = random.Random(seed)
g = torch.Generator().manual_seed(seed)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    scans = []
    for _ in range(rng.randint(1, 4)):
        N = rng.choice([rng.randint(1, 50), rng.randint(50, 5000), rng.randint(5000, 130000)])
        kind = rng.randint(0, 2)
        if kind == 0:
            xyz = torch.randn(N, 3, generator=g) * torch.tensor([rng.choice([5.0, 30.0, 80.0]), rng.choice([5.0, 30.0]), 2.0])
        elif kind == 1:
            xyz = torch.randint(-200, 201, (N, 3), generator=g).float() * 0.15      # on voxel edges: floor() boundaries
        else:
            xyz = (torch.rand(N, 3, generator=g) - 0.5) * 150
        scans.append(xyz + torch.randn(3, generator=g) * rng.choice([0.0, 10.0]))
    vs, lo, hi = rng.choice([0.3, 0.3, 0.1, 1.0]), rng.choice([1.0, 0.0, 3.0]), rng.choice([60.0, 20.0, 200.0])
    wants = [O.preprocess_scan(x, vs, lo, hi) for x in scans]
    for x, (wp, wi) in zip(scans, wants):
        try:
            pts, pad, idx = preprocess_scan(x, vs, lo, hi, return_index=True)
        except ValueError as e:
            if "max_cells" in str(e):
                continue
            raise
        if not (np.array_equal(pts[0].t().cpu().numpy(), wp.numpy()) and np.array_equal(idx.cpu().numpy(), wi.numpy().astype(np.int32))):
            bad += 1
            print(f"PREPROCESS MISMATCH seed {seed}: N {x.shape[0]} voxel {vs} crop [{lo}, {hi}]: {pts.shape[2]} vs {wp.shape[0]} points")
        n += 1
    try:
        bp, bpad, blen = preprocess_scans(scans, vs, lo, hi)
        for f, (wp, wi) in enumerate(wants):
            L = int(blen[f])
            if L != wp.shape[0] or not np.array_equal(bp[f, :, :L].t().cpu().numpy(), wp.numpy()):
                bad += 1
                print(f"BATCHED PREPROCESS MISMATCH seed {seed}: scan {f} of {len(scans)}: {L} vs {wp.shape[0]}")
    except ValueError as e:
        if "max_cells" not in str(e):
            raise
    # ---- map tile -----------------------------------------------------------------------------------------------------
    Kf, S = rng.randint(1, 16), rng.choice([256, 64, 100])
    kps = torch.randn(Kf, 131, S, generator=g)
    kps[:, 128:] *= 30
    poses = torch.eye(4).repeat(Kf, 1, 1)
    for i in range(Kf):
        a = float(torch.randn(1, generator=g)) * 0.5
        poses[i, :3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
        poses[i, :3, 3] = torch.randn(3, generator=g) * 20
    order = torch.randperm(Kf, generator=g)[:rng.randint(1, Kf)]
    tile = assemble_map_tile(kps.to("cuda:0"), poses, poses[int(order[0])], order.to(torch.int32))
    want = O.map_tile([kps[int(i)] for i in order], [poses[int(i)] for i in order], poses[int(order[0])])
    if not (np.array_equal(tile[:128].cpu().numpy(), want[:128].numpy()) and float((tile[128:].cpu() - want[128:]).abs().max()) < 5e-5):
        bad += 1
        print(f"MAP TILE MISMATCH seed {seed}: {Kf} key frames, S {S}, order {order.tolist()}: xyz err {float((tile[128:].cpu() - want[128:]).abs().max()):.2e}")
print(f"seed {seed}: {n} scans (+ batches, map tiles), {bad} mismatches, {time.time() - t0:.0f} s")
