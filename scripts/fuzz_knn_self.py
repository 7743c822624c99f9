"""One-off fuzz of the self-kNN behind the statistical pre-processing filters (dpm_knn_self) against the oracle's exact
search (scipy cKDTree): the K nearest OTHER points' squared distances must agree to the bit."""
import os, random, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deeppointmap_amd import preprocess as P
from oracle import dpm_oracle as O
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
rng = random.Random(seed); g = torch.Generator().manual_seed(seed)
t0, n, bad = time.time(), 0, 0
while time.time() - t0 < budget:
    N = rng.choice([rng.randint(2, 100), rng.randint(100, 3000), rng.randint(3000, 40000)])
    K = rng.randint(1, min(32, N - 1))
    kind = rng.randint(0, 3)
    if kind == 0:
        xyz = torch.randn(N, 3, generator=g) * torch.tensor([30.0, 30.0, 1.5])
    elif kind == 1:
        xyz = torch.randint(-20, 21, (N, 3), generator=g).float() * 0.3      # lattice with duplicates
    elif kind == 2:
        xyz = torch.rand(N, 3, generator=g) * rng.choice([0.5, 5.0, 100.0])
    else:
        xyz = torch.randn(N, 3, generator=g) * 10 + torch.tensor([5000.0, -3000.0, 100.0])
    xyz = xyz.contiguous()
    widx, wd2 = O.knn_self(xyz, K)
    out = P.knn_self(xyz.cuda(), K)
    d2 = out["dist2"].cpu()
    if not torch.equal(d2, wd2):
        # the oracle computes in float64 from the tree and rounds: compare after recomputing in fp32 from ITS indices
        diff = xyz[:, None, :] - xyz[widx]
        ref32 = (diff[..., 0] * diff[..., 0] + diff[..., 1] * diff[..., 1]) + diff[..., 2] * diff[..., 2]
        # the tree ranks in float64, the kernel (like pytorch3d) in float32: on a lattice many neighbours are equally far in
        # exact arithmetic and a few ulp apart in fp32, so WHICH of them are the K nearest may differ -- only a K-th
        # distance that is off by more than rounding counts
        a, b = d2.sort(1)[0], ref32.sort(1)[0]
        if not bool(((a - b).abs() <= 2e-6 * b.abs().clamp(min=1e-12)).all()):
            bad += 1
            r = int((d2.sort(1)[0] != ref32.sort(1)[0]).any(1).nonzero()[0])
            print(f"KNN-SELF MISMATCH seed {seed}: N {N} K {K} kind {kind} row {r}: got {d2[r].tolist()[:6]} want {ref32[r].tolist()[:6]}")
    n += 1
print(f"seed {seed}: {n} clouds, {bad} mismatches, {time.time() - t0:.0f} s")
