"""debug / check: points placed within a few ulps of the crop radius -- the device crop must keep exactly the points
torch.norm (the reference's DistanceSample) keeps."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
import torch
from deeppointmap_amd.preprocess import preprocess_scan
from oracle import dpm_oracle as O

torch.set_grad_enabled(False)
g = torch.Generator().manual_seed(3)
bad = tot = 0
for trial in range(20):
    n = 60000
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    r = torch.full((n, 1), 60.0)
    k = torch.randint(-6, 7, (n, 1), generator=g).float()
    x = d * (r + k * 3.8e-6)                    # radii within +-6 ulp of 60 m
    x += (torch.rand(n, 3, generator=g) - 0.5) * 1e-5
    wp, wi = O.preprocess_scan(x, 0.4, 1.0, 60.0)    # voxels of 40 cm on a 60 m sphere: most points keep their own voxel
    pts, pad, idx = preprocess_scan(x, 0.4, 1.0, 60.0, return_index=True)
    ok = pts.shape[2] == wp.shape[0] and np.array_equal(idx.cpu().numpy(), wi.numpy().astype(np.int32))
    tot += 1
    bad += not ok
    print(f"trial {trial}: {wp.shape[0]} of {n} kept by the reference, device {pts.shape[2]}: {'equal' if ok else 'DIFFERENT'}")
print(f"{bad} of {tot} scans differ")
