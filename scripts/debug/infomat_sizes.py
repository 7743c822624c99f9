"""Information matrix against the oracle at the edges of the chunked grid build: tiny, odd and very large scans, unequal sizes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from deeppointmap_amd import synthetic
from deeppointmap_amd.registration import calculate_information_matrix_from_pcd
from oracle import dpm_oracle as O
torch.set_grad_enabled(False)
g = torch.Generator().manual_seed(5)
for N1, N2 in [(1, 1), (17, 5), (4097, 4095), (20000, 70000), (70000, 20000), (150000, 150000)]:
    a = synthetic.frame(3, max(N1, 2))[:, :N1] * 60 if N1 > 1 else torch.tensor([[1.0], [2.0], [0.5]])
    b = synthetic.frame(4, max(N2, 2))[:, :N2] * 60 if N2 > 1 else torch.tensor([[1.2], [2.1], [0.4]])
    SE3 = torch.eye(4)
    SE3[:3, 3] = torch.tensor([0.4, -0.2, 0.05])
    want = O.information_matrix(a, b, SE3)
    got = calculate_information_matrix_from_pcd(a, b, SE3, device="cuda:0").cpu()
    scale = float(want.abs().max()) + 1e-30
    print(f"N1 {N1:6d} N2 {N2:6d}: matched {float(want[3, 3]):9.0f} vs {float(got[3, 3]):9.0f}, max rel err {float((got - want).abs().max()) / scale:.2e}")
