"""CPU: the algebra behind the folded / centred grouping layers (csrc/group_mlp.hip FOLD, CENTRED; ops._centred_layer), in fp64 on the
host -- the identities the kernels rely on, independent of any kernel:
  W_r (p - c) / r = W_r' p - W_r' c;  with W' = diag(sign gamma) (I - 11^T / C) W the pre-LayerNorm rows have zero mean, their mean
  square is the variance, and relu(max_k(gamma LN(y_k) + beta)) = relu(|gamma| max_k(y'_k rs_k) + beta)   (pointnext.py:52-61)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from deeppointmap_amd import ops  # noqa: E402


def test_centred_signed_layer_reproduces_layernorm_relu_max():
    g = torch.Generator().manual_seed(5)
    Cin, Cout, K, r = 32, 64, 16, 0.1
    W = torch.randn(Cout, Cin + 3, generator=g, dtype=torch.float64) + 0.7        # a common offset over the channels
    b = torch.randn(Cout, generator=g, dtype=torch.float64) + 0.3
    gamma, beta = torch.randn(Cout, generator=g, dtype=torch.float64), torch.randn(Cout, generator=g, dtype=torch.float64)
    gamma[5] = 0.0
    fea = torch.randn(K, Cin, generator=g, dtype=torch.float64)
    p, c = torch.rand(K, 3, generator=g, dtype=torch.float64) * 2 - 1, torch.rand(3, generator=g, dtype=torch.float64) * 2 - 1
    # the reference's arithmetic: Conv2d on [features ; (p - c) / r], LayerNorm over the channels, ReLU, max over the neighbours
    y = torch.cat([fea, (p - c) / r], 1) @ W.t() + b
    ln = (y - y.mean(1, keepdim=True)) / torch.sqrt(y.var(1, unbiased=False, keepdim=True) + 1e-5) * gamma + beta
    want = torch.relu(ln).max(0).values
    # the kernels' arithmetic on the derived layer
    Wf, Wr, bc = ops._centred_layer(W, b, gamma, Cin)                            # fp32 out of fp64 arithmetic
    sg = ops._gamma_sign(gamma)
    W2 = (W - W.mean(0, keepdim=True)) * sg                                      # the same in fp64, for the identity itself
    Wf64, Wr64, bc64 = W2[:, :Cin], W2[:, Cin:], ((b - b.mean()) * sg.reshape(-1))
    assert torch.allclose(Wf.double(), Wf64, atol=1e-6) and torch.allclose(Wr.double(), Wr64, atol=1e-6) and torch.allclose(bc.double(), bc64, atol=1e-6)
    P = fea @ Wf64.t() + bc64 + p @ (Wr64 / r).t()                               # projection + the point half (GEMM epilogue)
    yc = P - (Wr64 / r) @ c                                                      # gather: minus the centre half
    assert float((yc * sg.reshape(1, -1)).mean(1).abs().max()) < 1e-12           # zero mean by construction (before the signs)
    rs = torch.rsqrt((yc * yc).mean(1, keepdim=True) + 1e-5)
    got = torch.relu((yc * rs).max(0).values * gamma.abs() + beta)               # |gamma|, beta, ReLU once per centre
    torch.testing.assert_close(got, want, rtol=1e-11, atol=1e-11)
    # the affine first level: (A c + c0) + (A + W_r')(p - c) = A p + c0 + W_r'(p - c)
    A, c0 = torch.randn(Cout, 3, generator=g, dtype=torch.float64), torch.randn(Cout, generator=g, dtype=torch.float64)
    lhs = (A @ c + c0) + (p - c) @ (A + Wr64 / r).t()
    rhs = p @ A.t() + c0 + (p - c) @ (Wr64 / r).t()
    torch.testing.assert_close(lhs, rhs, rtol=1e-12, atol=1e-12)
