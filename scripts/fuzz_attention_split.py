"""Fuzz of the key-split attention (dpm_attention_split) against fp64 torch and against the plain kernel: random batch /
query / key counts (ragged last tiles and ranges), every admissible range count, shifted keys.
usage: python scripts/fuzz_attention_split.py [cases]"""
import sys

import torch

sys.path.insert(0, ".")
from deeppointmap_amd import _lib, ops  # noqa: E402
from deeppointmap_amd.ops import _ptr, _stream  # noqa: E402

DEV = "cuda:0"
lib = _lib.load()
g = torch.Generator().manual_seed(2026)
cases = int(sys.argv[1]) if len(sys.argv) > 1 else 300
worst, bad = 0.0, 0
for t in range(cases):
    B = int(torch.randint(1, 4, (1,), generator=g))
    M = int(torch.randint(1, 300, (1,), generator=g))
    N = int(torch.randint(65, 5000, (1,), generator=g))
    ns = int(torch.randint(2, 65, (1,), generator=g))
    chunk = -(-N // (64 * ns)) * 64
    ns = -(-N // chunk)
    if ns < 2:
        continue
    shift = int(torch.randint(0, B, (1,), generator=g))
    q, k, v = (torch.randn(B * n, 256, generator=g).to(DEV) for n in (M, N, N))
    out = torch.empty(B * M, 256, device=DEV)
    ws = torch.empty(lib.dpm_attention_split_workspace_bytes(B, M, 8, 32, ns), dtype=torch.uint8, device=DEV)
    _lib.check(lib.dpm_attention_split(_ptr(q), 256, M * 256, _ptr(k), 256, N * 256, _ptr(v), 256, N * 256, _ptr(out), 256, M * 256,
                                       B, M, N, 8, 32, shift, ns, _ptr(ws), _stream(q)), "split")
    plain = torch.empty_like(out)
    _lib.check(lib.dpm_attention_shifted(_ptr(q), 256, M * 256, _ptr(k), 256, N * 256, _ptr(v), 256, N * 256, _ptr(plain), 256,
                                         M * 256, B, M, N, 8, 32, shift, _stream(q)), "plain")
    qh = q.view(B, M, 8, 32).transpose(1, 2).double()
    kh = k.view(B, N, 8, 32).transpose(1, 2).double().roll(-shift, 0)
    vh = v.view(B, N, 8, 32).transpose(1, 2).double().roll(-shift, 0)
    want = (torch.softmax(qh @ kh.transpose(-1, -2) / 32 ** 0.5, -1) @ vh).transpose(1, 2).reshape(B * M, 256)
    e_split, e_plain = float((out.double() - want).abs().max()), float((plain.double() - want).abs().max())
    worst = max(worst, e_split)
    if not (e_split < 2e-5 and e_split < 4 * e_plain + 1e-6) or not torch.isfinite(out).all():
        bad += 1
        print("MISMATCH", B, M, N, ns, shift, e_split, e_plain)
print(f"{cases} cases, {bad} bad, worst |split - fp64| = {worst:.2e}")
