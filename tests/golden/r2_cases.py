"""Seeded INPUTS of the round-2 fixtures (large_reg.npz, knn_ties.npz, converged.npz): built from committed data and
torch CPU generators, so the fixtures store only what the reference answered.  Imported by
tests/golden/make_golden_r2.py (which runs the reference on them) and by the tests (which run the oracle and the HIP
path on them).  No reference code here."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))

# ---- key-frame pool for the map-sized registrations -------------------------------------------------------------
TILE_A, TILE_B, SCAN = 8, 25, 20          # centre tokens of the two 16-scan tiles; the scan registered against tile A
LOOP_SRC, LOOP_DST = list(range(0, 32, 2)), 33  # 16 loop-closure candidates against one new scan
N_POOL = 34


def keyframe_pool():
    """34 key-frame descriptors (131, 256) + poses: the 15 descriptors the reference's encoder produced in
    slam_trace.npz, re-used with seeded perturbations (non-negative feature noise, 0.3 m coordinate noise), on a
    gently curving path with 1.5 m between key-frames."""
    with np.load(os.path.join(HERE, "slam_trace.npz")) as z:
        base = torch.from_numpy(z["desc"].copy())
    g = torch.Generator().manual_seed(77)
    kps, poses = [], []
    for i in range(N_POOL):
        b = base[i % base.shape[0]]
        noise = torch.cat([0.02 * torch.rand(128, 256, generator=g), 0.3 * torch.randn(3, 256, generator=g)])
        kps.append((b + noise).contiguous() if i >= base.shape[0] else b.clone())
        a = 0.04 * i
        SE3 = torch.eye(4)
        SE3[:3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
        SE3[:3, 3] = torch.tensor([1.5 * i, 0.3 * np.sin(0.5 * i), 0.02 * i])
        poses.append(SE3)
    return kps, poses


# ---- lattices for the tie rule of torch.topk's std::nth_element branch --------------------------------------------
def knn_tie_cases():
    """name -> (points (N,3), centres (S,3), radius, K).  Lattice spacing 0.25: the expanded-form distances
    -2ab + |a|^2 + |b|^2 of a lattice are exact in fp32, so every row holds groups of EXACTLY equal distances, also
    across the K-th place; k * 64 > n in every case (nth_element, not partial_sort)."""
    g = torch.Generator().manual_seed(5)

    def lattice(nx, ny, nz):
        p = torch.stack(torch.meshgrid(torch.arange(nx * 1.0), torch.arange(ny * 1.0), torch.arange(nz * 1.0), indexing="ij"), -1)
        p = p.reshape(-1, 3) * 0.25
        return p[torch.randperm(p.shape[0], generator=g)].contiguous()

    p16, p64, p256, p1024 = lattice(4, 2, 2), lattice(4, 4, 4), lattice(8, 8, 4), lattice(16, 16, 4)
    return {
        "n16_k16_r1.6": (p16, p16.clone(), 1.6, 16),
        "n64_k16_r0.8": (p64, p64[:16].contiguous(), 0.8, 16),
        "n64_k32_r0.8": (p64, p64.clone(), 0.8, 32),
        "n64_k32_r0.4": (p64, p64.clone(), 0.4, 32),     # the radius cut removes part of every row
        "n256_k32_r0.4": (p256, p256.clone(), 0.4, 32),
        "n256_k32_r0.8": (p256, p256[::4].contiguous(), 0.8, 32),
        "n1024_k32_r0.6": (p1024, p1024[::4].contiguous(), 0.6, 32),  # the grid search's nth_element regime (1024 <= N < 2048)
    }


# ---- a registration that converges ----------------------------------------------------------------------------------
CONVERGED_TAU = 0.02


def converged_state_dict(sd):
    """procedural decoder weights, edited so that descriptor identity dominates the match: input projection x100,
    the attention blocks' output projections and MLP exits x0.01 (the residual stream keeps the projected descriptor),
    offset head output zero (correspondences are the matched key-points themselves).  With tau = CONVERGED_TAU."""
    sd = {k: v.clone() for k, v in sd.items()}
    sd["projection.weight"] *= 100.0
    for k in sd:
        if k.startswith("descriptor_attention") and (k.endswith("out_proj.weight") or k.endswith("mlp.2.weight")):
            sd[k] *= 0.01
    sd["offset_head.head.weight"].zero_()
    sd["offset_head.head.bias"].zero_()
    return sd


def converged_cases():
    """name -> (src (131,256), dst (131,256), R_gt, t_gt): zero-mean random descriptors at random key-points, the
    target = the source moved by a small rigid motion, permuted, with 2 cm coordinate noise (and feature noise in the
    second case, so that confident and unconfident pairs mix)."""
    out = {}
    for name, seed, noise in (("clean", 0, 0.0), ("noisy", 1, 0.6)):
        g = torch.Generator().manual_seed(seed)
        fea = torch.randn(128, 256, generator=g)
        xyz = torch.cat([(torch.rand(2, 256, generator=g) * 2 - 1) * 40, torch.randn(1, 256, generator=g)], 0)
        a = 0.05
        R = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
        t = torch.tensor([[0.8], [-0.3], [0.05]])
        perm = torch.randperm(256, generator=g)
        src = torch.cat([fea, xyz], 0)
        dst = torch.cat([fea + noise * torch.randn(128, 256, generator=g),
                         R @ xyz + t + 0.02 * torch.randn(3, 256, generator=g)], 0)[:, perm].contiguous()
        out[name] = (src, dst, R, t)
    return out


# ---- farthest point sampling from a random start point ------------------------------------------------------------------
RANDOM_START_SEED = 11


def random_start_cases():
    """name -> (points (B,N,3), padding (B,N) bool, K).  The start indices come from random.randint under
    random.seed(RANDOM_START_SEED), one draw per frame in batch order, cases in this dict's order."""
    g = torch.Generator().manual_seed(21)
    small = torch.rand(3, 1000, 3, generator=g) * 2 - 1
    pad = torch.zeros(3, 1000, dtype=torch.bool)
    pad[1, 700:] = True
    pad[2, 40:] = True            # fewer valid points than K: -1 padding after the 40th pick
    big = torch.randn(1, 20000, 3, generator=g) * torch.tensor([30.0, 30.0, 1.5])
    return {"small_ragged": (small, pad, 64), "bucket_20000": (big, torch.zeros(1, 20000, dtype=torch.bool), 256)}


# ---- padding masks (key_padding_mask of the attention blocks) ----------------------------------------------------------
def masked_cases():
    """name -> (src (B,131,M), dst (B,131,N), src_mask (B,M) bool or None, dst_mask (B,N) bool or None): key-frame
    descriptors of the pool above whose tail tokens are padding (zero columns, mask True), as a collated batch of
    ragged descriptor sets would look.  'reg_256' and 'reg_1024x256' are registrations (B = 1; equal and unequal
    token counts take different launch paths), 'loop4' a loop-detection batch with a different padding per row."""
    kps, _ = keyframe_pool()

    def pad(desc, n):
        d, m = desc.clone(), torch.zeros(desc.shape[1], dtype=torch.bool)
        if n:
            d[:, -n:] = 0.0
            m[-n:] = True
        return d, m

    s, ms = pad(kps[3], 40)
    d, md = pad(kps[4], 25)
    big, mb = pad(torch.cat([kps[0], kps[1], kps[2], kps[5]], dim=1), 100)
    rows = [pad(kps[i], n) for i, n in ((6, 0), (7, 64), (8, 200), (9, 1))]
    rows_d = [pad(kps[i], n) for i, n in ((10, 30), (10, 0), (11, 128), (12, 255))]
    return {
        "reg_256": (s[None], d[None], ms[None], md[None]),
        "reg_1024x256": (big[None], kps[13][None].clone(), mb[None], None),
        "loop4": (torch.stack([r[0] for r in rows]), torch.stack([r[0] for r in rows_d]),
                  torch.stack([r[1] for r in rows]), torch.stack([r[1] for r in rows_d])),
    }
