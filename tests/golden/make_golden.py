#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by running THE REFERENCE ITSELF.

Runs only in the build container (needs /root/reference, read-only).  The reference is a
Python package that cannot travel to the GPU box, so its inputs/outputs on seeded data are
committed as small .npz fixtures; tests compare the oracle (oracle/dpm_oracle.py) and the HIP
path against them.  Nothing here is imported by tests or by the product.

    python tests/golden/make_golden.py            # rewrites tests/golden/*.npz

Stubs: `colorlog` -> logging (only use: network/encoder/utils.py:7-9).  pytorch3d is absent, so
the reference auto-selects its pure-torch `fps` / `hybrid` operators
(network/encoder/utils.py:29-39,134-144) -- "the reference CPU path" of BASELINE.json.
"""
import logging
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.modules["colorlog"] = logging
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

from network.encoder.encoder import Encoder as RefEncoder  # noqa: E402  (reference)
from network.decoder.decoder import Decoder as RefDecoder  # noqa: E402  (reference)
from network.encoder.utils import Querier as RefQuerier, Sampler as RefSampler  # noqa: E402
from network.decoder.descriptor_attention import PositionEmbeddingCoordsSine as RefPosEmb  # noqa: E402

from deeppointmap_amd import synthetic  # noqa: E402
from deeppointmap_amd.config import default_args, reduced_args  # noqa: E402
from deeppointmap_amd.params import decoder_shapes, encoder_shapes  # noqa: E402
from deeppointmap_amd.weights import procedural_state_dict  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def save(name, **arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        out[k] = np.asarray(v)
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB, {len(out)} arrays")


def build_reference(cfg):
    enc, dec = RefEncoder(cfg).eval(), RefDecoder(cfg).eval()
    # strict=True on BOTH: proves our key/shape tables equal the reference's state dict
    enc.load_state_dict(procedural_state_dict(encoder_shapes(cfg)), strict=True)
    dec.load_state_dict(procedural_state_dict(decoder_shapes(cfg)), strict=True)
    return enc, dec


def kitti_sample(i, n=16384):
    """sample frame i: 1-60 m crop, every-k-th subsample to n points, /60 -> (3,n) f32."""
    raw = np.fromfile(f"{REF}/data/sample/seq06/velodyne/{i:06d}.bin", dtype=np.float32).reshape(-1, 4)[:, :3]
    d = np.linalg.norm(raw, axis=1)
    raw = raw[(d >= 1.0) & (d <= 60.0)]
    sel = np.linspace(0, raw.shape[0] - 1, n).astype(np.int64)
    return torch.from_numpy(np.ascontiguousarray((raw[sel] / 60.0).T.astype(np.float32)))


# ---------------------------------------------------------------------------------------------
def gen_fps():
    fps = RefSampler("fps")
    out = {}
    g = torch.Generator().manual_seed(11)
    cases = {
        "n64_k16": (torch.rand(1, 64, 3, generator=g) * 2 - 1, 64, 16),
        "n1000_k256": (torch.rand(1, 1000, 3, generator=g) * 2 - 1, 1000, 256),
        "pad_n1000_len700_k256": (torch.rand(1, 1000, 3, generator=g) * 2 - 1, 700, 256),
        "pad_n300_len100_k128": (torch.rand(1, 300, 3, generator=g) * 2 - 1, 100, 128),  # lengths < K -> -1
        "grid_ties_n512_k64": (torch.stack(torch.meshgrid(*[torch.arange(8.0)] * 3, indexing="ij"), -1).reshape(1, 512, 3) / 8, 512, 64),
        "kitti0_k4096": (kitti_sample(0).t().unsqueeze(0), 16384, 4096),
    }
    for name, (pts, length, K) in cases.items():
        pad = torch.arange(pts.shape[1]).unsqueeze(0) >= length
        new, mask = fps(points=pts, points_padding=pad, K=K)
        out[name + ".points"] = pts[0]
        out[name + ".length"] = length
        out[name + ".new"] = new[0]
        out[name + ".mask"] = mask[0]
    # full-size synthetic frame: the input is regenerated from synthetic.py, only the answer is stored
    pts = synthetic.frame(0).t().unsqueeze(0)
    new, mask = fps(points=pts, points_padding=torch.zeros(1, 65536, dtype=torch.bool), K=4096)
    out["synthetic0_k4096.new"] = new[0]
    save("fps.npz", **out)


def gen_knn():
    hyb = RefQuerier("hybrid")
    out = {}
    pts = synthetic.frame(0, 8192).t().unsqueeze(0)
    g = torch.Generator().manual_seed(5)
    cases = {"sa_r0.05_k32": (0.05, 32, 1024, 8192), "la_r0.1_k32": (0.1, 32, 1024, 1024),
             "la_r1.6_k16": (1.6, 16, 16, 16), "pad_r0.2_k32": (0.2, 32, 256, 700)}
    for name, (r, K, S, length) in cases.items():
        p = pts[:, :max(length, S)] if name.startswith("pad") else (pts if length == 8192 else pts[:, :length])
        pad = torch.arange(p.shape[1]).unsqueeze(0) >= length
        centers = p[:, torch.randperm(min(length, p.shape[1]), generator=g)[:S]]
        idx = hyb(radius=r, K=K, points=p, centers=centers, points_padding=pad)
        out[name + ".points"] = p[0]
        out[name + ".centers"] = centers[0]
        out[name + ".length"] = length
        out[name + ".radius"] = r
        out[name + ".idx"] = idx[0].to(torch.int32)
    save("knn.npz", **out)


def trace_reference_encoder(enc, pts, pad):
    """Run the reference encoder recording per-stage tensors through hooks / wrapped operators."""
    rec = {}

    def wrap(obj, attr, key, pick):
        orig = getattr(obj, attr)

        def f(*a, **k):
            r = orig(*a, **k)
            rec[key] = pick(r).clone()
            return r
        setattr(obj, attr, f)
        return lambda: setattr(obj, attr, orig)

    undo = []
    for i, st in enumerate(enc.downsampler):
        undo.append(wrap(st.sa, "sample", f"downsampler.{i}.fps.new", lambda r: r[0]))
        undo.append(wrap(st.sa, "query", f"downsampler.{i}.sa.idx", lambda r: r.to(torch.int32)))
        h = st.sa.register_forward_hook(lambda m, a, r, i=i: rec.__setitem__(f"downsampler.{i}.sa.out", r[1].transpose(1, 2).clone()))
        undo.append(h.remove)
        if not isinstance(st.irm, torch.nn.Identity):
            for j, irm in enumerate(st.irm):
                undo.append(wrap(irm.la, "query", f"downsampler.{i}.irm.{j}.la.idx", lambda r: r.to(torch.int32)))
                h = irm.la.register_forward_hook(lambda m, a, r, i=i, j=j: rec.__setitem__(f"downsampler.{i}.irm.{j}.la.out", r.transpose(1, 2).clone()))
                undo.append(h.remove)
                h = irm.register_forward_hook(lambda m, a, r, i=i, j=j: rec.__setitem__(f"downsampler.{i}.irm.{j}.out", r[1].transpose(1, 2).clone()))
                undo.append(h.remove)
    for i, up in enumerate(enc.upsampler):
        h = up.register_forward_hook(lambda m, a, r, i=i: rec.__setitem__(f"upsampler.{i}.out", r.transpose(1, 2).clone()))
        undo.append(h.remove)
    coor, fea, mask = enc(pts, pad)
    for u in undo:
        u()
    rec["coor"], rec["fea"], rec["mask"] = coor, fea, mask
    return rec


def gen_encoder():
    # (1) reduced topology, per-stage trace, two frames in one batch + a padded frame
    cfg = reduced_args()
    enc, _ = build_reference(cfg)
    pts, pad = synthetic.frames(2, 4096)
    rec = trace_reference_encoder(enc, pts, pad)
    save("encoder_reduced.npz", points=pts, **rec)
    pts1 = pts[:1].clone()
    pad1 = torch.arange(4096).unsqueeze(0) >= 3000
    pts1[:, :, 3000:] = 0.0
    rec = trace_reference_encoder(enc, pts1, pad1)
    save("encoder_reduced_padded.npz", points=pts1, length=3000, **rec)

    # (2) shipped topology: final descriptors only
    cfg = default_args()
    enc, _ = build_reference(cfg)
    out = {}
    for f in (0, 1):
        p = synthetic.frame(f).unsqueeze(0)
        coor, fea, mask = enc(p, torch.zeros(1, p.shape[2], dtype=torch.bool))
        out[f"synthetic{f}.coor"], out[f"synthetic{f}.fea"] = coor[0], fea[0]
    for f in (0, 1):
        p = kitti_sample(f).unsqueeze(0)
        coor, fea, mask = enc(p, torch.zeros(1, p.shape[2], dtype=torch.bool))
        out[f"kitti{f}.points"] = p[0]
        out[f"kitti{f}.coor"], out[f"kitti{f}.fea"] = coor[0], fea[0]
    save("encoder_full.npz", **out)
    return out


def gen_decoder(enc_out):
    cfg = default_args()
    _, dec = build_reference(cfg)
    out = {}
    xyz = torch.tensor([[0.0, 0.0, 0.0], [1.0, -2.0, 0.5], [59.5, 12.25, -3.0], [-40.0, 0.125, 2.0],
                        [7.0, 7.0, 7.0], [-0.001, 100.0, 1e-3], [3.14159, -2.71828, 1.41421], [25.0, -25.0, 0.0]])
    out["posemb.xyz"] = xyz
    out["posemb.out"] = RefPosEmb(3, 256)(xyz.t().unsqueeze(0))[0].t()

    def desc(tag):
        return torch.cat([enc_out[tag + ".fea"], enc_out[tag + ".coor"] * cfg.slam_system.coor_scale], dim=0)

    pairs = {"synthetic01": (desc("synthetic0"), desc("synthetic1")), "kitti01": (desc("kitti0"), desc("kitti1"))}
    # map-vs-scan shape (M != N): 4 noisy copies of frame 0's descriptors against frame 1
    g = torch.Generator().manual_seed(3)
    d0 = desc("synthetic0")
    big = torch.cat([d0 + torch.cat([0.01 * torch.randn(128, 256, generator=g).abs(), 0.05 * torch.randn(3, 256, generator=g)]) for _ in range(4)], dim=1)
    pairs["map1024_vs_256"] = (big, desc("synthetic1"))
    for name, (s, d) in pairs.items():
        sc, dc = dec._descriptor_attention_forward(s.unsqueeze(0), d.unsqueeze(0))
        ps, pd, conf = dec._descriptor_pairing(sc, dc, 0.5)
        src, dst, w = dec._get_corres_sets(ps, pd, conf)
        R, T, c, rmse = dec.registration_forward(s, d, num_sample=0.5)
        out[name + ".src_desc"], out[name + ".dst_desc"] = s, d
        out[name + ".src_corr"], out[name + ".dst_corr"] = sc[0], dc[0]
        out[name + ".pair_src_xyz"], out[name + ".pair_dst_xyz"], out[name + ".pair_conf"] = ps[0, -3:], pd[0, -3:], conf[0]
        out[name + ".corr_src"], out[name + ".corr_dst"], out[name + ".corr_w"] = src[0], dst[0], w[0]
        out[name + ".R"], out[name + ".T"], out[name + ".conf"], out[name + ".rmse"] = R, T, c, rmse
    # loop detection on a batch of 3 candidate pairs
    S = torch.stack([desc("synthetic0"), desc("kitti0"), desc("synthetic1")])
    D = torch.stack([desc("synthetic1"), desc("kitti1"), desc("kitti0")])
    out["loop.src"], out["loop.dst"] = S, D
    out["loop.prob"] = dec.loop_detection_forward(S, D)

    # hand-built correspondence sets for the Kabsch loop
    g = torch.Generator().manual_seed(9)

    def corr_case(n, noise, outliers, reflect=False, wlo=0.0):
        src = torch.randn(3, n, generator=g) * 10
        a = 0.3
        R = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
        if reflect:
            R = R @ torch.diag(torch.tensor([1.0, 1.0, -1.0]))
        dst = R @ src + torch.tensor([[1.0], [-2.0], [0.5]]) + noise * torch.randn(3, n, generator=g)
        dst[:, :outliers] += 5 * torch.randn(3, outliers, generator=g)
        w = wlo + (1 - wlo) * torch.rand(n, generator=g)
        return w, src, dst

    for name, args in {"svd_clean200": (200, 0.01, 0), "svd_outliers300": (300, 0.05, 40), "svd_few40": (40, 0.02, 8),
                       "svd_reflect120": (120, 0.01, 0, True), "svd_lowconf100": (100, 0.02, 10, False, 0.0)}.items():
        w, src, dst = corr_case(*args)
        if name == "svd_lowconf100":
            w = w * 0.4
        Rs, Ts, masks, rmses = RefDecoder._solve_transformation_SVD(w.unsqueeze(0), src.unsqueeze(0), dst.unsqueeze(0))
        out[name + ".w"], out[name + ".src"], out[name + ".dst"] = w, src, dst
        out[name + ".R"], out[name + ".T"], out[name + ".mask"], out[name + ".rmse"] = Rs[0], Ts[0], masks[0], rmses[0]
    save("decoder.npz", **out)


def gen_poses_full(n_frames=6):
    """Consecutive-frame registrations at BASELINE.json's full size (65 536 points), reference end to end:
    only the poses are stored; the inputs are regenerated from synthetic.py."""
    cfg = default_args()
    enc, dec = build_reference(cfg)
    descs = []
    for f in range(n_frames):
        p = synthetic.frame(f).unsqueeze(0)
        coor, fea, _ = enc(p, torch.zeros(1, p.shape[2], dtype=torch.bool))
        descs.append(torch.cat([fea[0], coor[0] * cfg.slam_system.coor_scale], dim=0))
    out = {}
    for f in range(1, n_frames):
        R, T, c, rmse = dec.registration_forward(descs[f - 1], descs[f], num_sample=0.5)
        out[f"pair{f - 1}_{f}.R"], out[f"pair{f - 1}_{f}.T"] = R, T
        out[f"pair{f - 1}_{f}.rmse"], out[f"pair{f - 1}_{f}.n_conf"] = rmse, c.shape[0]
        out[f"pair{f - 1}_{f}.conf30"] = c.flatten()[:30].mean()
    # num_sample variants on one pair (decoder.py:170-178)
    for tag, ns in (("int100", 100), ("float300", 300.0), ("float0.25", 0.25)):
        R, T, c, rmse = dec.registration_forward(descs[0], descs[1], num_sample=ns)
        out[f"ns_{tag}.R"], out[f"ns_{tag}.T"], out[f"ns_{tag}.n_conf"] = R, T, c.shape[0]
    save("poses_full.npz", **out)


def gen_maptile():
    """PoseGraph.global_map_query_graph (system/modules/pose_graph.py:471-511) on a 6-scan chain: the reference's own
    map-tile assembly (per-scan SE3 transform of the descriptor coordinates, concat in BFS order, re-centring).
    Needs import stubs for packages the container lacks (open3d, readerwriterlock); none of them is executed."""
    import types
    o3d = types.ModuleType("open3d")
    sys.modules.setdefault("open3d", o3d)
    rw = types.ModuleType("readerwriterlock")
    rwl = types.ModuleType("readerwriterlock.rwlock")

    class _L:
        def acquire(self, blocking=True):
            return True

        def release(self):
            pass

    class RWLockFair:
        def gen_rlock(self):
            return _L()

        def gen_wlock(self):
            return _L()

    rwl.RWLockFair = RWLockFair
    rw.rwlock = rwl
    sys.modules.setdefault("readerwriterlock", rw)
    sys.modules.setdefault("readerwriterlock.rwlock", rwl)
    from system.modules.pose_graph import PoseGraph, PoseGraph_Edge, ScanPack  # noqa: E402  (reference)

    g = torch.Generator().manual_seed(21)
    enc = np.load(os.path.join(HERE, "encoder_full.npz"))
    base = torch.cat([torch.from_numpy(enc["synthetic0.fea"]), torch.from_numpy(enc["synthetic0.coor"]) * 60.0], 0)
    pg = PoseGraph(args=None, agent_id=0, device="cpu")
    kps, poses = [], []
    for i in range(6):
        kp = base + torch.cat([0.01 * torch.rand(128, 256, generator=g), 0.5 * torch.randn(3, 256, generator=g)])
        a = 0.3 * i
        SE3 = torch.eye(4)
        SE3[:3, :3] = torch.tensor([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1]], dtype=torch.float32)
        SE3[:3, 3] = torch.tensor([4.0 * i, 1.5 * i * (-1) ** i, 0.1 * i])
        sp = ScanPack(timestamp=float(i), agent_id=0, timestep=i, key_points=kp, full_pcd=None, SE3_pred=SE3, coor_sys=0)
        if i == 4:
            sp = sp.nonkeyframe()  # must be skipped by the query
        pg.add_vertex(sp)
        kps.append(kp)
        poses.append(SE3)
    for i in range(5):
        pg.add_edge(PoseGraph_Edge(i, i + 1, torch.eye(4), torch.eye(6), "odom" if i != 2 else "loop"))
    center = poses[2].clone()
    tile, tok = pg.global_map_query_graph(token=2, neighbor_level=5, coor_sys=0, max_dist=11.0, centering_SE3=center)
    save("maptile.npz", key_points=torch.stack(kps), SE3=torch.stack(poses), centering=center, tile=tile, tokens=tok)


sys.path.insert(0, HERE)
from raw_scan import raw_scan  # noqa: E402  (tests/golden/raw_scan.py: the seeded synthetic raw scan)


def gen_preprocess():
    """VoxelSample(0.3,'first') -> DistanceSample(1,60) -> CoordinatesNormalization(60) with the reference's own
    transform classes (dataloader/transforms.py; open3d stubbed, never executed)."""
    import types
    sys.modules.setdefault("open3d", types.ModuleType("open3d"))
    from dataloader.transforms import CoordinatesNormalization, DistanceSample, PointCloud, VoxelSample  # noqa: E402
    out = {}
    enc = np.load(os.path.join(HERE, "encoder_full.npz"))
    cases = {"raw120k": raw_scan(), "kitti0_m": torch.from_numpy(enc["kitti0.points"]).t().contiguous() * 60.0}
    for name, xyz in cases.items():
        pcd = PointCloud(xyz.numpy().copy())
        for tf in (VoxelSample(voxel_size=0.3, retention="first"), DistanceSample(min_dis=1.0, max_dis=60.0),
                   CoordinatesNormalization(ratio=60.0)):
            pcd = tf(pcd)
        out[name + ".out"] = pcd.xyz
        if name != "raw120k":
            out[name + ".in"] = xyz
    save("preprocess.npz", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["fps", "knn", "encoder", "decoder", "poses", "maptile", "preprocess"]
    if "preprocess" in which:
        gen_preprocess()
    if "maptile" in which:
        gen_maptile()
    if "poses" in which:
        gen_poses_full()
    if "fps" in which:
        gen_fps()
    if "knn" in which:
        gen_knn()
    if "encoder" in which or "decoder" in which:
        enc_out = gen_encoder()
        gen_decoder(enc_out)
