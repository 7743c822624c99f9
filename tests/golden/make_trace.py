#!/usr/bin/env python3
"""Record every hot-path call the REFERENCE's SlamSystem.step makes over the 11 sample frames
(SURVEY.md 8c golden item 6; reference system/core.py:360-423) -> tests/golden/slam_trace.npz.

Runs only in the build container (imports /root/reference, read-only).  The reference modules are imported with
stubs for packages the container lacks (colorlog -> logging, easydict, readerwriterlock, open3d; none of the
stubbed code paths is executed), the networks get the procedural weights, and the SLAM thresholds are loosened so
that with those weights key-frames, scan-to-map refinements (mapping.py:136-170) and loop closures
(loop_closure.py:80-258) all occur.  The drive: the 11 frames forwards, then four of them again (a revisit, so that
loop candidates exist).  Recorded per call, in call order:

  enc   : frame index -> descriptors (131, 256)                                   (odometry.py:36-54)
  reg   : src / dst descriptor matrices (as column recipes, see below), num_sample -> R, T, conf30, n_conf, rmse
  loop  : candidate key-frame tokens + the new scan's token -> probabilities     (loop_closure.py:166-174)
  tile  : PoseGraph.global_map_query_graph: ordered scan tokens, their SE3_pred, the centring -> xyz rows
  info  : calculate_information_matrix_from_pcd(src scan, dst scan, SE3) -> 6x6.  NEITHER reference branch of this
          function runs here (pytorch3d and open3d are absent): the call sites are the reference's, the VALUES come
          from the oracle restatement of utils.py:72-104 -- "parity unpinned" for this one function, as in DESIGN.md.
  optim : PoseGraph.optim() as loop_closure.py:294-307 reaches it after a verified loop edge: the graph exactly as
          __optim_open3d would hand it to open3d (pose_graph.py:565-608: key-frame tokens + SE3_pred, every non-'locz' edge
          with inv(edge.SE3) and its information matrix, reference node = smallest token).  open3d is absent, so the
          call is RECORDED and skipped: the poses stay what odometry + scan-to-map made them (which is what every other
          entry of this trace was recorded with), and the optimiser's VALUES stay unpinned.
  final : exit code per step and every scan's SE3_pred (the trajectory).

A descriptor matrix handed to registration_forward is a column-wise selection of key-frame descriptors whose xyz
rows were moved by SE(3) transforms; its feature rows are bit-copies.  It is stored as (token, column) per column
plus the xyz rows, and rebuilt in the test from the recorded key-frame descriptors.
"""
import logging
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.modules["colorlog"] = logging
sys.modules.setdefault("open3d", types.ModuleType("open3d"))
sys.path.insert(0, REF)
sys.path.insert(1, ROOT)

from deeppointmap_amd.config import Cfg, default_args  # noqa: E402
from deeppointmap_amd.params import decoder_shapes, encoder_shapes  # noqa: E402
from deeppointmap_amd.weights import procedural_state_dict  # noqa: E402
from oracle import dpm_oracle as O  # noqa: E402

ed = types.ModuleType("easydict")
ed.EasyDict = Cfg
sys.modules["easydict"] = ed
rw, rwl = types.ModuleType("readerwriterlock"), types.ModuleType("readerwriterlock.rwlock")


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
sys.modules["readerwriterlock"], sys.modules["readerwriterlock.rwlock"] = rw, rwl

from network.encoder.encoder import Encoder as RefEncoder  # noqa: E402  (reference)
from network.decoder.decoder import Decoder as RefDecoder  # noqa: E402
from dataloader.transforms import CoordinatesNormalization, DistanceSample, PointCloud, VoxelSample  # noqa: E402
import system.modules.odometry as ref_odometry  # noqa: E402
import system.modules.mapping as ref_mapping  # noqa: E402
import system.modules.loop_closure as ref_loop  # noqa: E402
from system.core import SlamSystem  # noqa: E402
from system.modules.utils import EXIT_CODE  # noqa: E402

torch.set_grad_enabled(False)
torch.manual_seed(0)


def preprocessed_frame(i):
    raw = np.fromfile(f"{REF}/data/sample/seq06/velodyne/{i:06d}.bin", dtype=np.float32).reshape(-1, 4)[:, :3]
    pcd = PointCloud(raw.copy())
    for tf in (VoxelSample(voxel_size=0.3, retention="first"), DistanceSample(min_dis=1.0, max_dis=60.0),
               CoordinatesNormalization(ratio=60.0)):
        pcd = tf(pcd)
    xyz = pcd.xyz if isinstance(pcd.xyz, torch.Tensor) else torch.from_numpy(pcd.xyz)
    return xyz.float().t().contiguous()  # (3, N)


def main(variant="all"):
    """variant "all": every scan a key-frame (slam_trace.npz, with every array needed to replay each call on its own);
    variant "gated": thresholds under which scans are also DROPPED (the third drop in a row recovered), LOCALISED without
    becoming key-frames, and scan-to-map results refused (slam_trace_gated.npz: decisions, poses and per-call results only --
    the scans are those of slam_trace.npz)."""
    cfg = default_args()
    cfg.device = "cpu"
    cfg.use_ros = False
    cfg.infer_tgt = "/tmp/dpm_trace_log"
    cfg.slam_system = Cfg(dict(
        coor_scale=60, odometer_candidates_num=1, registration_sample_odometer=0.5,
        edge_confidence_drop=0.0, edge_rmse_drop=1e9, max_continuous_drop_scan=5, continuous_drop_scan_strategy="recover",
        key_frame_distance=0.0, enable_s2m_adjust=True, registration_sample_mapping=0.5,
        enable_loop_closure=True, loop_detection_gap=0, loop_detection_transaction_gap=0.0, loop_detection_trust_range=3,
        loop_detection_gnss_distance=-1, loop_detection_pred_distance=1e9, loop_detection_rotation_min=0.0,
        loop_detection_translation_min=0.0, loop_detection_prob_acpt_threshold=0.0, loop_detection_candidates_num=1,
        registration_sample_loop=0.5, loop_detection_confidence_acpt_threshold=0.0,
        enable_global_optimization=True, global_optimization_gap=0))
    if variant == "gated":
        cfg.slam_system.update(edge_rmse_drop=1.6, max_continuous_drop_scan=3, key_frame_distance=0.2, loop_detection_trust_range=2)
    enc, dec = RefEncoder(cfg).eval(), RefDecoder(cfg).eval()
    enc.load_state_dict(procedural_state_dict(encoder_shapes(cfg)), strict=True)
    dec.load_state_dict(procedural_state_dict(decoder_shapes(cfg)), strict=True)

    def info_from_oracle(p1, p2, SE3, device="cpu"):
        return O.information_matrix(p1, p2, SE3)
    for m in (ref_odometry, ref_mapping, ref_loop):
        m.calculate_information_matrix_from_pcd = lambda p1, p2, SE3, device="cpu": record_info(p1, p2, SE3)

    frames = [preprocessed_frame(i) for i in range(11)]
    order = list(range(11)) + ([7, 4, 2, 0] if variant == "all" else [9, 10, 7, 4, 2, 0, 3, 8, 5])
    calls, out = [], {}
    kf_desc = {}          # scan token -> descriptors (131, 256)
    col_of = {}           # feature-column bytes -> (token, column)
    pcd_token = {}        # id of a full_pcd tensor's storage -> scan token
    cur = {"step": -1, "token": None}

    def recipe(desc):
        tok = np.empty(desc.shape[1], np.int32)
        col = np.empty(desc.shape[1], np.int16)
        fea = desc[:128].t().contiguous().numpy()
        for j in range(desc.shape[1]):
            tok[j], col[j] = col_of[fea[j].tobytes()]
        return tok, col, desc[128:131].clone()

    def record_info(p1, p2, SE3):
        G = info_from_oracle(p1, p2, SE3)
        k = len(calls)
        calls.append(("info", k))
        out[f"c{k}.src"], out[f"c{k}.dst"] = pcd_token[p1.data_ptr()], pcd_token[p2.data_ptr()]
        out[f"c{k}.SE3"], out[f"c{k}.G"] = SE3.clone(), G
        return G

    system = SlamSystem(cfg, enc, dec, system_id=0, logger_dir="/tmp/dpm_trace_log", device="cpu")
    pg = system.posegraph_map

    reg0, loop0, tile0 = dec.registration_forward, dec.loop_detection_forward, pg.global_map_query_graph

    def reg(src, dst, src_padding_mask=None, dst_padding_mask=None, num_sample=0.5):
        R, T, conf, rmse = reg0(src, dst, src_padding_mask, dst_padding_mask, num_sample)
        k = len(calls)
        calls.append(("reg", k))
        for side, d in (("src", src), ("dst", dst)):
            tok, col, xyz = recipe(d.cpu())
            out[f"c{k}.{side}_tok"], out[f"c{k}.{side}_col"], out[f"c{k}.{side}_xyz"] = tok, col, xyz
        out[f"c{k}.num_sample"] = float(num_sample)
        out[f"c{k}.R"], out[f"c{k}.T"], out[f"c{k}.rmse"] = R.clone(), T.clone(), float(rmse)
        out[f"c{k}.n_conf"], out[f"c{k}.conf30"] = conf.numel(), float(conf.flatten()[:30].mean())
        print(f"  reg {tuple(src.shape)} x {tuple(dst.shape)}: rmse {float(rmse):.3f} n_conf {conf.numel()}", flush=True)
        return R, T, conf, rmse

    def loop(src, dst):
        p = loop0(src, dst)
        k = len(calls)
        calls.append(("loop", k))
        toks = []
        for c in range(src.shape[0]):
            tok, col, _ = recipe(src[c].cpu())
            assert (tok == tok[0]).all() and (col == np.arange(256)).all()
            toks.append(int(tok[0]))
        out[f"c{k}.src_tokens"], out[f"c{k}.dst_token"], out[f"c{k}.prob"] = np.array(toks, np.int32), cur["token"], p.clone()
        print(f"  loop detection over {src.shape[0]} candidates", flush=True)
        return p

    def tile(*a, **kw):
        t, tok = tile0(*a, **kw)
        k = len(calls)
        calls.append(("tile", k))
        order_tok = [int(x) for x in tok[::256].tolist()]
        assert (tok.view(-1, 256) == tok[::256].unsqueeze(1)).all()
        out[f"c{k}.tokens"] = np.array(order_tok, np.int32)
        out[f"c{k}.SE3"] = torch.stack([pg.get_scanpack(x).SE3_pred for x in order_tok])
        out[f"c{k}.centering"] = kw["centering_SE3"].clone()
        out[f"c{k}.xyz"] = t[128:131].clone()
        return t, tok

    dec.registration_forward, dec.loop_detection_forward, pg.global_map_query_graph = reg, loop, tile

    def optim(blocking=True):
        k = len(calls)
        calls.append(("optim", k))
        scans = [s for s in pg.get_all_scans() if s.type != "non-keyframe"]
        toks = [s.token for s in scans]
        edges = [e for e in pg.get_all_edges() if e.type != "locz" and e.src_scan_token in toks and e.dst_scan_token in toks]
        out[f"c{k}.tokens"] = np.array(toks, np.int32)
        out[f"c{k}.SE3"] = torch.stack([s.SE3_pred for s in scans])
        out[f"c{k}.edge_src"] = np.array([e.src_scan_token for e in edges], np.int32)
        out[f"c{k}.edge_dst"] = np.array([e.dst_scan_token for e in edges], np.int32)
        out[f"c{k}.edge_type"] = np.array([e.type for e in edges])
        out[f"c{k}.edge_T"] = torch.stack([torch.linalg.inv(e.SE3) for e in edges])         # pose_graph.py:593
        out[f"c{k}.edge_info"] = torch.stack([torch.as_tensor(e.information_mat).float() for e in edges])
        out[f"c{k}.reference"] = min(toks)
        print(f"  optim() on {len(toks)} key-frames, {len(edges)} edges ({sum(e.type == 'loop' for e in edges)} loop)", flush=True)
        return len(toks), len(edges), 0.0
    pg.optim = optim

    def enc_hook(mod, args, res):
        coor, fea, _ = res
        d = torch.cat([fea[0], coor[0] * 60.0], 0)
        k = len(calls)
        calls.append(("enc", k))
        out[f"c{k}.frame"], out[f"c{k}.desc"] = cur["frame"], d.clone()
    enc.register_forward_hook(enc_hook)

    add_vertex0 = pg.add_vertex

    def add_vertex(scan):
        r = add_vertex0(scan)
        if scan.key_points is not None and scan.token not in kf_desc:
            kf_desc[scan.token] = scan.key_points.clone()
            fea = scan.key_points[:128].t().contiguous().numpy()
            for j in range(fea.shape[0]):
                col_of.setdefault(fea[j].tobytes(), (scan.token, j))
        return r
    pg.add_vertex = add_vertex

    import system.core as ref_core
    ScanPack0 = ref_core.ScanPack

    def ScanPackRec(*a, **kw):
        sp = ScanPack0(*a, **kw)
        cur["token"] = sp.token
        if sp.full_pcd is not None:
            pcd_token[sp.full_pcd.data_ptr()] = sp.token
        if sp.key_points is not None:  # the new scan takes part in registrations before it becomes a vertex
            fea = sp.key_points[:128].t().contiguous().numpy()
            for j in range(fea.shape[0]):
                col_of.setdefault(fea[j].tobytes(), (sp.token, j))
            kf_desc.setdefault(sp.token, sp.key_points.clone())
        return sp
    ref_core.ScanPack = ScanPackRec

    codes = []
    for step, f in enumerate(order):
        cur["step"], cur["frame"] = step, f
        pts = frames[f]
        data = [pts.unsqueeze(0), torch.eye(3).unsqueeze(0), torch.zeros(1, 3, 1),
                torch.zeros(1, pts.shape[1], dtype=torch.bool), None]
        k0 = len(calls)
        code = system.step(data)
        codes.append(code.value if isinstance(code, EXIT_CODE) else -1)
        out[f"s{step}.frame"], out[f"s{step}.token"] = f, cur["token"]
        out[f"s{step}.calls"] = np.array([k0, len(calls)], np.int32)
        print(f"step {step} frame {f}: exit {code}, calls {[c[0] for c in calls[k0:]]}", flush=True)
    for i, p in enumerate(frames):
        out[f"frame{i}"] = p
    out["order"], out["codes"] = np.array(order, np.int32), np.array(codes, np.int32)
    out["call_kinds"] = np.array([c[0] for c in calls])
    toks = sorted(kf_desc)
    out["desc_tokens"] = np.array(toks, np.int32)
    out["desc"] = torch.stack([kf_desc[t] for t in toks])
    scans = sorted(pg.get_all_scans(), key=lambda s: s.token)
    out["final_tokens"] = np.array([s.token for s in scans], np.int32)
    out["final_SE3"] = torch.stack([s.SE3_pred for s in scans])
    out["final_type"] = np.array([s.type for s in scans])
    path = os.path.join(HERE, "slam_trace.npz")
    if variant == "gated":
        import json
        path = os.path.join(HERE, "slam_trace_gated.npz")
        big = ("_xyz", "_col", ".xyz", ".desc", ".centering", ".SE3")
        out = {k: v for k, v in out.items() if not (k.startswith("frame") or k == "desc" or (k.startswith("c") and k.endswith(big)))}
        for k in [k for k in out if k.endswith("_tok")]:        # one token per 256-column block
            out[k] = np.asarray(out[k])[::256].copy()
        out["slam_args"] = json.dumps(dict(cfg.slam_system))
    np.savez_compressed(path, **{k: (v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in out.items()})
    kinds = [c[0] for c in calls]
    print(f"{os.path.basename(path)}: {os.path.getsize(path) / 1024:.0f} KiB, {len(calls)} calls: " +
          ", ".join(f"{k} x{kinds.count(k)}" for k in ("enc", "reg", "loop", "tile", "info", "optim")))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "all")
