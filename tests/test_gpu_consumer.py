"""deeppointmap_amd/consumer.py::Rank0Consumer -- the reference's MappingThread rules on gathered edge rows -- driven by the
recorded SlamSystem.step run (tests/golden/slam_trace.npz): the same scans in the same order, the reference's own odometry
edges as the gathered rows, its verified loop edges added when it added them.  The consumer must then make the reference's
decisions: every scan accepted as a key-frame (the trace's thresholds), every scan-to-map tile made of the same key-frames in
the same order (PoseGraph.graph_search over the same graph), and -- its tiles and registrations being the device path's -- the
reference's trajectory."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden, rot_angle

pytestmark = pytest.mark.gpu


def test_consumer_makes_the_reference_s_mapping_decisions_on_the_trace(cfg_full):
    from deeppointmap_amd import ops
    from deeppointmap_amd.consumer import Rank0Consumer
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    g = load_golden("slam_trace.npz")
    kinds = [str(k) for k in g["call_kinds"]]
    desc = {int(t): T(g["desc"][i]) for i, t in enumerate(g["desc_tokens"])}
    dev = torch.device("cuda:0")
    dec = init_procedural(Decoder(cfg_full)).to(dev)
    # thresholds of tests/golden/make_trace.py: nothing dropped, key_frame_distance 0 (every accepted scan is a key-frame)
    cons = Rank0Consumer(dec, dev, slam_args=dict(edge_confidence_drop=0.0, edge_rmse_drop=1e9, key_frame_distance=0.0))
    loops_seen = set()
    for s in range(len(g["order"])):
        tok = int(g[f"s{s}.token"])
        k0, k1 = g[f"s{s}.calls"]
        row = torch.zeros(ops.RES_HDR + 36)
        regs = [k for k in range(k0, k1) if kinds[k] == "reg"]
        if regs:
            k = regs[0]                                  # the odometry registration of this step and its information matrix
            assert kinds[k + 1] == "info" and int(g[f"c{k + 1}.dst"]) == tok
            row[0:9], row[9:12] = T(g[f"c{k}.R"]).reshape(9), T(g[f"c{k}.T"]).reshape(3)
            row[12], row[14], row[16] = float(g[f"c{k}.rmse"]), float(g[f"c{k}.n_conf"]), float(g[f"c{k}.conf30"])
            row[ops.RES_HDR:] = T(g[f"c{k + 1}.G"]).reshape(36)
        assert cons.push(tok, desc[tok].to(dev), row) == "acpt"
        if regs:
            tiles = [k for k in range(k0, k1) if kinds[k] == "tile"]
            assert cons.tiles[-1] == [int(t) for t in g[f"c{tiles[0]}.tokens"]], s   # graph_search order, scan by scan
        for k in range(k0, k1):                          # the loop edges the reference verified in this step
            if kinds[k] != "optim":
                continue
            for a, b, ty, X, info in zip(g[f"c{k}.edge_src"], g[f"c{k}.edge_dst"], g[f"c{k}.edge_type"], g[f"c{k}.edge_T"],
                                         g[f"c{k}.edge_info"]):
                if str(ty) == "loop" and (int(a), int(b)) not in loops_seen:
                    loops_seen.add((int(a), int(b)))
                    cons.add_loop_edge(int(a), int(b), torch.linalg.inv(T(X)), information=T(info))
    assert cons.codes == ["acpt"] * len(g["order"]) and cons.stats["s2m"] == len(g["order"]) - 1
    assert [cons.type[int(t)] for t in g["final_tokens"]] == [str(x) for x in g["final_type"]]
    final = {int(t): T(x) for t, x in zip(g["final_tokens"], g["final_SE3"])}
    dev_t = max(float((cons.poses[t][:3, 3] - final[t][:3, 3]).norm()) for t in final)
    dev_r = max(rot_angle(cons.poses[t][:3, :3], final[t][:3, :3].numpy()) for t in final)
    assert dev_t < 1e-4 and dev_r < 1e-4, (dev_t, dev_r)
    # the odometry edges ended up with the scan-to-map values (mapping.py:196-200), the loop edges are the reference's
    assert sum(e["type"] == "loop" for e in cons.edges.values()) == len(loops_seen) == 8


def test_consumer_gating_rules():
    """valid_check and keyframe_check on hand-made edge rows: a scan that moved less than key_frame_distance is localised
    but not mapped, edges below the confidence / above the rmse threshold are dropped, the fifth drop in a row recovers the
    best of the bag (mapping.py:52-83), and the edge of a scan behind a non-key-frame is the product of the consecutive
    edges (module header)."""
    from deeppointmap_amd import ops
    from deeppointmap_amd.config import default_args
    from deeppointmap_amd.consumer import Rank0Consumer
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    dev = torch.device("cuda:0")
    dec = init_procedural(Decoder(default_args())).to(dev)
    gen = torch.Generator().manual_seed(0)
    cons = Rank0Consumer(dec, dev, slam_args=dict(edge_confidence_drop=0.5, edge_rmse_drop=1.0, key_frame_distance=3.0,
                                                  enable_s2m_adjust=False))

    def frame(dx, conf=0.9, rmse=0.2):
        d = torch.cat([torch.rand(128, 256, generator=gen), 60 * torch.randn(3, 256, generator=gen)]).to(dev)
        row = torch.zeros(ops.RES_HDR + 36)
        row[0:9] = torch.eye(3).reshape(9)
        row[9] = -dx                                     # registration takes source points into the new frame: x -> x - dx
        row[12], row[16] = rmse, conf
        return d, row

    moves = [0.0, 1.0, 1.0, 1.5, 1.0, 1.0, 1.0, 1.0]
    codes = [cons.push(i, *frame(m)) for i, m in enumerate(moves)]
    # 0 first scan; 1, 2: 1 m and 2 m from key-frame 0 -> non-key-frames; 3: 3.5 m -> key-frame; 4, 5: 1, 2 m from it; 6: 3 m ...
    assert codes == ["acpt", "dist", "dist", "acpt", "dist", "dist", "acpt", "dist"]
    assert cons.keyframes == [0, 3, 6] and abs(float(cons.poses[6][0, 3]) - 6.5) < 1e-5 and abs(float(cons.poses[7][0, 3]) - 7.5) < 1e-5
    # four bad edges are dropped, the fifth recovers the one with the smallest rmse
    n = len(moves)
    bad = [cons.push(n + i, *frame(4.0, conf=0.1, rmse=2.0 - 0.1 * (i == 2))) for i in range(5)]
    assert bad[:4] == ["drop"] * 4 and bad[4] in ("acpt", "dist") and cons.stats["dropped"] == 4
    # (scans behind a dropped one are registered against the key-frame on this rank, as the reference would: their rmse is
    # whatever the decoder makes of the random descriptors, so WHICH scan of the bag is recovered is not prescribed here)
    assert len([t for t in range(n, n + 5) if t in cons.type]) == 1 and cons.stats["re_registrations"] == 4
