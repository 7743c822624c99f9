"""The pose-graph queries of deeppointmap_amd/consumer.py against the reference's PoseGraph on random graphs
(tests/golden/graph_cases.json, made by tests/golden/make_graph_cases.py: 12 graphs of one or three agents with key-frames,
non-key-frames and loop edges; 900 queries).  Host logic: no GPU."""
import json
import os

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ALL = ("loop", "odom", "locz", "prxy")


def _graph(case):
    from deeppointmap_amd.consumer import Rank0Consumer
    c = Rank0Consumer(decoder=None, device="cpu")
    for s in case["scans"]:
        c.poses[s["token"]] = torch.tensor(s["SE3"])
        c._add_vertex(s["token"], s["type"], s["coor"])
    for a, b, ty, E, info in case["edges"]:
        c._add_edge(a, b, dict(src=a, dst=b, SE3=torch.tensor(E), type=ty, information=torch.tensor(info), confidence=1.0, rmse=0.0))
    return c


def test_graph_queries_equal_the_reference():
    cases = json.load(open(os.path.join(HERE, "golden", "graph_cases.json")))
    n = 0
    for case in cases:
        c = _graph(case)
        for q in case["queries"]:
            kinds = ALL if q.get("kinds") == "all" else tuple(q.get("kinds") or ())
            if q["fn"] == "graph_search":
                got = c.graph_search(q["token"], q["level"], kinds, q["max_k"])
            elif q["fn"] == "shortest":
                got = c.shortest_path_length(q["src"], q["dst"], kinds, q["inf"])
            else:
                got = c.map_tokens(q["token"], 5, q["max_dist"])
            assert got == q["out"], q
            n += 1
        c.repair_coor_sys()
        assert {str(t): v for t, v in c.coor.items()} == case["coor_after"]
    assert n == 900


def test_pose_products_are_order_free_and_close_to_blas():
    from deeppointmap_amd.consumer import se3_inv, se3_mul
    g = torch.Generator().manual_seed(0)
    for _ in range(50):
        q = torch.randn(4, generator=g)
        w, x, y, z = (q / q.norm()).tolist()
        A = torch.eye(4)
        A[:3, :3] = torch.tensor([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                                  [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                                  [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        A[:3, 3] = 50 * torch.randn(3, generator=g)
        B = se3_inv(A)
        ref = A.double() @ torch.linalg.inv(A.double())
        assert float((se3_mul(A, B).double() - ref).abs().max()) < 1e-4
        assert float((B.double() - torch.linalg.inv(A.double())).abs().max()) < 1e-5 * 100   # |t| ~ 100, float32 R
        assert torch.equal(se3_mul(A, torch.eye(4)), A) and torch.equal(se3_mul(torch.eye(4), A), A)


def test_gated_trace_fixture_exercises_every_exit():
    """tests/golden/slam_trace_gated.npz (make_trace.py gated): the reference's run in which scans are dropped, localised and
    mapped -- the GPU test replays it through the consumer; here only that the recording holds what it is there for."""
    import numpy as np
    g = np.load(os.path.join(HERE, "golden", "slam_trace_gated.npz"))
    codes = g["codes"].tolist()
    assert codes.count(0) >= 4 and codes.count(10) >= 8 and codes.count(11) >= 5         # acpt, drop, dist
    assert any(codes[i:i + 3] == [10, 10, 0] for i in range(len(codes) - 2))             # the third drop in a row recovered
    types = [str(t) for t in g["final_type"]]
    assert "non-keyframe" in types and "full" in types and len(types) == len(codes) - codes.count(10)
    kinds = [str(k) for k in g["call_kinds"]]
    assert kinds.count("optim") == 2 and kinds.count("loop") == 2 and kinds.count("tile") == 7


def test_text_writers_equal_the_reference_s(tmp_path):
    """system.ResultLogger.save_trajectory / save_posegraph against the files the reference's ResultLogger wrote for the same
    graphs (recoder.py:76-97, pose_graph.py:821-842): byte for byte"""
    from deeppointmap_amd.system import ResultLogger
    cases = json.load(open(os.path.join(HERE, "golden", "graph_cases.json")))
    n = 0
    for i, case in enumerate(cases):
        if not case["files"]:
            continue
        rl = ResultLogger(_graph(case), str(tmp_path / str(i)))
        rl.save_trajectory("traj")
        rl.save_posegraph("graph")
        for name, text in case["files"].items():
            assert open(tmp_path / str(i) / name).read() == text, (i, name)
            n += 1
    assert n == 20


def test_scan_cloud_store_bounds_device_residency_and_returns_the_same_bits():
    """consumer.ScanCloudStore (the consumer's full_pcd map): beyond its budget the least recently used clouds leave the device and
    come back unchanged when an edge asks for them (logic exercised on the CPU device; the consumer tests drive it on the GPU)."""
    from deeppointmap_amd.consumer import ScanCloudStore
    gen = torch.Generator().manual_seed(3)
    clouds = {t: torch.randn(3, 1000, generator=gen) for t in range(10)}
    st = ScanCloudStore("cpu", max_device_bytes=3 * 12000 + 100)      # room for three clouds
    for t, c in clouds.items():
        st[t] = c
        assert st.device_bytes() <= 3 * 12000 + 100
    assert len(st) == 10 and st.stats["evicted"] == 7 and all(t in st for t in clouds)
    assert torch.equal(st[0], clouds[0]) and st.stats["restored"] == 1            # back from the host, same bits
    assert st.device_bytes() <= 3 * 12000 + 100
    assert torch.equal(st.get(9), clouds[9]) and st.get(77) is None
    assert torch.equal(st.pop(3), clouds[3]) and 3 not in st and st.pop(3, None) is None
    st[0] = clouds[1]                                                             # overwrite keeps the accounting straight
    assert torch.equal(st[0], clouds[1]) and st.device_bytes() <= 3 * 12000 + 100
