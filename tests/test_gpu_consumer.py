"""deeppointmap_amd/consumer.py::Rank0Consumer -- the reference's odometry / mapping / loop-closure / optimisation rules
around the device path -- driven by the scans of the recorded SlamSystem.step run (tests/golden/slam_trace.npz) and held to
every call and every decision of that run, and by hand-made edge rows for the gating rules the run does not reach."""
import numpy as np
import pytest
import torch

from conftest import T, load_golden, rot_angle

pytestmark = pytest.mark.gpu


# the thresholds tests/golden/make_trace.py ran the reference with (loosened so that, with procedural weights, key-frames,
# scan-to-map refinements and loop closures all occur)
TRACE_SLAM = dict(coor_scale=60, odometer_candidates_num=1, registration_sample_odometer=0.5,
                  edge_confidence_drop=0.0, edge_rmse_drop=1e9, max_continuous_drop_scan=5, continuous_drop_scan_strategy="recover",
                  key_frame_distance=0.0, enable_s2m_adjust=True, registration_sample_mapping=0.5,
                  enable_loop_closure=True, loop_detection_gap=0, loop_detection_transaction_gap=0.0, loop_detection_trust_range=3,
                  loop_detection_gnss_distance=-1, loop_detection_pred_distance=1e9, loop_detection_rotation_min=0.0,
                  loop_detection_translation_min=0.0, loop_detection_prob_acpt_threshold=0.0, loop_detection_candidates_num=1,
                  registration_sample_loop=0.5, loop_detection_confidence_acpt_threshold=0.0,
                  enable_global_optimization=True, global_optimization_gap=0)


@pytest.mark.parametrize("mode", ["rows", "recorded", "own"])
def test_consumer_runs_the_reference_s_step_on_the_trace(cfg_full, mode):
    """The consumer alone decides: which key-frame a scan is registered against, whether it is dropped / localised / mapped,
    which scans form its scan-to-map tile, which key-frames are loop candidates, which of them is registered map against
    map with which columns, whether the loop edge is believed, when the optimiser runs and on which graph.  Every device
    call it makes is held to the call the reference made at that place (tests/golden/slam_trace.npz: 145 calls of
    SlamSystem.step over 15 scans), and the trajectory that comes out to the reference's.

      rows     : the sharded path -- the reference's odometry registrations arrive as gathered edge rows (push);
      recorded : the reference's flow (step): the odometry registration runs here too, on the recorded descriptors;
      own      : the same with descriptors from OUR encoder -- nothing but the scans comes from the recording."""
    from deeppointmap_amd import ops
    from deeppointmap_amd.consumer import Rank0Consumer
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.registration import make_descriptors
    from deeppointmap_amd.weights import init_procedural
    g = load_golden("slam_trace.npz")
    kinds = [str(k) for k in g["call_kinds"]]
    desc = {int(t): T(g["desc"][i]) for i, t in enumerate(g["desc_tokens"])}
    frames = [T(g[f"frame{i}"]) for i in range(11)]
    n_steps = len(g["order"])
    frame_of = {int(g[f"s{s}.token"]): int(g[f"s{s}.frame"]) for s in range(n_steps)}
    first = {}                                           # a revisited frame has the feature columns of its first visit, and
    for t in sorted(frame_of):                           # the recording names descriptor columns by (first token, column)
        first.setdefault(frame_of[t], t)
    alias = {t: first[f] for t, f in frame_of.items()}
    dev = torch.device("cuda:0")
    dec = init_procedural(Decoder(cfg_full)).to(dev)
    enc = init_procedural(Encoder(cfg_full)).to(dev) if mode == "own" else None
    optim_calls = []
    cons = Rank0Consumer(dec, dev, slam_args=TRACE_SLAM, keep_log=True,
                         optimiser=lambda nodes, es, base: optim_calls.append((nodes, es, base)))
    worst = dict(dT=0.0, dR=0.0, loop=0.0, info=0.0)
    for s in range(n_steps):
        tok, f = int(g[f"s{s}.token"]), int(g[f"s{s}.frame"])
        k0, k1 = g[f"s{s}.calls"]
        want = [k for k in range(k0, k1) if kinds[k] != "enc"]
        if mode == "own":
            p = frames[f].unsqueeze(0)
            coor, fea, _ = enc(p, torch.zeros(1, p.shape[2], dtype=torch.bool))
            d = make_descriptors(coor, fea, 60.0)[0].to(dev)
        else:
            d = desc[tok].to(dev)
        pcd = (frames[f] * 60.0).to(dev).contiguous()
        n0 = len(cons.log)
        if mode == "rows":
            row = torch.zeros(ops.RES_HDR + 36)
            if want:                                     # the odometry registration of this step and its information matrix
                k = want[0]
                assert kinds[k] == "reg" and kinds[k + 1] == "info" and int(g[f"c{k + 1}.dst"]) == tok
                row[0:9], row[9:12] = T(g[f"c{k}.R"]).reshape(9), T(g[f"c{k}.T"]).reshape(3)
                row[12], row[14], row[16] = float(g[f"c{k}.rmse"]), float(g[f"c{k}.n_conf"]), float(g[f"c{k}.conf30"])
                row[ops.RES_HDR:] = T(g[f"c{k + 1}.G"]).reshape(36)
                want = want[2:]
            assert cons.push(tok, d, row, pcd) == "acpt"
        else:
            assert cons.step(d, pcd) == (tok, "acpt")
        got = cons.log[n0:]
        assert [c["kind"] for c in got] == [kinds[k] for k in want], (s, [c["kind"] for c in got], [kinds[k] for k in want])
        for c, k in zip(got, want):
            if c["kind"] == "tile":                      # graph_search order and the 20 m cut, scan by scan
                assert c["tokens"] == [int(t) for t in g[f"c{k}.tokens"]], (s, k)
            elif c["kind"] == "reg":                     # the same columns on both sides (the overlap deal of a loop closure)
                assert c["cols"] == (g[f"c{k}.src_tok"].size, g[f"c{k}.dst_tok"].size), (s, k, c["cols"])
                side = c["src"] if isinstance(c["src"], list) else [c["src"]]
                assert [alias[t] for t in side] == [int(t) for t in g[f"c{k}.src_tok"][::256]], (s, k)
                dT = float((c["SE3"][:3, 3] - T(g[f"c{k}.T"]).reshape(3)).norm())
                dR = rot_angle(c["SE3"][:3, :3], g[f"c{k}.R"])
                assert dT < 1e-4 and dR < 1e-4 and abs(c["rmse"] - float(g[f"c{k}.rmse"])) < 1e-4, (s, k, dT, dR)
                worst["dT"], worst["dR"] = max(worst["dT"], dT), max(worst["dR"], dR)
            elif c["kind"] == "info":
                assert (c["src"], c["dst"]) == (int(g[f"c{k}.src"]), int(g[f"c{k}.dst"])), (s, k)
                want_G = g[f"c{k}.G"]
                worst["info"] = max(worst["info"], float(np.abs(c["G"].numpy() - want_G).max() / np.abs(want_G).max()))
            elif c["kind"] == "loop":                    # the candidates that survive the trusted zones, in graph order
                assert [alias[t] for t in c["src_tokens"]] == [int(t) for t in g[f"c{k}.src_tokens"]] and c["dst"] == tok, (s, k)
                worst["loop"] = max(worst["loop"], float((c["prob"] - T(g[f"c{k}.prob"])).abs().max()))
            else:                                        # the graph handed to the optimiser (pose_graph.py:565-608)
                assert c["tokens"] == [int(t) for t in g[f"c{k}.tokens"]]
                assert c["edges"] == [(int(a), int(b), str(ty)) for a, b, ty in
                                      zip(g[f"c{k}.edge_src"], g[f"c{k}.edge_dst"], g[f"c{k}.edge_type"])]
                nodes, es, base = optim_calls[-1]
                assert base == int(g[f"c{k}.reference"])
                for (a, b, X, info), X_ref, info_ref in zip(es, g[f"c{k}.edge_T"], g[f"c{k}.edge_info"]):
                    X_ref = np.linalg.inv(X_ref.astype(np.float64))                 # recorded: inv(edge.SE3)
                    assert np.abs(X[:3, 3] - X_ref[:3, 3]).max() < 2e-4 and np.abs(X[:3, :3] - X_ref[:3, :3]).max() < 1e-4
                    assert np.abs(info - info_ref).max() <= 1e-3 * np.abs(info_ref).max()
    print(f"consumer on the trace ({mode}): worst", worst)
    assert worst["loop"] < (1e-4 if mode == "own" else 2e-5) and worst["info"] < 1e-3
    assert cons.codes == ["acpt"] * n_steps and cons.stats["s2m"] == n_steps - 1
    assert cons.stats["loop_batches"] == kinds.count("loop") == 12 and cons.stats["loop_edges"] == 8 == len(optim_calls)
    assert [cons.type[int(t)] for t in g["final_tokens"]] == [str(x) for x in g["final_type"]]
    final = {int(t): T(x) for t, x in zip(g["final_tokens"], g["final_SE3"])}
    dev_t = max(float((cons.poses[t][:3, 3] - final[t][:3, 3]).norm()) for t in final)
    dev_r = max(rot_angle(cons.poses[t][:3, :3], final[t][:3, :3].numpy()) for t in final)
    assert dev_t < 1e-4 and dev_r < 1e-4, (dev_t, dev_r)


def test_consumer_with_the_native_optimiser_in_the_loop(cfg_full):
    """The same trace with the NATIVE pose-graph optimiser (csrc/posegraph.hip, posegraph_optim.py) where the tests above put a
    recorder: every loop edge the consumer believes triggers an optimisation of the key-frame graph on this host.  open3d's values
    cannot be had here (parity unpinned), so what is held is the call site and the solver's contract on the graphs that
    actually occur: the reference node keeps its pose, every refined pose is a finite rigid transform, the weighted residual of
    the graph does not grow, and the result equals the solver's numpy statement (oracle/posegraph_numpy.py)."""
    import deeppointmap_amd.consumer as C
    from deeppointmap_amd import ops, posegraph_optim as PG
    from deeppointmap_amd.consumer import Rank0Consumer
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.weights import init_procedural
    from oracle import posegraph_numpy as PN
    g = load_golden("slam_trace.npz")
    kinds = [str(k) for k in g["call_kinds"]]
    desc = {int(t): T(g["desc"][i]) for i, t in enumerate(g["desc_tokens"])}
    frames = [T(g[f"frame{i}"]) for i in range(11)]
    dev = torch.device("cuda:0")
    cons = Rank0Consumer(init_procedural(Decoder(cfg_full)).to(dev), dev, slam_args=TRACE_SLAM)
    seen = []
    orig = C.optimize_pose_graph

    def checked(nodes, es, base_token=None):
        refined, diff = orig(nodes, es, base_token=base_token)
        toks = sorted(nodes)
        index = {t: i for i, t in enumerate(toks)}
        poses = np.stack([nodes[t] for t in toks])
        edges = [(index[a], index[b], np.linalg.inv(X), info) for a, b, X, info in es]   # pose_graph.py:593: open3d gets inv(edge.SE3)
        want, st = PN.global_optimization(poses, edges, reference_node=index[base_token], return_stats=True)
        for t in toks:
            P = np.asarray(refined[t], dtype=np.float64)
            assert np.isfinite(P).all() and abs(np.linalg.det(P[:3, :3]) - 1.0) < 1e-5 and np.allclose(P[3], [0, 0, 0, 1])
            np.testing.assert_allclose(P, want[index[t]], atol=2e-5)
        np.testing.assert_array_equal(np.asarray(refined[base_token], dtype=np.float64), nodes[base_token])
        assert st["second"]["residual"] <= st["first"]["residual_start"] * (1 + 1e-9)
        seen.append((len(toks), len(es), float(diff)))
        return refined, diff

    C.optimize_pose_graph = checked
    try:
        for s in range(len(g["order"])):
            tok, f = int(g[f"s{s}.token"]), int(g[f"s{s}.frame"])
            assert cons.step(desc[tok].to(dev), (frames[f] * 60.0).to(dev).contiguous())[0] == tok
    finally:
        C.optimize_pose_graph = orig
    # procedural weights: every loop candidate is believed under the trace's thresholds, so the optimiser ran at least once per
    # believed loop edge, on graphs of growing size
    assert cons.stats["optimisations"] == len(seen) >= 1 and cons.stats["loop_edges"] >= 1
    assert all(torch.isfinite(P).all() for P in cons.poses.values())
    print("native optimiser in the loop:", seen)


def test_consumer_drops_localises_and_recovers_like_the_reference(cfg_full):
    """The second recorded run (tests/golden/slam_trace_gated.npz: 20 steps under thresholds that DROP scans -- the third drop
    in a row recovered --, LOCALISE scans without making them key-frames and refuse scan-to-map results): exit code by exit
    code, call by call, with descriptors from OUR encoder.  Tokens skip the dropped scans, non-key-frames hang on their
    key-frames by 'locz' edges, the odometry partner is searched from the last key-frame around the last known pose."""
    import json
    from deeppointmap_amd.consumer import Rank0Consumer
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.registration import make_descriptors
    from deeppointmap_amd.weights import init_procedural
    g, scans = load_golden("slam_trace_gated.npz"), load_golden("slam_trace.npz")
    kinds = [str(k) for k in g["call_kinds"]]
    frames = [T(scans[f"frame{i}"]) for i in range(11)]
    n_steps = len(g["order"])
    dev = torch.device("cuda:0")
    enc, dec = init_procedural(Encoder(cfg_full)).to(dev), init_procedural(Decoder(cfg_full)).to(dev)
    cons = Rank0Consumer(dec, dev, slam_args=json.loads(str(g["slam_args"])), keep_log=True, optimiser=lambda nodes, es, base: None)
    name = {0: "acpt", 10: "drop", 11: "dist"}
    first = {}
    worst = 0.0
    for s in range(n_steps):
        tok, f = int(g[f"s{s}.token"]), int(g[f"s{s}.frame"])
        first.setdefault(f, tok)
        k0, k1 = g[f"s{s}.calls"]
        want = [k for k in range(k0, k1) if kinds[k] != "enc"]
        p = frames[f].unsqueeze(0)
        coor, fea, _ = enc(p, torch.zeros(1, p.shape[2], dtype=torch.bool))
        n0 = len(cons.log)
        got_tok, code = cons.step(make_descriptors(coor, fea, 60.0)[0].to(dev), (frames[f] * 60.0).to(dev).contiguous())
        assert (got_tok, code) == (tok, name[int(g["codes"][s])]), (s, got_tok, code)
        got = cons.log[n0:]
        assert [c["kind"] for c in got] == [kinds[k] for k in want], (s, [c["kind"] for c in got], [kinds[k] for k in want])
        for c, k in zip(got, want):
            if c["kind"] == "tile":
                assert c["tokens"] == [int(t) for t in g[f"c{k}.tokens"]], (s, k)
            elif c["kind"] == "reg":
                dT = float((c["SE3"][:3, 3] - T(g[f"c{k}.T"]).reshape(3)).norm())
                dR = rot_angle(c["SE3"][:3, :3], g[f"c{k}.R"])
                assert dT < 1e-4 and dR < 1e-4 and abs(c["rmse"] - float(g[f"c{k}.rmse"])) < 1e-4, (s, k, dT, dR)
                worst = max(worst, dT)
            elif c["kind"] == "info":
                assert (c["src"], c["dst"]) == (int(g[f"c{k}.src"]), int(g[f"c{k}.dst"])), (s, k)
            elif c["kind"] == "loop":
                assert c["dst"] == tok and len(c["src_tokens"]) == g[f"c{k}.src_tokens"].size, (s, k)
                assert float((c["prob"] - T(g[f"c{k}.prob"])).abs().max()) < 1e-4
            else:
                assert c["tokens"] == [int(t) for t in g[f"c{k}.tokens"]]
                assert c["edges"] == [(int(a), int(b), str(ty)) for a, b, ty in
                                      zip(g[f"c{k}.edge_src"], g[f"c{k}.edge_dst"], g[f"c{k}.edge_type"])]
    codes = [name[int(c)] for c in g["codes"]]
    assert cons.codes == codes and {"acpt", "drop", "dist"} <= set(codes)
    assert cons.stats["dropped"] == codes.count("drop") and cons.stats["keyframes"] == codes.count("acpt")
    assert list(cons.type) == [int(t) for t in g["final_tokens"]]
    assert [cons.type[int(t)] for t in g["final_tokens"]] == [str(x) for x in g["final_type"]]
    final = {int(t): T(x) for t, x in zip(g["final_tokens"], g["final_SE3"])}
    dev_t = max(float((cons.poses[t][:3, 3] - final[t][:3, 3]).norm()) for t in final)
    dev_r = max(rot_angle(cons.poses[t][:3, :3], final[t][:3, :3].numpy()) for t in final)
    print(f"gated run: worst registration {worst:.2e} m, trajectory {dev_t:.2e} m / {dev_r:.2e} rad")
    assert dev_t < 1e-4 and dev_r < 1e-4, (dev_t, dev_r)


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
    cons = Rank0Consumer(dec, dev, slam_args=dict(edge_confidence_drop=0.5, edge_rmse_drop=1.0, key_frame_distance=3.0, enable_loop_closure=False,
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
