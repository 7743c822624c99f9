"""The multi-agent flow against the reference's own (tests/golden/cloud_trace.npz, recorded by make_trace_cloud.py from two
reference `AgentSystem`s and a `CloudSystem`, system/core.py:426-546):

  * the CLOUD: `Rank0Consumer.cloud_step` fed the reference's twelve uploads must build the reference's graph after every
    step -- the same scans in the same order, coordinate systems, edges (which cross-agent loops were closed, in which order)
    and every pose / edge transform within the registration tolerance;
  * the AGENTS: two `system.AgentSystem`s stepping through the same scans must upload what the reference's agents uploaded:
    tokens, poses, odometry edges and the loop edges each found in its own map."""
import pytest
import torch

from conftest import T, load_golden, rot_angle
from test_gpu_consumer import TRACE_SLAM

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _close(A, B):
    A, B = torch.as_tensor(A).float(), torch.as_tensor(B).float()
    return float((A[:3, 3] - B[:3, 3]).norm()) < TOL and rot_angle(A[:3, :3], B[:3, :3].numpy()) < TOL


def _edge(g, prefix):
    return dict(src=int(g[prefix + ".src"]), dst=int(g[prefix + ".dst"]), type=str(g[prefix + ".type"]), SE3=T(g[prefix + ".SE3"]),
                information=T(g[prefix + ".info"]), confidence=float(g[prefix + ".confidence"]), rmse=float(g[prefix + ".rmse"]))


def _models(cfg_full, dev):
    from deeppointmap_amd.decoder import Decoder
    from deeppointmap_amd.encoder import Encoder
    from deeppointmap_amd.weights import init_procedural
    return init_procedural(Encoder(cfg_full)).to(dev), init_procedural(Decoder(cfg_full)).to(dev)


def test_cloud_builds_the_reference_s_graph(cfg_full):
    from deeppointmap_amd.consumer import Rank0Consumer
    g, scans = load_golden("cloud_trace.npz"), load_golden("slam_trace.npz")
    dev = torch.device("cuda:0")
    _, dec = _models(cfg_full, dev)
    desc = {int(t): T(scans["desc"][i]) for i, t in enumerate(scans["desc_tokens"])}   # frame f <-> token f (first visit)
    cloud = Rank0Consumer(dec, dev, slam_args=TRACE_SLAM, agent_id=0, loop_targets="others", optimiser=lambda n, e, b: None)
    for u in range(int(g["n_uploads"])):
        p = f"u{u}"
        f = int(g[p + ".frame"])
        scan = dict(token=int(g[p + ".token"]), agent_id=int(g[p + ".agent"]), timestep=int(g[p + ".timestep"]),
                    type=str(g[p + ".type"]), key_points=desc[f].to(dev), full_pcd=(T(scans[f"frame{f}"]) * 60.0).to(dev).contiguous(),
                    SE3_pred=T(g[p + ".SE3_pred"]), coor_sys=int(g[p + ".coor_sys"]))
        odom = _edge(g, p + ".odom") if int(g[p + ".has_odom"]) else None
        nbrs = [_edge(g, f"{p}.nbr{j}") for j in range(int(g[p + ".n_nbr"]))]
        n_opt = cloud.stats["optimisations"]
        cloud.cloud_step(scan, odom, nbrs)
        assert list(cloud.type) == g[p + ".g_tokens"].tolist(), u
        assert [cloud.coor[t] for t in cloud.type] == g[p + ".g_coor"].tolist(), u
        assert [(a, b, e["type"]) for (a, b), e in cloud.edges.items()] == \
            [(int(a), int(b), str(ty)) for a, b, ty in zip(g[p + ".g_edge_src"], g[p + ".g_edge_dst"], g[p + ".g_edge_type"])], u
        for t, P in zip(cloud.type, g[p + ".g_SE3"]):
            assert _close(cloud.poses[t], P), (u, t)
        for e, X in zip(cloud.edges.values(), g[p + ".g_edge_SE3"]):
            assert _close(e["SE3"], X), (u, e["src"], e["dst"])
        assert cloud.stats["optimisations"] - n_opt == int(g[p + ".cloud_optim"]), u
    cross = [(a, b) for (a, b), e in cloud.edges.items() if e["type"] == "loop" and (a >> 16) != (b >> 16)]
    assert len(cross) >= 5 and len(set(cloud.coor.values())) == 1


def test_agents_upload_what_the_reference_s_agents_upload(cfg_full):
    from deeppointmap_amd.config import Cfg
    from deeppointmap_amd.system import AgentSystem, EXIT_CODE
    from test_gpu_multiagent import LocalComm
    g, scans = load_golden("cloud_trace.npz"), load_golden("slam_trace.npz")
    dev = torch.device("cuda:0")
    enc, dec = _models(cfg_full, dev)
    args = Cfg(dict(cfg_full))
    args.device, args.slam_system = "cuda:0", Cfg(TRACE_SLAM)
    comm = LocalComm()
    comm.add_member(0)
    agents = {a: AgentSystem(args, enc, dec, system_id=a, comm_module=comm, device=dev) for a in (1, 2)}
    for a in agents.values():
        a.backend.optimiser = lambda n, e, b: None
    plan = {1: [0, 1, 2, 3, 4, 5], 2: [10, 9, 8, 7, 6, 5]}
    for i in range(6):
        for a in (1, 2):
            p = T(scans[f"frame{plan[a][i]}"]).unsqueeze(0)
            code = agents[a].step([p, torch.eye(3).unsqueeze(0), torch.zeros(1, 3, 1), torch.zeros(1, p.shape[2], dtype=torch.bool), None])
            assert code == EXIT_CODE.acpt
    ups = [m for _, _, c, m in comm.sent if c == "UPLOAD_SCAN"]
    assert len(ups) == int(g["n_uploads"]) == 12
    for u, m in enumerate(ups):
        p, s = f"u{u}", m["new_scan"]
        assert (s["token"], s["agent_id"], s["timestep"], s["type"]) == \
            (int(g[p + ".token"]), int(g[p + ".agent"]), int(g[p + ".timestep"]), str(g[p + ".type"])), u
        # The reference's agents and cloud are threads handing each other the SAME ScanPack objects: once the cloud's
        # repair_coor_sys has relabelled agent 2's first scan, agent 2's own graph holds the relabelled object and its later
        # scans inherit the cloud's coordinate system (recorded: 1 from its second upload on).  Ranks exchange copies: an agent
        # keeps its own system, and the cloud gives an arriving scan the system of its odometry predecessor (core.py:478-485),
        # which is where the graph of the test above gets the same labels from.
        assert s["coor_sys"] == s["agent_id"] and int(g[p + ".coor_sys"]) in (1, s["agent_id"]), u
        assert _close(s["SE3_pred"], g[p + ".SE3_pred"]), u
        assert (m["odometer_edge"] is not None) == bool(int(g[p + ".has_odom"])) and len(m["neighbor_edges"]) == int(g[p + ".n_nbr"]), u
        pairs = ([(m["odometer_edge"], p + ".odom")] if m["odometer_edge"] is not None else []) + \
            [(e, f"{p}.nbr{j}") for j, e in enumerate(m["neighbor_edges"])]
        for e, q in pairs:
            assert (e["src"], e["dst"], e["type"]) == (int(g[q + ".src"]), int(g[q + ".dst"]), str(g[q + ".type"])), (u, q)
            assert _close(e["SE3"], g[q + ".SE3"]) and abs(e["rmse"] - float(g[q + ".rmse"])) < 1e-4, (u, q)
            want = T(g[q + ".info"])
            assert float((e["information"] - want).abs().max()) <= 1e-3 * float(want.abs().max()), (u, q)
    # ... and a cloud fed THESE uploads (copies carrying the agents' own coordinate systems) ends on the reference's graph too
    from deeppointmap_amd.consumer import Rank0Consumer
    cloud = Rank0Consumer(dec, dev, slam_args=TRACE_SLAM, agent_id=0, loop_targets="others", optimiser=lambda n, e, b: None)
    for m in ups:
        cloud.cloud_step(m["new_scan"], m["odometer_edge"], m["neighbor_edges"])
    p = f"u{len(ups) - 1}"
    assert list(cloud.type) == g[p + ".g_tokens"].tolist() and [cloud.coor[t] for t in cloud.type] == g[p + ".g_coor"].tolist()
    assert [(a, b, e["type"]) for (a, b), e in cloud.edges.items()] == \
        [(int(a), int(b), str(ty)) for a, b, ty in zip(g[p + ".g_edge_src"], g[p + ".g_edge_dst"], g[p + ".g_edge_type"])]
    assert all(_close(cloud.poses[t], P) for t, P in zip(cloud.type, g[p + ".g_SE3"]))

