"""Pose-graph optimisation for the SLAM back end.

Replaces the open3d calls of PoseGraph.__optim_open3d (reference system/modules/pose_graph.py:565-613):
`o3d.pipelines.registration.global_optimization(graph, GlobalOptimizationLevenbergMarquardt(),
GlobalOptimizationConvergenceCriteria(), GlobalOptimizationOption(edge_prune_threshold=0.0,
preference_loop_closure=2.0, reference_node=...))` on a graph whose edges are all `uncertain=False`
(pose_graph.py:597), and the g2o writer PoseGraph.to_g2o_file (pose_graph.py:821-842).

open3d (pinned 0.16.0, requirements.txt:15) is a third-party dependency that is absent here, so this is a
restatement of its published algorithm (cpp/open3d/pipelines/registration/GlobalOptimization.cpp) -- PARITY
UNPINNED against open3d itself; tests/test_posegraph_optim.py pins it against an independent least-squares solver
on the same objective (oracle/dpm_oracle.py::pose_graph_least_squares) and against closed-form cases.

  * node i has a 4x4 pose T_i (scan frame -> world); edge (s, t) carries X = the measured transform taking source-
    scan coordinates into the target scan's frame, and a 6x6 information matrix (rotation block first);
  * residual of an edge: e = vec6(X^-1 T_t^-1 T_s), vec6 = (rx, ry, rz from R = Rz Ry Rx, translation);
  * objective: sum_e e^T Lambda e (every edge certain: no line processes, nothing to prune -- the second
    optimisation pass open3d runs after pruning starts from the first pass's result with the same edges);
  * Levenberg-Marquardt on left-multiplicative pose updates T_i <- expm6(delta_i) T_i with open3d's linearised
    Jacobians, damping lambda = 1e-5 max diag(H) initially and the Nielsen update (scale by
    max(1/3, min(2/3, 1-(2 rho-1)^3)) on success, by 2, 4, 8... on failure);
  * no node is held fixed during the iterations (the damping fixes the gauge); afterwards every pose is multiplied
    by the transform that puts the reference node back where it started.

The problem is small (one 6-vector per keyframe), so this is host code -- native host code, like the open3d routine it
replaces: the iteration lives in csrc/posegraph.hip (`dpm_posegraph_optimize`; reverse Cuthill-McKee renumbering + skyline
Cholesky of the block-sparse normal equations), this module packs the reference's containers for it.  The numpy / scipy
statement of the same iteration is test infrastructure now (oracle/posegraph_numpy.py).  Nothing here touches the GPU.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Iterable, List, Sequence, Tuple

import numpy as np


@dataclass
class ConvergenceCriteria:
    """Defaults of open3d's GlobalOptimizationConvergenceCriteria."""
    max_iteration: int = 100
    min_relative_increment: float = 1e-6
    min_relative_residual_increment: float = 1e-6
    min_right_term: float = 1e-6
    min_residual: float = 1e-6
    max_iteration_lm: int = 20
    upper_scale_factor: float = 2.0 / 3.0
    lower_scale_factor: float = 1.0 / 3.0


def transform_to_vec6(T: np.ndarray) -> np.ndarray:
    """4x4 -> (rx, ry, rz, tx, ty, tz) with R = Rz(rz) Ry(ry) Rx(rx) (open3d TransformMatrix4dToVector6d)."""
    R = T[:3, :3]
    sy = math.sqrt(R[0, 0] * R[0, 0] + R[1, 0] * R[1, 0])
    if sy >= 1e-6:
        rx, ry, rz = math.atan2(R[2, 1], R[2, 2]), math.atan2(-R[2, 0], sy), math.atan2(R[1, 0], R[0, 0])
    else:
        rx, ry, rz = math.atan2(-R[1, 2], R[1, 1]), math.atan2(-R[2, 0], sy), 0.0
    return np.array([rx, ry, rz, T[0, 3], T[1, 3], T[2, 3]], dtype=np.float64)


def vec6_to_transform(v: Sequence[float]) -> np.ndarray:
    """Inverse of transform_to_vec6 (open3d TransformVector6dToMatrix4d)."""
    cx, sx, cy, sy, cz, sz = math.cos(v[0]), math.sin(v[0]), math.cos(v[1]), math.sin(v[1]), math.cos(v[2]), math.sin(v[2])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]], dtype=np.float64)
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]], dtype=np.float64)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]], dtype=np.float64)
    T = np.eye(4)
    T[:3, :3] = Rz @ Ry @ Rx
    T[:3, 3] = v[3:6]
    return T


def _inv(T: np.ndarray) -> np.ndarray:
    out = np.eye(4)
    out[:3, :3] = T[:3, :3].T
    out[:3, 3] = -T[:3, :3].T @ T[:3, 3]
    return out


def global_optimization(poses, edges: Iterable[Tuple[int, int, np.ndarray, np.ndarray]], reference_node: int = 0,
                        criteria: ConvergenceCriteria = None, uncertain: Sequence[bool] = None,
                        return_stats: bool = False):
    """poses (n,4,4) node poses; edges = (source_id, target_id, transformation 4x4, information 6x6) exactly as the
    reference fills o3d PoseGraphEdge (pose_graph.py:589-596: transformation = inv(edge.SE3)).  Returns the refined
    (n,4,4) float64 poses with node `reference_node` unchanged.  Only certain edges are supported -- the one
    configuration the reference uses.  The iteration runs in libdpm_hip.so (`dpm_posegraph_optimize`, host code)."""
    if uncertain is not None and any(bool(u) for u in uncertain):
        raise NotImplementedError("uncertain edges (line processes) are never produced by the reference (pose_graph.py:597)")
    edges = list(edges)
    P = np.ascontiguousarray(np.asarray(poses, dtype=np.float64).reshape(-1, 4, 4))
    n, E = P.shape[0], len(edges)
    src = np.ascontiguousarray([e[0] for e in edges], dtype=np.int32)
    dst = np.ascontiguousarray([e[1] for e in edges], dtype=np.int32)
    if E and (src.min() < 0 or dst.min() < 0 or src.max() >= n or dst.max() >= n):
        raise ValueError("edge endpoint outside the node list")
    if not 0 <= reference_node < max(n, 1):
        raise ValueError("reference_node outside the node list")
    X = np.ascontiguousarray(np.stack([np.asarray(e[2], dtype=np.float64).reshape(4, 4) for e in edges])) if E else np.zeros((0, 4, 4))
    info = np.ascontiguousarray(np.stack([np.asarray(e[3], dtype=np.float64).reshape(6, 6) for e in edges])) if E else np.zeros((0, 6, 6))
    crit = None
    if criteria is not None:
        crit = np.array([criteria.max_iteration, criteria.min_relative_increment, criteria.min_relative_residual_increment,
                         criteria.min_right_term, criteria.min_residual, criteria.max_iteration_lm,
                         criteria.upper_scale_factor, criteria.lower_scale_factor], dtype=np.float64)
    out, stats = np.empty_like(P), np.zeros(6)
    from . import _lib

    def ptr(a):
        return a.ctypes.data if a is not None and a.size else None
    _lib.check(_lib.load().dpm_posegraph_optimize(ptr(P), n, ptr(src), ptr(dst), ptr(X), ptr(info), E, int(reference_node),
                                                  ptr(crit), out.ctypes.data if n else stats.ctypes.data, stats.ctypes.data),
               "dpm_posegraph_optimize")
    if return_stats:
        return out, dict(first=dict(iterations=int(stats[0]), residual_start=float(stats[1]), residual=float(stats[2])),
                         second=dict(iterations=int(stats[3]), residual_start=float(stats[4]), residual=float(stats[5])))
    return out


def optimize_pose_graph(node_SE3: Dict[int, np.ndarray], edges: List[Tuple[int, int, np.ndarray, np.ndarray]],
                        base_token: int = None):
    """The body of PoseGraph.__optim_open3d on plain containers: node_SE3 {token: SE3_pred of a key-frame scan},
    edges [(src_token, dst_token, edge.SE3, edge.information_mat)] (non-'locz' edges; edges touching a token that is
    not a key-frame are skipped, as at pose_graph.py:588-599).  -> ({token: refined 4x4 float32}, mean translation
    change), the values the reference writes back (pose_graph.py:605-617)."""
    tokens = list(node_SE3.keys())
    index = {t: i for i, t in enumerate(tokens)}
    base = min(tokens) if base_token is None else base_token
    poses = np.stack([np.asarray(node_SE3[t], dtype=np.float64) for t in tokens]) if tokens else np.zeros((0, 4, 4))
    # edges with a non-finite transformation or information matrix (a zero-pair registration returns NaN poses, as the
    # reference's does) say nothing about the graph: they are left out, as hanging edges are
    packed = [(index[s], index[d], np.linalg.inv(np.asarray(T, dtype=np.float64)), np.asarray(info, dtype=np.float64))
              for s, d, T, info in edges if s in index and d in index
              and np.isfinite(np.asarray(T, dtype=np.float64)).all() and np.isfinite(np.asarray(info, dtype=np.float64)).all()]
    refined = global_optimization(poses, packed, reference_node=index[base]) if tokens else poses
    out = {t: refined[i].astype(np.float32) for t, i in index.items()}
    diff = [float(np.linalg.norm(poses[i][:3, 3].astype(np.float32) - out[t][:3, 3])) for t, i in index.items()]
    return out, (sum(diff) / len(diff) if diff else 0.0)


def _quat_xyzw(R: np.ndarray) -> np.ndarray:
    """(x, y, z, w), through the same scipy call the reference makes (pose_graph.py:831,835)."""
    from scipy.spatial.transform import Rotation
    return Rotation.from_matrix(np.asarray(R, dtype=np.float64)).as_quat()


def write_g2o(path: str, node_SE3: Dict[int, np.ndarray], edges: List[Tuple[int, int, np.ndarray, np.ndarray]]) -> None:
    """PoseGraph.to_g2o_file (pose_graph.py:821-842): VERTEX_SE3:QUAT token x y z qx qy qz qw, then
    EDGE_SE3:QUAT src dst x y z qx qy qz qw + the 21 upper-triangle information entries, row-major."""
    with open(path, "w+") as f:
        for token, T in node_SE3.items():
            T = np.asarray(T, dtype=np.float64)
            q = _quat_xyzw(T[:3, :3])
            f.write(f"VERTEX_SE3:QUAT {token} {T[0, 3]} {T[1, 3]} {T[2, 3]} {q[0]} {q[1]} {q[2]} {q[3]} \n")
        for s, d, T, info in edges:
            T, info = np.asarray(T, dtype=np.float64), np.asarray(info, dtype=np.float64)
            q = _quat_xyzw(T[:3, :3])
            upper = " ".join(" ".join(str(info[i, j]) for j in range(i, 6)) + " " for i in range(6))
            f.write(f"EDGE_SE3:QUAT {s} {d} {T[0, 3]} {T[1, 3]} {T[2, 3]} {q[0]} {q[1]} {q[2]} {q[3]} {upper}\n")


def read_g2o(path: str):
    """Inverse of write_g2o (for round-trip tests and for feeding external solvers' output back)."""
    nodes, edges = {}, []

    def pose(v):
        x, y, z, qx, qy, qz, qw = v
        T = np.eye(4)
        T[:3, :3] = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                              [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                              [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        T[:3, 3] = (x, y, z)
        return T

    with open(path) as f:
        for line in f:
            w = line.split()
            if not w:
                continue
            if w[0] == "VERTEX_SE3:QUAT":
                nodes[int(w[1])] = pose([float(v) for v in w[2:9]])
            elif w[0] == "EDGE_SE3:QUAT":
                vals = [float(v) for v in w[3:]]
                info = np.zeros((6, 6))
                info[np.triu_indices(6)] = vals[7:28]
                info = info + np.triu(info, 1).T
                edges.append((int(w[1]), int(w[2]), pose(vals[:7]), info))
    return nodes, edges
