"""TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.

numpy / scipy statement of the Levenberg-Marquardt iteration that deeppointmap_amd/csrc/posegraph.hip implements
(open3d 0.16 `global_optimization` on certain edges; reference system/modules/pose_graph.py:565-613 is its call site):
the same residual, Jacobians, damping schedule and stopping rules, with the normal equations solved by scipy (dense
Cholesky, or sparse LU with `DENSE_LIMIT = 0`) instead of the native skyline Cholesky.  It was the product code until the
native routine replaced it; tests/test_posegraph_optim.py holds the two to each other (1e-8 on the poses) and both to
the independent least-squares checker `dpm_oracle.pose_graph_least_squares`.  PARITY UNPINNED against open3d itself
(absent from the image).
"""
from __future__ import annotations

from typing import Iterable, Sequence, Tuple

import numpy as np
import scipy.linalg
import scipy.sparse
import scipy.sparse.linalg

from deeppointmap_amd.posegraph_optim import ConvergenceCriteria, _inv, transform_to_vec6, vec6_to_transform


def _generators() -> np.ndarray:
    G = np.zeros((6, 4, 4))
    G[0, 1, 2], G[0, 2, 1] = -1, 1
    G[1, 2, 0], G[1, 0, 2] = -1, 1
    G[2, 0, 1], G[2, 1, 0] = -1, 1
    G[3, 0, 3] = G[4, 1, 3] = G[5, 2, 3] = 1
    return G


_G = _generators()


def _linearized(M: np.ndarray) -> np.ndarray:
    """First-order vec6 of a near-identity-derivative matrix (open3d GetLinearized6DVector), batched over axis 0."""
    return np.stack([(-M[..., 1, 2] + M[..., 2, 1]) / 2, (-M[..., 2, 0] + M[..., 0, 2]) / 2,
                     (-M[..., 0, 1] + M[..., 1, 0]) / 2, M[..., 0, 3], M[..., 1, 3], M[..., 2, 3]], axis=-1)


class _Graph:
    def __init__(self, poses: np.ndarray, edges):
        self.poses = np.array(poses, dtype=np.float64).reshape(-1, 4, 4).copy()
        self.src = np.array([e[0] for e in edges], dtype=np.int64)
        self.dst = np.array([e[1] for e in edges], dtype=np.int64)
        self.Xinv = np.stack([_inv(np.asarray(e[2], dtype=np.float64)) for e in edges]) if len(edges) else np.zeros((0, 4, 4))
        self.info = np.stack([np.asarray(e[3], dtype=np.float64).reshape(6, 6) for e in edges]) if len(edges) else np.zeros((0, 6, 6))
        n = self.poses.shape[0]
        if len(edges) and (self.src.min() < 0 or self.dst.min() < 0 or self.src.max() >= n or self.dst.max() >= n):
            raise ValueError("edge endpoint outside the node list")

    def zeta(self, poses: np.ndarray) -> np.ndarray:
        """(E,6) misalignment vectors."""
        out = np.empty((self.src.size, 6))
        for k in range(self.src.size):
            out[k] = transform_to_vec6(self.Xinv[k] @ _inv(poses[self.dst[k]]) @ poses[self.src[k]])
        return out

    def residual(self, z: np.ndarray) -> float:
        return float(np.einsum("ei,eij,ej->", z, self.info, z))

    def linear_system(self, poses: np.ndarray, z: np.ndarray):
        """Block-sparse H (6n x 6n, CSC) and b of the Gauss-Newton step (open3d ComputeLinearSystem)."""
        n, E = poses.shape[0], self.src.size
        Js = np.empty((E, 6, 6))
        for k in range(E):
            A = self.Xinv[k] @ _inv(poses[self.dst[k]])          # (4,4)
            B = poses[self.src[k]]
            Js[k] = _linearized(A[None] @ _G @ B[None]).T        # column i = derivative along generator i
        Jt = -Js
        JsI = np.einsum("eji,ejk->eik", Js, self.info)           # Js^T Lambda
        JtI = -JsI
        blocks = {(0, 0): np.einsum("eij,ejk->eik", JsI, Js), (0, 1): np.einsum("eij,ejk->eik", JsI, Jt),
                  (1, 0): np.einsum("eij,ejk->eik", JtI, Js), (1, 1): np.einsum("eij,ejk->eik", JtI, Jt)}
        ends = (self.src, self.dst)
        rows, cols, vals = [], [], []
        ii, jj = np.meshgrid(np.arange(6), np.arange(6), indexing="ij")
        for (a, c), blk in blocks.items():
            rows.append((ends[a][:, None, None] * 6 + ii[None]).ravel())
            cols.append((ends[c][:, None, None] * 6 + jj[None]).ravel())
            vals.append(blk.ravel())
        H = scipy.sparse.coo_matrix((np.concatenate(vals), (np.concatenate(rows), np.concatenate(cols))),
                                    shape=(6 * n, 6 * n)).tocsc()
        b = np.zeros(6 * n)
        np.subtract.at(b.reshape(n, 6), self.src, np.einsum("eij,ej->ei", JsI, z))
        np.subtract.at(b.reshape(n, 6), self.dst, np.einsum("eij,ej->ei", JtI, z))
        return H, b


DENSE_LIMIT = 3000  # unknowns up to which (H + lambda I) is factorised densely


def _solve_damped(H, lam: float, b: np.ndarray) -> np.ndarray:
    n = H.shape[0]
    if n <= DENSE_LIMIT:
        A = H.toarray()
        A[np.diag_indices(n)] += lam
        try:
            return scipy.linalg.cho_solve(scipy.linalg.cho_factor(A, lower=True, check_finite=False), b, check_finite=False)
        except scipy.linalg.LinAlgError:
            return scipy.linalg.solve(A, b, assume_a="sym")
    return scipy.sparse.linalg.splu((H + lam * scipy.sparse.identity(n, format="csc")).tocsc()).solve(b)


def _levenberg_marquardt(g: _Graph, crit: ConvergenceCriteria) -> Tuple[np.ndarray, dict]:
    poses = g.poses
    n = poses.shape[0]
    stats = dict(iterations=0, residual_start=0.0, residual=0.0)
    if g.src.size == 0 or n == 0:
        return poses, stats
    z = g.zeta(poses)
    cur = g.residual(z)
    stats["residual_start"] = stats["residual"] = cur
    x = np.concatenate([transform_to_vec6(T) for T in poses])
    H, b = g.linear_system(poses, z)
    lam = 1e-5 * float(H.diagonal().max())
    ni, rho = 2.0, 0.0
    stop = float(np.abs(b).max()) < crit.min_right_term
    it = 0
    while not stop:
        lm = 0
        while True:
            delta = _solve_damped(H, lam, b)
            stop = stop or float(np.linalg.norm(delta)) < crit.min_relative_increment * (float(np.linalg.norm(x)) + crit.min_relative_increment)
            if not stop:
                new_poses = np.stack([vec6_to_transform(delta[6 * i:6 * i + 6]) @ poses[i] for i in range(n)])
                z_new = g.zeta(new_poses)
                new = g.residual(z_new)
                rho = (cur - new) / (float(delta @ (lam * delta + b)) + 1e-3)
                if rho > 0:
                    stop = stop or (cur - new) < crit.min_relative_residual_increment * cur
                    if stop:
                        break
                    alpha = min(1.0 - (2.0 * rho - 1.0) ** 3, crit.upper_scale_factor)
                    lam *= max(crit.lower_scale_factor, alpha)
                    ni = 2.0
                    cur, z, poses = new, z_new, new_poses
                    x = np.concatenate([transform_to_vec6(T) for T in poses])
                    H, b = g.linear_system(poses, z)
                    stop = stop or float(np.abs(b).max()) < crit.min_right_term
                    if stop:
                        break
                else:
                    lam *= ni
                    ni *= 2.0
            lm += 1
            stop = stop or lm >= crit.max_iteration_lm
            if rho > 0 or stop:
                break
        it += 1
        stop = stop or cur < crit.min_residual or it >= crit.max_iteration
    stats["iterations"], stats["residual"] = it, cur
    return poses, stats



def global_optimization(poses, edges: Iterable[Tuple[int, int, np.ndarray, np.ndarray]], reference_node: int = 0,
                        criteria: ConvergenceCriteria = None, return_stats: bool = False):
    """numpy counterpart of deeppointmap_amd.posegraph_optim.global_optimization (same arguments and return values)"""
    edges = list(edges)
    g = _Graph(np.asarray(poses), edges)
    n = g.poses.shape[0]
    crit = criteria or ConvergenceCriteria()
    original = g.poses.copy()
    first, s1 = _levenberg_marquardt(g, crit)
    g.poses = first
    second, s2 = _levenberg_marquardt(g, crit)
    if n:
        comp = original[reference_node] @ _inv(second[reference_node])
        second = comp[None] @ second
        second[reference_node] = original[reference_node]
    if return_stats:
        return second, dict(first=s1, second=s2)
    return second
