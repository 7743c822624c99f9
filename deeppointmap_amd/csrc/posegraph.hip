// Pose-graph optimisation of the SLAM back end -- HOST code (no kernel: one 6-vector per key-frame,
// sequential on rank 0 by design), native like the open3d routine it replaces:
//   o3d.pipelines.registration.global_optimization(graph, GlobalOptimizationLevenbergMarquardt(),
//       GlobalOptimizationConvergenceCriteria(), GlobalOptimizationOption(edge_prune_threshold=0, preference_loop_closure=2,
//       reference_node=...))                                  (reference system/modules/pose_graph.py:565-613)
// on a graph whose edges are all certain (pose_graph.py:597).  open3d 0.16.0 is absent here: this restates its published
// algorithm (cpp/open3d/pipelines/registration/GlobalOptimization.cpp); deeppointmap_amd/posegraph_optim.py has the full
// description, oracle/posegraph_numpy.py the numpy statement of the same iteration that the tests hold this file to.
//
//   residual of edge (s, t):  e = vec6(X^-1 T_t^-1 T_s),  vec6 = (rx, ry, rz of R = Rz Ry Rx, translation)
//   objective:                sum_e e^T Lambda e
//   Levenberg-Marquardt on T_i <- expm6(delta_i) T_i with the linearised Jacobians, lambda_0 = 1e-5 max diag(H), Nielsen
//   update; two passes; afterwards every pose is moved so that the reference node is back where it started.
//
// Normal equations: (H + lambda I) delta = b with one 6 x 6 block per key-frame and per edge.  The nodes are renumbered
// by reverse Cuthill-McKee (an odometry chain with loop closures becomes a narrow band: both arcs of a loop interleave)
// and the matrix is factorised as a skyline (envelope) Cholesky: row i keeps the columns from its first non-zero to the
// diagonal, fill never leaves the envelope, the inner loop is a dot product of two contiguous row pieces.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include "dpm_common.h"

namespace {

typedef double M4[16];  // row-major 4 x 4

inline void mul4(const double *A, const double *B, double *C) {
    double t[16];
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j) {
            double s = 0;
            for (int k = 0; k < 4; ++k) s += A[4 * i + k] * B[4 * k + j];
            t[4 * i + j] = s;
        }
    std::memcpy(C, t, sizeof t);
}
inline void inv_rigid(const double *T, double *O) {  // [R t; 0 1]^-1 = [R^T -R^T t; 0 1]
    double t[16] = {0};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) t[4 * i + j] = T[4 * j + i];
    for (int i = 0; i < 3; ++i) t[4 * i + 3] = -(t[4 * i] * T[3] + t[4 * i + 1] * T[7] + t[4 * i + 2] * T[11]);
    t[15] = 1;
    std::memcpy(O, t, sizeof t);
}
inline void to_vec6(const double *T, double *v) {  // open3d TransformMatrix4dToVector6d
    const double sy = std::sqrt(T[0] * T[0] + T[4] * T[4]);
    if (sy >= 1e-6) v[0] = std::atan2(T[9], T[10]), v[1] = std::atan2(-T[8], sy), v[2] = std::atan2(T[4], T[0]);
    else v[0] = std::atan2(-T[6], T[5]), v[1] = std::atan2(-T[8], sy), v[2] = 0.0;
    v[3] = T[3], v[4] = T[7], v[5] = T[11];
}
inline void from_vec6(const double *v, double *T) {  // open3d TransformVector6dToMatrix4d: R = Rz Ry Rx
    const double cx = std::cos(v[0]), sx = std::sin(v[0]), cy = std::cos(v[1]), sy = std::sin(v[1]), cz = std::cos(v[2]),
                 sz = std::sin(v[2]);
    const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx}, Ry[9] = {cy, 0, sy, 0, 1, 0, -sy, 0, cy},
                 Rz[9] = {cz, -sz, 0, sz, cz, 0, 0, 0, 1};
    double A[9], R[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += Rz[3 * i + k] * Ry[3 * k + j];
            A[3 * i + j] = s;
        }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += A[3 * i + k] * Rx[3 * k + j];
            R[3 * i + j] = s;
        }
    std::memset(T, 0, 16 * sizeof(double));
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) T[4 * i + j] = R[3 * i + j];
        T[4 * i + 3] = v[3 + i];
    }
    T[15] = 1;
}

struct Graph {
    int n = 0, E = 0;
    std::vector<double> poses;           // n * 16
    const int32_t *src = nullptr, *dst = nullptr;
    std::vector<double> Xinv;            // E * 16
    const double *info = nullptr;        // E * 36
    // skyline structure over the renumbered nodes
    std::vector<int> perm;               // node -> position
    std::vector<int> first;              // scalar row -> first stored column
    std::vector<size_t> rowptr;          // scalar row -> offset of its first stored entry
};

void zeta(const Graph &g, const std::vector<double> &poses, std::vector<double> &z) {
    z.resize((size_t)g.E * 6);
    for (int k = 0; k < g.E; ++k) {
        M4 a, b;
        inv_rigid(&poses[(size_t)g.dst[k] * 16], a);
        mul4(&g.Xinv[(size_t)k * 16], a, b);
        mul4(b, &poses[(size_t)g.src[k] * 16], a);
        to_vec6(a, &z[(size_t)k * 6]);
    }
}
double residual(const Graph &g, const std::vector<double> &z) {
    double r = 0;
    for (int k = 0; k < g.E; ++k)
        for (int i = 0; i < 6; ++i)
            for (int j = 0; j < 6; ++j) r += z[(size_t)k * 6 + i] * g.info[(size_t)k * 36 + 6 * i + j] * z[(size_t)k * 6 + j];
    return r;
}

// reverse Cuthill-McKee over the node graph (every component from a minimum-degree node)
void order_nodes(Graph &g) {
    const int n = g.n;
    std::vector<std::vector<int>> adj(n);
    for (int k = 0; k < g.E; ++k)
        if (g.src[k] != g.dst[k]) adj[g.src[k]].push_back(g.dst[k]), adj[g.dst[k]].push_back(g.src[k]);
    for (auto &a : adj) {
        std::sort(a.begin(), a.end());
        a.erase(std::unique(a.begin(), a.end()), a.end());
    }
    std::vector<int> order, by_deg(n);
    std::vector<char> seen(n, 0);
    for (int i = 0; i < n; ++i) by_deg[i] = i;
    std::stable_sort(by_deg.begin(), by_deg.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
    for (int root : by_deg) {
        if (seen[root]) continue;
        // pseudo-peripheral start: walk to the far end of the component twice
        int start = root;
        for (int rep = 0; rep < 2; ++rep) {
            std::vector<int> q{start};
            std::vector<char> s2(n, 0);
            s2[start] = 1;
            for (size_t h = 0; h < q.size(); ++h)
                for (int v : adj[q[h]])
                    if (!s2[v] && !seen[v]) s2[v] = 1, q.push_back(v);
            start = q.back();
        }
        size_t head = order.size();
        order.push_back(start), seen[start] = 1;
        for (; head < order.size(); ++head) {
            std::vector<int> nb;
            for (int v : adj[order[head]])
                if (!seen[v]) nb.push_back(v), seen[v] = 1;
            std::stable_sort(nb.begin(), nb.end(), [&](int a, int b) { return adj[a].size() < adj[b].size(); });
            order.insert(order.end(), nb.begin(), nb.end());
        }
    }
    std::reverse(order.begin(), order.end());
    g.perm.assign(n, 0);
    for (int p = 0; p < n; ++p) g.perm[order[p]] = p;
    std::vector<int> first_node(n);
    for (int p = 0; p < n; ++p) first_node[p] = p;
    for (int k = 0; k < g.E; ++k) {
        const int a = g.perm[g.src[k]], b = g.perm[g.dst[k]];
        first_node[std::max(a, b)] = std::min(first_node[std::max(a, b)], std::min(a, b));
    }
    g.first.resize((size_t)6 * n), g.rowptr.resize((size_t)6 * n + 1);
    size_t off = 0;
    for (int p = 0; p < n; ++p)
        for (int c = 0; c < 6; ++c) {
            const int row = 6 * p + c;
            g.first[row] = 6 * first_node[p];
            g.rowptr[row] = off;
            off += (size_t)(row - g.first[row] + 1);
        }
    g.rowptr[(size_t)6 * n] = off;
}

inline double &sky(const Graph &g, std::vector<double> &A, int i, int j) { return A[g.rowptr[i] + (size_t)(j - g.first[i])]; }

// lower skyline of H (renumbered) and b (original numbering) of the Gauss-Newton step (open3d ComputeLinearSystem)
void linear_system(const Graph &g, const std::vector<double> &poses, const std::vector<double> &z, std::vector<double> &H,
                   std::vector<double> &b) {
    static const int GI[6][2][2] = {{{1, 2}, {2, 1}}, {{2, 0}, {0, 2}}, {{0, 1}, {1, 0}}, {{0, 3}, {0, 3}}, {{1, 3}, {1, 3}}, {{2, 3}, {2, 3}}};
    H.assign(g.rowptr.back(), 0.0), b.assign((size_t)6 * g.n, 0.0);
    for (int k = 0; k < g.E; ++k) {
        M4 ti, A, Gm, M;
        inv_rigid(&poses[(size_t)g.dst[k] * 16], ti);
        mul4(&g.Xinv[(size_t)k * 16], ti, A);
        const double *B = &poses[(size_t)g.src[k] * 16];
        double Js[36];  // Js[r][i]: component r of the derivative along generator i
        for (int i = 0; i < 6; ++i) {
            std::memset(Gm, 0, sizeof Gm);
            if (i < 3) Gm[4 * GI[i][0][0] + GI[i][0][1]] = -1, Gm[4 * GI[i][1][0] + GI[i][1][1]] = 1;
            else Gm[4 * GI[i][0][0] + GI[i][0][1]] = 1;
            mul4(A, Gm, M);
            mul4(M, B, M);
            Js[6 * 0 + i] = (-M[6] + M[9]) / 2, Js[6 * 1 + i] = (-M[8] + M[2]) / 2, Js[6 * 2 + i] = (-M[1] + M[4]) / 2;
            Js[6 * 3 + i] = M[3], Js[6 * 4 + i] = M[7], Js[6 * 5 + i] = M[11];
        }
        const double *L = &g.info[(size_t)k * 36];
        double JsI[36], Q[36], v[6];
        for (int i = 0; i < 6; ++i)
            for (int c = 0; c < 6; ++c) {
                double s = 0;
                for (int j = 0; j < 6; ++j) s += Js[6 * j + i] * L[6 * j + c];
                JsI[6 * i + c] = s;
            }
        for (int i = 0; i < 6; ++i) {
            for (int c = 0; c < 6; ++c) {
                double s = 0;
                for (int j = 0; j < 6; ++j) s += JsI[6 * i + j] * Js[6 * j + c];
                Q[6 * i + c] = s;
            }
            double s = 0;
            for (int j = 0; j < 6; ++j) s += JsI[6 * i + j] * z[(size_t)k * 6 + j];
            v[i] = s;
        }
        const int s_ = g.src[k], t_ = g.dst[k], ps = g.perm[s_], pt = g.perm[t_];
        for (int i = 0; i < 6; ++i) {
            b[(size_t)6 * s_ + i] -= v[i], b[(size_t)6 * t_ + i] += v[i];
            for (int c = 0; c <= i; ++c) sky(g, H, 6 * ps + i, 6 * ps + c) += Q[6 * i + c], sky(g, H, 6 * pt + i, 6 * pt + c) += Q[6 * i + c];
        }
        if (ps != pt) {
            const int hi = std::max(ps, pt), lo = std::min(ps, pt);
            // block (hi, lo) of H is -Q whichever end is the source: J_s^T Lambda J_t = J_t^T Lambda J_s = -J_s^T Lambda J_s
            for (int i = 0; i < 6; ++i)
                for (int c = 0; c < 6; ++c) sky(g, H, 6 * hi + i, 6 * lo + c) -= Q[6 * i + c];
        } else {  // a self edge contributes Q - Q - Q + Q = 0 to its diagonal block
            for (int i = 0; i < 6; ++i)
                for (int c = 0; c <= i; ++c) sky(g, H, 6 * ps + i, 6 * ps + c) -= Q[6 * i + c] + Q[6 * c + i];
        }
    }
}

// solves (H + lam I) x = b; H lower skyline in the renumbered order, b / x in the original node order.  false: not SPD
bool solve_damped(const Graph &g, const std::vector<double> &H, double lam, const std::vector<double> &b, std::vector<double> &x) {
    const int N = 6 * g.n;
    std::vector<double> Lm(H);
    for (int i = 0; i < N; ++i) {
        sky(g, Lm, i, i) += lam;
        double *ri = &Lm[g.rowptr[i]];
        const int fi = g.first[i];
        for (int j = fi; j <= i; ++j) {
            const int fj = g.first[j], k0 = std::max(fi, fj);
            const double *rj = &Lm[g.rowptr[j]];
            double s = ri[j - fi];
            const double *pa = ri + (k0 - fi), *pb = rj + (k0 - fj);
            const int len = j - k0;
            for (int k = 0; k < len; ++k) s -= pa[k] * pb[k];
            if (j < i) ri[j - fi] = s / rj[j - fj];
            else {
                if (!(s > 0.0)) return false;
                ri[j - fi] = std::sqrt(s);
            }
        }
    }
    std::vector<double> y((size_t)N);
    for (int node = 0; node < g.n; ++node)
        for (int c = 0; c < 6; ++c) y[(size_t)6 * g.perm[node] + c] = b[(size_t)6 * node + c];
    for (int i = 0; i < N; ++i) {  // L y = b
        const double *ri = &Lm[g.rowptr[i]];
        const int fi = g.first[i];
        double s = y[i];
        for (int k = fi; k < i; ++k) s -= ri[k - fi] * y[k];
        y[i] = s / ri[i - fi];
    }
    for (int i = N - 1; i >= 0; --i) {  // L^T x = y, column sweep over the rows
        const double *ri = &Lm[g.rowptr[i]];
        const int fi = g.first[i];
        y[i] /= ri[i - fi];
        const double yi = y[i];
        for (int k = fi; k < i; ++k) y[k] -= ri[k - fi] * yi;
    }
    x.resize((size_t)N);
    for (int node = 0; node < g.n; ++node)
        for (int c = 0; c < 6; ++c) x[(size_t)6 * node + c] = y[(size_t)6 * g.perm[node] + c];
    return true;
}

struct Criteria {  // defaults of open3d's GlobalOptimizationConvergenceCriteria
    int max_iteration = 100;
    double min_relative_increment = 1e-6, min_relative_residual_increment = 1e-6, min_right_term = 1e-6, min_residual = 1e-6;
    int max_iteration_lm = 20;
    double upper_scale_factor = 2.0 / 3.0, lower_scale_factor = 1.0 / 3.0;
};

double max_abs(const std::vector<double> &v) {
    double m = 0;
    for (double x : v) m = std::max(m, std::fabs(x));
    return m;
}
double norm2(const std::vector<double> &v) {
    double s = 0;
    for (double x : v) s += x * x;
    return std::sqrt(s);
}

// one Levenberg-Marquardt pass over g.poses (updated in place); stats: iterations, residual at the start, at the end
int levenberg_marquardt(Graph &g, const Criteria &crit, double *stats) {
    stats[0] = stats[1] = stats[2] = 0;
    if (g.E == 0 || g.n == 0) return DPM_OK;
    std::vector<double> poses = g.poses, z, z_new, H, b, delta, x((size_t)6 * g.n), new_poses((size_t)16 * g.n);
    zeta(g, poses, z);
    double cur = residual(g, z);
    stats[1] = stats[2] = cur;
    auto refresh_x = [&] {
        for (int i = 0; i < g.n; ++i) to_vec6(&poses[(size_t)16 * i], &x[(size_t)6 * i]);
    };
    refresh_x();
    linear_system(g, poses, z, H, b);
    double hmax = 0;
    for (int i = 0; i < 6 * g.n; ++i) hmax = std::max(hmax, H[g.rowptr[i] + (size_t)(i - g.first[i])]);
    double lam = 1e-5 * hmax, ni = 2.0, rho = 0.0;
    bool stop = max_abs(b) < crit.min_right_term;
    int it = 0;
    while (!stop) {
        int lm = 0;
        while (true) {
            if (!solve_damped(g, H, lam, b, delta)) {
                // the damped matrix is not positive definite (a degenerate or indefinite information matrix): more damping,
                // as after a rejected step; out of tries, the pass ends with the poses it has (open3d reports a failed solve
                // and carries on with the current poses as well)
                lam = std::max(lam * ni, 1e-12 * std::max(hmax, 1.0)), ni *= 2.0;
                if (++lm >= crit.max_iteration_lm) {
                    stop = true;
                    break;
                }
                continue;
            }
            stop = stop || norm2(delta) < crit.min_relative_increment * (norm2(x) + crit.min_relative_increment);
            if (!stop) {
                for (int i = 0; i < g.n; ++i) {
                    M4 d;
                    from_vec6(&delta[(size_t)6 * i], d);
                    mul4(d, &poses[(size_t)16 * i], &new_poses[(size_t)16 * i]);
                }
                zeta(g, new_poses, z_new);
                const double nw = residual(g, z_new);
                double den = 1e-3;
                for (size_t i = 0; i < delta.size(); ++i) den += delta[i] * (lam * delta[i] + b[i]);
                rho = (cur - nw) / den;
                if (rho > 0) {
                    stop = stop || (cur - nw) < crit.min_relative_residual_increment * cur;
                    if (stop) break;
                    const double alpha = std::min(1.0 - std::pow(2.0 * rho - 1.0, 3), crit.upper_scale_factor);
                    lam *= std::max(crit.lower_scale_factor, alpha);
                    ni = 2.0;
                    cur = nw, z.swap(z_new), poses.swap(new_poses);
                    new_poses.resize(poses.size());
                    refresh_x();
                    linear_system(g, poses, z, H, b);
                    stop = stop || max_abs(b) < crit.min_right_term;
                    if (stop) break;
                } else {
                    lam *= ni;
                    ni *= 2.0;
                }
            }
            ++lm;
            stop = stop || lm >= crit.max_iteration_lm;
            if (rho > 0 || stop) break;
        }
        ++it;
        stop = stop || cur < crit.min_residual || it >= crit.max_iteration;
    }
    g.poses = poses;
    stats[0] = it, stats[2] = cur;
    return DPM_OK;
}

}  // namespace

extern "C" int dpm_posegraph_optimize(const double *poses, int n, const int32_t *src, const int32_t *dst, const double *X,
                                      const double *info, int E, int reference_node, const double *criteria,
                                      double *out_poses, double *stats) {
    DPM_CHECK_ARG(n >= 0 && E >= 0 && out_poses && stats && (n == 0 || poses) && (E == 0 || (src && dst && X && info)));
    DPM_CHECK_ARG(n == 0 || (reference_node >= 0 && reference_node < n));
    for (int k = 0; k < E; ++k) DPM_CHECK_ARG(src[k] >= 0 && src[k] < n && dst[k] >= 0 && dst[k] < n);
    for (size_t i = 0; i < (size_t)16 * n; ++i) DPM_CHECK_ARG(std::isfinite(poses[i]));
    for (size_t i = 0; i < (size_t)16 * E; ++i) DPM_CHECK_ARG(std::isfinite(X[i]));
    // the assembly reads an information matrix as symmetric: hand it its symmetric part (callers drop non-finite edges)
    std::vector<double> info_sym((size_t)36 * E);
    for (int k = 0; k < E; ++k)
        for (int r = 0; r < 6; ++r)
            for (int c = 0; c < 6; ++c) {
                const double v = 0.5 * (info[(size_t)36 * k + 6 * r + c] + info[(size_t)36 * k + 6 * c + r]);
                DPM_CHECK_ARG(std::isfinite(v));
                info_sym[(size_t)36 * k + 6 * r + c] = v;
            }
    Graph g;
    g.n = n, g.E = E, g.src = src, g.dst = dst, g.info = info_sym.data();
    g.poses.assign(poses, poses + (size_t)16 * n);
    g.Xinv.resize((size_t)16 * E);
    for (int k = 0; k < E; ++k) inv_rigid(X + (size_t)16 * k, &g.Xinv[(size_t)16 * k]);
    for (int i = 0; i < 6; ++i) stats[i] = 0;
    if (n == 0) return DPM_OK;
    order_nodes(g);
    const std::vector<double> original = g.poses;
    Criteria crit;
    if (criteria) {
        crit.max_iteration = (int)criteria[0], crit.min_relative_increment = criteria[1];
        crit.min_relative_residual_increment = criteria[2], crit.min_right_term = criteria[3], crit.min_residual = criteria[4];
        crit.max_iteration_lm = (int)criteria[5], crit.upper_scale_factor = criteria[6], crit.lower_scale_factor = criteria[7];
    }
    int rc = levenberg_marquardt(g, crit, stats);
    // second pass of open3d's GlobalOptimization: it follows the pruning of uncertain edges -- there are none -- and
    // starts from the first pass's result
    if (rc == DPM_OK) rc = levenberg_marquardt(g, crit, stats + 3);
    if (rc != DPM_OK) return rc;
    M4 inv_ref, comp;
    inv_rigid(&g.poses[(size_t)16 * reference_node], inv_ref);
    mul4(&original[(size_t)16 * reference_node], inv_ref, comp);
    for (int i = 0; i < n; ++i) mul4(comp, &g.poses[(size_t)16 * i], out_poses + (size_t)16 * i);
    std::memcpy(out_poses + (size_t)16 * reference_node, &original[(size_t)16 * reference_node], 16 * sizeof(double));
    return DPM_OK;
}
