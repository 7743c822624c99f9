/* Plain C against the C ABI (include/dpm_hip.h): no Python, no torch, no GPU -- the host entry points only.
 * Built and run by tests/test_abi.py::test_plain_c_program_links_and_runs. */
#include <math.h>
#include <stdio.h>
#include <string.h>

#include "dpm_hip.h"

static void eye(double *T) {
    memset(T, 0, 16 * sizeof(double));
    T[0] = T[5] = T[10] = T[15] = 1.0;
}

int main(void) {
    if (dpm_version() < 1000) return 1;
    if (strcmp(dpm_error_string(0), "ok") != 0) return 2;

    /* torch.topk replay: largest 3 of a row with ties */
    const float v[8] = {1.f, 5.f, 5.f, 2.f, 5.f, 0.f, 5.f, 3.f};
    int32_t idx[3];
    if (dpm_host_topk_replay(v, 8, 3, 1, idx) != 0) return 3;
    for (int i = 0; i < 3; ++i)
        if (v[idx[i]] != 5.f) return 4;

    /* pose graph: two nodes, one edge; node 1 is off its measurement and must be moved onto it */
    double poses[32], X[16], info[36], out[32], stats[6];
    eye(poses), eye(poses + 16), eye(X);
    poses[16 + 3] = 2.5;      /* node 1 at x = 2.5 ... */
    X[3] = -2.0;              /* ... but the edge says: source (node 0) sits at x = -2 in node 1's frame, i.e. node 1 at x = 2 */
    memset(info, 0, sizeof info);
    for (int i = 0; i < 6; ++i) info[7 * i] = 10.0;
    const int32_t src[1] = {0}, dst[1] = {1};
    if (dpm_posegraph_optimize(poses, 2, src, dst, X, info, 1, 0, NULL, out, stats) != 0) return 5;
    /* (the optimiser stops by open3d's rules, a few 1e-6 short of the exact minimiser) */
    if (fabs(out[3]) > 1e-12 || fabs(out[16 + 3] - 2.0) > 1e-5) return 6;
    if (!(stats[2] < stats[1])) return 7;
    printf("c abi ok: topk %d %d %d, node 1 x = %.9f, residual %.3e -> %.3e\n", idx[0], idx[1], idx[2], out[16 + 3], stats[1], stats[2]);
    return 0;
}
