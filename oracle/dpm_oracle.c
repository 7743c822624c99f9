/* TEST INFRASTRUCTURE -- NOT PART OF THE PRODUCT PATH.
 *
 * Plain-C restatement of the two serial/brute-force pieces of the hot path, used by the tests
 * (and the cpu_baseline leg of bench.py) where the torch restatement in dpm_oracle.py is too
 * slow at full size.  Built by oracle/Makefile into oracle/libdpm_oracle.so with
 * -ffp-contract=off so the arithmetic below is evaluated exactly as written.
 *
 *  - dpm_oracle_fps:  farthest point sampling, reference network/encoder/utils.py:232-262.
 *    Checked bit-for-bit against the torch loop (fps_indices) and against the reference's own
 *    output in tests/test_oracle_golden.py.
 *  - dpm_oracle_nn1:  exact nearest neighbour (direct-form squared distance), the search the
 *    reference delegates to pytorch3d.knn_points(K=1) at system/modules/utils.py:80.
 */
#include <math.h>
#include <stdint.h>

/* xyz: n_valid x 3 row-major.  idx_out: K entries, -1 where n_valid < K.
 * dist = (dx*dx + dy*dy) + dz*dz in fp32, closest = min(dist, closest), next = FIRST argmax. */
int dpm_oracle_fps(const float *xyz, int n_valid, int K, int64_t *idx_out, float *closest /* n_valid scratch */)
{
    for (int i = 0; i < K; ++i) idx_out[i] = -1;
    if (n_valid <= 0) return 0;
    for (int i = 0; i < n_valid; ++i) closest[i] = INFINITY;
    int sel = 0;
    idx_out[0] = 0;
    int kn = n_valid < K ? n_valid : K;
    for (int r = 1; r < kn; ++r) {
        const float sx = xyz[3 * sel], sy = xyz[3 * sel + 1], sz = xyz[3 * sel + 2];
        float best = -1.0f;
        int besti = 0;
        for (int i = 0; i < n_valid; ++i) {
            const float dx = sx - xyz[3 * i], dy = sy - xyz[3 * i + 1], dz = sz - xyz[3 * i + 2];
            const float d = (dx * dx + dy * dy) + dz * dz;
            const float c = d < closest[i] ? d : closest[i];
            closest[i] = c;
            if (c > best) { best = c; besti = i; }
        }
        sel = besti;
        idx_out[r] = sel;
    }
    return kn;
}

/* p1: n1 x 3, p2: n2 x 3 -> d[n1] (squared distance), idx[n1] (first index at the minimum). */
void dpm_oracle_nn1(const float *p1, int n1, const float *p2, int n2, float *d, int32_t *idx)
{
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n1; ++i) {
        const float x = p1[3 * i], y = p1[3 * i + 1], z = p1[3 * i + 2];
        float best = INFINITY;
        int bi = -1;
        for (int j = 0; j < n2; ++j) {
            const float dx = x - p2[3 * j], dy = y - p2[3 * j + 1], dz = z - p2[3 * j + 2];
            const float s = (dx * dx + dy * dy) + dz * dz;
            if (s < best) { best = s; bi = j; }
        }
        d[i] = best;
        idx[i] = bi;
    }
}
