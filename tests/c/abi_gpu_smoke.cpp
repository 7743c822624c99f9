// The C ABI driven WITHOUT torch: a small HIP host program (hipMalloc / hipMemcpy only) calls dpm_fps on a ragged batch
// and checks the picks against a plain loop of the reference's rule (network/encoder/utils.py:232-262: start at point 0,
// dist = (dx*dx + dy*dy) + dz*dz in fp32 without contraction, next pick = first arg-max of the running minimum).
// Built with hipcc and run by tests/test_gpu_ops.py::test_c_abi_from_a_program_without_torch.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "dpm_hip.h"

#define CK(x)                                                          \
    do {                                                               \
        hipError_t e_ = (x);                                           \
        if (e_ != hipSuccess) {                                        \
            std::printf("HIP error %d at line %d\n", (int)e_, __LINE__); \
            return 10;                                                 \
        }                                                              \
    } while (0)

int main() {
    const int B = 3, N = 20000, K = 512;
    const int len[B] = {20000, 12345, 300};  // the last frame is shorter than K: padded picks
    std::vector<float> xyz((size_t)B * N * 3);
    unsigned s = 12345u;
    for (auto &v : xyz) s = s * 1664525u + 1013904223u, v = (float)(s >> 8) / 16777216.0f * 2.f - 1.f;
    float *d_xyz, *d_new;
    int32_t *d_len, *d_idx, *d_nl;
    void *d_ws;
    const size_t ws = dpm_fps_workspace_bytes(B, N, K);
    std::fprintf(stderr, "workspace %zu bytes\n", ws);
    CK(hipMalloc(&d_xyz, xyz.size() * 4));
    CK(hipMalloc(&d_new, (size_t)B * K * 12));
    CK(hipMalloc(&d_len, B * 4));
    CK(hipMalloc(&d_idx, (size_t)B * K * 4));
    CK(hipMalloc(&d_nl, B * 4));
    CK(hipMalloc(&d_ws, ws));
    CK(hipMemcpy(d_xyz, xyz.data(), xyz.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_len, len, B * 4, hipMemcpyHostToDevice));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    std::fprintf(stderr, "buffers ready, launching\n");
    const int rc = dpm_fps(d_xyz, d_len, B, N, K, d_idx, d_new, d_nl, d_ws, (dpm_stream_t)st);
    if (rc != 0) {
        std::printf("dpm_fps: %d (%s)\n", rc, dpm_error_string(rc));
        return 11;
    }
    std::fprintf(stderr, "launched, rc %d\n", rc);
    CK(hipStreamSynchronize(st));
    std::fprintf(stderr, "synchronised\n");
    std::vector<int32_t> idx((size_t)B * K), nl(B);
    std::vector<float> nw((size_t)B * K * 3);
    CK(hipMemcpy(idx.data(), d_idx, idx.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nl.data(), d_nl, B * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(nw.data(), d_new, nw.size() * 4, hipMemcpyDeviceToHost));
    for (int b = 0; b < B; ++b) {
        const float *p = &xyz[(size_t)b * N * 3];
        std::vector<float> closest(len[b], __builtin_inff());
        int cur = 0;
        const int kn = len[b] < K ? len[b] : K;
        if (nl[b] != kn) return 20 + b;
        for (int k = 0; k < K; ++k) {
            const int got = idx[(size_t)b * K + k];
            if (k >= kn) {
                if (got != -1 || nw[((size_t)b * K + k) * 3] != 0.f) return 30 + b;
                continue;
            }
            if (got != cur) {
                std::printf("frame %d pick %d: %d, expected %d\n", b, k, got, cur);
                return 40 + b;
            }
            for (int a = 0; a < 3; ++a)
                if (nw[((size_t)b * K + k) * 3 + a] != p[3 * cur + a]) return 50 + b;
            float best = -1.f;
            int arg = 0;
            for (int i = 0; i < len[b]; ++i) {
                const float dx = p[3 * i] - p[3 * cur], dy = p[3 * i + 1] - p[3 * cur + 1], dz = p[3 * i + 2] - p[3 * cur + 2];
                const float d = (dx * dx + dy * dy) + dz * dz;
                if (d < closest[i]) closest[i] = d;
                if (closest[i] > best) best = closest[i], arg = i;
            }
            cur = arg;
        }
    }
    std::printf("c abi gpu ok: %d frames x %d picks equal the plain loop\n", B, K);
    return 0;
}
