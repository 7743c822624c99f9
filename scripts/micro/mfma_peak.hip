// What does the fp32 matrix pipe deliver when NOTHING else is in the way?  Register-only loops of independent
// v_mfma_f32_16x16x4_f32 / v_mfma_f32_32x32x2_f32 accumulations, W waves per SIMD on all 256 CUs.  The figure the GEMM
// kernels are priced against in DESIGN.md (157.3 TFLOP/s = 256 CUs x 256 FLOP/clk x 2.4 GHz) assumes the boost clock.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int ACC>
__global__ __launch_bounds__(256) void mfma16(float *out, int iters, float a, float b) {
    f32x4 acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

template <int ACC>
__global__ __launch_bounds__(256) void mfma32(float *out, int iters, float a, float b) {
    f32x16 acc[ACC];
#pragma unroll
    for (int i = 0; i < ACC; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.f;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < ACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[i], 0, 0, 0);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < ACC; ++i) s += acc[i][0] + acc[i][5];
    if (s == 12345.678f) out[0] = s;
}

int main() {
    float *out;
    (void)hipMalloc(&out, 64);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    const int iters = 4000;
    for (int kind = 0; kind < 2; ++kind)
        for (int wps : {1, 2, 4, 8}) {          // waves per SIMD: a 256-thread workgroup puts one wave on each SIMD
            const int grid = 256 * wps;
            float best = 1e9f;
            for (int rep = 0; rep < 3; ++rep) {
                (void)hipEventRecord(e0);
                if (kind == 0) hipLaunchKernelGGL(mfma16<4>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
                else hipLaunchKernelGGL(mfma32<2>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
                (void)hipEventRecord(e1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                best = ms < best ? ms : best;
            }
            const double n_mfma = (double)grid * 4 * iters * (kind == 0 ? 8 * 4 : 4 * 2);
            const double flop = n_mfma * (kind == 0 ? 16 * 16 * 4 * 2 : 32 * 32 * 2 * 2);
            printf("%s  %d waves/SIMD: %.3f ms  %.1f TFLOP/s\n", kind == 0 ? "v_mfma_f32_16x16x4_f32" : "v_mfma_f32_32x32x2_f32", wps,
                   best, flop / (best * 1e-3) / 1e12);
        }
    return 0;
}
