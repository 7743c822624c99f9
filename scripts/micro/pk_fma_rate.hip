// Issue rate of v_fma_f32 against v_pk_fma_f32 on gfx950 (no MFMA around): the same number of fused multiply-adds as 16
// independent scalar chains or 8 independent packed chains per thread.  hipcc --offload-arch=gfx950 -O3 pk_fma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void scalar_kernel(float *out, float a, float b, int iters) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

__global__ __launch_bounds__(256) void packed_kernel(float *out, float a, float b, int iters) {
    f2 x[8];
    f2 av = {a, a}, bv = {b, b};
#pragma unroll
    for (int i = 0; i < 8; ++i) x[i] = f2{threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(av), "v"(bv));
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i].x + x[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float *out;
    hipMalloc(&out, 256 * 4096 * sizeof(float));
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int iters = 4096;
    int clk = 0;
    hipDeviceGetAttribute(&clk, hipDeviceAttributeClockRate, 0);
    printf("device clock %d kHz\n", clk);
    for (int wg = 1; wg <= 8; wg *= 2) {
        const int blocks = 256 * wg;  // workgroups of 4 waves per CU = waves per SIMD
        for (int which = 0; which < 2; ++which) {
            float ms = 0;
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (which == 0) hipLaunchKernelGGL(scalar_kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
                else hipLaunchKernelGGL(packed_kernel, dim3(blocks), dim3(256), 0, 0, out, 1.0001f, 0.5f, iters);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                hipEventElapsedTime(&ms, e0, e1);
            }
            const double fma = (double)blocks * 256 * 16 * iters;
            printf("%s %d workgroups/CU: %.3f ms, %.1f TFLOP/s (fp32 FMA = 2 flop)\n", which ? "v_pk_fma_f32" : "v_fma_f32   ", wg,
                   ms, 2 * fma / ms / 1e9);
        }
    }
    return 0;
}
