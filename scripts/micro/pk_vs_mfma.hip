// Do packed fp32 vector instructions (v_pk_fma_f32) compute correctly while ANOTHER wave executes bf16 matrix instructions?
// Round 4 met wrong maxima in the encoder's first-level gather kernel (52 packed instructions) whenever the bf16x3 GEMM -- or
// any kernel issuing v_mfma_f32_16x16x32_bf16 -- ran on another stream or in another process; compiled without packed fp32
// instructions the kernel was right every time (csrc/build.py).  This is the minimal form: 512-thread workgroups, waves 0-3
// issue MFMAs (bf16 or fp32, or nothing), waves 4-7 run the SAME recurrence x <- x * a + b twice -- once as v_pk_fma_f32 on
// register pairs, once as v_fma_f32 on single registers -- and count the lanes whose two results differ (bit for bit they
// must not: both are single-rounded fused multiply-adds of the same operands).
// hipcc --offload-arch=gfx950 -O3 scripts/micro/pk_vs_mfma.hip -o deeppointmap_amd/csrc/build/pk_vs_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;

template <int MODE>   // 0: no matrix waves, 1: v_mfma_f32_16x16x32_bf16, 2: v_mfma_f32_16x16x4_f32
__global__ __launch_bounds__(512) void pk_kernel(unsigned long long *mismatch, int iters, float a, float b) {
    const int role = threadIdx.x >> 8;
    if (role == 0) {
        if (MODE == 0) return;
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 av, bv;
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = (short)(0x3f80 + threadIdx.x + i), bv[i] = (short)(0x3f80 + i);
        const float fa = a + threadIdx.x;
        for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
                else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, b, acc[i], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) mismatch[1] = 1;
        return;
    }
    // vector waves: 8 independent pairs
    f32x2 xp[8];
    float xs[16];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        xp[i] = f32x2{0.5f + 0.001f * (threadIdx.x + i), 0.25f + 0.002f * (threadIdx.x + i)};
        xs[2 * i] = xp[i][0], xs[2 * i + 1] = xp[i][1];
    }
    const f32x2 a2 = f32x2{a, a * 0.999f}, b2 = f32x2{b, b * 1.001f};
    unsigned long long bad = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(xp[i]) : "v"(a2), "v"(b2));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xs[2 * i]) : "v"(a2[0]), "v"(b2[0]));
            asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(xs[2 * i + 1]) : "v"(a2[1]), "v"(b2[1]));
        }
        if ((it & 63) == 63) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                bad += (__float_as_uint(xp[i][0]) != __float_as_uint(xs[2 * i])) + (__float_as_uint(xp[i][1]) != __float_as_uint(xs[2 * i + 1]));
                xp[i] = f32x2{xs[2 * i], xs[2 * i + 1]};   // resynchronise: every divergence is counted once
            }
        }
    }
    if (bad) atomicAdd(mismatch, bad);
}

int main() {
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 16);
    const int iters = 200000;
    printf("| matrix waves (one per SIMD, next to one vector wave per SIMD) | lanes x checks whose v_pk_fma_f32 result differed from v_fma_f32 | ms |\n|---|---|---|\n");
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 2; ++rep) {
            (void)hipMemset(d, 0, 16);
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
            (void)hipEventRecord(e0);
            if (mode == 0) hipLaunchKernelGGL(pk_kernel<0>, dim3(1024), dim3(512), 0, 0, d, iters, 0.9999f, 0.0001f);
            if (mode == 1) hipLaunchKernelGGL(pk_kernel<1>, dim3(1024), dim3(512), 0, 0, d, iters, 0.9999f, 0.0001f);
            if (mode == 2) hipLaunchKernelGGL(pk_kernel<2>, dim3(1024), dim3(512), 0, 0, d, iters, 0.9999f, 0.0001f);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("| %s | %llu of %.3g | %.1f |\n", mode == 0 ? "none" : mode == 1 ? "v_mfma_f32_16x16x32_bf16" : "v_mfma_f32_16x16x4_f32", h[0],
                   1024.0 * 256 * 16 * (iters / 64), ms);
        }
    return 0;
}
