// Can the chip run a matrix-pipe kernel and a vector-ALU kernel AT THE SAME TIME from two HIP streams -- in the cleanest
// possible case, and as the kernels get less pure?  (Round 4's decision gate; the product kernels are measured by
// scripts/corun_micro.py.)
//   M(lds)  register-only loop of independent v_mfma_f32_16x16x4_f32; lds > 0 adds that many ds_read_b32 per MFMA
//           (the 64 x 64 GEMM issues ~1.1 LDS reads and ~0.7 other vector instructions per MFMA)
//   V       register-only loop of independent v_fma_f32 chains
// every kernel is launched as 256 x w workgroups of 256 threads (w waves per SIMD when the dispatcher spreads them evenly;
// w_M + w_V <= 8, so both launches are resident together), alone and on two streams; printed: concurrent / (alone_M + alone_V), and max / sum = perfect overlap.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/corun_pure.hip -o deeppointmap_amd/csrc/build/corun_pure
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

using bf16x8 = __attribute__((ext_vector_type(8))) short;
template <int LDS>
__global__ __launch_bounds__(256) void mfma_kernel(float *out, int iters, float a, float b) {
    __shared__ float tile[64 * 34];
    for (int i = threadIdx.x; i < 64 * 34; i += 256) tile[i] = a + i;
    __syncthreads();
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int lane = threadIdx.x & 63;
    float av = a + threadIdx.x, bv = b;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (LDS >= 1) av = tile[((lane & 15) + i) * 34 + (lane >> 4) + 4 * u + (it & 7)];
                if (LDS >= 2) bv = tile[((lane & 15) + i + 16) * 34 + (lane >> 4) + 4 * u + (it & 7)];
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[i], 0, 0, 0);
            }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

// bf16 matrix instruction (the REAL matrix pipe: 16x the fp32 rate), register-only; CVT > 0 adds that many vector
// instructions per MFMA (what splitting fp32 operands into bf16 pieces costs when it is done next to the MFMAs)
template <int CVT>
__global__ __launch_bounds__(256) void mfma_bf16_kernel(float *out, int iters, float a, float b) {
    f32x4 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    bf16x8 av, bv;
#pragma unroll
    for (int i = 0; i < 8; ++i) av[i] = (short)(0x3f80 + threadIdx.x + i), bv[i] = (short)(0x3f80 + i);
    float x = a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
#pragma unroll
                for (int c = 0; c < CVT; ++c) x = fmaf(x, b, a);
                if (CVT) av[0] = (short)(__float_as_uint(x) >> 16);
                acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
            }
    }
    float s = x;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    if (s == 12345.678f) out[0] = s;
}

__global__ __launch_bounds__(256) void valu_kernel(float *out, int iters, float a, float b) {
    float x[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = a + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], b, a);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += x[i];
    if (s == 12345.678f) out[0] = s;
}

// ONE launch, role-specialised waves: a 512-thread workgroup puts two waves on every SIMD; waves 0-3 run the matrix loop,
// waves 4-7 the vector loop (mode 3), or one half exits at once (mode 1 = matrix only, 2 = vector only)
// GAP: the matrix waves idle 16 cycles (s_nop 15) after every MFMA -- the matrix pipe is then at most half busy and the
// waves are NOT waiting at the issue port; PRIO: the vector waves run at s_setprio 3
template <bool BF, int GAP = 0, bool PRIO = false>
__global__ __launch_bounds__(512) void mixed_kernel(float *out, int itm, int itv, float a, float b, int mode) {
    const int role = threadIdx.x >> 8;   // 0: matrix waves, 1: vector waves
    if (!((mode >> role) & 1)) return;
    if (PRIO && role == 1) __builtin_amdgcn_s_setprio(3);
    float s = 0.f;
    if (role == 0) {
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 av, bv;
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = (short)(0x3f80 + threadIdx.x + i), bv[i] = (short)(0x3f80 + i);
        const float fa = a + threadIdx.x;
        for (int it = 0; it < itm; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    if (BF) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
                    else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, b, acc[i], 0, 0, 0);
                    if (GAP > 0) asm volatile("s_nop %0" ::"n"(GAP > 0 ? GAP - 1 : 0));
                }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    } else {
        float x[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) x[i] = a + i + threadIdx.x;
        for (int it = 0; it < itv; ++it) {
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < 16; ++i) x[i] = fmaf(x[i], b, a);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += x[i];
    }
    if (s == 12345.678f) out[0] = s;
}

static hipStream_t s1, s2;
static hipEvent_t e0, e1, e2;
static float *out;

template <class FA, class FB>
static float timed(FA fa, FB fb, bool a, bool b) {
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0, 0);
    (void)hipStreamWaitEvent(s1, e0, 0), (void)hipStreamWaitEvent(s2, e0, 0);
    if (a) fa();
    if (b) fb();
    (void)hipEventRecord(e1, s1), (void)hipEventRecord(e2, s2);
    (void)hipStreamWaitEvent(0, e1, 0), (void)hipStreamWaitEvent(0, e2, 0);
    hipEvent_t e3;
    (void)hipEventCreate(&e3);
    (void)hipEventRecord(e3, 0);
    (void)hipEventSynchronize(e3);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e3);
    (void)hipEventDestroy(e3);
    return ms;
}

template <int LDS, int BF = -1>
static void sweep(const char *name) {
    printf("\n| %s: waves/SIMD matrix | waves/SIMD vector | M alone ms | TFLOP/s | V alone ms | TFLOP/s | concurrent ms | concurrent / sum | max / sum |\n|---|---|---|---|---|---|---|---|---|\n", name);
    for (int wm : {1, 2, 4, 6})
        for (int wv : {1, 2, 4, 6}) {
            if (wm + wv > 8) continue;
            const int gm = 256 * wm, gv = 256 * wv;
            // ~4 ms each: MFMA 32 cycles each on its SIMD; v_fma 4 cycles issue each
            const int itm = (BF >= 0 ? 40000 : 9000) / wm, itv = 36000 / wv;
            auto fa = [&] {
                if (BF >= 0) hipLaunchKernelGGL(mfma_bf16_kernel<(BF < 0 ? 0 : BF)>, dim3(gm), dim3(256), 0, s1, out, itm, 1.f, 1.0001f);
                else hipLaunchKernelGGL(mfma_kernel<LDS>, dim3(gm), dim3(256), 0, s1, out, itm, 1.f, 2.f);
            };
            auto fb = [&] { hipLaunchKernelGGL(valu_kernel, dim3(gv), dim3(256), 0, s2, out, itv, 1.f, 1.0001f); };
            timed(fa, fb, true, true);
            float ta = 1e9f, tb = 1e9f, tc = 1e9f;
            for (int r = 0; r < 3; ++r) {
                ta = fminf(ta, timed(fa, fb, true, false));
                tb = fminf(tb, timed(fa, fb, false, true));
                tc = fminf(tc, timed(fa, fb, true, true));
            }
            const double fm = (double)gm * 4 * itm * 32 * (BF >= 0 ? 16384.0 : 2048.0), fv = (double)gv * 256 * (double)itv * 64 * 2.0;
            printf("| %d | %d | %.3f | %.1f | %.3f | %.1f | %.3f | **%.3f** | %.3f |\n", wm, wv, ta, fm / ta * 1e-9, tb, fv / tb * 1e-9, tc,
                   tc / (ta + tb), fmaxf(ta, tb) / (ta + tb));
            fflush(stdout);
        }
}

template <bool BF, int GAP = 0, bool PRIO = false>
static void one_launch(const char *name) {
    printf("\n| %s, ONE launch of 256 x w workgroups of 512 threads: w | matrix waves only ms | vector waves only ms | both ms | both / sum | max / sum |\n|---|---|---|---|---|---|\n", name);
    for (int w : {1, 2, 4}) {
        const int itm = (BF ? 40000 : 9000) / w / (GAP >= 8 ? 2 : 1), itv = 36000 / w;
        float t[4] = {0, 1e9f, 1e9f, 1e9f};
        for (int r = 0; r < 3; ++r)
            for (int mode = 1; mode <= 3; ++mode) {
                (void)hipDeviceSynchronize();
                (void)hipEventRecord(e0, s1);
                hipLaunchKernelGGL((mixed_kernel<BF, GAP, PRIO>), dim3(256 * w), dim3(512), 0, s1, out, itm, itv, 1.f, 1.0001f, mode);
                (void)hipEventRecord(e1, s1);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                t[mode] = fminf(t[mode], ms);
            }
        printf("| %d | %.3f | %.3f | %.3f | **%.3f** | %.3f |\n", w, t[1], t[2], t[3], t[3] / (t[1] + t[2]), fmaxf(t[1], t[2]) / (t[1] + t[2]));
    }
}

int main(int argc, char **) {
    (void)hipMalloc(&out, 64);
    (void)hipStreamCreate(&s1), (void)hipStreamCreate(&s2);
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1), (void)hipEventCreate(&e2);
    printf("# pure matrix-pipe kernel x pure vector-ALU kernel on two streams (MI355X)\n");
    one_launch<false>("fp32 MFMA waves + v_fma waves");
    one_launch<true>("bf16 MFMA waves + v_fma waves");
    one_launch<true, 16>("bf16 MFMA waves, s_nop 15 (16 wait states) after every MFMA, + v_fma waves");
    one_launch<false, 16>("fp32 MFMA waves, s_nop 15 after every MFMA, + v_fma waves");
    one_launch<false, 1>("fp32 MFMA waves, s_nop 0 after every MFMA, + v_fma waves");
    one_launch<false, 2>("fp32 MFMA waves, s_nop 1 after every MFMA, + v_fma waves");
    one_launch<false, 4>("fp32 MFMA waves, s_nop 3 after every MFMA, + v_fma waves");
    one_launch<false, 6>("fp32 MFMA waves, s_nop 5 after every MFMA, + v_fma waves");
    one_launch<false, 8>("fp32 MFMA waves, s_nop 7 after every MFMA, + v_fma waves");
    one_launch<true, 0, true>("bf16 MFMA waves + v_fma waves at s_setprio 3");
    if (argc > 1) return 0;   // any argument: the one-launch tables only
    sweep<0>("fp32 MFMA only");
    sweep<2>("fp32 MFMA + 2 ds_read_b32 per MFMA");
    sweep<0, 0>("bf16 MFMA (16x16x32) only");
    sweep<0, 2>("bf16 MFMA + 2 vector instructions per MFMA");
    return 0;
}
