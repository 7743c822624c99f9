// POSITIVE reproducer of the round-4 corruption (csrc/build.py): v_pk_fma_f32 whose op_sel takes the HIGH dword of
// a 64-bit source for the LOW lane returns wrong results while another wave of the compute unit executes bf16 matrix instructions.
// Found by ISA bisection of the encoder's first-level gather kernel (scripts/debug/pk_isa_variants.py / pk_isa_run.py,
// profiles/r05_pk_opsel.md): of its 26 packed instructions only the four `v_pk_fma_f32 ... op_sel:[0,1,0]` matter -- replaced by
// scalar pairs the kernel is right in 1000 of 1000 launches next to the bf16x3 GEMM, with them wrong in ~900; wait states around
// the packed instructions change nothing; disjoint compute units never fail.
//
// 512-thread workgroups: waves 0-3 issue matrix instructions (or nothing), waves 4-7 evaluate r = x * a + b on register pairs
// in one packed form and as two v_fma_f32 on single registers with the halves the form selects, and count results that differ
// (both are single-rounded fused multiply-adds of the same operands: they must not).  The first mismatches are kept with the
// values every other half selection would have produced, which names the half the hardware took.
// hipcc --offload-arch=gfx950 -O3 scripts/micro/pk_opsel_vs_mfma.hip -o deeppointmap_amd/csrc/build/pk_opsel_vs_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x2 = __attribute__((ext_vector_type(2))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) short;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;

struct Sample {
    float x[2], a[2], b[2], got[2], want[2];
};
struct Result {
    unsigned long long bad_lo, bad_hi, sink;
    unsigned n_samples;
    Sample s[8];
};

// FORM: which halves of src1 (a) the two lanes read.  {lo lane, hi lane}: 0 = a.lo, 1 = a.hi
//   0 plain {0,1}   1 op_sel_hi:[1,0,1] {0,0}   2 op_sel:[0,1,0] {1,1}   3 op_sel:[0,1,0] op_sel_hi:[1,0,1] {1,0}
//   4 op_sel:[1,0,0] (src0 = x: both lanes read x.hi)   5 op_sel:[0,0,1] (src2 = b: both lanes read b.hi)
//   6 v_pk_mul_f32 op_sel:[0,1] (b unused)   7 v_pk_add_f32 op_sel:[0,1] (x + a, b unused)
template <int FORM>
__device__ __forceinline__ f32x2 packed(f32x2 x, f32x2 a, f32x2 b) {
    f32x2 r;
    if (FORM == 0) asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    if (FORM == 1) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    if (FORM == 2) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0]" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    if (FORM == 3) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    if (FORM == 4) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0]" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    if (FORM == 5) asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,1]" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    if (FORM == 6) asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(a));
    if (FORM == 7) asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1]" : "=v"(r) : "v"(x), "v"(a));
    return r;
}
__device__ __forceinline__ float sfma(float x, float a, float b) {
    float r;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float smul(float x, float a) {
    float r;
    asm volatile("v_mul_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a));
    return r;
}
__device__ __forceinline__ float sadd(float x, float a) {
    float r;
    asm volatile("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(a));
    return r;
}
template <int FORM>
__device__ __forceinline__ f32x2 scalar(f32x2 x, f32x2 a, f32x2 b) {
    if (FORM == 0) return f32x2{sfma(x[0], a[0], b[0]), sfma(x[1], a[1], b[1])};
    if (FORM == 1) return f32x2{sfma(x[0], a[0], b[0]), sfma(x[1], a[0], b[1])};
    if (FORM == 2) return f32x2{sfma(x[0], a[1], b[0]), sfma(x[1], a[1], b[1])};
    if (FORM == 3) return f32x2{sfma(x[0], a[1], b[0]), sfma(x[1], a[0], b[1])};
    if (FORM == 4) return f32x2{sfma(x[1], a[0], b[0]), sfma(x[1], a[1], b[1])};
    if (FORM == 5) return f32x2{sfma(x[0], a[0], b[1]), sfma(x[1], a[1], b[1])};
    if (FORM == 6) return f32x2{smul(x[0], a[1]), smul(x[1], a[1])};
    return f32x2{sadd(x[0], a[1]), sadd(x[1], a[1])};
}

// MODE: 0 no matrix waves, 1 v_mfma_f32_16x16x32_bf16, 2 v_mfma_f32_16x16x4_f32, 3 v_mfma_f32_32x32x16_bf16, 4 v_mfma_f32_16x16x32_f16
template <int MODE, int FORM>
__global__ __launch_bounds__(512) void pk_kernel(Result *res, int iters, float a0, float b0) {
    const int role = threadIdx.x >> 8;
    if (role == 0) {
        if (MODE == 0) return;
        using f32x16 = __attribute__((ext_vector_type(16))) float;
        f32x4 acc[8];
        f32x16 big[2];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 16; ++j) big[i][j] = 0.f;
        bf16x8 av, bv;
        f16x8 ah, bh;
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = (short)(0x3f80 + threadIdx.x + i), bv[i] = (short)(0x3f80 + i), ah[i] = (_Float16)(1.0f + i), bh[i] = (_Float16)(0.5f);
        const float fa = a0 + threadIdx.x;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
                if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, b0, acc[i], 0, 0, 0);
                if (MODE == 3) big[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, big[i & 1], 0, 0, 0);
                if (MODE == 4) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[i], 0, 0, 0);
            }
        }
        float s = big[0][0] + big[1][5];
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) res->sink = 1;
        return;
    }
    // vector waves: 4 independent pairs per iteration, inputs advanced by scalar instructions only
    f32x2 x[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) x[i] = f32x2{0.5f + 0.001f * (threadIdx.x + i), 0.25f + 0.002f * (threadIdx.x + 3 * i)};
    const f32x2 a = f32x2{a0, a0 * 0.75f}, b = f32x2{b0, b0 * 1.5f};
    unsigned long long bad_lo = 0, bad_hi = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 got = packed<FORM>(x[i], a, b);
            const f32x2 want = scalar<FORM>(x[i], a, b);
            const bool wl = __float_as_uint(got[0]) != __float_as_uint(want[0]), wh = __float_as_uint(got[1]) != __float_as_uint(want[1]);
            bad_lo += wl, bad_hi += wh;
            if ((wl || wh) && (bad_lo + bad_hi) <= 2) {
                const unsigned slot = atomicAdd(&res->n_samples, 1u);
                if (slot < 8) {
                    Sample &s = res->s[slot];
                    s.x[0] = x[i][0], s.x[1] = x[i][1], s.a[0] = a[0], s.a[1] = a[1], s.b[0] = b[0], s.b[1] = b[1];
                    s.got[0] = got[0], s.got[1] = got[1], s.want[0] = want[0], s.want[1] = want[1];
                }
            }
            x[i][0] = sfma(x[i][0], 0.9999f, 0.00013f), x[i][1] = sfma(x[i][1], 0.9998f, 0.00021f);
        }
    }
    if (bad_lo) atomicAdd(&res->bad_lo, bad_lo);
    if (bad_hi) atomicAdd(&res->bad_hi, bad_hi);
}

// ---- the gather kernel's situation: src1 of the packed instruction is a pair that a global load has just delivered (a
// point's x | y), several vector waves per SIMD keep loads in flight, the matrix waves run next to them.  WAVES = vector waves
// per SIMD (the workgroup is 256 * (1 + WAVES) threads: waves 0-3 issue matrix instructions).
template <int MODE, int FORM, int WAVES>
__global__ __launch_bounds__(256 * (1 + WAVES)) void pk_gather_kernel(Result *res, const float *__restrict__ pts, unsigned mask, int iters, float a0, float b0) {
    if (threadIdx.x < 256) {
        if (MODE == 0) return;
        f32x4 acc[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        bf16x8 av, bv;
#pragma unroll
        for (int i = 0; i < 8; ++i) av[i] = (short)(0x3f80 + threadIdx.x + i), bv[i] = (short)(0x3f80 + i);
        const float fa = a0 + threadIdx.x;
        for (int it = 0; it < iters * 2; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (MODE == 1) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(av, bv, acc[i], 0, 0, 0);
                if (MODE == 2) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, b0, acc[i], 0, 0, 0);
            }
        }
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
        if (s == 12345.678f) res->sink = 1;
        return;
    }
    const f32x2 w[4] = {f32x2{a0, a0 * 0.75f}, f32x2{a0 * 1.25f, a0 * 0.5f}, f32x2{a0 * 0.3f, a0 * 1.7f}, f32x2{a0 * 0.9f, a0 * 1.1f}};
    const f32x2 c = f32x2{b0, b0 * 1.5f};
    unsigned long long bad_lo = 0, bad_hi = 0;
    unsigned n = (blockIdx.x * 1024u + threadIdx.x) * 2654435761u;
    for (int it = 0; it < iters; ++it) {
        f32x2 p[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {   // four gathers in flight, like the kernel's row passes
            n = n * 1664525u + 1013904223u;
            const float *q = pts + 3u * ((n >> 8) & mask);
            p[i] = f32x2{q[0], q[1]};
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const f32x2 got = packed<FORM>(w[i], p[i], c);
            const f32x2 want = scalar<FORM>(w[i], p[i], c);
            const bool wl = __float_as_uint(got[0]) != __float_as_uint(want[0]), wh = __float_as_uint(got[1]) != __float_as_uint(want[1]);
            bad_lo += wl, bad_hi += wh;
            if ((wl || wh) && (bad_lo + bad_hi) <= 2) {
                const unsigned slot = atomicAdd(&res->n_samples, 1u);
                if (slot < 8) {
                    Sample &s = res->s[slot];
                    s.x[0] = w[i][0], s.x[1] = w[i][1], s.a[0] = p[i][0], s.a[1] = p[i][1], s.b[0] = c[0], s.b[1] = c[1];
                    s.got[0] = got[0], s.got[1] = got[1], s.want[0] = want[0], s.want[1] = want[1];
                }
            }
        }
    }
    if (bad_lo) atomicAdd(&res->bad_lo, bad_lo);
    if (bad_hi) atomicAdd(&res->bad_hi, bad_hi);
}

void report(const Result &h);
template <int MODE, int FORM, int WAVES>
void run_gather(Result *d, const float *pts, unsigned mask, int iters, const char *mname, const char *fname) {
    Result h;
    (void)hipMemset(d, 0, sizeof(Result));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((pk_gather_kernel<MODE, FORM, WAVES>), dim3(1024), dim3(256 * (1 + WAVES)), 0, 0, d, pts, mask, iters, 0.9999f, 0.0001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, d, sizeof(Result), hipMemcpyDeviceToHost);
    printf("| gather-like, %d vector waves per SIMD; %s | %s | %llu | %llu | %.3g | %.1f |", WAVES, mname, fname, h.bad_lo, h.bad_hi, 1024.0 * 256 * WAVES * 4 * iters, ms);
    report(h);
}

template <int MODE, int FORM>
void run(Result *d, int iters, const char *mname, const char *fname) {
    Result h;
    (void)hipMemset(d, 0, sizeof(Result));
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((pk_kernel<MODE, FORM>), dim3(1024), dim3(512), 0, 0, d, iters, 0.9999f, 0.0001f);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemcpy(&h, d, sizeof(Result), hipMemcpyDeviceToHost);
    printf("| %s | %s | %llu | %llu | %.3g | %.1f |", mname, fname, h.bad_lo, h.bad_hi, 1024.0 * 256 * 4 * iters, ms);
    report(h);
}

void report(const Result &h) {
    for (unsigned i = 0; i < h.n_samples && i < 2; ++i) {
        const Sample &s = h.s[i];
        // which selection reproduces the wrong value?
        const char *why_lo = "", *why_hi = "";
        auto same = [](float p, float q) { return memcmp(&p, &q, 4) == 0; };
        if (!same(s.got[0], s.want[0])) {
            why_lo = "lo: ?";
            for (int xi = 0; xi < 2; ++xi) for (int ai = 0; ai < 2; ++ai) for (int bi = 0; bi < 2; ++bi)
                if (same(s.got[0], fmaf(s.x[xi], s.a[ai], s.b[bi]))) { static char buf[64]; snprintf(buf, 64, "lo lane = fma(x.%s, a.%s, b.%s)", xi ? "hi" : "lo", ai ? "hi" : "lo", bi ? "hi" : "lo"); why_lo = buf; }
        }
        if (!same(s.got[1], s.want[1])) {
            why_hi = "hi: ?";
            for (int xi = 0; xi < 2; ++xi) for (int ai = 0; ai < 2; ++ai) for (int bi = 0; bi < 2; ++bi)
                if (same(s.got[1], fmaf(s.x[xi], s.a[ai], s.b[bi]))) { static char buf[64]; snprintf(buf, 64, "hi lane = fma(x.%s, a.%s, b.%s)", xi ? "hi" : "lo", ai ? "hi" : "lo", bi ? "hi" : "lo"); why_hi = buf; }
        }
        printf(" got (%.9g, %.9g) want (%.9g, %.9g) %s %s;", s.got[0], s.got[1], s.want[0], s.want[1], why_lo, why_hi);
    }
    printf("\n");
    fflush(stdout);
}

template <int MODE>
void forms(Result *d, int iters, const char *mname) {
    run<MODE, 0>(d, iters, mname, "v_pk_fma_f32 (plain)");
    run<MODE, 1>(d, iters, mname, "v_pk_fma_f32 op_sel_hi:[1,0,1]");
    run<MODE, 2>(d, iters, mname, "v_pk_fma_f32 op_sel:[0,1,0]");
    run<MODE, 3>(d, iters, mname, "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]");
    run<MODE, 4>(d, iters, mname, "v_pk_fma_f32 op_sel:[1,0,0]");
    run<MODE, 5>(d, iters, mname, "v_pk_fma_f32 op_sel:[0,0,1]");
    run<MODE, 6>(d, iters, mname, "v_pk_mul_f32 op_sel:[0,1]");
    run<MODE, 7>(d, iters, mname, "v_pk_add_f32 op_sel:[0,1]");
}

int main(int argc, char **argv) {
    Result *d;
    (void)hipMalloc(&d, sizeof(Result));
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    printf("| matrix waves (one per SIMD, next to one vector wave per SIMD) | packed form | wrong low lanes | wrong high lanes | of (per lane half) | ms | first mismatches |\n|---|---|---|---|---|---|---|\n");
    forms<0>(d, iters, "none");
    forms<1>(d, iters, "v_mfma_f32_16x16x32_bf16");
    forms<2>(d, iters, "v_mfma_f32_16x16x4_f32");
    forms<3>(d, iters, "v_mfma_f32_32x32x16_bf16");
    forms<4>(d, iters, "v_mfma_f32_16x16x32_f16");
    // gather-like
    const unsigned npts = 1u << 16;
    float *pts, *hp = (float *)malloc(npts * 12);
    for (unsigned i = 0; i < npts * 3; ++i) hp[i] = 0.25f + (float)((i * 2654435761u) >> 8) * (1.0f / 16777216.0f);
    (void)hipMalloc(&pts, npts * 12);
    (void)hipMemcpy(pts, hp, npts * 12, hipMemcpyHostToDevice);
    const int gi = iters / 8;
    run_gather<0, 2, 3>(d, pts, npts - 1, gi, "none", "v_pk_fma_f32 op_sel:[0,1,0]");
    run_gather<1, 0, 3>(d, pts, npts - 1, gi, "v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32 (plain)");
    run_gather<1, 1, 3>(d, pts, npts - 1, gi, "v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32 op_sel_hi:[1,0,1]");
    run_gather<1, 2, 3>(d, pts, npts - 1, gi, "v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32 op_sel:[0,1,0]");
    run_gather<1, 2, 1>(d, pts, npts - 1, gi, "v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32 op_sel:[0,1,0]");
    run_gather<1, 3, 3>(d, pts, npts - 1, gi, "v_mfma_f32_16x16x32_bf16", "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]");
    run_gather<1, 6, 3>(d, pts, npts - 1, gi, "v_mfma_f32_16x16x32_bf16", "v_pk_mul_f32 op_sel:[0,1]");
    run_gather<2, 2, 3>(d, pts, npts - 1, gi, "v_mfma_f32_16x16x4_f32", "v_pk_fma_f32 op_sel:[0,1,0]");
    return 0;
}
