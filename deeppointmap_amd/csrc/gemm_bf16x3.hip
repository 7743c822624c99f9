// fp32 GEMM through the bf16 matrix pipe: out = act(X W^T + bias + residual) with every fp32 operand split EXACTLY into
// three bf16 terms (x = hi + mid + lo: the top, middle and bottom 8 bits of its 24-bit significand, each a truncation,
// so the three terms reproduce x bit for bit) and the product formed from the six term pairs whose weight is above
// 2^-24 of the full product:
//     x w = hi_x hi_w + hi_x mid_w + mid_x hi_w + hi_x lo_w + lo_x hi_w + mid_x mid_w   (+ terms below 2^-24 |x w|).
// Each pair product of two bf16 values is exact in fp32 and v_mfma_f32_32x32x16_bf16 accumulates in fp32, so the result
// carries the rounding of an fp32 accumulation (measured against fp64 in tests/test_gpu_ops.py: the same error level
// as the fp32-MFMA kernel of gemm.hip, whose v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate).  Six bf16 MFMAs per
// fp32 product = 3/8 of the fp32-MFMA time.  Same "NT" layout as gemm.hip; weights arrive pre-split (they are constant:
// dpm_split_bf16x3 once per weight), activations are split while they are staged into LDS.
//   block tile 128 x 128, 4 waves as 2 x 2, wave tile 64 x 64 = 2 x 2 blocks of 32x32x16; K-tile 32; LDS holds three
//   bf16 planes per operand, rows 80 B apart (16-byte aligned ds_read_b128 fragments, 8 bf16 per lane).
#include "dpm_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;  // first-class vector values: the HIP uint4 struct array ended up in scratch

__device__ __forceinline__ float act_apply(float v, int act) {
    if (act == DPM_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DPM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// x -> (hi, mid, lo) as the upper 16 bits of three floats; the split is exact (each term a truncation)
__device__ __forceinline__ void split3(float x, unsigned &hi, unsigned &mid, unsigned &lo) {
    const unsigned xb = __float_as_uint(x);
    hi = xb & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mid);
    lo = __float_as_uint(r2) & 0xFFFF0000u;
}
// two such terms -> one dword holding [a | b] as consecutive bf16 (a at the lower address)
__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return (a >> 16) | b; }

__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ W, long long n, uint16_t *__restrict__ planes) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h, m, l;
    split3(W[i], h, m, l);
    planes[i] = (uint16_t)(h >> 16), planes[n + i] = (uint16_t)(m >> 16), planes[2 * n + i] = (uint16_t)(l >> 16);
}

constexpr int BM = 128, BN = 128, KT = 32, LD = KT + 8;  // LDS row stride in bf16 (80 bytes)

__global__ __launch_bounds__(256, 2) void gemm_bf16x3_kernel(const float *__restrict__ X, int ldx, const uint16_t *__restrict__ Wp,
                                                             int ldw, long long plane, const float *__restrict__ bias,
                                                             const float *__restrict__ res, int ldr, float *__restrict__ out,
                                                             int ldo, int R, int Cin, int Cout, int act) {
    __shared__ __attribute__((aligned(16))) uint16_t Xs[3][BM][LD];
    __shared__ __attribute__((aligned(16))) uint16_t Ws[3][BN][LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    int by = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.y & 7) == 0 && gridDim.y >= 64) {  // all column blocks of one row block on ONE XCD: its L2 serves X
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, slot = L >> 3;
        by = (int)((slot / gridDim.x) * 8 + xcd), bx = (int)(slot % gridDim.x);
    }
    const int row0 = by * BM, col0 = bx * BN;
    // staging maps: X as fp32 float4 (4 per thread: rows xr_ + 32 p), W planes as 8 bf16 = uint4 (6 per thread: plane
    // p >> 1, rows wr_ + 64 (p & 1)).  Every index below is a compile-time constant after unrolling: arrays indexed
    // through a lambda or a runtime value end up in scratch memory (the first version of this kernel did: 112 B / lane).
    const int xr_ = t >> 3, xk = (t & 7) * 4, wr_ = t >> 2, wk = (t & 3) * 8;
    const float *xp[4];
    const uint16_t *wp[2];
#pragma unroll
    for (int p = 0; p < 4; ++p) xp[p] = X + (size_t)min(row0 + p * 32 + xr_, R - 1) * ldx + xk;
#pragma unroll
    for (int p = 0; p < 2; ++p) wp[p] = Wp + (size_t)min(col0 + p * 64 + wr_, Cout - 1) * ldw + wk;
    f32x4 xr[4];
    u32x4 wr[6];
#pragma unroll
    for (int p = 0; p < 4; ++p) xr[p] = *reinterpret_cast<const f32x4 *>(xp[p]);
#pragma unroll
    for (int p = 0; p < 6; ++p) wr[p] = *reinterpret_cast<const u32x4 *>(wp[p & 1] + (size_t)(p >> 1) * plane);
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int fr = lane & 31, fk = (lane >> 5) * 8;
    for (int k0 = 0; k0 < Cin; k0 += KT) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
            split3(xr[p][0], h0, m0, l0), split3(xr[p][1], h1, m1, l1), split3(xr[p][2], h2, m2, l2), split3(xr[p][3], h3, m3, l3);
            const int r = p * 32 + xr_;
            *reinterpret_cast<uint2 *>(&Xs[0][r][xk]) = make_uint2(pack2(h0, h1), pack2(h2, h3));
            *reinterpret_cast<uint2 *>(&Xs[1][r][xk]) = make_uint2(pack2(m0, m1), pack2(m2, m3));
            *reinterpret_cast<uint2 *>(&Xs[2][r][xk]) = make_uint2(pack2(l0, l1), pack2(l2, l3));
        }
#pragma unroll
        for (int p = 0; p < 6; ++p) *reinterpret_cast<u32x4 *>(&Ws[p >> 1][(p & 1) * 64 + wr_][wk]) = wr[p];
        __syncthreads();
        {   // the next K-tile is requested while this one feeds the MFMAs -- unconditionally (the last trip re-reads its
            // own tile): behind a branch the compiler keeps the prefetch registers in scratch memory
            const int kn = min(k0 + KT, Cin - KT);
#pragma unroll
            for (int p = 0; p < 4; ++p) xr[p] = *reinterpret_cast<const f32x4 *>(xp[p] + kn);
#pragma unroll
            for (int p = 0; p < 6; ++p) wr[p] = *reinterpret_cast<const u32x4 *>(wp[p & 1] + (size_t)(p >> 1) * plane + kn);
        }
#pragma unroll
        for (int ks = 0; ks < KT; ks += 16) {
            bf16x8 a[3][2], b[3][2];  // a: W fragments (the instruction's A operand), b: X fragments
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int j = 0; j < 2; ++j) a[pl][j] = *reinterpret_cast<const bf16x8 *>(&Ws[pl][wn * 64 + j * 32 + fr][ks + fk]);
#pragma unroll
                for (int i = 0; i < 2; ++i) b[pl][i] = *reinterpret_cast<const bf16x8 *>(&Xs[pl][wm * 64 + i * 32 + fr][ks + fk]);
            }
            // smallest terms first; (plane of W, plane of X): (1,1) (2,0) (0,2) (1,0) (0,1) (0,0)
#define DPM_B3(PWQ, PXQ)                                                                                       \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)              \
        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[PWQ][j], b[PXQ][i], acc[i][j], 0, 0, 0)
            DPM_B3(1, 1);
            DPM_B3(2, 0);
            DPM_B3(0, 2);
            DPM_B3(1, 0);
            DPM_B3(0, 1);
            DPM_B3(0, 0);
#undef DPM_B3
        }
        __syncthreads();
    }
    // D[m][n]: n = lane & 31 -> output row, m = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) -> output column
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int r = row0 + wm * 64 + i * 32 + (lane & 31);
            if (r >= R) continue;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c = col0 + wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                if (c >= Cout) continue;  // Cout % 4 == 0: all four columns exist
                float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                if (bias) {
                    const float4 bv = *reinterpret_cast<const float4 *>(bias + c);
                    v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                }
                if (res) {
                    const float4 rv = *reinterpret_cast<const float4 *>(res + (size_t)r * ldr + c);
                    v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
                }
                v.x = act_apply(v.x, act), v.y = act_apply(v.y, act), v.z = act_apply(v.z, act), v.w = act_apply(v.w, act);
                *reinterpret_cast<float4 *>(out + (size_t)r * ldo + c) = v;
            }
        }
}

}  // namespace

extern "C" int dpm_split_bf16x3(const float *W, long long n, uint16_t *planes, dpm_stream_t stream) {
    DPM_CHECK_ARG(W && planes && n >= 1);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, W, n, planes);
    return dpm_launch_status();
}

extern "C" int dpm_linear_bf16x3(const float *x, int ldx, const uint16_t *w_planes, int ldw, long long plane_stride,
                                 const float *bias, const float *residual, int ldr, float *out, int ldo, int R, int Cin,
                                 int Cout, int act, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && w_planes && out && R >= 1 && Cin >= 1 && Cout >= 1 && ldx >= Cin && ldw >= Cin && ldo >= Cout);
    DPM_CHECK_ARG(act >= DPM_ACT_NONE && act <= DPM_ACT_SIGMOID && (!residual || ldr >= Cout));
    auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    if (!(Cin % KT == 0 && Cout % 4 == 0 && ldx % 4 == 0 && ldw % 8 == 0 && plane_stride % 8 == 0 && ldo % 4 == 0 && al(x) &&
          al(w_planes) && al(out) && al(bias) && (!residual || (al(residual) && ldr % 4 == 0))))
        return DPM_EUNSUPPORTED;
    hipLaunchKernelGGL(gemm_bf16x3_kernel, dim3(dpm_cdiv(Cout, BN), dpm_cdiv(R, BM)), dim3(256), 0, (hipStream_t)stream, x, ldx,
                       w_planes, ldw, plane_stride, bias, residual, ldr, out, ldo, R, Cin, Cout, act);
    return dpm_launch_status();
}
