// fp32 GEMM on the matrix cores: out = act(X W^T + bias + residual), optionally batched.
// Replaces every Conv1d(k=1) / nn.Linear of the path (reference network/encoder/utils.py:358-389,
// decoder heads, nn.MultiheadAttention projections) and the descriptor-vs-descriptor similarity
// contraction (decoder.py:185).
//
// v_mfma_f32_16x16x4_f32 is exact fp32 (a k-ordered fmaf chain, MI355X guide section 3), so the
// numerics equal a scalar fp32 loop.  "NT" layout: both operands have k contiguous (X rows and W
// rows), which is the reference's native weight layout -- no transposes anywhere.
//   block tile BM x BN, 4 waves as 2x2, wave tile (BM/2)x(BN/2) = MBxNB MFMA blocks of 16x16;
//   K-tile 32 staged through LDS with row stride 34 floats (conflict-free ds_read_b32 for the
//   A[i=l&15][k=l>>4] / B[k=l>>4][j=l&15] fragment pattern); the next K-tile is prefetched into
//   registers while the current one feeds the MFMAs.
#include "dpm_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DPM_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DPM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// 4 consecutive k of one row, zeros for k >= K.  Branch-free on purpose: rows beyond `rows` read the last valid
// row (their products land in output rows / columns the epilogue never stores) and the k tail is a clamped load
// plus a select, so the compiler issues the whole prefetch group back to back and waits for it only where the
// values are written to LDS one K-tile later.  (With guarded loads it serialised the group behind s_waitcnt
// vmcnt(0) and the HBM latency was exposed twice per K-tile.)
template <bool VEC>
__device__ __forceinline__ float4 load4(const float *__restrict__ base, int ld, int row, int rows, int k, int K) {
    const float *p = base + (size_t)min(row, rows - 1) * ld;
    if (VEC) {  // ld % 4 == 0, base 16-byte aligned, K % 4 == 0
        const float4 v = *reinterpret_cast<const float4 *>(p + min(k, K - 4));
        return k < K ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 v;
    v.x = p[min(k, K - 1)], v.y = p[min(k + 1, K - 1)], v.z = p[min(k + 2, K - 1)], v.w = p[min(k + 3, K - 1)];
    v.x = k < K ? v.x : 0.f, v.y = k + 1 < K ? v.y : 0.f, v.z = k + 2 < K ? v.z : 0.f, v.w = k + 3 < K ? v.w : 0.f;
    return v;
}

template <int BM, int BN, bool VEC, int KT = 32>
__global__ __launch_bounds__(256) void gemm_nt_mfma_kernel(const float *__restrict__ X, int ldx, long long sx,
                                                           const float *__restrict__ W, int ldw, long long sw,
                                                           const float *__restrict__ bias,
                                                           const float *__restrict__ res, int ldr, long long sr,
                                                           float *__restrict__ out, int ldo, long long so, int R,
                                                           int Cin, int Cout, int act) {
    constexpr int LDS_LD = KT + 2, LPR = KT / 4, RPP = 256 / LPR;  // lanes per staged row, rows per staging pass
    constexpr int WM = BM / 2, WN = BN / 2, MB = WM / 16, NB = WN / 16, PX = BM / RPP, PW = BN / RPP;
    __shared__ float Xs[BM][LDS_LD];
    __shared__ float Ws[BN][LDS_LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    const int bz = blockIdx.z;
    X += (size_t)bz * sx, W += (size_t)bz * sw, out += (size_t)bz * so;
    if (res) res += (size_t)bz * sr;
    int by = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.y & 7) == 0 && gridDim.y >= 64) {  // all column blocks of one row block on ONE XCD: its L2 serves X
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, slot = L >> 3;
        by = (int)((slot / gridDim.x) * 8 + xcd), bx = (int)(slot % gridDim.x);
    }
    const int row0 = by * BM, col0 = bx * BN;
    const int sr_ = t / LPR, sk = (t % LPR) * 4;  // staging: row within a pass, k offset

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 xr[PX], wr[PW];
#pragma unroll
    for (int p = 0; p < PX; ++p) xr[p] = load4<VEC>(X, ldx, row0 + p * RPP + sr_, R, sk, Cin);
#pragma unroll
    for (int p = 0; p < PW; ++p) wr[p] = load4<VEC>(W, ldw, col0 + p * RPP + sr_, Cout, sk, Cin);

    for (int k0 = 0; k0 < Cin; k0 += KT) {
        // registers -> LDS (row stride 136 B: 8-byte aligned, so two 8-byte stores per float4)
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            float2 *d = reinterpret_cast<float2 *>(&Xs[p * RPP + sr_][sk]);
            d[0] = make_float2(xr[p].x, xr[p].y), d[1] = make_float2(xr[p].z, xr[p].w);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            float2 *d = reinterpret_cast<float2 *>(&Ws[p * RPP + sr_][sk]);
            d[0] = make_float2(wr[p].x, wr[p].y), d[1] = make_float2(wr[p].z, wr[p].w);
        }
        __syncthreads();
        if (k0 + KT < Cin) {  // prefetch the next K-tile while this one is consumed
#pragma unroll
            for (int p = 0; p < PX; ++p) xr[p] = load4<VEC>(X, ldx, row0 + p * RPP + sr_, R, k0 + KT + sk, Cin);
#pragma unroll
            for (int p = 0; p < PW; ++p) wr[p] = load4<VEC>(W, ldw, col0 + p * RPP + sr_, Cout, k0 + KT + sk, Cin);
        }
        float a[2][MB], b[2][NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) a[0][i] = Xs[wm * WM + i * 16 + (lane & 15)][lane >> 4];
#pragma unroll
        for (int j = 0; j < NB; ++j) b[0][j] = Ws[wn * WN + j * 16 + (lane & 15)][lane >> 4];
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
            if (kk + 4 < KT) {  // fragments of the next k-step are in flight while this one's MFMAs issue
#pragma unroll
                for (int i = 0; i < MB; ++i) a[nxt][i] = Xs[wm * WM + i * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < NB; ++j) b[nxt][j] = Ws[wn * WN + j * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
            }
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[cur][i], b[cur][j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
    // C/D layout of the 16x16 block: col = lane & 15, row = (lane >> 4) * 4 + reg
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int c = col0 + wn * WN + j * 16 + (lane & 15);
            if (c >= Cout) continue;
            const float bv = bias ? bias[c] : 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int r = row0 + wm * WM + i * 16 + (lane >> 4) * 4 + q;
                if (r >= R) continue;
                float v = acc[i][j][q] + bv;
                if (res) v += res[(size_t)r * ldr + c];
                out[(size_t)r * ldo + c] = apply_act(v, act);
            }
        }
}

}  // namespace

extern "C" int dpm_linear_batched(const float *x, int ldx, long long sx, const float *W, int ldw, long long sw,
                                  const float *bias, const float *residual, int ldr, long long sr, float *out, int ldo,
                                  long long so, int batch, int R, int Cin, int Cout, int act, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && W && out && batch >= 1 && R >= 1 && Cin >= 1 && Cout >= 1);
    DPM_CHECK_ARG(ldx >= Cin && ldw >= Cin && ldo >= Cout && (!residual || ldr >= Cout));
    DPM_CHECK_ARG(act >= DPM_ACT_NONE && act <= DPM_ACT_SIGMOID);
    hipStream_t st = (hipStream_t)stream;
    const bool vec = ldx % 4 == 0 && ldw % 4 == 0 && Cin % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)W & 15) == 0 &&
                     sx % 4 == 0 && sw % 4 == 0;
    const long long big = (long long)batch * dpm_cdiv(R, 64) * dpm_cdiv(Cout, 64);
    // 64x64 tiles measured best or tied against 128x128 / 128x64 on every shape of the path (scripts/gemm_bench.py)
    const bool t64 = big >= 192 || (R > 1024 && Cout > 32);
    const dim3 grid = t64 ? dim3(dpm_cdiv(Cout, 64), dpm_cdiv(R, 64), batch) : dim3(dpm_cdiv(Cout, 32), dpm_cdiv(R, 32), batch);
#define DPM_GEMM_LAUNCH(BM, BN, V)                                                                                  \
    hipLaunchKernelGGL((gemm_nt_mfma_kernel<BM, BN, V>), grid, dim3(256), 0, st, x, ldx, sx, W, ldw, sw, bias, residual, \
                       ldr, sr, out, ldo, so, R, Cin, Cout, act)
    if (t64) {
        if (vec) DPM_GEMM_LAUNCH(64, 64, true);
        else DPM_GEMM_LAUNCH(64, 64, false);
    } else {
        if (vec) DPM_GEMM_LAUNCH(32, 32, true);
        else DPM_GEMM_LAUNCH(32, 32, false);
    }
#undef DPM_GEMM_LAUNCH
    return dpm_launch_status();
}

extern "C" int dpm_linear(const float *x, int ldx, const float *W, int ldw, const float *bias, const float *residual,
                          int ldr, float *out, int ldo, int R, int Cin, int Cout, int act, dpm_stream_t stream) {
    return dpm_linear_batched(x, ldx, 0, W, ldw, 0, bias, residual, ldr, 0, out, ldo, 0, 1, R, Cin, Cout, act, stream);
}
