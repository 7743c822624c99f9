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

#include <algorithm>
#include <type_traits>

#ifndef DPM_GEMM_WS_DEFAULT
#define DPM_GEMM_WS_DEFAULT 0   // 0: the 64 x 64 kernel everywhere (shipped: the wave-specialised kernel only ties it at K = 256, profiles/r04_corun.md); 1: the wave-specialised kernel takes the shapes it covers
#endif

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
// KFULL (K a multiple of the K-tile: every shape of the path): no k test at all -- with the select the compiler
// sinks each load into an exec-masked branch of its own, one basic block per load.
template <bool VEC, bool KFULL = false>
__device__ __forceinline__ float4 load4(const float *__restrict__ base, int ld, int row, int rows, int k, int K) {
    const float *p = base + (size_t)min(row, rows - 1) * ld;
    if (VEC && KFULL) return *reinterpret_cast<const float4 *>(p + k);
    if (VEC) {  // ld % 4 == 0, base 16-byte aligned, K % 4 == 0
        const float4 v = *reinterpret_cast<const float4 *>(p + min(k, K - 4));
        return k < K ? v : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 v;
    v.x = p[min(k, K - 1)], v.y = p[min(k + 1, K - 1)], v.z = p[min(k + 2, K - 1)], v.w = p[min(k + 3, K - 1)];
    v.x = k < K ? v.x : 0.f, v.y = k + 1 < K ? v.y : 0.f, v.z = k + 2 < K ? v.z : 0.f, v.w = k + 3 < K ? v.w : 0.f;
    return v;
}

template <int BM, int BN, bool VEC, int KT = 32, bool KFULL = false>
__global__ __launch_bounds__(256) void gemm_nt_mfma_kernel(const float *__restrict__ X, int ldx, long long sx,
                                                           const float *__restrict__ W, int ldw, long long sw,
                                                           const float *__restrict__ bias,
                                                           const float *__restrict__ res, int ldr, long long sr,
                                                           float *__restrict__ out, int ldo, long long so, int R,
                                                           int Cin, int Cout, int act) {
    constexpr int LDS_LD = KT + 2, LPR = KT / 4, RPP = 256 / LPR;  // lanes per staged row, rows per staging pass
    constexpr int WM = BM / 2, WN = BN / 2, MB = WM / 16, NB = WN / 16, PX = BM / RPP, PW = BN / RPP;
    constexpr int LDC = BN + 4;  // row stride of the output tile when it is staged for the epilogue
    static_assert(BM * LDC <= (BM + BN) * LDS_LD, "the staged output tile reuses the operand tiles' LDS");
    __shared__ __attribute__((aligned(16))) float smem[(BM + BN) * LDS_LD];
    float (*Xs)[LDS_LD] = reinterpret_cast<float (*)[LDS_LD]>(smem);
    float (*Ws)[LDS_LD] = reinterpret_cast<float (*)[LDS_LD]>(smem + BM * LDS_LD);
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
    for (int p = 0; p < PX; ++p) xr[p] = load4<VEC, KFULL>(X, ldx, row0 + p * RPP + sr_, R, sk, Cin);
#pragma unroll
    for (int p = 0; p < PW; ++p) wr[p] = load4<VEC, KFULL>(W, ldw, col0 + p * RPP + sr_, Cout, sk, Cin);

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
            for (int p = 0; p < PX; ++p) xr[p] = load4<VEC, KFULL>(X, ldx, row0 + p * RPP + sr_, R, k0 + KT + sk, Cin);
#pragma unroll
            for (int p = 0; p < PW; ++p) wr[p] = load4<VEC, KFULL>(W, ldw, col0 + p * RPP + sr_, Cout, k0 + KT + sk, Cin);
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[cur][j], a[cur][i], acc[i][j], 0, 0, 0), mfma_pace();
        }
        __syncthreads();
    }
    // W is the instruction's A operand and X its B operand, so the 16x16 result block is the TRANSPOSED output
    // block: D[m][n] with n = lane & 15 -> output row, m = (lane >> 4) * 4 + reg -> output column: a lane owns
    // four consecutive columns of one row.
    const bool vec_out = VEC && (ldo & 3) == 0 && (((uintptr_t)out) & 15) == 0 && (Cout & 3) == 0 &&
                         (!bias || (((uintptr_t)bias) & 15) == 0) &&
                         (!res || ((ldr & 3) == 0 && (((uintptr_t)res) & 15) == 0));
    if (vec_out) {
        // The tile goes through LDS once more (the operand tiles are dead after the loop's last barrier) so that
        // every store instruction writes whole rows of the tile: 256-byte runs instead of 64-byte pieces.
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                *reinterpret_cast<float4 *>(&smem[(wm * WM + i * 16 + (lane & 15)) * LDC + wn * WN + j * 16 + (lane >> 4) * 4]) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        __syncthreads();
        constexpr int TPR = BN / 4, RPS = 256 / TPR;  // threads per tile row, rows per store pass
        const int cr = t / TPR, cc = (t % TPR) * 4, c = col0 + cc;
        if (c < Cout) {
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) bv = *reinterpret_cast<const float4 *>(bias + c);
#pragma unroll
            for (int p = 0; p < BM / RPS; ++p) {
                const int r = row0 + p * RPS + cr;
                if (r >= R) continue;
                float4 v = *reinterpret_cast<const float4 *>(&smem[(p * RPS + cr) * LDC + cc]);
                v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                if (res) {
                    const float4 rv = *reinterpret_cast<const float4 *>(res + (size_t)r * ldr + c);
                    v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
                }
                v.x = apply_act(v.x, act), v.y = apply_act(v.y, act), v.z = apply_act(v.z, act), v.w = apply_act(v.w, act);
                *reinterpret_cast<float4 *>(out + (size_t)r * ldo + c) = v;
            }
        }
        return;
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int r = row0 + wm * WM + i * 16 + (lane & 15);
            const int c = col0 + wn * WN + j * 16 + (lane >> 4) * 4;
            if (r >= R) continue;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (c + q >= Cout) continue;
                float v = acc[i][j][q] + (bias ? bias[c + q] : 0.f);
                if (res) v += res[(size_t)r * ldr + c + q];
                out[(size_t)r * ldo + c + q] = apply_act(v, act);
            }
        }
}

// GEMM + LayerNorm in one kernel for layers whose output row fits one block (Cout = BN in {32, 64, 128, 256}: every
// Conv1d / Linear of the path that is followed by a LayerNorm except the widest expansions): out = act(LN(X W^T + bias +
// pre) * gamma + beta + post).  Same main loop as above (K-tile 32, register prefetch); the epilogue stages the tile in
// LDS and normalises it row by row with the lane-group arithmetic of layernorm_vec_kernel (two-pass mean / variance),
// so bias, residuals and output move as whole rows.  Saves the separate LayerNorm launch and the round trip of the
// pre-norm activations through HBM.
template <int BM, int BN, int WGM, int WGN, bool KFULL>
__global__ __launch_bounds__(256) void gemm_ln_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W, int ldw,
                                                      const float *__restrict__ bias, const float *__restrict__ pre,
                                                      const float *__restrict__ gamma, const float *__restrict__ beta,
                                                      const float *__restrict__ post, float *__restrict__ out, int ldo, int R,
                                                      int Cin, int act) {
    constexpr int KT = 32, LDS_LD = KT + 2, LPR = KT / 4, RPP = 256 / LPR;
    constexpr int WM = BM / WGM, WN = BN / WGN, MB = WM / 16, NB = WN / 16, PX = BM / RPP, PW = BN / RPP;
    static_assert(WGM * WGN == 4 && MB >= 1 && NB >= 1 && PX >= 1 && PW >= 1, "tile shape");
    constexpr int LDC = BN + 4;                  // row stride of the staged output tile
    constexpr int CH = BN >= 128 ? 32 : BM;       // rows staged at a time
    static_assert(CH % 16 == 0 && BM % CH == 0 && CH % (256 / (BN / 4)) == 0 && CH * LDC <= (BM + BN) * LDS_LD, "epilogue staging");
    constexpr int SMEM = (BM + BN) * LDS_LD;
    __shared__ __attribute__((aligned(16))) float smem[SMEM];
    float (*Xs)[LDS_LD] = reinterpret_cast<float (*)[LDS_LD]>(smem);
    float (*Ws)[LDS_LD] = reinterpret_cast<float (*)[LDS_LD]>(smem + BM * LDS_LD);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w / WGN, wn = w % WGN;
    const int row0 = blockIdx.x * BM;
    const int sr_ = t / LPR, sk = (t % LPR) * 4;

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 xr[PX], wr[PW];
#pragma unroll
    for (int p = 0; p < PX; ++p) xr[p] = load4<true, KFULL>(X, ldx, row0 + p * RPP + sr_, R, sk, Cin);
#pragma unroll
    for (int p = 0; p < PW; ++p) wr[p] = load4<true, KFULL>(W, ldw, p * RPP + sr_, BN, sk, Cin);
    for (int k0 = 0; k0 < Cin; k0 += KT) {
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
        if (k0 + KT < Cin) {
#pragma unroll
            for (int p = 0; p < PX; ++p) xr[p] = load4<true, KFULL>(X, ldx, row0 + p * RPP + sr_, R, k0 + KT + sk, Cin);
#pragma unroll
            for (int p = 0; p < PW; ++p) wr[p] = load4<true, KFULL>(W, ldw, p * RPP + sr_, BN, k0 + KT + sk, Cin);
        }
        float a[2][MB], b[2][NB];
#pragma unroll
        for (int i = 0; i < MB; ++i) a[0][i] = Xs[wm * WM + i * 16 + (lane & 15)][lane >> 4];
#pragma unroll
        for (int j = 0; j < NB; ++j) b[0][j] = Ws[wn * WN + j * 16 + (lane & 15)][lane >> 4];
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
            if (kk + 4 < KT) {
#pragma unroll
                for (int i = 0; i < MB; ++i) a[nxt][i] = Xs[wm * WM + i * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < NB; ++j) b[nxt][j] = Ws[wn * WN + j * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
            }
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[cur][j], a[cur][i], acc[i][j], 0, 0, 0), mfma_pace();
        }
        __syncthreads();
    }
    // ---- epilogue: the tile goes through LDS (the operand tiles are dead after the loop's last barrier), CH rows at a
    //      time so that it fits the operand tiles' footprint, and is walked ROW-wise: G = BN / 4 lanes hold one row (a
    //      float4 each), so bias / pre / post / out are whole-row accesses and the two LayerNorm sums are lane-group
    //      reductions -- the arithmetic of layernorm_vec_kernel
    constexpr int G = BN / 4, RPS = 256 / G;
    const int cr = t / G, cc = (t % G) * 4;
    const float4 bv = bias ? *reinterpret_cast<const float4 *>(bias + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g4 = *reinterpret_cast<const float4 *>(gamma + cc), b4 = *reinterpret_cast<const float4 *>(beta + cc);
    // neighbours first, like layernorm_vec_kernel's group_sum: the same association of the additions, so this kernel and
    // the GEMM kernel followed by the LayerNorm kernel give bit-identical rows (ops.linear_layernorm picks between them
    // by row count, and a frame's result must not depend on the batch it travels in)
    auto gsum = [](float v) {
#pragma unroll
        for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
#pragma unroll
    for (int ch = 0; ch < BM / CH; ++ch) {
        if (ch) __syncthreads();  // the previous chunk has been read
#pragma unroll
        for (int i = 0; i < MB; ++i) {
            const int rl = wm * WM + i * 16;  // first tile row of this 16-row block (wave-uniform)
            if (rl / CH != ch) continue;
#pragma unroll
            for (int j = 0; j < NB; ++j)
                *reinterpret_cast<float4 *>(&smem[(rl - ch * CH + (lane & 15)) * LDC + wn * WN + j * 16 + (lane >> 4) * 4]) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
        __syncthreads();
#pragma unroll 4
        for (int p = 0; p < CH / RPS; ++p) {
            const int r = row0 + ch * CH + p * RPS + cr, rr = min(r, R - 1);
            float4 v = *reinterpret_cast<const float4 *>(&smem[(p * RPS + cr) * LDC + cc]);
            v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
            if (pre) {
                const float4 pv = *reinterpret_cast<const float4 *>(pre + (size_t)rr * BN + cc);
                v.x += pv.x, v.y += pv.y, v.z += pv.z, v.w += pv.w;
            }
            const float mu = gsum((v.x + v.y) + (v.z + v.w)) / (float)BN;
            v.x -= mu, v.y -= mu, v.z -= mu, v.w -= mu;
            const float rs = rsqrtf(gsum(fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, v.w * v.w)))) / (float)BN + 1e-5f);
            float4 o = make_float4(fmaf(v.x * rs, g4.x, b4.x), fmaf(v.y * rs, g4.y, b4.y), fmaf(v.z * rs, g4.z, b4.z),
                                   fmaf(v.w * rs, g4.w, b4.w));
            if (post) {
                const float4 pv = *reinterpret_cast<const float4 *>(post + (size_t)rr * BN + cc);
                o.x += pv.x, o.y += pv.y, o.z += pv.z, o.w += pv.w;
            }
            o.x = apply_act(o.x, act), o.y = apply_act(o.y, act), o.z = apply_act(o.z, act), o.w = apply_act(o.w, act);
            if (r < R) *reinterpret_cast<float4 *>(out + (size_t)r * ldo + cc) = o;
        }
    }
}

// Large-shape variant: block tile 128x128, 4 waves as 2x2, wave tile 64x64 = 2x2 blocks of
// v_mfma_f32_32x32x2_f32 (64 accumulator registers; half the LDS fragment traffic and half the
// global->LDS staging per flop of the 64x64 kernel).  K-tile 32, LDS row stride 36 floats: 16-byte
// aligned rows for ds_write_b128 / ds_read_b128, and rows r..r+7 start 4 banks apart, so a
// quarter-wave's b128 reads cover all banks once.
// One float4 per lane feeds FOUR MFMAs: the instruction wants A[i = l & 31][k = l >> 5] (two k per
// issue); lanes < 32 read k = 8g..8g+3 and lanes >= 32 read k = 8g+4..8g+7 of their row, and MFMA c
// takes component c of both operands -- it contracts k in {8g+c, 8g+4+c}.  A and B use the same
// map, so the four issues together contract k = 8g..8g+7 exactly once.
using f32x16 = __attribute__((ext_vector_type(16))) float;

template <int KT = 32>
__global__ __launch_bounds__(256, 2) void gemm_nt_mfma128_kernel(const float *__restrict__ X0, int ldx, long long sx,
                                                                 const float *__restrict__ W0, int ldw, long long sw,
                                                                 const float *__restrict__ bias,
                                                                 const float *__restrict__ res0, int ldr, long long sr,
                                                                 float *__restrict__ out0, int ldo, long long so, int R,
                                                                 int Cin, int Cout, int act, int gx, int gy, int ntiles) {
    constexpr int BM = 128, BN = 128, LD = KT + 4, LPR = KT / 4, RPP = 256 / LPR, PX = BM / RPP, PW = BN / RPP;
    __shared__ __attribute__((aligned(16))) float Xs[BM][LD];
    __shared__ __attribute__((aligned(16))) float Ws[BN][LD];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    const int sr_ = t / LPR, sk = (t % LPR) * 4;
    const int fr = lane & 31, fk = (lane >> 5) * 4;
    const bool swz = (gy & 7) == 0 && gy >= 64 && (gridDim.x & 7) == 0;
    const int per = gx * gy;

    // Persistent blocks: block b walks tiles b, b + G, ...  (G % 8 == 0 keeps a block's tiles on the XCD-chunk
    // its first tile mapped to).  The first K-tile of the NEXT output tile is requested before the current
    // tile's epilogue, so neither the load latency nor the stores sit between two MFMA phases.
    int bz, row0, col0;
    auto place = [&](int L) {
        bz = L / per;
        const int l = L - bz * per;
        int by = l / gx, bx = l - by * gx;
        if (swz) {  // all column blocks of one row block on ONE XCD: its L2 serves X
            const int xcd = l & 7, slot = l >> 3;
            by = (slot / gx) * 8 + xcd, bx = slot % gx;
        }
        row0 = by * BM, col0 = bx * BN;
    };
    float4 xr[PX], wr[PW];
    auto request = [&](int k) {
        const float *X = X0 + (size_t)bz * sx, *W = W0 + (size_t)bz * sw;
#pragma unroll
        for (int p = 0; p < PX; ++p) xr[p] = load4<true>(X, ldx, row0 + p * RPP + sr_, R, k + sk, Cin);
#pragma unroll
        for (int p = 0; p < PW; ++p) wr[p] = load4<true>(W, ldw, col0 + p * RPP + sr_, Cout, k + sk, Cin);
    };
    int tile = blockIdx.x;
    if (tile < ntiles) place(tile), request(0);
    for (; tile < ntiles; tile += gridDim.x) {
        const int my_bz = bz, my_row0 = row0, my_col0 = col0;
        f32x16 acc[2][2];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
        for (int k0 = 0; k0 < Cin; k0 += KT) {
#pragma unroll
            for (int p = 0; p < PX; ++p) *reinterpret_cast<float4 *>(&Xs[p * RPP + sr_][sk]) = xr[p];
#pragma unroll
            for (int p = 0; p < PW; ++p) *reinterpret_cast<float4 *>(&Ws[p * RPP + sr_][sk]) = wr[p];
            __syncthreads();
            if (k0 + KT < Cin) {
                request(k0 + KT);
            } else if (tile + (int)gridDim.x < ntiles) {
                place(tile + gridDim.x), request(0);
            }
            float4 a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[0][i] = *reinterpret_cast<const float4 *>(&Xs[wm * 64 + i * 32 + fr][fk]);
#pragma unroll
            for (int j = 0; j < 2; ++j) b[0][j] = *reinterpret_cast<const float4 *>(&Ws[wn * 64 + j * 32 + fr][fk]);
#pragma unroll
            for (int g = 0; g < KT / 8; ++g) {
                const int cur = g & 1, nxt = cur ^ 1;
                if (g + 1 < KT / 8) {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
                        a[nxt][i] = *reinterpret_cast<const float4 *>(&Xs[wm * 64 + i * 32 + fr][(g + 1) * 8 + fk]);
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        b[nxt][j] = *reinterpret_cast<const float4 *>(&Ws[wn * 64 + j * 32 + fr][(g + 1) * 8 + fk]);
                }
#define DPM_MFMA4(c)                                                                                              \
    _Pragma("unroll") for (int i = 0; i < 2; ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j) acc[i][j] =        \
        __builtin_amdgcn_mfma_f32_32x32x2f32(b[cur][j].c, a[cur][i].c, acc[i][j], 0, 0, 0)
                DPM_MFMA4(x);
                DPM_MFMA4(y);
                DPM_MFMA4(z);
                DPM_MFMA4(w);
#undef DPM_MFMA4
            }
            __syncthreads();
        }
        // Transposed result blocks (W is the A operand): D[m][n], n = lane & 31 -> output row,
        // m = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) -> output column: four 16-byte stores per block.
        float *out = out0 + (size_t)my_bz * so;
        const float *res = res0 ? res0 + (size_t)my_bz * sr : nullptr;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int r = my_row0 + wm * 64 + i * 32 + (lane & 31);
                if (r >= R) continue;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int c = my_col0 + wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5);
                    if (c >= Cout) continue;  // Cout % 4 == 0 (dispatch): all four columns exist
                    float4 v = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
                    if (bias) {
                        const float4 bv = *reinterpret_cast<const float4 *>(bias + c);
                        v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                    }
                    if (res) {
                        const float4 rv = *reinterpret_cast<const float4 *>(res + (size_t)r * ldr + c);
                        v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
                    }
                    v.x = apply_act(v.x, act), v.y = apply_act(v.y, act), v.z = apply_act(v.z, act), v.w = apply_act(v.w, act);
                    *reinterpret_cast<float4 *>(out + (size_t)r * ldo + c) = v;
                }
            }
    }
}

// Wave-specialised variant for the big K = 256 .. 1024 shapes of the decoder (round 4).  What the measurements of
// profiles/r04_corun.md say about this chip: a SIMD's time is (matrix-pipe busy time) + (vector-ALU busy time) -- the two never
// overlap, neither across kernels nor across waves of one kernel -- so the only way to a shorter step is a matrix kernel
// whose pipe is busy while it runs.  The 64 x 64 kernel above keeps eight waves per SIMD that each load, stage, synchronise
// twice per K-tile and issue 32 MFMAs in between (48-58 % pipe utilisation).  Here a 512-thread workgroup owns a CU:
//   waves 0-3 (one per SIMD) ONLY feed the matrix pipe: a 64 x 64 register tile each (2 x 2 waves = a 128 x 128 block),
//             per k-step of four 8 LDS fragment reads against 16 MFMAs, ONE barrier per K-tile of 32 (128 MFMAs);
//   waves 4-7 ONLY move data: the next K-tile global -> registers -> LDS while the current one is consumed (two stages),
//             running ahead across tile boundaries (persistent workgroups: block b walks tiles b, b + G, ...).
// Same instruction and k order as the kernels above, so every output element has the same bits.
#ifdef DPM_EXPERIMENT
__device__ long long dpm_ws_trace_buf[1024];   // s_memtime stamps of block 0, matrix wave 0 (scripts/gemm_ws_check.py)
#define DPM_WS_STAMP(slot)                                                             \
    do {                                                                               \
        if (blockIdx.x == 0 && t == 0 && (slot) < 1024) dpm_ws_trace_buf[slot] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define DPM_WS_STAMP(slot) do { } while (0)
#endif
// PACE: wait states after every MFMA (A/B builds); ABL (timing experiments, -DDPM_EXPERIMENT builds only): 1 = no output
// stores, 2 = no global loads, 4 = no LDS staging stores
// The workgroup barrier of the wave-specialised kernel orders LDS traffic only: __syncthreads() also drains the vector
// memory counter, which made the matrix waves wait for the data waves' output stores (229 against 147 us).
__device__ __forceinline__ void ws_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <int PACE, int ABL = 0>
__global__ __launch_bounds__(512, 1) void gemm_ws_kernel(const float *__restrict__ X, int ldx, const float *__restrict__ W, int ldw,
                                                         const float *__restrict__ bias, const float *__restrict__ res, int ldr,
                                                         float *__restrict__ out, int ldo, int R, int Cin, int Cout, int act,
                                                         int gx, int gy, int ntiles) {
    constexpr int BM = 128, BN = 128, KT = 32, LD = KT + 2, STAGE = (BM + BN) * LD, NS = 2, LDC = BN + 4;
    // two operand stages (69 632 B) + the finished tile on its way out (67 584 B): more than half of the CU's 160 KB, so
    // ONE workgroup per CU and one matrix wave per SIMD
    __shared__ __attribute__((aligned(16))) float smem[NS * STAGE + BM * LDC];
    float *otile = smem + NS * STAGE;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int G = gridDim.x;
    const int my_tiles = ((int)blockIdx.x < ntiles) ? (ntiles - (int)blockIdx.x + G - 1) / G : 0;
    const int kpt = Cin / KT, total = my_tiles * kpt;   // K-tiles per output tile (>= 2: dispatch), K-tiles of this workgroup
    const bool swz = (gy & 7) == 0 && gy >= 64 && (G & 7) == 0;
    auto place = [&](int L, int &row0, int &col0) {
        int by = L / gx, bx = L - by * gx;
        if (swz) {  // all column blocks of one row block on ONE XCD: its L2 serves X
            const int xcd = L & 7, slot = L >> 3;
            by = (slot / gx) * 8 + xcd, bx = slot % gx;
        }
        row0 = by * BM, col0 = bx * BN;
    };
    if (total == 0) return;
    if (wave >= 4) {
        // ---------------------------------------------------------------- data waves: operand staging + the tiles' way out
        const int lt_ = t - 256, sr = lt_ >> 3, sk = (lt_ & 7) * 4;   // 8 lanes per row of 32 floats, 32 rows per pass
        float4 xr[4], wr[4];
        int rt = 0, rk = 0, row0, col0;   // the K-tile requested next: local tile, K-tile inside it
        place((int)blockIdx.x, row0, col0);
        auto request = [&]() {
            const int k0 = rk * KT + sk;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                if (ABL & 2) {
                    xr[p] = wr[p] = make_float4(1.f, 1.f, 1.f, 1.f);
                    continue;
                }
                xr[p] = *reinterpret_cast<const float4 *>(X + (size_t)min(row0 + p * 32 + sr, R - 1) * ldx + k0);
                wr[p] = *reinterpret_cast<const float4 *>(W + (size_t)min(col0 + p * 32 + sr, Cout - 1) * ldw + k0);
            }
            if (++rk == kpt) {
                rk = 0, ++rt;
                if (rt < my_tiles) place((int)blockIdx.x + rt * G, row0, col0);
            }
        };
        auto store = [&](int stage) {
            float *As = smem + stage * STAGE, *Bs = As + BM * LD;
            if (ABL & 4) return;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                float2 *d = reinterpret_cast<float2 *>(As + (p * 32 + sr) * LD + sk);
                d[0] = make_float2(xr[p].x, xr[p].y), d[1] = make_float2(xr[p].z, xr[p].w);
                float2 *e = reinterpret_cast<float2 *>(Bs + (p * 32 + sr) * LD + sk);
                e[0] = make_float2(wr[p].x, wr[p].y), e[1] = make_float2(wr[p].z, wr[p].w);
            }
        };
        // The finished tile (the matrix waves left it in `otile` one barrier ago): bias / residual / activation, whole
        // 512-byte rows per store instruction.  A thread keeps its four columns for all 16 row passes.
        const int cr = lt_ >> 5, cc = (lt_ & 31) * 4;
        auto emit = [&](int tile_index, int p0, int p1, auto act_tag) {   // row passes [p0, p1) of 16 (8 rows each)
            constexpr int ACT = decltype(act_tag)::value;
            int r0, c0;
            place((int)blockIdx.x + tile_index * G, r0, c0);
            const int c = c0 + cc;
            const bool cok = c < Cout;   // Cout % 4 == 0 (dispatch): a thread's four columns exist together
            const float4 bv = bias && cok ? *reinterpret_cast<const float4 *>(bias + c) : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int q = p0; q < p1; q += 2) {   // two passes' residual rows in flight together
                float4 rv[2];
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int r = r0 + (q + p) * 8 + cr;
                    rv[p] = res ? *reinterpret_cast<const float4 *>(res + (size_t)min(r, R - 1) * ldr + min(c, Cout - 4))
                                : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int p = 0; p < 2; ++p) {
                    const int rl = (q + p) * 8 + cr, r = r0 + rl;
                    float4 v = *reinterpret_cast<const float4 *>(otile + min(rl, BM - 1) * LDC + cc);
                    v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                    if (res) v.x += rv[p].x, v.y += rv[p].y, v.z += rv[p].z, v.w += rv[p].w;
                    v.x = apply_act(v.x, ACT), v.y = apply_act(v.y, ACT), v.z = apply_act(v.z, ACT), v.w = apply_act(v.w, ACT);
                    if (q + p < p1 && r < R && cok && !(ABL & 1)) *reinterpret_cast<float4 *>(out + (size_t)r * ldo + c) = v;
                }
            }
        };
        auto emit_tile = [&](int tile_index, int p0, int p1) {
            if (act == DPM_ACT_RELU) emit(tile_index, p0, p1, std::integral_constant<int, DPM_ACT_RELU>{});
            else if (act == DPM_ACT_SIGMOID) emit(tile_index, p0, p1, std::integral_constant<int, DPM_ACT_SIGMOID>{});
            else emit(tile_index, p0, p1, std::integral_constant<int, DPM_ACT_NONE>{});
        };
        const int per_step = (16 + kpt - 2) / (kpt - 1);   // row passes per step: a tile leaves over the kpt - 1 steps it has
        // K-tile m lives in stage m % NS; at step n the matrix waves consume K-tile n while K-tile n + NS - 1 goes into the
        // stage K-tile n - 1 left at the previous barrier.  K-tile n closes a tile when (n + 1) % kpt == 0; the matrix waves
        // park it in `otile` between barriers n and n + 1, so it is ours from step n + 2 on (and free again long before the
        // next tile closes, kpt >= 2 steps later).  It leaves in slices, a few row passes per step: all 64 KB at once was a
        // store burst on every CU at the same moment whose back-pressure held the data waves -- and with them the barrier.
        request();
        for (int m = 0; m < NS - 1 && m < total; ++m) {
            store(m);
            if (m + 1 < total) request();
        }
        ws_barrier();
        for (int n = 0; n < total; ++n) {
            if (n + NS - 1 < total) store((n + NS - 1) % NS);
            // the slice goes out BEFORE the next K-tile is requested: its bias / residual loads are waited for inside, and the
            // in-order memory counter would make that wait cover the operand loads too (a DRAM latency per step)
            if (n >= 2) {
                const int q = (n - 1) / kpt, sl = (n - 1) - q * kpt;   // step sl of the tile that closed at K-tile q * kpt - 1
                if (q >= 1 && sl <= kpt - 2 && sl * per_step < 16) emit_tile(q - 1, sl * per_step, min(16, (sl + 1) * per_step));
            }
            if (n + NS < total) request();
            ws_barrier();
        }
        ws_barrier();   // the last tile is parked behind this one
        emit_tile(my_tiles - 1, 0, 16);
        return;
    }
    // -------------------------------------------------------------------- matrix waves
    const int wm = wave >> 1, wn = wave & 1;
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fa = (wm * 64 + (lane & 15)) * LD + (lane >> 4), fb = (BM + wn * 64 + (lane & 15)) * LD + (lane >> 4);
    ws_barrier();
    int kin = 0;
    DPM_WS_STAMP(0);
    for (int n = 0; n < total; ++n) {
        const float *S = smem + (n % NS) * STAGE;
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[0][i] = S[fa + i * 16 * LD];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[0][j] = S[fb + j * 16 * LD];
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
            if (kk + 4 < KT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[nxt][i] = S[fa + i * 16 * LD + kk + 4];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[nxt][j] = S[fb + j * 16 * LD + kk + 4];
            }
            // the fragment reads of the NEXT k-step stay above this step's 16 MFMAs (512 cycles of cover); left alone
            // the scheduler sinks them to three MFMAs before their use and every other k-step waits for LDS
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[cur][j], a[cur][i], acc[i][j], 0, 0, 0);
                    if (PACE > 0) asm volatile("s_nop %0" ::"n"(PACE > 0 ? PACE - 1 : 0));
                }
            __builtin_amdgcn_sched_barrier(0);
        }
        DPM_WS_STAMP(1 + 3 * n);       // MFMAs of the K-tile issued
        ws_barrier();
        DPM_WS_STAMP(2 + 3 * n);       // barrier passed
        if (++kin == kpt) {
            // the tile is complete: park it in LDS for the data waves (a lane owns four consecutive columns of one row per
            // block: W is the instruction's A operand) and go on with the next tile's MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    *reinterpret_cast<float4 *>(otile + (wm * 64 + i * 16 + (lane & 15)) * LDC + wn * 64 + j * 16 + (lane >> 4) * 4) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
                    acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
            kin = 0;
        }
        DPM_WS_STAMP(3 + 3 * n);       // (tile parked, when the K-tile closed one)
    }
    ws_barrier();
}

}  // namespace

#ifdef DPM_EXPERIMENT
extern "C" int dpm_debug_ws_trace(long long *host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dpm_ws_trace_buf), sizeof(long long) * (size_t)std::min(n, 1024));
}
#endif

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
    // Long reductions over a big output (K >= 1024, >= 512 tiles of 128x128) amortise the 128x128 kernel's tile
    // switch: 128 vs 110 TFLOP/s at 4096^3.  Every shape of the registration path has K <= 512 with 32768 rows
    // or fewer, where the 64x64 kernel's higher occupancy measured equal or better (scripts/gemm_bench.py).
    const bool vec_out = (ldo & 3) == 0 && ((uintptr_t)out & 15) == 0 && so % 4 == 0 && Cout % 4 == 0 &&
                         (!bias || ((uintptr_t)bias & 15) == 0) &&
                         (!residual || ((ldr & 3) == 0 && ((uintptr_t)residual & 15) == 0 && sr % 4 == 0));
    const long long blocks128 = (long long)batch * dpm_cdiv(R, 128) * dpm_cdiv(Cout, 128);
    if (vec && vec_out && Cin >= 1024 && blocks128 >= 512) {
        const int gx = dpm_cdiv(Cout, 128), gy = dpm_cdiv(R, 128), ntiles = batch * gx * gy;
        hipLaunchKernelGGL((gemm_nt_mfma128_kernel<32>), dim3(min(ntiles, 512)), dim3(256), 0, st, x, ldx, sx, W, ldw, sw,
                           bias, residual, ldr, sr, out, ldo, so, R, Cin, Cout, act, gx, gy, ntiles);
        return dpm_launch_status();
    }
    // wave-specialised 128 x 128 kernel: big unbatched shapes with whole K-tiles and at least one tile per CU
    const long long tiles_ws = (long long)dpm_cdiv(R, 128) * dpm_cdiv(Cout, 128);
    const int ws_mode = dpm_knob("DPM_GEMM_WS", DPM_GEMM_WS_DEFAULT);
    if (ws_mode && batch == 1 && vec && vec_out && Cin % 32 == 0 && Cin >= 64 && Cout >= 128 && tiles_ws >= 256 && tiles_ws < (1 << 30)) {
        const int gx = dpm_cdiv(Cout, 128), gy = dpm_cdiv(R, 128), ntiles = gx * gy;
        const int G = (int)std::min<long long>(ntiles, 256);
#ifdef DPM_EXPERIMENT
#define DPM_WS_ABL(m, abl)                                                                                                   \
    if (ws_mode == m) {                                                                                                      \
        hipLaunchKernelGGL((gemm_ws_kernel<0, abl>), dim3(G), dim3(512), 0, st, x, ldx, W, ldw, bias, residual, ldr, out, ldo, R, \
                           Cin, Cout, act, gx, gy, ntiles);                                                                   \
        return dpm_launch_status();                                                                                          \
    }
        DPM_WS_ABL(11, 1) DPM_WS_ABL(12, 2) DPM_WS_ABL(13, 3) DPM_WS_ABL(17, 7)
#undef DPM_WS_ABL
#endif
        if (ws_mode == 2)
            hipLaunchKernelGGL((gemm_ws_kernel<2>), dim3(G), dim3(512), 0, st, x, ldx, W, ldw, bias, residual, ldr, out, ldo, R, Cin,
                               Cout, act, gx, gy, ntiles);
        else
            hipLaunchKernelGGL((gemm_ws_kernel<0>), dim3(G), dim3(512), 0, st, x, ldx, W, ldw, bias, residual, ldr, out, ldo, R, Cin,
                               Cout, act, gx, gy, ntiles);
        return dpm_launch_status();
    }
    const bool t64 = big >= 192 || (R > 1024 && Cout > 32);
    const dim3 grid = t64 ? dim3(dpm_cdiv(Cout, 64), dpm_cdiv(R, 64), batch) : dim3(dpm_cdiv(Cout, 32), dpm_cdiv(R, 32), batch);
#define DPM_GEMM_LAUNCH(BM, BN, V, KF)                                                                                \
    hipLaunchKernelGGL((gemm_nt_mfma_kernel<BM, BN, V, 32, KF>), grid, dim3(256), lds_pad, st, x, ldx, sx, W, ldw, sw, bias,   \
                       residual, ldr, sr, out, ldo, so, R, Cin, Cout, act)
    const bool kfull = vec && Cin % 32 == 0;
    // -DDPM_EXPERIMENT builds only: unused dynamic LDS caps the workgroups per CU (scripts/corun_micro.py: occupancy shapes)
    const size_t lds_pad = (size_t)dpm_knob("DPM_GEMM_LDS_PAD", 0);
    if (t64) {
        if (kfull) DPM_GEMM_LAUNCH(64, 64, true, true);
        else if (vec) DPM_GEMM_LAUNCH(64, 64, true, false);
        else DPM_GEMM_LAUNCH(64, 64, false, false);
    } else {
        if (kfull) DPM_GEMM_LAUNCH(32, 32, true, true);
        else if (vec) DPM_GEMM_LAUNCH(32, 32, true, false);
        else DPM_GEMM_LAUNCH(32, 32, false, false);
    }
#undef DPM_GEMM_LAUNCH
    return dpm_launch_status();
}

extern "C" int dpm_linear(const float *x, int ldx, const float *W, int ldw, const float *bias, const float *residual,
                          int ldr, float *out, int ldo, int R, int Cin, int Cout, int act, dpm_stream_t stream) {
    return dpm_linear_batched(x, ldx, 0, W, ldw, 0, bias, residual, ldr, 0, out, ldo, 0, 1, R, Cin, Cout, act, stream);
}

// Conv1d(k=1)/Linear + LayerNorm1d (+ residuals, + activation) as ONE kernel: out = act(LN(x W^T + bias + pre) * gamma +
// beta + post), rows of Cout in {32, 64, 128, 256}; pre / post are packed (R, Cout).  DPM_EUNSUPPORTED for other
// widths or unaligned operands: the caller then runs dpm_linear + dpm_layernorm.
extern "C" int dpm_linear_layernorm(const float *x, int ldx, const float *W, int ldw, const float *bias, const float *pre,
                                    const float *gamma, const float *beta, const float *post, float *out, int ldo, int R,
                                    int Cin, int Cout, int act, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && W && gamma && beta && out && R >= 1 && Cin >= 1 && Cout >= 1 && ldx >= Cin && ldw >= Cin && ldo >= Cout);
    DPM_CHECK_ARG(act >= DPM_ACT_NONE && act <= DPM_ACT_SIGMOID);
    auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    if (!(ldx % 4 == 0 && ldw % 4 == 0 && ldo % 4 == 0 && Cin % 4 == 0 && al(x) && al(W) && al(bias) && al(pre) && al(gamma) &&
          al(beta) && al(post) && al(out)))
        return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define DPM_GLN(BM, BN, WGM, WGN)                                                                                          \
    do {                                                                                                                   \
        if (Cin % 32 == 0)                                                                                                 \
            hipLaunchKernelGGL((gemm_ln_kernel<BM, BN, WGM, WGN, true>), dim3(dpm_cdiv(R, BM)), dim3(256), 0, st, x, ldx, W, ldw, \
                               bias, pre, gamma, beta, post, out, ldo, R, Cin, act);                                       \
        else                                                                                                               \
            hipLaunchKernelGGL((gemm_ln_kernel<BM, BN, WGM, WGN, false>), dim3(dpm_cdiv(R, BM)), dim3(256), 0, st, x, ldx, W, ldw, \
                               bias, pre, gamma, beta, post, out, ldo, R, Cin, act);                                       \
    } while (0)
    // 256 columns: 32-row tiles (39 KB of LDS: four workgroups per CU instead of two with 64 rows; 65 536 x 64 -> 256 takes
    // 39 us instead of 50, 16 384 x 256 -> 256 30 instead of 34, the 32 768-row decoder blocks are unchanged)
    if (Cout == 256) DPM_GLN(32, 256, 1, 4);
    else if (Cout == 128) DPM_GLN(64, 128, 2, 2);
    else if (Cout == 64) DPM_GLN(64, 64, 2, 2);
    else if (Cout == 32) DPM_GLN(128, 32, 4, 1);
    else return DPM_EUNSUPPORTED;
#undef DPM_GLN
    return dpm_launch_status();
}
