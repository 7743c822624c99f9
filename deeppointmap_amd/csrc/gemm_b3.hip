// fp32 GEMM through the bf16 matrix pipe: out = act(X W^T + bias + residual) with every fp32 operand split EXACTLY into
// three bf16 terms and the product formed from the six term pairs that matter (round 4).
//
// Why: profiles/r04_corun.md -- on this chip a SIMD's time is (matrix-pipe busy) + (vector-ALU busy), the two never overlap,
// so the cycles v_mfma_f32_16x16x4_f32 holds the matrix pipe (it runs at the fp32 VECTOR rate, 1/16 of the bf16 rate) are
// cycles nothing else can use.  x = hi + mid + lo with hi / mid / lo the top, middle and bottom 8 bits of the 24-bit
// significand, each a truncation, reproduces x bit for bit; a pair product of two bf16 values is exact in fp32;
//     x w = hi hi + hi mid + mid hi + hi lo + lo hi + mid mid        (+ three terms below 2^-23 |x w|),
// accumulated in fp32 by v_mfma_f32_16x16x32_bf16: six bf16 instructions of 4 passes do the work of eight fp32 ones of 8
// passes -- 3/8 of the matrix-pipe time.  The result carries fp32-accumulation rounding like the kernels of gemm.hip (measured
// against fp64 in tests/test_gpu_ops.py at the same error level) but not their bits: a layer uses ONE of the two kernels
// whatever the batch it sees (ops.linear decides by layer shape, never by row count).
//
// Same structure as gemm_nt_mfma_kernel<64,64> on purpose (eight small workgroups per CU hide each other's latencies: the
// 128 x 128 form of round 2 was faster alone and slower inside the pipeline): block tile 64 x 64, 4 waves as 2 x 2, wave tile
// 32 x 32 = 2 x 2 blocks of 16 x 16; K-tile 32 = ONE bf16 instruction deep.  Weights arrive pre-split (dpm_split_bf16x3, once
// per weight version), activations are split while they are staged: LDS holds three bf16 planes per operand in rows of 64
// bytes whose 16-byte chunks (a lane's 8 consecutive k: one ds_read_b128 per fragment) are swizzled against bank conflicts
// (b3_col, dpm_common.h), 24 576 B per workgroup.
#include "dpm_common.h"

namespace {

using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using f32x4 = __attribute__((ext_vector_type(4))) float;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

__device__ __forceinline__ float b3_act(float v, int act) {
    if (act == DPM_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DPM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// split3 / pack2: dpm_common.h
__global__ __launch_bounds__(256) void split_bf16x3_kernel(const float *__restrict__ W, long long n, uint16_t *__restrict__ planes) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    unsigned h, m, l;
    split3(W[i], h, m, l);
    planes[i] = (uint16_t)(h >> 16), planes[n + i] = (uint16_t)(m >> 16), planes[2 * n + i] = (uint16_t)(l >> 16);
}

// 64 x 128 tiles (wave tile 32 x 64: every split X fragment feeds four column blocks instead of two -- the split is vector-ALU
// work that ADDS to the matrix time on this chip) for problems with at least this many of them (one round of four workgroups
// per CU): 32 768 x 256 -> 768 97 -> 93 us alone, pipelined step 4.30 -> 4.28 ms.  Same bits as the other tile shapes.  0: never.
#ifndef DPM_B3_WIDE
#define DPM_B3_WIDE 1024
#endif
#ifndef DPM_B3_DEEP
#define DPM_B3_DEEP 256    // launches of at most this many workgroups (one per CU) keep four K-tiles in flight
#endif
constexpr int B3_KT = 32, B3_LD = B3_KT;       // K-tile; LDS rows of 64 bytes, chunks swizzled (b3_col, dpm_common.h)

// BM x BN = 64 x 128 (wave tile 32 x 64) for the large problems, 64 x 64 (wave tile 32 x 32 = 2 x 2 blocks), or 32 x 32 (one block
// per wave) for problems too small to fill the chip with 64 x 64 tiles; an output element sees the same instructions in the same order either way (same bits).
// XVEC: the rows of X are 16-byte aligned (false: four scalar loads per group -- a token matrix with rows of 131 floats must
// take the same kernel as one with rows of 132, or a layer's bits would depend on how its input happens to be laid out)
// PF: K-tiles in flight (a ring of PF register sets for both operands, the loop unrolled PF times so that the ring index is
// static).  1 for launches that fill the chip several times over -- there other workgroups cover a tile's load latency and the
// registers of a deeper ring cost occupancy (measured: 97 -> 99.9 / 99.4 / 106.5 us at 2 / 3 / 4 on 32 768 x 256 -> 768) --,
// 4 for small launches (at most a few workgroups per CU: the registration of ONE pair, the encoder's lower levels), where every
// trip of the K loop otherwise waits out a whole L2 / HBM round trip.  Same instructions per output element: same bits.
// Output columns >= col0 leave the kernel as the attention kernel's operand planes instead of fp32 rows (round 5; decoder_ops.hip,
// attention_kernel<PRE>): the layer is a q | k | v projection over sequences of `tokens` rows (a multiple of 64), columns col0 ..
// are K then V, `heads` heads of 32 each, and a 64-row tile is exactly one 64-key tile of one sequence.  Per (sequence, head, tile)
// one image of 2 x 6144 uint16 at p: the three bf16 planes of K[key][d] in b3_col-swizzled 64-byte rows, then the three planes of
// V TRANSPOSED, [d][slot] in 128-byte rows with the chunk swizzle and the key <-> slot order of the attention kernel's P V product.
// The values split are the ones the fp32 path would have stored (accumulator + bias): the attention kernel's own staging makes
// the same planes of them, bit for bit.
struct KvPlanes {
    uint16_t *p;
    int col0, tokens, heads;
};

template <int BM, int BN, bool XVEC, int PF, bool KVP>
__device__ __forceinline__ void gemm_b3_body(const float *__restrict__ X, int ldx, const uint16_t *__restrict__ Wp, int ldw,
                                             long long plane, const float *__restrict__ bias,
                                             const float *__restrict__ res, int ldr, float *__restrict__ out, int ldo,
                                             int R, int Cin, int Cout, int act, const float *__restrict__ r3x,
                                             const float *__restrict__ r3w, int ldr3, float r3s, KvPlanes kv) {
    // r3x / r3w (round 5): a rank-3 term in the epilogue, out += r3s * (r3x[row, 0:3] . r3w[col, 0:3]) in fp32 -- the relative-
    // coordinate columns of a grouping layer applied to the POINT's own coordinates (group_mlp.hip, "folded" gather)
    constexpr int LDC = BN + 4, MB = BM / 32, NB = BN / 32, PX = BM / 32, WT = BN * 4, PW = (WT + 255) / 256;   // WT: 16-byte pieces of a W plane tile
    constexpr int SM0 = 3 * (BM + BN) * B3_LD, SM1 = 2 * BM * LDC, SM = SM0 > SM1 ? SM0 : SM1;   // operand planes | staged output tile
    __shared__ __attribute__((aligned(16))) uint16_t smem[SM];   // 24 576 B at 64 x 64: X planes, then W planes
    uint16_t (*Xs)[BM][B3_LD] = reinterpret_cast<uint16_t (*)[BM][B3_LD]>(smem);
    uint16_t (*Ws)[BN][B3_LD] = reinterpret_cast<uint16_t (*)[BN][B3_LD]>(smem + 3 * BM * B3_LD);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    int by = blockIdx.y, bx = blockIdx.x;
    if ((gridDim.y & 7) == 0 && gridDim.y >= 64) {  // all column blocks of one row block on ONE XCD: its L2 serves X
        const unsigned L = blockIdx.y * gridDim.x + blockIdx.x, xcd = L & 7, slot = L >> 3;
        by = (int)((slot / gridDim.x) * 8 + xcd), bx = (int)(slot % gridDim.x);
    }
    const int row0 = by * BM, col0 = bx * BN;
    // staging: X as fp32 float4 (PX per thread: rows xr_ + 32 p, 4 consecutive k), W planes as 8 bf16 = 16 bytes (one per plane
    // per staging thread: row wr_, 8 consecutive k)
    const int xr_ = t >> 3, xk = (t & 7) * 4, wr_ = min(t >> 2, BN - 1), wk = (t & 3) * 8;   // W piece p: row wr_ + 64 p
    const float *xp[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) xp[p] = X + (size_t)min(row0 + p * 32 + xr_, R - 1) * ldx + xk;
    const uint16_t *wp[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) wp[p] = Wp + (size_t)min(col0 + p * 64 + wr_, Cout - 1) * ldw + wk;
    auto load_x = [&](const float *p) {
        if (XVEC) return *reinterpret_cast<const f32x4 *>(p);
        return f32x4{p[0], p[1], p[2], p[3]};
    };
    f32x4 xv[PF][PX];
    u32x4 wv[PF][PW][3];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kd = min(d * B3_KT, Cin - B3_KT);
#pragma unroll
        for (int p = 0; p < PX; ++p) xv[d][p] = load_x(xp[p] + kd);
#pragma unroll
        for (int p = 0; p < PW; ++p)
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) wv[d][p][pl] = *reinterpret_cast<const u32x4 *>(wp[p] + pl * plane + kd);
    }
    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
    auto stage_x = [&](const f32x4 &v, int r) {
        unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
        split3(v[0], h0, m0, l0), split3(v[1], h1, m1, l1), split3(v[2], h2, m2, l2), split3(v[3], h3, m3, l3);
        *reinterpret_cast<u32x2 *>(&Xs[0][r][b3_col(r, xk)]) = u32x2{pack2(h0, h1), pack2(h2, h3)};
        *reinterpret_cast<u32x2 *>(&Xs[1][r][b3_col(r, xk)]) = u32x2{pack2(m0, m1), pack2(m2, m3)};
        *reinterpret_cast<u32x2 *>(&Xs[2][r][b3_col(r, xk)]) = u32x2{pack2(l0, l1), pack2(l2, l3)};
    };
    for (int kb = 0; kb < Cin; kb += B3_KT * PF)
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int k0 = kb + d * B3_KT;
        if (PF > 1 && k0 >= Cin) break;   // uniform
#pragma unroll
        for (int p = 0; p < PX; ++p) stage_x(xv[d][p], p * 32 + xr_);
        if (WT >= 256 || t < WT) {
#pragma unroll
            for (int p = 0; p < PW; ++p)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(&Ws[pl][p * 64 + wr_][b3_col(wr_, wk)]) = wv[d][p][pl];
        }
        __syncthreads();
        {   // the next K-tile is requested while this one feeds the MFMAs -- unconditionally (the last trip re-reads its own
            // tile): behind a branch the prefetch group is serialised behind a full wait (gemm.hip, load4)
            const int kn = min(k0 + PF * B3_KT, Cin - B3_KT);
#pragma unroll
            for (int p = 0; p < PX; ++p) xv[d][p] = load_x(xp[p] + kn);
#pragma unroll
            for (int p = 0; p < PW; ++p)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wv[d][p][pl] = *reinterpret_cast<const u32x4 *>(wp[p] + pl * plane + kn);
        }
        // (the scheduler sinks these loads below the matrix instructions to save 20 registers; pinning them up here with
        // sched_barrier(0) was measured -- 116 us alone either way, 4.47 against 4.48 ms per pipelined step -- and not kept)
        bf16x8 a[3][NB], b[3][MB];   // a: W fragments (the instruction's A operand), b: X fragments
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int j = 0; j < NB; ++j) a[pl][j] = *reinterpret_cast<const bf16x8 *>(&Ws[pl][wn * (BN / 2) + j * 16 + fr][b3_col(fr, fk)]);
#pragma unroll
            for (int i = 0; i < MB; ++i) b[pl][i] = *reinterpret_cast<const bf16x8 *>(&Xs[pl][wm * (BM / 2) + i * 16 + fr][b3_col(fr, fk)]);
        }
        // smallest terms first; (plane of W, plane of X): (1,1) (2,0) (0,2) (1,0) (0,1) (0,0)
#define DPM_B3(PWQ, PXQ)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j)        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PWQ][j], b[PXQ][i], acc[i][j], 0, 0, 0)
        DPM_B3(1, 1);
        DPM_B3(2, 0);
        DPM_B3(0, 2);
        DPM_B3(1, 0);
        DPM_B3(0, 1);
        DPM_B3(0, 0);
#undef DPM_B3
        __syncthreads();
    }
    // D[m][n] with W as the A operand: n = lane & 15 -> output row, m = (lane >> 4) * 4 + reg -> output column: a lane owns
    // four consecutive columns of one row.  The tile goes through LDS once more so that every store writes whole rows.
    float *ct = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
            *reinterpret_cast<float4 *>(&ct[(wm * (BM / 2) + i * 16 + (lane & 15)) * LDC + wn * (BN / 2) + j * 16 + (lane >> 4) * 4]) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __syncthreads();
    if (KVP && BM == 64 && BN % 32 == 0 && col0 >= kv.col0) {   // (uniform) this tile is K or V of BN / 32 heads: planes, not rows
        const int E = kv.heads * 32, colk = col0 - kv.col0, isV = colk >= E ? 1 : 0, head0 = (colk - isV * E) >> 5;
        const int sq_ = row0 / kv.tokens, kt = (row0 - sq_ * kv.tokens) >> 6, ntile = kv.tokens >> 6;
#pragma unroll 1
        for (int hh = 0; hh < BN / 32; ++hh) {
            uint16_t *img = kv.p + ((((size_t)sq_ * kv.heads + head0 + hh) * ntile + kt) * 2 + isV) * 6144;
            const float *bh = bias ? bias + col0 + hh * 32 : nullptr;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const int e = t + p * 256;
                if (!isV) {   // K[key kr][d = c4 .. c4 + 3]: the attention kernel's staging, from the tile instead of from memory
                    const int kr = e >> 3, c4 = (e & 7) * 4;
                    float4 v = *reinterpret_cast<const float4 *>(&ct[kr * LDC + hh * 32 + c4]);
                    if (bh) {
                        const float4 bv = *reinterpret_cast<const float4 *>(bh + c4);
                        v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
                    }
                    unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
                    split3(v.x, h0, m0, l0), split3(v.y, h1, m1, l1), split3(v.z, h2, m2, l2), split3(v.w, h3, m3, l3);
                    uint16_t *o = img + kr * 32 + b3_col(kr, c4);
                    *reinterpret_cast<u32x2 *>(o) = u32x2{pack2(h0, h1), pack2(h2, h3)};
                    *reinterpret_cast<u32x2 *>(o + 2048) = u32x2{pack2(m0, m1), pack2(m2, m3)};
                    *reinterpret_cast<u32x2 *>(o + 4096) = u32x2{pack2(l0, l1), pack2(l2, l3)};
                } else {      // V^T[d][slots slot0 .. slot0 + 3] = four consecutive keys of channel d
                    const int d = e & 31, slot0 = (e >> 5) * 4;
                    const int key0 = 32 * (slot0 >> 5) + 16 * ((slot0 & 7) >> 2) + 4 * ((slot0 >> 3) & 3);
                    const float b = bh ? bh[d] : 0.f;
                    unsigned hq[4], mq[4], lq[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) split3(ct[(key0 + q) * LDC + hh * 32 + d] + b, hq[q], mq[q], lq[q]);
                    uint16_t *o = img + d * 64 + (((((slot0 >> 3) ^ (d >> 1)) & 7) << 3) | (slot0 & 7));
                    *reinterpret_cast<u32x2 *>(o) = u32x2{pack2(hq[0], hq[1]), pack2(hq[2], hq[3])};
                    *reinterpret_cast<u32x2 *>(o + 2048) = u32x2{pack2(mq[0], mq[1]), pack2(mq[2], mq[3])};
                    *reinterpret_cast<u32x2 *>(o + 4096) = u32x2{pack2(lq[0], lq[1]), pack2(lq[2], lq[3])};
                }
            }
        }
        return;
    }
    constexpr int TPR = BN / 4, RPS = 256 / TPR;   // threads per tile row, rows per store pass
    const int cr = t / TPR, cc = (t % TPR) * 4, c = col0 + cc;
    if (c < Cout) {   // Cout % 4 == 0 (dispatch): a thread's four columns exist together
        float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
        if (bias) bv = *reinterpret_cast<const float4 *>(bias + c);
        float w3[4][3] = {{0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}, {0.f, 0.f, 0.f}};
        if (r3x) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int d = 0; d < 3; ++d) w3[q][d] = r3w[(size_t)(c + q) * ldr3 + d] * r3s;
        }
#pragma unroll
        for (int p = 0; p < BM / RPS; ++p) {
            const int r = row0 + p * RPS + cr;
            if (r >= R) continue;
            float4 v = *reinterpret_cast<const float4 *>(&ct[(p * RPS + cr) * LDC + cc]);
            v.x += bv.x, v.y += bv.y, v.z += bv.z, v.w += bv.w;
            if (r3x) {
                const float x0 = r3x[3 * (size_t)r], x1 = r3x[3 * (size_t)r + 1], x2 = r3x[3 * (size_t)r + 2];
                v.x = fmaf(x2, w3[0][2], fmaf(x1, w3[0][1], fmaf(x0, w3[0][0], v.x)));
                v.y = fmaf(x2, w3[1][2], fmaf(x1, w3[1][1], fmaf(x0, w3[1][0], v.y)));
                v.z = fmaf(x2, w3[2][2], fmaf(x1, w3[2][1], fmaf(x0, w3[2][0], v.z)));
                v.w = fmaf(x2, w3[3][2], fmaf(x1, w3[3][1], fmaf(x0, w3[3][0], v.w)));
            }
            if (res) {
                const float4 rv = *reinterpret_cast<const float4 *>(res + (size_t)r * ldr + c);
                v.x += rv.x, v.y += rv.y, v.z += rv.z, v.w += rv.w;
            }
            v.x = b3_act(v.x, act), v.y = b3_act(v.y, act), v.z = b3_act(v.z, act), v.w = b3_act(v.w, act);
            *reinterpret_cast<float4 *>(out + (size_t)r * ldo + c) = v;
        }
    }
}

template <int BM, int BN, bool XVEC, int PF = 1>
__global__ __launch_bounds__(256) void gemm_b3_kernel(const float *__restrict__ X, int ldx, const uint16_t *__restrict__ Wp, int ldw,
                                                      long long plane, const float *__restrict__ bias,
                                                      const float *__restrict__ res, int ldr, float *__restrict__ out, int ldo,
                                                      int R, int Cin, int Cout, int act, const float *__restrict__ r3x = nullptr,
                                                      const float *__restrict__ r3w = nullptr, int ldr3 = 0, float r3s = 0.f) {
    gemm_b3_body<BM, BN, XVEC, PF, false>(X, ldx, Wp, ldw, plane, bias, res, ldr, out, ldo, R, Cin, Cout, act, r3x, r3w, ldr3, r3s,
                                          KvPlanes{nullptr, 0, 0, 0});
}

// the q | k | v projection with its K / V columns as planes (KvPlanes).  Held to the register budget of four waves per SIMD: the
// plain 64 x 128 kernel needs 122 registers, this one asked for 130 (three waves) before it was told
template <int BN, int PF>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PF == 1 ? 4 : 3, PF == 1 ? 4 : 3))) void gemm_b3_kvp_kernel(
    const float *__restrict__ X, int ldx, const uint16_t *__restrict__ Wp, int ldw, long long plane, const float *__restrict__ bias,
    float *__restrict__ out, int ldo, int R, int Cin, int Cout, KvPlanes kv) {
    gemm_b3_body<64, BN, true, PF, true>(X, ldx, Wp, ldw, plane, bias, nullptr, 0, out, ldo, R, Cin, Cout, DPM_ACT_NONE, nullptr, nullptr, 0,
                                         0.f, kv);
}

// GEMM + LayerNorm in one kernel on the bf16x3 product (the fp32 form: gemm_ln_kernel, gemm.hip): out = act(LN(X W^T + bias +
// pre) * gamma + beta + post) for layers whose output row fits one workgroup (Cout = BN in {32, 64, 128, 256}).  The main loop
// is gemm_b3_kernel's (same instructions in the same order per output element, so the pre-norm values are the plain kernel's
// bits and the two-kernel form -- dpm_linear_bf16x3 + dpm_layernorm -- gives identical rows); WGM x WGN waves of (BM / WGM) x
// (BN / WGN), 512 threads at 256 columns.  The epilogue stages the whole tile in LDS (inside the operand planes' footprint)
// and normalises it row-wise with the lane-group arithmetic of layernorm_vec_kernel / gemm_ln_kernel.
// NP > 1: the BN columns in NP passes of BN / NP over the same rows (X staged and split again per pass, W tile and LDS footprint
// 1 / NP as large, the accumulators of all passes kept): 32 x 256 with NP = 2 has gemm_b3_kernel's loop, wave tile and 38 KB.
template <int BM, int BN, int WGM, int WGN, int NP = 1>
__global__ __launch_bounds__(64 * WGM * WGN) void gemm_ln_b3_kernel(const float *__restrict__ X, int ldx, const uint16_t *__restrict__ Wp,
                                                                   int ldw, long long plane, const float *__restrict__ bias,
                                                                   const float *__restrict__ pre, const float *__restrict__ gamma,
                                                                   const float *__restrict__ beta, const float *__restrict__ post,
                                                                   float *__restrict__ out, int ldo, int R, int Cin, int act) {
    constexpr int BNP = BN / NP;                               // columns per pass
    constexpr int T = 64 * WGM * WGN, WM = BM / WGM, WN = BNP / WGN, MB = WM / 16, NB = WN / 16, LDC = BN + 4;
    constexpr int XF = BM * 8, PX = (XF + T - 1) / T;          // float4 groups of the X tile, per thread
    constexpr int WF = BNP * 4, PW = (WF + T - 1) / T;         // 16-byte groups of one W plane tile, per thread
    constexpr int SM0 = 3 * (BM + BNP) * B3_LD, SM1 = 2 * BM * LDC, SM = SM0 > SM1 ? SM0 : SM1;   // operand planes | staged output tile
    static_assert(MB >= 1 && NB >= 1 && BN % NP == 0 && WM % 16 == 0 && WN % 16 == 0, "tile shape");
    __shared__ __attribute__((aligned(16))) uint16_t smem[SM];
    uint16_t (*Xs)[BM][B3_LD] = reinterpret_cast<uint16_t (*)[BM][B3_LD]>(smem);
    uint16_t (*Ws)[BNP][B3_LD] = reinterpret_cast<uint16_t (*)[BNP][B3_LD]>(smem + 3 * BM * B3_LD);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w / WGN, wn = w % WGN;
    const int row0 = blockIdx.x * BM;
    const float *xp[PX];
    int xrow[PX], xk[PX];
#pragma unroll
    for (int p = 0; p < PX; ++p) {
        const int g = min(p * T + t, XF - 1);
        xrow[p] = g >> 3, xk[p] = (g & 7) * 4;
        xp[p] = X + (size_t)min(row0 + xrow[p], R - 1) * ldx + xk[p];
    }
    const uint16_t *wp[PW];
    int wrow[PW], wk[PW];
#pragma unroll
    for (int p = 0; p < PW; ++p) {
        const int g = min(p * T + t, WF - 1);
        wrow[p] = g >> 2, wk[p] = (g & 3) * 8;
        wp[p] = Wp + (size_t)wrow[p] * ldw + wk[p];
    }
    f32x4 xv[PX];
    u32x4 wv[PW][3];
#pragma unroll
    for (int p = 0; p < PX; ++p) xv[p] = *reinterpret_cast<const f32x4 *>(xp[p]);
#pragma unroll
    for (int p = 0; p < PW; ++p)
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) wv[p][pl] = *reinterpret_cast<const u32x4 *>(wp[p] + pl * plane);
    f32x4 acc[NP][MB][NB];
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[ps][i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int fr = lane & 15, fk = (lane >> 4) * 8;
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
    for (int k0 = 0; k0 < Cin; k0 += B3_KT) {
#pragma unroll
        for (int p = 0; p < PX; ++p) {
            if (XF % T == 0 || p * T + t < XF) {
                unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
                split3(xv[p][0], h0, m0, l0), split3(xv[p][1], h1, m1, l1), split3(xv[p][2], h2, m2, l2), split3(xv[p][3], h3, m3, l3);
                *reinterpret_cast<u32x2 *>(&Xs[0][xrow[p]][b3_col(xrow[p], xk[p])]) = u32x2{pack2(h0, h1), pack2(h2, h3)};
                *reinterpret_cast<u32x2 *>(&Xs[1][xrow[p]][b3_col(xrow[p], xk[p])]) = u32x2{pack2(m0, m1), pack2(m2, m3)};
                *reinterpret_cast<u32x2 *>(&Xs[2][xrow[p]][b3_col(xrow[p], xk[p])]) = u32x2{pack2(l0, l1), pack2(l2, l3)};
            }
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            if (WF % T == 0 || p * T + t < WF) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) *reinterpret_cast<u32x4 *>(&Ws[pl][wrow[p]][b3_col(wrow[p], wk[p])]) = wv[p][pl];
            }
        }
        __syncthreads();
        {
            // unconditional prefetch: the next K-tile of this pass, the first one of the next pass, or (the very last trip)
            // its own tile again
            const bool wrap = k0 + B3_KT >= Cin && ps + 1 < NP;
            const int kn = wrap ? 0 : min(k0 + B3_KT, Cin - B3_KT);
            const size_t wo = (size_t)(wrap ? ps + 1 : ps) * BNP * ldw + kn;
#pragma unroll
            for (int p = 0; p < PX; ++p) xv[p] = *reinterpret_cast<const f32x4 *>(xp[p] + kn);
#pragma unroll
            for (int p = 0; p < PW; ++p)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) wv[p][pl] = *reinterpret_cast<const u32x4 *>(wp[p] + pl * plane + wo);
        }
        bf16x8 a[3][NB], b[3][MB];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
            for (int j = 0; j < NB; ++j) a[pl][j] = *reinterpret_cast<const bf16x8 *>(&Ws[pl][wn * WN + j * 16 + fr][b3_col(fr, fk)]);
#pragma unroll
            for (int i = 0; i < MB; ++i) b[pl][i] = *reinterpret_cast<const bf16x8 *>(&Xs[pl][wm * WM + i * 16 + fr][b3_col(fr, fk)]);
        }
#define DPM_B3(PWQ, PXQ)                                                                                   \
    _Pragma("unroll") for (int i = 0; i < MB; ++i) _Pragma("unroll") for (int j = 0; j < NB; ++j)        \
        acc[ps][i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[PWQ][j], b[PXQ][i], acc[ps][i][j], 0, 0, 0)
        DPM_B3(1, 1);
        DPM_B3(2, 0);
        DPM_B3(0, 2);
        DPM_B3(1, 0);
        DPM_B3(0, 1);
        DPM_B3(0, 0);
#undef DPM_B3
        __syncthreads();
    }
    float *ct = reinterpret_cast<float *>(smem);
#pragma unroll
    for (int ps = 0; ps < NP; ++ps)
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int j = 0; j < NB; ++j)
                *reinterpret_cast<float4 *>(&ct[(wm * WM + i * 16 + (lane & 15)) * LDC + ps * BNP + wn * WN + j * 16 + (lane >> 4) * 4]) =
                    make_float4(acc[ps][i][j][0], acc[ps][i][j][1], acc[ps][i][j][2], acc[ps][i][j][3]);
    __syncthreads();
    // row-wise: G = BN / 4 lanes hold one row (a float4 each); the two LayerNorm sums are lane-group reductions, neighbours
    // first -- the association of layernorm_vec_kernel and gemm_ln_kernel (bit-identical rows in every form)
    constexpr int G = BN / 4, RPS = T / G;
    const int cr = t / G, cc = (t % G) * 4;
    const float4 bv = bias ? *reinterpret_cast<const float4 *>(bias + cc) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 g4 = *reinterpret_cast<const float4 *>(gamma + cc), b4 = *reinterpret_cast<const float4 *>(beta + cc);
    auto gsum = [](float v) {
#pragma unroll
        for (int off = 1; off < G; off <<= 1) v += __shfl_xor(v, off, 64);
        return v;
    };
#pragma unroll 4
    for (int p = 0; p < BM / RPS; ++p) {
        const int r = row0 + p * RPS + cr, rr = min(r, R - 1);
        float4 v = *reinterpret_cast<const float4 *>(&ct[(p * RPS + cr) * LDC + cc]);
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
        o.x = b3_act(o.x, act), o.y = b3_act(o.y, act), o.z = b3_act(o.z, act), o.w = b3_act(o.w, act);
        if (r < R) *reinterpret_cast<float4 *>(out + (size_t)r * ldo + cc) = o;
    }
}


// InvResMLP's point-wise pair (network/encoder/pointnext.py:118-138: pw_conv = Conv1d(C, 4C) -> LayerNorm -> ReLU -> Conv1d(4C, C)
// -> LayerNorm, then + residual and ReLU) as ONE kernel for C = 32 (the first level: 262 144 rows per 64-frame batch, where the
// two fused GEMM + LayerNorm kernels move 369 MB through HBM for 100 MB of input and output -- the 4C-wide intermediate is
// written and read back).  A wave owns 16 rows at a time and keeps the intermediate in REGISTERS:
//   h = relu(LN(x W1^T + b1)): x fragments straight from global memory (a row is 128 bytes, split in registers), W1 planes in
//       LDS (whole: 24 KB); the accumulators hold h[row = lane & 15][column 16 j + 4 g + q] (g = lane >> 4), whole rows per wave,
//       so the LayerNorm sums are two lane swaps;
//   y = LN(h W2^T + b2) + post, relu: the accumulator registers of column blocks 2 s, 2 s + 1 ARE the B operand of K-step s under
//       a permutation of k (slot 8 g + e <-> column 16 (2 s + (e >> 2)) + 4 g + (e & 3)), which the caller applies to W2's columns
//       once when it makes the planes (w2_planes_kperm): h is split in registers and never leaves them.
// Both products are bf16x3 (same arithmetic as gemm_b3_kernel; the row statistics are summed in another association than the
// two-kernel form's, so the results agree to rounding, not bit for bit).  The workgroup walks `tiles_per_wave` row tiles per wave
// with the weights resident in LDS.
#ifndef DPM_PW_GRID
#define DPM_PW_GRID 768
#endif
constexpr int PW_C = 32, PW_H = 128;
__device__ __forceinline__ float pw_rows4_sum(float v) {   // over the four lanes l, l+16, l+32, l+48 (decoder_ops.hip, rows4_sum)
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
__global__ __launch_bounds__(256) void pwconv_pair_b3_kernel(const float *__restrict__ X, int ldx, const uint16_t *__restrict__ W1p,
                                                            long long plane1, const float *__restrict__ b1,
                                                            const float *__restrict__ g1, const float *__restrict__ be1,
                                                            const uint16_t *__restrict__ W2p, long long plane2,
                                                            const float *__restrict__ b2, const float *__restrict__ g2,
                                                            const float *__restrict__ be2, const float *__restrict__ post,
                                                            float *__restrict__ out, int R, int tiles_per_wave) {
    __shared__ __attribute__((aligned(16))) uint16_t W1s[3][PW_H][PW_C];        // 24 KB, rows swizzled (b3_col)
    __shared__ __attribute__((aligned(16))) uint16_t W2s[3][PW_H / 32][PW_C][32];  // 24 KB: [plane][K-step][output column][slot]
    __shared__ __attribute__((aligned(16))) float vec1[3][PW_H];                 // b1 | gamma1 | beta1
    __shared__ __attribute__((aligned(16))) float vec2[3][PW_C];                 // b2 | gamma2 | beta2
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4, fr = lane & 15;
    // weights -> LDS: 16-byte pieces; W1 rows are [column][32 k], W2 rows (already K-permuted) [column][128 slots]
    for (int e = t; e < 3 * PW_H * 4; e += 256) {
        const int pl = e / (PW_H * 4), r = (e / 4) % PW_H, c = (e & 3) * 8;
        *reinterpret_cast<u32x4 *>(&W1s[pl][r][b3_col(r, c)]) = *reinterpret_cast<const u32x4 *>(W1p + pl * plane1 + r * PW_C + c);
    }
    for (int e = t; e < 3 * PW_C * 16; e += 256) {
        const int pl = e / (PW_C * 16), r = (e / 16) % PW_C, c = (e & 15) * 8;   // c: slot offset inside the row of 128
        *reinterpret_cast<u32x4 *>(&W2s[pl][c >> 5][r][b3_col(r, c & 31)]) =
            *reinterpret_cast<const u32x4 *>(W2p + pl * plane2 + r * PW_H + c);
    }
    if (t < PW_H) vec1[0][t] = b1 ? b1[t] : 0.f, vec1[1][t] = g1[t], vec1[2][t] = be1[t];
    if (t < PW_C) vec2[0][t] = b2 ? b2[t] : 0.f, vec2[1][t] = g2[t], vec2[2][t] = be2[t];
    __syncthreads();
    const long long tile0 = ((long long)blockIdx.x * 4 + w) * tiles_per_wave;
    // x fragment: this lane's 8 consecutive k of its row; the next tile's rows are requested while this one is computed
    f32x4 nx0, nx1, np0, np1;
    auto request = [&](long long r0) {
        const int r = (int)min(r0 + fr, (long long)R - 1);
        const float *xp = X + (size_t)r * ldx + 8 * g;
        nx0 = *reinterpret_cast<const f32x4 *>(xp), nx1 = *reinterpret_cast<const f32x4 *>(xp + 4);
        np0 = post ? *reinterpret_cast<const f32x4 *>(post + (size_t)r * PW_C + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
        np1 = post ? *reinterpret_cast<const f32x4 *>(post + (size_t)r * PW_C + 16 + 4 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    request(min(tile0 * 16, (long long)R - 1));
    for (int it = 0; it < tiles_per_wave; ++it) {
        const long long row0 = (tile0 + it) * 16;
        if (row0 >= R) break;
        const int row = (int)min(row0 + fr, (long long)R - 1);
        const f32x4 x0 = nx0, x1 = nx1, p0 = np0, p1 = np1;
        request(min(row0 + 16, (long long)R - 1));   // (clamped: the last trip re-reads a valid row)
        bf16x8 xb[3];
        {
            unsigned hh[8], mm[8], ll[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) split3(x0[e], hh[e], mm[e], ll[e]), split3(x1[e], hh[4 + e], mm[4 + e], ll[4 + e]);
            xb[0] = __builtin_bit_cast(bf16x8, u32x4{pack2(hh[0], hh[1]), pack2(hh[2], hh[3]), pack2(hh[4], hh[5]), pack2(hh[6], hh[7])});
            xb[1] = __builtin_bit_cast(bf16x8, u32x4{pack2(mm[0], mm[1]), pack2(mm[2], mm[3]), pack2(mm[4], mm[5]), pack2(mm[6], mm[7])});
            xb[2] = __builtin_bit_cast(bf16x8, u32x4{pack2(ll[0], ll[1]), pack2(ll[2], ll[3]), pack2(ll[4], ll[5]), pack2(ll[6], ll[7])});
        }
        // ---- h = x W1^T: 8 column blocks of 16, K = 32 = one instruction deep; two blocks at a time
        f32x4 h[PW_H / 16];
#pragma unroll
        for (int j = 0; j < PW_H / 16; j += 2) {
            bf16x8 a[2][3];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[jj][pl] = *reinterpret_cast<const bf16x8 *>(&W1s[pl][(j + jj) * 16 + fr][b3_col(fr, 8 * g)]);
            h[j] = h[j + 1] = f32x4{0.f, 0.f, 0.f, 0.f};
#define DPM_PW1(PWQ, PXQ)                                                                             \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                  \
        h[j + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[jj][PWQ], xb[PXQ], h[j + jj], 0, 0, 0)
            DPM_PW1(1, 1);
            DPM_PW1(2, 0);
            DPM_PW1(0, 2);
            DPM_PW1(1, 0);
            DPM_PW1(0, 1);
            DPM_PW1(0, 0);
#undef DPM_PW1
        }
        // + bias, LayerNorm over the row's 128 columns (32 here, the rest in the lanes 16 / 32 / 48 away), ReLU
        float sum = 0.f;
#pragma unroll
        for (int j = 0; j < PW_H / 16; ++j) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(&vec1[0][16 * j + 4 * g]);
#pragma unroll
            for (int q = 0; q < 4; ++q) h[j][q] += bv[q], sum += h[j][q];
        }
        const float mu = pw_rows4_sum(sum) * (1.f / PW_H);
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < PW_H / 16; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) h[j][q] -= mu, sq = fmaf(h[j][q], h[j][q], sq);
        const float rs = rsqrtf(pw_rows4_sum(sq) * (1.f / PW_H) + 1e-5f);
#pragma unroll
        for (int j = 0; j < PW_H / 16; ++j) {
            const f32x4 gv = *reinterpret_cast<const f32x4 *>(&vec1[1][16 * j + 4 * g]);
            const f32x4 ev = *reinterpret_cast<const f32x4 *>(&vec1[2][16 * j + 4 * g]);
#pragma unroll
            for (int q = 0; q < 4; ++q) h[j][q] = fmaxf(fmaf(h[j][q] * rs, gv[q], ev[q]), 0.f);
        }
        // ---- y = h W2^T: 2 column blocks, 4 K-steps; the B operand of K-step s = this lane's h values of blocks 2 s, 2 s + 1
        f32x4 y[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int s4 = 0; s4 < PW_H / 32; ++s4) {
            unsigned hh[8], mm[8], ll[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) split3(h[2 * s4 + (e >> 2)][e & 3], hh[e], mm[e], ll[e]);
            bf16x8 hb[3];
            hb[0] = __builtin_bit_cast(bf16x8, u32x4{pack2(hh[0], hh[1]), pack2(hh[2], hh[3]), pack2(hh[4], hh[5]), pack2(hh[6], hh[7])});
            hb[1] = __builtin_bit_cast(bf16x8, u32x4{pack2(mm[0], mm[1]), pack2(mm[2], mm[3]), pack2(mm[4], mm[5]), pack2(mm[6], mm[7])});
            hb[2] = __builtin_bit_cast(bf16x8, u32x4{pack2(ll[0], ll[1]), pack2(ll[2], ll[3]), pack2(ll[4], ll[5]), pack2(ll[6], ll[7])});
            bf16x8 a[2][3];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    a[jj][pl] = *reinterpret_cast<const bf16x8 *>(&W2s[pl][s4][jj * 16 + fr][b3_col(fr, 8 * g)]);
#define DPM_PW2(PWQ, PXQ)                                                                             \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj)                                                  \
        y[jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[jj][PWQ], hb[PXQ], y[jj], 0, 0, 0)
            DPM_PW2(1, 1);
            DPM_PW2(2, 0);
            DPM_PW2(0, 2);
            DPM_PW2(1, 0);
            DPM_PW2(0, 1);
            DPM_PW2(0, 0);
#undef DPM_PW2
        }
        // + bias, LayerNorm over 32 columns, + residual, ReLU; the lane owns columns 16 jj + 4 g .. + 3 of its row
        float s2 = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            const f32x4 bv = *reinterpret_cast<const f32x4 *>(&vec2[0][16 * jj + 4 * g]);
#pragma unroll
            for (int q = 0; q < 4; ++q) y[jj][q] += bv[q], s2 += y[jj][q];
        }
        const float mu2 = pw_rows4_sum(s2) * (1.f / PW_C);
        float q2 = 0.f;
#pragma unroll
        for (int jj = 0; jj < 2; ++jj)
#pragma unroll
            for (int q = 0; q < 4; ++q) y[jj][q] -= mu2, q2 = fmaf(y[jj][q], y[jj][q], q2);
        const float rs2 = rsqrtf(pw_rows4_sum(q2) * (1.f / PW_C) + 1e-5f);
        if (row0 + fr < R) {
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                const f32x4 gv = *reinterpret_cast<const f32x4 *>(&vec2[1][16 * jj + 4 * g]);
                const f32x4 ev = *reinterpret_cast<const f32x4 *>(&vec2[2][16 * jj + 4 * g]);
                const f32x4 pv = jj ? p1 : p0;
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = fmaxf(fmaf(y[jj][q] * rs2, gv[q], ev[q]) + pv[q], 0.f);
                *reinterpret_cast<f32x4 *>(out + (size_t)row * PW_C + 16 * jj + 4 * g) = o;
            }
        }
    }
}

}  // namespace

extern "C" int dpm_split_bf16x3(const float *W, long long n, void *planes, dpm_stream_t stream) {
    DPM_CHECK_ARG(W && planes && n >= 1);
    hipLaunchKernelGGL(split_bf16x3_kernel, dim3(dpm_cdiv(n, 256)), dim3(256), 0, (hipStream_t)stream, W, n, (uint16_t *)planes);
    return dpm_launch_status();
}

extern "C" int dpm_linear_bf16x3_rank3(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride,
                                       const float *bias, const float *residual, int ldr, float *out, int ldo, int R, int Cin,
                                       int Cout, int act, const float *x3, const float *w3, int ldw3, double scale,
                                       dpm_stream_t stream);

extern "C" int dpm_linear_bf16x3(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride, const float *bias,
                                 const float *residual, int ldr, float *out, int ldo, int R, int Cin, int Cout, int act,
                                 dpm_stream_t stream) {
    return dpm_linear_bf16x3_rank3(x, ldx, w_planes, ldw, plane_stride, bias, residual, ldr, out, ldo, R, Cin, Cout, act, nullptr,
                                   nullptr, 0, 0.0, stream);
}

// dpm_linear_bf16x3 plus a rank-3 term added in the epilogue: out[r, c] += scale * (x3[r, 0:3] . w3[c, 0:3]) (x3 (R,3) packed, w3
// rows ldw3 floats apart; fp32 multiply-adds in k order, before residual and activation).  x3 NULL: dpm_linear_bf16x3.
extern "C" int dpm_linear_bf16x3_rank3(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride,
                                       const float *bias, const float *residual, int ldr, float *out, int ldo, int R, int Cin,
                                       int Cout, int act, const float *x3, const float *w3, int ldw3, double scale,
                                       dpm_stream_t stream) {
    DPM_CHECK_ARG((x3 == nullptr) == (w3 == nullptr) && (!x3 || ldw3 >= 3));
    const float r3s = (float)scale;
    DPM_CHECK_ARG(x && w_planes && out && R >= 1 && Cin >= 1 && Cout >= 1 && ldx >= Cin && ldw >= Cin && ldo >= Cout);
    DPM_CHECK_ARG(act >= DPM_ACT_NONE && act <= DPM_ACT_SIGMOID && (!residual || ldr >= Cout) && plane_stride >= (long long)Cout * ldw);
    auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    if (Cin % B3_KT != 0 || Cout % 4 != 0 || ldw % 8 != 0 || plane_stride % 8 != 0 || ldo % 4 != 0 || !al(w_planes) || !al(out) ||
        !al(bias) || (residual && (!al(residual) || ldr % 4 != 0)))
        return DPM_EUNSUPPORTED;
    const bool xvec = ldx % 4 == 0 && al(x);
    // tile choice as in dpm_linear: 64 x 64 from 192 such tiles on (or tall problems), 32 x 32 below -- the bits do not depend on it
    const long long big = (long long)dpm_cdiv(R, 64) * dpm_cdiv(Cout, 64);
#define DPM_B3_LAUNCH(TM, TN, V, PF)                                                                                          \
    hipLaunchKernelGGL((gemm_b3_kernel<TM, TN, V, PF>), dim3(dpm_cdiv(Cout, TN), dpm_cdiv(R, TM)), dim3(256), 0, (hipStream_t)stream, x, ldx, \
                       (const uint16_t *)w_planes, ldw, plane_stride, bias, residual, ldr, out, ldo, R, Cin, Cout, act, x3, w3, ldw3, r3s)
    if (DPM_B3_WIDE && xvec && Cout % 128 == 0 && (long long)dpm_cdiv(R, 64) * (Cout / 128) >= DPM_B3_WIDE) {
        DPM_B3_LAUNCH(64, 128, true, 1);
    } else if (big >= 192 || (R > 1024 && Cout > 32)) {
        const bool small = big <= DPM_B3_DEEP;   // few workgroups per CU: K-tiles four ahead
        if (xvec && small) DPM_B3_LAUNCH(64, 64, true, 4);
        else if (xvec) DPM_B3_LAUNCH(64, 64, true, 1);
        else if (small) DPM_B3_LAUNCH(64, 64, false, 4);
        else DPM_B3_LAUNCH(64, 64, false, 1);
    } else {
        const bool small = (long long)dpm_cdiv(R, 32) * dpm_cdiv(Cout, 32) <= 2 * DPM_B3_DEEP;
        if (xvec && small) DPM_B3_LAUNCH(32, 32, true, 4);
        else if (xvec) DPM_B3_LAUNCH(32, 32, true, 1);
        else if (small) DPM_B3_LAUNCH(32, 32, false, 4);
        else DPM_B3_LAUNCH(32, 32, false, 1);
    }
#undef DPM_B3_LAUNCH
    return dpm_launch_status();
}

// dpm_linear_bf16x3 for a q | k | v projection whose K and V go straight to the attention kernel (KvPlanes above): columns
// [0, kv_col0) are written to `out` as fp32 rows, columns [kv_col0, Cout) = K (heads x 32) then V (heads x 32) are written to
// kv_planes as the per-(sequence, head, 64-key tile) images of dpm_attention_planes -- sequence = row / tokens.  Needs
// tokens % 64 == 0, R % tokens == 0, kv_col0 % 64 == 0, Cout - kv_col0 == 64 heads, 16-byte aligned operands; no residual, no
// activation.  DPM_EUNSUPPORTED otherwise (the caller runs dpm_linear_bf16x3 + dpm_attention_*: same results bit for bit).
extern "C" int dpm_linear_bf16x3_kvplanes(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride,
                                          const float *bias, float *out, int ldo, int R, int Cin, int Cout, int kv_col0, int tokens,
                                          int heads, void *kv_planes, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && w_planes && out && kv_planes && R >= 1 && Cin >= 1 && Cout >= 1 && ldx >= Cin && ldw >= Cin && heads >= 1);
    DPM_CHECK_ARG(kv_col0 >= 0 && kv_col0 < Cout && ldo >= kv_col0 && tokens >= 64 && plane_stride >= (long long)Cout * ldw);
    auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    if (Cin % B3_KT != 0 || ldw % 8 != 0 || plane_stride % 8 != 0 || ldo % 4 != 0 || !al(w_planes) || !al(out) || !al(bias) || !al(kv_planes) ||
        ldx % 4 != 0 || !al(x) || tokens % 64 != 0 || R % tokens != 0 || kv_col0 % 64 != 0 || Cout - kv_col0 != 64 * heads ||
        heads % 2 != 0)   // (a 64-column tile must not straddle the K | V boundary)
        return DPM_EUNSUPPORTED;
    const KvPlanes kv{(uint16_t *)kv_planes, kv_col0, tokens, heads};
#define DPM_B3_KV(TN, PF)                                                                                                       \
    hipLaunchKernelGGL((gemm_b3_kvp_kernel<TN, PF>), dim3(dpm_cdiv(Cout, TN), R / 64), dim3(256), 0, (hipStream_t)stream, x, ldx,   \
                       (const uint16_t *)w_planes, ldw, plane_stride, bias, out, ldo, R, Cin, Cout, kv)
    // 64-row tiles always (a tile is a key tile); the width by dpm_linear_bf16x3's rule -- the bits do not depend on it
    if (DPM_B3_WIDE && Cout % 128 == 0 && kv_col0 % 128 == 0 && heads % 4 == 0 && (long long)(R / 64) * (Cout / 128) >= DPM_B3_WIDE) DPM_B3_KV(128, 1);
    else if ((long long)(R / 64) * dpm_cdiv(Cout, 64) <= DPM_B3_DEEP) DPM_B3_KV(64, 4);
    else DPM_B3_KV(64, 1);
#undef DPM_B3_KV
    return dpm_launch_status();
}

// dpm_linear_layernorm on the bf16x3 product: same contract (pre / post packed (R, Cout), Cout in {32, 64, 128, 256}), weights as
// the planes of dpm_split_bf16x3.  DPM_EUNSUPPORTED for other widths, Cin % 32 != 0 or unaligned operands: the caller then runs
// dpm_linear_bf16x3 + dpm_layernorm (identical rows).
extern "C" int dpm_linear_layernorm_bf16x3(const float *x, int ldx, const void *w_planes, int ldw, long long plane_stride,
                                           const float *bias, const float *pre, const float *gamma, const float *beta,
                                           const float *post, float *out, int ldo, int R, int Cin, int Cout, int act,
                                           dpm_stream_t stream) {
    DPM_CHECK_ARG(x && w_planes && gamma && beta && out && R >= 1 && Cin >= 1 && Cout >= 1 && ldx >= Cin && ldw >= Cin && ldo >= Cout);
    DPM_CHECK_ARG(act >= DPM_ACT_NONE && act <= DPM_ACT_SIGMOID && plane_stride >= (long long)Cout * ldw);
    auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    if (Cin % B3_KT != 0 || ldx % 4 != 0 || ldw % 8 != 0 || plane_stride % 8 != 0 || ldo % 4 != 0 || !al(x) || !al(w_planes) ||
        !al(bias) || !al(pre) || !al(gamma) || !al(beta) || !al(post) || !al(out))
        return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
#define DPM_GLN3(BM, BN, WGM, WGN, NP)                                                                                         \
    hipLaunchKernelGGL((gemm_ln_b3_kernel<BM, BN, WGM, WGN, NP>), dim3(dpm_cdiv(R, BM)), dim3(64 * WGM * WGN), 0, st, x, ldx,       \
                       (const uint16_t *)w_planes, ldw, plane_stride, bias, pre, gamma, beta, post, out, ldo, R, Cin, act)
    // every configuration gives the same bits (same instructions in the same order per output element).  256 columns: 64 rows
    // in one pass, 512 threads (wave tile 32 x 64, X split once).  History: with the first version's padded LDS rows this form
    // was faster alone (44 against 59 us) and made the pipelined step LONGER, so 32 rows x two passes of 128 columns (256
    // threads, 38 KB) shipped for a while; with the swizzled rows (half the LDS cycles) the one-pass form wins both ways:
    // 4.19-4.20 against 4.27 ms per step.
    if (Cout == 256) {
        if (dpm_knob("DPM_GLN3_TWOPASS", 0)) DPM_GLN3(32, 256, 1, 4, 2);
        else DPM_GLN3(64, 256, 2, 4, 1);
    } else if (Cout == 128) DPM_GLN3(64, 128, 2, 2, 1);
    else if (Cout == 64) DPM_GLN3(64, 64, 2, 2, 1);
    else if (Cout == 32) DPM_GLN3(64, 32, 2, 2, 1);
    else return DPM_EUNSUPPORTED;
#undef DPM_GLN3
    return dpm_launch_status();
}

// InvResMLP's pw_conv pair in one kernel (pwconv_pair_b3_kernel): out = relu(LN2(relu(LN1(x W1^T + b1)) W2^T + b2) + post),
// x (R, 32) with row stride ldx, W1 (128, 32) and W2 (32, 128) as bf16x3 planes (dpm_split_bf16x3), W2's columns PERMUTED
// before the split: stored column 32 s + 8 g + e holds original column 16 (2 s + (e >> 2)) + 4 g + (e & 3).  post / out packed
// (R, 32).  DPM_EUNSUPPORTED for other widths or unaligned operands.
extern "C" int dpm_pwconv_pair_bf16x3(const float *x, int ldx, const void *w1_planes, long long plane1, const float *b1,
                                      const float *g1, const float *be1, const void *w2_planes_kperm, long long plane2,
                                      const float *b2, const float *g2, const float *be2, const float *post, float *out, int R,
                                      int C, int H, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && w1_planes && w2_planes_kperm && g1 && be1 && g2 && be2 && out && R >= 1 && ldx >= C);
    auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    if (C != PW_C || H != PW_H || ldx % 4 != 0 || plane1 % 8 != 0 || plane2 % 8 != 0 || !al(x) || !al(w1_planes) ||
        !al(w2_planes_kperm) || !al(post) || !al(out))
        return DPM_EUNSUPPORTED;
    const long long tiles = ((long long)R + 15) / 16;
    // the weights (48 KB) are staged once per workgroup: enough row tiles per wave to pay for it, enough workgroups to fill the chip
    // (one round of three resident workgroups per CU when there are rows enough; never fewer than 2 tiles per wave's worth)
    const int tpw = (int)std::max<long long>(1, (tiles + 4LL * DPM_PW_GRID - 1) / (4LL * DPM_PW_GRID));
    const unsigned grid = (unsigned)((tiles + 4LL * tpw - 1) / (4LL * tpw));
    hipLaunchKernelGGL(pwconv_pair_b3_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, x, ldx, (const uint16_t *)w1_planes,
                       plane1, b1, g1, be1, (const uint16_t *)w2_planes_kperm, plane2, b2, g2, be2, post, out, R, tpw);
    return dpm_launch_status();
}
