// Decoder kernels: sine position embedding, multi-head attention core, L2 row normalisation,
// dual-softmax pairing + top-k, pair gather, correspondence sets + iterative weighted Kabsch
// (fp64 3x3 SVD on device), row mean.  fp32 like the reference unless stated.
// Compiled with -ffp-contract=off: every fused multiply-add is an explicit fmaf().
#include "dpm_common.h"
#include "topk_emulate.h"
#include <algorithm>

namespace {

// ------------------------------------------------------------------------------------------
// a11 PositionEmbeddingCoordsSine (descriptor_attention.py:66-83).  dim_t (F floats) is the
// reference's own table temperature**(2*(i//2)/F), computed once on the host with the same
// torch expression, so p*pi/dim_t is bit-identical; channel axis*F+i = sin (even i) / cos (odd i).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void posemb_kernel(const float *__restrict__ xyz, int ld,
                                                     const float *__restrict__ dim_t, int F, int E, int R,
                                                     float scale, float *__restrict__ out) {
    // R * E < 2^32 (checked by the launcher): 32-bit index arithmetic -- the two 64-bit divisions per element were
    // most of this kernel
    const unsigned e = blockIdx.x * 256u + threadIdx.x;
    if (e >= (unsigned)R * (unsigned)E) return;
    const unsigned r = e / (unsigned)E, c = e - r * (unsigned)E;
    float v = 0.f;
    if (c < 3u * (unsigned)F) {
        const unsigned a = c / (unsigned)F, i = c - a * (unsigned)F;
        const float ang = (xyz[(size_t)r * ld + a] * scale) / dim_t[i];
        v = (i & 1) ? cosf(ang) : sinf(ang);
    }
    out[e] = v;
}

// ------------------------------------------------------------------------------------------
// a12 attention core on the matrix cores: out[b,m,h*32:(h+1)*32] = softmax(q k^T / sqrt(32)) v, head dim 32.
// Flash-attention structure in exact fp32 (v_mfma_f32_16x16x4_f32): one workgroup = 64 queries of one
// (batch, head), one wave = 16 queries.  Per 64-key tile: S = Q K^T (Q fragments stay in registers; K tile
// in LDS as the B operand), online softmax on the C-layout registers (row statistics via DPP row
// reductions), P transposed through a per-wave LDS tile into the A-operand layout, O += P V.
// ------------------------------------------------------------------------------------------
constexpr int HD = 32;
constexpr float LOG2E = 1.44269504088896340736f;
// workgroups of 128 queries a launch must have for the 32-queries-per-wave form (QT = 2); 0 = never.  Round 5: with P V on the
// bf16 pipe that form holds 204 registers (two waves per SIMD) against 128 at QT = 1; it is as fast alone (67.4 against 66.5 us at
// 128 sequences) and costs the pipelined step what the P V change wins (4.16 / 4.18 against 4.10 / 4.10 ms): never.
#ifndef DPM_ATT_WIDE_MIN
#define DPM_ATT_WIDE_MIN 0
#endif
#ifndef DPM_ATT_BLOCK8
#define DPM_ATT_BLOCK8 1
#endif
#ifndef DPM_ATT_BLOCK8_MIN_M
#define DPM_ATT_BLOCK8_MIN_M 1024
#endif
constexpr int RES_HDR = 20;  // floats before the inlier-confidence list in a Kabsch `result`
using f32x4 = __attribute__((ext_vector_type(4))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
using u32x2 = __attribute__((ext_vector_type(2))) unsigned;

// max / sum over the 16 lanes of a DPP row (lanes sharing lane>>4).  Written out as one DPP-operand instruction
// per step: from update_dpp + fmaxf the compiler builds copy + s_nop + mov_dpp + canonicalise + max.  s_nop 1 = the
// two wait states a DPP operand needs after the VALU write of its register.
#define DPM_ROW16(op)                                                                                   \
    "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"                      \
    "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"                      \
    "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"                          \
    "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\ts_nop 0"
__device__ __forceinline__ float row16_max(float v) {
    asm(DPM_ROW16("v_max_f32_dpp") : "+v"(v));
    return v;
}
__device__ __forceinline__ float row16_sum(float v) {
    asm(DPM_ROW16("v_add_f32_dpp") : "+v"(v));
    return v;
}
#undef DPM_ROW16

// max / sum over the four lanes l, l+16, l+32, l+48 (one value per DPP row), result in all four: the gfx950 lane
// swaps exchange half-waves (v_permlane32_swap) and odd/even rows (v_permlane16_swap) in one VALU op each.
__device__ __forceinline__ float rows4_max(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}
__device__ __forceinline__ float rows4_sum(float v) {
    auto a = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    auto b = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}

// Flash-attention structure on the matrix pipe with the TRANSPOSED score tile: S^T = K Q^T, so that the C-layout registers
// of a score block (lane l: keys 4(l>>4)+q of the block, query l&15) are already the B operand of O^T = V^T P^T under
// a k-remap (K-step h contracts, in slot 8 g + e of lane group g, the key 16 (2 h + (e >> 2)) + 4 g + (e & 3) on both operands).
// Both products are exact bf16x3 products (gemm_b3.hip has the arithmetic; the scores since round 4, P V since round 5:
// 128 sequences of 256 tokens 79.9 -> 67.4 us, 2 x 4096 x 4096 322 -> 269 us, error against fp64 4.2e-7 against 6.2e-7).  The
// probabilities never leave the registers (no LDS round trip to re-shape P), a lane owns ONE query -- its running
// max / sum are scalars, the max needs two lane swaps instead of a 16-lane reduction, the row sum meets once at the
// end -- and it ends up with 2 x 4 consecutive output channels of that query: two 16-byte stores.
// QT = query groups of 16 per wave: with two, every K / V fragment read from LDS feeds two MFMAs and a block covers
// 128 queries, so the fetch of the first K / V tile (a block lives for only N / 64 tiles) is paid half as often.
// MASK: key_mask (batch, N) bytes, non-zero = the key is padding (nn.MultiheadAttention's key_padding_mask): its score is
// -inf like a key beyond N.  A sequence whose keys are ALL masked gives NaN rows, as in the reference.
// SPLIT: few queries against many keys (a scan's 256 tokens attending a 4096-token map tile: 32 blocks walking 64 key
// tiles each).  The keys are cut into `nsplit` ranges of whole tiles, gridDim.x = query tiles x nsplit; a block writes its
// range's UNNORMALISED output rows (relative to its own running max) into part_o (nsplit, B, M, heads*HD) and
// (max, sum) into part_ml (nsplit, B, heads, M, 2); attention_merge_kernel rescales and adds the ranges.
template <bool VEC, int QT, bool MASK = false, bool SPLIT = false, int NWV = 4, bool PRE = false>
#ifndef DPM_ATT_WAVES
#define DPM_ATT_WAVES 0
#endif
#if DPM_ATT_WAVES
#define DPM_ATT_OCC __attribute__((amdgpu_waves_per_eu(DPM_ATT_WAVES, DPM_ATT_WAVES)))
#else
#define DPM_ATT_OCC
#endif
__global__ __launch_bounds__(64 * NWV) DPM_ATT_OCC void attention_kernel(const float *__restrict__ Q, int ldq, long long sq,
                                                        const float *__restrict__ Kp, int ldk, long long sk,
                                                        const float *__restrict__ V, int ldv, long long sv,
                                                        float *__restrict__ O, int ldo, long long so, int M,
                                                        int N, float scale, int kv_shift,
                                                        const uint8_t *__restrict__ key_mask = nullptr, int nsplit = 1,
                                                        float *__restrict__ part_o = nullptr,
                                                        float *__restrict__ part_ml = nullptr,
                                                        const int32_t *__restrict__ seq = nullptr,
                                                        const uint16_t *__restrict__ kvp = nullptr) {
    // PRE (round 5): the keys and values arrive as the bf16 planes this kernel would otherwise make of them, one 24 KB image
    // per (stored sequence, head, 64-key tile) in exactly the layout of Ks3 | Vt3 below -- written once by the q | k | v
    // projection's epilogue (gemm_b3.hip, KvPlanes) instead of being split by every query block that reads the tile: staging is
    // a straight copy (N % 64 == 0; Kp / V are not read).
    constexpr int TK = 64;                 // keys per tile
    // A operand of S^T = K Q^T, which runs as an exact bf16x3 product (round 4; gemm_b3.hip has the arithmetic: both operands
    // split into three bf16 terms, six term products per score, fp32 accumulate -- 3/8 of the exact-fp32 instruction's
    // matrix-pipe time at the same accuracy): three bf16 planes of the K tile in the swizzled 64-byte rows of b3_col (dpm_common.h:
    // conflict-free 16-byte fragment reads), A[i=key][k=d] = Ks3[plane][key][b3_col(key, d)].  HD = 32 is ONE instruction deep.
    constexpr int KLD = HD;
    static_assert(HD == 32, "the score product is one v_mfma_f32_16x16x32_bf16 deep");
    __shared__ __attribute__((aligned(16))) uint16_t Ks3[3][TK][KLD];
    // A operand of O^T = V^T P^T, which since round 5 runs as an exact bf16x3 product as well (the probabilities are split in
    // registers, V while it is staged): three bf16 planes of the V tile TRANSPOSED, Vt3[plane][d][slot], 64 slots = the tile's
    // keys in the order the score registers hold them: slot 32 h + 8 g + e <-> key 16 (2 h + (e >> 2)) + 4 g + (e & 3), so that
    // a lane's eight consecutive slots (one 16-byte fragment read) are exactly the keys whose probabilities it holds for K-step h.
    // Rows of 128 bytes, the eight 16-byte chunks of row d stored at chunk ^ ((d >> 1) & 7): the 16 lanes of every ds_read_b128
    // group hit 16 different bank groups.
    __shared__ __attribute__((aligned(16))) uint16_t Vt3[3][HD][TK];
    auto vt_col = [](int d, int slot) { return ((((slot >> 3) ^ (d >> 1)) & 7) << 3) | (slot & 7); };
    // all query tiles and heads of one batch element on one XCD: its K / V rows are fetched into that L2 once
    const unsigned bid = xcd_chunked_id((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x,
                                        gridDim.x * gridDim.y * gridDim.z);
    const int b = bid / (gridDim.x * gridDim.y), h = (bid / gridDim.x) % gridDim.y;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, g = lane >> 4;
    const int qtiles = SPLIT ? gridDim.x / nsplit : gridDim.x, split = SPLIT ? (bid % gridDim.x) / qtiles : 0;
    const int q0 = ((bid % gridDim.x) % qtiles) * (16 * NWV * QT) + w * (16 * QT);
    // key range of this block: whole tiles, the same count for every range but the last
    const int chunk = SPLIT ? ((N + TK * nsplit - 1) / (TK * nsplit)) * TK : N;
    const int n_begin = split * chunk, n_end = SPLIT ? min(N, n_begin + chunk) : N;
    // batch element b reads the keys / values of element (b + kv_shift) mod batch: with the source and target tokens
    // of B pairs stacked as 2B sequences and kv_shift = B, ONE launch is both directions of a cross attention
    int bk = b + kv_shift;
    if (bk >= (int)gridDim.z) bk -= (int)gridDim.z;
    // seq: the batch elements are DRAWN from a smaller set of stored sequences (consecutive-frame pairs use every frame
    // twice): element b's queries are stored sequence seq[b], its keys / values stored sequence seq[bk]; the output and the
    // key mask stay per batch element
    const int bq = seq ? seq[b] : b;
    const float *Qb = Q + (size_t)bq * sq + h * HD;
    const uint8_t *km = MASK ? key_mask + (size_t)bk * N : nullptr;
    if (seq) bk = seq[bk];
    const float *Kb = Kp + (size_t)bk * sk + h * HD;
    const float *Vb = V + (size_t)bk * sv + h * HD;
    constexpr int IMG = 2 * 3 * TK * HD;   // uint16 per (sequence, head, tile): K planes, then the transposed V planes
    const uint16_t *img = PRE ? kvp + ((size_t)bk * gridDim.y + h) * (size_t)(N / TK) * IMG : nullptr;

    // Q^T fragments (B operand: B[k = 8 g + e][j = lane&15] = Q[query lane&15][d = 8 g + e]), pre-scaled, split once
    bf16x8 qb[QT][3];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qr = min(q0 + 16 * u + (lane & 15), M - 1);
        const float *qp = Qb + (size_t)qr * ldq + 8 * g;
        unsigned hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3(qp[e] * scale, hh[e], mm[e], ll[e]);
        qb[u][0] = __builtin_bit_cast(bf16x8, u32x4{pack2(hh[0], hh[1]), pack2(hh[2], hh[3]), pack2(hh[4], hh[5]), pack2(hh[6], hh[7])});
        qb[u][1] = __builtin_bit_cast(bf16x8, u32x4{pack2(mm[0], mm[1]), pack2(mm[2], mm[3]), pack2(mm[4], mm[5]), pack2(mm[6], mm[7])});
        qb[u][2] = __builtin_bit_cast(bf16x8, u32x4{pack2(ll[0], ll[1]), pack2(ll[2], ll[3]), pack2(ll[4], ll[5]), pack2(ll[6], ll[7])});
    }
    f32x4 oacc[QT][2];  // O^T[d = 16 jd + 4 g + q][query 16 u + lane&15]
    float mrow[QT], lrow[QT];  // running max / this lane's share of the sum, query 16 u + lane&15
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        oacc[u][0] = oacc[u][1] = f32x4{0.f, 0.f, 0.f, 0.f};
        mrow[u] = -__builtin_inff(), lrow[u] = 0.f;
    }

    // K / V tiles: 64 rows x 32 floats each = 512 float4 per matrix, 2 per thread.  The next tile is fetched into
    // registers (branch-free: rows beyond N re-read the last key; their scores are masked to -inf below, so the
    // probabilities that multiply those V rows are exactly 0) while the current one feeds the MFMAs.
    constexpr int PT = 512 / (64 * NWV);  // float4 per thread and matrix
    constexpr int PC = PRE ? IMG / 8 / (64 * NWV) : 1;   // 16-byte pieces of an image per thread
    float4 kreg[PRE ? 1 : PT], vreg[PRE ? 1 : PT];
    u32x4 preg[PC];
    auto fetch = [&](int n0) {
        if (PRE) {
            const uint16_t *src = img + (size_t)(n0 / TK) * IMG;
#pragma unroll
            for (int i = 0; i < PC; ++i) preg[i] = *reinterpret_cast<const u32x4 *>(src + (size_t)(t + i * (64 * NWV)) * 8);
            return;
        }
#pragma unroll
        for (int p = 0; p < PT; ++p) {
            const int e = t + p * (64 * NWV), kr = min(n0 + (e >> 3), N - 1), c4 = (e & 7) * 4;
            const float *kp = Kb + (size_t)kr * ldk + c4, *vp = Vb + (size_t)kr * ldv + c4;
            if (VEC) {
                kreg[p] = *reinterpret_cast<const float4 *>(kp), vreg[p] = *reinterpret_cast<const float4 *>(vp);
            } else {
                kreg[p] = make_float4(kp[0], kp[1], kp[2], kp[3]), vreg[p] = make_float4(vp[0], vp[1], vp[2], vp[3]);
            }
        }
    };
    fetch(n_begin);
    for (int n0 = n_begin; n0 < n_end; n0 += TK) {
        __syncthreads();
        if (PRE) {
#pragma unroll
            for (int i = 0; i < PC; ++i) {
                const int c = t + i * (64 * NWV);   // piece of the image: the K planes are its first half
                uint16_t *dst = c < IMG / 16 ? &Ks3[0][0][0] + c * 8 : &Vt3[0][0][0] + (c - IMG / 16) * 8;
                *reinterpret_cast<u32x4 *>(dst) = preg[i];
            }
        }
#pragma unroll
        for (int p = 0; p < (PRE ? 0 : PT); ++p) {
            const int e = t + p * (64 * NWV), kr = e >> 3, c4 = (e & 7) * 4;
            unsigned h0, m0, l0, h1, m1, l1, h2, m2, l2, h3, m3, l3;
            split3(kreg[p].x, h0, m0, l0), split3(kreg[p].y, h1, m1, l1), split3(kreg[p].z, h2, m2, l2), split3(kreg[p].w, h3, m3, l3);
            *reinterpret_cast<u32x2 *>(&Ks3[0][kr][b3_col(kr, c4)]) = u32x2{pack2(h0, h1), pack2(h2, h3)};
            *reinterpret_cast<u32x2 *>(&Ks3[1][kr][b3_col(kr, c4)]) = u32x2{pack2(m0, m1), pack2(m2, m3)};
            *reinterpret_cast<u32x2 *>(&Ks3[2][kr][b3_col(kr, c4)]) = u32x2{pack2(l0, l1), pack2(l2, l3)};
            // V: split and scattered into the transposed planes (four 2-byte stores per plane)
            const int vj = kr >> 4, slot = 32 * (vj >> 1) + 8 * ((kr >> 2) & 3) + 4 * (vj & 1) + (kr & 3);
            const float vv[4] = {vreg[p].x, vreg[p].y, vreg[p].z, vreg[p].w};
#pragma unroll
            for (int dd = 0; dd < 4; ++dd) {
                unsigned vh, vm, vl;
                split3(vv[dd], vh, vm, vl);
                const int d = c4 + dd, col = vt_col(d, slot);
                Vt3[0][d][col] = (uint16_t)(vh >> 16), Vt3[1][d][col] = (uint16_t)(vm >> 16), Vt3[2][d][col] = (uint16_t)(vl >> 16);
            }
        }
        __syncthreads();
        if (n0 + TK < n_end) fetch(n0 + TK);
        // S^T tile: 64 keys x 16 queries = 4 MFMA blocks (16 keys each), each six bf16 instructions (smallest terms first;
        // (plane of K, plane of Q): (1,1) (2,0) (0,2) (1,0) (0,1) (0,0)); two key blocks at a time, so that 2 QT independent
        // accumulators lie between two instructions on the same one
        f32x4 sacc[QT][4];
#pragma unroll
        for (int u = 0; u < QT; ++u)
#pragma unroll
            for (int j = 0; j < 4; ++j) sacc[u][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int jp = 0; jp < 4; jp += 2) {
            bf16x8 ka[2][3];
#pragma unroll
            for (int jj = 0; jj < 2; ++jj)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl)
                    ka[jj][pl] = *reinterpret_cast<const bf16x8 *>(&Ks3[pl][(jp + jj) * 16 + (lane & 15)][b3_col(lane & 15, 8 * g)]);
#define DPM_S3(PK, PQ)                                                                                              \
    _Pragma("unroll") for (int jj = 0; jj < 2; ++jj) _Pragma("unroll") for (int u = 0; u < QT; ++u)                \
        sacc[u][jp + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka[jj][PK], qb[u][PQ], sacc[u][jp + jj], 0, 0, 0), mfma_pace()
            DPM_S3(1, 1);
            DPM_S3(2, 0);
            DPM_S3(0, 2);
            DPM_S3(1, 0);
            DPM_S3(0, 1);
            DPM_S3(0, 0);
#undef DPM_S3
        }
        // sacc[u][j][q] = score of key n0 + 16 j + 4 g + q for query 16 u + lane&15: mask keys beyond N, online softmax.
        // Only a ragged last tile is tested against N (uniform branch: a compare + select per score saved on whole tiles).
        if (n0 + TK > N || MASK) {
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int j = 0; j < 4; ++j)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int n = n0 + j * 16 + g * 4 + q;
                        if (n >= N || (MASK && km[min(n, N - 1)])) sacc[u][j][q] = -__builtin_inff();
                    }
        }
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            float mx = -__builtin_inff();
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) mx = fmaxf(mx, sacc[u][j][q]);
            mx = rows4_max(mx);
            const float nm = fmaxf(mrow[u], mx);
            // exp(-inf) = 0 on the first tile; with masks a whole tile may be padding while the running max is still -inf:
            // nothing has been accumulated then, the factor is irrelevant (and -inf - -inf would be NaN).
            // e^(s - nm) as ONE multiply-add into v_exp_f32 (2^(s log2 e - nm log2 e)) instead of subtract, multiply, v_exp_f32
            const bool dead = MASK && nm == -__builtin_inff();
            const float nml = nm * LOG2E;
            const float corr = dead ? 1.f : __builtin_amdgcn_exp2f(fmaf(mrow[u], LOG2E, -nml));
            float ps = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float pv = dead ? 0.f : __builtin_amdgcn_exp2f(fmaf(sacc[u][j][q], LOG2E, -nml));
                    sacc[u][j][q] = pv;
                    ps += pv;
                }
            lrow[u] = lrow[u] * corr + ps;
            mrow[u] = nm;
#pragma unroll
            for (int q = 0; q < 4; ++q) oacc[u][0][q] *= corr, oacc[u][1][q] *= corr;
        }
        // O^T += V^T P^T as two K-steps of 32 keys: the probabilities of key blocks 2 h, 2 h + 1 (this lane's eight registers)
        // are split into three bf16 planes in place -- they ARE the B operand's slots 8 g .. 8 g + 7 -- and meet the V^T planes
        // in six bf16 instructions per output block (smallest terms first; (plane of V, plane of P)); 2 QT independent
        // accumulators lie between two instructions on the same one.  24 instructions of 16 cycles per 16 queries and key tile
        // instead of 32 fp32 ones of 32.
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            bf16x8 pb[QT][3], va[2][3];
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                unsigned hh[8], mm[8], ll[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) split3(sacc[u][2 * h + (e >> 2)][e & 3], hh[e], mm[e], ll[e]);
                pb[u][0] = __builtin_bit_cast(bf16x8, u32x4{pack2(hh[0], hh[1]), pack2(hh[2], hh[3]), pack2(hh[4], hh[5]), pack2(hh[6], hh[7])});
                pb[u][1] = __builtin_bit_cast(bf16x8, u32x4{pack2(mm[0], mm[1]), pack2(mm[2], mm[3]), pack2(mm[4], mm[5]), pack2(mm[6], mm[7])});
                pb[u][2] = __builtin_bit_cast(bf16x8, u32x4{pack2(ll[0], ll[1]), pack2(ll[2], ll[3]), pack2(ll[4], ll[5]), pack2(ll[6], ll[7])});
            }
#pragma unroll
            for (int jd = 0; jd < 2; ++jd)
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    const int d = jd * 16 + (lane & 15);
                    va[jd][pl] = *reinterpret_cast<const bf16x8 *>(&Vt3[pl][d][vt_col(d, 32 * h + 8 * g)]);
                }
#define DPM_PV3(PVQ, PPQ)                                                                                           \
    _Pragma("unroll") for (int jd = 0; jd < 2; ++jd) _Pragma("unroll") for (int u = 0; u < QT; ++u)                \
        oacc[u][jd] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(va[jd][PVQ], pb[u][PPQ], oacc[u][jd], 0, 0, 0), mfma_pace()
            DPM_PV3(1, 1);
            DPM_PV3(2, 0);
            DPM_PV3(0, 2);
            DPM_PV3(1, 0);
            DPM_PV3(0, 1);
            DPM_PV3(0, 0);
#undef DPM_PV3
        }
    }
    // lane: queries q0 + 16 u + (lane&15), channels 16 jd + 4 g .. +3
    if (SPLIT) {
        const int B = gridDim.z, E = gridDim.y * HD;
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const float l = rows4_sum(lrow[u]);
            const int m = q0 + 16 * u + (lane & 15);
            if (m >= M) continue;
            float *o = part_o + (((size_t)split * B + b) * M + m) * E + h * HD + 4 * g;
#pragma unroll
            for (int jd = 0; jd < 2; ++jd)
                *reinterpret_cast<float4 *>(o + 16 * jd) = make_float4(oacc[u][jd][0], oacc[u][jd][1], oacc[u][jd][2], oacc[u][jd][3]);
            if (g == 0) {
                float *ml = part_ml + ((((size_t)split * B + b) * gridDim.y + h) * M + m) * 2;
                ml[0] = mrow[u], ml[1] = l;
            }
        }
        return;
    }
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const float inv = 1.f / rows4_sum(lrow[u]);
        const int m = q0 + 16 * u + (lane & 15);
        if (m >= M) continue;
        float *o = O + (size_t)b * so + (size_t)m * ldo + h * HD + 4 * g;
#pragma unroll
        for (int jd = 0; jd < 2; ++jd) {
            if (VEC) {
                *reinterpret_cast<float4 *>(o + 16 * jd) = make_float4(oacc[u][jd][0] * inv, oacc[u][jd][1] * inv,
                                                                       oacc[u][jd][2] * inv, oacc[u][jd][3] * inv);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) o[16 * jd + q] = oacc[u][jd][q] * inv;
            }
        }
    }
}

// out[b, m, c] = sum_s part_o[s, b, m, c] e^(m_s - m*) / sum_s l_s e^(m_s - m*), m* = max_s m_s (per head): the ranges of a
// key-split attention joined in range order (fixed order: deterministic)
__global__ __launch_bounds__(256) void attention_merge_kernel(const float *__restrict__ part_o, const float *__restrict__ part_ml,
                                                              int nsplit, int B, int M, int heads, float *__restrict__ O,
                                                              int ldo, long long so) {
    const int E = heads * HD;
    const size_t e = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (e >= (size_t)B * M * E) return;
    const int c = (int)(e % E), m = (int)((e / E) % M), b = (int)(e / ((size_t)E * M)), h = c / HD;
    float mx = -__builtin_inff();
    for (int s = 0; s < nsplit; ++s) mx = fmaxf(mx, part_ml[((((size_t)s * B + b) * heads + h) * M + m) * 2]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float *ml = part_ml + ((((size_t)s * B + b) * heads + h) * M + m) * 2;
        const float f = __expf(ml[0] - mx);
        num += part_o[(((size_t)s * B + b) * M + m) * E + c] * f;
        den += ml[1] * f;
    }
    O[(size_t)b * so + (size_t)m * ldo + c] = num / den;
}

// ------------------------------------------------------------------------------------------
// Attention for head widths other than 32 (decoder.model_channel != 256; no shipped config): one WAVE per (sequence, head,
// query); lane j looks at the keys j, j + 64, ... with its own running max / sum / weighted value sum, the lanes are
// merged at the end (rescaled to the wave's max).  Plain fp32 FMAs -- correct for any Decoder(args), not tuned.
// ------------------------------------------------------------------------------------------
template <int D, bool MASK>
__global__ __launch_bounds__(256) void attention_generic_kernel(const float *__restrict__ Q, int ldq, long long sq,
                                                                const float *__restrict__ Kp, int ldk, long long sk,
                                                                const float *__restrict__ V, int ldv, long long sv,
                                                                float *__restrict__ O, int ldo, long long so, int M, int N,
                                                                float scale, int kv_shift, const uint8_t *__restrict__ key_mask) {
    const int lane = threadIdx.x & 63, m = blockIdx.x * 4 + (threadIdx.x >> 6), h = blockIdx.y, b = blockIdx.z;
    if (m >= M) return;
    int bk = b + kv_shift;
    if (bk >= (int)gridDim.z) bk -= (int)gridDim.z;
    const float *q = Q + (size_t)b * sq + (size_t)m * ldq + h * D;
    const float *Kb = Kp + (size_t)bk * sk + h * D, *Vb = V + (size_t)bk * sv + h * D;
    const uint8_t *km = MASK ? key_mask + (size_t)bk * N : nullptr;
    float qv[D], o[D];
#pragma unroll
    for (int c = 0; c < D; ++c) qv[c] = q[c] * scale, o[c] = 0.f;
    float mx = -__builtin_inff(), l = 0.f;
    for (int n = lane; n < N; n += 64) {
        if (MASK && km[n]) continue;
        const float *kr = Kb + (size_t)n * ldk, *vr = Vb + (size_t)n * ldv;
        float sc = 0.f;
#pragma unroll
        for (int c = 0; c < D; ++c) sc = fmaf(qv[c], kr[c], sc);
        const float nm = fmaxf(mx, sc);
        const float corr = __expf(mx - nm), pv = __expf(sc - nm);  // exp(-inf) = 0 on a lane's first key
        l = l * corr + pv;
#pragma unroll
        for (int c = 0; c < D; ++c) o[c] = fmaf(pv, vr[c], o[c] * corr);
        mx = nm;
    }
    const float gm = wave_max(mx);
    const float f = mx == -__builtin_inff() ? 0.f : __expf(mx - gm);  // a lane that saw no key contributes nothing
    const float L = wave_sum(l * f);
    float mine = 0.f, mine2 = 0.f;  // lane c keeps channel c (and c + 64)
#pragma unroll
    for (int c = 0; c < D; ++c) {
        const float t = wave_sum(o[c] * f);
        if (c < 64) mine = lane == c ? t : mine;
        else mine2 = lane == c - 64 ? t : mine2;
    }
    float *orow = O + (size_t)b * so + (size_t)m * ldo + h * D;
    if (lane < D) orow[lane] = mine / L;  // every key masked: 0 / 0 = NaN, as in the reference
    if (D > 64 && lane + 64 < D) orow[lane + 64] = mine2 / L;
}

// ------------------------------------------------------------------------------------------
// F.normalize(x, p=2, dim=-1): x / max(||x||, 1e-12); one wave per row
// ------------------------------------------------------------------------------------------
template <bool VEC>
__global__ __launch_bounds__(256) void l2norm_kernel(const float *__restrict__ X, int R, int C,
                                                     float *__restrict__ out) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float *x = X + (size_t)r * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s = fmaf(x[c], x[c], s);
    const float nrm = fmaxf(sqrtf(wave_sum(s)), 1e-12f);
    if (VEC) {  // C % 4 == 0, 16-byte aligned rows: the quotients leave as 16-byte stores (same sums as above: same bits)
        for (int c = lane * 4; c < C; c += 256) {
            const float4 v = *reinterpret_cast<const float4 *>(x + c);
            *reinterpret_cast<float4 *>(out + (size_t)r * C + c) = make_float4(v.x / nrm, v.y / nrm, v.z / nrm, v.w / nrm);
        }
        return;
    }
    for (int c = lane; c < C; c += 64) out[(size_t)r * C + c] = x[c] / nrm;
}

// ------------------------------------------------------------------------------------------
// a13 dual softmax (decoder.py:186-188): x = S * (1/tau); P = softmax_row(x) * softmax_col(x)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void row_stats_kernel(const float *__restrict__ S, int M, int N, float itau,
                                                        float *__restrict__ rmax, float *__restrict__ rsum) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= M) return;
    const float *s = S + (size_t)r * N;
    float mx = -__builtin_inff();
    for (int c = lane; c < N; c += 64) mx = fmaxf(mx, s[c] * itau);
    mx = wave_max(mx);
    float sum = 0.f;
    for (int c = lane; c < N; c += 64) sum += expf(s[c] * itau - mx);
    sum = wave_sum(sum);
    if (lane == 0) rmax[r] = mx, rsum[r] = sum;
}

__global__ __launch_bounds__(256) void col_stats_kernel(const float *__restrict__ S, int M, int N, float itau,
                                                        float *__restrict__ cmax, float *__restrict__ csum) {
    // 64 columns per block, 4 row-groups; coalesced along columns; blockIdx.y = batch element
    __shared__ float sm[4][64], ss[4][64];
    S += (size_t)blockIdx.y * M * N, cmax += (size_t)blockIdx.y * N, csum += (size_t)blockIdx.y * N;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    float mx = -__builtin_inff(), sum = 0.f;
    if (c < N) {
#pragma unroll 8  // eight loads in flight per thread: the walk down a column is a chain of dependent trips otherwise
        for (int r = g; r < M; r += 4) mx = fmaxf(mx, S[(size_t)r * N + c] * itau);
    }
    sm[g][threadIdx.x & 63] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(sm[0][threadIdx.x & 63], sm[1][threadIdx.x & 63]),
               fmaxf(sm[2][threadIdx.x & 63], sm[3][threadIdx.x & 63]));
    if (c < N) {
#pragma unroll 8
        for (int r = g; r < M; r += 4) sum += expf(S[(size_t)r * N + c] * itau - mx);
    }
    ss[g][threadIdx.x & 63] = sum;
    __syncthreads();
    if (g == 0 && c < N) {
        cmax[c] = mx;
        csum[c] = (ss[0][threadIdx.x] + ss[1][threadIdx.x]) + (ss[2][threadIdx.x] + ss[3][threadIdx.x]);
    }
}

// Column statistics of a TALL score matrix (map-sized registrations: 4096 rows against 256 or 4096 columns, one pair per
// call): the kernel above gives a column to four threads, each walking M / 4 rows -- 1024 dependent trips with a handful
// of workgroups on the chip.  Here the rows are cut into `splits` slabs (blockIdx.z): slab maxima, then slab sums against
// the column's FULL maximum (read from the slab maxima), then one fixed-order sum over the slabs: deterministic, and the
// same max / the same exponentials as the one-kernel form -- only the order of the final additions differs.
__global__ __launch_bounds__(256) void col_slab_kernel(const float *__restrict__ S, int M, int N, float itau, int splits,
                                                       float *__restrict__ pmax, float *__restrict__ psum, int phase) {
    __shared__ float sh[4][64];
    const int batch = blockIdx.y / splits, z = blockIdx.y - batch * splits;
    S += (size_t)batch * M * N;
    const size_t slab = ((size_t)batch * splits + z) * N;
    const int c = blockIdx.x * 64 + (threadIdx.x & 63), g = threadIdx.x >> 6;
    const int rows = (M + splits - 1) / splits, r0 = z * rows, r1 = min(M, r0 + rows);
    float acc = phase == 0 ? -__builtin_inff() : 0.f, mx = 0.f;
    if (c < N) {
        if (phase == 0) {
#pragma unroll 8
            for (int r = r0 + g; r < r1; r += 4) acc = fmaxf(acc, S[(size_t)r * N + c] * itau);
        } else {
            mx = -__builtin_inff();
            for (int k = 0; k < splits; ++k) mx = fmaxf(mx, pmax[((size_t)batch * splits + k) * N + c]);
#pragma unroll 8
            for (int r = r0 + g; r < r1; r += 4) acc += expf(S[(size_t)r * N + c] * itau - mx);
        }
    }
    sh[g][threadIdx.x & 63] = acc;
    __syncthreads();
    if (g == 0 && c < N) {
        const int l = threadIdx.x;
        if (phase == 0) pmax[slab + c] = fmaxf(fmaxf(sh[0][l], sh[1][l]), fmaxf(sh[2][l], sh[3][l]));
        else psum[slab + c] = (sh[0][l] + sh[1][l]) + (sh[2][l] + sh[3][l]);
    }
}
__global__ __launch_bounds__(256) void col_slab_finish_kernel(int N, int splits, const float *__restrict__ pmax,
                                                              const float *__restrict__ psum, float *__restrict__ cmax,
                                                              float *__restrict__ csum) {
    const int batch = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= N) return;
    float mx = -__builtin_inff(), sum = 0.f;
    for (int k = 0; k < splits; ++k) {
        mx = fmaxf(mx, pmax[((size_t)batch * splits + k) * N + c]);
        sum += psum[((size_t)batch * splits + k) * N + c];
    }
    cmax[(size_t)batch * N + c] = mx, csum[(size_t)batch * N + c] = sum;
}

template <bool VEC>
__global__ __launch_bounds__(256) void dual_softmax_kernel(float *__restrict__ S, long long rows, int M, int N, float itau,
                                                           const float *__restrict__ rmax,
                                                           const float *__restrict__ rsum,
                                                           const float *__restrict__ cmax,
                                                           const float *__restrict__ csum) {
    // One block = 256 consecutive columns of ONE row (M = rows of one batch element, `rows` = batch * M): the
    // row and its batch element come from two scalar divisions per block, not from two 64-bit divisions per element.
    // VEC (N % 4 == 0): four columns per thread, 16-byte accesses; per element the same expression, the same bits.
    unsigned r, c;
    if (VEC) {  // flat over the rows' float4 groups (rows * N / 4 < 2^31, checked by the launcher): one 32-bit division
        const unsigned q = (unsigned)N >> 2, e4 = blockIdx.x * 256u + threadIdx.x;
        r = e4 / q, c = (e4 - r * q) * 4u;
    } else {
        const unsigned cblocks = ((unsigned)N + 255u) >> 8;
        r = blockIdx.x / cblocks, c = (blockIdx.x - r * cblocks) * 256u + threadIdx.x;
    }
    if (r >= rows || c >= (unsigned)N) return;
    const size_t e = (size_t)r * N + c, cb = (size_t)(r / (unsigned)M) * N + c;
    const float rm = rmax[r], rs = rsum[r];
    if (VEC) {
        const float4 x = *reinterpret_cast<const float4 *>(S + e);
        const float4 cm = *reinterpret_cast<const float4 *>(cmax + cb), cs = *reinterpret_cast<const float4 *>(csum + cb);
        float4 o;
        o.x = (expf(x.x * itau - rm) / rs) * (expf(x.x * itau - cm.x) / cs.x);
        o.y = (expf(x.y * itau - rm) / rs) * (expf(x.y * itau - cm.y) / cs.y);
        o.z = (expf(x.z * itau - rm) / rs) * (expf(x.z * itau - cm.z) / cs.z);
        o.w = (expf(x.w * itau - rm) / rs) * (expf(x.w * itau - cm.w) / cs.w);
        *reinterpret_cast<float4 *>(S + e) = o;
        return;
    }
    const float x = S[e] * itau;
    S[e] = (expf(x - rm) / rs) * (expf(x - cmax[cb]) / csum[cb]);
}

// ------------------------------------------------------------------------------------------
// top-k (largest) of n non-negative floats, sorted descending (ties: smaller index first).
// v1: ONE workgroup: 4-pass MSB radix select for the k-th value, ordered compaction, bitonic sort.
// ------------------------------------------------------------------------------------------
constexpr int TK_THREADS = 1024;
constexpr int TK_MAXK = 4096;

__global__ __launch_bounds__(TK_THREADS) void topk_kernel(const float *__restrict__ P, long long n, int k,
                                                          float *__restrict__ out_v, int32_t *__restrict__ out_i) {
    __shared__ unsigned hist[256];
    __shared__ unsigned s_prefix, s_mask, s_krem;
    __shared__ int s_wcnt[2][TK_THREADS / 64];
    __shared__ int s_base[2];
    __shared__ float sv[TK_MAXK];
    __shared__ int si[TK_MAXK];
    __shared__ int s_eq[TK_MAXK];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    P += (size_t)blockIdx.x * n, out_v += (size_t)blockIdx.x * k, out_i += (size_t)blockIdx.x * k;  // batch element
    if (t == 0) s_prefix = 0u, s_mask = 0u, s_krem = (unsigned)k;
    for (int pass = 0; pass < 4; ++pass) {
        const int shift = 24 - 8 * pass;
        if (t < 256) hist[t] = 0u;
        __syncthreads();
        const unsigned prefix = s_prefix, mask = s_mask;
        // four independent loads per trip; a thread counts runs of equal bins in a register and touches the LDS
        // histogram only when the bin changes (dual-softmax probabilities share their leading bits: without this the
        // first pass is 65 536 atomics on a handful of addresses)
        unsigned run_bin = 0xffffffffu, run_cnt = 0u;
        auto count = [&](unsigned u) {
            if ((u & mask) != prefix) return;
            const unsigned bin = (u >> shift) & 255u;
            if (bin != run_bin) {
                if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
                run_bin = bin, run_cnt = 0u;
            }
            ++run_cnt;
        };
        long long e = t;
        for (; e + 3 * TK_THREADS < n; e += 4 * TK_THREADS) {
            const unsigned u0 = __float_as_uint(P[e]), u1 = __float_as_uint(P[e + TK_THREADS]);
            const unsigned u2 = __float_as_uint(P[e + 2 * TK_THREADS]), u3 = __float_as_uint(P[e + 3 * TK_THREADS]);
            count(u0), count(u1), count(u2), count(u3);
        }
        for (; e < n; e += TK_THREADS) count(__float_as_uint(P[e]));
        if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
        __syncthreads();
        if (t < 64) {
            // the bin holding the `rem`-th largest key: lane l owns bins 255-4l .. 252-4l (descending), an inclusive
            // scan of the lane sums finds the lane, its four bins are then walked
            const int top = 255 - 4 * t;
            const unsigned c0 = hist[top], c1 = hist[top - 1], c2 = hist[top - 2], c3 = hist[top - 3];
            unsigned inc = c0 + c1 + c2 + c3;
            const unsigned own = inc;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const unsigned o = __shfl_up(inc, off, 64);
                if (t >= off) inc += o;
            }
            const unsigned rem0 = s_krem;
            const unsigned long long hit = __ballot(inc >= rem0);
            const int L = __builtin_ctzll(hit);  // k <= n: some lane always reaches rem0
            if (t == L) {
                unsigned rem = rem0 - (inc - own), bsel = (unsigned)top;
                if (c0 >= rem) {
                    bsel = (unsigned)top;
                } else if (c0 + c1 >= rem) {
                    bsel = (unsigned)(top - 1), rem -= c0;
                } else if (c0 + c1 + c2 >= rem) {
                    bsel = (unsigned)(top - 2), rem -= c0 + c1;
                } else {
                    bsel = (unsigned)(top - 3), rem -= c0 + c1 + c2;
                }
                s_krem = rem;
                s_prefix = prefix | (bsel << shift);
                s_mask = mask | (255u << shift);
            }
        }
        __syncthreads();
    }
    const unsigned thr = s_prefix;  // bit pattern of the k-th largest value
    const int take_eq = (int)s_krem;  // how many elements equal to it are taken (smallest flat indices first)
    const int c_eq = (int)hist[thr & 255u];  // how many elements equal it (last pass histogrammed exact values)
    __syncthreads();
    if (t == 0) s_base[0] = 0, s_base[1] = 0;
    __syncthreads();
    if (c_eq <= TK_MAXK) {
        // unordered collection with LDS atomics: the final sort restores a deterministic order; the
        // equal-to-threshold entries are collected separately and the smallest indices among them are kept
        auto collect = [&](float v, long long e) {
            const unsigned u = __float_as_uint(v);
            if (u > thr) {
                const int p = atomicAdd(&s_base[0], 1);
                sv[p] = v, si[p] = (int)e;
            } else if (u == thr) {
                s_eq[atomicAdd(&s_base[1], 1)] = (int)e;
            }
        };
        long long e = t;
        for (; e + 3 * TK_THREADS < n; e += 4 * TK_THREADS) {  // four independent loads per trip
            const float v0 = P[e], v1 = P[e + TK_THREADS], v2 = P[e + 2 * TK_THREADS], v3 = P[e + 3 * TK_THREADS];
            collect(v0, e), collect(v1, e + TK_THREADS), collect(v2, e + 2 * TK_THREADS), collect(v3, e + 3 * TK_THREADS);
        }
        for (; e < n; e += TK_THREADS) collect(P[e], e);
        __syncthreads();
        // sort the equal entries by index (ascending) and append the first take_eq of them
        int ne2 = 1;
        while (ne2 < c_eq) ne2 <<= 1;
        for (int e = c_eq + t; e < ne2; e += TK_THREADS) s_eq[e] = 0x7fffffff;
        __syncthreads();
        for (int size = 2; size <= ne2; size <<= 1)
            for (int stride = size >> 1; stride > 0; stride >>= 1) {
                for (int e = t; e < ne2 / 2; e += TK_THREADS) {
                    const int lo = 2 * e - (e & (stride - 1)), hi = lo + stride;
                    const bool up = (lo & size) == 0;
                    const int a = s_eq[lo], b2 = s_eq[hi];
                    if ((a > b2) == up) s_eq[lo] = b2, s_eq[hi] = a;
                }
                __syncthreads();
            }
        const float tv = __uint_as_float(thr);
        for (int e = t; e < take_eq; e += TK_THREADS) sv[k - take_eq + e] = tv, si[k - take_eq + e] = s_eq[e];
        __syncthreads();
    } else {
        // degenerate input (more exact ties at the threshold than the list holds): ordered chunk-wise compaction
        for (long long e0 = 0; e0 < n; e0 += TK_THREADS) {
            const long long e = e0 + t;
            unsigned u = 0;
            float v = 0.f;
            if (e < n) v = P[e], u = __float_as_uint(v);
            const bool gt = e < n && u > thr, eq = e < n && u == thr;
            const unsigned long long mg = __ballot(gt), me = __ballot(eq);
            if (lane == 0) s_wcnt[0][w] = __popcll(mg), s_wcnt[1][w] = __popcll(me);
            __syncthreads();
            int bg = s_base[0], be = s_base[1], tg = 0, te = 0;
            for (int x = 0; x < TK_THREADS / 64; ++x) {
                if (x < w) bg += s_wcnt[0][x], be += s_wcnt[1][x];
                tg += s_wcnt[0][x], te += s_wcnt[1][x];
            }
            const unsigned long long lt = (1ull << lane) - 1ull;
            if (gt) {
                const int p = bg + __popcll(mg & lt);
                sv[p] = v, si[p] = (int)e;
            }
            if (eq) {
                const int p = be + __popcll(me & lt);
                if (p < take_eq) sv[k - take_eq + p] = v, si[k - take_eq + p] = (int)e;
            }
            __syncthreads();
            if (t == 0) s_base[0] += tg, s_base[1] += te;
            __syncthreads();
        }
    }
    // bitonic sort, descending by value, ascending index on ties; pad to a power of two
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    for (int e = k + t; e < np2; e += TK_THREADS) sv[e] = -1.f, si[e] = 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < np2 / 2; e += TK_THREADS) {
                const int lo = 2 * e - (e & (stride - 1));
                const int hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const float a = sv[lo], b2 = sv[hi];
                const int ia = si[lo], ib = si[hi];
                const bool a_first = a > b2 || (a == b2 && ia < ib);  // a ranks before b
                if (a_first != desc) sv[lo] = b2, sv[hi] = a, si[lo] = ib, si[hi] = ia;
            }
            __syncthreads();
        }
    }
    for (int e = t; e < k; e += TK_THREADS) out_v[e] = sv[e], out_i[e] = si[e];
}

// ------------------------------------------------------------------------------------------
// top-k for LARGE inputs (map-vs-scan / map-vs-map similarity matrices, up to 4096 x 4096): the same MSB radix
// select, but every pass over the data runs on the whole chip -- per-workgroup LDS histograms folded into a
// global one with atomics, a one-workgroup bin choice between passes, a grid-wide unordered collect, and a
// one-workgroup finish (index sort of the threshold ties + bitonic sort), so the answer is identical to the
// single-workgroup kernel above.
// ------------------------------------------------------------------------------------------
struct TopkState {            // per batch element, in the workspace
    unsigned hist[256];
    unsigned prefix, mask, krem, c_eq;
    int n_gt, n_eq, pad0, pad1;
};

__global__ __launch_bounds__(256) void topk_hist_kernel(const float *__restrict__ P, long long n, int pass,
                                                        TopkState *__restrict__ st) {
    __shared__ unsigned h[256];
    const int t = threadIdx.x;
    P += (size_t)blockIdx.y * n;
    TopkState *s = st + blockIdx.y;
    h[t] = 0u;
    __syncthreads();
    const unsigned prefix = s->prefix, mask = s->mask;
    const int shift = 24 - 8 * pass;
    const long long stride = (long long)gridDim.x * 256 * 4;
    for (long long e = ((long long)blockIdx.x * 256 + t) * 4; e < n; e += stride) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        if (e + 3 < n && ((reinterpret_cast<uintptr_t>(P + e) & 15) == 0)) {
            const float4 q = *reinterpret_cast<const float4 *>(P + e);
            v[0] = q.x, v[1] = q.y, v[2] = q.z, v[3] = q.w;
        } else {
            for (int j = 0; j < 4; ++j) v[j] = (e + j < n) ? P[e + j] : -1.f;  // -1: never matches a non-negative key
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned u = __float_as_uint(v[j]);
            if (e + j < n && (u & mask) == prefix) atomicAdd(&h[(u >> shift) & 255u], 1u);
        }
    }
    __syncthreads();
    if (h[t]) atomicAdd(&s->hist[t], h[t]);
}

__global__ void topk_pick_kernel(TopkState *__restrict__ st, int pass, int k) {
    TopkState *s = st + blockIdx.x;
    __shared__ unsigned h[256];
    const int t = threadIdx.x;
    h[t] = s->hist[t];
    __syncthreads();
    if (t == 0) {
        if (pass == 0) s->krem = (unsigned)k;
        unsigned rem = s->krem, bsel = 0;
        for (int bin = 255; bin >= 0; --bin) {
            if (h[bin] >= rem) {
                bsel = (unsigned)bin;
                break;
            }
            rem -= h[bin];
        }
        const int shift = 24 - 8 * pass;
        s->krem = rem;
        s->prefix |= bsel << shift;
        s->mask |= 255u << shift;
        if (pass == 3) s->c_eq = h[bsel];
    }
    __syncthreads();
    s->hist[t] = 0u;
}

__global__ __launch_bounds__(256) void topk_collect_kernel(const float *__restrict__ P, long long n,
                                                           TopkState *__restrict__ st, float *__restrict__ gt_v,
                                                           int *__restrict__ gt_i, int *__restrict__ eq_i) {
    P += (size_t)blockIdx.y * n;
    TopkState *s = st + blockIdx.y;
    gt_v += (size_t)blockIdx.y * TK_MAXK, gt_i += (size_t)blockIdx.y * TK_MAXK, eq_i += (size_t)blockIdx.y * TK_MAXK;
    const unsigned thr = s->prefix;
    const long long stride = (long long)gridDim.x * 256;
    for (long long e = (long long)blockIdx.x * 256 + threadIdx.x; e < n; e += stride) {
        const float v = P[e];
        const unsigned u = __float_as_uint(v);
        if (u > thr) {
            const int p = atomicAdd(&s->n_gt, 1);
            gt_v[p] = v, gt_i[p] = (int)e;
        } else if (u == thr) {
            const int p = atomicAdd(&s->n_eq, 1);
            if (p < TK_MAXK) eq_i[p] = (int)e;
        }
    }
}

// status[b] = 1 when the threshold has more exact ties than the list holds (caller re-runs the one-workgroup kernel)
__global__ __launch_bounds__(TK_THREADS) void topk_finish_kernel(const TopkState *__restrict__ st, int k,
                                                                 const float *__restrict__ gt_v,
                                                                 const int *__restrict__ gt_i,
                                                                 const int *__restrict__ eq_i,
                                                                 float *__restrict__ out_v, int32_t *__restrict__ out_i) {
    __shared__ float sv[TK_MAXK];
    __shared__ int si[TK_MAXK];
    __shared__ int s_eq[TK_MAXK];
    const int t = threadIdx.x;
    const TopkState *s = st + blockIdx.x;
    gt_v += (size_t)blockIdx.x * TK_MAXK, gt_i += (size_t)blockIdx.x * TK_MAXK, eq_i += (size_t)blockIdx.x * TK_MAXK;
    out_v += (size_t)blockIdx.x * k, out_i += (size_t)blockIdx.x * k;
    const int take_eq = (int)s->krem, c_eq = min((int)s->c_eq, TK_MAXK), n_gt = k - take_eq;
    for (int e = t; e < n_gt; e += TK_THREADS) sv[e] = gt_v[e], si[e] = gt_i[e];
    int ne2 = 1;
    while (ne2 < c_eq) ne2 <<= 1;
    for (int e = t; e < ne2; e += TK_THREADS) s_eq[e] = e < c_eq ? eq_i[e] : 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= ne2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < ne2 / 2; e += TK_THREADS) {
                const int lo = 2 * e - (e & (stride - 1)), hi = lo + stride;
                const bool up = (lo & size) == 0;
                const int a = s_eq[lo], b2 = s_eq[hi];
                if ((a > b2) == up) s_eq[lo] = b2, s_eq[hi] = a;
            }
            __syncthreads();
        }
    const float tv = __uint_as_float(s->prefix);
    for (int e = t; e < take_eq; e += TK_THREADS) sv[n_gt + e] = tv, si[n_gt + e] = s_eq[e];
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    for (int e = k + t; e < np2; e += TK_THREADS) sv[e] = -1.f, si[e] = 0x7fffffff;
    __syncthreads();
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < np2 / 2; e += TK_THREADS) {
                const int lo = 2 * e - (e & (stride - 1)), hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const float a = sv[lo], b2 = sv[hi];
                const int ia = si[lo], ib = si[hi];
                const bool a_first = a > b2 || (a == b2 && ia < ib);
                if (a_first != desc) sv[lo] = b2, sv[hi] = a, si[lo] = ib, si[hi] = ia;
            }
            __syncthreads();
        }
    for (int e = t; e < k; e += TK_THREADS) out_v[e] = sv[e], out_i[e] = si[e];
}

// ------------------------------------------------------------------------------------------
// a14 gather: flat top-k index -> (src row, dst row); X[0:k] = [x[si] | y[di]], X[k:2k] = [y[di] | x[si]]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_pairs_kernel(const float *__restrict__ x, const float *__restrict__ y,
                                                           const int32_t *__restrict__ flat, int k, int M, int N, int E,
                                                           float *__restrict__ X, int32_t *__restrict__ si_out,
                                                           int32_t *__restrict__ di_out) {
    const int p = blockIdx.x, t = threadIdx.x, bz = blockIdx.y;
    x += (size_t)bz * M * E, y += (size_t)bz * N * E, flat += (size_t)bz * k;
    X += (size_t)bz * 2 * k * 2 * E, si_out += (size_t)bz * k, di_out += (size_t)bz * k;
    const int f = flat[p];
    const int si = f / N, di = f - si * N;
    if (t == 0) si_out[p] = si, di_out[p] = di;
    for (int c = t; c < E; c += 256) {
        const float a = x[(size_t)si * E + c], b2 = y[(size_t)di * E + c];
        X[(size_t)p * 2 * E + c] = a;
        X[(size_t)p * 2 * E + E + c] = b2;
        X[(size_t)(k + p) * 2 * E + c] = b2;
        X[(size_t)(k + p) * 2 * E + E + c] = a;
    }
}

// ------------------------------------------------------------------------------------------
// mean over the rows of each batch element: (B,R,C) -> out (B, ldo) at column offset
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void mean_rows_kernel(const float *__restrict__ X, int R, int C,
                                                        float *__restrict__ out, int ldo) {
    const int b = blockIdx.y, c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    const float *x = X + (size_t)b * R * C;
    float s = 0.f;
    for (int r = 0; r < R; ++r) s += x[(size_t)r * C + c];
    out[(size_t)b * ldo + c] = s / (float)R;
}

// ------------------------------------------------------------------------------------------
// a14 + a15: correspondence sets and the iterative weighted Kabsch (decoder.py:202-265)
// ------------------------------------------------------------------------------------------
constexpr int KB = 256;  // threads

template <typename T>
__device__ T block_sum(T v, T *scratch /* KB/64 entries */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = scratch[0];
    for (int i = 1; i < KB / 64; ++i) r += scratch[i];
    return r;
}

__device__ float block_max(float v, float *scratch /* KB/64 entries */) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scratch[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = scratch[0];
    for (int i = 1; i < KB / 64; ++i) r = fmaxf(r, scratch[i]);
    return r;
}

// one-sided Jacobi SVD of a 3x3 (fp64): A = U diag(s) V^T; returns R = V U^T
__device__ void rot_from_cov(const double A[9], double Rm[9]) {
    double G[9], V[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int i = 0; i < 9; ++i) G[i] = A[i];
    for (int sweep = 0; sweep < 30; ++sweep) {
        double off = 0.0;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                double al = 0, be = 0, ga = 0;
                for (int i = 0; i < 3; ++i) {
                    al += G[3 * i + p] * G[3 * i + p];
                    be += G[3 * i + q] * G[3 * i + q];
                    ga += G[3 * i + p] * G[3 * i + q];
                }
                if (al == 0.0 || be == 0.0) continue;
                const double lim = 1e-15 * sqrt(al * be);
                if (fabs(ga) <= lim) continue;
                off += fabs(ga);
                const double zeta = (be - al) / (2.0 * ga);
                const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                for (int i = 0; i < 3; ++i) {
                    const double gp = G[3 * i + p], gq = G[3 * i + q];
                    G[3 * i + p] = c * gp - s * gq;
                    G[3 * i + q] = s * gp + c * gq;
                    const double vp = V[3 * i + p], vq = V[3 * i + q];
                    V[3 * i + p] = c * vp - s * vq;
                    V[3 * i + q] = s * vp + c * vq;
                }
            }
        if (off == 0.0) break;
    }
    // columns of G are s_j * u_j.  R = V U^T = sum_j v_j u_j^T
    double U[9];
    double sig[3];
    for (int j = 0; j < 3; ++j) {
        sig[j] = sqrt(G[j] * G[j] + G[3 + j] * G[3 + j] + G[6 + j] * G[6 + j]);
    }
    int jmin = 0;
    for (int j = 1; j < 3; ++j)
        if (sig[j] < sig[jmin]) jmin = j;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) U[3 * i + j] = sig[j] > 0 ? G[3 * i + j] / sig[j] : 0.0;
    if (!(sig[jmin] > 1e-300)) {  // rank deficient: complete U with the cross product of the other two
        const int a = (jmin + 1) % 3, b2 = (jmin + 2) % 3;
        U[0 + jmin] = U[3 + a] * U[6 + b2] - U[6 + a] * U[3 + b2];
        U[3 + jmin] = U[6 + a] * U[0 + b2] - U[0 + a] * U[6 + b2];
        U[6 + jmin] = U[0 + a] * U[3 + b2] - U[3 + a] * U[0 + b2];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double r = 0;
            for (int l = 0; l < 3; ++l) r += V[3 * i + l] * U[3 * j + l];
            Rm[3 * i + j] = r;
        }
}

// result layout (floats): [0:9] R row-major, [9:12] T, [12] rmse, [13] n_corr, [14] n_inlier, [15] iterations,
// [16] mean of the first 30 inlier confidences (simvec_to_num, system/modules/utils.py:18), [17:20] reserved,
// [20 : 20+n_inlier] confidences of the inliers (in correspondence order).  `header` (optional) receives a
// copy of the first 20 floats (lets the caller assemble an edge table without extra kernels).
// ---- torch.topk's choice among EQUAL weights (decoder.py:233-235: inlier[topk(w, 64)] = True).  The weights are two
// copies of the confidences, so once some offsets are cut away the 64th and 65th largest weight are the same value
// about every other call, and which of the two correspondences becomes an initial inlier is whatever torch.topk's CPU
// kernel leaves in front.  A straddling tie is resolved by replaying it (topk_emulate.h); without a tie the set is
// unique and the parallel ranking stands.
__global__ __launch_bounds__(KB) void corr_kabsch_kernel(
    const float *__restrict__ off /* (2k,3) */, const float *__restrict__ sxyz, int lds_, long long ssrc,
    const float *__restrict__ dxyz, int ldd, long long sdst, const int32_t *__restrict__ si,
    const int32_t *__restrict__ di, const float *__restrict__ conf, int k, float eps2, int num_iter, float std_ratio,
    float *__restrict__ ws /* per batch element: 7*2k floats + 2*2k ints */, float *__restrict__ result,
    float *__restrict__ header, int header_stride) {
    {  // batch element = blockIdx.x
        const size_t bz = blockIdx.x;
        if (off) off += bz * 2 * k * 3;
        sxyz += bz * ssrc, dxyz += bz * sdst, conf += bz * k;
        if (si) si += bz * k, di += bz * k;
        ws += bz * (size_t)(2 * k) * 9;
        result += bz * (size_t)(RES_HDR + 2 * k);
        if (header) header += bz * (size_t)header_stride;
    }
    __shared__ int s_cnt[KB / 64];
    __shared__ int s_run;
    __shared__ float s_f[KB / 64];
    __shared__ double s_R[9], s_T[3];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int n2 = 2 * k;
    float *src = ws, *dst = ws + 3 * (size_t)n2, *wt = ws + 6 * (size_t)n2;
    int *inl = (int *)(ws + 7 * (size_t)n2), *rank = inl + n2;

    // ---- correspondence sets: [src+off_s2d ; src] <-> [dst ; dst+off_d2s], keep |off|^2 <= eps^2,
    //      stable compaction (decoder.py:208-223)
    const bool direct = (off == nullptr);  // test / generic mode: (conf, sxyz, dxyz) ARE the n = k correspondences
    if (t == 0) s_run = direct ? k : 0;
    __syncthreads();
    if (direct) {
        for (int p = t; p < k; p += KB) {
            for (int a = 0; a < 3; ++a) src[3 * p + a] = sxyz[(size_t)p * lds_ + a], dst[3 * p + a] = dxyz[(size_t)p * ldd + a];
            wt[p] = conf[p];
        }
    }
    for (int e0 = 0; !direct && e0 < n2; e0 += KB) {
        const int e = e0 + t;
        bool keep = false;
        float a[3] = {0, 0, 0}, b2[3] = {0, 0, 0}, wv = 0.f;
        int orig = 0;
        if (e < n2) {
            const int p = e < k ? e : e - k;
            const float ox = off[3 * (size_t)e], oy = off[3 * (size_t)e + 1], oz = off[3 * (size_t)e + 2];
            keep = ((ox * ox + oy * oy) + oz * oz) <= eps2;
            const float *sp = sxyz + (size_t)si[p] * lds_, *dp = dxyz + (size_t)di[p] * ldd;
            if (e < k) {
                a[0] = sp[0] + ox, a[1] = sp[1] + oy, a[2] = sp[2] + oz;
                b2[0] = dp[0], b2[1] = dp[1], b2[2] = dp[2];
            } else {
                a[0] = sp[0], a[1] = sp[1], a[2] = sp[2];
                b2[0] = dp[0] + ox, b2[1] = dp[1] + oy, b2[2] = dp[2] + oz;
            }
            wv = conf[p];
            orig = e;
        }
        const unsigned long long m = __ballot(keep);
        if (lane == 0) s_cnt[w] = __popcll(m);
        __syncthreads();
        int base = s_run, tot = 0;
        for (int x = 0; x < KB / 64; ++x) {
            if (x < w) base += s_cnt[x];
            tot += s_cnt[x];
        }
        if (keep) {
            const int p = base + __popcll(m & ((1ull << lane) - 1ull));
            src[3 * p] = a[0], src[3 * p + 1] = a[1], src[3 * p + 2] = a[2];
            dst[3 * p] = b2[0], dst[3 * p + 1] = b2[1], dst[3 * p + 2] = b2[2];
            wt[p] = wv;
            rank[p] = orig;
        }
        __syncthreads();
        if (t == 0) s_run += tot;
        __syncthreads();
    }
    const int n = s_run;
    // ---- initial inliers: w > 0.5, plus the 64 largest weights (decoder.py:233-235).  The weights are
    //      two copies of the descending top-k confidences, so "rank among the kept ones" is a merge:
    //      an entry is in the top 64 iff fewer than 64 kept entries precede it in (conf desc, copy) order.
    __syncthreads();
    if (direct) {
        // generic top-64: rank by (weight desc, index asc), O(n^2 / threads)
        for (int p = t; p < n; p += KB) {
            const float wp = wt[p];
            int before = 0;
            for (int q = 0; q < n; ++q) before += (wt[q] > wp) || (wt[q] == wp && q < p);
            inl[p] = (wp > 0.5f ? 1 : 0) | (before < min(64, n) ? 2 : 0);  // bit 1: among the 64 largest by (weight, index)
        }
    } else {
        auto lower = [&](int lo, int hi, int key) {  // first q in [lo,hi) with rank[q] >= key (rank is ascending)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (rank[mid] < key) lo = mid + 1;
                else hi = mid;
            }
            return lo;
        };
        const int nA = lower(0, n, k);  // kept entries of the first copy
        for (int p = t; p < n; p += KB) {
            const int e = rank[p];
            const bool isB = e >= k;
            const int j = isB ? e - k : e;
            const int a_lt = lower(0, nA, j), a_le = lower(0, nA, j + 1);
            const int b_lt = lower(nA, n, k + j) - nA;
            const int before = a_lt + b_lt + (isB ? (a_le - a_lt) : 0);
            inl[p] = (wt[p] > 0.5f ? 1 : 0) | (before < min(64, n) ? 2 : 0);
        }
    }
    __syncthreads();
    {   // ---- equal weights across the top-64 boundary: the reference's pick (see vi_nth_element above)
        extern __shared__ VI s_vi[];
        const int kk = min(64, n);
        float vmin = __builtin_inff();  // the kk-th largest weight = the smallest one the ranking admitted
        for (int p = t; p < n; p += KB) {
            s_vi[p].v = wt[p], s_vi[p].i = p;
            if (inl[p] & 2) vmin = fminf(vmin, wt[p]);
        }
        vmin = -block_max(-vmin, s_f);
        int ge = 0;
        for (int p = t; p < n; p += KB) {
            ge += wt[p] >= vmin;
            inl[p] = inl[p] != 0;
        }
        ge = block_sum(ge, s_cnt);
        if (ge > kk && !(vmin > 0.5f)) {  // a tie straddles the boundary and the 0.5 rule does not decide it
            if (kk * 64 <= n) {
                if (t == 0) vi_heap_select<true>(s_vi, 0, kk, n);
            } else if (t < 64) {
                LdsU16 sc = (LdsU16)(s_vi + 2 * k);  // scratch behind the 2k pairs: two lists of n 16-bit positions
                vi_nth_element_wave<true>((LdsVI)s_vi, n, kk - 1, sc, sc + 2 * k);  // one wave, a partition round per pass
            }
            __syncthreads();
            for (int p = t; p < n; p += KB)
                if (wt[p] == vmin) inl[p] = 0;
            __syncthreads();
            for (int j = t; j < kk; j += KB) inl[s_vi[j].i] = 1;
        }
        __syncthreads();
    }
    int iter = 0;
    float rmse = 0.f;
    int n_in = 0;
    while (true) {
        // weighted centroids (fp32, like the reference), covariance (fp32), SVD (fp64)
        float sw = 0, cs[3] = {0, 0, 0}, cd[3] = {0, 0, 0};
        for (int p = t; p < n; p += KB)
            if (inl[p]) {
                const float wv = wt[p];
                sw += wv;
                for (int a = 0; a < 3; ++a) cs[a] = fmaf(src[3 * p + a], wv, cs[a]), cd[a] = fmaf(dst[3 * p + a], wv, cd[a]);
            }
        sw = block_sum(sw, s_f);
        for (int a = 0; a < 3; ++a) cs[a] = block_sum(cs[a], s_f) / sw, cd[a] = block_sum(cd[a], s_f) / sw;
        float cov[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        for (int p = t; p < n; p += KB)
            if (inl[p]) {
                const float wv = wt[p];
                for (int a = 0; a < 3; ++a) {
                    const float sa = (src[3 * p + a] - cs[a]) * wv;
                    for (int c = 0; c < 3; ++c) cov[3 * a + c] = fmaf(sa, dst[3 * p + c] - cd[c], cov[3 * a + c]);
                }
            }
        for (int i = 0; i < 9; ++i) cov[i] = block_sum(cov[i], s_f);
        if (t == 0) {
            double A[9], Rm[9];
            for (int i = 0; i < 9; ++i) A[i] = (double)cov[i];
            rot_from_cov(A, Rm);
            for (int i = 0; i < 9; ++i) s_R[i] = Rm[i];
            for (int a = 0; a < 3; ++a)
                s_T[a] = (double)cd[a] - (Rm[3 * a] * (double)cs[0] + Rm[3 * a + 1] * (double)cs[1] + Rm[3 * a + 2] * (double)cs[2]);
        }
        __syncthreads();
        float Rf[9], Tf[3];
        for (int i = 0; i < 9; ++i) Rf[i] = (float)s_R[i];
        for (int a = 0; a < 3; ++a) Tf[a] = (float)s_T[a];
        // residuals for every correspondence; mean / unbiased std over the current inliers
        float es = 0.f;
        int cnt = 0;
        for (int p = t; p < n; p += KB) {
            float e2 = 0.f;
            for (int a = 0; a < 3; ++a) {
                const float r = fmaf(Rf[3 * a + 2], src[3 * p + 2], fmaf(Rf[3 * a + 1], src[3 * p + 1], Rf[3 * a] * src[3 * p])) + Tf[a] - dst[3 * p + a];
                e2 = fmaf(r, r, e2);
            }
            const float er = sqrtf(e2);
            rank[p] = __float_as_int(er);  // rank[] is free now: reuse as the residual buffer
            if (inl[p]) es += er, ++cnt;
        }
        const float tot_e = block_sum(es, s_f);
        const int tot_c = block_sum(cnt, s_cnt);
        const float mean = tot_e / (float)tot_c;
        float vs = 0.f;
        for (int p = t; p < n; p += KB)
            if (inl[p]) {
                const float d = __int_as_float(rank[p]) - mean;
                vs = fmaf(d, d, vs);
            }
        const float sd = sqrtf(block_sum(vs, s_f) / (float)(tot_c - 1));
        const float lim = mean + std_ratio * sd;
        int same = 1, newc = 0;
        for (int p = t; p < n; p += KB) {
            const int nw = __int_as_float(rank[p]) <= lim ? 1 : 0;
            same &= (nw == inl[p]);
            newc += nw;
            inl[p] = nw;  // the new mask is adopted in every exit case (decoder.py:252-256)
        }
        const int diff = block_sum(1 - same, s_cnt);
        n_in = block_sum(newc, s_cnt);
        ++iter;
        if (iter >= num_iter || diff == 0 || n_in < 30) break;
    }
    // rmse over the final inliers with the last R, T; inlier confidences in order
    {
        float Rf[9], Tf[3];
        for (int i = 0; i < 9; ++i) Rf[i] = (float)s_R[i];
        for (int a = 0; a < 3; ++a) Tf[a] = (float)s_T[a];
        float e2s = 0.f;
        for (int p = t; p < n; p += KB)
            if (inl[p]) {
                for (int a = 0; a < 3; ++a) {
                    const float r = fmaf(Rf[3 * a + 2], src[3 * p + 2], fmaf(Rf[3 * a + 1], src[3 * p + 1], Rf[3 * a] * src[3 * p])) + Tf[a] - dst[3 * p + a];
                    e2s = fmaf(r, r, e2s);
                }
            }
        rmse = sqrtf(block_sum(e2s, s_f) / (float)n_in);
        if (t == 0) {
            for (int i = 0; i < 9; ++i) result[i] = Rf[i];
            for (int a = 0; a < 3; ++a) result[9 + a] = Tf[a];
            result[12] = rmse, result[13] = (float)n, result[14] = (float)n_in, result[15] = (float)iter;
            s_run = 0;
        }
        __syncthreads();
        for (int e0 = 0; e0 < n; e0 += KB) {
            const int p = e0 + t;
            const bool keep = p < n && inl[p];
            const unsigned long long m = __ballot(keep);
            if (lane == 0) s_cnt[w] = __popcll(m);
            __syncthreads();
            int base = s_run, tot = 0;
            for (int x = 0; x < KB / 64; ++x) {
                if (x < w) base += s_cnt[x];
                tot += s_cnt[x];
            }
            if (keep) result[RES_HDR + base + __popcll(m & ((1ull << lane) - 1ull))] = wt[p];
            __syncthreads();
            if (t == 0) s_run += tot;
            __syncthreads();
        }
        if (t == 0) {
            const int m30 = min(30, n_in);
            float cm = 0.f;
            for (int q = 0; q < m30; ++q) cm += result[RES_HDR + q];
            result[16] = m30 > 0 ? cm / (float)m30 : 0.f;
            result[17] = result[18] = result[19] = 0.f;
            if (header)
                for (int q = 0; q < RES_HDR; ++q) header[q] = result[q];
        }
    }
}

// ------------------------------------------------------------------------------------------
// Map-tile assembly (PoseGraph.__global_mapping + the centring of global_map_query_graph,
// system/modules/pose_graph.py:373-409,504-510): tile[:, k*S + s] = key_points[sel[k]][:, s] with the last three
// rows (xyz, metres) mapped by  R_c^T ((R_k x + t_k) - t_c);  feature rows are copied.  Both 3x3 products are
// evaluated like the reference's fp32 matmuls (k-ordered fma chains, then the broadcast add / subtract).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void map_tile_kernel(const float *__restrict__ kp, const int32_t *__restrict__ sel,
                                                       const float *__restrict__ poses /* (n_scans,12) */,
                                                       const float *__restrict__ center /* 12 */, int C, int S, int K,
                                                       float *__restrict__ out) {
    const int k = blockIdx.y, sc = sel ? sel[k] : k;
    const float *src = kp + (size_t)sc * C * S;
    const int ld = K * S;
    const int cf = C - 3;
    for (int e = blockIdx.x * 256 + threadIdx.x; e < cf * S; e += gridDim.x * 256) {
        const int c = e / S, s_ = e - c * S;
        out[(size_t)c * ld + k * S + s_] = src[e];
    }
    const float *P = poses + (size_t)sc * 12;
    for (int s_ = blockIdx.x * 256 + threadIdx.x; s_ < S; s_ += gridDim.x * 256) {
        const float x = src[(size_t)cf * S + s_], y = src[(size_t)(cf + 1) * S + s_], z = src[(size_t)(cf + 2) * S + s_];
        float w[3], d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) w[a] = fmaf(P[3 * a + 2], z, fmaf(P[3 * a + 1], y, P[3 * a] * x)) + P[9 + a];
#pragma unroll
        for (int a = 0; a < 3; ++a) d[a] = w[a] - center[9 + a];
#pragma unroll
        for (int a = 0; a < 3; ++a)  // row a of R_c^T = column a of R_c
            out[(size_t)(cf + a) * ld + k * S + s_] = fmaf(center[6 + a], d[2], fmaf(center[3 + a], d[1], center[a] * d[0]));
    }
}

}  // namespace

extern "C" int dpm_map_tile(const float *key_points, const int32_t *select, const float *poses, const float *centering,
                            int C, int S, int K, float *out, dpm_stream_t stream) {
    DPM_CHECK_ARG(key_points && poses && centering && out && C > 3 && S >= 1 && K >= 1);
    hipLaunchKernelGGL(map_tile_kernel, dim3(dpm_cdiv((long long)(C - 3) * S, 256 * 4), K), dim3(256), 0, (hipStream_t)stream,
                       key_points, select, poses, centering, C, S, K, out);
    return dpm_launch_status();
}

extern "C" int dpm_posemb(const float *xyz, int ld, const float *dim_t, int F, int E, int R, float *out,
                          dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && dim_t && out && ld >= 3 && F >= 1 && E >= 3 * F && R >= 1 && (long long)R * E < (1ll << 32));
    hipLaunchKernelGGL(posemb_kernel, dim3(dpm_cdiv((long long)R * E, 256)), dim3(256), 0, (hipStream_t)stream, xyz, ld,
                       dim_t, F, E, R, 3.14159265358979323846f, out);
    return dpm_launch_status();
}

static int attention_launch(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                            const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B,
                            int M, int N, int heads, int head_dim, int kv_shift, const uint8_t *key_mask,
                            const int32_t *seq, dpm_stream_t stream) {
    DPM_CHECK_ARG(Q && K && V && out && B >= 1 && M >= 1 && N >= 1 && heads >= 1 && kv_shift >= 0 && kv_shift < B);
    if (seq && (head_dim != HD || key_mask)) return DPM_EUNSUPPORTED;
    if (head_dim != HD) {  // other decoder widths: the generic kernel
        const float sc = (float)(1.0 / sqrt((double)head_dim));
        const dim3 grid(dpm_cdiv(M, 4), heads, B);
#define DPM_ATTG(D)                                                                                                      \
    do {                                                                                                                 \
        if (key_mask)                                                                                                    \
            hipLaunchKernelGGL((attention_generic_kernel<D, true>), grid, dim3(256), 0, (hipStream_t)stream, Q, ldq, sq, K, ldk, \
                               sk, V, ldv, sv, out, ldo, so, M, N, sc, kv_shift, key_mask);                              \
        else                                                                                                             \
            hipLaunchKernelGGL((attention_generic_kernel<D, false>), grid, dim3(256), 0, (hipStream_t)stream, Q, ldq, sq, K, ldk, \
                               sk, V, ldv, sv, out, ldo, so, M, N, sc, kv_shift, key_mask);                              \
    } while (0)
        if (head_dim == 8) DPM_ATTG(8);
        else if (head_dim == 16) DPM_ATTG(16);
        else if (head_dim == 64) DPM_ATTG(64);
        else if (head_dim == 128) DPM_ATTG(128);
        else return DPM_EUNSUPPORTED;
#undef DPM_ATTG
        return dpm_launch_status();
    }
    const bool vec = ldk % 4 == 0 && ldv % 4 == 0 && sk % 4 == 0 && sv % 4 == 0 && ((uintptr_t)K & 15) == 0 && ((uintptr_t)V & 15) == 0 &&
                     ldo % 4 == 0 && so % 4 == 0 && ((uintptr_t)out & 15) == 0;  // 16-byte K / V loads and output stores
    const float scale = (float)(1.0 / sqrt((double)head_dim));
    // 32 queries per wave when the query count fills such blocks and there are enough of them for the chip
    const bool wide = DPM_ATT_WIDE_MIN > 0 && M % 128 == 0 && (long long)(M / 128) * heads * B >= DPM_ATT_WIDE_MIN;
#define DPM_ATT(V, QT)                                                                                                  \
    hipLaunchKernelGGL((attention_kernel<V, QT>), dim3(dpm_cdiv(M, 64 * QT), heads, B), dim3(256), (size_t)dpm_knob("DPM_ATT_LDS_PAD", 0), (hipStream_t)stream, \
                       Q, ldq, sq, K, ldk, sk, V_, ldv, sv, out, ldo, so, M, N, scale, kv_shift, nullptr, 1, nullptr,     \
                       nullptr, seq)
    const float *V_ = V;
    if (key_mask) {
        if (vec)
            hipLaunchKernelGGL((attention_kernel<true, 1, true>), dim3(dpm_cdiv(M, 64), heads, B), dim3(256), 0, (hipStream_t)stream,
                               Q, ldq, sq, K, ldk, sk, V, ldv, sv, out, ldo, so, M, N, scale, kv_shift, key_mask);
        else
            hipLaunchKernelGGL((attention_kernel<false, 1, true>), dim3(dpm_cdiv(M, 64), heads, B), dim3(256), 0, (hipStream_t)stream,
                               Q, ldq, sq, K, ldk, sk, V, ldv, sv, out, ldo, so, M, N, scale, kv_shift, key_mask);
        return dpm_launch_status();
    }
#ifndef DPM_ATT_BLOCK8
#define DPM_ATT_BLOCK8 1
#endif
#ifndef DPM_ATT_BLOCK8_MIN_M
#define DPM_ATT_BLOCK8_MIN_M 1024
#endif
    // Eight waves per workgroup (128 queries): a K / V tile is fetched, split and staged once per 128 queries instead of 64.
    // Alone it wins at every size (128 x 256 x 256: 65.4 -> 61.0 us; 2 x 4096 x 4096: 264 -> 233 us); in the pipelined step,
    // where the 256-token sequences run between the other stages' workgroups, 512-thread workgroups find their place later
    // and the step is 1.6 % LONGER (4.04 / 4.03 / 4.05 -> 4.09 / 4.12 / 4.11 ms).  So: map-sized query sets only (the
    // rank-0 registrations against a map tile run alone on their stream).
    if (DPM_ATT_BLOCK8 && vec && !wide && M % 128 == 0 && M >= DPM_ATT_BLOCK8_MIN_M) {
        hipLaunchKernelGGL((attention_kernel<true, 1, false, false, 8>), dim3(M / 128, heads, B), dim3(512), 0, (hipStream_t)stream,
                           Q, ldq, sq, K, ldk, sk, V, ldv, sv, out, ldo, so, M, N, scale, kv_shift, nullptr, 1, nullptr, nullptr, seq);
        return dpm_launch_status();
    }
    if (vec && wide) DPM_ATT(true, 2);
    else if (vec) DPM_ATT(true, 1);
    else if (wide) DPM_ATT(false, 2);
    else DPM_ATT(false, 1);
#undef DPM_ATT
    return dpm_launch_status();
}

extern "C" int dpm_attention_masked(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                                    const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B,
                                    int M, int N, int heads, int head_dim, int kv_shift, const uint8_t *key_mask,
                                    dpm_stream_t stream) {
    return attention_launch(Q, ldq, sq, K, ldk, sk, V, ldv, sv, out, ldo, so, B, M, N, heads, head_dim, kv_shift, key_mask,
                            nullptr, stream);
}

// dpm_attention_shifted over batch elements DRAWN from a smaller set of stored sequences: element b's queries are stored
// sequence seq_index[b] (Q + seq_index[b] * sq), its keys / values stored sequence seq_index[(b + kv_shift) mod B]; the
// output is per batch element.  The consecutive-frame registrations of a batch use every frame as a source and as a
// target, and the first cross-attention block's projections depend on the frame alone: they are computed once per frame
// and attended through this entry point (head_dim 32, no key mask).
extern "C" int dpm_attention_indexed(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                                     const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B,
                                     int M, int N, int heads, int head_dim, int kv_shift, const int32_t *seq_index,
                                     dpm_stream_t stream) {
    DPM_CHECK_ARG(seq_index);
    return attention_launch(Q, ldq, sq, K, ldk, sk, V, ldv, sv, out, ldo, so, B, M, N, heads, head_dim, kv_shift, nullptr,
                            seq_index, stream);
}

// Keys and values as pre-split planes (attention_kernel<PRE>): kv_planes holds, per (stored sequence, head, 64-key tile), the 24 KB
// image dpm_linear_bf16x3_kvplanes writes (K planes in the swizzled rows of the score product's A operand, then the transposed V
// planes).  head_dim 32, N % 64 == 0, no key mask.  seq_index as in dpm_attention_indexed (NULL: element b is stored sequence b).
extern "C" size_t dpm_attention_planes_bytes(int n_sequences, int N, int heads) {
    if (n_sequences <= 0 || N <= 0 || heads <= 0 || N % 64 != 0) return 0;
    return (size_t)n_sequences * heads * (N / 64) * (2 * 3 * 64 * HD * sizeof(uint16_t));
}

extern "C" int dpm_attention_planes(const float *Q, int ldq, long long sq, const void *kv_planes, float *out, int ldo, long long so,
                                    int B, int M, int N, int heads, int kv_shift, const int32_t *seq_index, dpm_stream_t stream) {
    DPM_CHECK_ARG(Q && kv_planes && out && B >= 1 && M >= 1 && N >= 64 && N % 64 == 0 && heads >= 1 && kv_shift >= 0 && kv_shift < B);
    DPM_CHECK_ARG(ldq >= heads * HD && ldo >= heads * HD);
    if (ldo % 4 != 0 || so % 4 != 0 || ((uintptr_t)out & 15) != 0 || ((uintptr_t)kv_planes & 15) != 0) return DPM_EUNSUPPORTED;
    const float scale = (float)(1.0 / sqrt((double)HD));
    hipStream_t st = (hipStream_t)stream;
    const uint16_t *kvp = (const uint16_t *)kv_planes;
    if (DPM_ATT_BLOCK8 && M % 128 == 0 && M >= DPM_ATT_BLOCK8_MIN_M)
        hipLaunchKernelGGL((attention_kernel<true, 1, false, false, 8, true>), dim3(M / 128, heads, B), dim3(512), 0, st, Q, ldq, sq, Q, 0,
                           0LL, Q, 0, 0LL, out, ldo, so, M, N, scale, kv_shift, nullptr, 1, nullptr, nullptr, seq_index, kvp);
    else
        hipLaunchKernelGGL((attention_kernel<true, 1, false, false, 4, true>), dim3(dpm_cdiv(M, 64), heads, B), dim3(256), 0, st, Q, ldq, sq,
                           Q, 0, 0LL, Q, 0, 0LL, out, ldo, so, M, N, scale, kv_shift, nullptr, 1, nullptr, nullptr, seq_index, kvp);
    return dpm_launch_status();
}

// Key-split form (few queries, many keys; see attention_kernel<SPLIT>): same arguments as dpm_attention_shifted plus the
// number of key ranges and a workspace of dpm_attention_split_workspace_bytes(B, M, heads, head_dim, nsplit) bytes.
extern "C" size_t dpm_attention_split_workspace_bytes(int B, int M, int heads, int head_dim, int nsplit) {
    if (B <= 0 || M <= 0 || heads <= 0 || head_dim <= 0 || nsplit <= 0) return 0;
    return sizeof(float) * (size_t)nsplit * B * M * ((size_t)heads * head_dim + 2 * (size_t)heads) + 256;
}

extern "C" int dpm_attention_split(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                                   const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B, int M,
                                   int N, int heads, int head_dim, int kv_shift, int nsplit, void *workspace,
                                   dpm_stream_t stream) {
    DPM_CHECK_ARG(Q && K && V && out && workspace && B >= 1 && M >= 1 && N >= 1 && heads >= 1 && kv_shift >= 0 && kv_shift < B);
    DPM_CHECK_ARG(nsplit >= 2 && nsplit <= 64 && (long long)(nsplit - 1) * 64 < N);
    if (head_dim != HD) return DPM_EUNSUPPORTED;
    DPM_CHECK_ARG(ldq >= heads * HD && ldk >= heads * HD && ldv >= heads * HD && ldo >= heads * HD);
    // every range must hold at least one key: ranges are ceil(N / (64 nsplit)) tiles long
    const int chunk = ((N + 64 * nsplit - 1) / (64 * nsplit)) * 64;
    DPM_CHECK_ARG((long long)(nsplit - 1) * chunk < N);
    hipStream_t st = (hipStream_t)stream;
    float *part_o = (float *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255);
    float *part_ml = part_o + (size_t)nsplit * B * M * heads * HD;
    const bool vec = ldk % 4 == 0 && ldv % 4 == 0 && sk % 4 == 0 && sv % 4 == 0 && ((uintptr_t)K & 15) == 0 && ((uintptr_t)V & 15) == 0;
    const float scale = (float)(1.0 / sqrt((double)head_dim));
    const dim3 grid(dpm_cdiv(M, 64) * nsplit, heads, B);
    if (vec)
        hipLaunchKernelGGL((attention_kernel<true, 1, false, true>), grid, dim3(256), 0, st, Q, ldq, sq, K, ldk, sk, V, ldv, sv,
                           out, ldo, so, M, N, scale, kv_shift, nullptr, nsplit, part_o, part_ml);
    else
        hipLaunchKernelGGL((attention_kernel<false, 1, false, true>), grid, dim3(256), 0, st, Q, ldq, sq, K, ldk, sk, V, ldv, sv,
                           out, ldo, so, M, N, scale, kv_shift, nullptr, nsplit, part_o, part_ml);
    hipLaunchKernelGGL(attention_merge_kernel, dim3(dpm_cdiv((long long)B * M * heads * HD, 256)), dim3(256), 0, st, part_o, part_ml,
                       nsplit, B, M, heads, out, ldo, so);
    return dpm_launch_status();
}

extern "C" int dpm_attention_shifted(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                                     const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B,
                                     int M, int N, int heads, int head_dim, int kv_shift, dpm_stream_t stream) {
    return dpm_attention_masked(Q, ldq, sq, K, ldk, sk, V, ldv, sv, out, ldo, so, B, M, N, heads, head_dim, kv_shift, nullptr,
                                stream);
}

extern "C" int dpm_attention(const float *Q, int ldq, long long sq, const float *K, int ldk, long long sk,
                             const float *V, int ldv, long long sv, float *out, int ldo, long long so, int B, int M,
                             int N, int heads, int head_dim, dpm_stream_t stream) {
    return dpm_attention_shifted(Q, ldq, sq, K, ldk, sk, V, ldv, sv, out, ldo, so, B, M, N, heads, head_dim, 0, stream);
}

extern "C" int dpm_l2_normalize(const float *x, int R, int C, float *out, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && out && R >= 1 && C >= 1);
    if (C % 4 == 0 && ((uintptr_t)x & 15) == 0 && ((uintptr_t)out & 15) == 0)
        hipLaunchKernelGGL(l2norm_kernel<true>, dim3(dpm_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, R, C, out);
    else
        hipLaunchKernelGGL(l2norm_kernel<false>, dim3(dpm_cdiv(R, 4)), dim3(256), 0, (hipStream_t)stream, x, R, C, out);
    return dpm_launch_status();
}

constexpr long long TOPK_BIG = 1LL << 18;  // elements per problem above which the grid-wide top-k is used

// row slabs of the column statistics: none up to 512 rows (the batched 256 x 256 hot path keeps its one-kernel form)
static int col_splits(int M) { return M <= 512 ? 1 : std::min(64, (M + 127) / 128); }

extern "C" size_t dpm_pairing_workspace_bytes(int batch, int M, int N) {
    size_t b = sizeof(float) * 2 * (size_t)batch * ((size_t)M + (size_t)N) + 256;
    if ((long long)M * N >= TOPK_BIG)
        b += 256 + (size_t)batch * (sizeof(TopkState) + (size_t)TK_MAXK * (sizeof(float) + 2 * sizeof(int)));
    if (col_splits(M) > 1) b += 256 + sizeof(float) * 2 * (size_t)batch * col_splits(M) * (size_t)N;
    return b;
}

extern "C" int dpm_dual_softmax_topk(float *S, int batch, int M, int N, double tau, int k, float *out_val,
                                     int32_t *out_idx, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(S && out_val && out_idx && workspace && batch >= 1 && M >= 1 && N >= 1 && tau > 0.0);
    DPM_CHECK_ARG(k >= 1 && (long long)k <= (long long)M * N);
    if (k > TK_MAXK || (long long)M * N > 0x7fffffffLL) return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const size_t BM = (size_t)batch * M, BN = (size_t)batch * N;
    float *rmax = (float *)workspace, *rsum = rmax + BM, *cmax = rsum + BM, *csum = cmax + BN;
    const float itau = 1.0f / (float)tau;  // torch divides by the scalar as a multiplication by 1/tau
    hipLaunchKernelGGL(row_stats_kernel, dim3(dpm_cdiv((long long)BM, 4)), dim3(256), 0, st, S, (int)BM, N, itau, rmax, rsum);
    const int splits = col_splits(M);
    if (splits == 1) {
        hipLaunchKernelGGL(col_stats_kernel, dim3(dpm_cdiv(N, 64), batch), dim3(256), 0, st, S, M, N, itau, cmax, csum);
    } else {  // slabs live behind everything else in the workspace
        size_t off = sizeof(float) * 2 * (BM + BN) + 256;
        if ((long long)M * N >= TOPK_BIG)
            off += 256 + (size_t)batch * (sizeof(TopkState) + (size_t)TK_MAXK * (sizeof(float) + 2 * sizeof(int)));
        float *pmax = (float *)((((uintptr_t)workspace + off) + 255) & ~(uintptr_t)255), *psum = pmax + (size_t)batch * splits * N;
        for (int phase = 0; phase < 2; ++phase)
            hipLaunchKernelGGL(col_slab_kernel, dim3(dpm_cdiv(N, 64), batch * splits), dim3(256), 0, st, S, M, N, itau, splits,
                               pmax, psum, phase);
        hipLaunchKernelGGL(col_slab_finish_kernel, dim3(dpm_cdiv(N, 256), batch), dim3(256), 0, st, N, splits, pmax, psum, cmax,
                           csum);
    }
    if (N % 4 == 0 && ((uintptr_t)S & 15) == 0 && ((uintptr_t)cmax & 15) == 0 && ((uintptr_t)csum & 15) == 0 &&
        (long long)BM * (N / 4) < (1ll << 31))
        hipLaunchKernelGGL(dual_softmax_kernel<true>, dim3((unsigned)dpm_cdiv((long long)BM * (N / 4), 256)), dim3(256), 0, st, S,
                           (long long)BM, M, N, itau, rmax, rsum, cmax, csum);
    else
        hipLaunchKernelGGL(dual_softmax_kernel<false>, dim3((unsigned)(BM * dpm_cdiv(N, 256))), dim3(256), 0, st, S, (long long)BM,
                           M, N, itau, rmax, rsum, cmax, csum);
    const long long n = (long long)M * N;
    if (n < TOPK_BIG) {
        hipLaunchKernelGGL(topk_kernel, dim3(batch), dim3(TK_THREADS), 0, st, S, n, k, out_val, out_idx);
        return dpm_launch_status();
    }
    // grid-wide radix select.  Exact ties at the threshold beyond TK_MAXK entries (degenerate inputs) would
    // overflow the tie list; they cannot occur for k <= TK_MAXK distinct-valued softmax products in practice and
    // are clamped (the one-workgroup kernel above has the fully general path).
    uintptr_t wp = ((uintptr_t)(csum + BN) + 255) & ~(uintptr_t)255;
    TopkState *state = (TopkState *)wp;
    float *gt_v = (float *)(state + batch);
    int *gt_i = (int *)(gt_v + (size_t)batch * TK_MAXK);
    int *eq_i = gt_i + (size_t)batch * TK_MAXK;
    hipError_t e = hipMemsetAsync(state, 0, sizeof(TopkState) * (size_t)batch, st);
    if (e != hipSuccess) return (int)e;
    const unsigned G = (unsigned)std::min<long long>(2048 / batch + 1, (n + 1023) / 1024);
    for (int pass = 0; pass < 4; ++pass) {
        hipLaunchKernelGGL(topk_hist_kernel, dim3(G, batch), dim3(256), 0, st, S, n, pass, state);
        hipLaunchKernelGGL(topk_pick_kernel, dim3(batch), dim3(256), 0, st, state, pass, k);
    }
    hipLaunchKernelGGL(topk_collect_kernel, dim3(G, batch), dim3(256), 0, st, S, n, state, gt_v, gt_i, eq_i);
    hipLaunchKernelGGL(topk_finish_kernel, dim3(batch), dim3(TK_THREADS), 0, st, state, k, gt_v, gt_i, eq_i, out_val,
                       out_idx);
    return dpm_launch_status();
}

extern "C" int dpm_gather_pairs(const float *x, const float *y, const int32_t *flat_idx, int batch, int k, int M, int N,
                                int E, float *X, int32_t *src_idx, int32_t *dst_idx, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && y && flat_idx && X && src_idx && dst_idx && batch >= 1 && k >= 1 && M >= 1 && N >= 1 && E >= 1);
    hipLaunchKernelGGL(gather_pairs_kernel, dim3(k, batch), dim3(256), 0, (hipStream_t)stream, x, y, flat_idx, k, M, N, E,
                       X, src_idx, dst_idx);
    return dpm_launch_status();
}

extern "C" int dpm_mean_rows(const float *x, int B, int R, int C, float *out, int ldo, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && out && B >= 1 && R >= 1 && C >= 1 && ldo >= C);
    hipLaunchKernelGGL(mean_rows_kernel, dim3(dpm_cdiv(C, 256), B), dim3(256), 0, (hipStream_t)stream, x, R, C, out, ldo);
    return dpm_launch_status();
}

extern "C" size_t dpm_kabsch_workspace_bytes(int batch, int k) {
    return (size_t)batch * (size_t)(2 * k) * 9 * sizeof(float) + 256;
}

extern "C" int dpm_corr_kabsch(const float *offsets, const float *src_xyz, int ld_src, long long stride_src,
                               const float *dst_xyz, int ld_dst, long long stride_dst, const int32_t *src_idx,
                               const int32_t *dst_idx, const float *conf, int batch, int k, double eps_offset,
                               int num_iter, double std_ratio, void *workspace, float *result, float *header,
                               int header_stride, dpm_stream_t stream) {
    DPM_CHECK_ARG(src_xyz && dst_xyz && conf && workspace && result && batch >= 1);
    DPM_CHECK_ARG(!offsets || (src_idx && dst_idx));
    DPM_CHECK_ARG(k >= 1 && ld_src >= 3 && ld_dst >= 3 && num_iter >= 1 && (!header || header_stride >= RES_HDR));
    // the 2k weights + replay scratch live in dynamic LDS, 24 B per pair, next to the kernel's static arrays: what fits the
    // CU's 160 KB bounds k (about 6000 on gfx950; the decoder asks for at most 4096)
    const size_t kabsch_lds = (sizeof(VI) + 4) * 2 * (size_t)k;
    static size_t kabsch_static = 0;
    if (!kabsch_static) {
        hipFuncAttributes fa;
        kabsch_static = hipFuncGetAttributes(&fa, (const void *)corr_kabsch_kernel) == hipSuccess ? fa.sharedSizeBytes + 1 : 8 * 1024;
    }
    if (kabsch_static + kabsch_lds > 160 * 1024) return DPM_EUNSUPPORTED;
    if (kabsch_lds > 48 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)corr_kabsch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kabsch_lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(corr_kabsch_kernel, dim3(batch), dim3(KB), kabsch_lds, (hipStream_t)stream, offsets, src_xyz, ld_src,
                       stride_src, dst_xyz, ld_dst, stride_dst, src_idx, dst_idx, conf, k,
                       (float)(eps_offset * eps_offset), num_iter, (float)std_ratio, (float *)workspace, result, header,
                       header_stride);
    return dpm_launch_status();
}
