// SetAbstraction / LocalAggregation body on the matrix cores (reference
// network/encoder/pointnext.py:52-61,97-107):
//   out[b,s,:] = max_k relu(LayerNorm(W [fea[idx[b,s,k]] | (xyz[idx]-centre)/radius] + bias))
//
// One workgroup = 64 gathered neighbour rows (2 centres at K=32, 4 at K=16) x all Cout columns.
// The gathered operand never exists in HBM: each K-tile of 32 input channels is gathered straight
// into LDS (one neighbour = one contiguous feature vector -> coalesced 128-byte row segments; the
// three relative-coordinate channels are synthesised in the last tile), multiplied on
// v_mfma_f32_16x16x4_f32 (exact fp32) against the weight tile (native Conv2d layout (Cout,Cin+3),
// k contiguous), and the epilogue does bias + two-pass LayerNorm over the Cout columns of every
// row + ReLU + max over the K rows of a centre without leaving the workgroup.  The next K-tile
// (gather + weights) is prefetched into registers while the current one feeds the MFMAs.
#include "dpm_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int TM = 64;  // gathered rows per workgroup
constexpr int KT = 32;
constexpr int LDS_LD = KT + 2;

template <int COUT, int WM, int WN>  // wave grid WM x WN = 4
__global__ __launch_bounds__(256) void group_mlp_mfma_kernel(
    const float *__restrict__ xyz_all, const float *__restrict__ fea_all, const float *__restrict__ ctr_all,
    const int32_t *__restrict__ idx_all, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ gamma, const float *__restrict__ beta, int N, int S, int K, int Cin, float inv_r,
    float *__restrict__ out_all, const float *__restrict__ W0, const float *__restrict__ b0) {
    // W0/b0 != NULL: the input features are not read but computed on the fly as the per-point affine map
    // fea[c] = b0[c] + W0[c,:] . xyz  (Encoder.point_mlp0, encoder.py:25,53) -- the level-0 feature tensor
    // (N x 16 floats per frame) then never exists in HBM.
    static_assert(WM * WN == 4, "four waves");
    constexpr int RW = TM / WM, CW = COUT / WN, MB = RW / 16, NB = CW / 16, PW = COUT / 32;
    static_assert(MB >= 1 && NB >= 1, "wave tile too small");
    __shared__ float Gs[TM][LDS_LD];
    __shared__ float Ws[COUT][LDS_LD];
    __shared__ float s_part[WN][TM];   // per-row partial sums across the waves that split the columns
    __shared__ int s_max[TM / 16][COUT];  // running max per (centre, column); values >= 0 compare as ints
    __shared__ int s_nidx[TM];
    __shared__ float s_ctr[TM / 16][3];

    const unsigned bid = xcd_chunked_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int b = bid / gridDim.x, t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w / WN, wn = w % WN;
    const int C3 = Cin + 3;
    const int cpb = TM / K;                       // centres per workgroup (2 or 4)
    const int s0 = (bid % gridDim.x) * cpb;       // first centre
    const float *xyz = xyz_all + (size_t)b * N * 3;
    const float *fea = fea_all ? fea_all + (size_t)b * N * Cin : nullptr;

    if (t < TM) {
        const int s = min(s0 + t / K, S - 1);
        const int i = idx_all[((size_t)b * S + s) * K + (t % K)];
        s_nidx[t] = min(max(i, 0), N - 1);
    }
    if (t < cpb * 3) s_ctr[t / 3][t % 3] = ctr_all[((size_t)b * S + min(s0 + t / 3, S - 1)) * 3 + (t % 3)];
    for (int e = t; e < (TM / 16) * COUT; e += 256) (&s_max[0][0])[e] = 0;
    __syncthreads();

    const int sr = t >> 3, sk = (t & 7) * 4;  // staging: row within a 32-row pass, k offset

    // Loads are branch-free (clamped address + select) except for conditions that are uniform over the workgroup,
    // so a prefetch group is issued back to back and only waited for when it is written to LDS one K-tile later.
    // Cin % 4 == 0 and 16-byte aligned features are guaranteed by the dispatcher, hence the three relative
    // coordinates start exactly at k == Cin of one float4 slot.
    auto load_g = [&](int row, int k0) -> float4 {  // 4 consecutive input channels of gathered row `row` in tile k0
        const int k = k0 + sk;
        const bool body_tile = k0 < Cin, tail_tile = Cin >= k0 && Cin < k0 + KT;  // uniform over the workgroup
        const int n = s_nidx[row];
        const bool in = k < Cin;
        // selects are written per component: a select between two float4 objects is lowered through scratch memory
        float v0 = 0.f, v1 = 0.f, v2 = 0.f, v3 = 0.f;
        if (fea && body_tile) {
            const float4 q = *reinterpret_cast<const float4 *>(fea + (size_t)n * Cin + min(k, Cin - 4));
            v0 = in ? q.x : 0.f, v1 = in ? q.y : 0.f, v2 = in ? q.z : 0.f, v3 = in ? q.w : 0.f;
        }
        if ((!fea && body_tile) || tail_tile) {
            const float px = xyz[(size_t)n * 3], py = xyz[(size_t)n * 3 + 1], pz = xyz[(size_t)n * 3 + 2];
            if (!fea && body_tile) {
                const int c = min(k, Cin - 4);
                const float f0 = fmaf(W0[3 * c + 2], pz, fmaf(W0[3 * c + 1], py, fmaf(W0[3 * c], px, b0[c])));
                const float f1 = fmaf(W0[3 * c + 5], pz, fmaf(W0[3 * c + 4], py, fmaf(W0[3 * c + 3], px, b0[c + 1])));
                const float f2 = fmaf(W0[3 * c + 8], pz, fmaf(W0[3 * c + 7], py, fmaf(W0[3 * c + 6], px, b0[c + 2])));
                const float f3 = fmaf(W0[3 * c + 11], pz, fmaf(W0[3 * c + 10], py, fmaf(W0[3 * c + 9], px, b0[c + 3])));
                v0 = in ? f0 : 0.f, v1 = in ? f1 : 0.f, v2 = in ? f2 : 0.f, v3 = in ? f3 : 0.f;
            }
            const int ci = row / K;
            const bool rel = k == Cin;
            const float r0 = (px - s_ctr[ci][0]) * inv_r, r1 = (py - s_ctr[ci][1]) * inv_r, r2_ = (pz - s_ctr[ci][2]) * inv_r;
            v0 = rel ? r0 : v0, v1 = rel ? r1 : v1, v2 = rel ? r2_ : v2, v3 = rel ? 0.f : v3;
        }
        return make_float4(v0, v1, v2, v3);
    };
    auto load_w = [&](int row, int k) -> float4 {  // rows of (Cout, Cin+3) are not 16-byte aligned: four dwords
        const float *p = W + (size_t)row * C3;
        float4 v;
        v.x = p[min(k, C3 - 1)], v.y = p[min(k + 1, C3 - 1)], v.z = p[min(k + 2, C3 - 1)], v.w = p[min(k + 3, C3 - 1)];
        v.x = k < C3 ? v.x : 0.f, v.y = k + 1 < C3 ? v.y : 0.f, v.z = k + 2 < C3 ? v.z : 0.f, v.w = k + 3 < C3 ? v.w : 0.f;
        return v;
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 gr[2], wr[PW];
#pragma unroll
    for (int p = 0; p < 2; ++p) gr[p] = load_g(p * 32 + sr, 0);
#pragma unroll
    for (int p = 0; p < PW; ++p) wr[p] = load_w(p * 32 + sr, sk);

    for (int k0 = 0; k0 < C3; k0 += KT) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float2 *d = reinterpret_cast<float2 *>(&Gs[p * 32 + sr][sk]);
            d[0] = make_float2(gr[p].x, gr[p].y), d[1] = make_float2(gr[p].z, gr[p].w);
        }
#pragma unroll
        for (int p = 0; p < PW; ++p) {
            float2 *d = reinterpret_cast<float2 *>(&Ws[p * 32 + sr][sk]);
            d[0] = make_float2(wr[p].x, wr[p].y), d[1] = make_float2(wr[p].z, wr[p].w);
        }
        __syncthreads();
        if (k0 + KT < C3) {
#pragma unroll
            for (int p = 0; p < 2; ++p) gr[p] = load_g(p * 32 + sr, k0 + KT);
#pragma unroll
            for (int p = 0; p < PW; ++p) wr[p] = load_w(p * 32 + sr, k0 + KT + sk);
        }
#pragma unroll
        for (int kk = 0; kk < KT; kk += 4) {
            float a[MB], bq[NB];
#pragma unroll
            for (int i = 0; i < MB; ++i) a[i] = Gs[wm * RW + i * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int j = 0; j < NB; ++j) bq[j] = Ws[wn * CW + j * 16 + (lane & 15)][kk + (lane >> 4)];
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int j = 0; j < NB; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], bq[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue.  C/D layout: value acc[i][j][q] is row wm*RW + i*16 + (lane>>4)*4 + q, column wn*CW + j*16 + (lane&15)
    float bv[NB], gm[NB], bt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const int c = wn * CW + j * 16 + (lane & 15);
        bv[j] = bias[c], gm[j] = gamma[c], bt[j] = beta[c];
    }
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[i][j][q] += bv[j];

    // LayerNorm pass 1: row means (sum over this wave's columns, then across the WN waves)
    auto row_reduce = [&](float (&v)[MB][4]) {  // in: per-lane partials; out: full-row sums in every lane
#pragma unroll
        for (int i = 0; i < MB; ++i)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float x = v[i][q];
                x += __shfl_xor(x, 1, 64), x += __shfl_xor(x, 2, 64), x += __shfl_xor(x, 4, 64), x += __shfl_xor(x, 8, 64);
                v[i][q] = x;
            }
        if (WN > 1) {
            if ((lane & 15) == 0) {
#pragma unroll
                for (int i = 0; i < MB; ++i)
#pragma unroll
                    for (int q = 0; q < 4; ++q) s_part[wn][wm * RW + i * 16 + (lane >> 4) * 4 + q] = v[i][q];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < MB; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float x = 0.f;
#pragma unroll
                    for (int ww = 0; ww < WN; ++ww) x += s_part[ww][wm * RW + i * 16 + (lane >> 4) * 4 + q];
                    v[i][q] = x;
                }
            __syncthreads();
        }
    };
    float mean[MB][4], var[MB][4];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float x = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) x += acc[i][j][q];
            mean[i][q] = x;
        }
    row_reduce(mean);
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            mean[i][q] *= (1.0f / (float)COUT);
            float x = 0.f;
#pragma unroll
            for (int j = 0; j < NB; ++j) {
                const float d = acc[i][j][q] - mean[i][q];
                x = fmaf(d, d, x);
            }
            var[i][q] = x;
        }
    row_reduce(var);
    // normalise, affine, ReLU, max over the rows of one centre
#pragma unroll
    for (int i = 0; i < MB; ++i) {
        const int row_base = wm * RW + i * 16;   // the 16 rows of this block belong to ONE centre (K >= 16)
        const int ctr = row_base / K;
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float m = 0.f;  // ReLU floor
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float rs = rsqrtf(var[i][q] * (1.0f / (float)COUT) + 1e-5f);
                m = fmaxf(m, fmaf((acc[i][j][q] - mean[i][q]) * rs, gm[j], bt[j]));
            }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            if (lane < 16) atomicMax(&s_max[ctr][wn * CW + j * 16 + lane], __float_as_int(m));
        }
    }
    __syncthreads();
    for (int e = t; e < cpb * COUT; e += 256) {
        const int cc = e / COUT, col = e - cc * COUT;
        if (s0 + cc < S) out_all[((size_t)b * S + s0 + cc) * COUT + col] = __int_as_float(s_max[cc][col]);
    }
}

template <int COUT, int WM, int WN>
int launch(const float *xyz, const float *fea, const float *centers, const int32_t *idx, const float *W,
           const float *bias, const float *gamma, const float *beta, int B, int N, int S, int K, int Cin, float inv_r,
           float *out, hipStream_t st, const float *W0 = nullptr, const float *b0 = nullptr) {
    const int cpb = TM / K;
    hipLaunchKernelGGL((group_mlp_mfma_kernel<COUT, WM, WN>), dim3(dpm_cdiv(S, cpb), B), dim3(256), 0, st, xyz, fea,
                       centers, idx, W, bias, gamma, beta, N, S, K, Cin, inv_r, out, W0, b0);
    return dpm_launch_status();
}

// ---- wave-autonomous variant for narrow layers (Cout <= 64) ------------------------------------------------
// One WAVE owns one centre at a time: no LDS, no barrier, no atomics.
//   * B (the weights) lives in registers for the whole kernel: a wave walks over `cpw` consecutive centres.
//   * A is read straight from the gathered rows.  Inside each 16-channel chunk the K index is remapped so that lane
//     group g = lane>>4 owns channels 16c+4g .. +3: one global_load_dwordx4 per lane feeds FOUR MFMAs (an MFMA only
//     needs A and B to agree on which k sits in which lane group).  The three relative coordinates are one more
//     MFMA with (dx, dy, dz, 0) spread over the four lane groups.
//   * The epilogue stays in registers: C/D rows (lane>>4)*4+q, columns lane&15, so LayerNorm's row sums are DPP
//     reductions over the 16 lanes of a DPP row, and the max over the K rows is max over q, the row blocks and
//     the four lane groups.
__device__ __forceinline__ float row16_sum(float v) {
    v += __int_as_float(dpp_i<0xB1, 0xF>(__float_as_int(v)));   // quad_perm [1,0,3,2]
    v += __int_as_float(dpp_i<0x4E, 0xF>(__float_as_int(v)));   // quad_perm [2,3,0,1]
    v += __int_as_float(dpp_i<0x141, 0xF>(__float_as_int(v)));  // row_half_mirror
    v += __int_as_float(dpp_i<0x140, 0xF>(__float_as_int(v)));  // row_mirror
    return v;
}

template <int COUT, int CIN, int KN, bool FUSED>
__global__ __launch_bounds__(256) void group_mlp_wave_kernel(
    const float *__restrict__ xyz_all, const float *__restrict__ fea_all, const float *__restrict__ ctr_all,
    const int32_t *__restrict__ idx_all, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ gamma, const float *__restrict__ beta, int N, int S, long long total, int cpw,
    float inv_r, float *__restrict__ out_all, const float *__restrict__ W0, const float *__restrict__ b0) {
    constexpr int MB = KN / 16, NB = COUT / 16, NC = CIN / 16, C3 = CIN + 3;
    static_assert(!FUSED || NC == 1, "the fused point_mlp0 path has 16 input channels");
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, i = lane & 15, g = lane >> 4;

    float Bf[NB][NC * 4], Bt[NB], bv[NB], gm[NB], bt[NB];
#pragma unroll
    for (int j = 0; j < NB; ++j) {
        const float *wr = W + (size_t)(j * 16 + i) * C3;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) Bf[j][c * 4 + jj] = wr[16 * c + 4 * g + jj];
        Bt[j] = g < 3 ? wr[CIN + min(g, 2)] : 0.f;
        bv[j] = bias[j * 16 + i], gm[j] = gamma[j * 16 + i], bt[j] = beta[j * 16 + i];
    }
    float w0[4][3], bb0[4];
    if (FUSED) {
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            bb0[jj] = b0[4 * g + jj];
#pragma unroll
            for (int d = 0; d < 3; ++d) w0[jj][d] = W0[3 * (4 * g + jj) + d];
        }
    }

    // centre ids fit 32 bits (dispatcher checks); the frame index is carried along instead of divided out per centre
    const int first = (int)((xcd_chunked_id(blockIdx.x, gridDim.x) * 4 + w) * (unsigned)cpw);
    const int last = (int)min((long long)first + cpw, total);
    const int gc = min(g, 2);  // lane group g carries relative coordinate g (group 3 carries the zero padding)
    struct Rows {  // the gathered A operand of one centre
        float4 a[MB][NC];
        float rel[MB];
    };
    auto load_idx = [&](int cc, int (&n)[MB]) {
        const int c = min(cc, last - 1);
#pragma unroll
        for (int m = 0; m < MB; ++m) n[m] = idx_all[(size_t)c * KN + m * 16 + i];
    };
    auto gather = [&](int cc, const int (&n)[MB], Rows &r) {
        const int c = min(cc, last - 1), b = c / S;
        const float *xyz = xyz_all + (size_t)b * N * 3;
        const float *fea = FUSED ? nullptr : fea_all + (size_t)b * N * CIN;
        const float cg = ctr_all[(size_t)c * 3 + gc];
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int nm = min(max(n[m], 0), N - 1);
            if (FUSED) {
                // features from the point itself: three loads per lane; the relative coordinate falls out of them
                const float px = xyz[(size_t)nm * 3], py = xyz[(size_t)nm * 3 + 1], pz = xyz[(size_t)nm * 3 + 2];
                float f[4];
#pragma unroll
                for (int jj = 0; jj < 4; ++jj)
                    f[jj] = fmaf(w0[jj][2], pz, fmaf(w0[jj][1], py, fmaf(w0[jj][0], px, bb0[jj])));
                r.a[m][0] = make_float4(f[0], f[1], f[2], f[3]);
                r.rel[m] = ((g == 0 ? px : (g == 1 ? py : pz)) - cg) * inv_r;
            } else {
#pragma unroll
                for (int c4 = 0; c4 < NC; ++c4)
                    r.a[m][c4] = *reinterpret_cast<const float4 *>(fea + (size_t)nm * CIN + 16 * c4 + 4 * g);
                r.rel[m] = (xyz[(size_t)nm * 3 + gc] - cg) * inv_r;  // one load per lane, no divergence
            }
            r.rel[m] = g == 3 ? 0.f : r.rel[m];
        }
    };
    auto compute = [&](int cc, const Rows &r) {
        float mx[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) mx[j] = 0.f;  // ReLU floor
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            f32x4 acc[NB];
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = f32x4{bv[j], bv[j], bv[j], bv[j]};  // bias folded into the accumulator
#pragma unroll
            for (int c = 0; c < NC; ++c)
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[m][c].x, Bf[j][c * 4 + 0], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[m][c].y, Bf[j][c * 4 + 1], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[m][c].z, Bf[j][c * 4 + 2], acc[j], 0, 0, 0);
                    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.a[m][c].w, Bf[j][c * 4 + 3], acc[j], 0, 0, 0);
                }
#pragma unroll
            for (int j = 0; j < NB; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(r.rel[m], Bt[j], acc[j], 0, 0, 0);
            // two-pass LayerNorm over the COUT columns of each of this lane group's 4 rows, ReLU, max over rows
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float sum = 0.f;
#pragma unroll
                for (int j = 0; j < NB; ++j) sum += acc[j][q];
                const float mean = row16_sum(sum) * (1.0f / (float)COUT);
                float sq = 0.f;
#pragma unroll
                for (int j = 0; j < NB; ++j) {
                    const float d = acc[j][q] - mean;
                    sq = fmaf(d, d, sq);
                }
                const float rs = rsqrtf(row16_sum(sq) * (1.0f / (float)COUT) + 1e-5f);
#pragma unroll
                for (int j = 0; j < NB; ++j) mx[j] = fmaxf(mx[j], fmaf((acc[j][q] - mean) * rs, gm[j], bt[j]));
            }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            float v = mx[j];
            v = fmaxf(v, __shfl_xor(v, 16, 64));
            v = fmaxf(v, __shfl_xor(v, 32, 64));
            if (g == 0) out_all[(size_t)cc * COUT + j * 16 + i] = v;
        }
    };
    // All gathers of a centre are issued before its first MFMA and the neighbour indices of the next centre are
    // fetched meanwhile.  (Keeping two centres in flight per wave measured slower: the kernel is bound by VALU issue
    // in the LayerNorm epilogue, not by load latency, and the extra registers cost occupancy.)
    if (first >= last) return;
    int n[MB];
    Rows A;
    load_idx(first, n);
    for (int cc = first; cc < last; ++cc) {
        gather(cc, n, A);
        load_idx(cc + 1, n);
        compute(cc, A);
    }
}

template <int COUT, int CIN, bool FUSED>
int launch_wave(const float *xyz, const float *fea, const float *centers, const int32_t *idx, const float *W,
                const float *bias, const float *gamma, const float *beta, int B, int N, int S, int K, float inv_r,
                float *out, hipStream_t st, const float *W0, const float *b0) {
    const long long total = (long long)B * S;
    if (total * 32 >= (1LL << 31)) return DPM_EUNSUPPORTED;
    const int cpw = total >= (1 << 16) ? 8 : (total >= (1 << 13) ? 2 : 1);  // centres per wave
    const unsigned grid = dpm_cdiv(total, 4LL * cpw);
    if (K == 32)
        hipLaunchKernelGGL((group_mlp_wave_kernel<COUT, CIN, 32, FUSED>), dim3(grid), dim3(256), 0, st, xyz, fea, centers,
                           idx, W, bias, gamma, beta, N, S, total, cpw, inv_r, out, W0, b0);
    else
        hipLaunchKernelGGL((group_mlp_wave_kernel<COUT, CIN, 16, FUSED>), dim3(grid), dim3(256), 0, st, xyz, fea, centers,
                           idx, W, bias, gamma, beta, N, S, total, cpw, inv_r, out, W0, b0);
    return dpm_launch_status();
}

}  // namespace

namespace {

// ---- "project before gather" (the default path of the encoder) ---------------------------------------------
// The layer is linear before its LayerNorm:  W [fea_n ; rel] + b  =  (W_f fea_n + b) + W_r rel.  The first term
// depends on the neighbour point alone, so it is computed ONCE PER POINT by a dense GEMM (P = fea W_f^T + b,
// dpm_linear with ldw = Cin+3) instead of once per (centre, neighbour) pair -- K times fewer multiply-adds -- and
// the per-pair work left is: gather the Cout-vector P[n], add the three relative-coordinate terms as an fma chain
// (same k order as the full dot product), two-pass LayerNorm, ReLU, max over the K neighbours.  For the first
// stage P itself is affine in the point (P = A xyz + c, A = W_f W0, c = W_f b0 + b) and is evaluated on the fly,
// so neither the level-0 features nor their projection ever exist in memory.
// One wave per centre; G = Cout/(4V) lanes share a row (float4 per lane, V float4 for Cout = 512), 64/G rows per
// wave pass, LayerNorm sums are DPP / shuffle reductions inside the lane group.
// The in-row steps are written as v_add_f32 with a DPP operand (one instruction + the two wait states a DPP read
// needs after the write of its register); from update_dpp the compiler builds copy + s_nop + mov_dpp + add.
#ifdef DPM_DPP_BUILTIN   // experimental builds: update_dpp + add (the compiler pads the hazards of what it builds from them)
template <int G>
__device__ __forceinline__ float lane_group_sum(float v) {
    if (G >= 2) v += __int_as_float(dpp_i<0xB1, 0xF>(__float_as_int(v)));
    if (G >= 4) v += __int_as_float(dpp_i<0x4E, 0xF>(__float_as_int(v)));
    if (G >= 8) v += __int_as_float(dpp_i<0x141, 0xF>(__float_as_int(v)));
    if (G >= 16) v += __int_as_float(dpp_i<0x140, 0xF>(__float_as_int(v)));
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
#else
#define DPM_ADD_DPP(ctrl) "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
template <int G>
__device__ __forceinline__ float lane_group_sum(float v) {
    if (G >= 16) asm(DPM_ADD_DPP("quad_perm:[1,0,3,2]") DPM_ADD_DPP("quad_perm:[2,3,0,1]") DPM_ADD_DPP("row_half_mirror")
                     DPM_ADD_DPP("row_mirror") "s_nop 0" : "+v"(v));
    else if (G >= 8) asm(DPM_ADD_DPP("quad_perm:[1,0,3,2]") DPM_ADD_DPP("quad_perm:[2,3,0,1]") DPM_ADD_DPP("row_half_mirror")
                         "s_nop 0" : "+v"(v));
    else if (G >= 4) asm(DPM_ADD_DPP("quad_perm:[1,0,3,2]") DPM_ADD_DPP("quad_perm:[2,3,0,1]") "s_nop 0" : "+v"(v));
    else if (G >= 2) asm(DPM_ADD_DPP("quad_perm:[1,0,3,2]") "s_nop 0" : "+v"(v));
    if (G >= 32) v += __shfl_xor(v, 16, 64);
    if (G >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}
#undef DPM_ADD_DPP
#endif
// max without the NaN canonicalisation fmaxf drags in (the operands are LayerNorm outputs of finite inputs)
__device__ __forceinline__ float vmax_raw(float a, float b) {
#ifdef DPM_VMAX_BUILTIN
    return fmaxf(a, b);
#endif
    float r;
    asm("v_max_f32_e32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// FOLD (round 5): the relative-coordinate term is linear in the POINT and in the CENTRE separately,
//   W_r (p_n - c) / r  =  W_r' p_n  -  W_r' c,      W_r' = W_r / r,
// so its point half is folded into what is gathered -- the projection GEMM's epilogue adds W_r' p_n to the projected row
// (dpm_linear_bf16x3_rank3) -- and the centre half is ONE vector per centre, subtracted per row: 4 subtractions instead of 3 + 12
// operations per gathered row and lane, and the projected path gathers no coordinates at all.  The price is cancellation:
// |W_r' p| is up to |p| / r (17 at the projected layers of radius 0.1, which carry most of it) times the term it replaces, i.e. a
// rounding error of ~1e-6 relative instead of ~1e-7 on the pre-LayerNorm values: feature error against the oracle 1.5e-5 median
// instead of 4e-6, poses unchanged (DESIGN.md section 2 has the figures per radius).  The affine first level does not pay it: see
// `FOLD && AFFINE` below.
// CENTRED (round 5, with FOLD): the caller has moved LayerNorm's mean removal into the layer -- (I - 11^T / C) applied to W_f, W_r,
// the bias (and, AFFINE, to the point map and its constant), so that every pre-LayerNorm row has zero mean over its channels by
// construction: the row sum, its lane-group reduction and the subtraction (about a third of the instructions per gathered row)
// are not executed.  What is left of the mean is the rounding of the row's own elements (~1e-7 of their magnitude / sqrt(C)).
// The same caller multiplies channel c of the layer by sign(gamma_c): gamma (y rs) + beta = |gamma| (sign(gamma) y rs) + beta is
// then a non-decreasing function of the gathered value, and a non-decreasing function commutes with the maximum over the
// neighbours bit for bit -- the kernel keeps the running maximum of y rs and applies |gamma|, beta and the ReLU once per centre
// (a multiply-add per channel and gathered row less; sign flips are exact, so the result equals the per-row form's).
template <int COUT, int V, bool AFFINE, bool FOLD = false, bool CENTRED = false>
__global__ __launch_bounds__(256) void group_gather_ln_max_kernel(
    const float *__restrict__ P_all, const float *__restrict__ A, const float *__restrict__ cvec,
    const float *__restrict__ xyz_all, const float *__restrict__ ctr_all, const int32_t *__restrict__ idx_all,
    const float *__restrict__ Wr, int ldwr, const float *__restrict__ gamma, const float *__restrict__ beta, int N,
    int S, int K, long long total, int cpw, float inv_r, float *__restrict__ out_all) {
    constexpr int G = COUT / (4 * V), RPW = 64 / G;  // lanes per row, rows per wave pass
    valu_bound_priority();
    // the wave index through readfirstlane: the centre loop, its frame pointers and the centre itself become scalar
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), gl = lane % G, gr = lane / G;
    // per-lane constants for its 4*V channels: relative-coordinate weights, LayerNorm affine, (AFFINE) the point map
    float wr[V][4][3], gm[V][4], bt[V][4], am[AFFINE ? V : 1][4][3], cv[AFFINE ? V : 1][4];
#pragma unroll
    for (int v = 0; v < V; ++v)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = 4 * (gl + G * v) + e;
            gm[v][e] = gamma[c], bt[v][e] = beta[c];
#pragma unroll
            for (int d = 0; d < 3; ++d) wr[v][e][d] = Wr[(size_t)c * ldwr + d] * inv_r;   // (p - centre) / r folded into the weight
            if (AFFINE) {
                cv[v][e] = cvec[c];
#pragma unroll
                for (int d = 0; d < 3; ++d) am[v][e][d] = A[3 * c + d];
            }
        }
    if (FOLD && AFFINE) {
        // The first level is affine in the point: A p + c + W_r' (p - centre) = (A centre + c) + (A + W_r') (p - centre) -- one
        // constant per CENTRE and one 3-term chain per row on the RELATIVE coordinates: 15 operations per row and lane instead of 27,
        // and none of the projected form's cancellation (|p| / r is 20-35 at this level, its largest: folding W_r' p into the point
        // map and subtracting W_r' centre, as the other levels must, cost 3 operations less and most of the path's rounding error).
        // wr <- A + W_r' (the per-row map), am stays A (the per-centre constant)
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int d = 0; d < 3; ++d) wr[v][e][d] += am[v][e][d];
    }
    const int first = (int)((xcd_chunked_id(blockIdx.x, gridDim.x) * 4 + w) * (unsigned)cpw);
    const int last = (int)min((long long)first + cpw, total);
    for (int cc = first; cc < last; ++cc) {
        const int b = cc / S;
        // the frame's coordinates, its projected rows and the centre's index row through buffer descriptors built from scalars:
        // a gather's address is then ONE 32-bit offset per lane (no 64-bit vector address arithmetic per load)
        const auto xyz_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(xyz_all + (size_t)b * N * 3), 0, N * 12, 0x00020000);
        const auto P_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(AFFINE ? xyz_all : P_all + (size_t)b * N * COUT), 0,
                                                            AFFINE ? 0 : (int)((size_t)N * COUT * 4 < 0x7fffffffu ? (size_t)N * COUT * 4 : 0x7fffffffu), 0x00020000);
        const auto idx_rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(idx_all + (size_t)cc * K), 0, K * 4, 0x00020000);
        const float cx = ctr_all[(size_t)cc * 3], cy = ctr_all[(size_t)cc * 3 + 1], cz = ctr_all[(size_t)cc * 3 + 2];
        float mx[V][4];
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) mx[v][e] = CENTRED ? -__builtin_inff() : 0.f;  // ReLU floor (CENTRED: applied at the end)
        float kc[FOLD ? V : 1][4];   // FOLD: the centre half, W_r' c (AFFINE: minus the point map's constant, so that one add serves both)
        if (FOLD) {
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (AFFINE) kc[v][e] = fmaf(am[v][e][2], cz, fmaf(am[v][e][1], cy, fmaf(am[v][e][0], cx, cv[v][e])));   // A centre + c
                    else kc[v][e] = fmaf(wr[v][e][2], cz, fmaf(wr[v][e][1], cy, wr[v][e][0] * cx));
                }
        }
        // UG row passes at a time: their UG index loads go out together, then the UG x (xyz, projected row) gathers,
        // and only then the arithmetic -- the chain index -> gather is paid once per group instead of once per pass
        constexpr int UG = V == 1 ? (RPW >= 8 ? 4 : 8) : 2;  // K >= 16 everywhere: never more passes than rows
        for (int r0 = 0; r0 < K; r0 += RPW * UG) {
            unsigned ng[UG];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                const int r = min(r0 + u * RPW + gr, K - 1);  // rows past K repeat the last one: a max does not mind
                ng[u] = (unsigned)__builtin_amdgcn_raw_buffer_load_b32(idx_rs, r * 4, 0, 0);
            }
            float pxg[UG], pyg[UG], pzg[UG];
            float4 pg[UG][AFFINE ? 1 : V];
#pragma unroll
            for (int u = 0; u < UG; ++u) {
                // 32-bit element offsets from the (scalar) frame pointers: a frame is far below 2^32 bytes
                ng[u] = (unsigned)min(max((int)ng[u], 0), N - 1);
                using u32x3 = __attribute__((ext_vector_type(3))) unsigned;
                using u32x4 = __attribute__((ext_vector_type(4))) unsigned;
                if (AFFINE || !FOLD) {
                    const u32x3 pn = __builtin_amdgcn_raw_buffer_load_b96(xyz_rs, (int)__umul24(ng[u], 12u), 0, 0);   // one 12-byte load (N < 2^24: launch_gather)
                    pxg[u] = __uint_as_float(pn[0]), pyg[u] = __uint_as_float(pn[1]), pzg[u] = __uint_as_float(pn[2]);
                } else {
                    pxg[u] = pyg[u] = pzg[u] = 0.f;
                }
                if (!AFFINE) {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const u32x4 q = __builtin_amdgcn_raw_buffer_load_b128(P_rs, (int)(ng[u] * (unsigned)(COUT * 4) + 16u * (unsigned)(gl + G * v)), 0, 0);
                        pg[u][v] = make_float4(__uint_as_float(q[0]), __uint_as_float(q[1]), __uint_as_float(q[2]), __uint_as_float(q[3]));
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < UG; ++u) {
            const float px = pxg[u], py = pyg[u], pz = pzg[u];
            const float rx = px - cx, ry = py - cy, rz = pz - cz;   // 1 / r sits in the weights
            float y[V][4], sum = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float4 p4;
                if (AFFINE && FOLD) {   // wr = A + W_r' on the relative coordinates, kc = A centre + c
                    p4.x = fmaf(wr[v][0][2], rz, fmaf(wr[v][0][1], ry, fmaf(wr[v][0][0], rx, kc[v][0])));
                    p4.y = fmaf(wr[v][1][2], rz, fmaf(wr[v][1][1], ry, fmaf(wr[v][1][0], rx, kc[v][1])));
                    p4.z = fmaf(wr[v][2][2], rz, fmaf(wr[v][2][1], ry, fmaf(wr[v][2][0], rx, kc[v][2])));
                    p4.w = fmaf(wr[v][3][2], rz, fmaf(wr[v][3][1], ry, fmaf(wr[v][3][0], rx, kc[v][3])));
                } else if (AFFINE) {
                    p4.x = fmaf(am[v][0][2], pz, fmaf(am[v][0][1], py, fmaf(am[v][0][0], px, cv[v][0])));
                    p4.y = fmaf(am[v][1][2], pz, fmaf(am[v][1][1], py, fmaf(am[v][1][0], px, cv[v][1])));
                    p4.z = fmaf(am[v][2][2], pz, fmaf(am[v][2][1], py, fmaf(am[v][2][0], px, cv[v][2])));
                    p4.w = fmaf(am[v][3][2], pz, fmaf(am[v][3][1], py, fmaf(am[v][3][0], px, cv[v][3])));
                } else {
                    p4 = pg[u][v];
                }
                if (FOLD) {
                    if (AFFINE) y[v][0] = p4.x, y[v][1] = p4.y, y[v][2] = p4.z, y[v][3] = p4.w;
                    else y[v][0] = p4.x - kc[v][0], y[v][1] = p4.y - kc[v][1], y[v][2] = p4.z - kc[v][2], y[v][3] = p4.w - kc[v][3];
                } else {
                    y[v][0] = fmaf(wr[v][0][2], rz, fmaf(wr[v][0][1], ry, fmaf(wr[v][0][0], rx, p4.x)));
                    y[v][1] = fmaf(wr[v][1][2], rz, fmaf(wr[v][1][1], ry, fmaf(wr[v][1][0], rx, p4.y)));
                    y[v][2] = fmaf(wr[v][2][2], rz, fmaf(wr[v][2][1], ry, fmaf(wr[v][2][0], rx, p4.z)));
                    y[v][3] = fmaf(wr[v][3][2], rz, fmaf(wr[v][3][1], ry, fmaf(wr[v][3][0], rx, p4.w)));
                }
                if (!CENTRED) sum += (y[v][0] + y[v][1]) + (y[v][2] + y[v][3]);
            }
            float mean = 0.f;
            if (!CENTRED) mean = lane_group_sum<G>(sum) * (1.0f / (float)COUT);
            float sq = 0.f;
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!CENTRED) y[v][e] -= mean;
                    sq = fmaf(y[v][e], y[v][e], sq);
                }
            // var + eps >= 1e-5 is a normal number: the bare v_rsq_f32 (what rsqrtf issues after its denormal scaling)
            const float rs = __builtin_amdgcn_rsqf(lane_group_sum<G>(sq) * (1.0f / (float)COUT) + 1e-5f);
#pragma unroll
            for (int v = 0; v < V; ++v)
#pragma unroll
                for (int e = 0; e < 4; ++e) mx[v][e] = vmax_raw(mx[v][e], CENTRED ? y[v][e] * rs : fmaf(y[v][e] * rs, gm[v][e], bt[v][e]));
                    }
        }
        // max over the row groups of the wave (lanes with equal gl), then the first group stores
#pragma unroll
        for (int v = 0; v < V; ++v)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float m = mx[v][e];
#pragma unroll
                for (int off = G; off < 64; off <<= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
                mx[v][e] = CENTRED ? fmaxf(fmaf(m, fabsf(gm[v][e]), bt[v][e]), 0.f) : m;
            }
        if (gr == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v)
                *reinterpret_cast<float4 *>(out_all + (size_t)cc * COUT + 4 * (gl + G * v)) =
                    make_float4(mx[v][0], mx[v][1], mx[v][2], mx[v][3]);
        }
    }
}

template <int COUT, int V, bool AFFINE, bool FOLD = false, bool CENTRED = false>
int launch_gather(const float *P, const float *A, const float *cvec, const float *xyz, const float *centers,
                  const int32_t *idx, const float *Wr, int ldwr, const float *gamma, const float *beta, int B, int N,
                  int S, int K, float inv_r, float *out, hipStream_t st) {
    const long long total = (long long)B * S;
    if (total >= (1LL << 30) || N >= (1 << 24) || (long long)N * COUT * 4 >= (1LL << 31)) return DPM_EUNSUPPORTED;   // 32-bit buffer offsets
    const int cpw = total >= (1 << 16) ? 8 : (total >= (1 << 13) ? 2 : 1);  // centres per wave
    static_assert(FOLD || !CENTRED, "the centred form is a folded form");
    hipLaunchKernelGGL((group_gather_ln_max_kernel<COUT, V, AFFINE, FOLD, CENTRED>), dim3(dpm_cdiv(total, 4LL * cpw)), dim3(256), (size_t)dpm_knob("DPM_GATHER_LDS_PAD", 0), st, P,
                       A, cvec, xyz, centers, idx, Wr, ldwr, gamma, beta, N, S, K, total, cpw, inv_r, out);
    return dpm_launch_status();
}

}  // namespace

extern "C" int dpm_group_gather_ln_max(const float *P, const float *xyz, const float *centers, const int32_t *idx,
                                       const float *W_rel, int ldw_rel, const float *gamma, const float *beta, int B,
                                       int N, int S, int K, int Cout, double radius, float *out, dpm_stream_t stream) {
    DPM_CHECK_ARG(P && xyz && centers && idx && W_rel && gamma && beta && out);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && ldw_rel >= 3 && radius > 0.0);
    DPM_CHECK_ARG(((uintptr_t)P & 15) == 0 && ((uintptr_t)out & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    const float inv_r = 1.0f / (float)radius;
#define DPM_GG(C, V) return launch_gather<C, V, false>(P, nullptr, nullptr, xyz, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, inv_r, out, st)
    switch (Cout) {
        case 32: DPM_GG(32, 1);
        case 64: DPM_GG(64, 1);
        case 128: DPM_GG(128, 1);
        case 256: DPM_GG(256, 1);
        case 512: DPM_GG(512, 2);
        default: return DPM_EUNSUPPORTED;
    }
#undef DPM_GG
}

// dpm_group_gather_ln_max for rows P' that ALREADY carry the point half of the relative-coordinate term (P' = fea W_f^T + b +
// xyz (W_rel / radius)^T: dpm_linear_bf16x3_rank3): the kernel subtracts the centre half, (W_rel / radius) centre, and gathers no
// coordinates.  Same contract otherwise (xyz is not read).
static int gather_folded(const float *P, const float *centers, const int32_t *idx, const float *W_rel, int ldw_rel,
                         const float *gamma, const float *beta, int B, int N, int S, int K, int Cout, double radius, float *out,
                         dpm_stream_t stream, bool centred) {
    DPM_CHECK_ARG(P && centers && idx && W_rel && gamma && beta && out);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && ldw_rel >= 3 && radius > 0.0);
    DPM_CHECK_ARG(((uintptr_t)P & 15) == 0 && ((uintptr_t)out & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    const float inv_r = 1.0f / (float)radius;
#define DPM_GF(C, V)                                                                                                              \
    return centred ? launch_gather<C, V, false, true, true>(P, nullptr, nullptr, centers, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, inv_r, out, st) \
                   : launch_gather<C, V, false, true>(P, nullptr, nullptr, centers, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, inv_r, out, st)
    switch (Cout) {
        case 32: DPM_GF(32, 1);
        case 64: DPM_GF(64, 1);
        case 128: DPM_GF(128, 1);
        case 256: DPM_GF(256, 1);
        case 512: DPM_GF(512, 2);
        default: return DPM_EUNSUPPORTED;
    }
#undef DPM_GF
}

extern "C" int dpm_group_gather_ln_max_folded(const float *P, const float *centers, const int32_t *idx, const float *W_rel,
                                              int ldw_rel, const float *gamma, const float *beta, int B, int N, int S, int K,
                                              int Cout, double radius, float *out, dpm_stream_t stream) {
    return gather_folded(P, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, Cout, radius, out, stream, false);
}

// dpm_group_gather_ln_max_folded for a layer whose mean removal sits in its weights: the caller PROMISES that every column of
// [W_f | W_rel] and the bias have zero mean over the Cout output channels (W' = (I - 11^T / Cout) W), so that the rows of P and
// the centre term have zero mean over their channels; the kernel then computes LayerNorm's variance from the rows as they are.
// AND that channel c of [W_f | W_rel | bias] has been multiplied by sign(gamma_c) (+1 for gamma_c = 0): the kernel applies
// |gamma|, beta and the ReLU to the maximum over the neighbours.  With weights that do not keep the promises the result is a
// LayerNorm without its mean removal / with |gamma| in place of gamma.
extern "C" int dpm_group_gather_ln_max_centred(const float *P, const float *centers, const int32_t *idx, const float *W_rel,
                                               int ldw_rel, const float *gamma, const float *beta, int B, int N, int S, int K,
                                               int Cout, double radius, float *out, dpm_stream_t stream) {
    return gather_folded(P, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, Cout, radius, out, stream, true);
}

static int gather_affine(const float *A, const float *cvec, const float *xyz, const float *centers, const int32_t *idx,
                         const float *W_rel, int ldw_rel, const float *gamma, const float *beta, int B, int N, int S, int K,
                         int Cout, double radius, float *out, dpm_stream_t stream, bool centred) {
    DPM_CHECK_ARG(A && cvec && xyz && centers && idx && W_rel && gamma && beta && out);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && ldw_rel >= 3 && radius > 0.0 && ((uintptr_t)out & 15) == 0);
    hipStream_t st = (hipStream_t)stream;
    const float inv_r = 1.0f / (float)radius;
#define DPM_GA(C, V)                                                                                                              \
    return centred ? launch_gather<C, V, true, true, true>(nullptr, A, cvec, xyz, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, inv_r, out, st) \
                   : launch_gather<C, V, true, true>(nullptr, A, cvec, xyz, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, inv_r, out, st)
    switch (Cout) {
        case 32: DPM_GA(32, 1);
        case 64: DPM_GA(64, 1);
        case 128: DPM_GA(128, 1);
        default: return DPM_EUNSUPPORTED;
    }
#undef DPM_GA
}

extern "C" int dpm_group_affine_ln_max(const float *A, const float *cvec, const float *xyz, const float *centers,
                                       const int32_t *idx, const float *W_rel, int ldw_rel, const float *gamma,
                                       const float *beta, int B, int N, int S, int K, int Cout, double radius,
                                       float *out, dpm_stream_t stream) {
    return gather_affine(A, cvec, xyz, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, Cout, radius, out, stream, false);
}

// dpm_group_affine_ln_max with the mean removal in the layer (see dpm_group_gather_ln_max_centred): the columns of A and of
// W_rel and the vector cvec have zero mean over the Cout channels.
extern "C" int dpm_group_affine_ln_max_centred(const float *A, const float *cvec, const float *xyz, const float *centers,
                                               const int32_t *idx, const float *W_rel, int ldw_rel, const float *gamma,
                                               const float *beta, int B, int N, int S, int K, int Cout, double radius,
                                               float *out, dpm_stream_t stream) {
    return gather_affine(A, cvec, xyz, centers, idx, W_rel, ldw_rel, gamma, beta, B, N, S, K, Cout, radius, out, stream, true);
}

// defined in encoder_ops.hip: generic VALU kernel for shapes the MFMA kernel does not cover
extern "C" int dpm_group_mlp_max_generic(const float *xyz, const float *fea, const float *centers, const int32_t *idx,
                                         const float *W, const float *bias, const float *gamma, const float *beta,
                                         int B, int N, int S, int K, int Cin, int Cout, double radius, float *out,
                                         dpm_stream_t stream);

extern "C" int dpm_group_mlp_max(const float *xyz, const float *fea, const float *centers, const int32_t *idx,
                                 const float *W, const float *bias, const float *gamma, const float *beta, int B,
                                 int N, int S, int K, int Cin, int Cout, double radius, float *out,
                                 dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && fea && centers && idx && W && bias && gamma && beta && out);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && K <= 64 && Cin >= 1 && Cout >= 1 && radius > 0.0);
    hipStream_t st = (hipStream_t)stream;
    const float inv_r = 1.0f / (float)radius;
    if ((K == 16 || K == 32) && Cin % 4 == 0 && ((uintptr_t)fea & 15) == 0) {
        if (Cout == 32 && Cin == 32) return launch_wave<32, 32, false>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, inv_r, out, st, nullptr, nullptr);
        if (Cout == 64 && Cin == 32) return launch_wave<64, 32, false>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, inv_r, out, st, nullptr, nullptr);
        if (Cout == 64 && Cin == 64) return launch_wave<64, 64, false>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, inv_r, out, st, nullptr, nullptr);
        switch (Cout) {
            case 32: return launch<32, 4, 1>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st);
            case 64: return launch<64, 2, 2>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st);
            case 128: return launch<128, 2, 2>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st);
            case 256: return launch<256, 1, 4>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st);
            case 512: return launch<512, 1, 4>(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st);
            default: break;
        }
    }
    return dpm_group_mlp_max_generic(xyz, fea, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, Cout, radius, out,
                                     stream);
}

// SetAbstraction of the FIRST stage with Encoder.point_mlp0 folded in: the per-point input features
// fea = W0 xyz + b0 (W0 (Cin,3), b0 (Cin)) are evaluated inside the gather instead of being read.
extern "C" int dpm_group_mlp_max_from_xyz(const float *xyz, const float *W0, const float *b0, const float *centers,
                                          const int32_t *idx, const float *W, const float *bias, const float *gamma,
                                          const float *beta, int B, int N, int S, int K, int Cin, int Cout,
                                          double radius, float *out, dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && W0 && b0 && centers && idx && W && bias && gamma && beta && out);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && Cin >= 1 && radius > 0.0);
    if (!(K == 16 || K == 32) || Cin % 4 != 0) return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const float inv_r = 1.0f / (float)radius;
    if (Cout == 32 && Cin == 16) return launch_wave<32, 16, true>(xyz, nullptr, centers, idx, W, bias, gamma, beta, B, N, S, K, inv_r, out, st, W0, b0);
    switch (Cout) {
        case 32: return launch<32, 4, 1>(xyz, nullptr, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st, W0, b0);
        case 64: return launch<64, 2, 2>(xyz, nullptr, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st, W0, b0);
        case 128: return launch<128, 2, 2>(xyz, nullptr, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st, W0, b0);
        default: return DPM_EUNSUPPORTED;
    }
}
