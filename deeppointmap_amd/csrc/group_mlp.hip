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

    const int b = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w / WN, wn = w % WN;
    const int C3 = Cin + 3;
    const int cpb = TM / K;                       // centres per workgroup (2 or 4)
    const int s0 = blockIdx.x * cpb;              // first centre
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
    const bool fvec = fea && (Cin % 4 == 0) && ((reinterpret_cast<uintptr_t>(fea) & 15) == 0);
    const bool wvec = (C3 % 4 == 0) && ((reinterpret_cast<uintptr_t>(W) & 15) == 0);

    auto load_g = [&](int row, int k) -> float4 {  // 4 consecutive input channels of gathered row `row`
        float v[4] = {0.f, 0.f, 0.f, 0.f};
        const int n = s_nidx[row];
        if (fvec && k + 3 < Cin) {
            const float4 q = *reinterpret_cast<const float4 *>(fea + (size_t)n * Cin + k);
            return q;
        }
        float px = 0.f, py = 0.f, pz = 0.f;
        if (!fea) px = xyz[(size_t)n * 3], py = xyz[(size_t)n * 3 + 1], pz = xyz[(size_t)n * 3 + 2];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = k + j;
            if (c < Cin) v[j] = fea ? fea[(size_t)n * Cin + c]
                                    : fmaf(W0[3 * c + 2], pz, fmaf(W0[3 * c + 1], py, fmaf(W0[3 * c], px, b0[c])));
            else if (c < C3) v[j] = (xyz[(size_t)n * 3 + (c - Cin)] - s_ctr[row / K][c - Cin]) * inv_r;
        }
        return make_float4(v[0], v[1], v[2], v[3]);
    };
    auto load_w = [&](int row, int k) -> float4 {
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *p = W + (size_t)row * C3 + k;
        if (wvec && k + 3 < C3) return *reinterpret_cast<const float4 *>(p);
        if (k < C3) v.x = p[0];
        if (k + 1 < C3) v.y = p[1];
        if (k + 2 < C3) v.z = p[2];
        if (k + 3 < C3) v.w = p[3];
        return v;
    };

    f32x4 acc[MB][NB];
#pragma unroll
    for (int i = 0; i < MB; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    float4 gr[2], wr[PW];
#pragma unroll
    for (int p = 0; p < 2; ++p) gr[p] = load_g(p * 32 + sr, sk);
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
            for (int p = 0; p < 2; ++p) gr[p] = load_g(p * 32 + sr, k0 + KT + sk);
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

}  // namespace

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
    if (K == 16 || K == 32) {
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
    if (!(K == 16 || K == 32)) return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const float inv_r = 1.0f / (float)radius;
    switch (Cout) {
        case 32: return launch<32, 4, 1>(xyz, nullptr, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st, W0, b0);
        case 64: return launch<64, 2, 2>(xyz, nullptr, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st, W0, b0);
        case 128: return launch<128, 2, 2>(xyz, nullptr, centers, idx, W, bias, gamma, beta, B, N, S, K, Cin, inv_r, out, st, W0, b0);
        default: return DPM_EUNSUPPORTED;
    }
}
