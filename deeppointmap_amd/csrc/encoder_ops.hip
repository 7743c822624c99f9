// Dense / gather kernels of the encoder (and the small linear algebra the decoder shares):
// input staging, grouped MLP + LayerNorm + ReLU + max (SetAbstraction / LocalAggregation),
// linear, LayerNorm, 3-NN feature propagation.  fp32 throughout, like the reference.
// Compiled with -ffp-contract=off: every fused multiply-add is an explicit fmaf().
#include "dpm_common.h"
#include "topk_emulate.h"

namespace {

// ------------------------------------------------------------------------------------------
// (B,C,N) channel-first + padding mask -> xyz (B,N,3), lengths (B)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void prepare_points_kernel(const float *__restrict__ pcf,
                                                             const uint8_t *__restrict__ pad, int C, int N,
                                                             float *__restrict__ xyz,
                                                             int32_t *__restrict__ lengths) {
    // 256 points per block: the three coordinate planes are read coalesced, interleaved through LDS, and the
    // 768 output floats are written coalesced
    __shared__ float tile[3 * 256];
    const int b = blockIdx.y, t = threadIdx.x;
    const int i0 = blockIdx.x * 256, i = i0 + t;
    const float *p = pcf + (size_t)b * C * N;
    if (i < N) tile[3 * t] = p[i], tile[3 * t + 1] = p[(size_t)N + i], tile[3 * t + 2] = p[2 * (size_t)N + i];
    __syncthreads();
    const int n = min(256, N - i0);
    float *o = xyz + ((size_t)b * N + i0) * 3;
    for (int e = t; e < 3 * n; e += 256) o[e] = tile[e];
    (void)pad, (void)lengths;
}

// lengths[b] = number of unpadded points of frame b: one workgroup per frame.  (Counting inside the kernel above
// meant ~1000 same-address device atomics per frame -- they, not the 100 MB of coordinates, set its run time.)
// 256 threads, not 1024: inside the stream pipeline a 16-wave workgroup waits for half a compute unit to fall free
// (5 us alone, 90 us on average between the other stages' workgroups); four waves find room at once.
__global__ __launch_bounds__(256) void count_valid_kernel(const uint8_t *__restrict__ pad, int N,
                                                          int32_t *__restrict__ lengths) {
    __shared__ int s_w[4];
    const int b = blockIdx.x, t = threadIdx.x;
    const uint8_t *p = pad + (size_t)b * N;
    int c = 0;
    if ((N & 15) == 0 && ((uintptr_t)p & 15) == 0) {
        const uint4 *q = reinterpret_cast<const uint4 *>(p);
#pragma unroll 8
        for (int i = t; i < N / 16; i += 256) {
            const uint4 v = q[i];  // a bool tensor holds 0 / 1 bytes: padded = number of non-zero bytes
            c += 16 - (__popc(v.x & 0x01010101u) + __popc(v.y & 0x01010101u) + __popc(v.z & 0x01010101u) +
                       __popc(v.w & 0x01010101u));
        }
    } else {
        for (int i = t; i < N; i += 256) c += p[i] ? 0 : 1;
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off, 64);
    if ((t & 63) == 0) s_w[t >> 6] = c;
    __syncthreads();
    if (t == 0) lengths[b] = (s_w[0] + s_w[1]) + (s_w[2] + s_w[3]);
}

__global__ __launch_bounds__(256) void to_channel_first_kernel(const float *__restrict__ x, int R, int C,
                                                               float *__restrict__ out, int ldo) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const float *xi = x + (size_t)b * R * C;
    float *oi = out + (size_t)b * C * ldo;   // output rows ldo >= R floats apart
    const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int k = ty; k < 32; k += 8)
        if (r0 + k < R && c0 + tx < C) tile[k][tx] = xi[(size_t)(r0 + k) * C + c0 + tx];
    __syncthreads();
    for (int k = ty; k < 32; k += 8)
        if (c0 + k < C && r0 + tx < R) oi[(size_t)(c0 + k) * ldo + r0 + tx] = tile[tx][k];
}

// ------------------------------------------------------------------------------------------
// grouped MLP: one workgroup per centre.  v1 (plain VALU): gather K neighbour rows into LDS,
// y[r][c] = b[c] + sum_k g[r][k] W[c][k], LayerNorm over c, ReLU, max over r.  Generic fallback for
// shapes the MFMA kernel (group_mlp.hip) does not cover.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void group_mlp_max_kernel(
    const float *__restrict__ xyz_all, const float *__restrict__ fea_all, const float *__restrict__ ctr_all,
    const int32_t *__restrict__ idx_all, const float *__restrict__ W, const float *__restrict__ bias,
    const float *__restrict__ gamma, const float *__restrict__ beta, int N, int S, int K, int Cin, int Cout,
    float inv_r, float *__restrict__ out_all) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int C3 = Cin + 3;
    float *G = smem;                    // [K][C3]
    float *Y = G + K * C3;              // [K][Cout]
    float *mean = Y + K * Cout;         // [K]
    float *rstd = mean + K;             // [K]
    int *nidx = (int *)(rstd + K);      // [K]
    const int b = blockIdx.y, s = blockIdx.x, t = threadIdx.x;
    const float *xyz = xyz_all + (size_t)b * N * 3;
    const float *fea = fea_all + (size_t)b * N * Cin;
    const float *ctr = ctr_all + ((size_t)b * S + s) * 3;
    const int32_t *idx = idx_all + ((size_t)b * S + s) * K;

    if (t < K) nidx[t] = min(max(idx[t], 0), N - 1);
    __syncthreads();
    for (int e = t; e < K * Cin; e += 256) {
        const int r = e / Cin, k = e - r * Cin;
        G[r * C3 + k] = fea[(size_t)nidx[r] * Cin + k];
    }
    if (t < K * 3) {
        const int r = t / 3, a = t - r * 3;
        G[r * C3 + Cin + a] = (xyz[(size_t)nidx[r] * 3 + a] - ctr[a]) * inv_r;
    }
    __syncthreads();
    for (int o = t; o < K * Cout; o += 256) {
        const int r = o / Cout, c = o - r * Cout;
        float acc = bias[c];
        const float *g = G + r * C3;
        for (int k = 0; k < C3; ++k) acc = fmaf(g[k], W[(size_t)c * C3 + k], acc);
        Y[o] = acc;
    }
    __syncthreads();
    const int w = t >> 6, lane = t & 63;
    for (int r = w; r < K; r += 4) {
        float sum = 0.f;
        for (int c = lane; c < Cout; c += 64) sum += Y[r * Cout + c];
        const float mu = wave_sum(sum) / (float)Cout;
        float sq = 0.f;
        for (int c = lane; c < Cout; c += 64) {
            const float d = Y[r * Cout + c] - mu;
            sq = fmaf(d, d, sq);
        }
        const float var = wave_sum(sq) / (float)Cout;
        if (lane == 0) mean[r] = mu, rstd[r] = rsqrtf(var + 1e-5f);
    }
    __syncthreads();
    float *out = out_all + ((size_t)b * S + s) * Cout;
    for (int c = t; c < Cout; c += 256) {
        const float gm = gamma[c], bt = beta[c];
        float m = 0.f;  // ReLU output is >= 0, so 0 is the identity of the max
        for (int r = 0; r < K; ++r) m = fmaxf(m, fmaf((Y[r * Cout + c] - mean[r]) * rstd[r], gm, bt));
        out[c] = m;
    }
}

// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DPM_ACT_RELU) return fmaxf(v, 0.f);
    if (act == DPM_ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// ------------------------------------------------------------------------------------------
// out = act(LN(x + pre) * gamma + beta + post); one wave per row
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_kernel(const float *__restrict__ X, int ldx,
                                                        const float *__restrict__ pre,
                                                        const float *__restrict__ gamma,
                                                        const float *__restrict__ beta,
                                                        const float *__restrict__ post,
                                                        float *__restrict__ out, int ldo, int R, int C, int act) {
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (r >= R) return;
    const float *x = X + (size_t)r * ldx;
    const float *pr = pre ? pre + (size_t)r * C : nullptr;
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += x[c] + (pr ? pr[c] : 0.f);
    const float mu = wave_sum(sum) / (float)C;
    float sq = 0.f;
    for (int c = lane; c < C; c += 64) {
        const float d = x[c] + (pr ? pr[c] : 0.f) - mu;
        sq = fmaf(d, d, sq);
    }
    const float rs = rsqrtf(wave_sum(sq) / (float)C + 1e-5f);
    const float *po = post ? post + (size_t)r * C : nullptr;
    float *o = out + (size_t)r * ldo;
    for (int c = lane; c < C; c += 64) {
        float v = fmaf((x[c] + (pr ? pr[c] : 0.f) - mu) * rs, gamma[c], beta[c]);
        if (po) v += po[c];
        o[c] = apply_act(v, act);
    }
}

// Vectorised variant for the widths of the path (C = 4 * G * V): G lanes share a row (64 / G rows per wave), every
// lane keeps its V float4 in registers, so x (+ pre) is read exactly once, and the two row sums are DPP / shuffle
// reductions inside the lane group.  Same two-pass arithmetic as the generic kernel above.
// (in-row steps as v_add_f32 with a DPP operand: one instruction + the two wait states of a DPP read, where
// update_dpp compiles to copy + s_nop + mov_dpp + add)
#define DPM_ADD_DPP(ctrl) "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 " ctrl " row_mask:0xf bank_mask:0xf\n\t"
template <int G>
__device__ __forceinline__ float group_sum(float v) {
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

template <int G, int V>
__global__ __launch_bounds__(256) void layernorm_vec_kernel(const float *__restrict__ X, int ldx,
                                                            const float *__restrict__ pre,
                                                            const float *__restrict__ gamma,
                                                            const float *__restrict__ beta,
                                                            const float *__restrict__ post,
                                                            float *__restrict__ out, int ldo, int R, int act) {
    constexpr int C = 4 * G * V, RPW = 64 / G;
    const int lane = threadIdx.x & 63, gl = lane % G;
    const int r = min((blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + lane / G, R - 1);  // surplus lanes redo the last row
    const float *x = X + (size_t)r * ldx;
    float4 v[V];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = 4 * (gl + G * j);
        v[j] = *reinterpret_cast<const float4 *>(x + c);
        if (pre) {
            const float4 p = *reinterpret_cast<const float4 *>(pre + (size_t)r * C + c);
            v[j].x += p.x, v[j].y += p.y, v[j].z += p.z, v[j].w += p.w;
        }
        sum += (v[j].x + v[j].y) + (v[j].z + v[j].w);
    }
    const float mu = group_sum<G>(sum) / (float)C;
    float sq = 0.f;
#pragma unroll
    for (int j = 0; j < V; ++j) {
        v[j].x -= mu, v[j].y -= mu, v[j].z -= mu, v[j].w -= mu;
        sq = fmaf(v[j].x, v[j].x, fmaf(v[j].y, v[j].y, fmaf(v[j].z, v[j].z, fmaf(v[j].w, v[j].w, sq))));
    }
    const float rs = rsqrtf(group_sum<G>(sq) / (float)C + 1e-5f);
#pragma unroll
    for (int j = 0; j < V; ++j) {
        const int c = 4 * (gl + G * j);
        const float4 g4 = *reinterpret_cast<const float4 *>(gamma + c), b4 = *reinterpret_cast<const float4 *>(beta + c);
        float4 o = make_float4(fmaf(v[j].x * rs, g4.x, b4.x), fmaf(v[j].y * rs, g4.y, b4.y), fmaf(v[j].z * rs, g4.z, b4.z),
                               fmaf(v[j].w * rs, g4.w, b4.w));
        if (post) {
            const float4 p = *reinterpret_cast<const float4 *>(post + (size_t)r * C + c);
            o.x += p.x, o.y += p.y, o.z += p.z, o.w += p.w;
        }
        o.x = apply_act(o.x, act), o.y = apply_act(o.y, act), o.z = apply_act(o.z, act), o.w = apply_act(o.w, act);
        *reinterpret_cast<float4 *>(out + (size_t)r * ldo + c) = o;
    }
}

// ------------------------------------------------------------------------------------------
// 3-NN inverse-distance interpolation + concat [skip | interpolated]; one WAVE per fine point.
// Lane j looks at the coarse points j, j + 64, ... and keeps its own three nearest (ascending (distance, index):
// strict <, the earlier index stays ahead on ties, as topk(k=3, largest=False) on the reference's distance row);
// the wave's three nearest are then the heads of three rounds of a wave arg-min, the winning lane popping its head.
// Distances in the expanded form -2ab + |a|^2 + |b|^2 like the reference (pointnext.py:205, utils.py:288-295).
// When coarse points at EXACTLY the third distance are left out (mirror-symmetric clouds: two key points equally far
// from a third), which of them torch.topk keeps is the data movement of std::nth_element (3 * 64 > S, every shipped
// level) or of std::partial_sort's heap-select: the row of S (distance, index) pairs goes to LDS (`rows` != 0: the launch
// reserved 4 * S pairs) and lane 0 replays libstdc++ on it (topk_emulate.h) -- rare, and S is 16 or 64.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void three_interp_cat_kernel(
    const float *__restrict__ xyz1_all, const float *__restrict__ xyz2_all, const int32_t *__restrict__ len2,
    const float *__restrict__ fea1_all, const float *__restrict__ fea2_all, int N, int S, int D1, int D2,
    float *__restrict__ out_all, int rows) {
    extern __shared__ __attribute__((aligned(8))) unsigned char s_rows_raw[];
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (n >= N) return;
    const float *p = xyz1_all + ((size_t)b * N + n) * 3;
    const float *q = xyz2_all + (size_t)b * S * 3;
    const float *f1 = fea1_all + ((size_t)b * N + n) * D1;
    const float *f2 = fea2_all + (size_t)b * S * D2;
    float *out = out_all + ((size_t)b * N + n) * (D1 + D2);
    for (int c = lane; c < D1; c += 64) out[c] = f1[c];
    if (S == 1) {
        for (int c = lane; c < D2; c += 64) out[D1 + c] = f2[c];
        return;
    }
    const int ls = min(max(len2[b], 0), S);
    const float px = p[0], py = p[1], pz = p[2];
    const float pp = (px * px + py * py) + pz * pz;  // torch.sum(p**2,-1): sequential, unfused
    const float INF = __builtin_inff();
    float d0 = INF, d1 = INF, d2 = INF;
    int i0 = 0x7fffffff, i1 = 0x7fffffff, i2 = 0x7fffffff;
    for (int j = lane; j < ls; j += 64) {
        const float x = q[3 * j], y = q[3 * j + 1], z = q[3 * j + 2];
        float d = -2.f * fmaf(pz, z, fmaf(py, y, px * x));
        d += pp;
        d += (x * x + y * y) + z * z;
        if (d < d2) {
            if (d < d1) {
                d2 = d1, i2 = i1;
                if (d < d0) d1 = d0, i1 = i0, d0 = d, i0 = j;
                else d1 = d, i1 = j;
            } else d2 = d, i2 = j;
        }
    }
    float bd[3];
    int bi[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        // smallest head over the lanes, ties to the smallest index
        float md = d0;
        int mi = i0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od = __shfl_xor(md, off, 64);
            const int oi = __shfl_xor(mi, off, 64);
            if (od < md || (od == md && oi < mi)) md = od, mi = oi;
        }
        bd[k] = md, bi[k] = mi == 0x7fffffff ? 0 : mi;
        if (i0 == mi && mi != 0x7fffffff) d0 = d1, i0 = i1, d1 = d2, i1 = i2, d2 = INF, i2 = 0x7fffffff;  // pop
    }
    if (rows && ls >= 3) {
        // does a point at exactly the third distance stay outside the three?
        const float t3 = bd[2];
        int eq = 0;
        for (int j = lane; j < ls; j += 64) {
            const float x = q[3 * j], y = q[3 * j + 1], z = q[3 * j + 2];
            float d = -2.f * fmaf(pz, z, fmaf(py, y, px * x));
            d += pp;
            d += (x * x + y * y) + z * z;
            eq += d == t3;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) eq += __shfl_xor(eq, off, 64);
        const int taken = 1 + (bd[1] == t3) + (bd[0] == t3);
        if (eq > taken) {  // wave-uniform
            VI *row = reinterpret_cast<VI *>(s_rows_raw) + (size_t)(threadIdx.x >> 6) * S;
            for (int j = lane; j < S; j += 64) {
                float d = INF;  // padded coarse points are pushed far away by the reference: above every real one, equal among themselves
                if (j < ls) {
                    const float x = q[3 * j], y = q[3 * j + 1], z = q[3 * j + 2];
                    d = -2.f * fmaf(pz, z, fmaf(py, y, px * x));
                    d += pp;
                    d += (x * x + y * y) + z * z;
                }
                row[j].v = d, row[j].i = j;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) {
                if (3 * 64 <= S) vi_heap_select<false>(row, 0, 3, S);  // torch.topk: partial_sort when k * 64 <= n
                else vi_nth_element<false>(row, S, 2);
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // the three survivors, ascending like the reference's sorted result (values tie, the set is what matters)
            VI a = row[0], b2 = row[1], c2 = row[2];
            if (b2.v < a.v) { VI t = a; a = b2; b2 = t; }
            if (c2.v < b2.v) { VI t = b2; b2 = c2; c2 = t; }
            if (b2.v < a.v) { VI t = a; a = b2; b2 = t; }
            bd[0] = a.v, bi[0] = a.i, bd[1] = b2.v, bi[1] = b2.i, bd[2] = c2.v, bi[2] = c2.i;
        }
    }
    float w0 = 1.f / fmaxf(bd[0], 1e-8f), w1 = 1.f / fmaxf(bd[1], 1e-8f), w2 = 1.f / fmaxf(bd[2], 1e-8f);
    if (ls < 3) w2 = 0.f;
    if (ls < 2) w1 = 0.f;
    const float ws = w0 + w1 + w2;
    w0 /= ws, w1 /= ws, w2 /= ws;
    for (int c = lane; c < D2; c += 64)
        out[D1 + c] = f2[(size_t)bi[0] * D2 + c] * w0 + f2[(size_t)bi[1] * D2 + c] * w1 + f2[(size_t)bi[2] * D2 + c] * w2;
}

}  // namespace

extern "C" int dpm_prepare_points(const float *points_cf, const uint8_t *padding, int B, int C, int N, float *xyz,
                                  int32_t *lengths, dpm_stream_t stream) {
    DPM_CHECK_ARG(points_cf && padding && xyz && lengths);
    DPM_CHECK_ARG(B >= 1 && C >= 3 && N >= 1);
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(count_valid_kernel, dim3(B), dim3(256), 0, st, padding, N, lengths);
    hipLaunchKernelGGL(prepare_points_kernel, dim3(dpm_cdiv(N, 256), B), dim3(256), 0, st, points_cf, padding, C, N,
                       xyz, lengths);
    return dpm_launch_status();
}

namespace {
// ------------------------------------------------------------------------------------------
// Bookkeeping that used to be a string of tiny tensor ops per step.
// ------------------------------------------------------------------------------------------
constexpr int MAX_NESTED = 8;
struct NestedLevels {
    int n;
    int npoint[MAX_NESTED];    // K of level i (descending)
    long long off[MAX_NESTED]; // element offset of level i in the packed outputs (per batch: sum over earlier levels of B * K)
};

// Levels below the first are prefixes of the first level's pick sequence (Encoder.nested_fps): per level the first K
// picked coordinates, idx = position (or -1 past the frame's valid count) and the clamped length.
__global__ __launch_bounds__(256) void nested_levels_kernel(const float *__restrict__ xyz0, const int32_t *__restrict__ len0,
                                                            int B, int K0, NestedLevels lv, float *__restrict__ xyz_out,
                                                            int32_t *__restrict__ idx_out, int32_t *__restrict__ len_out) {
    const int b = blockIdx.y, lvl = blockIdx.z, K = lv.npoint[lvl];
    const int l0 = len0[b];
    const int len = min(l0, K);
    float *xo = xyz_out + 3 * (lv.off[lvl] + (long long)b * K);
    int32_t *io = idx_out + lv.off[lvl] + (long long)b * K;
    const float *xi = xyz0 + (size_t)b * K0 * 3;
    for (int j = blockIdx.x * 256 + threadIdx.x; j < K; j += gridDim.x * 256) {
        xo[3 * j] = xi[3 * j], xo[3 * j + 1] = xi[3 * j + 1], xo[3 * j + 2] = xi[3 * j + 2];
        io[j] = j < len ? j : -1;
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) len_out[lvl * B + b] = len;
}

// The encoder's return values in one pass: coor (B,3,S), feat (B,C,S), padding (B,S) and, when asked for, the unified
// descriptor (B,C+3,S) = [feat ; coor * scale] of ExtractionThread.process (odometry.py:47-49).
__global__ __launch_bounds__(256) void emit_descriptors_kernel(const float *__restrict__ xyz, const float *__restrict__ fea,
                                                               const int32_t *__restrict__ lengths, int S, int C, float scale,
                                                               float *__restrict__ coor, float *__restrict__ feat,
                                                               uint8_t *__restrict__ padding, float *__restrict__ desc) {
    __shared__ float tile[32][33];
    const int b = blockIdx.z, r0 = blockIdx.y * 32, c0 = blockIdx.x * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const int CT = C + 3;  // channels of the virtual row [fea ; xyz]
    for (int k = ty; k < 32; k += 8) {
        const int r = r0 + k, c = c0 + tx;
        if (r < S && c < CT) tile[k][tx] = c < C ? fea[((size_t)b * S + r) * C + c] : xyz[((size_t)b * S + r) * 3 + (c - C)];
    }
    __syncthreads();
    for (int k = ty; k < 32; k += 8) {
        const int c = c0 + k, r = r0 + tx;
        if (c < CT && r < S) {
            const float v = tile[tx][k];
            if (c < C) feat[((size_t)b * C + c) * S + r] = v;
            else coor[((size_t)b * 3 + (c - C)) * S + r] = v;
            if (desc) desc[((size_t)b * CT + c) * S + r] = c < C ? v : v * scale;
        }
    }
    if (blockIdx.x == 0)
        for (int k = threadIdx.x; k < 32; k += 256)
            if (r0 + k < S) padding[(size_t)b * S + r0 + k] = (r0 + k) >= lengths[b];
}

// out[p, m, c] = src[index[p] * frame_stride + m * ld + c]: rows of selected frames, contiguous
__global__ __launch_bounds__(256) void gather_frames_kernel(const float *__restrict__ src, long long frame_stride, int rows,
                                                            int ld, int cols, const int32_t *__restrict__ index,
                                                            float *__restrict__ out) {
    const int p = blockIdx.y;
    const float *s = src + (size_t)index[p] * frame_stride;
    float *o = out + (size_t)p * rows * cols;
    const long long n = (long long)rows * cols;
    if (ld == cols && (cols & 3) == 0 && (((uintptr_t)s | (uintptr_t)o) & 15) == 0) {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n / 4; i += (long long)gridDim.x * 256)
            ((float4 *)o)[i] = ((const float4 *)s)[i];
    } else {
        for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
            const int m = (int)(i / cols), c = (int)(i % cols);
            o[i] = s[(size_t)m * ld + c];
        }
    }
}
}  // namespace

extern "C" int dpm_nested_levels(const float *xyz0, const int32_t *len0, int B, int K0, int n_levels, const int32_t *npoint,
                                 float *xyz_out, int32_t *idx_out, int32_t *len_out, dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz0 && len0 && npoint && xyz_out && idx_out && len_out && B >= 1 && K0 >= 1);
    DPM_CHECK_ARG(n_levels >= 1 && n_levels <= MAX_NESTED);
    NestedLevels lv;
    lv.n = n_levels;
    long long off = 0;
    int kmax = 1;
    for (int i = 0; i < n_levels; ++i) {
        DPM_CHECK_ARG(npoint[i] >= 1 && npoint[i] <= K0);
        lv.npoint[i] = npoint[i], lv.off[i] = off;
        off += (long long)B * npoint[i];
        kmax = npoint[i] > kmax ? npoint[i] : kmax;
    }
    hipLaunchKernelGGL(nested_levels_kernel, dim3(dpm_cdiv(kmax, 256), B, n_levels), dim3(256), 0, (hipStream_t)stream, xyz0,
                       len0, B, K0, lv, xyz_out, idx_out, len_out);
    return dpm_launch_status();
}

extern "C" int dpm_emit_descriptors(const float *xyz, const float *fea, const int32_t *lengths, int B, int S, int C,
                                    double coor_scale, float *coor, float *feat, uint8_t *padding, float *desc,
                                    dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && fea && lengths && coor && feat && padding && B >= 1 && S >= 1 && C >= 1);
    hipLaunchKernelGGL(emit_descriptors_kernel, dim3(dpm_cdiv(C + 3, 32), dpm_cdiv(S, 32), B), dim3(256), 0,
                       (hipStream_t)stream, xyz, fea, lengths, S, C, (float)coor_scale, coor, feat, padding, desc);
    return dpm_launch_status();
}

extern "C" int dpm_gather_frames(const float *src, long long frame_stride, int rows, int ld, int cols, const int32_t *index,
                                 int n, float *out, dpm_stream_t stream) {
    DPM_CHECK_ARG(src && index && out && rows >= 1 && cols >= 1 && ld >= cols && n >= 1);
    const long long per = (long long)rows * cols;
    hipLaunchKernelGGL(gather_frames_kernel, dim3((unsigned)((per / 4 + 255) / 256 < 64 ? ((per / 4 + 255) / 256 ? (per / 4 + 255) / 256 : 1) : 64), n),
                       dim3(256), 0, (hipStream_t)stream, src, frame_stride, rows, ld, cols, index, out);
    return dpm_launch_status();
}

extern "C" int dpm_to_channel_first_ld(const float *x, int B, int R, int C, float *out, int ldo, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && out && B >= 1 && R >= 1 && C >= 1 && ldo >= R);
    hipLaunchKernelGGL(to_channel_first_kernel, dim3(dpm_cdiv(C, 32), dpm_cdiv(R, 32), B), dim3(256), 0,
                       (hipStream_t)stream, x, R, C, out, ldo);
    return dpm_launch_status();
}

extern "C" int dpm_to_channel_first(const float *x, int B, int R, int C, float *out, dpm_stream_t stream) {
    return dpm_to_channel_first_ld(x, B, R, C, out, R, stream);
}

extern "C" int dpm_group_mlp_max_generic(const float *xyz, const float *fea, const float *centers, const int32_t *idx,
                                         const float *Wt, const float *bias, const float *gamma, const float *beta, int B,
                                         int N, int S, int K, int Cin, int Cout, double radius, float *out,
                                         dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && fea && centers && idx && Wt && bias && gamma && beta && out);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && K <= 64 && Cin >= 1 && Cout >= 1 && radius > 0.0);
    const size_t lds = sizeof(float) * ((size_t)K * (Cin + 3) + (size_t)K * Cout + 2 * K) + sizeof(int) * K;
    if (lds > 160 * 1024) return DPM_EUNSUPPORTED;
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)group_mlp_max_kernel,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
    }
    hipLaunchKernelGGL(group_mlp_max_kernel, dim3(S, B), dim3(256), lds, (hipStream_t)stream, xyz, fea, centers, idx,
                       Wt, bias, gamma, beta, N, S, K, Cin, Cout, 1.0f / (float)radius, out);
    return dpm_launch_status();
}

extern "C" int dpm_layernorm(const float *x, int ldx, const float *pre, const float *gamma, const float *beta,
                             const float *post, float *out, int ldo, int R, int C, int act, dpm_stream_t stream) {
    DPM_CHECK_ARG(x && gamma && beta && out && R >= 1 && C >= 1 && ldx >= C && ldo >= C);
    DPM_CHECK_ARG(act >= DPM_ACT_NONE && act <= DPM_ACT_SIGMOID);
    hipStream_t st = (hipStream_t)stream;
    auto al = [](const void *p) { return ((uintptr_t)p & 15) == 0; };
    const bool vec = ldx % 4 == 0 && ldo % 4 == 0 && al(x) && al(out) && al(gamma) && al(beta) && al(pre) && al(post);
#define DPM_LN(G, V)                                                                                              \
    hipLaunchKernelGGL((layernorm_vec_kernel<G, V>), dim3(dpm_cdiv(R, 4 * (64 / G))), dim3(256), 0, st, x, ldx, pre, gamma, \
                       beta, post, out, ldo, R, act)
    if (vec && C == 32) DPM_LN(8, 1);
    else if (vec && C == 64) DPM_LN(16, 1);
    else if (vec && C == 128) DPM_LN(32, 1);
    else if (vec && C == 256) DPM_LN(64, 1);
    else if (vec && C == 512) DPM_LN(64, 2);
    else if (vec && C == 1024) DPM_LN(64, 4);
    else if (vec && C == 2048) DPM_LN(64, 8);
    else
        hipLaunchKernelGGL(layernorm_kernel, dim3(dpm_cdiv(R, 4)), dim3(256), 0, st, x, ldx, pre, gamma, beta, post, out, ldo,
                           R, C, act);
#undef DPM_LN
    return dpm_launch_status();
}

extern "C" int dpm_three_interp_cat(const float *xyz1, const float *xyz2, const int32_t *lengths2, const float *fea1,
                                    const float *fea2, int B, int N, int S, int D1, int D2, float *out,
                                    dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz1 && xyz2 && lengths2 && fea1 && fea2 && out);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && D1 >= 0 && D2 >= 1);
    // rows of up to 1024 coarse points fit the tie replay's LDS scratch (4 waves x S pairs); larger ones keep the
    // smallest-index rule
    const int rows = S <= 1024 ? 1 : 0;
    hipLaunchKernelGGL(three_interp_cat_kernel, dim3(dpm_cdiv(N, 4), B), dim3(256), rows ? sizeof(VI) * 4 * (size_t)S : 0,
                       (hipStream_t)stream, xyz1, xyz2, lengths2, fea1, fea2, N, S, D1, D2, out, rows);
    return dpm_launch_status();
}

// an experimental build (-DDPM_EXPERIMENT: measurement switches read from the environment, dpm_common.h) says so: bench.py
// and the parity tests refuse a library whose version carries the flag
#ifdef DPM_EXPERIMENT
extern "C" int dpm_version(void) { return 1001 | DPM_VERSION_EXPERIMENT; }
#else
extern "C" int dpm_version(void) { return 1001; }
#endif

extern "C" const char *dpm_error_string(int status) {
    if (status == DPM_OK) return "ok";
    if (status == DPM_EINVAL) return "invalid argument";
    if (status == DPM_EUNSUPPORTED) return "unsupported shape";
    if (status > 0) return hipGetErrorString((hipError_t)status);
    return "unknown error";
}
