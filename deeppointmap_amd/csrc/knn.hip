// kNN-within-radius ("hybrid") neighbour grouping for gfx950.  Replaces Querier.hybrid_query /
// hybrid_query_t3d == pytorch3d.ops.knn_points + radius mask (reference
// network/encoder/utils.py:76-89,113-123).
//
// Semantics: the reference takes the K nearest points and then overwrites every slot whose
// squared distance exceeds r^2 with slot 0 (the nearest).  That equals "the K nearest among the
// points within r, padded with the nearest point", which is what is computed here: a scan keeps
// only candidates with d <= min(r^2, current K-th best), so the per-centre candidate list stays
// tiny and no (S,N) distance matrix is ever materialised.  Distances use the direct form
// (dx^2+dy^2+dz^2) like pytorch3d's kernel; the reference's CPU fallback uses the expanded form,
// which differs by a few 1e-7 -- tests compare index SETS with that borderline margin.
// Padded points (index >= lengths[b]) are ignored: the reference moves them to 3*max|coord|,
// which is never nearer than any valid point, so they can only ever be masked out.
//
// v1 structure: one wave handles CPW centres and streams all points of the frame (64 per step,
// coalesced); a wave-aggregated append (ballot + popcount) puts survivors into a per-centre LDS
// list; when the list is nearly full, or at the end, the K smallest are extracted by repeated
// wave arg-min (ties: smaller index).
#include "dpm_common.h"

namespace {

constexpr int CPW = 4;    // centres per wave
constexpr int WPB = 4;    // waves per block
constexpr int CAP = 512;  // candidate slots per centre
constexpr int KMAX = 64;

__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// Extract the k smallest (d asc, then index asc) of cd/ci[0..count) into od/oi[0..k); entries
// taken are overwritten with +inf.  All 64 lanes participate.  Returns nothing; k <= count.
__device__ __forceinline__ void select_smallest(volatile float *cd, volatile int *ci, int count, int k,
                                                volatile float *od, volatile int *oi) {
    const int lane = lane_id();
    for (int r = 0; r < k; ++r) {
        float bd = __builtin_inff();
        int bi = 0x7fffffff, bp = -1;
        for (int p = lane; p < count; p += 64) {
            const float d = cd[p];
            const int i = ci[p];
            if (d < bd || (d == bd && i < bi)) bd = d, bi = i, bp = p;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od_ = __shfl_xor(bd, off, 64);
            const int oi_ = __shfl_xor(bi, off, 64);
            const int op_ = __shfl_xor(bp, off, 64);
            if (od_ < bd || (od_ == bd && oi_ < bi)) bd = od_, bi = oi_, bp = op_;
        }
        if (lane == 0) {
            od[r] = bd;
            oi[r] = bi;
            if (bp >= 0) cd[bp] = __builtin_inff();
        }
        wave_mem_sync();
    }
}

__global__ __launch_bounds__(WPB * 64) void knn_hybrid_kernel(const float *__restrict__ points_all,
                                                              const int32_t *__restrict__ lengths,
                                                              const float *__restrict__ centers_all, int N,
                                                              int S, int K, float r2,
                                                              int32_t *__restrict__ idx_all) {
    __shared__ float s_d[WPB][CPW][CAP];
    __shared__ int s_i[WPB][CPW][CAP];
    __shared__ float s_td[WPB][KMAX];
    __shared__ int s_ti[WPB][KMAX];
    const int b = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s0 = (blockIdx.x * WPB + w) * CPW;
    if (s0 >= S) return;  // whole wave exits together; no block-wide barrier is used below
    const float *pts = points_all + (size_t)b * N * 3;
    const float *ctr = centers_all + (size_t)b * S * 3;
    const int len = min(max(lengths[b], 0), N);

    float cx[CPW], cy[CPW], cz[CPW], thr[CPW], gd[CPW];
    int gi[CPW], cnt[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        const int s = min(s0 + j, S - 1);
        cx[j] = ctr[3 * s], cy[j] = ctr[3 * s + 1], cz[j] = ctr[3 * s + 2];
        thr[j] = r2, gd[j] = __builtin_inff(), gi[j] = 0x7fffffff, cnt[j] = 0;
    }
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int base = 0; base < len; base += 64) {
        const int i = base + lane;
        const bool ok = i < len;
        float x = 0.f, y = 0.f, z = 0.f;
        if (ok) x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
#pragma unroll
        for (int j = 0; j < CPW; ++j) {
            const float dx = x - cx[j], dy = y - cy[j], dz = z - cz[j];
            const float d = ok ? fmaf(dz, dz, fmaf(dy, dy, dx * dx)) : __builtin_inff();
            if (d < gd[j]) gd[j] = d, gi[j] = i;
            const bool in = d <= thr[j];
            const unsigned long long m = __ballot(in);
            if (m) {
                if (in) {
                    const int pos = cnt[j] + __popcll(m & lt);
                    s_d[w][j][pos] = d;
                    s_i[w][j][pos] = i;
                }
                cnt[j] += __popcll(m);
                if (cnt[j] > CAP - 64) {  // compact: keep the K smallest, tighten the admission bound
                    wave_mem_sync();
                    const int k = min(K, cnt[j]);
                    select_smallest(s_d[w][j], s_i[w][j], cnt[j], k, s_td[w], s_ti[w]);
                    if (lane < k) s_d[w][j][lane] = s_td[w][lane], s_i[w][j][lane] = s_ti[w][lane];
                    wave_mem_sync();
                    cnt[j] = k;
                    if (k == K) thr[j] = fminf(thr[j], s_td[w][K - 1]);
                }
            }
        }
    }
    wave_mem_sync();
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        if (s0 + j >= S) break;
        // global nearest (slot 0 when nothing lies within the radius)
        float nd = gd[j];
        int ni = gi[j];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float od_ = __shfl_xor(nd, off, 64);
            const int oi_ = __shfl_xor(ni, off, 64);
            if (od_ < nd || (od_ == nd && oi_ < ni)) nd = od_, ni = oi_;
        }
        if (ni == 0x7fffffff) ni = 0;  // empty frame
        const int k = min(K, cnt[j]);
        select_smallest(s_d[w][j], s_i[w][j], cnt[j], k, s_td[w], s_ti[w]);
        const int first = (k > 0) ? s_ti[w][0] : ni;
        int32_t *out = idx_all + ((size_t)b * S + (s0 + j)) * K;
        if (lane < K) out[lane] = (lane < k) ? s_ti[w][lane] : first;
        wave_mem_sync();
    }
}

}  // namespace

extern "C" int dpm_knn_hybrid(const float *points, const int32_t *lengths, const float *centers, int B, int N,
                              int S, int K, double radius, int32_t *idx, dpm_stream_t stream) {
    DPM_CHECK_ARG(points && lengths && centers && idx);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && radius > 0.0);
    if (K > KMAX) return DPM_EUNSUPPORTED;
    dim3 grid(dpm_cdiv(S, WPB * CPW), B);
    hipLaunchKernelGGL(knn_hybrid_kernel, grid, dim3(WPB * 64), 0, (hipStream_t)stream, points, lengths,
                       centers, N, S, K, (float)(radius * radius), idx);
    return dpm_launch_status();
}
