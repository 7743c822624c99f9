// Information matrix of a registered scan pair.  Replaces calculate_information_matrix_from_pcd
// (reference system/modules/utils.py:60-113): p1 = R*pcd1 + T; nearest neighbour of every p1 in
// pcd2 (pytorch3d.knn_points K=1 at utils.py:80); keep d^2 <= radius^2; G^T G summed over the
// matched TARGET points t=(x,y,z) of the Jacobian rows [0,z,-y,1,0,0], [-z,0,x,0,1,0],
// [y,-x,0,0,0,1] (utils.py:86-103).
//
// The search is exact but not brute force: only neighbours within `radius` can be kept, so pcd2
// is counting-sorted into a 2-D xy grid with cell edge >= radius and each query looks at the 3x3
// cells around it (65 536^2 = 4.3e9 pair evaluations become ~6e6).  Summing the three outer
// products gives a matrix that depends on ten moments (n, sum x, y, z, xx, yy, zz, xy, xz, yz);
// they are accumulated in fp64 and rounded once to the fp32 6x6 the reference returns.
#include "dpm_common.h"
#include <type_traits>

namespace {

constexpr int GMAX = 512;  // grid cells per axis (upper bound)

// Batched operation: pair p reads its clouds from pcd1 + f1[p]*stride / pcd2 + f2[p]*stride (f1/f2 NULL: the
// pair index itself), its pose from Rt + p*rt_stride, and owns workspace slice p.
struct PairArgs {
    const float *pcd1, *pcd2;
    const int32_t *f1, *f2;
    long long stride1, stride2;  // floats between consecutive frames
    const float *Rt;
    int rt_stride;
    float *out;
    int out_stride;
    char *ws;
    size_t ws_stride;
    int N1, N2;
};

struct GridHdr {       // lives at the start of each workspace slice
    float lox, loy, inv_cs;
    int gx, gy, ncell;
    int H;             // the (2H+1)^2 cells around a query's cell contain every point within the radius (1 or 2)
    int qexp;          // coordinates enter the moments as integers rint(v * 2^qexp) (see nn1_match_kernel)
};

__device__ __forceinline__ const float *pair_p1(const PairArgs &a, int p) {
    return a.pcd1 + (size_t)(a.f1 ? a.f1[p] : p) * a.stride1;
}
__device__ __forceinline__ const float *pair_p2(const PairArgs &a, int p) {
    return a.pcd2 + (size_t)(a.f2 ? a.f2[p] : p) * a.stride2;
}
__device__ __forceinline__ GridHdr *pair_hdr(const PairArgs &a, int p) { return (GridHdr *)(a.ws + (size_t)p * a.ws_stride); }
__device__ __forceinline__ int *pair_count(const PairArgs &a, int p) { return (int *)(a.ws + (size_t)p * a.ws_stride + 256); }
__device__ __forceinline__ float4 *pair_sorted(const PairArgs &a, int p) {
    return (float4 *)(a.ws + (size_t)p * a.ws_stride + 256 + sizeof(int) * (size_t)(GMAX * GMAX + 1) + 12);
}
// per-block partial moments [cdiv(N1,256)][10] (64-bit integers), after the sorted points
__device__ __forceinline__ long long *pair_partial(const PairArgs &a, int p) {
    return (long long *)(pair_sorted(a, p) + a.N2);
}

// Bounds of the target scan in chunks of GB_CHUNK points (a 1024-thread workgroup per scan walked it alone: 41 us, and 71 us
// between the other stages' workgroups, where sixteen waves wait for half a compute unit to fall free), then the header.
constexpr int GB_CHUNK = 4096;  // points per workgroup of the chunk passes (16 per thread)

__device__ __forceinline__ float *pair_bounds(const PairArgs &a, int p);  // [chunks][8], defined with the build's scratch below

__global__ __launch_bounds__(256) void grid_bounds_kernel(PairArgs A) {
    const int chunk = blockIdx.x, pair = blockIdx.y, chunks = gridDim.x;
    const float *p2 = pair_p2(A, pair);
    const int N2 = A.N2;
    __shared__ float red[5][4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float lox = __builtin_inff(), loy = lox, hix = -lox, hiy = -lox, amax = 0.f;
    constexpr int PPT = GB_CHUNK / 256;
    const int i0 = chunk * GB_CHUNK + t;
    float xs[PPT], ys[PPT], zs[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {  // clamped duplicates change neither a minimum nor a maximum
        const int i = min(i0 + u * 256, N2 - 1);
        xs[u] = p2[i], ys[u] = p2[(size_t)N2 + i], zs[u] = p2[2 * (size_t)N2 + i];
    }
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        lox = fminf(lox, xs[u]), hix = fmaxf(hix, xs[u]), loy = fminf(loy, ys[u]), hiy = fmaxf(hiy, ys[u]);
        amax = fmaxf(amax, fmaxf(fmaxf(fabsf(xs[u]), fabsf(ys[u])), fabsf(zs[u])));
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lox = fminf(lox, __shfl_xor(lox, off, 64)), loy = fminf(loy, __shfl_xor(loy, off, 64));
        hix = fmaxf(hix, __shfl_xor(hix, off, 64)), hiy = fmaxf(hiy, __shfl_xor(hiy, off, 64));
        amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    }
    if (lane == 0) red[0][w] = lox, red[1][w] = loy, red[2][w] = hix, red[3][w] = hiy, red[4][w] = amax;
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < 4; ++k) {
            lox = fminf(lox, red[0][k]), loy = fminf(loy, red[1][k]);
            hix = fmaxf(hix, red[2][k]), hiy = fmaxf(hiy, red[3][k]);
            amax = fmaxf(amax, red[4][k]);
        }
        float *o = pair_bounds(A, pair) + (size_t)chunk * 8;
        o[0] = lox, o[1] = loy, o[2] = hix, o[3] = hiy, o[4] = amax;
        (void)chunks;
    }
}

__global__ __launch_bounds__(64) void grid_setup_kernel(PairArgs A, float radius, int fine, int chunks) {
    const int pair = blockIdx.x;
    GridHdr *hdr = pair_hdr(A, pair);
    if (threadIdx.x == 0) {
        float lox = __builtin_inff(), loy = lox, hix = -lox, hiy = -lox, amax = 0.f;
        const float *o = pair_bounds(A, pair);
        for (int k = 0; k < chunks; ++k, o += 8) {
            lox = fminf(lox, o[0]), loy = fminf(loy, o[1]);
            hix = fmaxf(hix, o[2]), hiy = fmaxf(hiy, o[3]);
            amax = fmaxf(amax, o[4]);
        }
        const float ext = fmaxf(fmaxf(hix - lox, hiy - loy), 1e-6f);
        // cell edge = half the radius (5x5 cells cover it: a scan's nearest neighbour is a fraction of the radius away, and
        // the 5x5 block of half-size cells holds 25/36 of the 3x3 block of full-size ones), unless that needs more cells per
        // axis than the grid has; then, and with fine == 0, edge >= radius and 3x3 cells
        float cs = fmaxf(radius * 0.5005f, ext / (float)(GMAX - 1));
        int H = 2;
        if (!fine || cs >= radius) cs = fmaxf(radius, ext / (float)(GMAX - 1)), H = 1;
        hdr->lox = lox, hdr->loy = loy, hdr->inv_cs = 1.0f / cs, hdr->H = H;
        hdr->gx = min(GMAX, (int)((hix - lox) / cs) + 1);
        hdr->gy = min(GMAX, (int)((hiy - loy) / cs) + 1);
        hdr->ncell = hdr->gx * hdr->gy;
        // fixed-point scale of the moments: |v * 2^qexp| < 2^bits with bits such that N1 squares still fit 63 bits
        int lg = 0;
        while (lg < 31 && (1ll << lg) < (long long)A.N1) ++lg;
        const int bits = min(21, (62 - lg) / 2);
        int e = 0;
        (void)frexpf(fmaxf(amax, 1e-30f), &e);  // amax < 2^e
        hdr->qexp = bits - e;
    }
}

// Grid (nblk, n_pairs) -> (block within the pair, pair) such that ALL blocks of one pair run on the SAME XCD.
// Workgroups are dealt round-robin to the 8 XCDs by linear id and every XCD has a private 4 MB L2; with the
// plain mapping each L2 sees the grids of all pairs at once (64 x 1 MB: every candidate load misses to the
// fabric -- 4.8 GB per launch measured), with this one it holds the one or two pairs it is working on.
__device__ __forceinline__ void pair_block(int &blk, int &pair) {
    const int nblk = gridDim.x, npair = gridDim.y;
    if (npair % 8 == 0) {
        const unsigned L = blockIdx.y * nblk + blockIdx.x;
        const unsigned xcd = L & 7, slot = L >> 3;
        pair = (int)((slot / nblk) * 8 + xcd), blk = (int)(slot % nblk);
    } else {
        pair = blockIdx.y, blk = blockIdx.x;
    }
}

__device__ __forceinline__ int cell_coord(float v, float lo, float inv_cs, int g) {
    return min(max((int)floorf((v - lo) * inv_cs), 0), g - 1);
}

// Counting sort of pcd2 into the grid as FOUR short chip-wide kernels with a few KB of LDS each (round 3).  The one-kernel
// form it replaces kept a scan's 57 600 cell counters in the LDS of ONE 1024-thread workgroup (131 KB): 64 compute units
// were closed to every other workgroup of the pipeline for the 0.23 ms (0.47 ms under load) it took, and the pipelined
// step was 0.2 ms shorter without it (ablation).  Now:
//   rows:    a workgroup takes a chunk of GB_CHUNK points, counts them per grid ROW (<= 512 counters in LDS) and leaves
//            the histogram in the workspace;
//   offsets: one workgroup per scan turns the histograms into start offsets per (chunk, row) and per row;
//   scatter: the chunks again -- a point goes to its row's range in a temporary array (points of one chunk and row are
//            consecutive there);
//   cells:   one WAVE per grid row: counting sort of the row's points by cell (<= 512 counters in LDS per wave) into the
//            final array, plus the row's cell start offsets.
// The order of the points inside a cell depends on scheduling (LDS atomics); nothing downstream depends on it: the search
// takes the minimum of (distance, original index) keys and sums integers.
__device__ __forceinline__ float4 *pair_tmp(const PairArgs &a, int p) {
    return (float4 *)(pair_partial(a, p) + 10 * (size_t)((a.N1 + 255) / 256));
}
__device__ __forceinline__ int *pair_rowhist(const PairArgs &a, int p) { return (int *)(pair_tmp(a, p) + a.N2); }  // [chunks][GMAX]
__device__ __forceinline__ int *pair_rowstart(const PairArgs &a, int p) {                                           // [GMAX + 1]
    return pair_rowhist(a, p) + (size_t)((a.N2 + GB_CHUNK - 1) / GB_CHUNK) * GMAX;
}
__device__ __forceinline__ float *pair_bounds(const PairArgs &a, int p) {                                        // [chunks][8]
    return (float *)(pair_rowstart(a, p) + GMAX + 4);
}

// PLACE = false: row histogram of the chunk; PLACE = true: the chunk's points into their rows' ranges of `tmp`
template <bool PLACE>
__global__ __launch_bounds__(256) void grid_rows_kernel(PairArgs A) {
    int pair, chunk;
    pair_block(chunk, pair);
    const float *p2 = pair_p2(A, pair);
    const int N2 = A.N2;
    const GridHdr *hdr = pair_hdr(A, pair);
    const int gy = hdr->gy;
    const float loy = hdr->loy, inv_cs = hdr->inv_cs;
    __shared__ int cnt[GMAX];
    int *hist = pair_rowhist(A, pair) + (size_t)chunk * GMAX;
    const int t = threadIdx.x;
    for (int r = t; r < gy; r += 256) cnt[r] = PLACE ? hist[r] : 0;
    __syncthreads();
    constexpr int PPT = GB_CHUNK / 256;
    const int i0 = chunk * GB_CHUNK + t;
    float xs[PPT], ys[PPT], zs[PPT];
#pragma unroll
    for (int u = 0; u < PPT; ++u) {  // all loads of the chunk in flight before the first is used
        const int i = min(i0 + u * 256, N2 - 1);
        ys[u] = p2[(size_t)N2 + i];
        if (PLACE) xs[u] = p2[i], zs[u] = p2[2 * (size_t)N2 + i];
    }
    float4 *tmp = pair_tmp(A, pair);
#pragma unroll
    for (int u = 0; u < PPT; ++u) {
        const int i = i0 + u * 256;
        if (i >= N2) break;
        const int row = cell_coord(ys[u], loy, inv_cs, gy);
        const int pos = atomicAdd(&cnt[row], 1);
        if (PLACE) tmp[pos] = make_float4(xs[u], ys[u], zs[u], __int_as_float(i));
    }
    if (PLACE) return;
    __syncthreads();
    for (int r = t; r < gy; r += 256) hist[r] = cnt[r];
}

// histograms [chunk][row] -> start offset of every (chunk, row) run in `tmp` (in place) and of every row (rowstart)
__global__ __launch_bounds__(256) void grid_offsets_kernel(PairArgs A) {
    const int pair = blockIdx.x;
    const GridHdr *hdr = pair_hdr(A, pair);
    const int gy = hdr->gy, chunks = (A.N2 + GB_CHUNK - 1) / GB_CHUNK;
    int *hist = pair_rowhist(A, pair), *rowstart = pair_rowstart(A, pair);
    __shared__ int wsum[4];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    // thread t owns rows 2t and 2t + 1 (gy <= GMAX = 512)
    int tot[2] = {0, 0};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = 2 * t + k;
        if (r < gy)
            for (int c = 0; c < chunks; ++c) tot[k] += hist[(size_t)c * GMAX + r];
    }
    const int sum = tot[0] + tot[1];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int run = inc - sum;
    for (int k = 0; k < w; ++k) run += wsum[k];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int r = 2 * t + k;
        if (r < gy) {
            rowstart[r] = run;
            int at = run;
            for (int c = 0; c < chunks; ++c) {
                const int n = hist[(size_t)c * GMAX + r];
                hist[(size_t)c * GMAX + r] = at, at += n;
            }
            run = at;
        }
    }
    if (t == 0) rowstart[gy] = A.N2;
}

// one wave per grid row: the row's points (a contiguous range of `tmp`) counting-sorted by cell into `sorted`
__global__ __launch_bounds__(256) void grid_cells_kernel(PairArgs A) {
    int pair, blk;
    pair_block(blk, pair);
    const GridHdr *hdr = pair_hdr(A, pair);
    const int gx = hdr->gx, gy = hdr->gy;
    const float lox = hdr->lox, inv_cs = hdr->inv_cs;
    __shared__ int cnt[4][GMAX];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6, row = blk * 4 + w;
    if (row >= gy) return;  // whole waves; no workgroup barrier below
    const int *rowstart = pair_rowstart(A, pair);
    int *start = pair_count(A, pair);
    const float4 *tmp = pair_tmp(A, pair);
    float4 *sorted = pair_sorted(A, pair);
    const int lo = rowstart[row], hi = rowstart[row + 1];
    int *c = cnt[w];
    for (int k = lane; k < gx; k += 64) c[k] = 0;
    // the row's first RB * 64 points are fetched together and stay in registers for both passes (a typical row holds a few
    // hundred points: one round trip instead of a dozen dependent ones); longer rows loop over the rest
    constexpr int RB = 8;
    float4 v[RB];
#pragma unroll
    for (int u = 0; u < RB; ++u) v[u] = tmp[min(lo + lane + u * 64, max(hi, 1) - 1)];  // clamped: an empty row reads a neighbour's point and drops it
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < RB; ++u)
        if (lo + lane + u * 64 < hi) atomicAdd(&c[cell_coord(v[u].x, lox, inv_cs, gx)], 1);
    for (int p = lo + lane + RB * 64; p < hi; p += 64) atomicAdd(&c[cell_coord(tmp[p].x, lox, inv_cs, gx)], 1);
    __builtin_amdgcn_wave_barrier();
    // exclusive scan over the row's gx counters: lane l owns cells [l * per, (l + 1) * per)
    const int per = (gx + 63) / 64, a = min(lane * per, gx), b = min(a + per, gx);
    int sum = 0;
    for (int k = a; k < b; ++k) sum += c[k];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    int run = lo + inc - sum;
    for (int k = a; k < b; ++k) {
        const int n = c[k];
        c[k] = run, run += n;
    }
    __builtin_amdgcn_wave_barrier();
    for (int k = lane; k < gx; k += 64) start[(size_t)row * gx + k] = c[k];  // cursors still at the cell starts
    if (row == gy - 1 && lane == 0) start[(size_t)gy * gx] = A.N2;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int u = 0; u < RB; ++u)
        if (lo + lane + u * 64 < hi) sorted[atomicAdd(&c[cell_coord(v[u].x, lox, inv_cs, gx)], 1)] = v[u];
    for (int p = lo + lane + RB * 64; p < hi; p += 64) {
        const float4 q = tmp[p];
        sorted[atomicAdd(&c[cell_coord(q.x, lox, inv_cs, gx)], 1)] = q;
    }
}

// quad_perm exchanges inside groups of 4 lanes (DPP, no LDS traffic)
__device__ __forceinline__ float quad_xor1(float v) { return __int_as_float(dpp_i<0xB1, 0xF>(__float_as_int(v))); }
__device__ __forceinline__ float quad_xor2(float v) { return __int_as_float(dpp_i<0x4E, 0xF>(__float_as_int(v))); }
__device__ __forceinline__ float quad_min(float v) {
    v = fminf(v, quad_xor1(v));
    return fminf(v, quad_xor2(v));
}

// FOUR lanes per query: they scan consecutive candidates of the same cell range, so one 64-byte request serves
// the quad (a lane-per-query scan issues 64 unrelated 16-byte requests per load instruction).  A block covers 256
// queries, each quad taking four of them in turn.
// Round 2's kernel was bound by VALU issue (PMC: vector ALU busy 98 % of the time); this one is laid out for few
// instructions per candidate and, once those were gone, for the memory system:
//  * the (2H+1) grid rows around the query's cell are (2H+1) contiguous ranges of the sorted array (cells cx-H .. cx+H of
//    one row are neighbours in memory); all range ends are fetched up front, the first eight candidates of the three
//    nearest rows are requested together, and the running best is ONE 64-bit key (distance bits, original index):
//    smaller key = nearer, then smaller index -- one compare and two selects per candidate (193 M -> 105 M vector
//    instructions per 64-pair launch);
//  * the rows two cells away are pruned by an exact lower bound (cell-unit distance to the row, shrunk by a margin for
//    the rounding of the cell assignment);
//  * the queries are walked in the CELL ORDER of the source scan's own grid when another pair of the batch has built one of
//    it (consecutive-frame edges: always): neighbouring quads then read the same rows and the 32 KB L1 serves them; in
//    index order every load went to the L2 (355 -> 250 us);
//  * the moments are summed as 64-bit integers (see the end of the kernel), so the result does not depend on the order the
//    queries are walked in.
__global__ __launch_bounds__(256) void nn1_match_kernel(PairArgs A, float r2, int ordered) {
    valu_bound_priority();
    int pair, blk;
    pair_block(blk, pair);
    const float *p1 = pair_p1(A, pair);
    const float *p2 = pair_p2(A, pair);
    const int N1 = A.N1, N2 = A.N2;
    const float *Rt = A.Rt + (size_t)pair * A.rt_stride;  // 12 floats: R row-major, T
    const GridHdr *hdr = pair_hdr(A, pair);
    const int *start = pair_count(A, pair);
    const float4 *sorted = pair_sorted(A, pair);
    __shared__ int s_qp;
    __shared__ long long sred[4][10];
    const int quad = threadIdx.x >> 2, ql = threadIdx.x & 3;
    // the pair (if any) whose target is this pair's source scan: its sorted array is the source scan in cell order
    if (threadIdx.x == 0) s_qp = -1;
    __syncthreads();
    if (ordered && A.f1 && A.f2 && N1 == N2)
        for (int t = threadIdx.x; t < (int)gridDim.y; t += 256)
            if (A.f2[t] == A.f1[pair]) atomicMax(&s_qp, t);
    __syncthreads();
    const float4 *qsorted = s_qp >= 0 ? pair_sorted(A, s_qp) : nullptr;
    const int gx = hdr->gx, gy = hdr->gy, H = hdr->H;
    const float inv_cs = hdr->inv_cs, cs = 1.0f / inv_cs, lox = hdr->lox, loy = hdr->loy;
    const float k2 = cs * cs * (1.f - 1e-5f);
    constexpr int NR = 5;  // rows of the largest block (H <= 2), visited nearest first: 0, -1, +1, -2, +2
    int win[4] = {-1, -1, -1, -1};  // original index of the match of the quad's j-th query (-1: none within the radius)
    // Everything about a query that does not depend on the candidates -- the transformed point, its cell, the ten range ends of
    // the rows around it -- is prepared ONCE, by lane j of the quad for the quad's j-th query (round 6; until then all four lanes
    // prepared all four queries: ~80 of the kernel's ~160 vector instructions per query and lane, and four times the range-end
    // loads), and handed to the other three lanes with quad-permute DPP moves when the query's turn comes.
    const int i_own = blk * 256 + ql * 64 + quad;
    float o_qx, o_qy, o_qz, o_below, o_above;
    int o_rlo[NR], o_rhi[NR];
    {
        const int ic = min(i_own, N1 - 1);   // lanes past the end prepare a copy of the last query; their turn never comes
        float x, y, z;
        if (qsorted) {
            const float4 q4 = qsorted[ic];
            x = q4.x, y = q4.y, z = q4.z;
        } else {
            x = p1[ic], y = p1[(size_t)N1 + ic], z = p1[2 * (size_t)N1 + ic];
        }
        // R @ pcd1 + T in fp32 (sgemm k-order fma chain, then the broadcast add)
        o_qx = fmaf(Rt[2], z, fmaf(Rt[1], y, Rt[0] * x)) + Rt[9];
        o_qy = fmaf(Rt[5], z, fmaf(Rt[4], y, Rt[3] * x)) + Rt[10];
        o_qz = fmaf(Rt[8], z, fmaf(Rt[7], y, Rt[6] * x)) + Rt[11];
        const float fx = (o_qx - lox) * inv_cs, fy = (o_qy - loy) * inv_cs;  // position in cell units
        const float flx = floorf(fx), fly = floorf(fy);
        const int cx = (int)fmaxf(fminf(flx, 1e6f), -1e6f), cy = (int)fmaxf(fminf(fly, 1e6f), -1e6f);
        // range ends of the rows, all requested before the first is used.  Columns outside the grid collapse through the
        // clamp, rows outside the grid are empty.
        const int xa = min(max(cx - H, 0), gx), xb = min(max(cx + H + 1, 0), gx);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const int k = (r + 1) / 2 * ((r & 1) ? -1 : 1);  // 0, -1, +1, -2, +2
            const int yy = cy + k;
            const bool in = yy >= 0 && yy < gy && (r < 3 || H > 1);
            const int yc = min(max(yy, 0), gy - 1);
            o_rlo[r] = in ? start[yc * gx + xa] : 0, o_rhi[r] = in ? start[yc * gx + xb] : 0;
        }
        o_below = fy - fly, o_above = fly + 1.f - fy;  // cell units to the lower / upper edge of the query's row
    }
    auto one_query = [&](auto Jc) {
        constexpr int j = decltype(Jc)::value;
        constexpr int BC = j * 0x55;  // quad_perm:[j,j,j,j]
        const int i = blk * 256 + j * 64 + quad;
        if (i >= N1) return;  // uniform inside a quad
        auto from_j = [&](int v) { return __builtin_amdgcn_update_dpp(v, v, BC, 0xF, 0xF, false); };
        auto from_jf = [&](float v) { return __int_as_float(from_j(__float_as_int(v))); };
        const float qx = from_jf(o_qx), qy = from_jf(o_qy), qz = from_jf(o_qz);
        int rlo[NR], rhi[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) rlo[r] = from_j(o_rlo[r]), rhi[r] = from_j(o_rhi[r]);
        const float mg = 1e-3f;
        unsigned long long best = ~0ull;
        auto offer = [&](const float4 t, bool ok) {
            const float dx = qx - t.x, dy = qy - t.y, dz = qz - t.z;
            const float d = (dx * dx + dy * dy) + dz * dz;
            const unsigned long long key = ((unsigned long long)__float_as_uint(d) << 32) | (unsigned)__float_as_int(t.w);
            best = (ok && key < best) ? key : best;
        };
        // candidates [p, hi) of one row from position `from` on, two per lane and trip
        auto rest = [&](int from, int hi) {
            for (int p = from; p < hi; p += 8) {
                const float4 t0 = sorted[p], t1 = sorted[min(p + 4, hi - 1)];  // unconditional, the second from a clamped slot
                offer(t0, true), offer(t1, p + 4 < hi);
            }
        };
        // The query's own row and its two neighbours: the first eight candidates of each (two per lane) are requested
        // together -- six loads in flight instead of a chain of dependent round trips (the kernel waits for memory, not
        // for the ALU) -- rows with more than eight finish in a loop.
        {
            float4 t[3][2];
            bool ok[3][2];
#pragma unroll
            for (int r = 0; r < 3; ++r) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int p = rlo[r] + ql + 4 * u;
                    ok[r][u] = p < rhi[r];
                    t[r][u] = sorted[ok[r][u] ? p : 0];
                }
            }
#pragma unroll
            for (int r = 0; r < 3; ++r) offer(t[r][0], ok[r][0]), offer(t[r][1], ok[r][1]);
#pragma unroll
            for (int r = 0; r < 3; ++r) rest(rlo[r] + ql + 8, rhi[r]);
        }
        // the rows two cells away, pruned by an exact lower bound of the distance to anything stored there
        if (H > 1) {
            const float bd = best == ~0ull ? r2 : fminf(__uint_as_float((unsigned)(best >> 32)), r2);
            const float below = from_jf(o_below), above = from_jf(o_above);
            const float g0 = fmaxf(below + 1.f - mg, 0.f), g1 = fmaxf(above + 1.f - mg, 0.f);
            const int h3 = (g0 * g0 * k2 > bd) ? 0 : rhi[3], h4 = (g1 * g1 * k2 > bd) ? 0 : rhi[4];
            const int p3 = rlo[3] + ql, p4 = rlo[4] + ql;
            const float4 a0 = sorted[p3 < h3 ? p3 : 0], a1 = sorted[p3 + 4 < h3 ? p3 + 4 : 0];
            const float4 b0 = sorted[p4 < h4 ? p4 : 0], b1 = sorted[p4 + 4 < h4 ? p4 + 4 : 0];
            offer(a0, p3 < h3), offer(a1, p3 + 4 < h3), offer(b0, p4 < h4), offer(b1, p4 + 4 < h4);
            rest(p3 + 8, h3), rest(p4 + 8, h4);
        }
        // quad-wide best
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            const float blo = __int_as_float((int)(unsigned)best), bhi = __int_as_float((int)(unsigned)(best >> 32));
            const unsigned olo = (unsigned)__float_as_int(step ? quad_xor2(blo) : quad_xor1(blo));
            const unsigned ohi = (unsigned)__float_as_int(step ? quad_xor2(bhi) : quad_xor1(bhi));
            const unsigned long long o = ((unsigned long long)ohi << 32) | olo;
            best = o < best ? o : best;
        }
        win[j] = (best != ~0ull && __uint_as_float((unsigned)(best >> 32)) <= r2) ? (int)(unsigned)best : -1;
    };
    one_query(std::integral_constant<int, 0>{});
    one_query(std::integral_constant<int, 1>{});
    one_query(std::integral_constant<int, 2>{});
    one_query(std::integral_constant<int, 3>{});
    // The matched target points of the lane's (up to) four queries, fetched together.  Their moments are summed as INTEGERS
    // (coordinates rounded to 2^-qexp: 21 significant bits of the scan's largest coordinate, 3e-5 m on a 60 m scan, with
    // errors that average out over tens of thousands of terms; the reference itself sums in fp32): integer addition is
    // associative, so the result depends neither on the order the queries are walked in nor on the scheduling-dependent
    // order of the points inside a grid cell.
    long long m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (ql == 0) {
        const float qs = ldexpf(1.f, hdr->qexp);
        float X[4], Y[4], Z[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int oi = max(win[j], 0);
            X[j] = p2[oi], Y[j] = p2[(size_t)N2 + oi], Z[j] = p2[2 * (size_t)N2 + oi];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (win[j] < 0) continue;
            const long long x = (long long)__float2int_rn(X[j] * qs), y = (long long)__float2int_rn(Y[j] * qs),
                            z = (long long)__float2int_rn(Z[j] * qs);
            m[0] += 1, m[1] += x, m[2] += y, m[3] += z, m[4] += x * x, m[5] += y * y, m[6] += z * z;
            m[7] += x * y, m[8] += x * z, m[9] += y * z;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        long long v = m[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) sred[w][k] = v;
    }
    __syncthreads();
    // one partial per block, summed by the finalize kernel: no same-address atomics
    if (threadIdx.x < 10)
        pair_partial(A, pair)[(size_t)blk * 10 + threadIdx.x] =
            (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void infomat_finalize_kernel(PairArgs A) {
    const long long *part = pair_partial(A, blockIdx.x);
    const int nblk = (A.N1 + 255) / 256, lane = threadIdx.x;
    __shared__ long long s[10];
    for (int k = 0; k < 10; ++k) {
        long long v = 0;
        for (int b = lane; b < nblk; b += 64) v += part[(size_t)b * 10 + k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) s[k] = v;
    }
    __syncthreads();
    if (lane != 0) return;
    float *out = A.out + (size_t)blockIdx.x * A.out_stride;
    const int qexp = pair_hdr(A, blockIdx.x)->qexp;
    const double u = ldexp(1.0, -qexp), u2 = ldexp(1.0, -2 * qexp);
    const double n = (double)s[0], x = (double)s[1] * u, y = (double)s[2] * u, z = (double)s[3] * u, xx = (double)s[4] * u2,
                 yy = (double)s[5] * u2, zz = (double)s[6] * u2, xy = (double)s[7] * u2, xz = (double)s[8] * u2,
                 yz = (double)s[9] * u2;
    const double G[36] = {zz + yy, -xy,     -xz,     0,  -z, y,   //
                          -xy,     zz + xx, -yz,     z,  0,  -x,  //
                          -xz,     -yz,     yy + xx, -y, x,  0,   //
                          0,       z,       -y,      n,  0,  0,   //
                          -z,      0,       x,       0,  n,  0,   //
                          y,       -x,      0,       0,  0,  n};
    for (int i = 0; i < 36; ++i) out[i] = (float)G[i];
}

}  // namespace

static size_t ws_slice_bytes(int N1, int N2) {
    size_t b = 256 + sizeof(int) * (size_t)(GMAX * GMAX + 1) + 12 + sizeof(float4) * (size_t)N2 +
               10 * sizeof(long long) * (size_t)dpm_cdiv(N1, 256) +
               sizeof(float4) * (size_t)N2 + sizeof(int) * ((size_t)dpm_cdiv(N2, GB_CHUNK) * (GMAX + 8) + GMAX + 4);  // row-sorted copy, row histograms, row starts, chunk bounds
    return (b + 255) & ~(size_t)255;
}

extern "C" size_t dpm_infomat_workspace_bytes(int n_pairs, int N1, int N2) {
    return 256 + (size_t)n_pairs * ws_slice_bytes(N1, N2);
}

static int launch_grid(PairArgs A, int n_pairs, double radius, hipStream_t st) {
    const int fine = dpm_knob("DPM_NN1_FINE", 1);  // -DDPM_EXPERIMENT builds only; 0: cells of one radius and 3x3 blocks (the round-2 layout; A/B measurements)
    const int chunks = dpm_cdiv(A.N2, GB_CHUNK);
    hipLaunchKernelGGL(grid_bounds_kernel, dim3(chunks, n_pairs), dim3(256), 0, st, A);
    hipLaunchKernelGGL(grid_setup_kernel, dim3(n_pairs), dim3(64), 0, st, A, (float)radius, fine, chunks);
    if (dpm_knob("DPM_ABLATE_GRID", 0)) return dpm_launch_status();  // -DDPM_EXPERIMENT builds only (with DPM_ABLATE_NN1: nothing reads the grid)
    hipLaunchKernelGGL(grid_rows_kernel<false>, dim3(chunks, n_pairs), dim3(256), 0, st, A);
    hipLaunchKernelGGL(grid_offsets_kernel, dim3(n_pairs), dim3(256), 0, st, A);
    hipLaunchKernelGGL(grid_rows_kernel<true>, dim3(chunks, n_pairs), dim3(256), 0, st, A);
    hipLaunchKernelGGL(grid_cells_kernel, dim3(GMAX / 4, n_pairs), dim3(256), 0, st, A);
    return dpm_launch_status();
}

static int launch_search(PairArgs A, int n_pairs, double radius, hipStream_t st) {
    if (dpm_knob("DPM_ABLATE_NN1", 0)) return DPM_OK;  // -DDPM_EXPERIMENT builds only
    const int ord = dpm_knob("DPM_NN1_ORDERED", 1);  // -DDPM_EXPERIMENT builds only; 0: queries in index order (A/B measurements)
    hipLaunchKernelGGL(nn1_match_kernel, dim3(dpm_cdiv(A.N1, 256), n_pairs), dim3(256), 0, st, A,
                       (float)(radius * radius), ord);
    hipLaunchKernelGGL(infomat_finalize_kernel, dim3(n_pairs), dim3(64), 0, st, A);
    return dpm_launch_status();
}

static int launch_infomat(PairArgs A, int n_pairs, double radius, hipStream_t st) {
    const int rc = launch_grid(A, n_pairs, radius, st);
    return rc ? rc : launch_search(A, n_pairs, radius, st);
}

static PairArgs batched_args(const float *pcd, int N, const int32_t *src_frame, const int32_t *dst_frame,
                             void *workspace) {
    PairArgs A{};
    A.pcd1 = pcd, A.pcd2 = pcd, A.f1 = src_frame, A.f2 = dst_frame, A.stride1 = 3LL * N, A.stride2 = 3LL * N;
    A.ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), A.ws_stride = ws_slice_bytes(N, N);
    A.N1 = N, A.N2 = N;
    return A;
}

extern "C" int dpm_information_matrix(const float *pcd1, int N1, const float *pcd2, int N2, const float *Rt,
                                      double radius, float *out6x6, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(pcd1 && pcd2 && Rt && out6x6 && workspace && N1 >= 1 && N2 >= 1 && radius > 0.0);
    PairArgs A{};
    A.pcd1 = pcd1, A.pcd2 = pcd2, A.f1 = nullptr, A.f2 = nullptr, A.stride1 = 0, A.stride2 = 0;
    A.Rt = Rt, A.rt_stride = 0, A.out = out6x6, A.out_stride = 0;
    A.ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), A.ws_stride = ws_slice_bytes(N1, N2);
    A.N1 = N1, A.N2 = N2;
    return launch_infomat(A, 1, radius, (hipStream_t)stream);
}

extern "C" int dpm_information_matrix_batched(const float *pcd, int N, const int32_t *src_frame,
                                              const int32_t *dst_frame, int n_pairs, const float *Rt, int rt_stride,
                                              double radius, float *out, int out_stride, void *workspace,
                                              dpm_stream_t stream) {
    DPM_CHECK_ARG(pcd && src_frame && dst_frame && Rt && out && workspace && N >= 1 && n_pairs >= 1 && radius > 0.0);
    DPM_CHECK_ARG(rt_stride >= 12 && out_stride >= 36);
    PairArgs A = batched_args(pcd, N, src_frame, dst_frame, workspace);
    A.Rt = Rt, A.rt_stride = rt_stride, A.out = out, A.out_stride = out_stride;
    return launch_infomat(A, n_pairs, radius, (hipStream_t)stream);
}

// The two halves of dpm_information_matrix_batched.  The grid depends on the target scans only -- not on the
// pose -- so a pipeline builds it while the frames are still being encoded and runs only the search after
// the registration.
extern "C" int dpm_infomat_build_grids(const float *pcd, int N, const int32_t *dst_frame, int n_pairs, double radius,
                                       void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(pcd && dst_frame && workspace && N >= 1 && n_pairs >= 1 && radius > 0.0);
    return launch_grid(batched_args(pcd, N, nullptr, dst_frame, workspace), n_pairs, radius, (hipStream_t)stream);
}

extern "C" int dpm_infomat_search_grids(const float *pcd, int N, const int32_t *src_frame, const int32_t *dst_frame,
                                        int n_pairs, const float *Rt, int rt_stride, double radius, float *out,
                                        int out_stride, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(pcd && src_frame && dst_frame && Rt && out && workspace && N >= 1 && n_pairs >= 1 && radius > 0.0);
    DPM_CHECK_ARG(rt_stride >= 12 && out_stride >= 36);
    PairArgs A = batched_args(pcd, N, src_frame, dst_frame, workspace);
    A.Rt = Rt, A.rt_stride = rt_stride, A.out = out, A.out_stride = out_stride;
    return launch_search(A, n_pairs, radius, (hipStream_t)stream);
}
