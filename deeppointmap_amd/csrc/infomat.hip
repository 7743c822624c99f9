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
    int pad[2];
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
// per-block partial moments [cdiv(N1,256)][10], after the sorted points
__device__ __forceinline__ double *pair_partial(const PairArgs &a, int p) {
    return (double *)(pair_sorted(a, p) + a.N2);
}

__global__ __launch_bounds__(1024) void grid_setup_kernel(PairArgs A, float radius) {
    const int pair = blockIdx.x;
    const float *p2 = pair_p2(A, pair);
    const int N2 = A.N2;
    GridHdr *hdr = pair_hdr(A, pair);
    __shared__ float red[4][16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float lox = __builtin_inff(), loy = lox, hix = -lox, hiy = -lox;
    for (int i = t; i < N2; i += 1024) {
        const float x = p2[i], y = p2[(size_t)N2 + i];
        lox = fminf(lox, x), hix = fmaxf(hix, x), loy = fminf(loy, y), hiy = fmaxf(hiy, y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lox = fminf(lox, __shfl_xor(lox, off, 64)), loy = fminf(loy, __shfl_xor(loy, off, 64));
        hix = fmaxf(hix, __shfl_xor(hix, off, 64)), hiy = fmaxf(hiy, __shfl_xor(hiy, off, 64));
    }
    if (lane == 0) red[0][w] = lox, red[1][w] = loy, red[2][w] = hix, red[3][w] = hiy;
    __syncthreads();
    if (t == 0) {
        for (int k = 1; k < 16; ++k) {
            lox = fminf(lox, red[0][k]), loy = fminf(loy, red[1][k]);
            hix = fmaxf(hix, red[2][k]), hiy = fmaxf(hiy, red[3][k]);
        }
        const float ext = fmaxf(fmaxf(hix - lox, hiy - loy), 1e-6f);
        const float cs = fmaxf(radius, ext / (float)(GMAX - 1));  // cell edge >= radius: 3x3 search is exact
        hdr->lox = lox, hdr->loy = loy, hdr->inv_cs = 1.0f / cs;
        hdr->gx = min(GMAX, (int)((hix - lox) / cs) + 1);
        hdr->gy = min(GMAX, (int)((hiy - loy) / cs) + 1);
        hdr->ncell = hdr->gx * hdr->gy;
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

// Counting sort of pcd2 into the grid, entirely in LDS: the cells are split into SLABS contiguous index ranges
// and one 1024-thread workgroup owns one slab of one pair.  It streams the pair's points twice (count, then
// place), keeps its cells' counters / cursors in LDS and derives its global base offset from the number of
// points that fall into earlier slabs, so slabs need no communication and there is no global atomic at all
// (the three-kernel count / scan / scatter version spent 0.35 ms per launch in device-scope atomics).
constexpr int SLABS = 8;
constexpr int SLAB_CELLS = GMAX * GMAX / SLABS;  // 32768 counters = 128 KB of LDS

__global__ __launch_bounds__(1024) void grid_build_kernel(PairArgs A) {
    int pair, slab;
    pair_block(slab, pair);
    const float *p2 = pair_p2(A, pair);
    const int N2 = A.N2;
    const GridHdr *hdr = pair_hdr(A, pair);
    int *start = pair_count(A, pair);
    float4 *sorted = pair_sorted(A, pair);
    __shared__ int cnt[SLAB_CELLS];
    __shared__ int wsum[16];
    __shared__ int s_before;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int ncell = hdr->ncell, gx = hdr->gx, gy = hdr->gy;
    const float lox = hdr->lox, loy = hdr->loy, inv_cs = hdr->inv_cs;
    // as few slabs as the LDS allows (one for any grid up to 32768 cells, i.e. 180 m x 180 m at the 1 m radius):
    // every slab streams all points, so extra slabs only cost; the surplus workgroups leave at once
    const int used = (ncell + SLAB_CELLS - 1) / SLAB_CELLS;
    if (slab >= used) return;
    const int per = (ncell + used - 1) / used;
    const int c0 = min(slab * per, ncell), c1 = min(c0 + per, ncell), nc = c1 - c0;
    for (int c = t; c < nc; c += 1024) cnt[c] = 0;
    if (t == 0) s_before = 0;
    __syncthreads();
    int before = 0;
    constexpr int UB = 8;  // points per thread per batch: all loads of a batch are in flight before any is used
    for (int i0 = t; i0 < N2; i0 += 1024 * UB) {
        float xs[UB], ys[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = min(i0 + u * 1024, N2 - 1);
            xs[u] = p2[i], ys[u] = p2[(size_t)N2 + i];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            if (i0 + u * 1024 >= N2) break;
            const int cell = cell_coord(ys[u], loy, inv_cs, gy) * gx + cell_coord(xs[u], lox, inv_cs, gx);
            if (cell < c0) ++before;
            else if (cell < c1) atomicAdd(&cnt[cell - c0], 1);
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) before += __shfl_xor(before, off, 64);
    if (lane == 0 && before) atomicAdd(&s_before, before);
    __syncthreads();
    // exclusive scan of the slab's counters (base = points of earlier slabs) -> cell start offsets
    const int chunk = (nc + 1023) / 1024;
    const int a = min(t * chunk, nc), b = min(a + chunk, nc);
    int sum = 0;
    for (int c = a; c < b; ++c) sum += cnt[c];
    int inc = sum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int run = s_before + inc - sum;
    for (int k = 0; k < w; ++k) run += wsum[k];
    for (int c = a; c < b; ++c) {
        const int v = cnt[c];
        cnt[c] = run, start[c0 + c] = run;
        run += v;
    }
    if (slab == used - 1 && t == 0) start[ncell] = N2;
    __syncthreads();
    for (int i0 = t; i0 < N2; i0 += 1024 * UB) {
        float xs[UB], ys[UB], zs[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = min(i0 + u * 1024, N2 - 1);
            xs[u] = p2[i], ys[u] = p2[(size_t)N2 + i], zs[u] = p2[2 * (size_t)N2 + i];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + u * 1024;
            if (i >= N2) break;
            const int cell = cell_coord(ys[u], loy, inv_cs, gy) * gx + cell_coord(xs[u], lox, inv_cs, gx);
            if (cell >= c0 && cell < c1) {
                const int pos = atomicAdd(&cnt[cell - c0], 1);
                sorted[pos] = make_float4(xs[u], ys[u], zs[u], __int_as_float(i));
            }
        }
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
// the quad (a lane-per-query scan issues 64 unrelated 16-byte requests per load instruction and is bound by the
// texture-address unit, not by bytes).  A block still covers 256 queries, each quad taking four of them in turn.
__global__ __launch_bounds__(256) void nn1_moments_kernel(PairArgs A, float r2) {
    int pair, blk;
    pair_block(blk, pair);
    const float *p1 = pair_p1(A, pair);
    const int N1 = A.N1;
    const float *Rt = A.Rt + (size_t)pair * A.rt_stride;  // 12 floats: R row-major, T
    const GridHdr *hdr = pair_hdr(A, pair);
    const int *start = pair_count(A, pair);
    const float4 *sorted = pair_sorted(A, pair);
    __shared__ double sred[4][10];
    const int quad = threadIdx.x >> 2, ql = threadIdx.x & 3;
    const int gx = hdr->gx, gy = hdr->gy;
    const float inv_cs = hdr->inv_cs, cs = 1.0f / inv_cs, lox = hdr->lox, loy = hdr->loy;
    const float k2 = cs * cs * (1.f - 1e-5f);
    double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int j = 0; j < 4; ++j) {
        const int i = blk * 256 + j * 64 + quad;
        if (i >= N1) break;  // uniform inside a quad
        const float x = p1[i], y = p1[(size_t)N1 + i], z = p1[2 * (size_t)N1 + i];
        // R @ pcd1 + T in fp32 (sgemm k-order fma chain, then the broadcast add)
        const float qx = fmaf(Rt[2], z, fmaf(Rt[1], y, Rt[0] * x)) + Rt[9];
        const float qy = fmaf(Rt[5], z, fmaf(Rt[4], y, Rt[3] * x)) + Rt[10];
        const float qz = fmaf(Rt[8], z, fmaf(Rt[7], y, Rt[6] * x)) + Rt[11];
        const float fx = (qx - lox) * inv_cs, fy = (qy - loy) * inv_cs;  // position in cell units
        const float flx = floorf(fx), fly = floorf(fy);
        const int cx = (int)fmaxf(fminf(flx, 1e6f), -1e6f), cy = (int)fmaxf(fminf(fly, 1e6f), -1e6f);
        // start offsets of the 3x3 neighbourhood, loaded up front (12 independent loads).  Cells of one grid row
        // are contiguous in `sorted`, so rs[r][k] .. rs[r][k+1] is cell (cx-1+k, cy-1+r); columns outside the
        // grid collapse to empty ranges through the clamp, rows outside the grid are all-zero.
        int rs[3][4];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = cy - 1 + r;
            const bool in = yy >= 0 && yy < gy;
#pragma unroll
            for (int k = 0; k < 4; ++k) rs[r][k] = in ? start[yy * gx + min(max(cx - 1 + k, 0), gx)] : 0;
        }
        // Distances (cell units) from the query to the neighbouring columns / rows, shrunk by a margin that
        // covers the fp32 rounding of the cell assignment: (d - margin)^2 * cs^2 * (1 - 1e-5) is a strict lower
        // bound of the computed distance to any point stored there, so skipping a cell whose bound exceeds the
        // best distance so far (or radius^2 -- farther matches are discarded anyway) never changes the result.
        const float mg = 1e-3f;
        const float dl = fmaxf(fx - flx - mg, 0.f), dr = fmaxf(flx + 1.f - fx - mg, 0.f);
        const float dd = fmaxf(fy - fly - mg, 0.f), du = fmaxf(fly + 1.f - fy - mg, 0.f);
        const float l2 = dl * dl * k2, rr2 = dr * dr * k2, d2 = dd * dd * k2, u2 = du * du * k2;
        float best = __builtin_inff();
        int bi = 0x7fffffff;
        float bx = 0, by = 0, bz = 0;
        // five segments visited as ONE flat loop (own cell, left, right, row below, row above): a quad moves on
        // as soon as its own segment is exhausted, so a wave costs max-over-quads of the candidates actually
        // visited instead of the sum over nine cells of the per-cell maxima.  p, e, seg are quad-uniform.
        int seg = 0, p = 0, e = 0;
        for (;;) {
            while (p >= e && seg < 5) {
                const float bound = fminf(quad_min(best), r2);
                if (seg == 0) {
                    p = rs[1][1], e = rs[1][2];
                } else if (seg == 1) {
                    p = rs[1][0], e = l2 > bound ? p : rs[1][1];
                } else if (seg == 2) {
                    p = rs[1][2], e = rr2 > bound ? p : rs[1][3];
                } else if (seg == 3) {
                    p = (l2 + d2 > bound) ? rs[0][1] : rs[0][0];
                    e = d2 > bound ? p : ((rr2 + d2 > bound) ? rs[0][2] : rs[0][3]);
                } else {
                    p = (l2 + u2 > bound) ? rs[2][1] : rs[2][0];
                    e = u2 > bound ? p : ((rr2 + u2 > bound) ? rs[2][2] : rs[2][3]);
                }
                ++seg;
            }
            if (p >= e) break;
            // branch-free body: an unconditional load from a clamped slot (p < e here, so e - 1 is a valid one) and
            // selects; the lanes of a quad past the end of the segment re-read its last point and are masked out
            // NU candidates per lane and trip (4 NU per quad): their loads are in flight together
            constexpr int NU = 2;  // 3 and 4 measured slower: most segments hold fewer than eight candidates
            const int pp = p + ql;
            p += 4 * NU;
            float4 tc[NU];
#pragma unroll
            for (int u = 0; u < NU; ++u) tc[u] = sorted[min(pp + 4 * u, e - 1)];
#pragma unroll
            for (int u = 0; u < NU; ++u) {
                const float dx = qx - tc[u].x, dy = qy - tc[u].y, dz = qz - tc[u].z;
                const float d = (dx * dx + dy * dy) + dz * dz;
                const int oi = __float_as_int(tc[u].w);
                const bool take = (pp + 4 * u < e) & ((d < best) | ((d == best) & (oi < bi)));
                best = take ? d : best, bi = take ? oi : bi;
                bx = take ? tc[u].x : bx, by = take ? tc[u].y : by, bz = take ? tc[u].z : bz;
            }
        }
        // quad-wide arg-best: (smallest distance, then smallest original index), with the winner's coordinates
#pragma unroll
        for (int step = 0; step < 2; ++step) {
            const float od = step ? quad_xor2(best) : quad_xor1(best);
            const int oi = __float_as_int(step ? quad_xor2(__int_as_float(bi)) : quad_xor1(__int_as_float(bi)));
            const float ox = step ? quad_xor2(bx) : quad_xor1(bx), oy = step ? quad_xor2(by) : quad_xor1(by);
            const float oz = step ? quad_xor2(bz) : quad_xor1(bz);
            if (od < best || (od == best && oi < bi)) best = od, bi = oi, bx = ox, by = oy, bz = oz;
        }
        if (ql == 0 && best <= r2) {
            const double X = bx, Y = by, Z = bz;
            m[0] += 1.0, m[1] += X, m[2] += Y, m[3] += Z, m[4] += X * X, m[5] += Y * Y, m[6] += Z * Z;
            m[7] += X * Y, m[8] += X * Z, m[9] += Y * Z;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        double v = m[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) sred[w][k] = v;
    }
    __syncthreads();
    // one partial per block, summed in a fixed order by the finalize kernel: no same-address fp64 atomics
    // (256 blocks of a pair hammering one cache line was the kernel's real bottleneck) and a reproducible result
    if (threadIdx.x < 10)
        pair_partial(A, pair)[(size_t)blk * 10 + threadIdx.x] =
            (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
}

__global__ __launch_bounds__(64) void infomat_finalize_kernel(PairArgs A) {
    const double *part = pair_partial(A, blockIdx.x);
    const int nblk = (A.N1 + 255) / 256, lane = threadIdx.x;
    __shared__ double s[10];
    for (int k = 0; k < 10; ++k) {
        double v = 0.0;
        for (int b = lane; b < nblk; b += 64) v += part[(size_t)b * 10 + k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        if (lane == 0) s[k] = v;
    }
    __syncthreads();
    if (lane != 0) return;
    float *out = A.out + (size_t)blockIdx.x * A.out_stride;
    const double n = s[0], x = s[1], y = s[2], z = s[3], xx = s[4], yy = s[5], zz = s[6], xy = s[7], xz = s[8],
                 yz = s[9];
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
               10 * sizeof(double) * (size_t)dpm_cdiv(N1, 256);
    return (b + 255) & ~(size_t)255;
}

extern "C" size_t dpm_infomat_workspace_bytes(int n_pairs, int N1, int N2) {
    return 256 + (size_t)n_pairs * ws_slice_bytes(N1, N2);
}

static int launch_grid(PairArgs A, int n_pairs, double radius, hipStream_t st) {
    hipLaunchKernelGGL(grid_setup_kernel, dim3(n_pairs), dim3(1024), 0, st, A, (float)radius);
    hipLaunchKernelGGL(grid_build_kernel, dim3(SLABS, n_pairs), dim3(1024), 0, st, A);
    return dpm_launch_status();
}

static int launch_search(PairArgs A, int n_pairs, double radius, hipStream_t st) {
    hipLaunchKernelGGL(nn1_moments_kernel, dim3(dpm_cdiv(A.N1, 256), n_pairs), dim3(256), 0, st, A,
                       (float)(radius * radius));
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
