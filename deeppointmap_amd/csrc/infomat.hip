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
    double sums[10];
};

__device__ __forceinline__ const float *pair_p1(const PairArgs &a, int p) {
    return a.pcd1 + (size_t)(a.f1 ? a.f1[p] : p) * a.stride1;
}
__device__ __forceinline__ const float *pair_p2(const PairArgs &a, int p) {
    return a.pcd2 + (size_t)(a.f2 ? a.f2[p] : p) * a.stride2;
}
__device__ __forceinline__ GridHdr *pair_hdr(const PairArgs &a, int p) { return (GridHdr *)(a.ws + (size_t)p * a.ws_stride); }
__device__ __forceinline__ int *pair_count(const PairArgs &a, int p) { return (int *)(a.ws + (size_t)p * a.ws_stride + 256); }
__device__ __forceinline__ int *pair_cursor(const PairArgs &a, int p) { return pair_count(a, p) + (GMAX * GMAX + 1); }
__device__ __forceinline__ float4 *pair_sorted(const PairArgs &a, int p) {
    return (float4 *)(a.ws + (size_t)p * a.ws_stride + 256 + 2 * sizeof(int) * (size_t)(GMAX * GMAX + 1) + 8);
}

__global__ __launch_bounds__(1024) void grid_setup_kernel(PairArgs A, float radius) {
    const int pair = blockIdx.x;
    const float *p2 = pair_p2(A, pair);
    const int N2 = A.N2;
    GridHdr *hdr = pair_hdr(A, pair);
    int *count = pair_count(A, pair);
    __shared__ float red[4][16];
    __shared__ int s_ncell;
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
        s_ncell = hdr->ncell;
        for (int k = 0; k < 10; ++k) hdr->sums[k] = 0.0;
    }
    __syncthreads();
    for (int c = t; c <= s_ncell; c += 1024) count[c] = 0;
}

__device__ __forceinline__ int cell_coord(float v, float lo, float inv_cs, int g) {
    return min(max((int)floorf((v - lo) * inv_cs), 0), g - 1);
}

__global__ __launch_bounds__(256) void grid_count_kernel(PairArgs A) {
    const int pair = blockIdx.y;
    const float *p2 = pair_p2(A, pair);
    const int N2 = A.N2;
    const GridHdr *hdr = pair_hdr(A, pair);
    int *count = pair_count(A, pair);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N2) return;
    const int cx = cell_coord(p2[i], hdr->lox, hdr->inv_cs, hdr->gx);
    const int cy = cell_coord(p2[(size_t)N2 + i], hdr->loy, hdr->inv_cs, hdr->gy);
    atomicAdd(&count[cy * hdr->gx + cx], 1);
}

// exclusive scan of count[0..ncell) in place -> cell start offsets; cursor = copy for the scatter
__global__ __launch_bounds__(1024) void grid_scan_kernel(PairArgs A) {
    const GridHdr *hdr = pair_hdr(A, blockIdx.x);
    int *count = pair_count(A, blockIdx.x), *cursor = pair_cursor(A, blockIdx.x);
    __shared__ int wsum[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int ncell = hdr->ncell;
    const int per = (ncell + 1023) / 1024;
    const int c0 = t * per, c1 = min(c0 + per, ncell);
    int s = 0;
    for (int c = c0; c < c1; ++c) s += count[c];
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int base = 0;
    for (int k = 0; k < w; ++k) base += wsum[k];
    int run = base + inc - s;
    for (int c = c0; c < c1; ++c) {
        const int v = count[c];
        count[c] = run, cursor[c] = run;
        run += v;
    }
    if (t == 1023) count[ncell] = run;  // == N2
}

__global__ __launch_bounds__(256) void grid_scatter_kernel(PairArgs A) {
    const int pair = blockIdx.y;
    const float *p2 = pair_p2(A, pair);
    const int N2 = A.N2;
    const GridHdr *hdr = pair_hdr(A, pair);
    int *cursor = pair_cursor(A, pair);
    float4 *sorted = pair_sorted(A, pair);
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N2) return;
    const float x = p2[i], y = p2[(size_t)N2 + i], z = p2[2 * (size_t)N2 + i];
    const int cx = cell_coord(x, hdr->lox, hdr->inv_cs, hdr->gx);
    const int cy = cell_coord(y, hdr->loy, hdr->inv_cs, hdr->gy);
    const int pos = atomicAdd(&cursor[cy * hdr->gx + cx], 1);
    sorted[pos] = make_float4(x, y, z, __int_as_float(i));
}

__global__ __launch_bounds__(256) void nn1_moments_kernel(PairArgs A, float r2) {
    const int pair = blockIdx.y;
    const float *p1 = pair_p1(A, pair);
    const int N1 = A.N1;
    const float *Rt = A.Rt + (size_t)pair * A.rt_stride;  // 12 floats: R row-major, T
    GridHdr *hdr = pair_hdr(A, pair);
    const int *start = pair_count(A, pair);
    const float4 *sorted = pair_sorted(A, pair);
    __shared__ double sred[4][10];
    const int i = blockIdx.x * 256 + threadIdx.x;
    double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    if (i < N1) {
        const float x = p1[i], y = p1[(size_t)N1 + i], z = p1[2 * (size_t)N1 + i];
        // R @ pcd1 + T in fp32 (sgemm k-order fma chain, then the broadcast add)
        const float qx = fmaf(Rt[2], z, fmaf(Rt[1], y, Rt[0] * x)) + Rt[9];
        const float qy = fmaf(Rt[5], z, fmaf(Rt[4], y, Rt[3] * x)) + Rt[10];
        const float qz = fmaf(Rt[8], z, fmaf(Rt[7], y, Rt[6] * x)) + Rt[11];
        const int gx = hdr->gx, gy = hdr->gy;
        const int cx = (int)floorf((qx - hdr->lox) * hdr->inv_cs), cy = (int)floorf((qy - hdr->loy) * hdr->inv_cs);
        float best = __builtin_inff();
        int bi = 0x7fffffff;
        float bx = 0, by = 0, bz = 0;
        for (int yy = max(cy - 1, 0); yy <= min(cy + 1, gy - 1); ++yy)
            for (int xx = max(cx - 1, 0); xx <= min(cx + 1, gx - 1); ++xx) {
                const int c = yy * gx + xx;
                for (int p = start[c]; p < start[c + 1]; ++p) {
                    const float4 t4 = sorted[p];
                    const float dx = qx - t4.x, dy = qy - t4.y, dz = qz - t4.z;
                    const float d = (dx * dx + dy * dy) + dz * dz;
                    const int oi = __float_as_int(t4.w);
                    if (d < best || (d == best && oi < bi)) best = d, bi = oi, bx = t4.x, by = t4.y, bz = t4.z;
                }
            }
        if (best <= r2) {
            const double X = bx, Y = by, Z = bz;
            m[0] = 1.0, m[1] = X, m[2] = Y, m[3] = Z, m[4] = X * X, m[5] = Y * Y, m[6] = Z * Z;
            m[7] = X * Y, m[8] = X * Z, m[9] = Y * Z;
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
    if (threadIdx.x < 10) {
        const double v = (sred[0][threadIdx.x] + sred[1][threadIdx.x]) + (sred[2][threadIdx.x] + sred[3][threadIdx.x]);
        if (v != 0.0) atomicAdd(&hdr->sums[threadIdx.x], v);
    }
}

__global__ void infomat_finalize_kernel(PairArgs A) {
    if (threadIdx.x != 0) return;
    const GridHdr *hdr = pair_hdr(A, blockIdx.x);
    float *out = A.out + (size_t)blockIdx.x * A.out_stride;
    const double *s = hdr->sums;
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

static size_t ws_slice_bytes(int N2) {
    size_t b = 256 + 2 * sizeof(int) * (size_t)(GMAX * GMAX + 1) + 8 + sizeof(float4) * (size_t)N2;
    return (b + 255) & ~(size_t)255;
}

extern "C" size_t dpm_infomat_workspace_bytes(int n_pairs, int N1, int N2) {
    (void)N1;
    return 256 + (size_t)n_pairs * ws_slice_bytes(N2);
}

static int launch_infomat(PairArgs A, int n_pairs, double radius, hipStream_t st) {
    hipLaunchKernelGGL(grid_setup_kernel, dim3(n_pairs), dim3(1024), 0, st, A, (float)radius);
    hipLaunchKernelGGL(grid_count_kernel, dim3(dpm_cdiv(A.N2, 256), n_pairs), dim3(256), 0, st, A);
    hipLaunchKernelGGL(grid_scan_kernel, dim3(n_pairs), dim3(1024), 0, st, A);
    hipLaunchKernelGGL(grid_scatter_kernel, dim3(dpm_cdiv(A.N2, 256), n_pairs), dim3(256), 0, st, A);
    hipLaunchKernelGGL(nn1_moments_kernel, dim3(dpm_cdiv(A.N1, 256), n_pairs), dim3(256), 0, st, A,
                       (float)(radius * radius));
    hipLaunchKernelGGL(infomat_finalize_kernel, dim3(n_pairs), dim3(64), 0, st, A);
    return dpm_launch_status();
}

extern "C" int dpm_information_matrix(const float *pcd1, int N1, const float *pcd2, int N2, const float *Rt,
                                      double radius, float *out6x6, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(pcd1 && pcd2 && Rt && out6x6 && workspace && N1 >= 1 && N2 >= 1 && radius > 0.0);
    PairArgs A{};
    A.pcd1 = pcd1, A.pcd2 = pcd2, A.f1 = nullptr, A.f2 = nullptr, A.stride1 = 0, A.stride2 = 0;
    A.Rt = Rt, A.rt_stride = 0, A.out = out6x6, A.out_stride = 0;
    A.ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), A.ws_stride = ws_slice_bytes(N2);
    A.N1 = N1, A.N2 = N2;
    return launch_infomat(A, 1, radius, (hipStream_t)stream);
}

extern "C" int dpm_information_matrix_batched(const float *pcd, int N, const int32_t *src_frame,
                                              const int32_t *dst_frame, int n_pairs, const float *Rt, int rt_stride,
                                              double radius, float *out, int out_stride, void *workspace,
                                              dpm_stream_t stream) {
    DPM_CHECK_ARG(pcd && src_frame && dst_frame && Rt && out && workspace && N >= 1 && n_pairs >= 1 && radius > 0.0);
    DPM_CHECK_ARG(rt_stride >= 12 && out_stride >= 36);
    PairArgs A{};
    A.pcd1 = pcd, A.pcd2 = pcd, A.f1 = src_frame, A.f2 = dst_frame, A.stride1 = 3LL * N, A.stride2 = 3LL * N;
    A.Rt = Rt, A.rt_stride = rt_stride, A.out = out, A.out_stride = out_stride;
    A.ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255), A.ws_stride = ws_slice_bytes(N);
    A.N1 = N, A.N2 = N;
    return launch_infomat(A, n_pairs, radius, (hipStream_t)stream);
}
