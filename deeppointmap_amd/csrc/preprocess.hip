// Scan pre-processing on the GPU: the first transforms of every shipped inference config,
//   VoxelSample(voxel_size, retention='first') -> DistanceSample(min_dis, max_dis) -> CoordinatesNormalization(ratio)
// (reference dataloader/transforms.py:322-356, 387-397, 400-407; configs/infer/*.yaml:21-27).
//
// VoxelSample('first') keeps, for every occupied voxel, the point with the smallest index, and np.unique returns the
// voxels in ascending voxel-id order (transforms.py:346).  Instead of sorting 120 k keys, the voxel grid itself lives
// in HBM (one int32 per cell, tens of MB -- nothing on a 288 GB part): atomicMin(first index) per point, then an
// ordered stream compaction over the cells, which IS ascending voxel-id order.  The distance filter is a per-point
// predicate that preserves order, so it is folded into the same compaction, and the division by `ratio` into the
// final gather.  Arithmetic mirrors numpy/torch fp32 on the CPU: (x - min) / voxel_size truncated toward zero,
// true division for the normalisation (torch's in-place `/=` on CPU tensors).
#include "dpm_common.h"

namespace {

constexpr int CH = 4096;  // cells per compaction block

struct PreHdr {
    float lo[3], hi[3];
    int X, Y, Z;
    long long ncell;
    int n_out;
    int overflow;  // 1: the voxel grid does not fit the workspace
};

__global__ __launch_bounds__(1024) void pre_bbox_kernel(const float *__restrict__ xyz, int N, int stride, float vs,
                                                        long long max_cells, PreHdr *__restrict__ hdr) {
    __shared__ float red[6][16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-lo[0], -lo[0], -lo[0]};
    for (int i = t; i < N; i += 1024)
        for (int a = 0; a < 3; ++a) {
            const float v = xyz[(size_t)i * stride + a];
            lo[a] = fminf(lo[a], v), hi[a] = fmaxf(hi[a], v);
        }
    for (int a = 0; a < 3; ++a) {
        lo[a] = -wave_max_dpp(-lo[a]), hi[a] = wave_max_dpp(hi[a]);
        if (lane == 0) red[a][w] = lo[a], red[3 + a][w] = hi[a];
    }
    __syncthreads();
    if (t == 0) {
        for (int a = 0; a < 3; ++a) {
            for (int k = 0; k < 16; ++k) lo[a] = fminf(lo[a], red[a][k]), hi[a] = fmaxf(hi[a], red[3 + a][k]);
            hdr->lo[a] = lo[a], hdr->hi[a] = hi[a];
        }
        const int X = (int)((hi[0] - lo[0]) / vs) + 1, Y = (int)((hi[1] - lo[1]) / vs) + 1, Z = (int)((hi[2] - lo[2]) / vs) + 1;
        hdr->X = X, hdr->Y = Y, hdr->Z = Z;
        hdr->ncell = (long long)X * Y * Z;
        hdr->overflow = (N <= 0 || hdr->ncell > max_cells || hdr->ncell > 0x7fffffffLL) ? 1 : 0;
        hdr->n_out = 0;
    }
}

__global__ __launch_bounds__(256) void pre_fill_kernel(const PreHdr *__restrict__ hdr, int *__restrict__ grid) {
    if (hdr->overflow) return;
    const long long n = hdr->ncell;
    for (long long c = (long long)blockIdx.x * 256 + threadIdx.x; c < n; c += (long long)gridDim.x * 256) grid[c] = 0x7fffffff;
}

__global__ __launch_bounds__(256) void pre_mark_kernel(const float *__restrict__ xyz, int N, int stride, float vs,
                                                       const PreHdr *__restrict__ hdr, int *__restrict__ grid) {
    if (hdr->overflow) return;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float *p = xyz + (size_t)i * stride;
    const int vx = (int)((p[0] - hdr->lo[0]) / vs), vy = (int)((p[1] - hdr->lo[1]) / vs), vz = (int)((p[2] - hdr->lo[2]) / vs);
    const int id = vx + vy * hdr->X + vz * hdr->X * hdr->Y;
    atomicMin(&grid[id], i);
}

__device__ __forceinline__ bool keep_point(const float *__restrict__ xyz, int stride, int i, float dmin, float dmax) {
    const float x = xyz[(size_t)i * stride], y = xyz[(size_t)i * stride + 1], z = xyz[(size_t)i * stride + 2];
    const float d = sqrtf((x * x + y * y) + z * z);
    return dmin <= d && d <= dmax;
}

__global__ __launch_bounds__(256) void pre_count_kernel(const float *__restrict__ xyz, int stride, float dmin, float dmax,
                                                        const PreHdr *__restrict__ hdr, const int *__restrict__ grid,
                                                        int *__restrict__ bcount) {
    __shared__ int s[4];
    if (hdr->overflow) return;
    const long long c0 = (long long)blockIdx.x * CH;
    if (c0 >= hdr->ncell) return;
    int cnt = 0;
    for (int k = threadIdx.x; k < CH; k += 256) {
        const long long c = c0 + k;
        if (c < hdr->ncell) {
            const int g = grid[c];
            if (g != 0x7fffffff && keep_point(xyz, stride, g, dmin, dmax)) ++cnt;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cnt += __shfl_xor(cnt, off, 64);
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) bcount[blockIdx.x] = s[0] + s[1] + s[2] + s[3];
}

__global__ __launch_bounds__(1024) void pre_scan_kernel(PreHdr *__restrict__ hdr, int *__restrict__ bcount, int max_blocks) {
    __shared__ int wsum[16];
    if (hdr->overflow) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int nblk = (int)((hdr->ncell + CH - 1) / CH);
    const int per = (max_blocks + 1023) / 1024;
    const int b0 = t * per, b1 = min(b0 + per, nblk);
    int s = 0;
    for (int b = b0; b < b1; ++b) s += bcount[b];
    int inc = s;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int run = inc - s;
    for (int k = 0; k < w; ++k) run += wsum[k];
    for (int b = b0; b < b1; ++b) {
        const int v = bcount[b];
        bcount[b] = run;
        run += v;
    }
    if (t == 1023) hdr->n_out = run;
}

__global__ __launch_bounds__(256) void pre_write_kernel(const float *__restrict__ xyz, int stride, float dmin, float dmax,
                                                        float ratio, const PreHdr *__restrict__ hdr,
                                                        const int *__restrict__ grid, const int *__restrict__ boff,
                                                        float *__restrict__ out_xyz, int32_t *__restrict__ out_idx,
                                                        int out_cap) {
    __shared__ int s_w[4];
    if (hdr->overflow) return;
    const long long c0 = (long long)blockIdx.x * CH;
    if (c0 >= hdr->ncell) return;
    // thread t owns 16 CONSECUTIVE cells so that the block-level order equals cell order
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    int idx[CH / 256], cnt = 0;
#pragma unroll
    for (int k = 0; k < CH / 256; ++k) {
        const long long c = c0 + (long long)t * (CH / 256) + k;
        int g = 0x7fffffff;
        if (c < hdr->ncell) g = grid[c];
        if (g != 0x7fffffff && !keep_point(xyz, stride, g, dmin, dmax)) g = 0x7fffffff;
        idx[k] = g;
        cnt += g != 0x7fffffff;
    }
    int inc = cnt;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    int pos = boff[blockIdx.x] + inc - cnt;
    for (int k = 0; k < w; ++k) pos += s_w[k];
#pragma unroll
    for (int k = 0; k < CH / 256; ++k) {
        const int g = idx[k];
        if (g == 0x7fffffff) continue;
        if (pos < out_cap) {
            const float *p = xyz + (size_t)g * stride;
            out_xyz[3 * (size_t)pos] = p[0] / ratio, out_xyz[3 * (size_t)pos + 1] = p[1] / ratio, out_xyz[3 * (size_t)pos + 2] = p[2] / ratio;
            if (out_idx) out_idx[pos] = g;
        }
        ++pos;
    }
}

}  // namespace

extern "C" size_t dpm_preprocess_workspace_bytes(long long max_cells) {
    return 1024 + sizeof(int) * (size_t)max_cells + sizeof(int) * (size_t)(max_cells / CH + 2);
}

extern "C" int dpm_preprocess_scan(const float *xyz, int N, int stride, double voxel_size, double min_dis, double max_dis,
                                   double ratio, long long max_cells, float *out_xyz, int32_t *out_idx, int out_capacity,
                                   int32_t *status /* [n_out, overflow] */, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && out_xyz && status && workspace && N >= 1 && stride >= 3 && voxel_size > 0.0 && ratio != 0.0);
    DPM_CHECK_ARG(max_cells >= CH && out_capacity >= 1);
    hipStream_t st = (hipStream_t)stream;
    uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    PreHdr *hdr = (PreHdr *)p;
    int *grid = (int *)(p + 256);
    const int max_blocks = (int)(max_cells / CH + 1);
    int *bcount = grid + max_cells;
    const float vs = (float)voxel_size;
    hipLaunchKernelGGL(pre_bbox_kernel, dim3(1), dim3(1024), 0, st, xyz, N, stride, vs, max_cells, hdr);
    hipLaunchKernelGGL(pre_fill_kernel, dim3(2048), dim3(256), 0, st, hdr, grid);
    hipLaunchKernelGGL(pre_mark_kernel, dim3(dpm_cdiv(N, 256)), dim3(256), 0, st, xyz, N, stride, vs, hdr, grid);
    hipLaunchKernelGGL(pre_count_kernel, dim3(max_blocks), dim3(256), 0, st, xyz, stride, (float)min_dis, (float)max_dis, hdr,
                       grid, bcount);
    hipLaunchKernelGGL(pre_scan_kernel, dim3(1), dim3(1024), 0, st, hdr, bcount, max_blocks);
    hipLaunchKernelGGL(pre_write_kernel, dim3(max_blocks), dim3(256), 0, st, xyz, stride, (float)min_dis, (float)max_dis,
                       (float)ratio, hdr, grid, bcount, out_xyz, out_idx, out_capacity);
    // status = [n_out, overflow]: two ints copied device-to-device so the caller reads them with its own sync
    hipError_t e = hipMemcpyAsync(status, &hdr->n_out, 2 * sizeof(int), hipMemcpyDeviceToDevice, st);
    if (e != hipSuccess) return (int)e;
    return dpm_launch_status();
}
