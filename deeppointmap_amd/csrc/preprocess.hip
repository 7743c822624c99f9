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
    // torch.norm(xyz, p=2, dim=1) on the CPU (DistanceSample, dataloader/transforms.py:394) accumulates acc = fma(v, v, acc) in
    // float over x, y, z -- its vectorised reduction is built with FMA contraction -- and takes a float sqrt: measured on the
    // build host, 400 000 random points, this chain equals it bit for bit, (x*x + y*y) + z*z differs in 10 % of them by one
    // ulp, which moves a point sitting exactly on the crop radius across it (scripts/fuzz_preprocess.py: 2 of 2 000 scans)
    const float d = sqrtf(fmaf(z, z, fmaf(y, y, x * x)));
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

// ------------------------------------------------------------------------------------------
// LowPassFilter similarity (transforms.py:279-281): sim[i] = sum of the `flux` largest |n_i . n_j| over the K
// nearest neighbours j of i.  One thread per point.
// ------------------------------------------------------------------------------------------
constexpr int FLUX_MAX = 8;

__global__ __launch_bounds__(256) void lowpass_sim_kernel(const float *__restrict__ normals,
                                                          const int32_t *__restrict__ idx, int N, int K, int flux,
                                                          float *__restrict__ sim) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const float nx = normals[3 * (size_t)i], ny = normals[3 * (size_t)i + 1], nz = normals[3 * (size_t)i + 2];
    float top[FLUX_MAX];
#pragma unroll
    for (int f = 0; f < FLUX_MAX; ++f) top[f] = -1.f;  // similarities are >= 0
    for (int k = 0; k < K; ++k) {
        const int j = idx[(size_t)i * K + k];
        // (grouped_normals @ normals.unsqueeze(-1)): a K=3 dot product accumulated in index order
        float v = fabsf(fmaf(normals[3 * (size_t)j + 2], nz, fmaf(normals[3 * (size_t)j + 1], ny, normals[3 * (size_t)j] * nx)));
#pragma unroll
        for (int f = 0; f < FLUX_MAX; ++f) {  // insertion into the descending list
            if (f < flux && v > top[f]) {
                const float t = top[f];
                top[f] = v, v = t;
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int f = 0; f < FLUX_MAX; ++f)
        if (f < flux) s += fmaxf(top[f], 0.f);
    sim[i] = s;
}

// ------------------------------------------------------------------------------------------
// Statistical cut + stable compaction shared by OutlierFilter and LowPassFilter (transforms.py:241-246,
// 282-287): mean and unbiased std of stat[0..N) (fp64 accumulation, rounded to fp32 like the torch scalars),
// mode 0 keeps stat <= mean + k*std, mode 1 keeps stat > mean - k*std; survivors keep their order.
// One 1024-thread workgroup (scans after voxel sampling have a few 10^4 points).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void stat_filter_kernel(const float *__restrict__ stat, int N, float k_std, int mode,
                                                           float ratio,
                                                           const float *__restrict__ xyz_in,
                                                           const int32_t *__restrict__ idx_in,
                                                           float *__restrict__ xyz_out, int32_t *__restrict__ idx_out,
                                                           int32_t *__restrict__ n_out) {
    __shared__ double s_red[16];
    __shared__ int s_cnt[16];
    __shared__ double s_bcast;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    auto block_sum = [&](double v) -> double {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        __syncthreads();
        if (lane == 0) s_red[w] = v;
        __syncthreads();
        if (t == 0) {
            double a = 0.0;
            for (int k = 0; k < 16; ++k) a += s_red[k];
            s_bcast = a;
        }
        __syncthreads();
        return s_bcast;
    };
    double a = 0.0;
    for (int i = t; i < N; i += 1024) a += (double)stat[i];
    const double mean = block_sum(a) / (double)N;
    double q = 0.0;
    for (int i = t; i < N; i += 1024) {
        const double d = (double)stat[i] - mean;
        q += d * d;
    }
    const double var = N > 1 ? block_sum(q) / (double)(N - 1) : 0.0;
    const float mean_f = (float)mean, std_f = (float)sqrt(var);
    const float thr = mode == 0 ? mean_f + k_std * std_f : mean_f - k_std * std_f;
    auto keep = [&](int i) -> bool { return i < N && (mode == 0 ? stat[i] <= thr : stat[i] > thr); };
    // stable compaction: wave w owns the contiguous range [w*per, (w+1)*per) and walks it 64 points at a time
    const int per = ((N + 15) / 16 + 63) & ~63, i0 = min(w * per, N), i1 = min(i0 + per, N);
    int c = 0;
    for (int i = i0; i < i1; i += 64) c += __popcll(__ballot(keep(i + lane)));
    if (lane == 0) s_cnt[w] = c;
    __syncthreads();
    int pos = 0;
    for (int k = 0; k < w; ++k) pos += s_cnt[k];
    const unsigned long long ltm = (1ull << lane) - 1ull;
    for (int i = i0; i < i1; i += 64) {
        const int j = i + lane;
        const bool kp = keep(j);
        const unsigned long long m = __ballot(kp);
        if (kp) {
            const size_t o = (size_t)(pos + __popcll(m & ltm));
            // CoordinatesNormalization folded into the last filter: a true division (x / 1 is exact)
            xyz_out[3 * o] = xyz_in[3 * (size_t)j] / ratio, xyz_out[3 * o + 1] = xyz_in[3 * (size_t)j + 1] / ratio;
            xyz_out[3 * o + 2] = xyz_in[3 * (size_t)j + 2] / ratio;
            if (idx_out) idx_out[o] = idx_in ? idx_in[j] : j;
        }
        pos += __popcll(m);
    }
    if (t == 1023) *n_out = pos;
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

extern "C" int dpm_lowpass_similarity(const float *normals, const int32_t *idx, int N, int K, int flux, float *sim,
                                      dpm_stream_t stream) {
    DPM_CHECK_ARG(normals && idx && sim && N >= 1 && K >= 1 && flux >= 1 && flux <= K);
    if (flux > FLUX_MAX) return DPM_EUNSUPPORTED;
    hipLaunchKernelGGL(lowpass_sim_kernel, dim3(dpm_cdiv(N, 256)), dim3(256), 0, (hipStream_t)stream, normals, idx, N, K, flux,
                       sim);
    return dpm_launch_status();
}

extern "C" int dpm_stat_filter(const float *stat, int N, double k_std, int mode, double ratio, const float *xyz_in,
                               const int32_t *idx_in, float *xyz_out, int32_t *idx_out, int32_t *n_out,
                               dpm_stream_t stream) {
    DPM_CHECK_ARG(stat && xyz_in && xyz_out && n_out && N >= 1 && (mode == 0 || mode == 1) && ratio != 0.0);
    hipLaunchKernelGGL(stat_filter_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, stat, N, (float)k_std, mode,
                       (float)ratio, xyz_in,
                       idx_in, xyz_out, idx_out, n_out);
    return dpm_launch_status();
}
