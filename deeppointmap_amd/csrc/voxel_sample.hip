// Sampler('voxel') of the reference's operator table (network/encoder/utils.py:150-207).
//
// What the reference computes per frame: points outside `sample_range` (and padded points, which it first moves to
// 2 * sample_range -- they still stretch the bounding box) are dropped; every remaining point gets the id of its voxel
// and the squared distance to that voxel's centre; after a torch.sort by that distance np.unique(return_index,
// return_counts) keeps, per occupied voxel, the first point in sorted order -- the point nearest the centre -- in
// ascending voxel-id order, with the voxel's population; if more than K voxels are occupied, torch.topk(population, K)
// picks the K fullest and their order is the output order.
//
// No sort here: the voxel grid lives in HBM (one 64-bit key + one counter per cell).  key = (distance bits << 32 | index)
// under atomicMin is "first after the sort by distance" whenever a voxel's nearest point is unique (non-negative floats
// order like their bit patterns), atomicAdd gives the population, an ordered stream compaction over the cells IS
// np.unique's ascending order, and the top-k over populations -- integers that tie all the time -- is torch.topk's CPU
// kernel replayed step by step (topk_emulate.h: which K survive AND the order they come out in follow libstdc++'s data
// movement).  When two points of a voxel are EXACTLY equally near its centre (lattice data, duplicated points) the
// reference's choice is whatever its unstable torch.sort (std::sort over the frame's N distances) left first: frames
// where that happens are detected and re-keyed by one thread replaying that sort (slow path, exact).
// Arithmetic mirrors torch's fp32 CPU kernels operation by operation (the library is built with -ffp-contract=off).
#include <vector>

#include "dpm_common.h"
#include "topk_emulate.h"

namespace {

constexpr int CH = 4096;  // cells per compaction block (256 threads x 16 cells)

struct VoxHdr {  // 8 floats per frame
    float lo[3];
    float X, Y, Z;  // integer-valued floats, as in the reference (utils.py:159-161)
    float over;     // 1: the grid does not fit max_cells (set by the select entry point)
    float pad;      // 1: some voxel's nearest distance is shared by several points (slow path taken)
};

__device__ __forceinline__ void vox_point(const float *__restrict__ p, bool padded, float two_r, float &x, float &y,
                                          float &z) {
    x = padded ? two_r : p[0], y = padded ? two_r : p[1], z = padded ? two_r : p[2];
}

__global__ __launch_bounds__(1024) void vox_bounds_kernel(const float *__restrict__ points, const uint8_t *__restrict__ padding,
                                                          int N, int D, float vs, float two_r, VoxHdr *__restrict__ hdr) {
    __shared__ float red[6][16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *pts = points + (size_t)b * N * D;
    const uint8_t *pad = padding + (size_t)b * N;
    float lo[3] = {__builtin_inff(), __builtin_inff(), __builtin_inff()}, hi[3] = {-lo[0], -lo[0], -lo[0]};
    for (int i = t; i < N; i += 1024) {
        float v[3];
        vox_point(pts + (size_t)i * D, pad[i] != 0, two_r, v[0], v[1], v[2]);
        for (int a = 0; a < 3; ++a) lo[a] = fminf(lo[a], v[a]), hi[a] = fmaxf(hi[a], v[a]);
    }
    for (int a = 0; a < 3; ++a) {
        lo[a] = -wave_max_dpp(-lo[a]), hi[a] = wave_max_dpp(hi[a]);
        if (lane == 0) red[a][w] = lo[a], red[3 + a][w] = hi[a];
    }
    __syncthreads();
    if (t == 0) {
        VoxHdr h;
        float dim[3];
        for (int a = 0; a < 3; ++a) {
            for (int k = 0; k < 16; ++k) lo[a] = fminf(lo[a], red[a][k]), hi[a] = fmaxf(hi[a], red[3 + a][k]);
            h.lo[a] = lo[a];
            dim[a] = truncf((hi[a] - lo[a]) / vs) + 1.0f;  // torch.div(max - min, voxel_size, 'trunc') + 1
        }
        h.X = dim[0], h.Y = dim[1], h.Z = dim[2], h.over = 0.f, h.pad = 0.f;
        hdr[b] = h;
    }
}

__device__ __forceinline__ long long vox_ncell(const VoxHdr &h) { return (long long)((double)h.X * (double)h.Y * (double)h.Z); }

// voxel id, squared distance to the voxel centre and the range predicate of one point, in the reference's fp32 steps
struct VoxPoint {
    long long id;
    float dis;
    bool inside;
};
__device__ __forceinline__ VoxPoint vox_classify(const VoxHdr &h, float x, float y, float z, float vs, float half_vs, float r2) {
    VoxPoint p;
    p.inside = (x * x + y * y) + z * z <= r2;  // dis_mask (utils.py:163)
    const float rx = x - h.lo[0], ry = y - h.lo[1], rz = z - h.lo[2];
    const int vx = (int)truncf(rx / vs), vy = (int)truncf(ry / vs), vz = (int)truncf(rz / vs);
    // voxel_id in float arithmetic, left to right, as utils.py:168 evaluates it
    p.id = (long long)(int)(((float)vx + (float)vy * h.X) + ((float)vz * h.X) * h.Y);
    const float dx = (rx - (float)vx * vs) - half_vs, dy = (ry - (float)vy * vs) - half_vs, dz = (rz - (float)vz * vs) - half_vs;
    p.dis = (dx * dx + dy * dy) + dz * dz;
    return p;
}

// TIE = false: key = (distance bits, index) minimum and the population of every cell.
// TIE = true (second pass): flags the frame (hdr.pad) if some voxel's nearest distance is shared by two of its points.
template <bool TIE>
__global__ __launch_bounds__(256) void vox_scatter_kernel(const float *__restrict__ points, const uint8_t *__restrict__ padding,
                                                          int N, int D, float vs, float half_vs, float two_r, float r2,
                                                          VoxHdr *__restrict__ hdr, long long max_cells,
                                                          unsigned long long *__restrict__ keys, unsigned *__restrict__ cnt) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    const VoxHdr h = hdr[b];
    const long long nc = vox_ncell(h);
    if (!(nc >= 1 && nc <= max_cells)) {  // also catches NaN / inf boxes
        if (i == 0 && !TIE) hdr[b].over = 1.f;  // (this kernel reads lo / X / Y / Z of the header only)
        return;
    }
    if (i >= N) return;
    float x, y, z;
    vox_point(points + ((size_t)b * N + i) * D, padding[(size_t)b * N + i] != 0, two_r, x, y, z);
    const VoxPoint p = vox_classify(h, x, y, z, vs, half_vs, r2);
    if (!p.inside || p.id < 0 || p.id > nc) return;  // ids outside the grid cannot happen for finite inputs
    const size_t cell = (size_t)b * (size_t)(max_cells + 1) + (size_t)p.id;
    const unsigned long long key = ((unsigned long long)__float_as_uint(p.dis) << 32) | (unsigned)i;
    if (!TIE) {
        atomicMin(&keys[cell], key);
        atomicAdd(&cnt[cell], 1u);
    } else {
        const unsigned long long best = keys[cell];
        if ((best >> 32) == (key >> 32) && best != key) hdr[b].pad = 1.f;  // every writer stores the same value; read by later kernels only
    }
}

// Slow path of a flagged frame: one thread replays torch.sort's std::sort over ALL N distances of the frame (padded and
// out-of-range points take part in its data movement), then every voxel is re-keyed by (position in that order, index).
__global__ __launch_bounds__(64) void vox_resort_kernel(const float *__restrict__ points, const uint8_t *__restrict__ padding,
                                                        int N, int D, float vs, float half_vs, float two_r, float r2,
                                                        const VoxHdr *__restrict__ hdr, long long max_cells,
                                                        unsigned long long *__restrict__ keys, VI *__restrict__ scratch) {
    constexpr int LDS_ROWS = 7680;
    __shared__ VI s_row[LDS_ROWS];
    const int b = blockIdx.x, lane = threadIdx.x;
    const VoxHdr h = hdr[b];
    if (h.over != 0.f || h.pad == 0.f) return;
    const long long nc = vox_ncell(h);
    VI *row = N <= LDS_ROWS ? s_row : scratch + (size_t)b * N;
    const float *pts = points + (size_t)b * N * D;
    const uint8_t *pad = padding + (size_t)b * N;
    unsigned long long *k = keys + (size_t)b * (size_t)(max_cells + 1);
    for (int i = lane; i < N; i += 64) {
        float x, y, z;
        vox_point(pts + (size_t)i * D, pad[i] != 0, two_r, x, y, z);
        VI e;
        e.v = vox_classify(h, x, y, z, vs, half_vs, r2).dis, e.i = i;
        row[i] = e;
    }
    for (long long c = lane; c <= nc; c += 64) k[c] = ~0ull;
    __threadfence();
    __syncthreads();
    if (lane == 0) vi_sort<false>(row, 0, N);
    __threadfence();
    __syncthreads();
    for (int j = lane; j < N; j += 64) {
        const int i = row[j].i;
        float x, y, z;
        vox_point(pts + (size_t)i * D, pad[i] != 0, two_r, x, y, z);
        const VoxPoint p = vox_classify(h, x, y, z, vs, half_vs, r2);
        if (p.inside && p.id >= 0 && p.id <= nc) atomicMin(&k[p.id], ((unsigned long long)(unsigned)j << 32) | (unsigned)i);
    }
}

__global__ __launch_bounds__(256) void vox_count_kernel(const VoxHdr *__restrict__ hdr, long long max_cells,
                                                        const unsigned *__restrict__ cnt, int nchunk,
                                                        int *__restrict__ bcount) {
    __shared__ int s[4];
    const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x;
    const VoxHdr h = hdr[b];
    const long long nc = h.over != 0.f ? 0 : vox_ncell(h) + 1;
    const unsigned *c = cnt + (size_t)b * (size_t)(max_cells + 1);
    int n = 0;
    for (int k = 0; k < CH / 256; ++k) {
        const long long cell = (long long)chunk * CH + k * 256 + t;
        n += (cell < nc && c[cell] != 0u) ? 1 : 0;
    }
    for (int off = 32; off > 0; off >>= 1) n += __shfl_xor(n, off, 64);
    if ((t & 63) == 0) s[t >> 6] = n;
    __syncthreads();
    if (t == 0) bcount[(size_t)b * nchunk + chunk] = s[0] + s[1] + s[2] + s[3];
}

// exclusive scan of a frame's chunk counts (in place) + number of occupied voxels
__global__ __launch_bounds__(1024) void vox_scan_kernel(int nchunk, int *__restrict__ bcount, int32_t *__restrict__ n_unique) {
    __shared__ int wsum[16];
    __shared__ int carry;
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    int *c = bcount + (size_t)b * nchunk;
    if (t == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < nchunk; base += 1024) {
        const int i = base + t;
        const int v = i < nchunk ? c[i] : 0;
        int inc = v;
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        int before = carry;
        for (int k = 0; k < w; ++k) before += wsum[k];
        if (i < nchunk) c[i] = before + inc - v;
        __syncthreads();
        if (t == 1023) carry = before + inc;
        __syncthreads();
    }
    if (t == 0) n_unique[b] = carry;
}

// ordered compaction of one chunk: thread t owns cells [16 t, 16 t + 16) of the chunk
__global__ __launch_bounds__(256) void vox_write_kernel(const VoxHdr *__restrict__ hdr, long long max_cells,
                                                        const unsigned long long *__restrict__ keys,
                                                        const unsigned *__restrict__ cnt, int nchunk,
                                                        const int *__restrict__ boff, int N, int32_t *__restrict__ ufirst,
                                                        int32_t *__restrict__ ucnt) {
    __shared__ int wsum[4];
    const int b = blockIdx.y, chunk = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const VoxHdr h = hdr[b];
    const long long nc = h.over != 0.f ? 0 : vox_ncell(h) + 1;
    const size_t base = (size_t)b * (size_t)(max_cells + 1);
    const long long c0 = (long long)chunk * CH + t * 16;
    unsigned occ = 0;
    for (int k = 0; k < 16; ++k) occ |= (c0 + k < nc && cnt[base + c0 + k] != 0u) ? (1u << k) : 0u;
    const int mine = __popc(occ);
    int inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int pos = boff[(size_t)b * nchunk + chunk] + inc - mine;
    for (int k = 0; k < w; ++k) pos += wsum[k];
    for (int k = 0; k < 16; ++k)
        if ((occ >> k) & 1u) {
            ufirst[(size_t)b * N + pos] = (int32_t)(unsigned)(keys[base + c0 + k] & 0xffffffffull);
            ucnt[(size_t)b * N + pos] = (int32_t)cnt[base + c0 + k];
            ++pos;
        }
}

// sel (B, cap): original indices of the sampled points in the reference's output order, -1 = padding
__global__ __launch_bounds__(64) void vox_pick_kernel(const VoxHdr *__restrict__ hdr, const int32_t *__restrict__ n_unique,
                                                      const int32_t *__restrict__ ufirst, const int32_t *__restrict__ ucnt,
                                                      int N, int K, int cap, VI *__restrict__ scratch,
                                                      int32_t *__restrict__ sel) {
    constexpr int LDS_ROWS = 7680;  // rows of up to this many voxels are replayed in LDS (60 KB), longer ones in HBM
    __shared__ VI s_row[LDS_ROWS];
    const int b = blockIdx.x, lane = threadIdx.x;
    const int nu = n_unique[b];
    const int32_t *uf = ufirst + (size_t)b * N, *uc = ucnt + (size_t)b * N;
    int32_t *out = sel + (size_t)b * cap;
    if (K < 0 || nu <= K) {  // utils.py:187: no top-k, ascending voxel id
        for (int j = lane; j < cap; j += 64) out[j] = j < nu ? uf[j] : -1;
        return;
    }
    VI *row = nu <= LDS_ROWS ? s_row : scratch + (size_t)b * N;
    for (int j = lane; j < nu; j += 64) {
        VI e;
        e.v = (float)uc[j], e.i = j;  // populations are below 2^24: the float order is the integer order
        row[j] = e;
    }
    __syncthreads();
    if (lane == 0) vi_topk_sorted<true>(row, nu, K);
    __syncthreads();
    for (int j = lane; j < cap; j += 64) out[j] = j < K ? uf[row[j].i] : -1;
}

}  // namespace

extern "C" {

int dpm_voxel_sampler_bounds(const float *points, const uint8_t *padding, int B, int N, int D, double voxel_size,
                             double sample_range, float *hdr, dpm_stream_t stream) {
    DPM_CHECK_ARG(points && padding && hdr && B > 0 && N > 0 && D >= 3 && voxel_size > 0 && sample_range > 0);
    hipStream_t st = (hipStream_t)stream;
    vox_bounds_kernel<<<B, 1024, 0, st>>>(points, padding, N, D, (float)voxel_size, (float)(2 * sample_range), (VoxHdr *)hdr);
    return dpm_launch_status();
}

static size_t vox_align(size_t x) { return (x + 255) & ~(size_t)255; }

size_t dpm_voxel_sampler_workspace_bytes(int B, int N, long long max_cells) {
    if (B <= 0 || N <= 0 || max_cells <= 0) return 0;
    const size_t cells = (size_t)B * (size_t)(max_cells + 1), rows = (size_t)B * N;
    const size_t nchunk = (size_t)((max_cells + 1 + CH - 1) / CH);
    return vox_align(cells * 8) + vox_align(cells * 4) + 2 * vox_align(rows * 4) + vox_align(rows * sizeof(VI)) +
           vox_align((size_t)B * nchunk * 4);
}

int dpm_voxel_sampler_select(const float *points, const uint8_t *padding, int B, int N, int D, double voxel_size,
                             double sample_range, float *hdr, long long max_cells, int K, int32_t *sel, int cap,
                             int32_t *n_unique, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(points && padding && hdr && sel && n_unique && workspace && B > 0 && N > 0 && D >= 3);
    DPM_CHECK_ARG(voxel_size > 0 && sample_range > 0 && max_cells > 0 && max_cells < 0x7fffffffLL && N <= (1 << 24));
    DPM_CHECK_ARG(K < 0 ? cap == N : (K >= 1 && cap == K));
    hipStream_t st = (hipStream_t)stream;
    const size_t cells = (size_t)B * (size_t)(max_cells + 1), rows = (size_t)B * N;
    const int nchunk = (int)((max_cells + 1 + CH - 1) / CH);
    char *w = (char *)workspace;
    unsigned long long *keys = (unsigned long long *)w;
    w += vox_align(cells * 8);
    unsigned *cnt = (unsigned *)w;
    w += vox_align(cells * 4);
    int32_t *ufirst = (int32_t *)w;
    w += vox_align(rows * 4);
    int32_t *ucnt = (int32_t *)w;
    w += vox_align(rows * 4);
    VI *scratch = (VI *)w;
    w += vox_align(rows * sizeof(VI));
    int *bcount = (int *)w;
    hipError_t e = hipMemsetAsync(keys, 0xff, cells * 8, st);
    if (e == hipSuccess) e = hipMemsetAsync(cnt, 0, cells * 4, st);
    if (e != hipSuccess) return (int)e;
    const float vs = (float)voxel_size, half_vs = (float)(voxel_size / 2), two_r = (float)(2 * sample_range),
                r2 = (float)(sample_range * sample_range);
    const dim3 pgrid(dpm_cdiv(N, 256), B);
    vox_scatter_kernel<false><<<pgrid, 256, 0, st>>>(points, padding, N, D, vs, half_vs, two_r, r2, (VoxHdr *)hdr, max_cells, keys, cnt);
    vox_scatter_kernel<true><<<pgrid, 256, 0, st>>>(points, padding, N, D, vs, half_vs, two_r, r2, (VoxHdr *)hdr, max_cells, keys, cnt);
    vox_resort_kernel<<<B, 64, 0, st>>>(points, padding, N, D, vs, half_vs, two_r, r2, (const VoxHdr *)hdr, max_cells, keys, scratch);
    vox_count_kernel<<<dim3(nchunk, B), 256, 0, st>>>((const VoxHdr *)hdr, max_cells, cnt, nchunk, bcount);
    vox_scan_kernel<<<B, 1024, 0, st>>>(nchunk, bcount, n_unique);
    vox_write_kernel<<<dim3(nchunk, B), 256, 0, st>>>((const VoxHdr *)hdr, max_cells, keys, cnt, nchunk, bcount, N, ufirst,
                                                      ucnt);
    vox_pick_kernel<<<B, 64, 0, st>>>((const VoxHdr *)hdr, n_unique, ufirst, ucnt, N, K, cap, scratch, sel);
    return dpm_launch_status();
}

// Host-side run of the torch.topk replay the kernels use (same source, compiled for the host): lets the CPU test suite
// hold topk_emulate.h to torch.topk without a GPU.  out_idx (k): indices in output order.
int dpm_host_topk_replay(const float *values, int n, int k, int largest, int32_t *out_idx) {
    DPM_CHECK_ARG(values && out_idx && n >= 1 && k >= 1 && k <= n);
    std::vector<VI> row((size_t)n);
    for (int i = 0; i < n; ++i) row[i].v = values[i], row[i].i = i;
    if (largest) vi_topk_sorted<true>(row.data(), n, k);
    else vi_topk_sorted<false>(row.data(), n, k);
    for (int i = 0; i < k; ++i) out_idx[i] = row[i].i;
    return DPM_OK;
}


// torch.sort(values, stable=False) on one row (CPU kernel: std::sort over (value, index) pairs, NOT stable from 17
// elements up), replayed by the same source: out_idx (n) = the permutation torch returns.
int dpm_host_sort_replay(const float *values, int n, int descending, int32_t *out_idx) {
    DPM_CHECK_ARG(values && out_idx && n >= 1);
    std::vector<VI> row((size_t)n);
    for (int i = 0; i < n; ++i) row[i].v = values[i], row[i].i = i;
    if (descending) vi_sort<true>(row.data(), 0, n);
    else vi_sort<false>(row.data(), 0, n);
    for (int i = 0; i < n; ++i) out_idx[i] = row[i].i;
    return DPM_OK;
}

}  // extern "C"
