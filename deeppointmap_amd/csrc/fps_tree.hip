// Sort-Tile-Recursive packing of a frame for the bucket kernel of fps.hip (farthest point sampling, N up to 65 536;
// replaces Sampler.fps / pytorch3d.sample_farthest_points, reference network/encoder/utils.py:210-285, together with it).
//
// Geometry.  The valid points of a frame are packed Sort-Tile-Recursive style: a counting sort by x (4096 bins)
// cut BY POSITION into <= 32 slabs of whole leaves, then every slab counting-sorted by y (512 bins) and cut by
// position into leaves of 64 points (= one wave-wide float4 load).  Leaves are addressed node-major: a node is a
// 4 x 4 block of (slab, tile) leaves, 64 nodes of 16 leaves -- bucket b of the sampling kernel is owned by wave b % 16, so
// the 16 neighbours of a block land on 16 different waves.  Compared with Z-ordered grid cells cut into runs,
// leaves are compact rectangles adapted to the point density: 6.5 instead of 10.8 leaves survive the pruning test
// of a round on the benchmark scans (the order inside a bin is arbitrary; no result depends on it).
// Rounds 1-2 also carried a one-wave-per-frame TREE sampling kernel over this packing (removed in round 3: exact, slower)
// and, until the end of round 3, the packing as ONE 1024-thread workgroup per frame (git history: "FPS packing: ...").
#include "fps_util.h"

#pragma clang fp contract(off)

namespace {

constexpr int TL = 1024;      // leaves per frame
constexpr int LEAF = 64;      // points per leaf
constexpr int XB = 4096;      // x bins of the first counting sort
constexpr int YB = 512;       // y bins per slab of the second
constexpr int MAXSLAB = 32;

// slab / tile geometry of a frame with `len` valid points
__device__ __forceinline__ void str_shape(int len, int &nsx, int &lps) {
    const int nl = (len + LEAF - 1) / LEAF;
    nsx = 1;
    while (nsx * nsx < nl) ++nsx;      // ceil(sqrt(nl)) <= 32
    nsx = min(nsx, MAXSLAB);
    lps = (nl + nsx - 1) / nsx;        // leaves per slab <= 32
}

// ------------------------------------------------------------------------------------------
// The packing as FIVE short chip-wide kernels.  As one 1024-thread workgroup per frame (rounds 2-3) it held 64 compute
// units with 64 KB of LDS each for 0.26 ms (0.34 ms under load) per batch; run twice per batch it lengthened the pipelined
// step by 0.19 ms -- such a kernel costs the other stages almost its whole duration.  Output layout (the bucket kernel's):
// points (x, y, z, original index) + a separate `closest` array (+inf; -1 in unused slots, whose index field is INT_MAX),
// TL * LEAF slots per frame.  Here a frame's
// points are cut into chunks of SC_CHUNK: bounds per chunk, x histogram per chunk, offsets per frame, scatter per chunk
// (the frame sorted by x bin, as above), then one workgroup per SLAB for the y sort into leaves.  Which points share a
// leaf depends on the arbitrary order inside an x bin, as it did before; no sampling result depends on it.
// ------------------------------------------------------------------------------------------
#ifndef DPM_SC_CHUNK
#define DPM_SC_CHUNK 8192
#endif
constexpr int SC_CHUNK = DPM_SC_CHUNK, SC_T = 256;

struct StrAux {       // per frame, behind the sort's scratch array
    float *part;      // [chunks][4]: lox, loy, hix, hiy of the chunk
    int *xhist;       // [chunks][XB]: counts, then start offsets
};
__host__ __device__ inline size_t str_aux_bytes(int N) {
    const size_t chunks = (size_t)(N + SC_CHUNK - 1) / SC_CHUNK;
    return (chunks * 4 * sizeof(float) + chunks * XB * sizeof(int) + 255) & ~(size_t)255;
}
__device__ __forceinline__ StrAux str_aux(char *aux, int b, int N) {
    char *p = aux + (size_t)b * str_aux_bytes(N);
    const int chunks = (N + SC_CHUNK - 1) / SC_CHUNK;
    return StrAux{(float *)p, (int *)(p + (size_t)chunks * 4 * sizeof(float))};
}
// the frame's bounds from the chunks' (min / max: exact in any order)
__device__ __forceinline__ void str_bounds(const float *part, int len, float &lox, float &loy, float &hix, float &hiy) {
    lox = loy = __builtin_inff(), hix = hiy = -__builtin_inff();
    for (int c = 0; c * SC_CHUNK < len; ++c) {
        lox = fminf(lox, part[4 * c]), loy = fminf(loy, part[4 * c + 1]);
        hix = fmaxf(hix, part[4 * c + 2]), hiy = fmaxf(hiy, part[4 * c + 3]);
    }
}
__device__ __forceinline__ int str_xbin(float x, float lox, float sxc) { return min(max((int)((x - lox) * sxc), 0), XB - 1); }
__device__ __forceinline__ int str_ybin(float y, float loy, float syc) { return min(max((int)((y - loy) * syc), 0), YB - 1); }

// MODE 0: chunk bounds + sentinel fill of the chunk's share of the frame's slots; 1: x histogram; 2: scatter by x bin
template <int MODE>
__global__ __launch_bounds__(SC_T) void str_chunk_kernel(const float *__restrict__ xyz_all, const int32_t *__restrict__ lengths,
                                                         int N, float4 *__restrict__ bpts_all, float *__restrict__ bclosest_all,
                                                         float4 *__restrict__ btmp_all, char *__restrict__ aux) {
    const int c = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *xyz = xyz_all + (size_t)b * N * 3;
    const int len = min(max(lengths[b], 0), N);
    const StrAux A = str_aux(aux, b, N);
    __shared__ int s_hist[MODE == 0 ? 1 : XB];
    __shared__ float s_red[4][SC_T / 64];
    if (MODE == 0) {
        const int chunks = gridDim.x, per = (TL * LEAF + chunks - 1) / chunks;
        float4 *pts = bpts_all + (size_t)b * TL * LEAF;
        float *closest = bclosest_all + (size_t)b * TL * LEAF;
        for (int i = c * per + t; i < min((c + 1) * per, TL * LEAF); i += SC_T)
            pts[i] = make_float4(0.f, 0.f, 0.f, __int_as_float(0x7fffffff)), closest[i] = -1.f;
    }
    const int p0 = c * SC_CHUNK, p1 = min(len, p0 + SC_CHUNK);
    if (p0 >= p1) return;
    float lox, loy, hix, hiy;
    if (MODE == 0) {
        lox = loy = __builtin_inff(), hix = hiy = -__builtin_inff();
    } else {
        str_bounds(A.part, len, lox, loy, hix, hiy);
        int *hist = A.xhist + (size_t)c * XB;
        for (int k = t; k < XB; k += SC_T) s_hist[k] = MODE == 2 ? hist[k] : 0;
        __syncthreads();
    }
    const float sxc = (hix > lox) ? (float)XB / (hix - lox) : 0.f;
    float4 *tmp = btmp_all + (size_t)b * N;
    constexpr int UB = 8;  // loads in flight per thread
    for (int i0 = p0 + t; i0 < p1; i0 += SC_T * UB) {
        float xs[UB], ys[UB], zs[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = min(i0 + u * SC_T, p1 - 1);
            xs[u] = xyz[3 * i], ys[u] = xyz[3 * i + 1];
            if (MODE == 2) zs[u] = xyz[3 * i + 2];
        }
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            const int i = i0 + u * SC_T;
            if (i >= p1) break;
            if (MODE == 0) {
                lox = fminf(lox, xs[u]), hix = fmaxf(hix, xs[u]), loy = fminf(loy, ys[u]), hiy = fmaxf(hiy, ys[u]);
            } else if (MODE == 1) {
                atomicAdd(&s_hist[str_xbin(xs[u], lox, sxc)], 1);
            } else {
                const int pos = atomicAdd(&s_hist[str_xbin(xs[u], lox, sxc)], 1);
                tmp[pos] = make_float4(xs[u], ys[u], zs[u], __int_as_float(i));
            }
        }
    }
    if (MODE == 0) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            lox = fminf(lox, __shfl_xor(lox, off, 64)), loy = fminf(loy, __shfl_xor(loy, off, 64));
            hix = fmaxf(hix, __shfl_xor(hix, off, 64)), hiy = fmaxf(hiy, __shfl_xor(hiy, off, 64));
        }
        if (lane == 0) s_red[0][w] = lox, s_red[1][w] = loy, s_red[2][w] = hix, s_red[3][w] = hiy;
        __syncthreads();
        if (t == 0) {
            for (int k = 1; k < SC_T / 64; ++k) {
                lox = fminf(lox, s_red[0][k]), loy = fminf(loy, s_red[1][k]);
                hix = fmaxf(hix, s_red[2][k]), hiy = fmaxf(hiy, s_red[3][k]);
            }
            A.part[4 * c] = lox, A.part[4 * c + 1] = loy, A.part[4 * c + 2] = hix, A.part[4 * c + 3] = hiy;
        }
    } else if (MODE == 1) {
        __syncthreads();
        int *hist = A.xhist + (size_t)c * XB;
        for (int k = t; k < XB; k += SC_T) hist[k] = s_hist[k];
    }
}

// x histograms [chunk][bin] -> start offset of every (chunk, bin) run in the x-sorted array, in place.  256 threads of 16
// bins each: a 16-wave workgroup waits for half a compute unit to fall free inside the pipeline (5 us alone, 38 us there).
__global__ __launch_bounds__(256) void str_xoffsets_kernel(const int32_t *__restrict__ lengths, int N, char *__restrict__ aux) {
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int len = min(max(lengths[b], 0), N), chunks = (len + SC_CHUNK - 1) / SC_CHUNK;
    if (chunks == 0) return;
    const StrAux A = str_aux(aux, b, N);
    __shared__ int s_wsum[4];
    constexpr int PER = XB / 256;  // bins per thread (16), as int4 groups
    int tot[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) tot[k] = 0;
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int g = 0; g < PER / 4; ++g) {
            const int4 h = *reinterpret_cast<const int4 *>(A.xhist + (size_t)c * XB + PER * t + 4 * g);
            tot[4 * g] += h.x, tot[4 * g + 1] += h.y, tot[4 * g + 2] += h.z, tot[4 * g + 3] += h.w;
        }
    }
    int tsum = 0;
#pragma unroll
    for (int k = 0; k < PER; ++k) tsum += tot[k];
    int inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
    int base = inc - tsum;
    for (int k = 0; k < w; ++k) base += s_wsum[k];
    int at[PER];
#pragma unroll
    for (int k = 0; k < PER; ++k) at[k] = base, base += tot[k];
    for (int c = 0; c < chunks; ++c) {
#pragma unroll
        for (int g = 0; g < PER / 4; ++g) {
            int4 *hp = reinterpret_cast<int4 *>(A.xhist + (size_t)c * XB + PER * t + 4 * g);
            const int4 h = *hp;
            *hp = make_int4(at[4 * g], at[4 * g + 1], at[4 * g + 2], at[4 * g + 3]);
            at[4 * g] += h.x, at[4 * g + 1] += h.y, at[4 * g + 2] += h.z, at[4 * g + 3] += h.w;
        }
    }
}

// one workgroup per slab (a run of lps * 64 positions of the x-sorted array): counting sort by y bin into leaves
__global__ __launch_bounds__(SC_T) void str_ysort_kernel(const int32_t *__restrict__ lengths, int N,
                                                         float4 *__restrict__ bpts_all, float *__restrict__ bclosest_all,
                                                         const float4 *__restrict__ btmp_all, char *__restrict__ aux) {
    const int slab = blockIdx.x, b = blockIdx.y, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int len = min(max(lengths[b], 0), N);
    if (len == 0) return;
    int nsx, lps;
    str_shape(len, nsx, lps);
    if (slab >= nsx) return;
    const int slab_pts = lps * LEAF, p0 = slab * slab_pts, p1 = min(len, p0 + slab_pts);
    if (p0 >= p1) return;
    const StrAux A = str_aux(aux, b, N);
    float lox, loy, hix, hiy;
    str_bounds(A.part, len, lox, loy, hix, hiy);
    const float syc = (hiy > loy) ? (float)YB / (hiy - loy) : 0.f;
    __shared__ int s_hist[YB];
    __shared__ int s_wsum[SC_T / 64];
    for (int k = t; k < YB; k += SC_T) s_hist[k] = 0;
    __syncthreads();
    const float4 *tmp = btmp_all + (size_t)b * N;
    constexpr int UB = (MAXSLAB * LEAF) / SC_T;  // a slab holds at most 32 leaves: 8 points per thread, kept in registers
    float4 e[UB];
#pragma unroll
    for (int u = 0; u < UB; ++u) e[u] = tmp[min(p0 + t + u * SC_T, p1 - 1)];
#pragma unroll
    for (int u = 0; u < UB; ++u)
        if (p0 + t + u * SC_T < p1) atomicAdd(&s_hist[str_ybin(e[u].y, loy, syc)], 1);
    __syncthreads();
    {   // exclusive scan over the 512 bins, two per thread
        const int c0 = s_hist[2 * t], c1 = s_hist[2 * t + 1];
        int inc = c0 + c1;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();
        int base = inc - (c0 + c1);
        for (int k = 0; k < w; ++k) base += s_wsum[k];
        s_hist[2 * t] = base, s_hist[2 * t + 1] = base + c0;
    }
    __syncthreads();
    float4 *pts = bpts_all + (size_t)b * TL * LEAF;
    float *closest = bclosest_all + (size_t)b * TL * LEAF;
#pragma unroll
    for (int u = 0; u < UB; ++u) {
        if (p0 + t + u * SC_T >= p1) break;
        const int j = atomicAdd(&s_hist[str_ybin(e[u].y, loy, syc)], 1);
        const int tile = j >> 6;
        const int leaf = (((slab >> 2) * 8 + (tile >> 2)) << 4) + ((slab & 3) << 2) + (tile & 3);
        const int q = leaf * LEAF + (j & 63);
        pts[q] = e[u];  // (x, y, z, original index)
        closest[q] = __builtin_inff();
    }
}

}  // namespace

// algo 5: the Sort-Tile-Recursive packing for fps.hip's bucket kernel (bucket = leaf; a 4 x 4 block of neighbouring
// leaves lands on 16 different waves).  Workspace: TL * LEAF float4 + TL * LEAF float per frame, then N float4 scratch.
size_t dpm_fps_str_bucket_workspace_bytes(int B, int N) {
    return (size_t)B * ((size_t)TL * LEAF * (sizeof(float4) + sizeof(float)) + (size_t)N * sizeof(float4) + str_aux_bytes(N)) + 1536;
}
// tmp: B * N float4 of scratch, followed (256-byte aligned) by B * str_aux_bytes(N) bytes for the chunk passes
int dpm_fps_str_bucket_sort(const float *xyz, const int32_t *lengths, int B, int N, float4 *pts, float *closest, float4 *tmp,
                            hipStream_t st) {
    if (N > TL * LEAF) return DPM_EUNSUPPORTED;
    const int extra = dpm_knob("DPM_PRICE_FPS_SORT", 0);   // -DDPM_EXPERIMENT builds only: the (idempotent) sort n more times = its price inside the pipelined step
    char *aux = (char *)(((uintptr_t)(tmp + (size_t)B * N) + 255) & ~(uintptr_t)255);
    const int chunks = (N + SC_CHUNK - 1) / SC_CHUNK;
    for (int rep = 0; rep <= extra; ++rep) {
        hipLaunchKernelGGL(str_chunk_kernel<0>, dim3(chunks, B), dim3(SC_T), 0, st, xyz, lengths, N, pts, closest, tmp, aux);
        hipLaunchKernelGGL(str_chunk_kernel<1>, dim3(chunks, B), dim3(SC_T), 0, st, xyz, lengths, N, pts, closest, tmp, aux);
        hipLaunchKernelGGL(str_xoffsets_kernel, dim3(B), dim3(256), 0, st, lengths, N, aux);
        hipLaunchKernelGGL(str_chunk_kernel<2>, dim3(chunks, B), dim3(SC_T), 0, st, xyz, lengths, N, pts, closest, tmp, aux);
        hipLaunchKernelGGL(str_ysort_kernel, dim3(MAXSLAB, B), dim3(SC_T), 0, st, lengths, N, pts, closest, tmp, aux);
    }
    return dpm_launch_status();
}
