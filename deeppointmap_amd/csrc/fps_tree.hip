// Farthest point sampling, TREE algorithm (N up to 65536): one WAVE per frame, no workgroup barrier in the round loop.
// Replaces Sampler.fps / pytorch3d.sample_farthest_points (reference network/encoder/utils.py:210-285) with the
// bit-exact contract of fps.hip: d = (dx*dx + dy*dy) + dz*dz without fused multiply-add, closest = min(d, closest),
// next pick = FIRST index attaining the maximum.
//
// Geometry.  The valid points of a frame are packed Sort-Tile-Recursive style: a counting sort by x (4096 bins)
// cut BY POSITION into <= 32 slabs of whole leaves, then every slab counting-sorted by y (512 bins) and cut by
// position into leaves of 64 points (= one wave-wide float4 load).  Leaves are addressed node-major: a node is a
// 4 x 4 block of (slab, tile) leaves, 64 nodes of 16 leaves.  Compared with Z-ordered grid cells cut into runs,
// leaves are compact rectangles adapted to the point density: 6.5 instead of 10.8 leaves survive the pruning test
// of a round on the benchmark scans (the order inside a bin is arbitrary; no result depends on it).
//
// A round, for the one wave that owns the frame:
//   select : lane n holds node n's box and max(closest) in registers -> DPP wave max -> the winning node's 16 leaf
//            maxima (LDS) -> the leaf's recorded best point (LDS: x, y, z, position).  Ties anywhere fall into an
//            exact slow path that compares ORIGINAL indices (the first-index rule).
//   apply  : box test of the 64 nodes (same fp32 expression as the point distance; every operation is monotone, so
//            box distance <= point distance and the pruning is exact), then the 16 leaf boxes of up to four surviving
//            nodes at once (lane group g tests node g's children), then ALL surviving leaves are requested from
//            memory before the first is evaluated (one round trip per round): distance, conditional store of the
//            new closest value, DPP arg-max, leaf record back to LDS, node maxima patched into their owner lanes.
// ~350 wave-instructions per pick instead of ~1900 (16 waves x 117) for the barrier-synchronised bucket kernel at
// about the same latency per round, so the sampling chain of a whole batch costs the chip's other kernels a sixth of
// the issue slots; `NW` > 1 is reserved for a latency-mode variant.
#include "fps_util.h"

#pragma clang fp contract(off)

namespace {

constexpr int TL = 1024;      // leaves per frame
constexpr int LEAF = 64;      // points per leaf
constexpr int XB = 4096;      // x bins of the first counting sort
constexpr int YB = 512;       // y bins per slab of the second
constexpr int MAXSLAB = 32;
constexpr int SB = 1024;      // threads of the sort workgroup

// per-frame workspace (bytes)
constexpr size_t WS_PTS = (size_t)TL * LEAF * sizeof(float4);   // (x, y, z, closest); closest = -1 in unused slots
constexpr size_t WS_ORIG = (size_t)TL * LEAF * sizeof(int32_t); // original index of every slot
constexpr size_t WS_META = (size_t)TL * (6 + 1 + 4) * sizeof(float);  // leaf boxes (SoA) | leaf max | leaf best (float4)
__host__ __device__ inline size_t ws_frame_bytes(int N) { return WS_PTS + WS_ORIG + WS_META + (size_t)N * sizeof(float4); }

struct FrameWs {
    float4 *pts;
    int32_t *orig;
    float *box;    // [6][TL]
    float *lmax;   // [TL]
    float4 *best;  // [TL]
    float4 *tmp;   // [N] scratch of the sort
};
__device__ __forceinline__ FrameWs frame_ws(char *ws, int b, int N) {
    char *p = ws + (size_t)b * ws_frame_bytes(N);
    FrameWs f;
    f.pts = (float4 *)p;
    f.orig = (int32_t *)(p + WS_PTS);
    f.box = (float *)(p + WS_PTS + WS_ORIG);
    f.lmax = f.box + 6 * TL;
    f.best = (float4 *)(f.lmax + TL);
    f.tmp = (float4 *)(p + WS_PTS + WS_ORIG + WS_META);
    return f;
}

// slab / tile geometry of a frame with `len` valid points
__device__ __forceinline__ void str_shape(int len, int &nsx, int &lps) {
    const int nl = (len + LEAF - 1) / LEAF;
    nsx = 1;
    while (nsx * nsx < nl) ++nsx;      // ceil(sqrt(nl)) <= 32
    nsx = min(nsx, MAXSLAB);
    lps = (nl + nsx - 1) / nsx;        // leaves per slab <= 32
}

// ------------------------------------------------------------------------------------------
// sort: Sort-Tile-Recursive packing of one frame per workgroup
// ------------------------------------------------------------------------------------------
// BUCKETS = true writes the layout of fps.hip's bucket kernel instead (algo 5): points (x, y, z, original index) +
// a separate `closest` array (+inf; -1 in unused slots, whose index field is INT_MAX), TL * LEAF slots per frame.
template <bool BUCKETS>
__global__ __launch_bounds__(SB) void fps_tree_sort_kernel(const float *__restrict__ xyz_all,
                                                           const int32_t *__restrict__ lengths, int N, char *ws,
                                                           float4 *__restrict__ bpts_all, float *__restrict__ bclosest_all,
                                                           float4 *__restrict__ btmp_all) {
    __shared__ int s_hist[MAXSLAB * YB];  // 64 KB; the x pass uses the first XB counters
    __shared__ float s_red[4][SB / 64];
    __shared__ int s_wsum[SB / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *xyz = xyz_all + (size_t)b * N * 3;
    FrameWs f;
    float *bclosest = nullptr;
    if (BUCKETS) {
        f.pts = bpts_all + (size_t)b * TL * LEAF, f.tmp = btmp_all + (size_t)b * N, bclosest = bclosest_all + (size_t)b * TL * LEAF;
        f.orig = nullptr, f.box = f.lmax = nullptr, f.best = nullptr;
    } else {
        f = frame_ws(ws, b, N);
    }
    const int len = min(max(lengths[b], 0), N);
    if (len == 0 && !BUCKETS) return;
    for (int i = t; i < TL * LEAF; i += SB) {
        f.pts[i] = make_float4(0.f, 0.f, 0.f, BUCKETS ? __int_as_float(0x7fffffff) : -1.f);
        if (BUCKETS) bclosest[i] = -1.f;
    }
    if (len == 0) return;

    float lox = __builtin_inff(), loy = __builtin_inff(), hix = -__builtin_inff(), hiy = -__builtin_inff();
    for (int i = t; i < len; i += SB) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1];
        lox = fminf(lox, x), hix = fmaxf(hix, x), loy = fminf(loy, y), hiy = fmaxf(hiy, y);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lox = fminf(lox, __shfl_xor(lox, off, 64));
        loy = fminf(loy, __shfl_xor(loy, off, 64));
        hix = fmaxf(hix, __shfl_xor(hix, off, 64));
        hiy = fmaxf(hiy, __shfl_xor(hiy, off, 64));
    }
    if (lane == 0) s_red[0][w] = lox, s_red[1][w] = loy, s_red[2][w] = hix, s_red[3][w] = hiy;
    for (int c = t; c < XB; c += SB) s_hist[c] = 0;
    __syncthreads();
    for (int k = 0; k < SB / 64; ++k) {
        lox = fminf(lox, s_red[0][k]), loy = fminf(loy, s_red[1][k]);
        hix = fmaxf(hix, s_red[2][k]), hiy = fmaxf(hiy, s_red[3][k]);
    }
    const float sxc = (hix > lox) ? (float)XB / (hix - lox) : 0.f;
    const float syc = (hiy > loy) ? (float)YB / (hiy - loy) : 0.f;
    auto xbin = [&](float x) { return min(max((int)((x - lox) * sxc), 0), XB - 1); };
    auto ybin = [&](float y) { return min(max((int)((y - loy) * syc), 0), YB - 1); };

    // ---- pass 1: counting sort by x bin into tmp (x, y, z, original index)
    for (int i = t; i < len; i += SB) atomicAdd(&s_hist[xbin(xyz[3 * i])], 1);
    __syncthreads();
    {
        const int c0 = s_hist[4 * t], c1 = s_hist[4 * t + 1], c2 = s_hist[4 * t + 2], c3 = s_hist[4 * t + 3];
        const int tsum = c0 + c1 + c2 + c3;
        int inc = tsum;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off, 64);
            if (lane >= off) inc += o;
        }
        if (lane == 63) s_wsum[w] = inc;
        __syncthreads();
        int base = 0;
        for (int k = 0; k < w; ++k) base += s_wsum[k];
        const int excl = base + inc - tsum;
        s_hist[4 * t] = excl;
        s_hist[4 * t + 1] = excl + c0;
        s_hist[4 * t + 2] = excl + c0 + c1;
        s_hist[4 * t + 3] = excl + c0 + c1 + c2;
    }
    __syncthreads();
    for (int i = t; i < len; i += SB) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const int pos = atomicAdd(&s_hist[xbin(x)], 1);
        f.tmp[pos] = make_float4(x, y, z, __int_as_float(i));
    }
    __threadfence_block();
    __syncthreads();

    // ---- pass 2: every slab (a run of lps * 64 consecutive positions) counting-sorted by y bin; a leaf is a run of
    //      64 positions inside its slab
    int nsx, lps;
    str_shape(len, nsx, lps);
    const int slab_pts = lps * LEAF;
    for (int c = t; c < nsx * YB; c += SB) s_hist[c] = 0;
    __syncthreads();
    for (int pos = t; pos < len; pos += SB) atomicAdd(&s_hist[(pos / slab_pts) * YB + ybin(f.tmp[pos].y)], 1);
    __syncthreads();
    {
        // 32 threads per slab, 16 bins each; exclusive scan inside the slab
        const int slab = t >> 5, sub = t & 31;
        int c[16], tsum = 0;
        if (slab < nsx) {
#pragma unroll
            for (int k = 0; k < 16; ++k) c[k] = s_hist[slab * YB + sub * 16 + k], tsum += c[k];
        }
        int inc = tsum;
#pragma unroll
        for (int off = 1; off < 32; off <<= 1) {
            const int o = __shfl_up(inc, off, 32);
            if (sub >= off) inc += o;
        }
        if (slab < nsx) {
            int run = inc - tsum;
#pragma unroll
            for (int k = 0; k < 16; ++k) s_hist[slab * YB + sub * 16 + k] = run, run += c[k];
        }
    }
    __syncthreads();
    const float s0x = xyz[0], s0y = xyz[1], s0z = xyz[2];  // the first pick is index 0 (utils.py:249-250)
    for (int pos = t; pos < len; pos += SB) {
        const float4 p = f.tmp[pos];
        const int slab = pos / slab_pts;
        const int j = atomicAdd(&s_hist[slab * YB + ybin(p.y)], 1);
        const int tile = j >> 6;
        const int leaf = (((slab >> 2) * 8 + (tile >> 2)) << 4) + ((slab & 3) << 2) + (tile & 3);
        const int q = leaf * LEAF + (j & 63);
        if (BUCKETS) {
            f.pts[q] = p;  // (x, y, z, original index)
            bclosest[q] = __builtin_inff();
        } else {
            // closest distance after the first pick: min(+inf, d) = d
            f.pts[q] = make_float4(p.x, p.y, p.z, sqdist(s0x, s0y, s0z, p.x, p.y, p.z));
            f.orig[q] = __float_as_int(p.w);
        }
    }
}

// ------------------------------------------------------------------------------------------
// leaf records: box, max(closest) and the point attaining it (smallest original index among equal maxima)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void fps_tree_leaf_kernel(const int32_t *__restrict__ lengths, int N, char *ws) {
    const int b = blockIdx.x >> 2, w = ((blockIdx.x & 3) << 2) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const FrameWs f = frame_ws(ws, b, N);
    if (min(max(lengths[b], 0), N) == 0) return;
    for (int l = w; l < TL; l += 16) {
        const float4 p = f.pts[l * LEAF + lane];
        const bool ok = p.w >= 0.f;
        const float inf = __builtin_inff();
        const float x0 = -wave_max_dpp(ok ? -p.x : -inf), y0 = -wave_max_dpp(ok ? -p.y : -inf), z0 = -wave_max_dpp(ok ? -p.z : -inf);
        const float x1 = wave_max_dpp(ok ? p.x : -inf), y1 = wave_max_dpp(ok ? p.y : -inf), z1 = wave_max_dpp(ok ? p.z : -inf);
        float vmax = wave_max_ordered(p.w);
        unsigned long long eq = __ballot(p.w == vmax);
        if (vmax >= 0.f && __popcll(eq) > 1) {
            const int o = f.orig[l * LEAF + lane];
            const int imin = wave_min_dpp(p.w == vmax ? o : 0x7fffffff);
            eq = __ballot(p.w == vmax && o == imin);
        }
        const int L = __builtin_ctzll(eq);
        if (lane == L) {
            f.box[0 * TL + l] = x0, f.box[1 * TL + l] = y0, f.box[2 * TL + l] = z0;
            f.box[3 * TL + l] = x1, f.box[4 * TL + l] = y1, f.box[5 * TL + l] = z1;
            f.lmax[l] = vmax;  // -1: empty leaf, never active and never a winner
            f.best[l] = make_float4(p.x, p.y, p.z, __int_as_float(l * LEAF + lane));
        }
    }
}

// ------------------------------------------------------------------------------------------
// the sampling rounds: one wave per frame
// ------------------------------------------------------------------------------------------
#ifdef DPM_FPS_STATS
#define TREE_T(i) do { const long long _n = clock64(); tacc[i] += _n - tprev; tprev = _n; } while (0)
#define TREE_C(i, n) do { cacc[i] += (n); } while (0)
#else
#define TREE_T(i) do { } while (0)
#define TREE_C(i, n) do { } while (0)
#endif

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef FPS_TREE_PRIO
#define FPS_TREE_PRIO 0
#endif
constexpr int OB = 512;  // picks buffered in LDS between flushes
constexpr int G = 12;    // leaves requested from memory before the first one is evaluated (90 % of the rounds need <= 11)

__global__ __launch_bounds__(64) void fps_tree_kernel(const float *__restrict__ xyz_all,
                                                      const int32_t *__restrict__ lengths, int N, int K, char *ws,
                                                      int32_t *__restrict__ idx_all, float *__restrict__ new_xyz_all,
                                                      int32_t *__restrict__ new_len) {
    __shared__ float s_box[6][TL];
    __shared__ float s_lmax[TL];
    __shared__ float4 s_best[TL];
    __shared__ float4 s_pick[OB];
    const int b = blockIdx.x, lane = threadIdx.x;
    const float *xyz = xyz_all + (size_t)b * N * 3;
    const FrameWs f = frame_ws(ws, b, N);
    int32_t *idx = idx_all + (size_t)b * K;
    float *new_xyz = new_xyz_all + (size_t)b * K * 3;
    const int len = min(max(lengths[b], 0), N);
    const int kn = min(len, K);
    if (FPS_TREE_PRIO) __builtin_amdgcn_s_setprio(FPS_TREE_PRIO);  // a latency chain: its few instructions go first on a shared SIMD

    if (lane == 0) {
        idx[0] = 0;  // slot 0 is index 0 even for an empty frame (utils.py:249-250)
        new_xyz[0] = xyz[0], new_xyz[1] = xyz[1], new_xyz[2] = xyz[2];
        new_len[b] = max(kn, 1);
    }
    if (kn > 1) {
        for (int i = lane; i < 6 * TL; i += 64) (&s_box[0][0])[i] = f.box[i];
        for (int i = lane; i < TL; i += 64) s_lmax[i] = f.lmax[i], s_best[i] = f.best[i];
        __syncthreads();  // one wave: just the waitcnt
        // node `lane`: union box and maximum of its 16 leaves (registers, for the whole run)
        float nx0 = __builtin_inff(), ny0 = nx0, nz0 = nx0, nx1 = -nx0, ny1 = -nx0, nz1 = -nx0, nmax = -1.f;
        for (int c = 0; c < 16; ++c) {
            const int l = lane * 16 + c;
            const float m = s_lmax[l];
            if (m >= 0.f) {
                nx0 = fminf(nx0, s_box[0][l]), ny0 = fminf(ny0, s_box[1][l]), nz0 = fminf(nz0, s_box[2][l]);
                nx1 = fmaxf(nx1, s_box[3][l]), ny1 = fmaxf(ny1, s_box[4][l]), nz1 = fmaxf(nz1, s_box[5][l]);
                nmax = fmaxf(nmax, m);
            }
        }
        if (!(nmax >= 0.f)) nx0 = ny0 = nz0 = nx1 = ny1 = nz1 = 0.f;  // empty node: nmax = -1 fails every test

#ifdef DPM_FPS_STATS
        long long tacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, cacc[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tprev = clock64();
#endif
        // the frame's points as a raw buffer: base in four SGPRs, a leaf = one scalar offset, the lane = one VGPR offset
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)f.pts, 0, (int)WS_PTS, 0x00020000);
        const int lane16 = lane * 16;
        for (int r = 1; r < kn; ++r) {
            TREE_T(7);
            // ---------------- select: the point with the largest closest distance (first original index on ties)
            const float g = wave_max_ordered(nmax);
            const unsigned long long neq = __ballot(nmax == g);
            int leaf = 0;
            bool slow = __popcll(neq) != 1;
            if (!slow) {
                const int n = __builtin_ctzll(neq);
                const float lm = s_lmax[n * 16 + (lane & 15)];
                const unsigned ceq = (unsigned)__ballot(lm == g) & 0xFFFFu;
                slow = __popc(ceq) != 1;
                leaf = n * 16 + __builtin_ctz(ceq | 0x10000u);
            }
            TREE_C(3, slow ? 1 : 0);
            if (slow) {  // equal maxima in several leaves: the smallest original index among their recorded points
                int bo = 0x7fffffff, bl = 0;
                for (int c = 0; c < TL / 64; ++c) {
                    const int l = c * 64 + lane;
                    if (s_lmax[l] == g) {
                        const int o = f.orig[__float_as_int(s_best[l].w)];
                        if (o < bo) bo = o, bl = l;
                    }
                }
                const int imin = wave_min_dpp(bo);
                leaf = lane_i(bl, __builtin_ctzll(__ballot(bo == imin)));
            }
            const float4 pick = s_best[leaf];
            const float sx = pick.x, sy = pick.y, sz = pick.z;
            if (lane == 0) s_pick[r & (OB - 1)] = pick;
            if ((r & (OB - 1)) == OB - 1 || r == kn - 1) {  // flush the buffered picks
                for (int q = max(r & ~(OB - 1), 1) + lane; q <= r; q += 64) {
                    const float4 p = s_pick[q & (OB - 1)];
                    idx[q] = f.orig[__float_as_int(p.w)];
                    new_xyz[3 * q] = p.x, new_xyz[3 * q + 1] = p.y, new_xyz[3 * q + 2] = p.z;
                }
            }
            if (r == kn - 1) break;
            TREE_T(0);

            // ---------------- apply: nodes whose box the new point can reach
            unsigned long long mm;
            {
                const float cx = __builtin_amdgcn_fmed3f(sx, nx0, nx1), cy = __builtin_amdgcn_fmed3f(sy, ny0, ny1),
                            cz = __builtin_amdgcn_fmed3f(sz, nz0, nz1);
                mm = __ballot(sqdist(sx, sy, sz, cx, cy, cz) < nmax);
            }
            TREE_T(1);
            while (mm) {
                TREE_C(0, 1);
                // up to four nodes per pass: lane group g = lane >> 4 looks at node n[g]'s 16 leaves
                int n[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    n[k] = mm ? __builtin_ctzll(mm) : -1;
                    if (mm) mm &= mm - 1;
                }
                const int grp = lane >> 4;
                const int node = grp == 0 ? n[0] : grp == 1 ? n[1] : grp == 2 ? n[2] : n[3];
                const int lf = max(node, 0) * 16 + (lane & 15);
                unsigned long long ma;
                {
                    const float cx = __builtin_amdgcn_fmed3f(sx, s_box[0][lf], s_box[3][lf]),
                                cy = __builtin_amdgcn_fmed3f(sy, s_box[1][lf], s_box[4][lf]),
                                cz = __builtin_amdgcn_fmed3f(sz, s_box[2][lf], s_box[5][lf]);
                    ma = __ballot(node >= 0 && sqdist(sx, sy, sz, cx, cy, cz) < s_lmax[lf]);
                }
                bool touched = false;
                TREE_T(6);
                while (ma) {
                    // Every surviving leaf is requested before the first one is looked at: one memory round trip per
                    // round.  Loads and stores share one in-order counter (vmcnt) and behind a branch the compiler has
                    // to assume the worst, so: all requests, ONE wait, then leaf by leaf (no load is outstanding any
                    // more when the first store goes out).
                    const int cnt = __builtin_amdgcn_readfirstlane(min((int)__popcll(ma), G));  // 32-bit scalar: s_cmp, not a 64-bit VALU compare
                    TREE_C(1, 1);
                    TREE_C(2, cnt);
                    int lfs[G];
                    u32x4 p[G];
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        if (j < cnt) {  // wave-uniform
                            lfs[j] = lane_i(lf, __builtin_ctzll(ma));
                            ma &= ma - 1;
                            p[j] = __builtin_amdgcn_raw_buffer_load_b128(rs, lane16, lfs[j] * (LEAF * 16), 0);
                        }
                    }
                    TREE_T(2);
#pragma unroll
                    for (int j = 0; j < G; ++j) {
                        if (j >= cnt) continue;
                        const float px = __uint_as_float(p[j].x), py = __uint_as_float(p[j].y), pz = __uint_as_float(p[j].z),
                                    pc = __uint_as_float(p[j].w);
                        const float d = sqdist(sx, sy, sz, px, py, pz);
                        const bool lt = d < pc;  // unused slots of a ragged leaf hold -1: never
#ifdef DPM_FPS_STATS
                        if (j == 0) { if (__ballot(lt) == 0x123456789ull) tacc[0] += 1; TREE_T(4); }
#endif
                        if (__ballot(lt) == 0) continue;  // a reachable box, but no point of the leaf is closer
                        TREE_C(4, 1);
                        touched = true;
                        if (lt) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(d), rs, lane16 + 12, lfs[j] * (LEAF * 16), 0);
                        const float v = lt ? d : pc;
                        const int q = lfs[j] * LEAF + lane;
                        const float vmax = wave_max_ordered(v);
                        unsigned long long eq = __ballot(v == vmax);
                        if (__popcll(eq) > 1) {  // equal maxima inside the leaf: smallest original index
                            const int o = f.orig[q];
                            const int imin = wave_min_dpp(v == vmax ? o : 0x7fffffff);
                            eq = __ballot(v == vmax && o == imin);
                        }
                        if (lane == (int)__builtin_ctzll(eq)) {
                            s_lmax[lfs[j]] = vmax;
                            s_best[lfs[j]] = make_float4(px, py, pz, __int_as_float(q));
                        }
                    }
                    TREE_T(3);
                }
                if (touched) {  // node maxima of this pass, patched into their owner lanes
                    const float rm = row16_max_f(node >= 0 ? s_lmax[lf] : -1.f);
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (n[k] >= 0)
                            nmax = lane == n[k] ? lane_f(rm, 16 * k) : nmax;
                }
                TREE_T(5);
            }
        }
#ifdef DPM_FPS_STATS
        if (b == 0 && lane == 0)
            for (int i = 0; i < 8; ++i) {
                atomicAdd((unsigned long long *)ws - 32 + i, (unsigned long long)tacc[i]);
                atomicAdd((unsigned long long *)ws - 32 + 8 + i, (unsigned long long)cacc[i]);
            }
#endif
    }
    for (int r = max(kn, 1) + lane; r < K; r += 64) {
        idx[r] = -1;
        new_xyz[3 * r] = 0.f, new_xyz[3 * r + 1] = 0.f, new_xyz[3 * r + 2] = 0.f;
    }
}

}  // namespace

size_t dpm_fps_tree_workspace_bytes(int B, int N) { return (size_t)B * ws_frame_bytes(N) + 512; }

int dpm_fps_tree_launch(const float *xyz, const int32_t *lengths, int B, int N, int K, int32_t *idx, float *new_xyz,
                        int32_t *new_lengths, void *workspace, hipStream_t st) {
    if (N > TL * LEAF) return DPM_EUNSUPPORTED;
    char *ws = (char *)(((uintptr_t)workspace + 255) & ~(uintptr_t)255) + 256;  // 256 B of debug counters in front
#ifdef DPM_FPS_STATS
    (void)hipMemsetAsync(ws - 256, 0, 256, st);
#endif
    hipLaunchKernelGGL(fps_tree_sort_kernel<false>, dim3(B), dim3(SB), 0, st, xyz, lengths, N, ws, (float4 *)nullptr,
                       (float *)nullptr, (float4 *)nullptr);
    hipLaunchKernelGGL(fps_tree_leaf_kernel, dim3(B * 4), dim3(256), 0, st, lengths, N, ws);
    hipLaunchKernelGGL(fps_tree_kernel, dim3(B), dim3(64), 0, st, xyz, lengths, N, K, ws, idx, new_xyz, new_lengths);
    return dpm_launch_status();
}

// algo 5: the Sort-Tile-Recursive packing for fps.hip's bucket kernel (bucket = leaf; a 4 x 4 block of neighbouring
// leaves lands on 16 different waves).  Workspace: TL * LEAF float4 + TL * LEAF float per frame, then N float4 scratch.
size_t dpm_fps_str_bucket_workspace_bytes(int B, int N) {
    return (size_t)B * ((size_t)TL * LEAF * (sizeof(float4) + sizeof(float)) + (size_t)N * sizeof(float4)) + 1024;
}
int dpm_fps_str_bucket_sort(const float *xyz, const int32_t *lengths, int B, int N, float4 *pts, float *closest, float4 *tmp,
                            hipStream_t st) {
    if (N > TL * LEAF) return DPM_EUNSUPPORTED;
    hipLaunchKernelGGL(fps_tree_sort_kernel<true>, dim3(B), dim3(SB), 0, st, xyz, lengths, N, (char *)nullptr, pts, closest, tmp);
    return dpm_launch_status();
}
