// Farthest point sampling for gfx950.  Replaces Sampler.fps / pytorch3d.sample_farthest_points
// (reference network/encoder/utils.py:210-285).
//
// Bit-exact contract (checked against the reference's own outputs in tests): the squared
// distance is (dx*dx + dy*dy) + dz*dz in fp32 with NO fused multiply-add, closest = min(d,
// closest), and the next pick is the FIRST index attaining the maximum.  This translation unit
// is compiled with -ffp-contract=off and additionally pins the pragma below.
//
// One workgroup per frame (the K-1 rounds of a frame are strictly serial); two algorithms:
//
//  * REGISTER (N <= 16384): every thread keeps PPT points (x,y,z,closest) in registers for the
//    whole run; a round = PPT distance updates + one DPP wave arg-max + one LDS exchange (the
//    winner's coordinates travel with it, so nothing is re-read from memory).
//
//  * BUCKET (N up to 65536): points are counting-sorted by a 64x64 xy grid cell in Z-order and
//    cut into buckets of 64 consecutive points (one bucket = one wave-wide load).  Thread t owns
//    bucket t: its bounding box, its current max(closest) and the original index attaining it,
//    all in registers.  A round tests every bucket's box against the newly selected point with
//    the SAME fp32 expression as the point distance; because every fp32 operation involved is
//    monotonic, box distance <= distance to any point in the box, so a bucket whose box
//    distance >= its current max cannot change and is skipped -- the surviving ("active")
//    buckets are the only global-memory traffic of the round.  Results are identical to brute
//    force, bit for bit, including the first-index tie rule (buckets track the smallest
//    original index among their maxima).  The caller's workspace holds the sorted read-only
//    float4 (x, y, z, original index) array and the running `closest` array.
#include "fps_util.h"

#pragma clang fp contract(off)

namespace {

// ------------------------------------------------------------------------------------------
// REGISTER algorithm.  REG=false keeps `closest` in the workspace and re-reads xyz (fallback /
// cross-check path for large N).
// ------------------------------------------------------------------------------------------
template <int BLOCK, int PPT, bool REG>
__global__ __launch_bounds__(BLOCK) void fps_kernel(const float *__restrict__ xyz_all,
                                                    const int32_t *__restrict__ lengths, int N, int K,
                                                    int32_t *__restrict__ idx_all,
                                                    float *__restrict__ new_xyz_all,
                                                    int32_t *__restrict__ new_len, float *__restrict__ cd_ws,
                                                    const int32_t *__restrict__ start) {
    constexpr int NW = BLOCK / 64;
    constexpr int OB = 1024;  // picks buffered in LDS between flushes
    __shared__ float s_v[2][NW], s_x[2][NW], s_y[2][NW], s_z[2][NW];
    __shared__ int s_i[2][NW];
    __shared__ int s_oidx[OB];
    __shared__ float s_oxyz[OB][3];
    const int b = blockIdx.x, t = threadIdx.x;
    const float *xyz = xyz_all + (size_t)b * N * 3;
    int32_t *idx = idx_all + (size_t)b * K;
    float *new_xyz = new_xyz_all + (size_t)b * K * 3;
    float *cdg = REG ? nullptr : cd_ws + (size_t)b * N;
    const int len = min(max(lengths[b], 0), N);
    const int kn = min(len, K);

    float px[REG ? PPT : 1], py[REG ? PPT : 1], pz[REG ? PPT : 1], cd[REG ? PPT : 1];
    if (REG) {
#pragma unroll
        for (int j = 0; j < PPT; ++j) {
            const int i = t + j * BLOCK;
            const bool ok = i < len;
            cd[j] = __builtin_inff();
            px[j] = ok ? xyz[3 * i] : 0.f;
            py[j] = ok ? xyz[3 * i + 1] : 0.f;
            pz[j] = ok ? xyz[3 * i + 2] : 0.f;
        }
    } else {
        for (int i = t; i < len; i += BLOCK) cdg[i] = __builtin_inff();
    }
    // slot 0 is index 0 even for an empty frame (utils.py:249-250) -- or the caller's start index
    // (`random_start_point`, utils.py:248)
    const int s0 = start ? min(max(start[b], 0), max(len - 1, 0)) : 0;
    int cur = s0;
    if (t == 0) {
        idx[0] = s0;
        new_xyz[0] = xyz[3 * s0];
        new_xyz[1] = xyz[3 * s0 + 1];
        new_xyz[2] = xyz[3 * s0 + 2];
        new_len[b] = max(kn, 1);
    }
    float sx = xyz[3 * s0], sy = xyz[3 * s0 + 1], sz = xyz[3 * s0 + 2];
    for (int r = 1; r < kn; ++r) {
        Best best{-1.f, 0x7fffffff};
        float bx = 0.f, by = 0.f, bz = 0.f;  // coordinates of this thread's best point
        if (REG) {
#pragma unroll
            for (int j = 0; j < PPT; ++j) {
                const int i = t + j * BLOCK;
                if (i < len) {
                    const float c = fminf(sqdist(sx, sy, sz, px[j], py[j], pz[j]), cd[j]);
                    cd[j] = c;
                    if (c > best.v) best = Best{c, i}, bx = px[j], by = py[j], bz = pz[j];  // strict >: first maximum
                }
            }
        } else {
            for (int i = t; i < len; i += BLOCK) {
                const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
                const float c = fminf(sqdist(sx, sy, sz, x, y, z), cdg[i]);
                cdg[i] = c;
                if (c > best.v) best = Best{c, i}, bx = x, by = y, bz = z;
            }
        }
        const int lane = t & 63;
        float vmax;
        const int L = wave_argbest(best.v, best.i, vmax);
        const int p = r & 1;
        if (lane == L) s_v[p][t >> 6] = vmax, s_i[p][t >> 6] = best.i, s_x[p][t >> 6] = bx, s_y[p][t >> 6] = by, s_z[p][t >> 6] = bz;
        if (NW > 1) lds_barrier();
        // every lane reads entry (lane % NW); the best of the NW entries is found with a DPP max + ballot
        const int e = lane & (NW - 1);
        const float ev = s_v[p][e];
        const int ei = s_i[p][e];
        const float ex = s_x[p][e], ey = s_y[p][e], ez = s_z[p][e];
        float gv;
        const int gl = wave_argbest(ev, ei, gv);
        cur = lane_i(ei, gl);
        sx = lane_f(ex, gl), sy = lane_f(ey, gl), sz = lane_f(ez, gl);
        if (t == 0) {
            const int o = r & (OB - 1);
            s_oidx[o] = cur, s_oxyz[o][0] = sx, s_oxyz[o][1] = sy, s_oxyz[o][2] = sz;
        }
        if ((r & (OB - 1)) == OB - 1 || r == kn - 1) {  // flush the buffered picks (uniform condition)
            __syncthreads();
            const int r0 = r & ~(OB - 1);
            for (int q = r0 + t; q <= r; q += BLOCK) {
                if (q == 0) continue;
                const int o = q & (OB - 1);
                idx[q] = s_oidx[o];
                new_xyz[3 * q] = s_oxyz[o][0], new_xyz[3 * q + 1] = s_oxyz[o][1], new_xyz[3 * q + 2] = s_oxyz[o][2];
            }
            __syncthreads();
        }
    }
    // padding: -1 / zeros where the frame has fewer than K valid points
    for (int r = max(kn, 1) + t; r < K; r += BLOCK) {
        idx[r] = -1;
        new_xyz[3 * r] = 0.f;
        new_xyz[3 * r + 1] = 0.f;
        new_xyz[3 * r + 2] = 0.f;
    }
}

// ------------------------------------------------------------------------------------------
// BUCKET algorithm, part 1: counting sort by Z-ordered grid cell -> sorted float4 + original ids
// ------------------------------------------------------------------------------------------
constexpr int FB = 1024;        // threads per frame workgroup
constexpr int CELLS = 4096;     // 64 x 64 grid cells
constexpr int MAXBUCKETS = FB;  // one bucket per thread -> N <= 65536

__device__ __forceinline__ unsigned morton2_6(unsigned x, unsigned y) {
    // interleave the low 6 bits of x and y (x in even positions)
    unsigned r = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) r |= ((x >> k) & 1u) << (2 * k) | ((y >> k) & 1u) << (2 * k + 1);
    return r;
}

__global__ __launch_bounds__(FB) void fps_bucket_sort_kernel(const float *__restrict__ xyz_all,
                                                             const int32_t *__restrict__ lengths, int N,
                                                             float4 *__restrict__ pts_all,
                                                             float *__restrict__ closest_all) {
    __shared__ int s_hist[CELLS];
    __shared__ float s_red[4][FB / 64];
    __shared__ int s_wsum[FB / 64];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *xyz = xyz_all + (size_t)b * N * 3;
    float4 *pts = pts_all + (size_t)b * N;
    float *closest = closest_all + (size_t)b * N;
    const int len = min(max(lengths[b], 0), N);

    // xy bounding box of the valid points
    float lox = __builtin_inff(), loy = __builtin_inff(), hix = -__builtin_inff(), hiy = -__builtin_inff();
    for (int i = t; i < len; i += FB) {
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
    for (int c = t; c < CELLS; c += FB) s_hist[c] = 0;
    __syncthreads();
    for (int k = 0; k < FB / 64; ++k) {
        lox = fminf(lox, s_red[0][k]), loy = fminf(loy, s_red[1][k]);
        hix = fmaxf(hix, s_red[2][k]), hiy = fmaxf(hiy, s_red[3][k]);
    }
    const float sxc = (hix > lox) ? 64.f / (hix - lox) : 0.f;
    const float syc = (hiy > loy) ? 64.f / (hiy - loy) : 0.f;
    auto cell_of = [&](float x, float y) -> int {
        const int cx = min(max((int)((x - lox) * sxc), 0), 63);
        const int cy = min(max((int)((y - loy) * syc), 0), 63);
        return (int)morton2_6((unsigned)cx, (unsigned)cy);
    };
    for (int i = t; i < len; i += FB) atomicAdd(&s_hist[cell_of(xyz[3 * i], xyz[3 * i + 1])], 1);
    __syncthreads();
    // exclusive scan of the 4096 counters: 4 per thread -> wave scan -> cross-wave offsets
    int c0 = s_hist[4 * t], c1 = s_hist[4 * t + 1], c2 = s_hist[4 * t + 2], c3 = s_hist[4 * t + 3];
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
    __syncthreads();
    for (int i = t; i < len; i += FB) {
        const float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
        const int pos = atomicAdd(&s_hist[cell_of(x, y)], 1);
        pts[pos] = make_float4(x, y, z, __int_as_float(i));  // read-only from here on: coordinates + original index
        closest[pos] = __builtin_inff();
    }
}

// ------------------------------------------------------------------------------------------
// BUCKET algorithm, part 2: the sampling rounds.
// Bucket b is owned by wave (b % NW), lane (b / NW): spatially adjacent buckets (consecutive b) belong to
// different waves, so the handful of buckets a new point can change are updated by different waves in
// parallel, each wave working only on its OWN buckets -- no work list, no atomics, and one barrier per round
// (the cross-wave arg-max exchange, double-buffered by round parity).
// ------------------------------------------------------------------------------------------
// REGCL (experiment of round 6, -DDPM_FPS_REGCL=1; profiles/r06_fps.md): the running `closest` of the wave's 64 buckets in 64 registers
// per lane (bucket slot l of the wave = register l, wave-uniform index) instead of the workspace array: one global load per touched
// bucket instead of two, no store.
#ifndef DPM_FPS_REGCL
#define DPM_FPS_REGCL 0
#endif
// DPM_FPS_NT (round 6, profiles/r06_step_model.md): the rounds' re-reads of the frame state with the non-temporal cache policy.  A frame's
// state is 1.3 MB, sixteen frames share an XCD's 4 MB L2 while two launches are in flight, and the rounds stream through all of it
// every few tens of microseconds: with the default policy they keep evicting what the feature and registration kernels re-use.
//   bit 0: the point loads (16 B x 64 per touched bucket), bit 1: the `closest` loads and stores.
#ifndef DPM_FPS_NT
#define DPM_FPS_NT 0
#endif
#ifndef DPM_FPS_XBCAST
#define DPM_FPS_XBCAST 0
#endif
typedef float fps_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 fps_load_point(const float4 *p) {
    if (DPM_FPS_NT & 1) {
        const fps_f4 v = __builtin_nontemporal_load((const fps_f4 *)p);
        return make_float4(v.x, v.y, v.z, v.w);
    }
    return *p;
}
__device__ __forceinline__ float fps_load_closest(const float *p) { return (DPM_FPS_NT & 2) ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ void fps_store_closest(float *p, float v) {
    if (DPM_FPS_NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// Timing-only ablations of a round (-DDPM_FPS_ABLATE=bits, WRONG RESULTS; profiles/r06_step_model.md asks which part of a round the
// kernels next to it pay for): bit 0 = no bucket updates (no global loads / stores, no per-bucket arg-max), bit 1 = no exchange (no
// LDS, no barrier: every wave follows its own candidate).  -DDPM_FPS_PACE=n holds every round until n ticks of the 100 MHz clock
// after the previous one, so that all variants keep their workgroups resident equally long.
// Both are kernel arguments of the EXPERIMENT instantiation only (run-time values of -DDPM_EXPERIMENT builds: DPM_FPS_ABLATE, DPM_FPS_PACE in
// the environment); the shipped instantiation has them folded to 0.
template <bool REGCL, bool EXP = false>
__global__ __launch_bounds__(FB) void fps_bucket_kernel(const float *__restrict__ xyz_all,
                                                        const int32_t *__restrict__ lengths, int N, int K,
                                                        const float4 *__restrict__ pts_all,
                                                        float *__restrict__ closest_all,
                                                        int32_t *__restrict__ idx_all,
                                                        float *__restrict__ new_xyz_all,
                                                        int32_t *__restrict__ new_len, int slots,
                                                        const int32_t *__restrict__ start = nullptr, int exp_ablate = 0,
                                                        int exp_pace = 0, int exp_xcds = 0, int exp_xcd_base = 0) {
    const int DPM_FPS_ABLATE = EXP ? exp_ablate : 0, DPM_FPS_PACE = EXP ? exp_pace : 0;
    // (experiment, EXP builds: DPM_FPS_XCDS = n) the launch is 8 / n times wider and only the workgroups that the dispatcher's
    // round-robin puts on XCDs base .. base + n - 1 (linear id mod 8) take a frame; the others leave at once
    int frame = blockIdx.x;
    if (EXP && exp_xcds > 0) {
        const int x = (int)(blockIdx.x & 7) - exp_xcd_base;
        if (x < 0 || x >= exp_xcds) return;
        frame = (int)(blockIdx.x >> 3) * exp_xcds + x;
    }
#ifdef DPM_FPS_PRIO   // wave priority of the sampling waves (A/B builds; round 3 and round 6 measured no effect on the rounds)
    __builtin_amdgcn_s_setprio(DPM_FPS_PRIO);
#endif
    constexpr int NW = FB / 64;
#ifndef DPM_FPS_OB
#define DPM_FPS_OB 2048
#endif
    constexpr int OB = DPM_FPS_OB;  // picks buffered in LDS between flushes to global memory
    // per-wave bests, double-buffered by round parity: [parity][value, index bits, x, y, z][wave] in ONE block, so
    // a wave's five fields are one address plus immediate offsets
    __shared__ float s_ex[2][5][NW];
    __shared__ int s_oidx[OB];
    __shared__ float s_oxyz[OB][3];

    const int b = frame, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *xyz = xyz_all + (size_t)b * N * 3;
    // slots != 0 (Sort-Tile-Recursive packing, algo 5): every frame owns `slots` point slots; unused ones carry the index
    // INT_MAX and closest = -1 (they never move and never win), so the slot count plays the part of the length below
    const size_t fstride = slots ? (size_t)slots : (size_t)N;
    const float4 *pts = pts_all + (size_t)b * fstride;
    float *closest = closest_all + (size_t)b * fstride;
    int32_t *idx = idx_all + (size_t)b * K;
    float *new_xyz = new_xyz_all + (size_t)b * K * 3;
    const int true_len = min(max(lengths[b], 0), N);
    const int kn = min(true_len, K);
    const int len = slots ? (true_len > 0 ? slots : 0) : true_len;
    const int nb = (len + 63) >> 6;
    const int my_bucket = lane * NW + w;
    const bool mine = my_bucket < nb;

    // bounding boxes of this wave's buckets, straight into the owner lane's registers
    float bx0 = 0.f, by0 = 0.f, bz0 = 0.f, bx1 = 0.f, by1 = 0.f, bz1 = 0.f;
    for (int l = 0; l * NW + w < nb; ++l) {
        const int q = (l * NW + w) * 64 + lane;
        float x0 = __builtin_inff(), y0 = x0, z0 = x0, x1 = -x0, y1 = -x0, z1 = -x0;
        if (q < len) {
            const float4 p = pts[q];
            if (__float_as_int(p.w) != 0x7fffffff) x0 = x1 = p.x, y0 = y1 = p.y, z0 = z1 = p.z;
        }
        x0 = -wave_max_dpp(-x0), y0 = -wave_max_dpp(-y0), z0 = -wave_max_dpp(-z0);
        x1 = wave_max_dpp(x1), y1 = wave_max_dpp(y1), z1 = wave_max_dpp(z1);
        if (lane == l) bx0 = x0, by0 = y0, bz0 = z0, bx1 = x1, by1 = y1, bz1 = z1;
    }
    // two 32-wide register vectors (a C array indexed by a run-time value goes to scratch memory; a vector element addressed by a
    // wave-uniform index is one v_movrels / v_movreld through M0)
    typedef float f32x32 __attribute__((ext_vector_type(32)));
    f32x32 cl_lo, cl_hi;
    if (REGCL) {
#pragma unroll
        for (int l = 0; l < 32; ++l) {
            const int qa = (l * NW + w) * 64 + lane, qb = ((l + 32) * NW + w) * 64 + lane;
            cl_lo[l] = qa < len ? closest[qa] : -1.f;   // +inf for points, -1 for the packing's unused slots (written by the sort)
            cl_hi[l] = qb < len ? closest[qb] : -1.f;
        }
    }
    auto cl_get = [&](int l) -> float { return l < 32 ? cl_lo[l & 31] : cl_hi[l & 31]; };   // l is wave-uniform: a scalar branch
    auto cl_put = [&](int l, float v) {
        if (l < 32) cl_lo[l & 31] = v;
        else cl_hi[l & 31] = v;
    };
    const int s0 = start ? min(max(start[b], 0), max(true_len - 1, 0)) : 0;  // `random_start_point` (utils.py:248)
    if (t == 0) {
        idx[0] = s0;  // slot 0 is index 0 even for an empty frame (utils.py:249-250)
        new_xyz[0] = xyz[3 * s0], new_xyz[1] = xyz[3 * s0 + 1], new_xyz[2] = xyz[3 * s0 + 2];
        new_len[b] = max(kn, 1);
    }
    // every closest distance starts at +inf; lanes without a bucket hold the "cannot win" value for good, which also
    // fails every box test below (no distance is < -1), so the round loop needs no `mine` masks
    float bmax = mine ? __builtin_inff() : -1.f;
    int bidx = 0x7fffffff;
    float wx = 0.f, wy = 0.f, wz = 0.f;  // coordinates of this bucket's current best point
    float sx = xyz[3 * s0], sy = xyz[3 * s0 + 1], sz = xyz[3 * s0 + 2];
    float wv = -1.f, wbx = 0.f, wby = 0.f, wbz = 0.f;  // this wave's best over ALL its buckets, valid across rounds
    int wi = 0x7fffffff;
    int wl = 0;  // the bucket slot (lane) that holds the wave's best

#ifdef DPM_FPS_STATS
    long long tacc[6] = {0, 0, 0, 0, 0, 0}, tprev = clock64();
#endif
    const long long pace0 = DPM_FPS_PACE ? (long long)wall_clock64() : 0;
    for (int r = 1; r < kn; ++r) {
        FPS_T(5);
        // ---- which of my wave's buckets can change?  box distance with the point-distance expression:
        //      (s - clamp(s)) reproduces (s - x) monotonically, so box distance <= every point distance
        // clamp(s, lo, hi) as ONE v_med3_f32 (lo <= hi: the median is the clamp; fminf(fmaxf()) costs a NaN
        // canonicalisation per operand on top of the two instructions)
        const float cx = __builtin_amdgcn_fmed3f(sx, bx0, bx1), cy = __builtin_amdgcn_fmed3f(sy, by0, by1),
                    cz = __builtin_amdgcn_fmed3f(sz, bz0, bz1);
        const bool act = sqdist(sx, sy, sz, cx, cy, cz) < bmax;
        unsigned long long m = __ballot(act);
        if (DPM_FPS_ABLATE & 1) m = 0;
        // Bucket maxima only ever fall.  If the bucket that held this wave's best is not touched this round, the
        // wave's best is what it was -- whatever happens to the touched ones -- and nothing needs re-deriving.
        const bool keep = wv >= 0.f && ((m >> wl) & 1ull) == 0;
        FPS_T(0);
#ifdef DPM_FPS_STATS
        if (lane == 0 && m) atomicAdd(&((unsigned long long *)pts_all)[-1 - (b & 0)], (unsigned long long)__popcll(m)), atomicMax(&((int *)pts_all)[-4], __popcll(m));
#endif
        // wave-level best so far, all values wave-uniform: (value, original index, coordinates).  It is carried over
        // from the previous round: a wave none of whose buckets can change this round (almost half of the
        // wave-rounds) has nothing to recompute and goes straight to the exchange.
        bool first = true;
        if (m != 0) do {
            // up to two active buckets per pass; their global loads are issued first ...
            const int l0 = m ? __builtin_ctzll(m) : 0;
            const bool one = m != 0;
            if (one) m &= m - 1;
            const bool two = m != 0;
            const int l1 = two ? __builtin_ctzll(m) : l0;
            if (two) m &= m - 1;
            const int q0 = (l0 * NW + w) * 64 + lane, q1 = (l1 * NW + w) * 64 + lane;
            const bool ok0 = one && q0 < len, ok1 = two && q1 < len;
            // unconditional loads from a clamped slot (a ragged last bucket re-reads the frame's last point; its
            // lanes are masked out of the values below): no exec-mask juggling around the four loads
            const int q0c = min(q0, len - 1), q1c = min(q1, len - 1);
            const float4 p0 = fps_load_point(pts + q0c);
            const float c0 = REGCL ? cl_get(l0) : fps_load_closest(closest + q0c);
            float4 p1;
            float c1;
            if (two) p1 = fps_load_point(pts + q1c), c1 = REGCL ? cl_get(l1) : fps_load_closest(closest + q1c);  // wave-uniform: a scalar branch; an unused load would still
                                                                        // have to be waited for before its registers are reused
            if (first && !keep) {
                // ... and while they are in flight: the best among this wave's UNCHANGED buckets
                wl = wave_argbest(act ? -1.f : bmax, bidx, wv);
                wi = lane_i(bidx, wl), wbx = lane_f(wx, wl), wby = lane_f(wy, wl), wbz = lane_f(wz, wl);
            }
            first = false;
            if (one) {
                const int o0 = ok0 ? __float_as_int(p0.w) : 0x7fffffff;
                const float d = sqdist(sx, sy, sz, p0.x, p0.y, p0.z);
                const bool lt = ok0 && d < c0;
                if (REGCL) cl_put(l0, lt ? d : c0);
                else if (lt) fps_store_closest(closest + q0, d);
                const float v0 = lt ? d : (ok0 ? c0 : -1.f);
                float vmax;
                const int L = wave_argbest(v0, o0, vmax);
                const int bi = lane_i(o0, L);
                const float px = lane_f(p0.x, L), py = lane_f(p0.y, L), pz = lane_f(p0.z, L);
                if (lane == l0) bmax = vmax, bidx = bi, wx = px, wy = py, wz = pz;
                if (!keep && (vmax > wv || (vmax == wv && bi < wi))) wv = vmax, wi = bi, wbx = px, wby = py, wbz = pz, wl = l0;
            }
            if (two) {
                const int o1 = ok1 ? __float_as_int(p1.w) : 0x7fffffff;
                const float d = sqdist(sx, sy, sz, p1.x, p1.y, p1.z);
                const bool lt = ok1 && d < c1;
                if (REGCL) cl_put(l1, lt ? d : c1);
                else if (lt) fps_store_closest(closest + q1, d);
                const float v1 = lt ? d : (ok1 ? c1 : -1.f);
                float vmax;
                const int L = wave_argbest(v1, o1, vmax);
                const int bi = lane_i(o1, L);
                const float px = lane_f(p1.x, L), py = lane_f(p1.y, L), pz = lane_f(p1.z, L);
                if (lane == l1) bmax = vmax, bidx = bi, wx = px, wy = py, wz = pz;
                if (!keep && (vmax > wv || (vmax == wv && bi < wi))) wv = vmax, wi = bi, wbx = px, wby = py, wbz = pz, wl = l1;
            }
        } while (m);
        FPS_T(1);
        const int par = r & 1;
        if (DPM_FPS_ABLATE & 2) {   // timing-only: no exchange, the wave follows its own candidate
            sx = wbx + 1e-3f * r, sy = wby, sz = wbz;
            if (DPM_FPS_PACE) {
                const long long due = pace0 + (long long)r * DPM_FPS_PACE;
                while ((long long)wall_clock64() < due) __builtin_amdgcn_s_sleep(4);
            }
            continue;
        }
        if (lane == 0) {
            float *ex = &s_ex[par][0][w];
            ex[0] = wv, ex[NW] = __int_as_float(wi), ex[2 * NW] = wbx, ex[3 * NW] = wby, ex[4 * NW] = wbz;
        }
        FPS_T(2);
        lds_barrier();
        FPS_T(3);
        // cross-wave arg-max: lane reads entry (lane & 15), 16-lane row reduction, winner's fields by broadcast reads
        const int e = lane & (NW - 1);
#if DPM_FPS_XBCAST
        // (experiment of round 6, profiles/r06_fps.md) ONE LDS round trip: every lane reads candidate (lane & 15) whole, the winner's
        // fields are spread over the row with row_newbcast DPP moves -- the lane is an immediate, hence the 16-way scalar branch
        const float *rec = &s_ex[par][0][e];
        const float ev = rec[0];
        const int ei = __float_as_int(rec[NW]);
        const float ex_ = rec[2 * NW], ey_ = rec[3 * NW], ez_ = rec[4 * NW];
        const float gv = row16_max_f(ev);
        unsigned eqm = (unsigned)(__ballot(ev == gv) & 0xFFFFull);
        if (__popc(eqm) > 1) {  // equal maxima in different waves: smallest original index wins
            const int imin = row16_min_i(ev == gv ? ei : 0x7fffffff);
            eqm = (unsigned)(__ballot(ev == gv && ei == imin) & 0xFFFFull);
        }
        const int gw = __builtin_ctz(eqm);
        int gi;
#define DPM_BC(K)                                                                                                           \
    case K:                                                                                                                 \
        gi = __builtin_amdgcn_update_dpp(ei, ei, 0x150 + K, 0xF, 0xF, false);                                               \
        sx = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ex_), __float_as_int(ex_), 0x150 + K, 0xF, 0xF, false)); \
        sy = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ey_), __float_as_int(ey_), 0x150 + K, 0xF, 0xF, false)); \
        sz = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ez_), __float_as_int(ez_), 0x150 + K, 0xF, 0xF, false)); \
        break;
        switch (gw) {
            DPM_BC(0) DPM_BC(1) DPM_BC(2) DPM_BC(3) DPM_BC(4) DPM_BC(5) DPM_BC(6) DPM_BC(7) DPM_BC(8) DPM_BC(9) DPM_BC(10) DPM_BC(11)
            DPM_BC(12) DPM_BC(13) DPM_BC(14)
            default:
                gi = __builtin_amdgcn_update_dpp(ei, ei, 0x15F, 0xF, 0xF, false);
                sx = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ex_), __float_as_int(ex_), 0x15F, 0xF, 0xF, false));
                sy = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ey_), __float_as_int(ey_), 0x15F, 0xF, 0xF, false));
                sz = __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(ez_), __float_as_int(ez_), 0x15F, 0xF, 0xF, false));
        }
#undef DPM_BC
#else
        const float ev = s_ex[par][0][e];
        const float gv = row16_max_f(ev);
        unsigned eqm = (unsigned)(__ballot(ev == gv) & 0xFFFFull);
        if (__popc(eqm) > 1) {  // equal maxima in different waves: smallest original index wins
            const int ei = __float_as_int(s_ex[par][1][e]);
            const int imin = row16_min_i(ev == gv ? ei : 0x7fffffff);
            eqm = (unsigned)(__ballot(ev == gv && ei == imin) & 0xFFFFull);
        }
        const int gw = __builtin_ctz(eqm);
        const float *gx = &s_ex[par][1][gw];
        const int gi = __float_as_int(gx[0]);
        sx = gx[NW], sy = gx[2 * NW], sz = gx[3 * NW];
#endif
        if (t == 0) {
            const int o = r & (OB - 1);
            s_oidx[o] = gi, s_oxyz[o][0] = sx, s_oxyz[o][1] = sy, s_oxyz[o][2] = sz;
        }
        FPS_T(4);
        if (DPM_FPS_PACE) {
            const long long due = pace0 + (long long)r * DPM_FPS_PACE;
            while ((long long)wall_clock64() < due) __builtin_amdgcn_s_sleep(4);
        }
        if ((r & (OB - 1)) == OB - 1 || r == kn - 1) {  // flush the buffered picks (uniform condition)
            __syncthreads();
            const int r0 = r & ~(OB - 1);
            for (int q = r0 + t; q <= r; q += FB) {
                if (q == 0) continue;  // slot 0 was written up front
                const int o = q & (OB - 1);
                idx[q] = s_oidx[o];
                new_xyz[3 * q] = s_oxyz[o][0], new_xyz[3 * q + 1] = s_oxyz[o][1], new_xyz[3 * q + 2] = s_oxyz[o][2];
            }
            __syncthreads();
        }
    }
    for (int r = max(kn, 1) + t; r < K; r += FB) {
        idx[r] = -1;
        new_xyz[3 * r] = 0.f, new_xyz[3 * r + 1] = 0.f, new_xyz[3 * r + 2] = 0.f;
    }
#ifdef DPM_FPS_STATS
    if (b == 0 && lane == 0)
        for (int i = 0; i < 6; ++i) atomicAdd((unsigned long long *)pts_all - 28 + i, (unsigned long long)tacc[i] / NW);
#endif
}

template <int BLOCK, int PPT, bool REG>
int launch(const float *xyz, const int32_t *lengths, int B, int N, int K, int32_t *idx, float *new_xyz,
           int32_t *new_len, float *ws, hipStream_t st, const int32_t *start = nullptr) {
    hipLaunchKernelGGL((fps_kernel<BLOCK, PPT, REG>), dim3(B), dim3(BLOCK), 0, st, xyz, lengths, N, K, idx,
                       new_xyz, new_len, ws, start);
    return dpm_launch_status();
}

}  // namespace

// fps_tree.hip
size_t dpm_fps_str_bucket_workspace_bytes(int B, int N);
int dpm_fps_str_bucket_sort(const float *xyz, const int32_t *lengths, int B, int N, float4 *pts, float *closest, float4 *tmp,
                            hipStream_t st);

extern "C" size_t dpm_fps_workspace_bytes(int B, int N, int K) {
    (void)K;
    // bucket algorithms: float4 sorted points + closest, per frame (Sort-Tile-Recursive packing: see fps_tree.hip)
    const size_t bucket = (size_t)B * (size_t)N * (sizeof(float4) + sizeof(int32_t)) + 512;
    const size_t strb = N > 16384 && N <= 65536 ? dpm_fps_str_bucket_workspace_bytes(B, N) : 0;
    return bucket > strb ? bucket : strb;
}

// algo: 0 = by size; 1 = plain kernel (any N); 2 = bucket kernel over Z-ordered grid cells (N <= 65 536); 5 = bucket kernel
// over the Sort-Tile-Recursive packing (16 384 < N <= 65 536; the default there).  Rounds 1-2 also carried a speculative
// multi-pick kernel (3 / 6 / 7) and a one-wave-per-frame tree kernel (4): exact, measured slower in every setting for two
// rounds, removed in round 3 (git history: csrc/fps.hip, csrc/fps_tree.hip before "FPS: experimental variants removed").
static int fps_dispatch(const float *xyz, const int32_t *lengths, const int32_t *start, int B, int N, int K, int32_t *idx,
                        float *new_xyz, int32_t *new_lengths, void *workspace, int algo, dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && lengths && idx && new_xyz && new_lengths);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && K >= 1);
    DPM_CHECK_ARG(algo >= 0 && algo <= 7);
    if (algo == 3 || algo == 4 || algo == 6 || algo == 7) return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    // beyond 65 536 points per frame (no shipped pipeline produces such frames: raw scans are voxel-sampled first) the
    // bucket kernels' one-bucket-per-lane layout ends; the plain kernel with `closest` in the workspace takes over
    if (algo == 0) algo = (N > 16384 && N <= 65536) ? 5 : (N > 16384 && N <= 64 * MAXBUCKETS ? 2 : 1);
    if (algo == 5) {  // the bucket kernel over the Sort-Tile-Recursive packing of fps_tree.hip (fewer buckets survive a round)
        DPM_CHECK_ARG(workspace != nullptr);
        if (N <= 16384 || N > 65536) return DPM_EUNSUPPORTED;
        const int slots = 65536;
        uintptr_t p = (((uintptr_t)workspace + 255) & ~(uintptr_t)255) + 256;
        float4 *pts = (float4 *)p;
        float *closest = (float *)(pts + (size_t)B * slots);
        float4 *tmp = (float4 *)(((uintptr_t)(closest + (size_t)B * slots) + 255) & ~(uintptr_t)255);
        const int rc = dpm_fps_str_bucket_sort(xyz, lengths, B, N, pts, closest, tmp, st);
        if (rc != DPM_OK) return rc;
        if (dpm_knob("DPM_ABLATE_FPS_ROUNDS", 0)) return dpm_launch_status();  // -DDPM_EXPERIMENT builds only: the packing without the rounds (scripts/step_model.py)
#ifdef DPM_EXPERIMENT
        {
            static int launches = 0;
            int xcds = dpm_knob("DPM_FPS_XCDS", 0);
            if (xcds > 0 && (B % xcds != 0 || 8 % xcds != 0)) xcds = 0;
            const int base = xcds > 0 ? (launches++ % (8 / xcds)) * xcds : 0;   // consecutive launches take consecutive XCD groups
            hipLaunchKernelGGL((fps_bucket_kernel<DPM_FPS_REGCL != 0, true>), dim3(xcds > 0 ? B / xcds * 8 : B), dim3(FB), 0, st, xyz, lengths, N, K,
                               pts, closest, idx, new_xyz, new_lengths, slots, start, dpm_knob("DPM_FPS_ABLATE", 0), dpm_knob("DPM_FPS_PACE", 0),
                               xcds, base);
        }
#else
        hipLaunchKernelGGL(fps_bucket_kernel<DPM_FPS_REGCL != 0>, dim3(B), dim3(FB), 0, st, xyz, lengths, N, K, pts, closest, idx, new_xyz,
                           new_lengths, slots, start, 0, 0);
#endif
        return dpm_launch_status();
    }
    if (algo >= 2) {
        if (N > 64 * MAXBUCKETS) return DPM_EUNSUPPORTED;
        DPM_CHECK_ARG(workspace != nullptr);
        uintptr_t p = (((uintptr_t)workspace + 255) & ~(uintptr_t)255) + 256;  // 256 B of debug counters in front
        float4 *pts = (float4 *)p;
#ifdef DPM_FPS_STATS
        (void)hipMemsetAsync((void *)(p - 256), 0, 256, st);
#endif
        float *closest = (float *)(pts + (size_t)B * N);
        hipLaunchKernelGGL(fps_bucket_sort_kernel, dim3(B), dim3(FB), 0, st, xyz, lengths, N, pts, closest);
        hipLaunchKernelGGL(fps_bucket_kernel<DPM_FPS_REGCL != 0>, dim3(B), dim3(FB), 0, st, xyz, lengths, N, K, pts, closest, idx,
                           new_xyz, new_lengths, 0, start, 0, 0);
        return dpm_launch_status();
    }
    float *ws = (float *)workspace;
    if (N <= 64) return launch<64, 1, true>(xyz, lengths, B, N, K, idx, new_xyz, new_lengths, ws, st, start);
    if (N <= 256) return launch<256, 1, true>(xyz, lengths, B, N, K, idx, new_xyz, new_lengths, ws, st, start);
    // block shapes measured per level (scripts/fps_small_bench.py): fewer waves = cheaper barrier and second-level
    // reduction, more points per thread = more independent work per round
    if (N <= 1024) return launch<256, 4, true>(xyz, lengths, B, N, K, idx, new_xyz, new_lengths, ws, st, start);
    if (N <= 4096) return launch<512, 8, true>(xyz, lengths, B, N, K, idx, new_xyz, new_lengths, ws, st, start);
    if (N <= 16384) return launch<1024, 16, true>(xyz, lengths, B, N, K, idx, new_xyz, new_lengths, ws, st, start);
    DPM_CHECK_ARG(workspace != nullptr);
    return launch<1024, 1, false>(xyz, lengths, B, N, K, idx, new_xyz, new_lengths, ws, st, start);
}

extern "C" int dpm_fps_ex(const float *xyz, const int32_t *lengths, int B, int N, int K, int32_t *idx,
                          float *new_xyz, int32_t *new_lengths, void *workspace, int algo,
                          dpm_stream_t stream) {
    return fps_dispatch(xyz, lengths, nullptr, B, N, K, idx, new_xyz, new_lengths, workspace, algo, stream);
}

extern "C" int dpm_fps_start(const float *xyz, const int32_t *lengths, const int32_t *start, int B, int N, int K,
                             int32_t *idx, float *new_xyz, int32_t *new_lengths, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(start != nullptr);
    return fps_dispatch(xyz, lengths, start, B, N, K, idx, new_xyz, new_lengths, workspace, 0, stream);
}

extern "C" int dpm_fps(const float *xyz, const int32_t *lengths, int B, int N, int K, int32_t *idx,
                       float *new_xyz, int32_t *new_lengths, void *workspace, dpm_stream_t stream) {
    return dpm_fps_ex(xyz, lengths, B, N, K, idx, new_xyz, new_lengths, workspace, 0, stream);
}
