// kNN-within-radius ("hybrid") neighbour grouping for gfx950.  Replaces Querier.hybrid_query /
// hybrid_query_t3d == pytorch3d.ops.knn_points + radius mask (reference
// network/encoder/utils.py:76-89,113-123).
//
// Semantics: the reference takes the K nearest points and then overwrites every slot whose
// squared distance exceeds r^2 with slot 0 (the nearest).  That equals "the K nearest among the
// points within r, padded with the nearest point", which is what is computed here: only
// candidates with d <= min(r^2, current K-th best) are kept, so the per-centre candidate list stays
// tiny and no (S,N) distance matrix is ever materialised.  Distances reproduce the reference's
// CPU path bit for bit: coordinate_distance (utils.py:288-295) evaluates the EXPANDED form
//   d = ((-2 * dot) + |a|^2) + |b|^2,  dot = fma(az,bz, fma(ay,by, ax*bx)),  |v|^2 = (x*x+y*y)+z*z
// (the K=3 sgemm is an fma chain in k order; verified equal on 3.3e7 pairs in the build container),
// so radius cuts and K-th-neighbour cuts fall exactly where the reference's fall.  This file is
// compiled with -ffp-contract=off; every fused operation below is an explicit fmaf.
// Padded points (index >= lengths[b]) are ignored: the reference moves them to 3*max|coord|,
// which is never nearer than any valid point, so they can only ever be masked out.
//
// Exact ties: when the K-th and (K+1)-th nearest have bit-equal distances, which of them the
// reference keeps is decided by libstdc++'s std::partial_sort (heap-select), which torch.topk's
// CPU kernel uses whenever K*64 <= N.  Such rows are rare (about 3 of 4096 at the first stage,
// where the expanded form quantises distances to ~1e-7) but one flipped neighbour moves the final
// pose by ~1e-4 m, so rows flagged with a boundary tie are re-run through a sequential emulation
// of heap-select (make_heap over the first K points, then pop/replace in index order) and give
// exactly the reference's set.  For K*64 > N the reference uses std::nth_element; ties there
// keep the smaller index (none occur on the shipped shapes).
//
// Two search strategies, same candidate logic:
//  * BRUTE (N < 1024): one wave handles 4 centres and streams all points of the frame.
//  * GRID  (N >= 1024): the frame's points are counting-sorted into a 2-D xy grid whose cell edge
//    exceeds sqrt(r^2 + 2e-5) (the 2e-5 covers the rounding of the expanded form), so every
//    admissible point lies in the 3x3 cells around the centre; cells of one grid row are contiguous
//    in the sorted array, so a centre reads three short ranges.  65 536 x 4096 pair evaluations
//    become ~150 per centre.  A centre with nothing inside the radius (only padded centres) falls
//    back to a full scan for its nearest point.
#include "dpm_common.h"
#include "topk_emulate.h"
#include <type_traits>

namespace {

constexpr int CPW = 4;    // centres per wave (brute force)
constexpr int WPB = 4;    // waves per block
constexpr int CAP = 512;  // candidate slots per centre
constexpr int KMAX = 64;
constexpr int TMPN = 128;  // scratch entries per wave (>= KMAX + 1 and >= 64)
constexpr int GDIM = 128;  // grid cells per axis (upper bound)
constexpr int GRID_MIN_N = 1024;
constexpr int KNN_TODO_MARK = -2;  // first output slot of a row knn_grid_fast_kernel left to knn_grid_kernel

// LDS scratch is handed to the helpers below as address-space-3 pointers: through generic pointers every access
// would compile to a FLAT instruction (the address-space check plus both wait counters) instead of ds_read/ds_write.
using LdsF = __attribute__((address_space(3))) volatile float *;
using LdsI = __attribute__((address_space(3))) volatile int *;

__device__ __forceinline__ void wave_mem_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ float sq3(float x, float y, float z) { return (x * x + y * y) + z * z; }

// the reference's expanded-form squared distance (see header)
__device__ __forceinline__ float exp_dist(float cx, float cy, float cz, float caa, float x, float y, float z,
                                          float bb) {
    return ((-2.f * fmaf(cz, z, fmaf(cy, y, cx * x))) + caa) + bb;
}

// order-preserving map float -> int32 (signed compare): negative floats reversed, positives kept
__device__ __forceinline__ int fkey(float d) {
    const int u = __float_as_int(d);
    return u >= 0 ? u : (u ^ 0x7fffffff);
}

// Select the K smallest of the C > K candidates cd/ci[0..C) (distance, then index).  The result is written
// UNSORTED to od/oi[0..K) (downstream is a max-pool; only slot 0 matters and is fixed up by the caller).
// Returns the K-th smallest distance; *tie is set when the K-th and (K+1)-th distances are bit-equal.
// Bitwise search for the K-th smallest key with ballots instead of K rounds of arg-min extraction.  A wave64
// VALU instruction occupies its SIMD for four cycles, so the search is trimmed to what the data needs:
//   * it starts below the bits all keys share (distances inside one radius share sign and most of the exponent),
//   * it only looks at the 64-candidate chunks that exist,
//   * it stops at the first probe that has exactly K keys below it -- that probe already separates the answer
//     (and proves there is no tie at the boundary).  Typically ~10 probes instead of 32.
__device__ __forceinline__ float select_k(LdsF cd, LdsI ci, int C, int K, LdsF od, LdsI oi, bool *tie) {
    const int lane = lane_id();
    constexpr int R = 4;  // keys cached in registers per lane (C <= 256); the rest is re-read from LDS
    const int nch = min((C + 63) >> 6, R);
    int kr[R];
    int kmin = 0x7fffffff, kmax = -0x7fffffff;
#pragma unroll
    for (int j = 0; j < R; ++j) {
        const int p = lane + 64 * j;
        kr[j] = 0x7fffffff;
        if (j < nch && p < C) {
            kr[j] = fkey(cd[p]);
            kmin = min(kmin, kr[j]), kmax = max(kmax, kr[j]);
        }
    }
    for (int p = lane + 64 * R; p < C; p += 64) {
        const int k = fkey(cd[p]);
        kmin = min(kmin, k), kmax = max(kmax, k);
    }
    kmin = wave_min_dpp(kmin), kmax = -wave_min_dpp(-kmax);
    auto count_lt = [&](int probe) -> int {  // wave-uniform number of candidates with key < probe
        int c = __popcll(__ballot(kr[0] < probe));  // padding keys are INT_MAX: never below a probe
        if (nch > 1) c += __popcll(__ballot(kr[1] < probe));
        if (nch > 2) c += __popcll(__ballot(kr[2] < probe));
        if (nch > 3) c += __popcll(__ballot(kr[3] < probe));
        for (int p = lane + 64 * R; p < ((C + 63) & ~63); p += 64)
            c += __popcll(__ballot(p < C && fkey(cd[p]) < probe));
        return c;
    };
    // t = largest value with count(key < t) < K  ==  the K-th smallest key.  Keys are compared as signed ints,
    // so search in the biased domain (key ^ 0x80000000 is unsigned-monotone), below the common prefix.
    const unsigned umin = (unsigned)kmin ^ 0x80000000u, umax = (unsigned)kmax ^ 0x80000000u;
    const unsigned diff = umin ^ umax;
    int t = kmin, c_lt = 0;
    bool exact = false;  // t has exactly K keys below it (then it is a separator, not a key)
    if (diff != 0) {
        const int top = 31 - __clz(diff);
        unsigned tb = top == 31 ? 0u : (umax >> (top + 1)) << (top + 1);
        for (int bit = top; bit >= 0; --bit) {
            const unsigned cand = tb | (1u << bit);
            const int c = count_lt((int)(cand ^ 0x80000000u));
            if (c == K) {
                exact = true, tb = cand;
                break;
            }
            if (c < K) tb = cand;
        }
        t = (int)(tb ^ 0x80000000u);
        c_lt = exact ? K : count_lt(t);
    }
    const int need = K - c_lt;  // entries equal to t that are taken (none when t is a separator)
    int c_eq = 0;
    if (!exact) {
        if (diff == 0) {
            c_eq = C;
        } else {
#pragma unroll
            for (int j = 0; j < R; ++j) c_eq += __popcll(__ballot(kr[j] == t));
            for (int p = lane + 64 * R; p < ((C + 63) & ~63); p += 64) c_eq += __popcll(__ballot(p < C && fkey(cd[p]) == t));
        }
    }
    *tie = c_eq > need;
    // emit: everything below t, then `need` of the entries equal to t (smallest indices first when tied)
    const unsigned long long ltm = (1ull << lane) - 1ull;
    int base = 0, eq_taken = 0;
    int last_idx = -1;
    float below = -__builtin_inff();  // largest emitted distance below t
    for (int p0 = 0; p0 < C; p0 += 64) {
        const int p = p0 + lane;
        const bool ok = p < C;
        const float d = ok ? cd[p] : 0.f;
        const int k = ok ? fkey(d) : 0x7fffffff;
        const bool lt = ok && k < t;
        const unsigned long long m = __ballot(lt);
        if (lt) od[base + __popcll(m & ltm)] = d, oi[base + __popcll(m & ltm)] = ci[p], below = fmaxf(below, d);
        base += __popcll(m);
        if (!exact && !*tie) {
            const bool eq = ok && k == t;
            const unsigned long long me = __ballot(eq);
            if (eq) od[c_lt + eq_taken + __popcll(me & ltm)] = d, oi[c_lt + eq_taken + __popcll(me & ltm)] = ci[p];
            eq_taken += __popcll(me);
        }
    }
    if (*tie) {  // rare: take the `need` smallest indices among the equal ones
        for (int n = 0; n < need; ++n) {
            int best = 0x7fffffff;
            for (int p = lane; p < C; p += 64) {
                const int i = ci[p];
                if (fkey(cd[p]) == t && i > last_idx && i < best) best = i;
            }
            best = wave_min_dpp(best);
            if (lane == 0) od[c_lt + n] = cd[0] * 0.f, oi[c_lt + n] = best;
            last_idx = best;
        }
    }
    wave_mem_sync();
    // the K-th smallest distance itself: the largest one below a separator, else the value of key t
    float kth;
    if (exact) {
        kth = wave_max_dpp(below);
    } else {
        float v = -__builtin_inff();
        for (int p = lane; p < C; p += 64)
            if (fkey(cd[p]) == t) v = cd[p];
        kth = wave_max_dpp(v);
    }
    if (*tie && lane < need) od[c_lt + lane] = kth;
    wave_mem_sync();
    return kth;
}

// libstdc++ __adjust_heap + __push_heap on (value, index) pairs ordered by value only
__device__ void heap_adjust(LdsF hv, LdsI hi, int hole, int len, float val, int vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (hv[child] < hv[child - 1]) child--;
        hv[hole] = hv[child], hi[hole] = hi[child];
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        hv[hole] = hv[child - 1], hi[hole] = hi[child - 1];
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && hv[parent] < val) {
        hv[hole] = hv[parent], hi[hole] = hi[parent];
        hole = parent, parent = (hole - 1) / 2;
    }
    hv[hole] = val, hi[hole] = vi;
}

// Sequential emulation of std::partial_sort's heap-select over the whole row in ORIGINAL index order.
// hv/hi: K-entry heap in LDS; tv: 64-entry staging.  On return hv/hi hold the K survivors.
__device__ void heap_select_exact(const float *__restrict__ pts, int len, int K, float cx, float cy, float cz,
                                  float caa, LdsF hv, LdsI hi, LdsF tv) {
    const int lane = lane_id();
    auto dist = [&](int i) -> float {
        if (i >= len) return __builtin_inff();
        const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        return exp_dist(cx, cy, cz, caa, x, y, z, sq3(x, y, z));
    };
    if (lane < K) hv[lane] = dist(lane), hi[lane] = lane;
    wave_mem_sync();
    if (lane == 0 && K >= 2) {  // std::__make_heap
        for (int parent = (K - 2) / 2;; --parent) {
            heap_adjust(hv, hi, parent, K, hv[parent], hi[parent]);
            if (parent == 0) break;
        }
    }
    wave_mem_sync();
    for (int base = K; base < len; base += 64) {
        const int i = base + lane;
        const float d = dist(i);
        const float top = hv[0];
        const unsigned long long m = __ballot(d < top);
        if (m) {
            tv[lane] = d;
            wave_mem_sync();
            if (lane == 0) {
                unsigned long long mm = m;
                while (mm) {
                    const int l = __builtin_ctzll(mm);
                    mm &= mm - 1;
                    const float dl = tv[l];
                    if (dl < hv[0]) heap_adjust(hv, hi, 0, K, dl, base + l);  // std::__pop_heap(first, middle, i)
                }
            }
            wave_mem_sync();
        }
    }
}

// torch.topk's OTHER branch, k * 64 > n (every encoder level below 2048 points): std::nth_element over the whole row of
// N distances in index order (padded points are pushed far away by the reference, utils.py:80-81: they compare above
// every real point and equal among themselves, which +inf reproduces).  One wave fills the row, lane 0 replays
// libstdc++ (topk_emulate.h), the K survivors land in hv/hi.  `row` needs N entries (LDS).
__device__ void nth_select_exact(const float *__restrict__ pts, int len, int N, int K, float cx, float cy, float cz,
                                 float caa, VI *row, LdsF hv, LdsI hi) {
    const int lane = lane_id();
    for (int i = lane; i < N; i += 64) {
        float d = __builtin_inff();
        if (i < len) {
            const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
            d = exp_dist(cx, cy, cz, caa, x, y, z, sq3(x, y, z));
        }
        row[i].v = d, row[i].i = i;
    }
    wave_mem_sync();
    LdsU16 sc = (LdsU16)(row + N);  // scratch behind the row: two lists of N 16-bit positions
    vi_nth_element_wave<false>((LdsVI)row, N, K - 1, sc, sc + N);  // the whole wave: a partition round per pass
    wave_mem_sync();
    if (lane < K) hv[lane] = row[lane].v, hi[lane] = row[lane].i;
    wave_mem_sync();
}

// K survivors of the heap-select in hv/hi -> the K output slots (executed by one full wave): slots beyond the
// radius are replaced by the nearest point (utils.py:85-87) and the nearest point itself goes to slot 0.
__device__ __forceinline__ void emit_heap_result(LdsF hv, LdsI hi, int K, float r2, int32_t *__restrict__ out) {
    const int lane = lane_id();
    float mv = (lane < K) ? hv[lane] : __builtin_inff();
    int mi = (lane < K) ? hi[lane] : 0x7fffffff;
    const float myv = mv;
    const int myi = mi;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const float ov = __shfl_xor(mv, off, 64);
        const int oi = __shfl_xor(mi, off, 64);
        if (ov < mv || (ov == mv && oi < mi)) mv = ov, mi = oi;
    }
    int outv = (myv > r2) ? mi : myi;
    // slot 0 must be the nearest point: swap it (in registers) with whoever holds it
    const unsigned long long hm = __ballot(lane < K && myi == mi);
    const int L = hm ? __builtin_ctzll(hm) : 0;
    const int v0 = __shfl(outv, 0, 64);
    if (lane == L) outv = v0;
    if (lane == 0) outv = mi;
    if (lane < K) out[lane] = outv;
}

constexpr int TIE_CAP = 4096;  // queued boundary-tie rows per call (beyond that they are resolved in place)

// Per-centre running state (all fields wave-uniform except gd/gi which are per lane).
struct Ctr {
    float x, y, z, aa, thr, gd;
    int gi, cnt;
    bool tie;
};

__device__ __forceinline__ void ctr_init(Ctr &c, const float *p, float r2) {
    c.x = p[0], c.y = p[1], c.z = p[2];
    c.aa = sq3(c.x, c.y, c.z);
    c.thr = r2, c.gd = __builtin_inff(), c.gi = 0x7fffffff, c.cnt = 0, c.tie = false;
}

// Offer one point per lane (ok = lane holds a valid point with original index i) to centre c.
// Offer one candidate per lane with a precomputed squared distance d (inf for lanes without a point).
__device__ __forceinline__ void offer_d(Ctr &c, float d, int i, int K, LdsF cd, LdsI ci, LdsF td, LdsI ti);

__device__ __forceinline__ void offer(Ctr &c, bool ok, float x, float y, float z, float bb, int i, int K,
                                      LdsF cd, LdsI ci, LdsF td, LdsI ti) {
    offer_d(c, ok ? exp_dist(c.x, c.y, c.z, c.aa, x, y, z, bb) : __builtin_inff(), i, K, cd, ci, td, ti);
}

__device__ __forceinline__ void offer_d(Ctr &c, float d, int i, int K, LdsF cd, LdsI ci, LdsF td, LdsI ti) {
    const int lane = lane_id();
    const bool nearer = (d < c.gd) | ((d == c.gd) & (i < c.gi));  // bitwise on purpose: selects, no exec-mask branches
    c.gd = nearer ? d : c.gd, c.gi = nearer ? i : c.gi;
    const bool in = d <= c.thr;
    const unsigned long long m = __ballot(in);
    if (!m) return;
    if (in) {
        const int pos = c.cnt + __popcll(m & ((1ull << lane) - 1ull));
        cd[pos] = d, ci[pos] = i;
    }
    c.cnt += __popcll(m);
    if (c.cnt > CAP - 64) {  // compact: keep the K smallest, tighten the admission bound
        wave_mem_sync();
        if (c.cnt > K) {
            bool tie = false;
            const float kth = select_k(cd, ci, c.cnt, K, td, ti, &tie);
            c.tie |= tie;
            if (lane < K) cd[lane] = td[lane], ci[lane] = ti[lane];
            wave_mem_sync();
            c.cnt = K;
            c.thr = fminf(c.thr, kth);
        }
    }
}

// Turn the candidate list of centre c into the K output slots.
__device__ __forceinline__ void finish(Ctr &c, const float *__restrict__ pts, int len, int N, int K, float r2,
                                       LdsF cd, LdsI ci, LdsF td, LdsI ti,
                                       int32_t *__restrict__ out, int *tie_count = nullptr,
                                       int32_t *tie_rows = nullptr, int row = 0, VI *nth_row = nullptr) {
    const int lane = lane_id();
    wave_mem_sync();
    // Nearest of all points examined (smallest distance, then smallest index): slot 0.  When anything lies within the
    // radius it is also the nearest candidate -- every offer updated c.gd / c.gi before the admission test, and the
    // nearest point is never above an admission bound that admitted something -- so the candidate list is not
    // searched again.  One DPP min over the order-preserving keys, ties (rare) broken by a second one on the index.
    const int nk = fkey(c.gd);
    const int kbest = wave_min_dpp(nk);
    unsigned long long best = __ballot(nk == kbest);
    if (__popcll(best) > 1) {
        const int imin = wave_min_dpp(nk == kbest ? c.gi : 0x7fffffff);
        best = __ballot(nk == kbest && c.gi == imin);
    }
    int ni = __builtin_amdgcn_readlane(c.gi, __builtin_ctzll(best));
    if (ni == 0x7fffffff) ni = 0;  // empty frame
    const int first = ni;
    const int k = min(K, c.cnt);
    if (c.cnt > K) {
        bool tie = false;
        select_k(cd, ci, c.cnt, K, td, ti, &tie);
        c.tie |= tie;
    } else {
        for (int p = lane; p < c.cnt; p += 64) ti[p] = ci[p];
        wave_mem_sync();
    }
    const bool heap_regime = (long long)K * 64 <= (long long)N;  // torch.topk: partial_sort vs nth_element
    bool exact_here = c.tie && len >= K && (heap_regime || tie_count || nth_row);
    if (exact_here && tie_count) {
        // Grid path: the sequential emulation streams the whole frame and would leave this wave running long after
        // the rest of the kernel has drained, so the row is queued for knn_tie_kernel (one workgroup
        // per row) and gets the plain selection for now.  A full queue falls back to doing it here.
        int slot = 0;
        if (lane == 0) slot = atomicAdd(tie_count, 1);
        slot = __shfl(slot, 0, 64);
        if (slot < TIE_CAP) {
            if (lane == 0) tie_rows[slot] = row;
            exact_here = false;
        }
    }
    if (exact_here && !heap_regime && !nth_row) exact_here = false;  // (queue full and no row scratch: smallest indices)
    if (exact_here) {
        // boundary tie: reproduce the reference's choice exactly (rare, sequential)
        wave_mem_sync();
        if (heap_regime) heap_select_exact(pts, len, K, c.x, c.y, c.z, c.aa, cd, ci, td);
        else nth_select_exact(pts, len, N, K, c.x, c.y, c.z, c.aa, nth_row, cd, ci);
        emit_heap_result(cd, ci, K, r2, out);
    } else {
        // unsorted selection: put the nearest point into slot 0 by swapping it with whatever sits there
        int outv = (lane < k) ? ti[lane] : first;
        const unsigned long long hm = __ballot(lane < k && outv == first);
        const int L = hm ? __builtin_ctzll(hm) : 0;
        const int v0 = __shfl(outv, 0, 64);
        if (lane == L) outv = v0;
        if (lane == 0) outv = first;
        if (lane < K) out[lane] = outv;
    }
    wave_mem_sync();
}

// ---------------------------------------------------------------------------------------------
// BRUTE: one wave = CPW centres, streams every point
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WPB * 64) void knn_hybrid_kernel(const float *__restrict__ points_all,
                                                              const int32_t *__restrict__ lengths,
                                                              const float *__restrict__ centers_all, int N,
                                                              int S, int K, float r2,
                                                              int32_t *__restrict__ idx_all,
                                                              const int32_t *__restrict__ reuse_idx,
                                                              const int32_t *__restrict__ center_src, int nth_rows) {
    extern __shared__ VI s_nth[];  // nth_rows != 0: N entries per wave for the nth_element replay of tied rows
    __shared__ float s_d[WPB][CPW][CAP];
    __shared__ int s_i[WPB][CPW][CAP];
    __shared__ float s_td[WPB][TMPN];
    __shared__ int s_ti[WPB][TMPN];
    const int b = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s0 = (blockIdx.x * WPB + w) * CPW;
    if (s0 >= S) return;  // whole wave exits together; no block-wide barrier is used below
    if (reuse_idx) {
        // centres that ARE points of this frame (center_src >= 0) were already answered by the self-query over
        // all points with the same radius and K: copy those rows; only padded centres are computed
        bool all = true;
        for (int j = 0; j < CPW && s0 + j < S; ++j) all &= center_src[(size_t)b * S + s0 + j] >= 0;
        if (all) {
            for (int j = 0; j < CPW && s0 + j < S; ++j) {
                const int src = center_src[(size_t)b * S + s0 + j];
                if (lane < K) idx_all[((size_t)b * S + s0 + j) * K + lane] = reuse_idx[((size_t)b * N + src) * K + lane];
            }
            return;
        }
    }
    const float *pts = points_all + (size_t)b * N * 3;
    const float *ctr = centers_all + (size_t)b * S * 3;
    const int len = min(max(lengths[b], 0), N);
    Ctr c[CPW];
#pragma unroll
    for (int j = 0; j < CPW; ++j) ctr_init(c[j], ctr + 3 * (size_t)min(s0 + j, S - 1), r2);
    for (int base = 0; base < len; base += 64) {
        const int i = base + lane;
        const bool ok = i < len;
        float x = 0.f, y = 0.f, z = 0.f;
        if (ok) x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
        const float bb = sq3(x, y, z);
#pragma unroll
        for (int j = 0; j < CPW; ++j) offer(c[j], ok, x, y, z, bb, i, K, (LdsF)s_d[w][j], (LdsI)s_i[w][j], (LdsF)s_td[w], (LdsI)s_ti[w]);
    }
#pragma unroll
    for (int j = 0; j < CPW; ++j) {
        if (s0 + j >= S) break;
        finish(c[j], pts, len, N, K, r2, (LdsF)s_d[w][j], (LdsI)s_i[w][j], (LdsF)s_td[w], (LdsI)s_ti[w],
               idx_all + ((size_t)b * S + (s0 + j)) * K, nullptr, nullptr, 0,
               nth_rows ? (VI *)((char *)s_nth + (size_t)w * N * (sizeof(VI) + 4)) : nullptr);
    }
}

// ---------------------------------------------------------------------------------------------
// GRID build: counting sort of one frame into row-major xy cells (one 1024-thread block per frame)
// ---------------------------------------------------------------------------------------------
struct KnnGrid {  // per frame header in the workspace
    float lox, loy, inv_cs;
    int g;       // cells per axis
    float err2;  // what the expanded-form distance may undershoot the true squared distance by (frame-dependent)
    int H;       // the (2H+1)^2 cells around a centre's cell contain every point whose computed distance is <= r^2
    int h;       // half-width of the INNER block knn_grid_fast_kernel tries first (h = H: the block covers the radius)
    int pad;
};

// Cell edge of the search grid.  With edge = covering edge (> the radius) the 3x3 cells around a centre hold everything
// within the radius -- and, on a dense first-stage frame, 15 times K points.  The K nearest sit much closer: within
// rho_t, the radius that holds about GRID_RHO_POINTS points of a surface of the frame's mean density (4 K for K = 32: a scan
// is not a surface everywhere, its points spread along z as well, and the centres -- farthest-point samples -- are its
// outliers: on the benchmark scans the 32nd neighbour of a first-stage centre is 0.033 away in the median, 0.045 at the
// 90th percentile, with 0.05 the radius).  The fast search therefore looks
// at an inner block of (2h+1)^2 finer cells whose border is at least rho_t away from the centre, and proves afterwards
// that the K-th distance found stays inside it.  Three layouts, the one with the fewest points per inner block wins:
//   m = 1: edge = covering edge,            inner = full = 3x3 (sparse levels: every lower level of the encoder)
//   m = 2: edge = max(rho_t, cover / 2),     inner 3x3, full 5x5
//   m = 3: edge = max(rho_t / 2, cover / 3), inner 5x5, full 7x7
constexpr float GRID_RHO_POINTS = 128.f;
constexpr float GRID_PI = 3.14159265f;

__global__ __launch_bounds__(1024) void knn_grid_build_kernel(const float *__restrict__ points_all,
                                                              const int32_t *__restrict__ lengths, int N,
                                                              float cs_min, float r2_margin, KnnGrid *__restrict__ hdr_all,
                                                              int *__restrict__ start_all,
                                                              float4 *__restrict__ sorted_all,
                                                              int *__restrict__ tie_count) {
    if (blockIdx.x == 0 && threadIdx.x == 0) tie_count[0] = 0, tie_count[1] = 0;  // tie queue, todo list
    __shared__ int s_hist[GDIM * GDIM];
    __shared__ float s_red[4][16];
    __shared__ float s_m2[16];
    __shared__ int s_wsum[16];
    const int b = blockIdx.x, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const float *pts = points_all + (size_t)b * N * 3;
    const int len = min(max(lengths[b], 0), N);
    float lox = __builtin_inff(), loy = lox, hix = -lox, hiy = -lox, m2 = 0.f;
    // every pass over the frame fetches UB points per thread before it uses the first (unconditional loads from clamped
    // positions): a pass is 64 dependent round trips per thread otherwise -- the kernel is a latency chain on 64 CUs
    constexpr int UB = 8;
    auto for_points = [&](auto &&fn) {
        for (int i0 = t; i0 < len; i0 += 1024 * UB) {
            float xs[UB], ys[UB], zs[UB];
#pragma unroll
            for (int u = 0; u < UB; ++u) {
                const int i = min(i0 + u * 1024, len - 1);
                xs[u] = pts[3 * i], ys[u] = pts[3 * i + 1], zs[u] = pts[3 * i + 2];
            }
#pragma unroll
            for (int u = 0; u < UB; ++u)
                if (i0 + u * 1024 < len) fn(i0 + u * 1024, xs[u], ys[u], zs[u]);
        }
    };
    for_points([&](int, float x, float y, float z) {
        lox = fminf(lox, x), hix = fmaxf(hix, x), loy = fminf(loy, y), hiy = fmaxf(hiy, y);
        m2 = fmaxf(m2, sq3(x, y, z));
    });
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        lox = fminf(lox, __shfl_xor(lox, off, 64)), loy = fminf(loy, __shfl_xor(loy, off, 64));
        hix = fmaxf(hix, __shfl_xor(hix, off, 64)), hiy = fmaxf(hiy, __shfl_xor(hiy, off, 64));
        m2 = fmaxf(m2, __shfl_xor(m2, off, 64));
    }
    if (lane == 0) s_red[0][w] = lox, s_red[1][w] = loy, s_red[2][w] = hix, s_red[3][w] = hiy, s_m2[w] = m2;
    for (int c = t; c < GDIM * GDIM; c += 1024) s_hist[c] = 0;
    __syncthreads();
    for (int k = 0; k < 16; ++k) {
        lox = fminf(lox, s_red[0][k]), loy = fminf(loy, s_red[1][k]);
        hix = fmaxf(hix, s_red[2][k]), hiy = fmaxf(hiy, s_red[3][k]);
        m2 = fmaxf(m2, s_m2[k]);
    }
    if (len == 0) lox = loy = hix = hiy = 0.f;
    const float ext = fmaxf(fmaxf(hix - lox, hiy - loy), 1e-6f);
    // The search compares the reference's EXPANDED-form distance with r^2, and that form's rounding error grows with
    // the squared magnitude of the coordinates (it is a difference of numbers of size |a|^2 + |b|^2): a point whose
    // computed distance is <= r^2 can truly be sqrt(r^2 + err) away.  The cell edge covers it: 2e-5 for coordinates
    // normalised to the unit ball (what the encoder feeds), scaled up for clouds that are not.
    const float err2 = r2_margin > 0.f ? 2e-5f * fmaxf(1.f, m2) : 0.f;
    if (r2_margin > 0.f) cs_min = fmaxf(cs_min, sqrtf(r2_margin + err2) * 1.002f);
    float cs = fmaxf(cs_min, ext / (float)(GDIM - 1));  // the covering edge: 3x3 cells hold everything within the radius
    int H = 1, hin = 1;
    if (r2_margin > 0.f && len > 0) {  // hybrid query: density-adapted edge (see GRID_RHO_POINTS)
        const float area = fmaxf(hix - lox, 1e-6f) * fmaxf(hiy - loy, 1e-6f);
        // bounding-box area: a disc-shaped scan fills pi/4 of it
        const float rho_t = sqrtf(GRID_RHO_POINTS * (area * (GRID_PI / 4.f)) / (GRID_PI * (float)len));
        const float floor_cs = ext / (float)(GDIM - 1);
        const float cs2 = fmaxf(fmaxf(rho_t, cs * 0.5005f), floor_cs), cs3 = fmaxf(fmaxf(rho_t * 0.5f, cs * 0.3337f), floor_cs);
        float best = 9.f * cs * cs;
        const float cover = cs;
        if (cs2 < cover && 9.f * cs2 * cs2 < best) best = 9.f * cs2 * cs2, cs = cs2, H = 2, hin = 1;
        if (cs3 < cover * 0.5f && 25.f * cs3 * cs3 < best) best = 25.f * cs3 * cs3, cs = cs3, H = 3, hin = 2;
    }
    const float inv_cs = 1.0f / cs;
    const int g = min(GDIM, (int)(ext * inv_cs) + 1);
    auto cell = [&](float x, float y) -> int {
        const int cx = min(max((int)floorf((x - lox) * inv_cs), 0), g - 1);
        const int cy = min(max((int)floorf((y - loy) * inv_cs), 0), g - 1);
        return cy * g + cx;
    };
    for_points([&](int, float x, float y, float) { atomicAdd(&s_hist[cell(x, y)], 1); });
    __syncthreads();
    // exclusive scan over the counters, 16 per thread
    constexpr int PER = GDIM * GDIM / 1024;
    int local[PER], tsum = 0;
#pragma unroll
    for (int q = 0; q < PER; ++q) local[q] = s_hist[t * PER + q], tsum += local[q];
    int inc = tsum;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(inc, off, 64);
        if (lane >= off) inc += o;
    }
    if (lane == 63) s_wsum[w] = inc;
    __syncthreads();
    int run = inc - tsum;
    for (int k = 0; k < w; ++k) run += s_wsum[k];
    int *start = start_all + (size_t)b * (GDIM * GDIM + 1);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        s_hist[t * PER + q] = run;
        start[t * PER + q] = run;
        run += local[q];
    }
    if (t == 1023) start[GDIM * GDIM] = run;
    if (t == 0) hdr_all[b] = KnnGrid{lox, loy, inv_cs, g, err2, H, hin, 0};
    __syncthreads();
    float4 *sorted = sorted_all + (size_t)b * N;
    for_points([&](int i, float x, float y, float z) {
        const int pos = atomicAdd(&s_hist[cell(x, y)], 1);
        sorted[pos] = make_float4(x, y, z, __int_as_float(i));
    });
}

// ---------------------------------------------------------------------------------------------
// GRID search: one wave per centre
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WPB * 64) void knn_grid_kernel(const float *__restrict__ points_all,
                                                            const int32_t *__restrict__ lengths,
                                                            const float *__restrict__ centers_all, int N, int S,
                                                            int K, float r2, const KnnGrid *__restrict__ hdr_all,
                                                            const int *__restrict__ start_all,
                                                            const float4 *__restrict__ sorted_all,
                                                            int32_t *__restrict__ idx_all,
                                                            const int32_t *__restrict__ reuse_idx,
                                                            const int32_t *__restrict__ center_src,
                                                            int *__restrict__ tie_count,
                                                            int32_t *__restrict__ tie_rows,
                                                            const int *__restrict__ todo_count,
                                                            const int32_t *__restrict__ todo_rows, int todo_cap, int n_rows) {
    __shared__ float s_d[WPB][CAP];
    __shared__ int s_i[WPB][CAP];
    __shared__ float s_td[WPB][TMPN];
    __shared__ int s_ti[WPB][TMPN];
    // the wave index is wave-uniform, but the compiler only knows that through readfirstlane: with it the centre,
    // its cell, the cell ranges and the chunk loop live in scalar registers (scalar loads, scalar loop control)
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    // Two ways to be told which rows (row = frame * S + centre) to search:
    //  * todo_count == NULL: every row, one wave each (gridDim.x * WPB >= n_rows), frames kept on one XCD;
    //  * otherwise the rows knn_grid_fast_kernel could not finish: the first min(*todo_count, todo_cap) entries of
    //    todo_rows, and when more rows failed than the list holds, every row whose first output slot carries the
    //    fast kernel's mark (KNN_TODO_MARK).  The grid is fixed; waves stride over the work.
    const int n_list = todo_count ? *todo_count : 0;
    const bool by_mark = todo_count && n_list > todo_cap;
    const int n_work = !todo_count || by_mark ? n_rows : n_list;
    const int n_waves = gridDim.x * WPB;
    for (int it = (todo_count ? blockIdx.x : (int)xcd_chunked_id(blockIdx.x, gridDim.x)) * WPB + w; it < n_work; it += n_waves) {
    int row = it;
    if (todo_count && !by_mark) row = todo_rows[it];
    if (by_mark && idx_all[(size_t)row * K] != KNN_TODO_MARK) continue;
    const int b = row / S, s = row - b * S;
    if (reuse_idx) {
        const int src = center_src[(size_t)b * S + s];
        if (src >= 0) {  // this centre is point `src` of the frame: its row of the self-query is the answer
            if (lane < K) idx_all[((size_t)b * S + s) * K + lane] = reuse_idx[((size_t)b * N + src) * K + lane];
            continue;
        }
    }
    const float *pts = points_all + (size_t)b * N * 3;
    const int len = min(max(lengths[b], 0), N);
    const KnnGrid G = hdr_all[b];
    const int *start = start_all + (size_t)b * (GDIM * GDIM + 1);
    const float4 *sorted = sorted_all + (size_t)b * N;
    Ctr c;
    ctr_init(c, centers_all + ((size_t)b * S + s) * 3, r2);
    // cell coordinates clamped in float first: a centre far outside the grid must not overflow the int conversion
    const int cx = (int)fminf(fmaxf(floorf((c.x - G.lox) * G.inv_cs), -8.f), (float)(GDIM + 8));
    const int cy = (int)fminf(fmaxf(floorf((c.y - G.loy) * G.inv_cs), -8.f), (float)(GDIM + 8));
    const int x0 = max(cx - G.H, 0), x1 = min(cx + G.H, G.g - 1);
    if (x0 <= x1) {
        for (int yy = max(cy - G.H, 0); yy <= min(cy + G.H, G.g - 1); ++yy) {
            const int lo = start[yy * G.g + x0], hi = start[yy * G.g + x1 + 1];
            // unconditional loads from a clamped slot, the next chunk requested before the current one is offered
            float4 p = lo < hi ? sorted[min(lo + lane, hi - 1)] : make_float4(0.f, 0.f, 0.f, 0.f);
            for (int base = lo; base < hi; base += 64) {
                const int q = base + lane;
                const bool ok = q < hi;
                const float4 cur = p;
                if (base + 64 < hi) p = sorted[min(q + 64, hi - 1)];
                offer(c, ok, cur.x, cur.y, cur.z, sq3(cur.x, cur.y, cur.z), ok ? __float_as_int(cur.w) : 0x7fffffff, K,
                      (LdsF)s_d[w], (LdsI)s_i[w], (LdsF)s_td[w], (LdsI)s_ti[w]);
            }
        }
    }
    if (c.cnt == 0) {
        // nothing within the radius (a padded centre): the answer is the nearest point overall
        c.gd = __builtin_inff(), c.gi = 0x7fffffff;
        for (int base = 0; base < len; base += 64) {
            const int i = base + lane;
            if (i < len) {
                const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
                const float d = exp_dist(c.x, c.y, c.z, c.aa, x, y, z, sq3(x, y, z));
                if (d < c.gd) c.gd = d, c.gi = i;
            }
        }
    }
    finish(c, pts, len, N, K, r2, (LdsF)s_d[w], (LdsI)s_i[w], (LdsF)s_td[w], (LdsI)s_ti[w], idx_all + ((size_t)b * S + s) * K,
           tie_count, tie_rows, b * S + s);
    }
}

// ---------------------------------------------------------------------------------------------
// GRID search, fast path: a QUARTER wave per centre, everything in registers.
//
// knn_grid_kernel above spends a whole wave on a centre whose answer hangs on ~120 candidates: its time is the fixed
// instruction count of one wave (candidate lists in LDS, a ballot bit search, several wave reductions), ~1 % of the
// VALU peak.  Here the 16 lanes of one DPP row own a centre (four centres per wave share every instruction), each lane
// keeps up to CPL candidates as (key, index) in registers, all reductions are four-step DPP row operations and nothing
// touches LDS:
//   * block: the centre's (2H+1)^2 cells when they hold at most QCAP points (then the block contains every point
//     within the radius and the answer is exact as in knn_grid_kernel); otherwise, H = 2, only the inner 3x3 cells,
//     accepted when the K-th distance found there is provably smaller than the distance to the block's border
//     (rho: every point outside the block is at least rho away; the expanded form may undershoot a true squared
//     distance by err2, both from the grid header).  The rows of the block are contiguous ranges of the sorted
//     array; a lane walks the concatenation of the (up to five) ranges with stride 16.
//   * selection: the K-th smallest key by probing count(key < t): three interpolation probes (the number of points
//     within squared distance d grows linearly in d on a surface), then bisection on the integer keys; a probe
//     with exactly K keys below it separates the answer and ends the search.  No separator exists iff the K-th and
//     (K+1)-th keys are equal: the row goes to the tie replay kernel, as in knn_grid_kernel.
//   * output: unsorted, nearest point in slot 0, slots beyond the radius filled with the nearest (utils.py:85-87).
// What it cannot finish -- block too populous for the registers, border test failed, nothing within the radius, centre
// outside the grid, tie queue full -- gets KNN_TODO_MARK in its first slot and a place in the todo list, which
// knn_grid_kernel works off afterwards.
// ---------------------------------------------------------------------------------------------
constexpr int QL = 16;           // lanes per centre (one DPP row)
constexpr int CPL = 20;          // candidate slots per lane, in groups of QB
constexpr int QCAP = QL * CPL;   // candidates per centre
constexpr int QROWS = 5;         // grid rows of a block (inner: h <= 2; full: only while 2H+1 <= 5)
constexpr int QB = 5;            // slots per group: loaded together, skipped together

template <int CTRL>
__device__ __forceinline__ int dpp_row(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xF, 0xF, false); }
template <int CTRL>
__device__ __forceinline__ int dpp_row_zero(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }
// reductions over the 16 lanes of a DPP row; every lane of the row gets the result
__device__ __forceinline__ int row_sum(int v) {
    v += dpp_row<0xB1>(v), v += dpp_row<0x4E>(v), v += dpp_row<0x141>(v), v += dpp_row<0x140>(v);
    return v;
}
__device__ __forceinline__ int row_min(int v) {
    v = min(v, dpp_row<0xB1>(v)), v = min(v, dpp_row<0x4E>(v)), v = min(v, dpp_row<0x141>(v)), v = min(v, dpp_row<0x140>(v));
    return v;
}
// inclusive prefix sum along the row (row_shr:1,2,4,8 with zero fill)
__device__ __forceinline__ int row_scan(int v) {
    v += dpp_row_zero<0x111>(v), v += dpp_row_zero<0x112>(v), v += dpp_row_zero<0x114>(v), v += dpp_row_zero<0x118>(v);
    return v;
}
// inverse of fkey
__device__ __forceinline__ float fkey_inv(int k) { return __int_as_float(k >= 0 ? k : (k ^ 0x7fffffff)); }

__global__ __launch_bounds__(WPB * 64) void knn_grid_fast_kernel(const int32_t *__restrict__ lengths,
                                                                 const float *__restrict__ centers_all, int N, int S, int K,
                                                                 float r2, const KnnGrid *__restrict__ hdr_all,
                                                                 const int *__restrict__ start_all,
                                                                 const float4 *__restrict__ sorted_all,
                                                                 int32_t *__restrict__ idx_all,
                                                                 const int32_t *__restrict__ reuse_idx,
                                                                 const int32_t *__restrict__ center_src,
                                                                 int *__restrict__ tie_count, int32_t *__restrict__ tie_rows,
                                                                 int *__restrict__ todo_count, int32_t *__restrict__ todo_rows,
                                                                 int todo_cap) {
    constexpr int CPB = WPB * 64 / QL;  // centres per block
    valu_bound_priority();
    const unsigned bid = xcd_chunked_id(blockIdx.y * gridDim.x + blockIdx.x, gridDim.x * gridDim.y);
    const int b = bid / gridDim.x;
    const int ql = threadIdx.x & (QL - 1);
    const int s = (bid % gridDim.x) * CPB + (threadIdx.x >> 4);
    bool live = s < S;                       // row-uniform, like every flag below
    const int sc = min(s, S - 1);
    const size_t row = (size_t)b * S + sc;
    int32_t *out = idx_all + row * K;
    if (reuse_idx && live) {
        const int src = center_src[row];
        if (src >= 0) {  // this centre is point `src` of the frame: its row of the self-query is the answer
            for (int k = ql; k < K; k += QL) out[k] = reuse_idx[((size_t)b * N + src) * K + k];
            live = false;
        }
    }
    if (!__ballot(live)) return;
    const KnnGrid G = hdr_all[b];
    const int *start = start_all + (size_t)b * (GDIM * GDIM + 1);
    const float4 *sorted = sorted_all + (size_t)b * N;
    const float *cp = centers_all + row * 3;
    const float cx_ = cp[0], cy_ = cp[1], cz_ = cp[2], caa = sq3(cx_, cy_, cz_);
    const float fx = (cx_ - G.lox) * G.inv_cs, fy = (cy_ - G.loy) * G.inv_cs;
    const float flx = floorf(fx), fly = floorf(fy);
    bool todo = false;  // leave this row to knn_grid_kernel
    // a centre outside the grid (a padded or free-standing one): not for this path (NaN coordinates fail the test too)
    if (!(flx >= 0.f && flx < (float)G.g && fly >= 0.f && fly < (float)G.g)) todo = live, live = false;
    const int cx = live ? (int)flx : 0, cy = live ? (int)fly : 0;
    // ---- the block's rows: [lo_r, lo_r + n_r) of the sorted array, row r = grid row cy + r - 2.  First the inner block
    //      (h = H: it is the whole block); the full block's offsets are fetched only when a centre of this wave is sparse
    //      enough to use it
    int lo[QROWS], n[QROWS];
    int tot = 0;
    {
        const int xa = max(cx - G.h, 0), xb = min(cx + G.h, G.g - 1) + 1;
#pragma unroll
        for (int r = 0; r < QROWS; ++r) {
            lo[r] = 0, n[r] = 0;
            if (abs(r - 2) > G.h) continue;  // uniform
            const int yy = cy + r - 2;
            const int yc = min(max(yy, 0), G.g - 1);
            const int a = start[yc * G.g + xa], e = start[yc * G.g + xb];
            lo[r] = a, n[r] = (live && yy >= 0 && yy < G.g) ? e - a : 0, tot += n[r];
        }
    }
    bool inner = G.H > G.h;  // the block does not cover the radius: its answer needs the border test
    const int full_cells = (2 * G.H + 1) * (2 * G.H + 1), inner_cells = (2 * G.h + 1) * (2 * G.h + 1);
    const bool want_full = live && inner && 2 * G.H + 1 <= QROWS && tot * full_cells <= QCAP * inner_cells;
    if (__ballot(want_full)) {
        int flo[QROWS], fn[QROWS], ftot = 0;
        const int xa = max(cx - G.H, 0), xb = min(cx + G.H, G.g - 1) + 1;
#pragma unroll
        for (int r = 0; r < QROWS; ++r) {
            flo[r] = 0, fn[r] = 0;
            if (abs(r - 2) > G.H) continue;  // uniform
            const int yy = cy + r - 2;
            const int yc = min(max(yy, 0), G.g - 1);
            const int a = start[yc * G.g + xa], e = start[yc * G.g + xb];
            flo[r] = a, fn[r] = (yy >= 0 && yy < G.g) ? e - a : 0, ftot += fn[r];
        }
        if (want_full && ftot <= QCAP) {
#pragma unroll
            for (int r = 0; r < QROWS; ++r) lo[r] = flo[r], n[r] = fn[r];
            tot = ftot, inner = false;
        }
    }
    if (live && tot > QCAP) todo = true, live = false;  // too many for the registers
    if (!live) tot = 0;
    // lane position q = ql + 16 j in the concatenated rows -> sorted position q + off, off = lo_r - (n_0 + .. + n_{r-1}) for
    // the row r that q falls into; as a sum of masked steps (from a chain of selects the compiler builds a table in scratch
    // memory and branches)
    int cum[QROWS], step[QROWS];
    {
        int run = 0, prev = lo[0];
        cum[0] = 0, step[0] = lo[0];
#pragma unroll
        for (int r = 1; r < QROWS; ++r) {
            run += n[r - 1];
            cum[r] = run;
            step[r] = (lo[r] - run) - prev, prev = lo[r] - run;
        }
    }
    int key[CPL], idx[CPL];
    const int last = max(N - 1, 0);
#pragma unroll
    for (int j = 0; j < CPL; ++j) key[j] = 0x7fffffff, idx[j] = 0x7fffffff;
    // Slots in groups of QB; a group no centre of this wave reaches is skipped everywhere below (nsl is wave-uniform).  Inside
    // a group every load is issued before the first distance is computed -- unconditional 16-byte loads from clamped
    // positions, pinned by an empty asm so that the compiler neither splits them nor sinks the index word behind the
    // radius test.
    int nsl = (tot + QL - 1) / QL;
    {
        unsigned long long m = __ballot(nsl > QB) ;
        nsl = m ? (__ballot(nsl > 2 * QB) ? (__ballot(nsl > 3 * QB) ? 4 * QB : 3 * QB) : 2 * QB) : QB;
    }
    auto batch = [&, tot, ql, last](auto J0) __attribute__((always_inline)) {
        constexpr int j0 = decltype(J0)::value, j1 = j0 + QB;
        float4 p[QB];
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const int q = ql + QL * j;
            int off = step[0];
#pragma unroll
            for (int r = 1; r < QROWS; ++r) off += step[r] & -(int)(q >= cum[r]);
            p[j - j0] = sorted[q < tot ? min(q + off, last) : 0];
        }
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            float4 &c = p[j - j0];
            asm volatile("" : "+v"(c.x), "+v"(c.y), "+v"(c.z), "+v"(c.w));
        }
#pragma unroll
        for (int j = j0; j < j1; ++j) {
            const float4 c = p[j - j0];
            const float d = exp_dist(cx_, cy_, cz_, caa, c.x, c.y, c.z, sq3(c.x, c.y, c.z));
            const bool in = (ql + QL * j < tot) && d <= r2;
            key[j] = in ? fkey(d) : 0x7fffffff;
            idx[j] = in ? __float_as_int(c.w) : 0x7fffffff;
        }
    };
    batch(std::integral_constant<int, 0>{});
    if (nsl > QB) batch(std::integral_constant<int, QB>{});
    if (nsl > 2 * QB) batch(std::integral_constant<int, 2 * QB>{});
    if (nsl > 3 * QB) batch(std::integral_constant<int, 3 * QB>{});
    // ---- how many within the radius, and the nearest (smallest key, then smallest index)
    int cnt = 0, kb = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (j % QB == 0 && j >= nsl) break;
        cnt += key[j] != 0x7fffffff, kb = min(kb, key[j]);
    }
    cnt = row_sum(cnt), kb = row_min(kb);
    int ib = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (j % QB == 0 && j >= nsl) break;
        ib = min(ib, key[j] == kb ? idx[j] : 0x7fffffff);
    }
    ib = row_min(ib);
    if (live && cnt == 0) todo = true, live = false;  // nothing within the radius here: the full search decides
    // ---- K-th smallest key: probes t with c = count(key < t); invariant c(lo_k) < K < c(hi_k)
    bool sel = live && cnt > K, exact = false;
    int lo_k = kb, hi_k = fkey(r2) + 1, c_lo = 0, c_hi = cnt;
    for (int it = 0; __ballot(sel && !exact && (unsigned)hi_k - (unsigned)lo_k > 1u); ++it) {
        const bool act = sel && !exact && (unsigned)hi_k - (unsigned)lo_k > 1u;
        int t;
        if (it < 12) {  // interpolate in the distance domain (simulated on the benchmark scans: 6.9 probes per wave against 9.1
                       // with three interpolations + bisection; bisection afterwards bounds the tie case)
            const float dl = fmaxf(fkey_inv(lo_k), 0.f), dh = fkey_inv(hi_k);
            // (the probe is a guess that is clamped into the bracket below: a reciprocal approximation instead of an IEEE
            // division, ten instructions fewer per probe)
            const float td = dl + (dh - dl) * ((float)(K - c_lo) * __builtin_amdgcn_rcpf((float)(c_hi - c_lo)));
            t = fkey(td);
            t = (t <= lo_k || t >= hi_k) ? lo_k + (int)(((unsigned)hi_k - (unsigned)lo_k) >> 1) : t;
        } else {
            t = lo_k + (int)(((unsigned)hi_k - (unsigned)lo_k) >> 1);
        }
        int c = 0;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            if (j % QB == 0 && j >= nsl) break;
            c += key[j] < t;
        }
        c = row_sum(c);
        if (act) {
            if (c == K) exact = true, hi_k = t;
            else if (c < K) lo_k = t, c_lo = c;
            else hi_k = t, c_hi = c;
        }
    }
    // separator: keys below it are the answer (all valid keys when no selection was needed)
    const int sep = sel ? hi_k : 0x7fffffff;
    const bool tie = sel && !exact;
    // ---- inner block: is everything at or below the bound inside it?
    if (live && inner && !tie) {
        const float u = fx - flx, v = fy - fly;
        const float m = (float)G.h + fminf(fminf(u, 1.f - u), fminf(v, 1.f - v)) - 1e-3f;  // cell units to the border
        const float rho = m / G.inv_cs;
        const float bound = sel ? fkey_inv(sep) : r2;  // sep >= the K-th distance
        if (!((bound + G.err2) * 1.004f < rho * rho)) todo = true, live = false;
    }
    if (live && tie) {  // exact replay of torch.topk's choice by the tie kernel (it rewrites the whole row)
        int slot = 0;
        if (ql == 0) slot = atomicAdd(tie_count, 1);
        slot = __shfl(slot, (threadIdx.x & 63) & ~(QL - 1), 64);  // from the row's first lane
        if (slot < TIE_CAP) {
            if (ql == 0) tie_rows[slot] = (int)row;
        } else {
            todo = true;  // queue full: resolved in place by knn_grid_kernel
        }
        live = false;
    }
    {   // one atomic per wave for the rows left to the full search
        const unsigned long long tm = __ballot(todo && ql == 0);
        if (tm) {
            const int lane = threadIdx.x & 63;
            int base = 0;
            if (lane == __builtin_ctzll(tm)) base = atomicAdd(todo_count, __popcll(tm));
            base = __shfl(base, __builtin_ctzll(tm), 64);
            if (todo && ql == 0) {
                out[0] = KNN_TODO_MARK;
                const int slot = base + __popcll(tm & ((1ull << lane) - 1ull));
                if (slot < todo_cap) todo_rows[slot] = (int)row;
            }
        }
    }
    if (!live) return;
    // ---- output: nearest first, then the other selected candidates, then the nearest again as filler
    int mine = 0;
    bool take[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        take[j] = false;
        if (j >= nsl) continue;
        take[j] = key[j] < sep && !(key[j] == kb && idx[j] == ib);
        mine += take[j];
    }
    int pos = 1 + row_scan(mine) - mine;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        if (j % QB == 0 && j >= nsl) break;
        if (take[j]) out[pos] = idx[j], ++pos;
    }
    const int nsel = min(cnt, K);
    if (ql == 0) out[0] = ib;
    for (int k = nsel + ql; k < K; k += QL) out[k] = ib;
}

// ---------------------------------------------------------------------------------------------
// Boundary-tie rows of the grid search: exact emulation of torch.topk's heap-select, one workgroup
// per queued row.  The whole workgroup evaluates TIE_U * TIE_T distances per step (in index order: chunk (u, wave) covers
// 64 consecutive points) and flags the ones below the heap top; only those are replayed sequentially by thread 0,
// re-checked against the moving top -- exactly the elements std::__heap_select would have touched, in its order.
// The next batch of points is fetched while the replay runs.
// ---------------------------------------------------------------------------------------------
constexpr int TIE_U = 8;    // points per thread per step (loads in flight per thread: latency, not bandwidth, bounds a row)
constexpr int TIE_HEAD = 4096;  // points walked step by step before the filter pass (a multiple of TIE_U * TIE_T)
constexpr int TIE_LIST = 512;   // candidates one wave may keep in the filter pass
constexpr int TIE_T = 256;  // threads per row: the sequential replay bounds a row, so many small workgroups beat few large ones

__device__ __forceinline__ float rlane(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int rlane(int v, int l) { return __builtin_amdgcn_readlane(v, l); }
// "write lane l": a select on the lane id (this toolchain has no v_writelane builtin; two VALU ops are as cheap)
__device__ __forceinline__ void wlane(int &v, int l, int x) { v = lane_id() == l ? x : v; }
__device__ __forceinline__ void wlane(float &v, int l, float x) { v = lane_id() == l ? x : v; }

// heap_adjust with the heap held in registers: lane j of the wave owns entry j, every index is wave-uniform,
// so the walk is v_readlane plus lane-id selects with scalar indices -- no LDS round trips on the serial path.
__device__ __forceinline__ void heap_adjust_reg(float &hv, int &hi, int hole, int len, float val, int vi) {
    const int top = hole;
    int child = hole;
    while (child < (len - 1) / 2) {
        child = 2 * (child + 1);
        if (rlane(hv, child) < rlane(hv, child - 1)) child--;
        wlane(hv, hole, rlane(hv, child)), wlane(hi, hole, rlane(hi, child));
        hole = child;
    }
    if ((len & 1) == 0 && child == (len - 2) / 2) {
        child = 2 * (child + 1);
        wlane(hv, hole, rlane(hv, child - 1)), wlane(hi, hole, rlane(hi, child - 1));
        hole = child - 1;
    }
    int parent = (hole - 1) / 2;
    while (hole > top && rlane(hv, parent) < val) {
        wlane(hv, hole, rlane(hv, parent)), wlane(hi, hole, rlane(hi, parent));
        hole = parent, parent = (hole - 1) / 2;
    }
    wlane(hv, hole, val), wlane(hi, hole, vi);
}

// std::__pop_heap + push of `val` at the root (the only heap_adjust the replay issues), evaluated for all nodes at
// once instead of walking the tree with dependent lane reads (one wave runs the replay: every dependent instruction is
// exposed latency):
//   * __adjust_heap sends the hole down through the bigger child of each node -- the right one unless it is smaller
//     than the left, the single (left) child of node (len-2)/2 when len is even.  "I am my parent's bigger child"
//     needs only my sibling's value: one ds_bpermute;
//   * a node is on the hole's way down iff it and all its ancestors below the root are their parent's bigger child:
//     six shift-and-test steps on the ballot of that predicate, independent per lane;
//   * along that chain the old values fall monotonically (heap property), so the push-up of `val` from the leaf
//     stops at depth k = number of chain nodes below the root whose value is >= val: chain nodes above depth k take
//     their bigger child's entry (two more bpermute pairs, requested up front), the one at depth k takes (val, vi),
//     deeper ones end up where they started.
__device__ __forceinline__ void heap_replace_top(float &hv, int &hi, int len, float val, int vi) {
    const int j = lane_id();
    const int left = 2 * j + 1, right = 2 * j + 2, sib = (j & 1) ? j + 1 : j - 1;
    const float sv = __int_as_float(__builtin_amdgcn_ds_bpermute(sib << 2, __float_as_int(hv)));
    const float cl = __int_as_float(__builtin_amdgcn_ds_bpermute(left << 2, __float_as_int(hv)));
    const float cr = __int_as_float(__builtin_amdgcn_ds_bpermute(right << 2, __float_as_int(hv)));
    const int il = __builtin_amdgcn_ds_bpermute(left << 2, hi), ir = __builtin_amdgcn_ds_bpermute(right << 2, hi);
    const bool bigger = j < len && ((j & 1) ? (j + 1 >= len || sv < hv) : !(hv < sv));
    const unsigned long long chain = __ballot(bigger) | 1ull;  // bit 0: the root starts the chain
    bool on = true;
    int a = j;
#pragma unroll
    for (int step = 0; step < 6; ++step) {  // 64 nodes: at most five ancestors; the root repeats harmlessly
        on = on && ((chain >> a) & 1ull);
        a = max((a - 1) >> 1, 0);
    }
    const int depth = 31 - __clz(j + 1);
    const bool take_left = right < len ? (cr < cl) : true;
    const float cv = take_left ? cl : cr;
    const int ci = take_left ? il : ir;
    const int k = __popcll(__ballot(on && j != 0 && hv >= val));
    if (on && depth < k) hv = cv, hi = ci;
    if (on && depth == k) hv = val, hi = vi;
}

// Queued boundary-tie rows of the grid search in torch.topk's nth_element regime: one wave per row.
__global__ __launch_bounds__(64) void knn_tie_nth_kernel(const float *__restrict__ points_all,
                                                         const int32_t *__restrict__ lengths,
                                                         const float *__restrict__ centers_all, int N, int S, int K,
                                                         float r2, const int *__restrict__ tie_count,
                                                         const int32_t *__restrict__ tie_rows,
                                                         int32_t *__restrict__ idx_all) {
    extern __shared__ VI s_row[];
    __shared__ float s_hv[KMAX];
    __shared__ int s_hi[KMAX];
    const int n_rows = min(*tie_count, TIE_CAP);
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int row = tie_rows[r], b = row / S;
        const float *pts = points_all + (size_t)b * N * 3;
        const int len = min(max(lengths[b], 0), N);
        const float *cp = centers_all + (size_t)row * 3;
        nth_select_exact(pts, len, N, K, cp[0], cp[1], cp[2], sq3(cp[0], cp[1], cp[2]), s_row, (LdsF)s_hv, (LdsI)s_hi);
        emit_heap_result((LdsF)s_hv, (LdsI)s_hi, K, r2, idx_all + (size_t)row * K);
        wave_mem_sync();
    }
}

__global__ __launch_bounds__(TIE_T) void knn_tie_kernel(const float *__restrict__ points_all,
                                                       const int32_t *__restrict__ lengths,
                                                       const float *__restrict__ centers_all, int N, int S, int K,
                                                       float r2, const int *__restrict__ tie_count,
                                                       const int32_t *__restrict__ tie_rows,
                                                       int32_t *__restrict__ idx_all) {
    __shared__ float s_hv[KMAX];
    __shared__ int s_hi[KMAX];
    __shared__ float s_dist[TIE_U * TIE_T];
    __shared__ unsigned long long s_mask[TIE_U * (TIE_T / 64)];
    __shared__ float s_top;
    __shared__ float s_cd[TIE_T / 64][TIE_LIST];
    __shared__ int s_ci[TIE_T / 64][TIE_LIST];
    __shared__ int s_cn[TIE_T / 64];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int n_rows = min(*tie_count, TIE_CAP);
    for (int r = blockIdx.x; r < n_rows; r += gridDim.x) {
        const int row = tie_rows[r], b = row / S;
        const float *pts = points_all + (size_t)b * N * 3;
        const int len = min(max(lengths[b], 0), N);
        const float *cp = centers_all + (size_t)row * 3;
        const float cx = cp[0], cy = cp[1], cz = cp[2], caa = sq3(cx, cy, cz);
        auto dist = [&](float x, float y, float z) -> float { return exp_dist(cx, cy, cz, caa, x, y, z, sq3(x, y, z)); };
        // wave 0 owns the heap: entry j in lane j
        float hv = __builtin_inff();
        int hi = 0x7fffffff;
        if (w == 0) {
            if (lane < K) {
                const int i = min(lane, len - 1);
                hv = dist(pts[3 * i], pts[3 * i + 1], pts[3 * i + 2]), hi = lane;
            }
            if (K >= 2) {  // std::__make_heap
                for (int parent = (K - 2) / 2;; --parent) {
                    heap_adjust_reg(hv, hi, parent, K, rlane(hv, parent), rlane(hi, parent));
                    if (parent == 0) break;
                }
            }
            if (lane == 0) s_top = hv;
        }
        float px[TIE_U], py[TIE_U], pz[TIE_U];
        auto fetch = [&](int base) {
#pragma unroll
            for (int u = 0; u < TIE_U; ++u) {
                const int i = min(base + u * TIE_T + t, len - 1);
                px[u] = pts[3 * i], py[u] = pts[3 * i + 1], pz[u] = pts[3 * i + 2];
            }
        };
        // replay of one 64-entry chunk (distance d, point index ii per lane) against the moving top
        // (the ballot is retaken against the new top after every insertion: the loop runs once per insertion, not
        // once per entry that was below the top when the chunk started -- early in a frame that is nearly all of them)
        auto replay = [&](float d, int ii) {
            unsigned long long mm = __ballot(d < rlane(hv, 0));
            while (mm) {
                const int l = __builtin_ctzll(mm);
                heap_replace_top(hv, hi, K, rlane(d, l), rlane(ii, l));  // std::__pop_heap
                mm = __ballot(d < rlane(hv, 0)) & ((~1ull) << l);
            }
        };
        // points [from, to) step by step: TIE_U * TIE_T distances per step, flagged chunks replayed by wave 0
        auto scan_steps = [&](int from, int to) {
            fetch(from);
            for (int base = from; base < to; base += TIE_U * TIE_T) {
                __syncthreads();  // s_top is final for this step; s_dist / s_mask are free again
                const float top = s_top;
                bool any = false;
#pragma unroll
                for (int u = 0; u < TIE_U; ++u) {
                    const int i = base + u * TIE_T + t;
                    const float d = i < to ? dist(px[u], py[u], pz[u]) : __builtin_inff();
                    s_dist[u * TIE_T + t] = d;
                    const unsigned long long m = __ballot(d < top);
                    if (lane == 0) s_mask[u * (TIE_T / 64) + w] = m;
                    any |= m != 0;
                }
                if (base + TIE_U * TIE_T < to) fetch(base + TIE_U * TIE_T);
                if (__syncthreads_or(any) && w == 0) {
                    for (int c = 0; c < TIE_U * (TIE_T / 64); ++c)
                        if (s_mask[c] != 0) replay(s_dist[c * 64 + lane], base + c * 64 + lane);
                    if (lane == 0) s_top = hv;
                }
            }
        };
        // The top only falls, and it falls fast: after the first few thousand points it is the K-th smallest of
        // them, which less than 1 % of the rest undercut.  So the head of the frame is walked step by step, and the
        // rest goes through ONE filter pass against the top reached there (a superset of everything heap-select
        // would touch): each wave compacts its contiguous quarter of the range, in index order, into its own LDS
        // list, and wave 0 replays the four lists back to back.  A list that overflows (a frame whose head is far
        // from the centre) sends the row down the step-by-step path for the whole range.
        const int stop = min(len, K + TIE_HEAD);
        scan_steps(K, stop);
        if (stop < len) {
            __syncthreads();
            const float top = s_top;
            const int seg = (((len - stop + TIE_T / 64 - 1) / (TIE_T / 64)) + 63) & ~63;
            const int beg = stop + w * seg, end = min(len, beg + seg);
            int cn = 0;
            for (int i0 = beg; i0 < end; i0 += 64 * TIE_U) {
                float fx[TIE_U], fy[TIE_U], fz[TIE_U];
#pragma unroll
                for (int u = 0; u < TIE_U; ++u) {
                    const int i = min(i0 + u * 64 + lane, len - 1);
                    fx[u] = pts[3 * i], fy[u] = pts[3 * i + 1], fz[u] = pts[3 * i + 2];
                }
#pragma unroll
                for (int u = 0; u < TIE_U; ++u) {
                    const int i = i0 + u * 64 + lane;
                    const float d = i < end ? dist(fx[u], fy[u], fz[u]) : __builtin_inff();
                    const unsigned long long m = __ballot(d < top);
                    if (m == 0) continue;
                    const int pos = cn + __popcll(m & ((1ull << lane) - 1ull));
                    if (d < top && pos < TIE_LIST) s_cd[w][pos] = d, s_ci[w][pos] = i;
                    cn += __popcll(m);
                }
            }
            if (lane == 0) s_cn[w] = cn;
            __syncthreads();
            bool overflow = false;
#pragma unroll
            for (int k = 0; k < TIE_T / 64; ++k) overflow |= s_cn[k] > TIE_LIST;
            if (overflow) {
                scan_steps(stop, len);
            } else if (w == 0) {
                for (int k = 0; k < TIE_T / 64; ++k) {
                    const int n = s_cn[k];
                    for (int c0 = 0; c0 < n; c0 += 64) {
                        const bool in = c0 + lane < n;
                        replay(in ? s_cd[k][c0 + lane] : __builtin_inff(), in ? s_ci[k][c0 + lane] : 0);
                    }
                }
            }
        }
        if (w == 0) {
            if (lane < K) s_hv[lane] = hv, s_hi[lane] = hi;
            wave_mem_sync();
            emit_heap_result((LdsF)s_hv, (LdsI)s_hi, K, r2, idx_all + (size_t)row * K);
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// Self kNN of one cloud (pre-processing filters: OutlierFilter / LowPassFilter, reference
// dataloader/transforms.py:230-289, which call pytorch3d.knn_points(p, p, K+1) and drop the first column).
// Exact, no radius: one wave per point searches the (2R+1)^2 cell block around it and accepts the result when
// the K-th distance is within R cells (every point outside the block is farther than that); otherwise R doubles
// (sparse far-range points) until the block covers the grid.  Distances are the direct form (dx^2+dy^2)+dz^2;
// the K+1 nearest (the point itself first) are ordered by (distance, index) and column 0 is dropped.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WPB * 64) void knn_self_kernel(const float *__restrict__ pts, int N, int K,
                                                            const KnnGrid *__restrict__ hdr,
                                                            const int *__restrict__ start,
                                                            const float4 *__restrict__ sorted,
                                                            int32_t *__restrict__ idx_out,
                                                            float *__restrict__ dist2_out,
                                                            float *__restrict__ mean_dist_out) {
    __shared__ float s_d[WPB][CAP];
    __shared__ int s_i[WPB][CAP];
    __shared__ float s_td[WPB][TMPN];
    __shared__ int s_ti[WPB][TMPN];
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * WPB + w;
    if (q >= N) return;
    LdsF cd = (LdsF)s_d[w], td = (LdsF)s_td[w];
    LdsI ci = (LdsI)s_i[w], ti = (LdsI)s_ti[w];
    const KnnGrid G = hdr[0];
    const float cs = 1.0f / G.inv_cs;
    const int K1 = K + 1;
    const float qx = pts[3 * q], qy = pts[3 * q + 1], qz = pts[3 * q + 2];
    const int cx = min(max((int)floorf((qx - G.lox) * G.inv_cs), 0), G.g - 1);
    const int cy = min(max((int)floorf((qy - G.loy) * G.inv_cs), 0), G.g - 1);
    Ctr c;
    int cnt = 0;
    for (int R = 1;; R *= 2) {
        c.x = qx, c.y = qy, c.z = qz, c.aa = 0.f;
        c.thr = __builtin_inff(), c.gd = __builtin_inff(), c.gi = 0x7fffffff, c.cnt = 0, c.tie = false;
        const int x0 = max(cx - R, 0), x1 = min(cx + R, G.g - 1);
        for (int yy = max(cy - R, 0); yy <= min(cy + R, G.g - 1); ++yy) {
            const int lo = start[yy * G.g + x0], hi = start[yy * G.g + x1 + 1];
            for (int base = lo; base < hi; base += 64) {
                const int p = base + lane;
                float d = __builtin_inff();
                int i = 0x7fffffff;
                if (p < hi) {
                    const float4 t4 = sorted[p];
                    const float dx = qx - t4.x, dy = qy - t4.y, dz = qz - t4.z;
                    d = (dx * dx + dy * dy) + dz * dz;
                    i = __float_as_int(t4.w);
                }
                offer_d(c, d, i, K1, cd, ci, td, ti);
            }
        }
        wave_mem_sync();
        const bool whole = x0 == 0 && x1 == G.g - 1 && cy - R <= 0 && cy + R >= G.g - 1;
        cnt = min(c.cnt, K1);
        float kth = __builtin_inff();
        if (c.cnt > K1) {
            bool tie = false;
            kth = select_k(cd, ci, c.cnt, K1, td, ti, &tie);
        } else {
            for (int p = lane; p < c.cnt; p += 64) td[p] = cd[p], ti[p] = ci[p];
            wave_mem_sync();
            if (c.cnt == K1) {
                float v = lane < K1 ? td[lane] : -__builtin_inff();
                kth = wave_max_dpp(v);
            }
        }
        const float reach = (float)R * cs;
        if (whole || kth <= reach * reach * (1.f - 1e-5f)) break;
    }
    // order the (<= K+1) survivors by (distance, index): lane j ranks its own entry against the others
    const float dj = lane < cnt ? td[lane] : __builtin_inff();
    const int ij = lane < cnt ? ti[lane] : 0x7fffffff;
    int rank = 0;
    for (int o = 0; o < cnt; ++o) {
        const float dv = rlane(dj, o);
        const int iv = rlane(ij, o);
        rank += (dv < dj || (dv == dj && iv < ij)) ? 1 : 0;
    }
    // column 0 (the point itself) is dropped; missing neighbours (cloud smaller than K+1) repeat index q at d = 0
    if (lane < cnt && rank >= 1) {
        if (idx_out) idx_out[(size_t)q * K + rank - 1] = ij;
        if (dist2_out) dist2_out[(size_t)q * K + rank - 1] = dj;
    }
    if (lane >= cnt && lane < K1 && lane >= 1) {
        if (idx_out) idx_out[(size_t)q * K + lane - 1] = q;
        if (dist2_out) dist2_out[(size_t)q * K + lane - 1] = 0.f;
    }
    if (mean_dist_out) {  // mean over the K columns of sqrt(d^2), summed in column order
        float acc = 0.f;
        const float sd = (lane < cnt && rank >= 1) ? sqrtf(dj) : 0.f;
        for (int r = 1; r < K1; ++r) {
            const unsigned long long m = __ballot(lane < cnt && rank == r);
            acc += m ? rlane(sd, (int)__builtin_ctzll(m)) : 0.f;
        }
        if (lane == 0) mean_dist_out[q] = acc / (float)K;
    }
}

// ---------------------------------------------------------------------------------------------
// Point normals for LowPassFilter (reference dataloader/transforms.py:268-271: open3d 0.16
// PointCloud.estimate_normals(KDTreeSearchParamRadius(r))): covariance of the points within r of the point
// (itself included) from fp64 cumulants, normal = unit eigenvector of its smallest eigenvalue (cyclic Jacobi in
// fp64; the sign is arbitrary -- the caller only uses |n_i . n_j|); fewer than 3 points in range -> (0,0,1).
// One wave per point over the 3x3 cells of a grid with cell edge >= r.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__global__ __launch_bounds__(WPB * 64) void point_normals_kernel(const float *__restrict__ pts, int N, float r2,
                                                                 const KnnGrid *__restrict__ hdr,
                                                                 const int *__restrict__ start,
                                                                 const float4 *__restrict__ sorted,
                                                                 float *__restrict__ normals) {
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int q = blockIdx.x * WPB + w;
    if (q >= N) return;
    const KnnGrid G = hdr[0];
    const float qx = pts[3 * q], qy = pts[3 * q + 1], qz = pts[3 * q + 2];
    const int cx = min(max((int)floorf((qx - G.lox) * G.inv_cs), 0), G.g - 1);
    const int cy = min(max((int)floorf((qy - G.loy) * G.inv_cs), 0), G.g - 1);
    double m[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0};  // n, x, y, z, xx, xy, xz, yy, yz, zz
    const int x0 = max(cx - 1, 0), x1 = min(cx + 1, G.g - 1);
    for (int yy = max(cy - 1, 0); yy <= min(cy + 1, G.g - 1); ++yy) {
        const int lo = start[yy * G.g + x0], hi = start[yy * G.g + x1 + 1];
        for (int p = lo + lane; p < hi; p += 64) {
            const float4 t4 = sorted[p];
            const float dx = qx - t4.x, dy = qy - t4.y, dz = qz - t4.z;
            if ((dx * dx + dy * dy) + dz * dz < r2) {
                const double X = t4.x, Y = t4.y, Z = t4.z;
                m[0] += 1.0, m[1] += X, m[2] += Y, m[3] += Z;
                m[4] += X * X, m[5] += X * Y, m[6] += X * Z, m[7] += Y * Y, m[8] += Y * Z, m[9] += Z * Z;
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) m[k] = wave_sum_f64(m[k]);
    if (lane != 0) return;
    float nx = 0.f, ny = 0.f, nz = 1.f;
    if (m[0] >= 3.0) {
        const double n = m[0], mx = m[1] / n, my = m[2] / n, mz = m[3] / n;
        double A[3][3] = {{m[4] / n - mx * mx, m[5] / n - mx * my, m[6] / n - mx * mz},
                          {0, m[7] / n - my * my, m[8] / n - my * mz},
                          {0, 0, m[9] / n - mz * mz}};
        A[1][0] = A[0][1], A[2][0] = A[0][2], A[2][1] = A[1][2];
        double V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
        for (int sweep = 0; sweep < 12; ++sweep) {
            const double off = fabs(A[0][1]) + fabs(A[0][2]) + fabs(A[1][2]);
            if (off <= 1e-300 || off <= 1e-18 * (fabs(A[0][0]) + fabs(A[1][1]) + fabs(A[2][2]))) break;
            for (int p = 0; p < 2; ++p)
                for (int r = p + 1; r < 3; ++r) {
                    if (A[p][r] == 0.0) continue;
                    const double theta = (A[r][r] - A[p][p]) / (2.0 * A[p][r]);
                    const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                    const double cs_ = 1.0 / sqrt(t * t + 1.0), sn = t * cs_;
                    for (int k = 0; k < 3; ++k) {  // A <- A J
                        const double akp = A[k][p], akr = A[k][r];
                        A[k][p] = cs_ * akp - sn * akr, A[k][r] = sn * akp + cs_ * akr;
                    }
                    for (int k = 0; k < 3; ++k) {  // A <- J^T A,  V <- V J
                        const double apk = A[p][k], ark = A[r][k];
                        A[p][k] = cs_ * apk - sn * ark, A[r][k] = sn * apk + cs_ * ark;
                        const double vkp = V[k][p], vkr = V[k][r];
                        V[k][p] = cs_ * vkp - sn * vkr, V[k][r] = sn * vkp + cs_ * vkr;
                    }
                }
        }
        int e = 0;
        if (A[1][1] < A[e][e]) e = 1;
        if (A[2][2] < A[e][e]) e = 2;
        const double vx = V[0][e], vy = V[1][e], vz = V[2][e], nr = sqrt(vx * vx + vy * vy + vz * vz);
        if (nr > 0.0) nx = (float)(vx / nr), ny = (float)(vy / nr), nz = (float)(vz / nr);
    }
    normals[3 * (size_t)q] = nx, normals[3 * (size_t)q + 1] = ny, normals[3 * (size_t)q + 2] = nz;
}

// ---------------------------------------------------------------------------------------------
// Querier.ball_query (utils.py:57-73): the K smallest INDICES among the points within the radius, ascending,
// padded with the first of them.  One wave per centre scans the frame in index order and stops at K hits.
// (A centre with no point in range gets index N in every slot in the reference -- an out-of-range gather
// there -- here the slots are filled with 0.)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(WPB * 64) void ball_query_kernel(const float *__restrict__ points_all,
                                                              const int32_t *__restrict__ lengths,
                                                              const float *__restrict__ centers_all, int N, int S,
                                                              int K, float r2, int32_t *__restrict__ idx_all) {
    const int b = blockIdx.y, w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int s = blockIdx.x * WPB + w;
    if (s >= S) return;
    const float *pts = points_all + (size_t)b * N * 3;
    const int len = min(max(lengths[b], 0), N);
    const float *c = centers_all + ((size_t)b * S + s) * 3;
    const float cx = c[0], cy = c[1], cz = c[2], caa = sq3(cx, cy, cz);
    int32_t *out = idx_all + ((size_t)b * S + s) * K;
    int cnt = 0, first = 0;
    for (int base = 0; base < len && cnt < K; base += 64) {
        const int i = base + lane;
        bool in = false;
        if (i < len) {
            const float x = pts[3 * i], y = pts[3 * i + 1], z = pts[3 * i + 2];
            in = !(exp_dist(cx, cy, cz, caa, x, y, z, sq3(x, y, z)) > r2);
        }
        const unsigned long long m = __ballot(in);
        if (m) {
            if (cnt == 0) first = base + __builtin_ctzll(m);
            const int pos = cnt + __popcll(m & ((1ull << lane) - 1ull));
            if (in && pos < K) out[pos] = i;
            cnt += __popcll(m);
        }
    }
    for (int p = min(cnt, K) + lane; p < K; p += 64) out[p] = first;
}

}  // namespace

extern "C" int dpm_ball_query(const float *points, const int32_t *lengths, const float *centers, int B, int N, int S,
                              int K, double radius, int32_t *idx, dpm_stream_t stream) {
    DPM_CHECK_ARG(points && lengths && centers && idx);
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && radius > 0.0);
    hipLaunchKernelGGL(ball_query_kernel, dim3(dpm_cdiv(S, WPB), B), dim3(WPB * 64), 0, (hipStream_t)stream, points,
                       lengths, centers, N, S, K, (float)(radius * radius), idx);
    return dpm_launch_status();
}

extern "C" size_t dpm_knn_workspace_bytes(int B, int N) {
    if (N < GRID_MIN_N) return 0;
    return 1024 + sizeof(KnnGrid) * (size_t)B + sizeof(int) * (size_t)B * (GDIM * GDIM + 1) +
           (size_t)B * (size_t)N * sizeof(float4) + 256 + sizeof(int32_t) * (size_t)TIE_CAP + sizeof(int32_t) * (size_t)B * (size_t)N;
}

namespace {
struct KnnWs {  // layout of the grid workspace (dpm_knn_workspace_bytes)
    KnnGrid *hdr;
    int *start;
    float4 *sorted;
    int *tie_count;  // [0] = queued tie rows, [1] = rows the fast search left to the full one; the lists follow
    int32_t *tie_rows;
    int32_t *todo_rows;
    int todo_cap;
};
KnnWs carve(void *workspace, int B, int N) {
    KnnWs w;
    uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    w.hdr = (KnnGrid *)p;
    p = (p + sizeof(KnnGrid) * (size_t)B + 255) & ~(uintptr_t)255;
    w.start = (int *)p;
    p = (p + sizeof(int) * (size_t)B * (GDIM * GDIM + 1) + 255) & ~(uintptr_t)255;
    w.sorted = (float4 *)p;
    p = (p + sizeof(float4) * (size_t)B * (size_t)N + 255) & ~(uintptr_t)255;
    w.tie_count = (int *)p;
    w.tie_rows = (int32_t *)(p + 64);
    w.todo_rows = w.tie_rows + TIE_CAP;
    w.todo_cap = (size_t)B * (size_t)N < (size_t)0x7fffffff ? (int)((size_t)B * (size_t)N) : 0x7fffffff;
    return w;
}
void launch_grid_build(const float *points, const int32_t *lengths, int B, int N, double radius, const KnnWs &w,
                       hipStream_t st) {
    // cell edge > sqrt(r^2 + 2e-5 max(1, max |p|^2)): the expanded-form distance can undershoot the true one by ~1.5e-6 on
    // unit-ball coordinates and proportionally more on larger ones (the kernel knows the frame's extent)
    const float cs_min = (float)(sqrt(radius * radius + 2e-5) * 1.002);
    // -DDPM_EXPERIMENT builds only, DPM_PRICE_KNN_BUILD=n: the (idempotent) build n more times -- what the kernel costs the
    // pipelined step is the step's growth
    const int extra = N >= 16384 ? dpm_knob("DPM_PRICE_KNN_BUILD", 0) : 0;
    for (int rep = 0; rep <= extra; ++rep)
        hipLaunchKernelGGL(knn_grid_build_kernel, dim3(B), dim3(1024), 0, st, points, lengths, N, cs_min, (float)(radius * radius),
                           w.hdr, w.start, w.sorted, w.tie_count);
}
// -DDPM_EXPERIMENT builds only, DPM_KNN_FAST=0: every row through the one-wave-per-centre search (the round-2 path; A/B
// measurements, scripts/knn_bench.py)
bool knn_fast_enabled() { return dpm_knob("DPM_KNN_FAST", 1) != 0; }
int launch_grid_search(const float *points, const int32_t *lengths, const float *centers, int B, int N, int S, int K,
                       float r2, int32_t *idx, const KnnWs &w, const int32_t *reuse_idx, const int32_t *center_src,
                       hipStream_t st) {
    const long long rows = (long long)B * S;
    if (rows > 0x7fffffffLL / KMAX) return DPM_EUNSUPPORTED;
    if (knn_fast_enabled()) {
        // a quarter wave per centre; what it cannot finish is listed for the full search (few rows: a fixed small grid)
        hipLaunchKernelGGL(knn_grid_fast_kernel, dim3(dpm_cdiv(S, WPB * 64 / QL), B), dim3(WPB * 64), (size_t)dpm_knob("DPM_KNN_LDS_PAD", 0), st, lengths, centers, N, S,
                           K, r2, w.hdr, w.start, w.sorted, idx, reuse_idx, center_src, w.tie_count, w.tie_rows, w.tie_count + 1,
                           w.todo_rows, w.todo_cap);
        hipLaunchKernelGGL(knn_grid_kernel, dim3((unsigned)(rows / WPB + 1 < 2048 ? rows / WPB + 1 : 2048)), dim3(WPB * 64), 0, st, points, lengths,
                           centers, N, S, K, r2, w.hdr, w.start, w.sorted, idx, reuse_idx, center_src, w.tie_count, w.tie_rows,
                           (const int *)(w.tie_count + 1), (const int32_t *)w.todo_rows, w.todo_cap, (int)rows);
    } else {
        hipLaunchKernelGGL(knn_grid_kernel, dim3(dpm_cdiv(rows, WPB)), dim3(WPB * 64), 0, st, points, lengths, centers, N,
                           S, K, r2, w.hdr, w.start, w.sorted, idx, reuse_idx, center_src, w.tie_count, w.tie_rows,
                           (const int *)nullptr, (const int32_t *)nullptr, 0, (int)rows);
    }
    if (dpm_knob("DPM_ABLATE_TIE", 0)) return dpm_launch_status();  // -DDPM_EXPERIMENT builds only: tied rows keep the plain selection
    if ((long long)K * 64 <= (long long)N)  // torch.topk's partial_sort regime: heap-select replay of the queued rows
        hipLaunchKernelGGL(knn_tie_kernel, dim3(2048), dim3(TIE_T), 0, st, points, lengths, centers, N, S, K, r2,
                           w.tie_count, w.tie_rows, idx);
    else  // nth_element regime (N < 64 K <= 2048)
        hipLaunchKernelGGL(knn_tie_nth_kernel, dim3(1024), dim3(64), (sizeof(VI) + 4) * (size_t)N, st, points, lengths, centers, N,
                           S, K, r2, w.tie_count, w.tie_rows, idx);
    return dpm_launch_status();
}
}  // namespace

extern "C" int dpm_knn_hybrid_reuse(const float *points, const int32_t *lengths, const float *centers, int B, int N,
                                    int S, int K, double radius, int32_t *idx, void *workspace,
                                    const int32_t *reuse_idx, const int32_t *center_src, dpm_stream_t stream) {
    DPM_CHECK_ARG(points && lengths && centers && idx);
    DPM_CHECK_ARG((reuse_idx == nullptr) == (center_src == nullptr));
    DPM_CHECK_ARG(B >= 1 && N >= 1 && S >= 1 && K >= 1 && radius > 0.0);
    if (K > KMAX) return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    const float r2 = (float)(radius * radius);
    if (N >= GRID_MIN_N && workspace) {
        const KnnWs w = carve(workspace, B, N);
        launch_grid_build(points, lengths, B, N, radius, w, st);
        return launch_grid_search(points, lengths, centers, B, N, S, K, r2, idx, w, reuse_idx, center_src, st);
    }
    int nth_rows = (long long)K * 64 > (long long)N;  // torch.topk's nth_element regime: tied rows are replayed
    // The replay keeps a row of N (value, index) pairs + scratch per wave in dynamic LDS, next to the kernel's static lists.
    // A frame too long for what is left of the CU's 160 KB (about 1900 points; the encoder sends frames from 1024 points on
    // to the grid search, so only a forced brute-force call gets here) runs without it: tied rows then keep the smaller index.
    size_t dyn = nth_rows ? (sizeof(VI) + 4) * (size_t)WPB * N : 0;
    if (dyn) {
        static size_t static_lds = 0;
        if (!static_lds) {
            hipFuncAttributes fa;
            static_lds = hipFuncGetAttributes(&fa, (const void *)knn_hybrid_kernel) == hipSuccess ? fa.sharedSizeBytes : 72 * 1024;
        }
        if (static_lds + dyn > 160 * 1024) nth_rows = 0, dyn = 0;
        else if (dyn > 32 * 1024)
            (void)hipFuncSetAttribute((const void *)knn_hybrid_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)dyn);
    }
    hipLaunchKernelGGL(knn_hybrid_kernel, dim3(dpm_cdiv(S, WPB * CPW), B), dim3(WPB * 64), dyn, st, points, lengths, centers, N, S,
                       K, r2, idx, reuse_idx, center_src, nth_rows);
    return dpm_launch_status();
}

// The two halves of the grid path as separate calls sharing one workspace: the grid depends on the points and the
// radius only, so a pipeline builds it next to the sampling (before any feature exists) and runs only the search in
// its feature stage.  One search per build (the build also resets the tie queue the search fills).
extern "C" int dpm_knn_build_grid(const float *points, const int32_t *lengths, int B, int N, double radius,
                                  void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(points && lengths && workspace && B >= 1 && N >= GRID_MIN_N && radius > 0.0);
    launch_grid_build(points, lengths, B, N, radius, carve(workspace, B, N), (hipStream_t)stream);
    return dpm_launch_status();
}

extern "C" int dpm_knn_hybrid_prebuilt(const float *points, const int32_t *lengths, const float *centers, int B, int N,
                                       int S, int K, double radius, int32_t *idx, void *workspace,
                                       dpm_stream_t stream) {
    DPM_CHECK_ARG(points && lengths && centers && idx && workspace);
    DPM_CHECK_ARG(B >= 1 && N >= GRID_MIN_N && S >= 1 && K >= 1 && radius > 0.0);
    if (K > KMAX) return DPM_EUNSUPPORTED;
    return launch_grid_search(points, lengths, centers, B, N, S, K, (float)(radius * radius), idx, carve(workspace, B, N),
                              nullptr, nullptr, (hipStream_t)stream);
}

extern "C" int dpm_knn_hybrid(const float *points, const int32_t *lengths, const float *centers, int B, int N,
                              int S, int K, double radius, int32_t *idx, void *workspace, dpm_stream_t stream) {
    return dpm_knn_hybrid_reuse(points, lengths, centers, B, N, S, K, radius, idx, workspace, nullptr, nullptr, stream);
}

// ---- self kNN (pre-processing filters) ------------------------------------------------------------------
extern "C" size_t dpm_knn_self_workspace_bytes(int N) { return dpm_knn_workspace_bytes(1, N > GRID_MIN_N ? N : GRID_MIN_N); }

extern "C" int dpm_knn_self(const float *xyz, int N, int K, double cell, int32_t *idx, float *dist2, float *mean_dist,
                            void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && workspace && N >= 1 && K >= 1 && cell > 0.0 && (idx || dist2 || mean_dist));
    if (K + 1 > KMAX) return DPM_EUNSUPPORTED;
    hipStream_t st = (hipStream_t)stream;
    uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    KnnGrid *hdr = (KnnGrid *)p;
    p = (p + sizeof(KnnGrid) + 255) & ~(uintptr_t)255;
    int *start = (int *)p;
    p = (p + sizeof(int) * (size_t)(GDIM * GDIM + 1) + 255) & ~(uintptr_t)255;
    float4 *sorted = (float4 *)p;
    p = (p + sizeof(float4) * (size_t)N + 255) & ~(uintptr_t)255;
    int *tie_count = (int *)p;
    // one "frame" of N points, all valid: the length lives in the workspace header area
    int32_t *len_dev = (int32_t *)(tie_count + 8);
    // written by the device itself (a fill, not a copy from this function's stack: the runtime may stage a pageable
    // source after we have returned, and a fill is capturable in a HIP graph)
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)len_dev, N, 1, st);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(knn_grid_build_kernel, dim3(1), dim3(1024), 0, st, xyz, len_dev, N, (float)cell, 0.f, hdr, start, sorted,
                       tie_count);
    hipLaunchKernelGGL(knn_self_kernel, dim3(dpm_cdiv(N, WPB)), dim3(WPB * 64), 0, st, xyz, N, K, hdr, start, sorted, idx,
                       dist2, mean_dist);
    return dpm_launch_status();
}

extern "C" int dpm_point_normals(const float *xyz, int N, double radius, float *normals, void *workspace,
                                 dpm_stream_t stream) {
    DPM_CHECK_ARG(xyz && normals && workspace && N >= 1 && radius > 0.0);
    hipStream_t st = (hipStream_t)stream;
    uintptr_t p = ((uintptr_t)workspace + 255) & ~(uintptr_t)255;
    KnnGrid *hdr = (KnnGrid *)p;
    p = (p + sizeof(KnnGrid) + 255) & ~(uintptr_t)255;
    int *start = (int *)p;
    p = (p + sizeof(int) * (size_t)(GDIM * GDIM + 1) + 255) & ~(uintptr_t)255;
    float4 *sorted = (float4 *)p;
    p = (p + sizeof(float4) * (size_t)N + 255) & ~(uintptr_t)255;
    int *tie_count = (int *)p;
    int32_t *len_dev = (int32_t *)(tie_count + 8);
    // written by the device itself (a fill, not a copy from this function's stack: the runtime may stage a pageable
    // source after we have returned, and a fill is capturable in a HIP graph)
    hipError_t e = hipMemsetD32Async((hipDeviceptr_t)len_dev, N, 1, st);
    if (e != hipSuccess) return (int)e;
    // cell edge slightly above the radius: the 3x3 block then contains every point strictly within it
    hipLaunchKernelGGL(knn_grid_build_kernel, dim3(1), dim3(1024), 0, st, xyz, len_dev, N, (float)(radius * 1.001), 0.f, hdr, start,
                       sorted, tie_count);
    hipLaunchKernelGGL(point_normals_kernel, dim3(dpm_cdiv(N, WPB)), dim3(WPB * 64), 0, st, xyz, N,
                       (float)(radius * radius), hdr, start, sorted, normals);
    return dpm_launch_status();
}
