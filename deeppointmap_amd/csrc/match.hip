// a13, the match kernel north_star names: pairwise descriptor similarity + soft-assignment match
// (reference network/decoder/decoder.py:181-191, `_descriptor_pairing`):
//     S = A B^T  (L2-normalised head outputs, M x N, contraction over C = 256: the one dense contraction -> MFMA)
//     P = softmax_row(S / tau) * softmax_col(S / tau);  (values, flat indices) = topk(P.flatten(), k)
// without the M x N matrix ever existing in memory.  Two launches over (row strips of 64) x (pairs):
//   match_stats_kernel  S strip (64 x N, fp32 MFMA, tile in registers) -> the strip's row maxima / row sums (whole rows
//                       live in one workgroup) and its PARTIAL column maxima / sums (64 of the M rows)
//   match_topk_kernel   the same S strip again (8.4 MFLOP: cheaper than 64 KB through HBM and back), column statistics
//                       folded over the strips, P computed in registers and parked in LDS (66 KB over the dead operand tiles), the strip's k largest by radix select over that tile;
//                       the LAST workgroup of a pair to finish (one atomic ticket per pair, no spinning) merges the
//                       strips' candidates into the pair's top-k.
// Strip-local selection and merge use ONE total order (value descending, flat index ascending), so the result is the
// exact top-k of the whole matrix under the rule of dpm_dual_softmax_topk -- which stays for N > 256 (map-vs-map).
// S is bit-identical to gemm_nt_mfma_kernel's (same instruction, same k order); the column sums are folded as
// sum_s psum_s * exp(pmax_s - cmax) instead of one pass against the final maximum (last-bit differences).
#include "dpm_common.h"

namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int MT_ROWS = 64, MT_COLS = 256, MT_KT = 32, MT_LD = MT_KT + 2, MT_T = 256;
constexpr int MT_KCAP = 2048;                                  // largest k of the fused path (lists alias the operand tiles)
constexpr int MT_SMEM = (MT_ROWS + MT_COLS) * MT_LD;           // floats: 43 520 B of operand tiles
constexpr int MT_PLD = MT_COLS + 4;                            // row stride of the P tile in LDS
constexpr int MT_SMEM2 = MT_ROWS * MT_PLD + 2 * MT_KCAP;       // second kernel: P tile (over the dead operand tiles) + lists
constexpr int MT_MCAP = MT_ROWS * MT_PLD / 2;                  // candidates (value, index) the dead P tile holds for the merge: 8 320
static_assert(MT_SMEM2 >= MT_SMEM && MT_SMEM >= 2 * MT_COLS, "the P tile and the column statistics reuse the operand tiles");

#ifdef DPM_EXPERIMENT
__device__ long long dpm_match_trace_buf[64];   // cycle stamps of pair 0 (scripts/debug/match_trace.py): slots 0-15 of the strip
                                                // that merges, 16-31 of match_stats_kernel's strip 0
#define DPM_MT_STAMP(slot)                                                                                        \
    do {                                                                                                          \
        if (blockIdx.y == 0 && threadIdx.x == 0) dpm_match_trace_buf[slot] = (long long)__builtin_readcyclecounter(); \
    } while (0)
#else
#define DPM_MT_STAMP(slot) do { } while (0)
#endif

struct MatchWs {
    float *rmax, *rsum;      // (batch, M)
    float *pmax, *psum;      // (batch, strips, N)
    int *ticket;             // (batch)
    float *cand_v;           // (batch, strips, kl)
    int *cand_i;
};

// S strip = A[row0 .. row0+63] B^T, all N <= 256 columns: wave w owns columns 64 w .. 64 w + 63 as 4 x 4 blocks of
// 16 x 16.  Lane layout of block (i, j) (B is the instruction's A operand, as in gemm.hip): row i*16 + (lane & 15),
// columns j*16 + (lane >> 4)*4 + 0..3.  Rows / columns beyond M / N read the last valid one (masked by the callers).
__device__ __forceinline__ void match_strip_gemm(const float *__restrict__ A, int rows_a, const float *__restrict__ B, int N,
                                                 int C, float *smem, f32x4 (&acc)[4][4]) {
    float (*As)[MT_LD] = reinterpret_cast<float (*)[MT_LD]>(smem);
    float (*Bs)[MT_LD] = reinterpret_cast<float (*)[MT_LD]>(smem + MT_ROWS * MT_LD);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int sr = t >> 3, sk = (t & 7) * 4;   // staging: 8 lanes per row (32 floats), 32 rows per pass
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 ar[2], br[8];
    auto request = [&](int k0) {
#pragma unroll
        for (int p = 0; p < 2; ++p)
            ar[p] = *reinterpret_cast<const float4 *>(A + (size_t)min(p * 32 + sr, rows_a - 1) * C + k0 + sk);
#pragma unroll
        for (int p = 0; p < 8; ++p)
            br[p] = *reinterpret_cast<const float4 *>(B + (size_t)min(p * 32 + sr, N - 1) * C + k0 + sk);
    };
    request(0);
    for (int k0 = 0; k0 < C; k0 += MT_KT) {
#pragma unroll
        for (int p = 0; p < 2; ++p) {
            float2 *d = reinterpret_cast<float2 *>(&As[p * 32 + sr][sk]);
            d[0] = make_float2(ar[p].x, ar[p].y), d[1] = make_float2(ar[p].z, ar[p].w);
        }
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            float2 *d = reinterpret_cast<float2 *>(&Bs[p * 32 + sr][sk]);
            d[0] = make_float2(br[p].x, br[p].y), d[1] = make_float2(br[p].z, br[p].w);
        }
        __syncthreads();
        if (k0 + MT_KT < C) request(k0 + MT_KT);
        float a[2][4], b[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) a[0][i] = As[i * 16 + (lane & 15)][lane >> 4];
#pragma unroll
        for (int j = 0; j < 4; ++j) b[0][j] = Bs[w * 64 + j * 16 + (lane & 15)][lane >> 4];
#pragma unroll
        for (int kk = 0; kk < MT_KT; kk += 4) {
            const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
            if (kk + 4 < MT_KT) {
#pragma unroll
                for (int i = 0; i < 4; ++i) a[nxt][i] = As[i * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
#pragma unroll
                for (int j = 0; j < 4; ++j) b[nxt][j] = Bs[w * 64 + j * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[cur][j], a[cur][i], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }
}

__device__ __forceinline__ float xor_max(float v, int m) { return fmaxf(v, __shfl_xor(v, m, 64)); }
__device__ __forceinline__ float xor_add(float v, int m) { return v + __shfl_xor(v, m, 64); }

__global__ __launch_bounds__(MT_T) void match_stats_kernel(const float *__restrict__ A, const float *__restrict__ B, int M, int N,
                                                           int C, float itau, MatchWs ws) {
    __shared__ __attribute__((aligned(16))) float smem[MT_SMEM];
    const int strip = blockIdx.x, pair = blockIdx.y, strips = gridDim.x;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int row0 = strip * MT_ROWS;
    if (strip == 0 && t == 0) ws.ticket[pair] = 0;   // the next launch counts the pair's finished strips from zero
    f32x4 acc[4][4];
    if (strip == 0) DPM_MT_STAMP(16);
    match_strip_gemm(A + ((size_t)pair * M + row0) * C, M - row0, B + (size_t)pair * N * C, N, C, smem, acc);
    if (strip == 0) DPM_MT_STAMP(17);
    const float NEG = -__builtin_inff();
    // x = S / tau, invalid entries -inf (they fall out of every maximum and add exp(-inf) = 0 to every sum)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool rv = row0 + i * 16 + (lane & 15) < M;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const bool cv = w * 64 + j * 16 + (lane >> 4) * 4 + q < N;
                acc[i][j][q] = rv && cv ? acc[i][j][q] * itau : NEG;
            }
    }
    // ---- rows: the wave holds 64 of a row's columns, spread over the four lanes with equal (lane & 15)
    float *xm = smem, *xs = smem + 4 * MT_ROWS;   // [4 waves][64 rows] each (the operand tiles are dead)
    float rm[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float m = NEG;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) m = fmaxf(m, acc[i][j][q]);
        m = xor_max(xor_max(m, 16), 32);
        if (lane < 16) xm[w * MT_ROWS + i * 16 + lane] = m;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = i * 16 + (lane & 15);
        rm[i] = fmaxf(fmaxf(xm[r], xm[MT_ROWS + r]), fmaxf(xm[2 * MT_ROWS + r], xm[3 * MT_ROWS + r]));
        float s = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) s += expf(acc[i][j][q] - rm[i]);
        s = xor_add(xor_add(s, 16), 32);
        if (lane < 16) xs[w * MT_ROWS + i * 16 + lane] = s;
    }
    __syncthreads();
    if (t < MT_ROWS && row0 + t < M) {
        ws.rmax[(size_t)pair * M + row0 + t] = fmaxf(fmaxf(xm[t], xm[MT_ROWS + t]), fmaxf(xm[2 * MT_ROWS + t], xm[3 * MT_ROWS + t]));
        ws.rsum[(size_t)pair * M + row0 + t] = (xs[t] + xs[MT_ROWS + t]) + (xs[2 * MT_ROWS + t] + xs[3 * MT_ROWS + t]);
    }
    // ---- columns: a column's 64 rows of this strip all sit in this wave (4 blocks x the 16 lanes with equal lane >> 4)
    float *pm_out = ws.pmax + ((size_t)pair * strips + strip) * N, *ps_out = ws.psum + ((size_t)pair * strips + strip) * N;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float m = fmaxf(fmaxf(acc[0][j][q], acc[1][j][q]), fmaxf(acc[2][j][q], acc[3][j][q]));
            m = xor_max(xor_max(xor_max(xor_max(m, 1), 2), 4), 8);
            float s = (expf(acc[0][j][q] - m) + expf(acc[1][j][q] - m)) + (expf(acc[2][j][q] - m) + expf(acc[3][j][q] - m));
            s = xor_add(xor_add(xor_add(xor_add(s, 1), 2), 4), 8);
            const int c = w * 64 + j * 16 + (lane >> 4) * 4 + q;
            if ((lane & 15) == 0 && c < N) pm_out[c] = m, ps_out[c] = s;
        }
    if (strip == 0) DPM_MT_STAMP(18);
}

// ---- block-wide exact top-k over items a thread enumerates itself (registers or memory) -------------------------------
// The bin that holds the rem-th key in walking order (descending: largest first): lane l of wave 0 owns four bins.
template <bool DESC>
__device__ __forceinline__ void pick_bin(const unsigned *hist, unsigned *sc /* [0] prefix [1] mask [2] rem */, int shift) {
    const int t = threadIdx.x;   // t < 64
    const int first = DESC ? 255 - 4 * t : 4 * t, step = DESC ? -1 : 1;
    const unsigned c0 = hist[first], c1 = hist[first + step], c2 = hist[first + 2 * step], c3 = hist[first + 3 * step];
    unsigned inc = c0 + c1 + c2 + c3;
    const unsigned own = inc;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const unsigned o = __shfl_up(inc, off, 64);
        if (t >= off) inc += o;
    }
    const unsigned rem0 = sc[2];
    const unsigned long long hit = __ballot(inc >= rem0);
    const int L = hit ? __builtin_ctzll(hit) : 63;
    if (t == L) {
        unsigned rem = rem0 - (inc - own);
        int b = first;
        if (c0 >= rem) {
        } else if (c0 + c1 >= rem) {
            b = first + step, rem -= c0;
        } else if (c0 + c1 + c2 >= rem) {
            b = first + 2 * step, rem -= c0 + c1;
        } else {
            b = first + 3 * step, rem -= c0 + c1 + c2;
        }
        sc[2] = rem, sc[0] |= (unsigned)b << shift, sc[1] |= 255u << shift;
        sc[3] = hist[b];   // population of the chosen bin (after the last pass: keys equal to the selected one)
    }
}

// Bitonic sort of np2 = 256 EPT (value, index) pairs in sv / si into (value descending, index ascending), EPT consecutive
// elements per thread in registers: compare-exchange steps at distances below EPT stay inside a thread, below 64 EPT inside a
// wave (lane exchanges), and only the last ones (partner in another wave) go through LDS with barriers -- 3 of the 55 steps at
// 1024 elements.  (All steps through LDS with a barrier each: 15 of the merge's 26 us.)
template <int EPT>
__device__ __forceinline__ void block_sort_regs(float *sv, int *si) {
    const int t = threadIdx.x;
    constexpr int NP2 = EPT * MT_T;
    float v[EPT];
    int ix[EPT];
#pragma unroll
    for (int i = 0; i < EPT; ++i) v[i] = sv[t * EPT + i], ix[i] = si[t * EPT + i];
    // own (a, ia) against partner (b, ib); take_first: this slot keeps the pair that sorts first
    auto keep = [](float a, int ia, float b, int ib, bool take_first, float &o, int &oi) {
        const bool a_first = a > b || (a == b && ia < ib);
        const bool own = a_first == take_first;
        o = own ? a : b, oi = own ? ia : ib;
    };
    for (int size = 2; size <= NP2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            if (stride >= EPT * 64) {          // partner in another wave
                __syncthreads();
#pragma unroll
                for (int i = 0; i < EPT; ++i) sv[t * EPT + i] = v[i], si[t * EPT + i] = ix[i];
                __syncthreads();
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int e = t * EPT + i, pe = e ^ stride;
                    const bool take_first = ((e & stride) == 0) == ((e & size) == 0);
                    keep(v[i], ix[i], sv[pe], si[pe], take_first, v[i], ix[i]);
                }
            } else if (stride >= EPT) {        // partner lane of the same wave
                const int m = stride / EPT;
#pragma unroll
                for (int i = 0; i < EPT; ++i) {
                    const int e = t * EPT + i;
                    const float b = __shfl_xor(v[i], m, 64);
                    const int ib = __shfl_xor(ix[i], m, 64);
                    const bool take_first = ((e & stride) == 0) == ((e & size) == 0);
                    keep(v[i], ix[i], b, ib, take_first, v[i], ix[i]);
                }
            } else {                           // both elements in this thread: static register pairs
#pragma unroll
                for (int sd = 1; sd < EPT; sd <<= 1) {
                    if (sd != stride) continue;
#pragma unroll
                    for (int i = 0; i < EPT; ++i) {
                        if (i & sd) continue;
                        const int j = i | sd;
                        const bool desc = ((t * EPT + i) & size) == 0;   // this block sorts "first" to the low slot
                        const bool a_first = v[i] > v[j] || (v[i] == v[j] && ix[i] < ix[j]);
                        if (a_first != desc) {
                            const float tv = v[i];
                            const int ti = ix[i];
                            v[i] = v[j], ix[i] = ix[j], v[j] = tv, ix[j] = ti;
                        }
                    }
                }
            }
        }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < EPT; ++i) sv[t * EPT + i] = v[i], si[t * EPT + i] = ix[i];
    __syncthreads();
}

// items(f): calls f(value bits, flat index) for each of the thread's valid items (values >= 0: their bit patterns order
// like the floats).  Leaves the k largest -- ties at the k-th value resolved toward smaller flat indices -- sorted
// (value descending, index ascending) in sv / si -- or, with sorted == false, in arbitrary order (a strip's candidates: the
// merge selects and sorts again).  k <= number of valid items, k <= MT_KCAP.
template <class Items>
__device__ __forceinline__ void block_topk(Items &&items, int k, float *sv, int *si, unsigned *hist, unsigned *sc, bool sorted) {
    const int t = threadIdx.x;
    auto select = [&](auto &&key_of, auto &&member, bool desc, unsigned want) {
        // 4 x 8-bit MSB radix select of the want-th key in walking order among the items with member(u, idx)
        if (t == 0) sc[0] = 0u, sc[1] = 0u, sc[2] = want;
        for (int pass = 0; pass < 4; ++pass) {
            const int shift = 24 - 8 * pass;
            hist[t] = 0u;   // MT_T == 256 bins
            __syncthreads();
            const unsigned prefix = sc[0], mask = sc[1];
            // runs of equal bins are counted in a register (dual-softmax products share their leading bits)
            unsigned run_bin = 0xffffffffu, run_cnt = 0u;
            items([&](unsigned u, int idx) {
                if (!member(u, idx)) return;
                const unsigned key = key_of(u, idx);
                if ((key & mask) != prefix) return;
                const unsigned bin = (key >> shift) & 255u;
                if (bin != run_bin) {
                    if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
                    run_bin = bin, run_cnt = 0u;
                }
                ++run_cnt;
            });
            if (run_cnt) atomicAdd(&hist[run_bin], run_cnt);
            __syncthreads();
            if (t < 64) {
                if (desc) pick_bin<true>(hist, sc, shift);
                else pick_bin<false>(hist, sc, shift);
            }
            __syncthreads();
        }
    };
    select([](unsigned u, int) { return u; }, [](unsigned, int) { return true; }, true, (unsigned)k);
    const unsigned thr = sc[0], take_eq = sc[2], c_eq = sc[3];
    __syncthreads();
    if (sorted) DPM_MT_STAMP(8);
    unsigned ithr = 0x7fffffffu;
    if (take_eq < c_eq) {   // not every element equal to the k-th value fits: the take_eq smallest flat indices do
        select([](unsigned, int idx) { return (unsigned)idx; }, [thr](unsigned u, int) { return u == thr; }, false, take_eq);
        ithr = sc[0];
        __syncthreads();
    }
    if (t == 0) sc[4] = 0u;
    int np2 = 1;
    while (np2 < k) np2 <<= 1;
    if (sorted)
        for (int e = k + t; e < np2; e += MT_T) sv[e] = -1.f, si[e] = 0x7fffffff;
    __syncthreads();
    const int lane = t & 63;
    items([&](unsigned u, int idx) {   // one counter update per wave and step, not one per survivor
        const bool take = u > thr || (u == thr && (unsigned)idx <= ithr);
        const unsigned long long m = __ballot(take);
        if (!take) return;
        const int leader = __builtin_ctzll(m);
        unsigned base = 0u;
        if (lane == leader) base = atomicAdd(&sc[4], (unsigned)__builtin_popcountll(m));
        base = (unsigned)__shfl((int)base, leader, 64);
        const unsigned p = base + (unsigned)__builtin_popcountll(m & ((1ull << lane) - 1ull));
        sv[p] = __uint_as_float(u), si[p] = idx;
    });
    __syncthreads();
    if (sorted) DPM_MT_STAMP(9);
    if (!sorted) return;
    if (np2 == 4 * MT_T) return block_sort_regs<4>(sv, si);
    if (np2 == 8 * MT_T) return block_sort_regs<8>(sv, si);
    for (int size = 2; size <= np2; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            for (int e = t; e < np2 / 2; e += MT_T) {
                const int lo = 2 * e - (e & (stride - 1)), hi = lo + stride;
                const bool desc = (lo & size) == 0;
                const float a = sv[lo], b = sv[hi];
                const int ia = si[lo], ib = si[hi];
                const bool a_first = a > b || (a == b && ia < ib);
                if (a_first != desc) sv[lo] = b, sv[hi] = a, si[lo] = ib, si[hi] = ia;
            }
            __syncthreads();
        }
}

__global__ __launch_bounds__(MT_T) void match_topk_kernel(const float *__restrict__ A, const float *__restrict__ B, int M, int N,
                                                          int C, float itau, int k, MatchWs ws, float *__restrict__ out_v,
                                                          int32_t *__restrict__ out_i) {
    __shared__ __attribute__((aligned(16))) float smem[MT_SMEM2];
    __shared__ unsigned hist[256];
    __shared__ unsigned sc[8];
    const int strip = blockIdx.x, pair = blockIdx.y, strips = gridDim.x;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int row0 = strip * MT_ROWS;
    f32x4 acc[4][4];
    if (strip == 0) DPM_MT_STAMP(0);
    match_strip_gemm(A + ((size_t)pair * M + row0) * C, M - row0, B + (size_t)pair * N * C, N, C, smem, acc);
    if (strip == 0) DPM_MT_STAMP(1);
    // column statistics over ALL rows: fold the strips' partial (max, sum) pairs, in strip order
    float *cm = smem, *cs = smem + MT_COLS;   // the operand tiles are dead
    float *pt = smem, *sv = smem + MT_ROWS * MT_PLD;
    int *si = reinterpret_cast<int *>(sv + MT_KCAP);
    if (t < N) {
        const float *pm = ws.pmax + (size_t)pair * strips * N + t, *ps = ws.psum + (size_t)pair * strips * N + t;
        float m = -__builtin_inff();
        for (int s = 0; s < strips; ++s) m = fmaxf(m, pm[(size_t)s * N]);
        float sum = 0.f;
        for (int s = 0; s < strips; ++s) sum += ps[(size_t)s * N] * expf(pm[(size_t)s * N] - m);
        cm[t] = m, cs[t] = sum;
    }
    __syncthreads();
    // P = softmax_row * softmax_col, the expression of dual_softmax_kernel; invalid entries keep a value no item reads
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = min(row0 + i * 16 + (lane & 15), M - 1);
        const float rm = ws.rmax[(size_t)pair * M + r], rs = ws.rsum[(size_t)pair * M + r];
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int c = min(w * 64 + j * 16 + (lane >> 4) * 4 + q, N - 1);
                const float x = acc[i][j][q] * itau;
                acc[i][j][q] = (expf(x - rm) / rs) * (expf(x - cm[c]) / cs[c]);
            }
    }
    __syncthreads();   // cm / cs are read: the P tile goes over them (and the rest of the dead operand tiles)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *reinterpret_cast<float4 *>(&pt[(i * 16 + (lane & 15)) * MT_PLD + w * 64 + j * 16 + (lane >> 4) * 4]) =
                make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
    __syncthreads();
    const int rows_here = min(MT_ROWS, M - row0);
    const int kl = min(k, rows_here * N);   // the strip's contribution: its k largest (all of them when it has fewer)
    // thread t walks column t of the tile (conflict-free LDS reads); the selection passes run over LDS, not over 64 registers
    // (sixteen LDS reads in flight per step: one read per item, waited for before it is looked at, made every walk of the
    // selection a chain of 64 LDS round trips -- 27 of the kernel's 73 us, scripts/debug/match_trace.py)
    auto tile_items = [&](auto &&f) {
        if (t < N)
            for (int r0 = 0; r0 < rows_here; r0 += 16) {
                unsigned u[16];
#pragma unroll
                for (int q = 0; q < 16; ++q) u[q] = __float_as_uint(pt[min(r0 + q, rows_here - 1) * MT_PLD + t]);
#pragma unroll
                for (int q = 0; q < 16; ++q)
                    if (r0 + q < rows_here) f(u[q], (row0 + r0 + q) * N + t);
            }
    };
    if (strip == 0) DPM_MT_STAMP(2);
    block_topk(tile_items, kl, sv, si, hist, sc, strips == 1);
    if (strip == 0) DPM_MT_STAMP(3);
    if (strips == 1) {
        for (int e = t; e < k; e += MT_T) out_v[(size_t)pair * k + e] = sv[e], out_i[(size_t)pair * k + e] = si[e];
        return;
    }
    // candidates of this strip -> memory; the pair's last strip to arrive merges them all
    const int kcap = min(k, MT_ROWS * N);   // slots per strip
    float *cv = ws.cand_v + ((size_t)pair * strips + strip) * kcap;
    int *ci = ws.cand_i + ((size_t)pair * strips + strip) * kcap;
    for (int e = t; e < kl; e += MT_T) cv[e] = sv[e], ci[e] = si[e];
    __syncthreads();
    if (t == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        sc[5] = (unsigned)__hip_atomic_fetch_add(&ws.ticket[pair], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if ((int)sc[5] != strips - 1) return;
    DPM_MT_STAMP(4);
    if (t == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    __syncthreads();
    const float *av = ws.cand_v + (size_t)pair * strips * kcap;
    const int *ai = ws.cand_i + (size_t)pair * strips * kcap;
    const int last_rows = M - (strips - 1) * MT_ROWS, kl_last = min(k, last_rows * N);
    const int total = (strips - 1) * kcap + kl_last;   // the lists lie back to back, only the last one is short
    if (total <= MT_MCAP) {
        // the candidates once through LDS (over the dead P tile), every load of a thread in flight together: the selection's
        // five walks over memory with agent-scope loads were most of this kernel's time
        float *mv = pt;
        int *mi = reinterpret_cast<int *>(pt + MT_MCAP);
        for (int e0 = 0; e0 < total; e0 += 8 * MT_T) {
            float v[8];
            int ix[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = min(e0 + q * MT_T + t, total - 1);
                v[q] = __hip_atomic_load(av + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ix[q] = __hip_atomic_load(ai + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int e = e0 + q * MT_T + t;
                if (e < total) mv[e] = v[q], mi[e] = ix[q];
            }
        }
        __syncthreads();
        DPM_MT_STAMP(5);
        auto lds_items = [&](auto &&f) {
            for (int e0 = t; e0 < total; e0 += 8 * MT_T) {
                unsigned u[8];
                int ix[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int e = min(e0 + q * MT_T, total - 1);
                    u[q] = __float_as_uint(mv[e]), ix[q] = mi[e];
                }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (e0 + q * MT_T < total) f(u[q], ix[q]);
            }
        };
        block_topk(lds_items, k, sv, si, hist, sc, true);
        DPM_MT_STAMP(6);
        for (int e = t; e < k; e += MT_T) out_v[(size_t)pair * k + e] = sv[e], out_i[(size_t)pair * k + e] = si[e];
        DPM_MT_STAMP(7);
        return;
    }
    auto mem_items = [&](auto &&f) {
        for (int s = 0; s < strips; ++s) {
            const int n_s = s == strips - 1 ? kl_last : kcap;
            for (int e = t; e < n_s; e += MT_T) {
                // candidates another workgroup wrote: agent-scope loads (this CU's L1 may hold stale lines of the slots)
                const float v = __hip_atomic_load(av + (size_t)s * kcap + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const int ix = __hip_atomic_load(ai + (size_t)s * kcap + e, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                f(__float_as_uint(v), ix);
            }
        }
    };
    __syncthreads();
    block_topk(mem_items, k, sv, si, hist, sc, true);
    for (int e = t; e < k; e += MT_T) out_v[(size_t)pair * k + e] = sv[e], out_i[(size_t)pair * k + e] = si[e];
}

size_t align256(size_t b) { return (b + 255) & ~(size_t)255; }

MatchWs carve(void *workspace, int batch, int M, int N, int k) {
    const int strips = (M + MT_ROWS - 1) / MT_ROWS;
    const size_t kcap = (size_t)std::min<long long>(k, (long long)MT_ROWS * N);
    char *p = (char *)align256((size_t)(uintptr_t)workspace);
    MatchWs w;
    w.rmax = (float *)p, p += align256(sizeof(float) * (size_t)batch * M);
    w.rsum = (float *)p, p += align256(sizeof(float) * (size_t)batch * M);
    w.pmax = (float *)p, p += align256(sizeof(float) * (size_t)batch * strips * N);
    w.psum = (float *)p, p += align256(sizeof(float) * (size_t)batch * strips * N);
    w.ticket = (int *)p, p += align256(sizeof(int) * (size_t)batch);
    w.cand_v = (float *)p, p += align256(sizeof(float) * (size_t)batch * strips * kcap);
    w.cand_i = (int *)p;
    return w;
}

}  // namespace

#ifdef DPM_EXPERIMENT
extern "C" int dpm_debug_match_trace(long long *host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dpm_match_trace_buf), sizeof(long long) * (size_t)std::min(n, 64));
}
#endif

extern "C" size_t dpm_match_workspace_bytes(int batch, int M, int N, int k) {
    if (batch < 1 || M < 1 || N < 1 || k < 1) return 0;
    const size_t strips = (size_t)(M + MT_ROWS - 1) / MT_ROWS;
    const size_t kcap = (size_t)std::min<long long>(k, (long long)MT_ROWS * N);
    return 256 + 2 * align256(sizeof(float) * (size_t)batch * M) + 2 * align256(sizeof(float) * (size_t)batch * strips * N) +
           align256(sizeof(int) * (size_t)batch) + 2 * align256(4 * (size_t)batch * strips * kcap);
}

extern "C" int dpm_match_topk(const float *a, const float *b, int batch, int M, int N, int C, double tau, int k, float *out_val,
                              int32_t *out_idx, void *workspace, dpm_stream_t stream) {
    DPM_CHECK_ARG(a && b && out_val && out_idx && workspace && batch >= 1 && M >= 1 && N >= 1 && C >= 1 && tau > 0.0);
    DPM_CHECK_ARG(k >= 1 && (long long)k <= (long long)M * N);
    const int strips = (M + MT_ROWS - 1) / MT_ROWS;
    if (N > MT_COLS || k > MT_KCAP || C % MT_KT != 0 || strips > 65535 || batch > 65535 || (long long)M * N > 0x7fffffffLL ||
        ((uintptr_t)a & 15) != 0 || ((uintptr_t)b & 15) != 0)
        return DPM_EUNSUPPORTED;
    const MatchWs ws = carve(workspace, batch, M, N, k);
    const float itau = 1.0f / (float)tau;  // torch divides by the scalar as a multiplication by 1/tau
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(match_stats_kernel, dim3(strips, batch), dim3(MT_T), 0, st, a, b, M, N, C, itau, ws);
    hipLaunchKernelGGL(match_topk_kernel, dim3(strips, batch), dim3(MT_T), 0, st, a, b, M, N, C, itau, k, ws, out_val, out_idx);
    return dpm_launch_status();
}
