// Shared helpers for the gfx950 kernels of the DeepPointMap hot path.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/dpm_hip.h"

#define DPM_WAVE 64

#define DPM_CHECK_ARG(cond)            \
    do {                               \
        if (!(cond)) return DPM_EINVAL; \
    } while (0)

static inline int dpm_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? DPM_OK : (int)e;
}

static inline unsigned dpm_cdiv(long long a, long long b) { return (unsigned)((a + b - 1) / b); }

// Measurement switches (ablations that skip work, A/B layouts, "run it n more times" pricing).  The shipped library has
// NONE: dpm_knob() folds to its default and no entry point reads the environment.  Only a library built with
// -DDPM_EXPERIMENT (`csrc/build.py --out <lib> -DDPM_EXPERIMENT`, selected by scripts through DPM_LIB, which bench.py
// refuses without --allow-knobs) looks the name up, and dpm_version() of such a build carries DPM_VERSION_EXPERIMENT.
#ifdef DPM_EXPERIMENT
#include <stdlib.h>
static inline int dpm_knob(const char *name, int dflt) {
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
static inline constexpr int dpm_knob(const char *, int dflt) { return dflt; }
#endif

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Issue priority of the kernels that are bound by vector-ALU issue (neighbour searches, gather + LayerNorm kernels, the
// NN-1 search): inside the stream pipeline they share every SIMD with matrix-pipe kernels and the sampling chains.
// -DDPM_VALU_PRIO=n builds a library whose VALU-bound kernels run at wave priority n (A/B measurements; 0 = default).
#ifndef DPM_VALU_PRIO
#define DPM_VALU_PRIO 0
#endif
__device__ __forceinline__ void valu_bound_priority() {
    if (DPM_VALU_PRIO) __builtin_amdgcn_s_setprio(DPM_VALU_PRIO);
}

// Pacing of matrix instructions (experiment of round 4, profiles/r04_corun.md): a wave whose NEXT instruction is an MFMA
// while the matrix pipe is busy waits AT the SIMD's vector issue port and keeps every other wave's vector instructions
// out (scripts/micro/corun_pure.hip: a matrix-only and a vector-only wave on one SIMD take the SUM of their times; with
// the matrix wave idling after each MFMA the vector wave's work disappears in the gaps).  -DDPM_MFMA_PACE=n makes the
// dense kernels idle n wait states (4 cycles each) after every matrix instruction.  0 = shipped.
#ifndef DPM_MFMA_PACE
#define DPM_MFMA_PACE 0
#endif
__device__ __forceinline__ void mfma_pace() {
#if DPM_MFMA_PACE > 0
#pragma unroll
    for (int left = DPM_MFMA_PACE; left > 0; left -= 16) {
        if (left >= 16) asm volatile("s_nop 15");
        else asm volatile("s_nop %0" ::"n"((DPM_MFMA_PACE - 1) & 15));
    }
#endif
}

// ---- exact three-way bf16 split of an fp32 value (csrc/gemm_b3.hip says what it is for): x = hi + mid + lo, each term a
// truncation to the top 16 bits of a float, each remainder exact.  hi / mid come back masked, lo unmasked (its truncation
// happens where the term is stored: pack2 / a 16-bit store of the upper half).
// Finite values only: for x = +-Inf the first remainder is Inf - Inf = NaN, so a bf16x3 product turns an infinite operand into
// NaN where the fp32 kernels propagate the infinity (NaN stays NaN).  Guarding it costs a class test and a select per value on
// kernels that are bound by exactly these instructions; no layer of the path makes a non-finite activation from finite inputs
// (LayerNorm follows every product), and tests/test_gpu_ops.py::test_bf16x3_non_finite_inputs_give_nan_not_garbage pins the behaviour.
__device__ __forceinline__ void split3(float x, unsigned &hi, unsigned &mid, unsigned &lo) {
    hi = __float_as_uint(x) & 0xFFFF0000u;
    const float r1 = x - __uint_as_float(hi);
    mid = __float_as_uint(r1) & 0xFFFF0000u;
    const float r2 = r1 - __uint_as_float(mid);
    lo = __float_as_uint(r2);
}
// two such terms -> one dword holding [a | b] as consecutive bf16 (a at the lower address): the upper halves of both
// registers in one v_perm_b32
__device__ __forceinline__ unsigned pack2(unsigned a, unsigned b) { return __builtin_amdgcn_perm(b, a, 0x07060302u); }

// LDS image of a bf16 operand plane of the bf16x3 kernels: rows of 32 k = 64 bytes, NO padding, the four 16-byte chunks of
// a row (8 consecutive k: what one lane of v_mfma_f32_16x16x32_bf16 holds) stored at chunk ^ F(row), F = (-(row >> 2)) & 3.
// ds_read_b128 serves a fragment read in four groups of 16 lanes -- rows {0-3, 12-15} of k-chunk c with rows {4-11} of
// chunk c + 1, and vice versa (MI355X guide, LDS table) --: with this F the 16 lanes of every group hit 16 different 16-byte
// bank groups; the staging writes (8 or 16 bytes per lane, a row's lanes side by side) cover whole rows = contiguous bytes.
// The padded image of the first version (rows 80 bytes apart) spent 49 % of its LDS cycles in bank conflicts and kept the
// LDS pipe 74 % busy (PMC, profiles/r04_gemm_b3_pmc.md).  b3_col: element offset of k inside row `row`.
__device__ __forceinline__ int b3_col(int row, int k) { return ((((k >> 3) ^ (0 - (row >> 2))) & 3) << 3) | (k & 7); }

// XCD-aware workgroup order.  Workgroups are dealt round-robin to the 8 XCDs by linear id and every XCD has its own
// 4 MB L2; with the plain order all XCDs walk through all frames at once and every L2 holds a slice of everything.
// This maps the hardware id to a logical id such that XCD x processes the contiguous chunk [x*n/8, (x+1)*n/8) in
// order, i.e. whole frames / pairs stay on one XCD and their gather sources stay L2-resident.
__device__ __forceinline__ unsigned xcd_chunked_id(unsigned linear, unsigned total) {
    return (total & 7u) == 0u ? (linear & 7u) * (total >> 3) + (linear >> 3) : linear;
}

// ---- wave64 reductions on DPP (no LDS crossbar round trips).  After the call every lane holds the result.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, ROW_MASK, 0xF, false); }

// Each step is ONE instruction with a DPP operand (plus the two wait states a DPP read needs after the VALU write of
// its register): from update_dpp + fmaxf / min the compiler builds copy + s_nop + mov_dpp (+ a NaN canonicalisation
// for floats) + the operation.  Rows 1,3 then take row 0,2's last lane and rows 2,3 take lane 31's, so lane 63 holds
// the wave's result.  (v_max_f32 on non-NaN inputs returns one of its operands, exactly like fmaxf.)
#define DPM_WAVE_REDUCE(op)                                                                  \
    "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"           \
    "s_nop 1\n\t" op " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"           \
    "s_nop 1\n\t" op " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"               \
    "s_nop 1\n\t" op " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"                    \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"                  \
    "s_nop 1\n\t" op " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
#ifdef DPM_DPP_BUILTIN   // experimental builds (csrc/build.py --out): the same steps through update_dpp, hazards padded by the compiler
template <typename F>
__device__ __forceinline__ int wave_reduce_builtin(int v, F op) {
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x141, 0xF, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x140, 0xF, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x142, 0xA, 0xF, false));
    v = op(v, __builtin_amdgcn_update_dpp(v, v, 0x143, 0xC, 0xF, false));
    return __builtin_amdgcn_readlane(v, 63);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    return __int_as_float(wave_reduce_builtin(__float_as_int(v), [](int a, int b) { return __float_as_int(fmaxf(__int_as_float(a), __int_as_float(b))); }));
}
__device__ __forceinline__ int wave_min_dpp(int v) {
    return wave_reduce_builtin(v, [](int a, int b) { return min(a, b); });
}
#else
__device__ __forceinline__ float wave_max_dpp(float v) {
    asm(DPM_WAVE_REDUCE("v_max_f32_dpp") : "+v"(v));
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_dpp(int v) {
    asm(DPM_WAVE_REDUCE("v_min_i32_dpp") : "+v"(v));
    return __builtin_amdgcn_readlane(v, 63);
}
#endif
#undef DPM_WAVE_REDUCE
