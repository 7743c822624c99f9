// Wave-level helpers shared by the farthest-point-sampling kernels (fps.hip, fps_tree.hip).
#pragma once
#include "dpm_common.h"

#pragma clang fp contract(off)

namespace {

struct Best {
    float v;
    int i;
};

__device__ __forceinline__ float sqdist(float sx, float sy, float sz, float x, float y, float z) {
    const float dx = sx - x, dy = sy - y, dz = sz - z;
    return (dx * dx + dy * dy) + dz * dz;
}


// Workgroup barrier that orders LDS traffic only.  __syncthreads() also drains vmcnt, i.e. waits for the round
// trip of every outstanding global store; in the sampling loops global data is wave-private (or written once and
// read after the kernel), so only the LDS exchange needs ordering.
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Squared distances are >= +0 and the "must not win" sentinel is -1, and on that domain the order of the floats
// is the order of their bit patterns as signed integers.  Integer max needs no NaN canonicalisation, so each
// reduction step is ONE v_max_i32 with a DPP operand instead of mov_dpp + two v_max_f32.
// The instruction is written out: from update_dpp the compiler builds copy + s_nop + mov_dpp + max per step.
// s_nop 1 = the two wait states a DPP operand needs after the VALU write of its register (each step reads the
// previous step's result; the first one covers whatever produced v).
#define DPM_IMAX_STEP(ctrl) "s_nop 1\n\tv_max_i32_dpp %0, %0, %0 " ctrl " bank_mask:0xf\n\t"
#define DPM_IMAX_ROW                                                                              \
    DPM_IMAX_STEP("quad_perm:[1,0,3,2] row_mask:0xf") DPM_IMAX_STEP("quad_perm:[2,3,0,1] row_mask:0xf") \
    DPM_IMAX_STEP("row_half_mirror row_mask:0xf") DPM_IMAX_STEP("row_mirror row_mask:0xf")
__device__ __forceinline__ int imax_dpp_row(int v) {
    asm(DPM_IMAX_ROW "s_nop 0" : "+v"(v));
    return v;
}
// max / min over the 16 lanes of a DPP row (lanes sharing lane>>4); every lane of the row gets the result
__device__ __forceinline__ float row16_max_f(float v) { return __int_as_float(imax_dpp_row(__float_as_int(v))); }
// wave-wide max of such values (wave-uniform result): rows 1,3 take row 0,2's last lane, rows 2,3 take lane 31's
__device__ __forceinline__ float wave_max_ordered(float v) {
    int i = __float_as_int(v);
    asm(DPM_IMAX_ROW DPM_IMAX_STEP("row_bcast:15 row_mask:0xa") DPM_IMAX_STEP("row_bcast:31 row_mask:0xc") "s_nop 0"
        : "+v"(i));
    return __int_as_float(__builtin_amdgcn_readlane(i, 63));
}
#undef DPM_IMAX_ROW
#undef DPM_IMAX_STEP
__device__ __forceinline__ int row16_min_i(int v) {
    v = min(v, dpp_i<0xB1, 0xF>(v));
    v = min(v, dpp_i<0x4E, 0xF>(v));
    v = min(v, dpp_i<0x141, 0xF>(v));
    v = min(v, dpp_i<0x140, 0xF>(v));
    return v;
}

#ifdef DPM_FPS_STATS
#define FPS_T(i) do { const long long _n = clock64(); tacc[i] += _n - tprev; tprev = _n; } while (0)
#else
#define FPS_T(i) do { } while (0)
#endif

// lane holding the wave's best (largest v; among equal v the smallest idx).  Lanes that must not win pass v < 0.
__device__ __forceinline__ int wave_argbest(float v, int idx, float &vmax) {
    vmax = wave_max_ordered(v);
    unsigned long long eq = __ballot(v == vmax);
    if (__popcll(eq) > 1) {  // ties are rare: break them by the smallest original index
        const int imin = wave_min_dpp((v == vmax) ? idx : 0x7fffffff);
        eq = __ballot(v == vmax && idx == imin);
    }
    return __builtin_ctzll(eq);
}
__device__ __forceinline__ float lane_f(float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ int lane_i(int v, int l) { return __builtin_amdgcn_readlane(v, l); }


}  // namespace
