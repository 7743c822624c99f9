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

__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, __int_as_float(dpp_i<0xB1, 0xF>(__float_as_int(v))));   // quad_perm [1,0,3,2]
    v = fmaxf(v, __int_as_float(dpp_i<0x4E, 0xF>(__float_as_int(v))));   // quad_perm [2,3,0,1]
    v = fmaxf(v, __int_as_float(dpp_i<0x141, 0xF>(__float_as_int(v))));  // row_half_mirror
    v = fmaxf(v, __int_as_float(dpp_i<0x140, 0xF>(__float_as_int(v))));  // row_mirror
    v = fmaxf(v, __int_as_float(dpp_i<0x142, 0xA>(__float_as_int(v))));  // row_bcast:15 -> rows 1,3
    v = fmaxf(v, __int_as_float(dpp_i<0x143, 0xC>(__float_as_int(v))));  // row_bcast:31 -> rows 2,3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}
__device__ __forceinline__ int wave_min_dpp(int v) {
    v = min(v, dpp_i<0xB1, 0xF>(v));
    v = min(v, dpp_i<0x4E, 0xF>(v));
    v = min(v, dpp_i<0x141, 0xF>(v));
    v = min(v, dpp_i<0x140, 0xF>(v));
    v = min(v, dpp_i<0x142, 0xA>(v));
    v = min(v, dpp_i<0x143, 0xC>(v));
    return __builtin_amdgcn_readlane(v, 63);
}
