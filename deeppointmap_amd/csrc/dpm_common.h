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
