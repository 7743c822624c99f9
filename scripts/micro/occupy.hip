// Workgroups that hold a footprint (wave slots, LDS) and do a CHOSEN part of what the sampling rounds do, once per `period`:
// scripts/step_model.py keeps them resident next to the feature / registration stages to find out WHICH activity of the sampling
// kernel costs the other stages time (its residency alone does not: mode 0).
//   mode 0: sleep.  bit 0: ~40 dependent vector instructions per wave.  bit 1: one wave-wide 16-byte-per-lane load from a random
//   1 KB block of `buf` (+ a 4-byte-per-lane load and store 256 B wide), as a bucket update does.  bit 2: the exchange -- LDS write,
//   workgroup barrier, two dependent LDS reads.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(1024) void occupy(long long cycles, int lds_words, int mode, int period, float4 *buf, unsigned nblk, float *sink) {
    extern __shared__ float pad[];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    if (lds_words && t == 0) pad[0] = 1.f;
    const long long t0 = wall_clock64();   // 100 MHz constant clock
    unsigned rng = blockIdx.x * 9781u + w * 6271u + 1u;
    float acc = (float)t;
    long long next = t0;
    while (wall_clock64() - t0 < cycles) {
        if (mode == 0) { __builtin_amdgcn_s_sleep(64); continue; }
        while (wall_clock64() < next) __builtin_amdgcn_s_sleep(8);
        next += period;
        if (mode & 2) {
            rng = rng * 1664525u + 1013904223u;
            const size_t blk = (size_t)(rng >> 8) % nblk;           // wave-uniform: one 1 KB block
            const float4 p = buf[blk * 64 + lane];
            float *c = (float *)(buf + (size_t)nblk * 64) + blk * 64 + lane;
            const float d = p.x * p.x + p.y * p.y + p.z * p.z;
            if (d < *c + acc) *c = d;
            acc += d * 1e-30f;
        }
        if (mode & 1) {
#pragma unroll
            for (int i = 0; i < 40; ++i) acc = acc * 1.0000001f + 0.5f;
        }
        if (mode & 4) {
            if (lane == 0) pad[16 + w] = acc;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const float a = pad[16 + (lane & 15)];
            const int k = ((int)a) & 15;
            acc += pad[16 + k] * 1e-30f;
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
extern "C" int launch_occupy(int wgs, int threads, long long cycles, int lds_bytes, void *stream) {
    hipLaunchKernelGGL(occupy, dim3(wgs), dim3(threads), lds_bytes, (hipStream_t)stream, cycles, lds_bytes / 4, 0, 0, nullptr, 1u, nullptr);
    return (int)hipGetLastError();
}
extern "C" int launch_active(int wgs, int threads, long long cycles, int lds_bytes, int mode, int period, void *buf, unsigned nblk, void *sink, void *stream) {
    hipLaunchKernelGGL(occupy, dim3(wgs), dim3(threads), lds_bytes, (hipStream_t)stream, cycles, lds_bytes / 4, mode, period, (float4 *)buf, nblk, (float *)sink);
    return (int)hipGetLastError();
}
