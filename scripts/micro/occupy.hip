// Workgroups that hold a footprint (wave slots, LDS, optionally registers) and do nothing: scripts/step_model.py keeps them
// resident next to the feature / registration stages to price what the sampling kernels' RESIDENCY costs, apart from their work.
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(1024) void occupy(long long cycles, int lds_words) {
    extern __shared__ int pad[];
    if (lds_words && threadIdx.x == 0) pad[0] = 1;
    const long long t0 = wall_clock64();   // 100 MHz constant clock
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(64);
}
extern "C" int launch_occupy(int wgs, int threads, long long cycles, int lds_bytes, void *stream) {
    hipLaunchKernelGGL(occupy, dim3(wgs), dim3(threads), lds_bytes, (hipStream_t)stream, cycles, lds_bytes / 4);
    return (int)hipGetLastError();
}
