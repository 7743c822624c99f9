// Three chip-filling "victim" kernels with one bound each, timed next to the real first-level sampling (scripts/fps_interference.py):
// which shared resource does the sampling kernel take from the others?
#include <hip/hip_runtime.h>
extern "C" __global__ __launch_bounds__(256) void victim_alu(float *out, int iters) {   // vector ALU only
    float a = threadIdx.x * 1e-3f, b = blockIdx.x * 1e-3f, c = 1.0001f, d = 0.5f;
    for (int i = 0; i < iters; ++i) {
        a = fmaf(a, c, d), b = fmaf(b, c, d), a = fmaf(a, c, b), b = fmaf(b, c, a);
        a = fmaf(a, c, d), b = fmaf(b, c, d), a = fmaf(a, c, b), b = fmaf(b, c, a);
    }
    if (a + b == 12345.678f) out[0] = a;
}
extern "C" __global__ __launch_bounds__(256) void victim_stream(const float4 *__restrict__ in, float4 *__restrict__ out, size_t n) {  // HBM stream
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        float4 v = in[i];
        v.x += 1.f;
        out[i] = v;
    }
}
extern "C" __global__ __launch_bounds__(256) void victim_gather(const float4 *__restrict__ tab, const int *__restrict__ idx, float *out, int n, int reps) {  // L2-resident random gather
    float acc = 0.f;
    const int t = blockIdx.x * 256 + threadIdx.x;
    for (int r = 0; r < reps; ++r) {
        const int j = idx[(t * 17 + r * 9973) % n];
        const float4 v = tab[j];
        acc += v.x + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;
}
extern "C" __global__ __launch_bounds__(256) void victim_lds(float *out, int iters) {   // LDS traffic + workgroup barriers
    __shared__ float s[4096];
    const int t = threadIdx.x;
    float acc = 0.f;
    for (int i = 0; i < iters; ++i) {
        s[(t * 33 + i) & 4095] = acc + t;
        __syncthreads();
        acc += s[(t * 7 + i * 5) & 4095];
        __syncthreads();
    }
    if (acc == 12345.678f) out[0] = acc;
}
#define L(name, grid, ...) hipLaunchKernelGGL(name, dim3(grid), dim3(256), 0, (hipStream_t)stream, __VA_ARGS__); return (int)hipGetLastError();
extern "C" int run_alu(int grid, float *out, int iters, void *stream) { L(victim_alu, grid, out, iters) }
extern "C" int run_stream(int grid, const void *in, void *out, size_t n, void *stream) { L(victim_stream, grid, (const float4 *)in, (float4 *)out, n) }
extern "C" int run_gather(int grid, const void *tab, const int *idx, float *out, int n, int reps, void *stream) { L(victim_gather, grid, (const float4 *)tab, idx, out, n, reps) }
extern "C" int run_lds(int grid, float *out, int iters, void *stream) { L(victim_lds, grid, out, iters) }
