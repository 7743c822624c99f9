// Micro-benchmark behind DESIGN.md's latency-mode decision: what does ONE winner exchange between G workgroups cost when it
// goes through L2 (the only memory G compute units share)?  A cooperative single-frame FPS pays this once per pick.
//   variant A: every workgroup publishes (round tag, candidate) in its own slot (one 64-bit store), lanes 0..G-1 of every
//              workgroup poll the G slots until all carry the round's tag;
//   variant B: atomicMax on a shared 64-bit key + atomicAdd on an arrival counter, polled.
// G workgroups on ONE XCD (workgroups are dealt round-robin to the 8 XCDs: only blockIdx % 8 == 0 take part) or spread.
// build + run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/micro/l2_exchange.hip -o /tmp/l2x && /tmp/l2x
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void exchange_slots(unsigned long long *slots, int G, int rounds, int same_xcd, unsigned long long *sink) {
    int g = blockIdx.x;
    if (same_xcd) {
        if (g & 7) return;
        g >>= 3;
    }
    if (g >= G) return;
    const int lane = threadIdx.x;
    unsigned long long acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (lane == 0)
            __hip_atomic_store(&slots[g * 16], ((unsigned long long)r << 32) | (unsigned)(g * 977 + r), __ATOMIC_RELEASE,
                               __HIP_MEMORY_SCOPE_AGENT);
        unsigned long long v = 0;
        bool done;
        do {
            if (lane < G) v = __hip_atomic_load(&slots[lane * 16], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT);
            done = __all(lane >= G || (v >> 32) >= (unsigned long long)r);
        } while (!done);
        acc += v;
        __syncthreads();
    }
    if (lane == 0) sink[g] = acc;
}

__global__ void exchange_atomics(unsigned long long *key, unsigned *count, int G, int rounds, int same_xcd, unsigned long long *sink) {
    int g = blockIdx.x;
    if (same_xcd) {
        if (g & 7) return;
        g >>= 3;
    }
    if (g >= G) return;
    const int lane = threadIdx.x;
    unsigned long long acc = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (lane == 0) {
            atomicMax(&key[(r & 1) * 16], ((unsigned long long)r << 32) | (unsigned)(g * 977 + r));
            __hip_atomic_fetch_add(count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)(G * r)) {}
            acc += __hip_atomic_load(&key[(r & 1) * 16], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
    }
    if (lane == 0) sink[g] = acc;
}

int main() {
    unsigned long long *buf, *sink;
    hipMalloc(&buf, 1 << 16);
    hipMalloc(&sink, 1 << 12);
    hipEvent_t e0, e1;
    hipEventCreate(&e0), hipEventCreate(&e1);
    const int rounds = 4095;
    for (int same = 1; same >= 0; --same)
        for (int variant = 0; variant < 2; ++variant)
            for (int G : {1, 2, 4, 8, 16}) {
                float best = 1e9f;
                for (int rep = 0; rep < 4; ++rep) {
                    hipMemset(buf, 0, 1 << 16);
                    hipDeviceSynchronize();
                    hipEventRecord(e0);
                    const int grid = same ? G * 8 : G;
                    if (variant == 0)
                        hipLaunchKernelGGL(exchange_slots, dim3(grid), dim3(64), 0, 0, buf, G, rounds, same, sink);
                    else
                        hipLaunchKernelGGL(exchange_atomics, dim3(grid), dim3(64), 0, 0, buf, (unsigned *)(buf + 1024), G, rounds, same, sink);
                    hipEventRecord(e1);
                    hipEventSynchronize(e1);
                    float ms;
                    hipEventElapsedTime(&ms, e0, e1);
                    best = ms < best ? ms : best;
                }
                printf("%s  %-28s G=%2d: %.3f us per exchange\n", same ? "one XCD " : "spread  ",
                       variant == 0 ? "slots (store + polled loads)" : "atomicMax + arrival counter", G, best * 1e3f / rounds);
            }
    return 0;
}
