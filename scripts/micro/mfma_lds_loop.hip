// Which part of the GEMM's K-tile loop costs the matrix pipe its time?  The 64 x 64 kernel's inner structure (4 waves as
// 2 x 2, wave tile 32 x 32 = 2 x 2 MFMA blocks, K-tile 32, row stride 34) rebuilt piece by piece on static LDS contents:
//   0: MFMAs only (registers)                       1: + the ds_read_b32 fragment reads (double-buffered over kk)
//   2: + the two barriers per K-tile                3: + the register -> LDS staging writes (8 x ds_write_b64 per thread)
//   4: like 3 with b128 fragment reads (k-remap, row stride 36) and b128 staging writes
//   5: like 3 plus the global prefetch loads of the next K-tile (L2-resident operands)
//   6: like 5 plus the kernel's epilogue: tile staged through LDS, whole 256-byte rows stored (100 MB per launch)
//   7: like 5 plus an epilogue that stores each lane's float4 straight from the accumulators
//   8: like 6 with nontemporal stores
// hipcc --offload-arch=gfx950 -O3 scripts/micro/mfma_lds_loop.hip -o /tmp/mll && /tmp/mll
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;

template <int V>
__global__ __launch_bounds__(256) void loop(const float *__restrict__ src, float *out, int ktiles) {
    constexpr int LD = V == 4 ? 36 : 34;
    __shared__ __attribute__((aligned(16))) float smem[128 * LD];
    float (*Xs)[LD] = reinterpret_cast<float (*)[LD]>(smem);
    float (*Ws)[LD] = reinterpret_cast<float (*)[LD]>(smem + 64 * LD);
    const int t = threadIdx.x, lane = t & 63, w = t >> 6, wm = w >> 1, wn = w & 1;
    for (int i = t; i < 128 * LD; i += 256) smem[i] = (float)(i % 7) * 0.125f;
    __syncthreads();
    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int sr = t >> 3, sk = (t & 7) * 4;
    const float *gp = src + (size_t)(blockIdx.x % 512) * 64 * 256 + sr * 256 + sk;
    float4 xr[2], wr[2];
#pragma unroll
    for (int p = 0; p < 2; ++p) xr[p] = make_float4(1.f + t, 2.f, 3.f, 4.f), wr[p] = make_float4(0.5f, 0.25f, t, 1.f);
    float ra = 1.f + lane, rb = 2.f;
    for (int kt = 0; kt < ktiles; ++kt) {
        if (V >= 3) {
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                if (V == 4) {
                    *reinterpret_cast<float4 *>(&Xs[p * 32 + sr][sk]) = xr[p];
                    *reinterpret_cast<float4 *>(&Ws[p * 32 + sr][sk]) = wr[p];
                } else {
                    float2 *d = reinterpret_cast<float2 *>(&Xs[p * 32 + sr][sk]);
                    d[0] = make_float2(xr[p].x, xr[p].y), d[1] = make_float2(xr[p].z, xr[p].w);
                    float2 *e = reinterpret_cast<float2 *>(&Ws[p * 32 + sr][sk]);
                    e[0] = make_float2(wr[p].x, wr[p].y), e[1] = make_float2(wr[p].z, wr[p].w);
                }
            }
        }
        if (V >= 2) __syncthreads();
        if (V >= 5) {
            const int k0 = (kt & 7) * 32;
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                xr[p] = *reinterpret_cast<const float4 *>(gp + p * 32 * 256 + k0);
                wr[p] = *reinterpret_cast<const float4 *>(gp + 8 * 1024 * 1024 + p * 32 * 256 + k0);
            }
        }
        if (V == 0) {
#pragma unroll
            for (int kk = 0; kk < 32; kk += 4)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(rb, ra, acc[i][j], 0, 0, 0);
        } else if (V == 4) {
            f32x4 a4[2][2], b4[2][2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < 2; ++i) a4[h][i] = *reinterpret_cast<const f32x4 *>(&Xs[wm * 32 + i * 16 + (lane & 15)][h * 16 + (lane >> 4) * 4]);
#pragma unroll
                for (int j = 0; j < 2; ++j) b4[h][j] = *reinterpret_cast<const f32x4 *>(&Ws[wn * 32 + j * 16 + (lane & 15)][h * 16 + (lane >> 4) * 4]);
            }
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b4[h][j][s], a4[h][i][s], acc[i][j], 0, 0, 0);
        } else {
            float a[2][2], b[2][2];
#pragma unroll
            for (int i = 0; i < 2; ++i) a[0][i] = Xs[wm * 32 + i * 16 + (lane & 15)][lane >> 4];
#pragma unroll
            for (int j = 0; j < 2; ++j) b[0][j] = Ws[wn * 32 + j * 16 + (lane & 15)][lane >> 4];
#pragma unroll
            for (int kk = 0; kk < 32; kk += 4) {
                const int cur = (kk >> 2) & 1, nxt = cur ^ 1;
                if (kk + 4 < 32) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) a[nxt][i] = Xs[wm * 32 + i * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
#pragma unroll
                    for (int j = 0; j < 2; ++j) b[nxt][j] = Ws[wn * 32 + j * 16 + (lane & 15)][kk + 4 + (lane >> 4)];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[cur][j], a[cur][i], acc[i][j], 0, 0, 0);
            }
        }
        if (V >= 2) __syncthreads();
    }
    if (V >= 6) {
        // output tile (by, bx) of a 32768 x 768 matrix, ktiles == 8 launches only
        const int by = blockIdx.x / 12, bx = blockIdx.x % 12;
        float *o = out + 64 + (size_t)by * 64 * 768 + bx * 64;
        if (V == 7) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    *reinterpret_cast<float4 *>(o + (size_t)(wm * 32 + i * 16 + (lane & 15)) * 768 + wn * 32 + j * 16 + (lane >> 4) * 4) =
                        make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
            return;
        }
        constexpr int LDC = 68;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
                *reinterpret_cast<float4 *>(&smem[(wm * 32 + i * 16 + (lane & 15)) * LDC + wn * 32 + j * 16 + (lane >> 4) * 4]) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        __syncthreads();
        const int cr = t >> 4, cc = (t & 15) * 4;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            float4 v = *reinterpret_cast<const float4 *>(&smem[(p * 16 + cr) * LDC + cc]);
            v.x += 1.f, v.y += 1.f, v.z += 1.f, v.w += 1.f;
            float4 *dst = reinterpret_cast<float4 *>(o + (size_t)(p * 16 + cr) * 768 + cc);
            if (V == 8) {
                __builtin_nontemporal_store(v.x, &dst->x), __builtin_nontemporal_store(v.y, &dst->y);
                __builtin_nontemporal_store(v.z, &dst->z), __builtin_nontemporal_store(v.w, &dst->w);
            } else {
                *dst = v;
            }
        }
        return;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    if (s == 12345.678f) out[0] = s + xr[0].x + wr[1].y;
}

template <int V>
void run(const float *src, float *out, hipEvent_t e0, hipEvent_t e1, const char *what) {
    for (int ktiles : {8, 64}) {
        if (V >= 6 && ktiles != 8) continue;
        const int grid = ktiles == 8 ? 6144 : 2048;  // the 32768 x 256 -> 768 projection has 6144 tiles of 8 K-tiles
        float best = 1e9f;
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(loop<V>, dim3(grid), dim3(256), 0, 0, src, out, ktiles);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms;
            (void)hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        const double flop = (double)grid * ktiles * 4 * 32 * 2048.0;
        printf("%-58s %2d K-tiles x %d tiles: %7.1f us  %6.1f TFLOP/s\n", what, ktiles, grid, best * 1e3, flop / (best * 1e-3) / 1e12);
    }
}

int main() {
    float *out, *src;
    (void)hipMalloc(&out, 64 * 4 + (size_t)32768 * 768 * 4);
    (void)hipMalloc(&src, 80u << 20);
    (void)hipMemset(src, 0, 80u << 20);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
    run<0>(src, out, e0, e1, "0 MFMA only");
    run<1>(src, out, e0, e1, "1 + ds_read_b32 fragments");
    run<2>(src, out, e0, e1, "2 + two barriers per K-tile");
    run<3>(src, out, e0, e1, "3 + staging writes (ds_write_b64)");
    run<4>(src, out, e0, e1, "4 = 3 with b128 fragment reads and writes (k-remap)");
    run<5>(src, out, e0, e1, "5 = 3 + global prefetch loads");
    run<6>(src, out, e0, e1, "6 = 5 + epilogue through LDS, whole-row stores");
    run<7>(src, out, e0, e1, "7 = 5 + epilogue straight from the accumulators");
    run<8>(src, out, e0, e1, "8 = 6 with nontemporal stores");
    return 0;
}
