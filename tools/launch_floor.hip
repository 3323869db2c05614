// launch_floor.hip -- tuning aid (not part of the product): what one more kernel in a dependent same-stream chain costs on
// this chip, i.e. the floor under the ~4 us fixed cost per decode launch discussed in DESIGN.md.
//   empty      1 block x 64 threads, no work
//   wide       512 blocks x 512 threads (the decode GEMV grid), no work
//   ramp       512 x 512, every wave issues the 16 16-byte loads of a GEMV ring (1 KiB each) and waits for them: kernel
//              boundary + launch of 4096 waves + one full HBM round trip, no streaming
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/launch_floor.hip -o tools/bin/launch_floor
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__global__ void k_empty(int* p) { if (p && threadIdx.x == 12345) *p = 1; }
__global__ __launch_bounds__(512) void k_ramp(const u32x4* __restrict__ w, uint32_t* out, size_t stride) {
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    const u32x4* base = w + wave * stride + (threadIdx.x & 63);
    u32x4 acc = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int u = 0; u < 16; ++u) {
        const u32x4 v = __builtin_nontemporal_load(base + u * 64);
        acc[0] ^= v[0]; acc[1] ^= v[1]; acc[2] ^= v[2]; acc[3] ^= v[3];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;   // never true for the fill pattern; keeps the loads
}

// rolling variant: every wave streams `rounds` consecutive 16-load ring-fulls (the next 16 loads are issued as the previous
// ones are consumed), like a GEMV wave that owns `rounds` x 16 KiB of weights
__global__ __launch_bounds__(512) void k_stream(const u32x4* __restrict__ w, uint32_t* out, size_t stride, int rounds) {
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    const u32x4* base = w + wave * stride + (threadIdx.x & 63);
    u32x4 r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = __builtin_nontemporal_load(base + u * 64);
    u32x4 acc = {0u, 0u, 0u, 0u};
    for (int it = 1; it <= rounds; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            acc[0] ^= r[u][0]; acc[1] ^= r[u][1]; acc[2] ^= r[u][2]; acc[3] ^= r[u][3];
            if (it < rounds) r[u] = __builtin_nontemporal_load(base + (size_t)(it * 16 + u) * 64);
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[0] = 1;
}

// the same stream, consumed the way the GEMV consumes it: one 16-byte x chunk from LDS per pair of weight loads, 8 v_dot2c
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
__device__ __forceinline__ float dot2(uint32_t a, uint32_t b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
__global__ __launch_bounds__(512) void k_stream_dot(const u32x4* __restrict__ w, float* out, size_t stride, int rounds) {
    __shared__ u32x4 xs[2048];
    for (int i = threadIdx.x; i < 2048; i += 512) xs[i] = (u32x4){0x3f803f80u + i, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u};
    const size_t wave = (size_t)blockIdx.x * 8 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const u32x4* base = w + wave * stride + lane;
    u32x4 r[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) r[u] = __builtin_nontemporal_load(base + u * 64);
    __syncthreads();
    float a0 = 0.f, a1 = 0.f;
    for (int it = 1; it <= rounds; ++it) {
#pragma unroll
        for (int u = 0; u < 16; u += 2) {
            const u32x4 x = xs[((it * 8 + (u >> 1)) * 64 + lane) & 2047];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a0 = dot2(r[u][j], x[j], a0);
                a1 = dot2(r[u + 1][j], x[j], a1);
            }
            if (it < rounds) {
                r[u] = __builtin_nontemporal_load(base + (size_t)(it * 16 + u) * 64);
                r[u + 1] = __builtin_nontemporal_load(base + (size_t)(it * 16 + u + 1) * 64);
            }
        }
    }
    if (a0 + a1 == 12345.678f) out[0] = a0;
}

template <class F>
static double time_chain(F launch, int n) {
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int i = 0; i < 50; ++i) launch(i);
    CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < n; ++i) launch(i);
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, e0, e1));
    return ms * 1e3 / n;
}

int main() {
    int* flag;
    CHECK(hipMalloc(&flag, 4));
    const size_t per_launch = (size_t)4096 * 16 * 1024;            // 64 MiB touched per ramp launch
    const int nbuf = 24;                                            // rotate through 1.5 GiB so nothing is cache resident
    u32x4* w;
    CHECK(hipMalloc(&w, per_launch * nbuf));
    CHECK(hipMemset(w, 0x5a, per_launch * nbuf));
    uint32_t* out;
    CHECK(hipMalloc(&out, 4));
    const int N = 2000;
    printf("empty (1 x 64)        %.2f us per launch\n", time_chain([&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, (int*)nullptr); }, N));
    printf("wide  (512 x 512)     %.2f us per launch\n", time_chain([&](int) { hipLaunchKernelGGL(k_empty, dim3(512), dim3(512), 0, 0, (int*)nullptr); }, N));
    printf("ramp  (512 x 512, 16 KiB per wave = 64 MiB per launch, HBM)  %.2f us per launch  (64 MiB at 6.3 TB/s alone = %.1f us)\n",
           time_chain([&](int i) { hipLaunchKernelGGL(k_ramp, dim3(512), dim3(512), 0, 0, w + (size_t)(i % nbuf) * (per_launch / 16), out, (size_t)16 * 64); }, N),
           per_launch / 6.3e6);
    // the same bytes per launch (~90 MB = the down projection) from 4096 waves x 22 KiB vs 2048 waves x 44 KiB
    for (int cfg = 0; cfg < 4; ++cfg) {
        const int grid = (cfg & 1) ? 256 : 512, rounds = (cfg & 2) ? ((cfg & 1) ? 6 : 3) : ((cfg & 1) ? 3 : 2);
        const size_t bytes = (size_t)grid * 8 * rounds * 16 * 1024;
        const int nb = (int)((per_launch * nbuf) / bytes);
        const double us = time_chain([&](int i) { hipLaunchKernelGGL(k_stream, dim3(grid), dim3(512), 0, 0, w + (size_t)(i % nb) * (bytes / 16), out, (size_t)rounds * 16 * 64, rounds); }, N);
        printf("stream %3d blocks x 8 waves x %d ring-fulls = %6.1f MB per launch: %.2f us  %.2f TB/s\n", grid, rounds, bytes / 1e6, us, bytes / us / 1e6);
    }
    for (int cfg = 0; cfg < 2; ++cfg) {
        const int grid = cfg ? 256 : 512, rounds = cfg ? 3 : 3;
        const size_t bytes = (size_t)grid * 8 * rounds * 16 * 1024;
        const int nb = (int)((per_launch * nbuf) / bytes);
        const double us = time_chain([&](int i) { hipLaunchKernelGGL(k_stream_dot, dim3(grid), dim3(512), 0, 0, w + (size_t)(i % nb) * (bytes / 16), (float*)out, (size_t)rounds * 16 * 64, rounds); }, N);
        printf("stream+dot2 %3d blocks x 8 waves x %d ring-fulls = %6.1f MB per launch: %.2f us  %.2f TB/s\n", grid, rounds, bytes / 1e6, us, bytes / us / 1e6);
    }
    return 0;
}
