// kernarg_preload.hip -- what does reading the kernel arguments cost at the start of a decode launch, and does the gfx950
// kernarg preload (the first <= 16 argument dwords delivered in SGPRs at wave launch; hipcc -mllvm
// -amdgpu-kernarg-preload-count=16, scalar arguments only -- not a by-value struct) remove it?  (not part of the product)
// A chain of dependent launches of a 512 x 512 grid; every wave stamps s_memrealtime at entry and again when it has ISSUED a
// global load whose address comes from the arguments; printed: median / max over blocks of that gap, and the time per launch.
// Measured (MI355X, ROCm 7.2): the gap drops from ~600-660 to ~200 clocks for the scalar-argument kernel, but EVERY launch of the
// binary built with the option takes 4.1-4.4 us in a dependent chain instead of 3.05-3.4 -- the dispatch gets slower by more
// than the waves save.  Not used.
// Build twice: hipcc --offload-arch=gfx950 -O3 tools/kernarg_preload.hip -o tools/bin/kernarg_preload [-mllvm -amdgpu-kernarg-preload-count=16]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct P { const uint32_t* x; uint32_t* y; unsigned long long* tr; int n, a, b, c, d, e, f, g; const uint32_t* more[8]; };   // 120 bytes, like a trimmed GemvParams

__device__ __forceinline__ unsigned long long now() { return __builtin_readcyclecounter(); }

__global__ __launch_bounds__(512) void k_struct(P p) {
    const unsigned long long t0 = now();
    const uint32_t v = p.x[(blockIdx.x * 512 + threadIdx.x) % p.n];
    asm volatile("" ::: "memory");
    const unsigned long long t1 = now();
    if (threadIdx.x == 0) { p.tr[blockIdx.x * 2] = t1 - t0; }
    if (v == 0x12345u) p.y[0] = v + p.a + p.b + p.c + p.d + p.e + p.f + p.g;
}
__global__ __launch_bounds__(512) void k_scalar(const uint32_t* x, uint32_t* y, unsigned long long* tr, int n, int a, int b, int c, int d, int e, int f, int g, P rest) {
    const unsigned long long t0 = now();
    const uint32_t v = x[(blockIdx.x * 512 + threadIdx.x) % n];
    asm volatile("" ::: "memory");
    const unsigned long long t1 = now();
    if (threadIdx.x == 0) { tr[blockIdx.x * 2] = t1 - t0; }
    if (v == 0x12345u) y[0] = v + a + b + c + d + e + f + g + rest.a;
}

int main() {
    uint32_t *x, *y;
    unsigned long long* tr;
    const int n = 1 << 20;
    CHECK(hipMalloc(&x, n * 4)); CHECK(hipMemset(x, 0, n * 4));
    CHECK(hipMalloc(&y, 4));
    CHECK(hipMalloc(&tr, 512 * 2 * 8));
    P p = {x, y, tr, n, 1, 2, 3, 4, 5, 6, 7, {x, x, x, x, x, x, x, x}};
    for (int kind = 0; kind < 2; ++kind) {
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int L = 200;
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < L; ++i) {
                if (kind == 0) hipLaunchKernelGGL(k_struct, dim3(512), dim3(512), 0, 0, p);
                else hipLaunchKernelGGL(k_scalar, dim3(512), dim3(512), 0, 0, (const uint32_t*)x, y, tr, n, 1, 2, 3, 4, 5, 6, 7, p);
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
        }
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        std::vector<unsigned long long> h(1024);
        CHECK(hipMemcpy(h.data(), tr, 1024 * 8, hipMemcpyDeviceToHost));
        std::vector<unsigned long long> g;
        for (int b = 0; b < 512; ++b) g.push_back(h[b * 2]);
        std::sort(g.begin(), g.end());
        printf("%-28s %.2f us per launch; entry -> first argument-dependent load issued: min %llu median %llu max %llu clocks\n",
               kind == 0 ? "by-value struct:" : "scalar arguments (+struct):", ms * 1e3 / L, g[0], g[256], g[511]);
    }
    return 0;
}
