// xpath_probe.hip -- how long does a block wait for a small activation vector while the chip floods HBM with weight requests, and
// does the SCALAR path (s_load through the scalar cache) get it sooner than a vector load issued first (not part of the product)?
// 512 blocks x 8 waves; every wave requests 16 KiB of "weights" (16 non-temporal 16-byte loads per lane, as the decode GEMV's head);
// wave 0 of every block ALSO gets 8 KB of x, written just before by another kernel (so it is not in this XCD's L2):
//   mode 0: vector loads of x AFTER the head (waited for with the head: in-order return)
//   mode 1: vector loads of x BEFORE the head, counted wait (vmcnt(16))
//   mode 2: s_load_dwordx16 x 8 (512 B per wave: every wave takes its 1/16 of x) before the head, s_waitcnt lgkmcnt(0)
// Prints min / median / max over blocks of (x arrived - block entry) in us.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/xpath_probe.hip -o tools/bin/xpath_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <algorithm>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(16))) uint32_t u32x16;

__device__ __forceinline__ unsigned long long wall() { return __builtin_readcyclecounter() * 0 + __builtin_amdgcn_s_memrealtime(); }

__global__ void write_x(uint32_t* x, uint32_t v) { x[blockIdx.x * 256 + threadIdx.x] = v + threadIdx.x; }

template <int MODE>
__global__ __launch_bounds__(512, 4) void probe(const u32x4* __restrict__ W, const uint32_t* __restrict__ x, unsigned long long* stamps, uint32_t* sink) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned long long t0 = 0, t1 = 0;
    if (tid == 0) t0 = wall();
    const u32x4* wp = W + ((size_t)blockIdx.x * 8 + wave) * 1024 + lane;   // 16 KiB per wave
    u32x4 w[16];
    uint32_t acc = 0;
    if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = __builtin_nontemporal_load(wp + u * 64);
        const u32x4 xv = *((const u32x4*)x + tid);          // 8 KB over the block
        acc = xv[0] ^ xv[1] ^ xv[2] ^ xv[3];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (tid == 0) t1 = wall();
    } else if (MODE == 1) {
        const u32x4 xv = *((const u32x4*)x + tid);
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = __builtin_nontemporal_load(wp + u * 64);
        asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        acc = xv[0] ^ xv[1] ^ xv[2] ^ xv[3];
        if (tid == 0) t1 = wall();
    } else if (MODE == 3) {   // no weight requests at all: what does the 8 KB cost by itself at the start of a launch?
        const u32x4 xv = *((const u32x4*)x + tid);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        acc = xv[0] ^ xv[1] ^ xv[2] ^ xv[3];
        if (tid == 0) t1 = wall();
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = (u32x4){0u, 0u, 0u, 0u};
    } else {
        const uint32_t* xs = x + __builtin_amdgcn_readfirstlane(wave) * 256;   // this wave's 1 KB of x ...
        u32x16 s[8];                                                             // ... half of it here (8 x 64 B = 512 B keeps the SGPR budget)
#pragma unroll
        for (int j = 0; j < 8; ++j) s[j] = *(const u32x16*)(xs + j * 16);       // uniform address: s_load_dwordx16
#pragma unroll
        for (int u = 0; u < 16; ++u) w[u] = __builtin_nontemporal_load(wp + u * 64);
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc ^= s[j][k];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (tid == 0) t1 = wall();
    }
#pragma unroll
    for (int u = 0; u < 16; ++u) acc ^= w[u][0] ^ w[u][1] ^ w[u][2] ^ w[u][3];
    if (tid == 0) {
        stamps[blockIdx.x * 4 + 0] = t0;
        stamps[blockIdx.x * 4 + 1] = t1;
        stamps[blockIdx.x * 4 + 2] = wall();
    }
    if (acc == 0x12345678u) sink[tid] = acc;
}

int main() {
    const int G = 512, NB = 6;
    void* W[NB];
    for (int i = 0; i < NB; ++i) { CHECK(hipMalloc(&W[i], (size_t)G * 8 * 16384)); CHECK(hipMemset(W[i], i + 1, (size_t)G * 8 * 16384)); }
    uint32_t *x, *sink;
    unsigned long long* st;
    CHECK(hipMalloc(&x, 8192)); CHECK(hipMalloc(&sink, 4096)); CHECK(hipMalloc(&st, G * 4 * 8));
    const char* names[] = {"vector x behind the head (vmcnt(0))", "vector x first, counted wait", "scalar x first (s_load_dwordx16)",
                           "x alone, no weight requests", "x alone, NOT rewritten before the launch"};
    for (int mode = 0; mode < 5; ++mode) {
        std::vector<double> med;
        double mn = 1e9, mx = 0, all = 0;
        for (int rep = 0; rep < NB; ++rep) {
            if (mode != 4) hipLaunchKernelGGL(write_x, dim3(8), dim3(256), 0, 0, x, (uint32_t)(rep * 977 + mode));
            if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(G), dim3(512), 0, 0, (const u32x4*)W[rep], x, st, sink);
            if (mode == 1) hipLaunchKernelGGL(probe<1>, dim3(G), dim3(512), 0, 0, (const u32x4*)W[rep], x, st, sink);
            if (mode == 2) hipLaunchKernelGGL(probe<2>, dim3(G), dim3(512), 0, 0, (const u32x4*)W[rep], x, st, sink);
            if (mode >= 3) hipLaunchKernelGGL(probe<3>, dim3(G), dim3(512), 0, 0, (const u32x4*)W[rep], x, st, sink);
            CHECK(hipDeviceSynchronize());
            if (rep == 0) continue;   // warm-up
            std::vector<unsigned long long> h(G * 4);
            CHECK(hipMemcpy(h.data(), st, G * 4 * 8, hipMemcpyDeviceToHost));
            std::vector<double> v;
            unsigned long long e0 = ~0ull, e1 = 0;
            for (int b = 0; b < G; ++b) { v.push_back((double)(h[b * 4 + 1] - h[b * 4]) * 0.01); e0 = std::min(e0, h[b * 4]); e1 = std::max(e1, h[b * 4 + 2]); }
            std::sort(v.begin(), v.end());
            med.push_back(v[G / 2]); mn = std::min(mn, v[0]); mx = std::max(mx, v[G - 1]); all += (double)(e1 - e0) * 0.01;
        }
        std::sort(med.begin(), med.end());
        printf("%-40s x arrives min %.2f / median %.2f / max %.2f us after the block's entry; launch done %.1f us after the first entry\n", names[mode], mn,
               med[med.size() / 2], mx, all / (NB - 1));
    }
    return 0;
}
