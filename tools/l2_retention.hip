// l2_retention.hip -- does data read by kernel k survive in the XCD L2 for kernel k+1 (same block -> same XCD), and does a
// "prefetch the next kernel's first bytes while this one drains" tail pay?  (decode launch-floor study)
//   part 1: 512 blocks x 256 threads stream a buffer of S MB; "same" = the same buffer every launch, "cold" = rotating through a
//           4 GB pool.  us per launch.
//   part 2: chain of launches, each streaming its own cold 100 MB region; with pf > 0 every block ends by touching the first pf
//           bytes of the slice it will read first in the NEXT launch.  us per launch with / without.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/bin/l2_retention tools/l2_retention.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

__global__ __launch_bounds__(256) void stream_read(const u32x4* __restrict__ p, size_t vec_per_block, const u32x4* __restrict__ next,
                                                   size_t pf_vec_per_block, size_t next_vec_per_block, unsigned int* sink) {
    const u32x4* s = p + (size_t)blockIdx.x * vec_per_block;
    u32x4 acc = {0u, 0u, 0u, 0u};
    size_t i = threadIdx.x;
    for (; i + 7 * 256 < vec_per_block; i += 8 * 256) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = __builtin_nontemporal_load(s + i + j * 256);
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    for (; i < vec_per_block; i += 256) acc ^= __builtin_nontemporal_load(s + i);
    if (pf_vec_per_block) {   // touch the head of the slice block b reads first in the next launch (default policy: stays in L2)
        const u32x4* n = next + (size_t)blockIdx.x * next_vec_per_block;
        for (size_t k = threadIdx.x; k < pf_vec_per_block; k += 256) acc ^= n[k];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

// default-policy variant for part 1 (nt loads may bypass L2 retention)
__global__ __launch_bounds__(256) void stream_read_plain(const u32x4* __restrict__ p, size_t vec_per_block, unsigned int* sink) {
    const u32x4* s = p + (size_t)blockIdx.x * vec_per_block;
    u32x4 acc = {0u, 0u, 0u, 0u};
    size_t i = threadIdx.x;
    for (; i + 7 * 256 < vec_per_block; i += 8 * 256) {
        u32x4 v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = s[i + j * 256];
#pragma unroll
        for (int j = 0; j < 8; ++j) acc ^= v[j];
    }
    for (; i < vec_per_block; i += 256) acc ^= s[i];
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const size_t POOL = (size_t)4 << 30;
    char* pool;
    unsigned int* sink;
    CK(hipMalloc(&pool, POOL));
    CK(hipMalloc(&sink, 4));
    CK(hipMemset(pool, 1, POOL));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    const int G = 512, REPS = 200;
    printf("part 1: re-read of the same buffer vs cold, us per launch (plain loads)\n");
    for (size_t mb : {4, 8, 16, 32, 64, 128, 256}) {
        const size_t bytes = mb << 20, vpb = bytes / 16 / G;
        for (int cold = 0; cold < 2; ++cold) {
            float ms;
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < REPS; ++r) {
                    const size_t off = cold ? ((size_t)r * bytes) % (POOL - bytes) : 0;
                    hipLaunchKernelGGL(stream_read_plain, dim3(G), dim3(256), 0, 0, (const u32x4*)(pool + off), vpb, sink);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("  %4zu MB %s: %7.2f us  (%.2f TB/s)\n", mb, cold ? "cold" : "same", ms * 1e3 / REPS, bytes / (ms * 1e-3 / REPS) / 1e12);
        }
    }
    printf("part 2: chain of cold launches (nt stream), tail prefetch of the next launch's head, us per launch\n");
    for (size_t mb : {34, 90, 100, 180}) {
        const size_t bytes = mb << 20, vpb = bytes / 16 / G;
        for (size_t pf_kb : {0, 16, 32, 64, 128}) {
            const size_t pfv = pf_kb * 1024 / 16;
            float ms;
            for (int pass = 0; pass < 2; ++pass) {
                CK(hipEventRecord(e0));
                for (int r = 0; r < REPS; ++r) {
                    const size_t off = ((size_t)r * bytes) % (POOL - 2 * bytes), noff = ((size_t)(r + 1) * bytes) % (POOL - 2 * bytes);
                    hipLaunchKernelGGL(stream_read, dim3(G), dim3(256), 0, 0, (const u32x4*)(pool + off), vpb, (const u32x4*)(pool + noff), pfv, vpb, sink);
                }
                CK(hipEventRecord(e1));
                CK(hipEventSynchronize(e1));
                CK(hipEventElapsedTime(&ms, e0, e1));
            }
            printf("  %4zu MB, prefetch %3zu KB/block (%5.1f MB): %7.2f us per launch\n", mb, pf_kb, pf_kb * G / 1024.0, ms * 1e3 / REPS);
        }
    }
    return 0;
}
