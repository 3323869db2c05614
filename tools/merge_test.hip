// Self-check: attn_merge_chunk<8,4> (branch-free) vs attn_merge_chunk_loop on random split partials, bit for bit.
// build: hipcc --offload-arch=gfx950 -O3 -I emma-x_amd/csrc -o tools/bin/merge_test tools/merge_test.hip
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "common.h"

__global__ void k(const float* part, u32x4_t* a, u32x4_t* b, int Hq) {
    const int cg = blockIdx.x * blockDim.x + threadIdx.x;   // chunk: head cg>>4, elements (cg&15)*8
    if (cg >= Hq * 16) return;
    const float* pp = part + (size_t)(cg >> 4) * 8 * EMMAX_PSTRIDE;
    a[cg] = attn_merge_chunk<8, 4>(pp, (cg & 15) * 8);
    b[cg] = attn_merge_chunk_loop(pp, (cg & 15) * 8, 8);
}

int main() {
    const int Hq = 32, n = Hq * 8 * EMMAX_PSTRIDE;
    std::vector<float> h(n);
    int bad = 0;
    for (int trial = 0; trial < 64; ++trial) {
        srand(trial);
        for (int hd = 0; hd < Hq; ++hd)
            for (int s = 0; s < 8; ++s) {
                float* p = &h[(hd * 8 + s) * EMMAX_PSTRIDE];
                const bool empty = (rand() % 5) == 0;
                for (int j = 0; j < 128; ++j) p[j] = empty ? 0.f : (rand() / (float)RAND_MAX - 0.5f) * 50.f;
                p[128] = empty ? -INFINITY : (rand() / (float)RAND_MAX - 0.5f) * (trial % 4 == 0 ? 200.f : 20.f);
                p[129] = empty ? 0.f : rand() / (float)RAND_MAX * 80.f + 0.01f;
                p[130] = p[131] = 0.f;
            }
        float* d; u32x4_t *da, *db;
        (void)hipMalloc(&d, n * 4); (void)hipMalloc(&da, Hq * 16 * 16); (void)hipMalloc(&db, Hq * 16 * 16);
        (void)hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(Hq * 16 / 64), dim3(64), 0, 0, d, da, db, Hq);
        std::vector<uint32_t> ra(Hq * 16 * 4), rb(Hq * 16 * 4);
        (void)hipMemcpy(ra.data(), da, Hq * 16 * 16, hipMemcpyDeviceToHost);
        (void)hipMemcpy(rb.data(), db, Hq * 16 * 16, hipMemcpyDeviceToHost);
        for (size_t i = 0; i < ra.size(); ++i)
            if (ra[i] != rb[i]) { if (bad < 8) printf("trial %d word %zu: %08x vs %08x\n", trial, i, ra[i], rb[i]); ++bad; }
        (void)hipFree(d); (void)hipFree(da); (void)hipFree(db);
    }
    printf(bad ? "merge variants DIFFER in %d words\n" : "merge variants identical\n", bad);
    return bad != 0;
}
