// decode_lab.hip -- phase stamps of the small-batch MFMA decode projection (not part of the product): includes decode_mfma.hip
// with -DDECODE_LAB_TRACE and prints, over the 256 blocks of one plain launch (rotating weights, so HBM cold), when each phase
// is reached relative to the first block's entry: entry, activations requested, staged (barrier), first ring block consumed,
// stream done, last reduction + epilogue done.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -DDECODE_LAB_TRACE -Iemma-x_amd/csrc -Iinclude tools/decode_lab.hip -o tools/bin/decode_lab
#include "../emma-x_amd/csrc/decode_mfma.hip"

#include <algorithm>
#include <string.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 8;
    struct Shape { int N, K; };
    const Shape shapes[] = {{4096, 4096}, {12288, 4096}, {22016, 4096}, {4096, 11008}};
    if (decode_mfma_init() != 0) { printf("init failed\n"); return 1; }
    for (const Shape& s : shapes) {
        const size_t nw = (size_t)s.N * s.K;
        const int NBUF = 5;
        void *raw, *fm[NBUF], *x, *y;
        CHECK(hipMalloc(&raw, nw * 2));
        CHECK(hipMemset(raw, 0x11, nw * 2));
        for (int i = 0; i < NBUF; ++i) {
            CHECK(hipMalloc(&fm[i], nw * 2));
            if (launch_repack_fm(raw, s.K, fm[i], s.N, s.K, 0) != 0) { printf("repack failed\n"); return 1; }
        }
        CHECK(hipMalloc(&x, (size_t)8 * s.K * 2));
        CHECK(hipMemset(x, 0x11, (size_t)8 * s.K * 2));
        CHECK(hipMalloc(&y, (size_t)8 * s.N * 2));
        void* skws;   // stream-K granules, as the session provides them (all zero between launches); SK=0: whole tasks per block
        CHECK(hipMalloc(&skws, (size_t)256 * 2 * 256 * 8));
        CHECK(hipMemset(skws, 0, (size_t)256 * 2 * 256 * 8));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        float ms = 0.f;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < NBUF; ++i) {
                GemvParams p;
                memset(&p, 0, sizeof(p));
                p.x = x; p.ldx = s.K; p.W = fm[i]; p.ldw = s.K; p.K = s.K; p.y = y; p.ldy = s.N; p.n_rows = s.N;
                if (!(getenv("SK") && atoi(getenv("SK")) == 0)) p.sk_ws = (unsigned long long*)skws;
                if (launch_decode_mfma(GEMV_PLAIN, p, B, 0) != 0) { printf("launch failed\n"); return 1; }
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        static unsigned long long tr[256 * 8];
        CHECK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_dec_trace), sizeof(tr)));
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < 256; ++b) t0 = std::min(t0, tr[b * 8]);
        printf("N=%d K=%d B=%d: %.1f us per launch (%.0f MB, %.2f TB/s); us from the first block's entry (min / median / max over blocks):\n", s.N, s.K, B,
               ms * 1e3 / NBUF, nw * 2 / 1e6, nw * 2 / (ms * 1e-3 / NBUF) / 1e12);
        const char* names[] = {"entry", "x in LDS", "x staged (barrier)", "first block consumed", "stream done", "epilogue done", "x loads issued", "weight head issued"};
        const int order[] = {0, 6, 7, 1, 2, 3, 4, 5};
        for (int kk = 0; kk < 8; ++kk) {
            const int k = order[kk];
            std::vector<double> v;
            for (int b = 0; b < 256; ++b) v.push_back((double)(tr[b * 8 + k] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            printf("  %-22s %6.2f / %6.2f / %6.2f\n", names[k], v[0], v[128], v[255]);
        }
        {   // who finishes late?  "stream done" per XCD (block b runs on XCD b % 8) and per eighth of the block range
            printf("  stream done by XCD (mean / max):");
            for (int xc = 0; xc < 8; ++xc) {
                double m = 0, mx = 0;
                for (int b = xc; b < 256; b += 8) { const double t = (double)(tr[b * 8 + 4] - t0) * 0.01; m += t / 32; mx = std::max(mx, t); }
                printf("  %5.1f/%5.1f", m, mx);
            }
            printf("\n  stream done by block range of 32 (mean / max):");
            for (int r = 0; r < 8; ++r) {
                double m = 0, mx = 0;
                for (int b = r * 32; b < r * 32 + 32; ++b) { const double t = (double)(tr[b * 8 + 4] - t0) * 0.01; m += t / 32; mx = std::max(mx, t); }
                printf("  %5.1f/%5.1f", m, mx);
            }
            printf("\n");
        }
        for (int i = 0; i < NBUF; ++i) CHECK(hipFree(fm[i]));
        CHECK(hipFree(raw)); CHECK(hipFree(x)); CHECK(hipFree(y)); CHECK(hipFree(skws));
    }
    return 0;
}
