// gemv_lab.hip -- phase stamps of the batch-1/2 dot2 GEMV decode kernel (not part of the product): includes decode.hip with
// -DDECODE_LAB_TRACE and prints, over the blocks of one launch (rotating weights: HBM cold), when each phase is reached relative
// to the first block's entry: entry, prologue done, activations staged (barrier), first weight block consumed, stream + epilogues
// done.  Shapes: LLaMA-7B gate/up (RMSNorm prologue, SwiGLU epilogue), down (+residual), o-proj sized plain rows.  Build:
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-inline-asm -DDECODE_LAB_TRACE -Iemma-x_amd/csrc -Iinclude tools/gemv_lab.hip -o tools/bin/gemv_lab
#include "../emma-x_amd/csrc/decode.hip"

#include <algorithm>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 1;
    const bool fp8 = argc > 2 && !strcmp(argv[2], "fp8");   // the e4m3 row copy (half the bytes), batch 1-2
    struct Shape { const char* name; int mode, N, K; };
    const Shape shapes[] = {{"gate/up (norm, SwiGLU)", MODE_GATEUP, 22016, 4096}, {"down (+residual)", MODE_RESID, 4096, 11008}, {"plain 4096 x 4096", MODE_PLAIN, 4096, 4096},
                            {"plain 12288 x 4096", MODE_PLAIN, 12288, 4096}};
    if (decode_gemv_init() != 0) { printf("init failed\n"); return 1; }
    for (const Shape& s : shapes) {
        const size_t nw = (size_t)s.N * s.K, wb = fp8 ? 1 : 2;
        const int NBUF = 5;
        void *W[NBUF], *x, *y, *nwp, *wsc;
        for (int i = 0; i < NBUF; ++i) { CHECK(hipMalloc(&W[i], nw * wb)); CHECK(hipMemset(W[i], 0x11, nw * wb)); }
        CHECK(hipMalloc(&wsc, (size_t)s.N * 4)); CHECK(hipMemset(wsc, 0, (size_t)s.N * 4));
        CHECK(hipMalloc(&x, (size_t)8 * s.K * 2)); CHECK(hipMemset(x, 0x11, (size_t)8 * s.K * 2));
        CHECK(hipMalloc(&nwp, (size_t)s.K * 2)); CHECK(hipMemset(nwp, 0x11, (size_t)s.K * 2));
        CHECK(hipMalloc(&y, (size_t)8 * s.N * 2)); CHECK(hipMemset(y, 0, (size_t)8 * s.N * 2));
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        float ms = 0.f;
        int grid = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CHECK(hipEventRecord(e0, 0));
            for (int i = 0; i < NBUF; ++i) {
                GemvParams p;
                memset(&p, 0, sizeof(p));
                p.x = x; p.ldx = s.K; p.W = W[i]; p.ldw = s.K; p.K = s.K; p.y = y; p.ldy = s.mode == MODE_GATEUP ? s.N / 2 : s.N; p.n_rows = s.N;
                p.norm_w = nwp; p.eps = 1e-5f;
                if (fp8) p.wscale = (const float*)wsc;
                if (launch_decode_gemv(s.mode, p, B, 0, &grid) != 0) { printf("launch failed\n"); return 1; }
            }
            CHECK(hipEventRecord(e1, 0));
            CHECK(hipDeviceSynchronize());
            CHECK(hipEventElapsedTime(&ms, e0, e1));
        }
        static unsigned long long tr[1024 * 8];
        CHECK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_gemv_trace), sizeof(tr)));
        const int nb = std::min(grid, 1024);
        unsigned long long t0 = ~0ull;
        for (int b = 0; b < nb; ++b) t0 = std::min(t0, tr[b * 8]);
        printf("%s N=%d K=%d B=%d grid %d: %.1f us per launch (%.0f MB, %.2f TB/s); us from the first block's entry (min / median / max over blocks):\n", s.name, s.N,
               s.K, B, grid, ms * 1e3 / NBUF, nw * wb / 1e6, nw * wb / (ms * 1e-3 / NBUF) / 1e12);
        const char* names[] = {"entry", "prologue done", "x staged (barrier)", "first block consumed", "stream + epilogues done", "x arrived, stats done", "stats barrier passed"};
        const int order[] = {0, 5, 6, 1, 2, 3, 4};
        for (int kk = 0; kk < 7; ++kk) {
            const int k = order[kk];
            std::vector<double> v;
            for (int b = 0; b < nb; ++b) v.push_back((double)(tr[b * 8 + k] - t0) * 0.01);
            std::sort(v.begin(), v.end());
            printf("  %-26s %6.2f / %6.2f / %6.2f\n", names[k], v[0], v[nb / 2], v[nb - 1]);
        }
        {   // who finishes late?  per XCD (block b runs on XCD b % 8) and per eighth of the block range
            printf("  done by XCD (mean / max):");
            for (int xc = 0; xc < 8; ++xc) {
                double m = 0, mx = 0; int n = 0;
                for (int b = xc; b < nb; b += 8) { const double t = (double)(tr[b * 8 + 4] - t0) * 0.01; m += t; mx = std::max(mx, t); ++n; }
                printf("  %5.1f/%5.1f", m / n, mx);
            }
            printf("\n  done by eighth of the block range (mean / max):");
            for (int r = 0; r < 8; ++r) {
                double m = 0, mx = 0; int n = 0;
                for (int b = r * nb / 8; b < (r + 1) * nb / 8; ++b) { const double t = (double)(tr[b * 8 + 4] - t0) * 0.01; m += t; mx = std::max(mx, t); ++n; }
                printf("  %5.1f/%5.1f", m / n, mx);
            }
            printf("\n");
        }
        for (int i = 0; i < NBUF; ++i) CHECK(hipFree(W[i]));
        CHECK(hipFree(x)); CHECK(hipFree(y)); CHECK(hipFree(nwp)); CHECK(hipFree(wsc));
    }
    return 0;
}
