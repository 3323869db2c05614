// gemm_trace.hip -- tuning aid (not part of the product): phase timestamps of block 0 of the GEMM (argv[5]: 1 = 256x256 geometry (default), 0 = 128x128;
// argv[6]: epilogue 0 = plain, 1 = LayerNorm fold, 2 = bias + LayerScale + residual, 3 = LayerNorm fold + GELU).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -DGEMM_LAB_TRACE -Iemma-x_amd/csrc tools/gemm_trace.hip emma-x_amd/csrc/gemm.hip -o tools/bin/gemm_trace
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>
#include "kernels.h"

// the library's tuning table lives in model.hip; this tool links gemm.hip alone: defaults + GEMM_DEEP from the environment
const EmmaxTune& emmax_tune() {
    static EmmaxTune t = [] {
        EmmaxTune x;
        memset(&x, 0, sizeof(x));
        x.gemm_big = -1; x.gemm_splitk = 1; x.gemm_deep = getenv("GEMM_DEEP") ? atoi(getenv("GEMM_DEEP")) : -1; x.gemm_lnfuse = 1;
        return x;
    }();
    return t;
}

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); exit(1); } } while (0)

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 1024;
    void *A, *W, *C;
    long long* tr;
    CHECK(hipMalloc(&A, (size_t)M * K * 2));
    CHECK(hipMalloc(&W, (size_t)N * K * 2));
    CHECK(hipMalloc(&C, (size_t)M * N * 2));
    CHECK(hipMalloc(&tr, 4096 * 8));
    std::vector<uint16_t> h((size_t)M * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3c00 + (rand() & 0x1ff);
    CHECK(hipMemcpy(A, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    h.resize((size_t)N * K);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0x3a00 + (rand() & 0x3ff);
    CHECK(hipMemcpy(W, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    GemmParams p;
    memset(&p, 0, sizeof(p));
    p.A = A; p.lda = K; p.W = W; p.ldw = K; p.C = C; p.ldc = N; p.M = M; p.N = N; p.K = K; p.N_store = N;
    p.dbg = argc > 4 ? atoi(argv[4]) : 0;
    const int big = argc > 5 ? atoi(argv[5]) : 1;
    const int epi = argc > 6 ? atoi(argv[6]) : 0;
    if (epi == 1 || epi == 3) {
        float *st, *ls, *lc;
        CHECK(hipMalloc(&st, (size_t)M * 8)); CHECK(hipMalloc(&ls, (size_t)N * 4)); CHECK(hipMalloc(&lc, (size_t)N * 4));
        std::vector<float> f((size_t)M * 2);
        for (size_t i = 0; i < f.size(); i += 2) { f[i] = 0.01f; f[i + 1] = 1.0f; }
        CHECK(hipMemcpy(st, f.data(), f.size() * 4, hipMemcpyHostToDevice));
        f.assign(N, 0.5f);
        CHECK(hipMemcpy(ls, f.data(), (size_t)N * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(lc, f.data(), (size_t)N * 4, hipMemcpyHostToDevice));
        p.ln_stats = st; p.ln_s = ls; p.ln_c = lc;
        if (epi == 3) p.act = 1;
    } else if (epi == 2) {
        void *b, *sc, *r;
        CHECK(hipMalloc(&b, (size_t)N * 2)); CHECK(hipMalloc(&sc, (size_t)N * 2)); CHECK(hipMalloc(&r, (size_t)M * N * 2));
        CHECK(hipMemset(b, 0x3c, (size_t)N * 2)); CHECK(hipMemset(sc, 0x3c, (size_t)N * 2)); CHECK(hipMemset(r, 0x3c, (size_t)M * N * 2));
        p.bias = b; p.scale = sc; p.residual = r; p.ldr = N;
    }
    for (int rep = 0; rep < 3; ++rep) {
        CHECK(hipMemset(tr, 0, 4096 * 8));
        p.trace = tr;
        if (launch_gemm_geom(p, big, 0) != 0) { printf("launch failed\n"); return 1; }
        CHECK(hipDeviceSynchronize());
    }
    std::vector<long long> t(4096);
    CHECK(hipMemcpy(t.data(), tr, 4096 * 8, hipMemcpyDeviceToHost));
    printf("M=%d N=%d K=%d big=%d epi=%d (100 MHz ticks -> us)\n tile  main_loop  setup+issue  epilogue_issue  wait_next\n", M, N, K, big, epi);
    for (int i = 0; i + 4 < 4096 && t[i + 3] != 0; i += 4) {
        const double ml = (t[i + 1] - t[i]) * 0.01, si = (t[i + 2] - t[i + 1]) * 0.01, ep = (t[i + 3] - t[i + 2]) * 0.01;
        const double wn = t[i + 4] ? (t[i + 4] - t[i + 3]) * 0.01 : 0.0;
        printf(" %3d   %8.2f   %8.2f   %8.2f   %8.2f\n", i / 4, ml, si, ep, wn);
    }
    return 0;
}
