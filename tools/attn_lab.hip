// attn_lab.hip -- tuning aid for the prefill / ViT attention kernel (not part of the product): includes attention.hip as it is,
// optionally with -DATTN_LAB_KO_* knock-outs (timing only, results are wrong by construction), and times the three shapes of
// profiles/r01_vision_prefill_kernel_stats.csv with HIP events.  Build: tools/attn_lab.sh
#include "../emma-x_amd/csrc/attention.hip"

#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main(int argc, char** argv) {
    struct Shape { const char* name; int B, N, H, hd, causal; };
    const Shape shapes[] = {{"dino<64> B=256 N=261", 256, 261, 16, 64, 0},
                            {"siglip<72> B=256 N=256", 256, 256, 16, 72, 0},
                            {"llama<128> causal B=8 S=768", 8, 768, 32, 128, 1},
                            {"llama<128> causal B=1 S=768", 1, 768, 32, 128, 1},
                            {"dino<64> B=16", 16, 261, 16, 64, 0}, {"siglip<72> B=16", 16, 256, 16, 72, 0},
                            {"dino<64> B=32", 32, 261, 16, 64, 0}, {"siglip<72> B=32", 32, 256, 16, 72, 0},
                            {"dino<64> B=64", 64, 261, 16, 64, 0}, {"siglip<72> B=64", 64, 256, 16, 72, 0},
                            {"long<64> B=16 N=4096", 16, 4096, 16, 64, 0},
                            {"long<72> B=16 N=4096", 16, 4096, 16, 72, 0},
                            {"long<128> B=8 N=4096", 8, 4096, 32, 128, 0}};
    const int reps = argc > 1 ? atoi(argv[1]) : 20;
    for (const Shape& s : shapes) {
        const int D = s.H * s.hd, ld = (3 * D + 127) / 128 * 128;
        const size_t n = (size_t)s.B * s.N * ld;
        std::vector<uint16_t> h(n);
        uint32_t r = 12345u;
        for (size_t i = 0; i < n; ++i) {   // uniform in [-1, 1) as bf16
            r = r * 1664525u + 1013904223u;
            const float f = ((int)(r >> 8) - (1 << 23)) * (1.0f / (1 << 23));
            h[i] = (uint16_t)(__builtin_bit_cast(uint32_t, f) >> 16);
        }
        std::vector<int32_t> cu(s.B + 1);
        for (int i = 0; i <= s.B; ++i) cu[i] = i * s.N;
        void *qkv, *out, *dcu;
        CHECK(hipMalloc(&qkv, n * 2));
        CHECK(hipMalloc(&out, (size_t)s.B * s.N * D * 2));
        CHECK(hipMalloc(&dcu, cu.size() * 4));
        CHECK(hipMemcpy(qkv, h.data(), n * 2, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dcu, cu.data(), cu.size() * 4, hipMemcpyHostToDevice));
        AttnParams p = {qkv, out, (const int32_t*)dcu, ld, 0, D, 2 * D, D, s.B, s.N, s.H, s.H, 1.0f / sqrtf((float)s.hd), s.causal};
        if (launch_attention(p, s.hd, 0) != 0) { printf("launch failed\n"); return 1; }
        CHECK(hipDeviceSynchronize());
        hipEvent_t e0, e1;
        CHECK(hipEventCreate(&e0));
        CHECK(hipEventCreate(&e1));
        CHECK(hipEventRecord(e0, 0));
        for (int i = 0; i < reps; ++i) launch_attention(p, s.hd, 0);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipDeviceSynchronize());
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, fl = 4.0 * s.N * s.N * s.hd * s.H * s.B * (s.causal ? 0.5 : 1.0);
        printf("%-30s %8.1f us  %6.0f TFLOP/s executed\n", s.name, us, fl / us / 1e6);
#ifdef ATTN_LAB_TRACE
        if (s.B == 256 && getenv("TRACE")) {
            static unsigned long long tr[16 * 12 * 8 * 8];
            CHECK(hipMemcpyFromSymbol(tr, HIP_SYMBOL(g_attn_trace), sizeof(tr)));
            const int L = 8, it = atoi(getenv("TRACE"));
            const unsigned long long t0 = tr[((L * 12 + 0) * 8 + it) * 8 + 0];
            printf("block %d item %d, clocks from wave 0's start: start, requested, computed, stored, landed, barrier\n", L, it);
            for (int w = 0; w < 12; ++w) {
                printf("  wave %2d:", w);
                for (int k = 0; k < 6; ++k) printf(" %8lld", (long long)(tr[((L * 12 + w) * 8 + it) * 8 + k] - t0));
                printf("\n");
            }
        }
#endif
        CHECK(hipFree(qkv));
        CHECK(hipFree(out));
        CHECK(hipFree(dcu));
    }
    return 0;
}
