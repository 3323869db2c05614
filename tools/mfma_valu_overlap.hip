// mfma_valu_overlap.hip -- do MFMA and VALU work of DIFFERENT waves on one SIMD overlap (not part of the product)?
// Blocks of 8 waves (two per SIMD): waves 0-3 run a loop of independent v_mfma_f32_32x32x16_bf16, waves 4-7 a loop of v_fma_f32 /
// v_exp_f32; timed alone and together.  Together ~ max(alone) = the two pipes run concurrently; ~ sum = they do not.
// Also: one wave alternating MFMA and independent VALU in program order (what software pipelining inside a wave relies on).
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o tools/bin/mfma_valu_overlap
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// mode bit 0: waves 0-3 do MFMA; bit 1: waves 4-7 do VALU; mode 4: every wave interleaves 1 MFMA with NV VALU instructions
template <int NV>
__global__ __launch_bounds__(512) void k(float* out, int iters, int mode) {
    const int wave = threadIdx.x >> 6;
    f32x16_t a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    bf16x8_t x, y;
    for (int i = 0; i < 8; ++i) { x[i] = (__bf16)(float)(threadIdx.x & 7); y[i] = (__bf16)0.5f; }
    float v0 = threadIdx.x, v1 = 1, v2 = 2, v3 = 3, v4 = 4, v5 = 5, v6 = 6, v7 = 7;
    if (mode == 4) {
        for (int it = 0; it < iters; ++it) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; j += 8) {
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
            a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NV; j += 8) {
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        }
    } else if (wave < 4) {
        if (mode & 1)
            for (int it = 0; it < iters; ++it) {
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a0, 0, 0, 0);
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a1, 0, 0, 0);
                a2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a2, 0, 0, 0);
                a3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, a3, 0, 0, 0);
            }
    } else if (mode & 8) {   // LDS reads: 8 conflict-free ds_read_b128 per iteration
        __shared__ __attribute__((aligned(16))) unsigned char sm[8192];
        typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
        u32x4 acc = {0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const u32x4 r = *(volatile u32x4*)(sm + ((j & 7) * 1024 + (threadIdx.x & 63) * 16) % 8192);
                acc[0] ^= r[0]; acc[1] ^= r[1]; acc[2] ^= r[2]; acc[3] ^= r[3];
            }
        }
        v0 += (float)(acc[0] ^ acc[1] ^ acc[2] ^ acc[3]);
    } else if (mode & 2) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {   // 32 v_fma_f32 per iteration
                v0 = fmaf(v0, 1.0001f, 0.5f); v1 = fmaf(v1, 1.0001f, 0.5f); v2 = fmaf(v2, 1.0001f, 0.5f); v3 = fmaf(v3, 1.0001f, 0.5f);
                v4 = fmaf(v4, 1.0001f, 0.5f); v5 = fmaf(v5, 1.0001f, 0.5f); v6 = fmaf(v6, 1.0001f, 0.5f); v7 = fmaf(v7, 1.0001f, 0.5f);
            }
        }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int i = 0; i < 16; ++i) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int NV>
static float run(float* out, int iters, int mode) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NV>, dim3(256), dim3(512), 0, 0, out, iters, mode);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* out;
    CHECK(hipMalloc(&out, 256 * 512 * 4));
    const int iters = 20000;
    const float tm = run<8>(out, iters, 1), tv = run<8>(out, iters, 2), tb = run<8>(out, iters, 3);
    printf("one MFMA wave per SIMD (4 x 32x32x16 per iteration):  %.3f ms = %.1f clk per MFMA @2.1 GHz\n", tm, tm * 1e-3 * 2.1e9 / (iters * 4.0));
    printf("one VALU wave per SIMD (32 v_fma_f32 per iteration):   %.3f ms = %.1f clk per fma\n", tv, tv * 1e-3 * 2.1e9 / (iters * 32.0));
    printf("both on the same SIMD:                                 %.3f ms  (max %.3f, sum %.3f)\n", tb, tm > tv ? tm : tv, tm + tv);
    const float tl = run<8>(out, iters, 8), tlb = run<8>(out, iters, 9);
    printf("one LDS wave per SIMD (8 ds_read_b128 per iteration):  %.3f ms = %.1f clk per read (4 waves share the CU's LDS)\n", tl, tl * 1e-3 * 2.1e9 / (iters * 8.0));
    printf("MFMA wave + LDS wave on the same SIMD:                 %.3f ms  (max %.3f, sum %.3f)\n", tlb, tm > tl ? tm : tl, tm + tl);
    const float i0 = run<8>(out, iters, 4), i1 = run<16>(out, iters, 4), i2 = run<32>(out, iters, 4);
    printf("two waves per SIMD, each: MFMA then N independent fmas, per MFMA: N=8 %.1f clk, N=16 %.1f clk, N=32 %.1f clk (both waves together)\n",
           i0 * 1e-3 * 2.1e9 / (iters * 2.0), i1 * 1e-3 * 2.1e9 / (iters * 2.0), i2 * 1e-3 * 2.1e9 / (iters * 2.0));
    return 0;
}
