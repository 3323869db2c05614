// valu_rate.hip -- instruction-rate probe for the decode GEMV inner loop candidates (not part of the product).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(2))) float f32x2;
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s\n", hipGetErrorString(e)); return 1; } } while (0)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, uint32_t seed) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    uint32_t w = seed + threadIdx.x, x = seed * 3 + 1;
    f32x2 p0 = {1, 2}, p1 = {3, 4}, p2 = {5, 6}, p3 = {7, 8};
    const f32x2 m = {__uint_as_float(0x3f800001u + seed), 1.0f}, c = {0.5f, 0.25f};
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {   // v_dot2c_f32_bf16, 8 independent accumulators
            bf16x2 ww = __builtin_bit_cast(bf16x2, w), xx = __builtin_bit_cast(bf16x2, x);
            a0 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a0, false); a1 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a1, false);
            a2 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a2, false); a3 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a3, false);
            a4 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a4, false); a5 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a5, false);
            a6 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a6, false); a7 = __builtin_amdgcn_fdot2_f32_bf16(ww, xx, a7, false);
        } else if (KIND == 1) {   // v_fma_f32
            const float ww = __uint_as_float(w), xx = __uint_as_float(x);
            a0 = fmaf(ww, xx, a0); a1 = fmaf(ww, xx, a1); a2 = fmaf(ww, xx, a2); a3 = fmaf(ww, xx, a3);
            a4 = fmaf(ww, xx, a4); a5 = fmaf(ww, xx, a5); a6 = fmaf(ww, xx, a6); a7 = fmaf(ww, xx, a7);
        } else if (KIND == 3) {   // v_exp_f32, 8 independent
            a0 = __builtin_amdgcn_exp2f(a0); a1 = __builtin_amdgcn_exp2f(a1); a2 = __builtin_amdgcn_exp2f(a2); a3 = __builtin_amdgcn_exp2f(a3);
            a4 = __builtin_amdgcn_exp2f(a4); a5 = __builtin_amdgcn_exp2f(a5); a6 = __builtin_amdgcn_exp2f(a6); a7 = __builtin_amdgcn_exp2f(a7);
        } else if (KIND == 4) {   // v_exp_f32 and v_fma_f32 alternating, 4 + 4 independent
            a0 = __builtin_amdgcn_exp2f(a0); a4 = fmaf(a4, 1.0001f, 0.5f); a1 = __builtin_amdgcn_exp2f(a1); a5 = fmaf(a5, 1.0001f, 0.5f);
            a2 = __builtin_amdgcn_exp2f(a2); a6 = fmaf(a6, 1.0001f, 0.5f); a3 = __builtin_amdgcn_exp2f(a3); a7 = fmaf(a7, 1.0001f, 0.5f);
        } else if (KIND == 5) {   // v_max3_f32, 8 independent
            const float ww = __uint_as_float(w), xx = __uint_as_float(x);
            a0 = __builtin_fmaxf(__builtin_fmaxf(a0, ww), xx); a1 = __builtin_fmaxf(__builtin_fmaxf(a1, ww), xx); a2 = __builtin_fmaxf(__builtin_fmaxf(a2, ww), xx); a3 = __builtin_fmaxf(__builtin_fmaxf(a3, ww), xx);
            a4 = __builtin_fmaxf(__builtin_fmaxf(a4, ww), xx); a5 = __builtin_fmaxf(__builtin_fmaxf(a5, ww), xx); a6 = __builtin_fmaxf(__builtin_fmaxf(a6, ww), xx); a7 = __builtin_fmaxf(__builtin_fmaxf(a7, ww), xx);
        } else if (KIND == 6) {   // v_cvt_scalef32_pk_bf16_fp8 (2 values per instruction), 8 independent
            uint32_t r0 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, false)), r1 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w, 1.0f, true));
            uint32_t r2 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x, 1.0f, false)), r3 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x, 1.0f, true));
            uint32_t r4 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w + 1, 1.0f, false)), r5 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(w + 1, 1.0f, true));
            uint32_t r6 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x + 1, 1.0f, false)), r7 = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(x + 1, 1.0f, true));
            w = r0 ^ r1 ^ r4 ^ r5; x = r2 ^ r3 ^ r6 ^ r7;   // (+ 8 cheap xor / add per 8 conversions: see the v_fma row for their cost)
        } else if (KIND == 7) {   // v_cvt_pk_f32_fp8 (2 values per instruction), 8 independent
            f32x2 r0 = __builtin_amdgcn_cvt_pk_f32_fp8(w, false), r1 = __builtin_amdgcn_cvt_pk_f32_fp8(w, true);
            f32x2 r2 = __builtin_amdgcn_cvt_pk_f32_fp8(x, false), r3 = __builtin_amdgcn_cvt_pk_f32_fp8(x, true);
            f32x2 r4 = __builtin_amdgcn_cvt_pk_f32_fp8(w + 1, false), r5 = __builtin_amdgcn_cvt_pk_f32_fp8(w + 1, true);
            f32x2 r6 = __builtin_amdgcn_cvt_pk_f32_fp8(x + 1, false), r7 = __builtin_amdgcn_cvt_pk_f32_fp8(x + 1, true);
            w = __float_as_uint(r0[0]) ^ __float_as_uint(r1[1]) ^ __float_as_uint(r4[0]) ^ __float_as_uint(r5[1]);
            x = __float_as_uint(r2[0]) ^ __float_as_uint(r3[1]) ^ __float_as_uint(r6[0]) ^ __float_as_uint(r7[1]);
        } else if (KIND == 8) {   // v_perm_b32 + v_and_or_b32 pairs (the integer route from e4m3 to bf16), 4 + 4
            uint32_t t0 = __builtin_amdgcn_perm(w, x, 0x010c000cu), t1 = __builtin_amdgcn_perm(w, x, 0x030c020cu);
            uint32_t t2 = __builtin_amdgcn_perm(x, w, 0x010c000cu), t3 = __builtin_amdgcn_perm(x, w, 0x030c020cu);
            t0 = (t0 & 0x80008000u) | (w >> 4); t1 = (t1 & 0x80008000u) | (x >> 4);
            t2 = (t2 & 0x80008000u) | (w >> 5); t3 = (t3 & 0x80008000u) | (x >> 5);
            w = t0 ^ t1; x = t2 ^ t3;
        } else {   // v_pk_fma_f32 (4 per iteration = 8 MACs)
            p0 = __builtin_elementwise_fma(p0, m, c); p1 = __builtin_elementwise_fma(p1, m, c);
            p2 = __builtin_elementwise_fma(p2, m, c); p3 = __builtin_elementwise_fma(p3, m, c);
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(w ^ x) + p0[0] + p0[1] + p1[0] + p1[1] + p2[0] + p2[1] + p3[0] + p3[1];
}

int main() {
    float* out; CHECK(hipMalloc(&out, 256 * 4 * 256 * 16));
    const int iters = 20000, blocks = 256 * 8;
    const char* names[] = {"v_dot2c_f32_bf16 (2 MAC/lane)", "v_fma_f32 (1 MAC/lane)", "v_pk_fma_f32 (2 MAC/lane)", "v_exp_f32", "v_exp_f32 + v_fma_f32 alternating", "v_max3_f32",
                           "v_cvt_scalef32_pk_bf16_fp8 (+1 xor each)", "v_cvt_pk_f32_fp8 (+1 xor each)", "v_perm_b32 / v_and_or mix (+shift, xor)"};
    for (int kind = 0; kind < 9; ++kind) {
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep) {
            CHECK(hipEventRecord(e0));
            if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 5) hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 6) hipLaunchKernelGGL(k<6>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 7) hipLaunchKernelGGL(k<7>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            if (kind == 8) hipLaunchKernelGGL(k<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 7u);
            CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        }
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double instr = (double)blocks * 4 /*waves*/ * iters * (kind == 2 ? 4 : 8);
        const double per_simd = instr / (256.0 * 4);
        printf("%-34s %8.3f ms  -> %.2f ns per wave-instruction per SIMD (~%.1f cycles @2.1GHz)\n", names[kind], ms, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.1);
    }
    return 0;
}
