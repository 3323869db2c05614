// mfma_rate.hip -- issue-rate ceilings of the two bf16 MFMA shapes on gfx950 with ONE and TWO waves per SIMD (round 5: is the
// 16x16x32 loop of gemm.hip capped by its instruction, and does a second wave on the SIMD lift the cap?).  Registers only: NACC
// independent accumulators per wave, operands constant.  hipcc --offload-arch=gfx950 -O3 tools/mfma_rate.hip -o tools/bin/mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;

template <int NACC>
__global__ __launch_bounds__(512) void k16(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(512) void k32(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(1.0f + i * 0.01f); }
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i], 0, 0, 0);
    }
    float s = 0.f;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <class K>
static void run(const char* name, K kern, int threads, double flop_per_mfma, int nacc, float* out) {
    const int iters = 4000, grid = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * (threads / 64) * iters * nacc;
    printf("%-34s %d waves/SIMD: %8.3f ms  %7.1f TFLOP/s  (%.1f ns per MFMA per SIMD)\n", name, threads / 256, ms, mfma * flop_per_mfma / ms / 1e9,
           ms * 1e6 / ((double)iters * nacc * (threads / 256)));
}
int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        run("16x16x32 bf16, 8 accumulators", k16<8>, 256, 16384.0, 8, out);
        run("16x16x32 bf16, 8 accumulators", k16<8>, 512, 16384.0, 8, out);
        run("16x16x32 bf16, 32 accumulators", k16<32>, 256, 16384.0, 32, out);
        run("16x16x32 bf16, 32 accumulators", k16<32>, 512, 16384.0, 32, out);
        run("32x32x16 bf16, 4 accumulators", k32<4>, 256, 32768.0, 4, out);
        run("32x32x16 bf16, 4 accumulators", k32<4>, 512, 32768.0, 4, out);
        run("32x32x16 bf16, 8 accumulators", k32<8>, 256, 32768.0, 8, out);
        run("32x32x16 bf16, 8 accumulators", k32<8>, 512, 32768.0, 8, out);
    }
    return 0;
}
