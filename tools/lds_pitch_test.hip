// lds_pitch_test.hip -- bank behaviour of the two fragment read patterns of the attention kernels as a function of the LDS row
// pitch (not part of the product): clocks per wave-level read, one wave per CU so nothing else competes.
//   K pattern: ds_read_b128, lane (ql = lane & 31, hi = lane >> 5) reads 16 B at row ql, byte 16 hi (+ 32 kk)
//   V pattern: ds_read_b64_tr_b16, lane reads 8 B at row 4 hi + (i16 >> 2), byte 32 b4 + 8 (i16 & 3)  (i16 = lane & 15, b4 = (lane >> 4) & 1)
// Build: hipcc --offload-arch=gfx950 -O3 tools/lds_pitch_test.hip -o tools/bin/lds_pitch_test
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <int KIND>
__global__ __launch_bounds__(64) void k(unsigned long long* out, uint32_t* sink, int pitch, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[65536];
    const int lane = threadIdx.x, ql = lane & 31, hi = lane >> 5, i16 = lane & 15, b4 = (lane >> 4) & 1;
    for (int i = lane * 16; i < 65536; i += 64 * 16) *(u32x4_t*)(sm + i) = (u32x4_t){1u, 2u, 3u, 4u};
    __syncthreads();
    const int offK = ql * pitch + hi * 16, offV = (4 * hi + (i16 >> 2)) * pitch + 32 * b4 + 8 * (i16 & 3);
    uint32_t acc = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND == 0) {
                const u32x4_t v = *(volatile u32x4_t*)(sm + offK + (j & 3) * 32 + (j >> 2) * 32 * pitch);
                acc += v[0] ^ v[3];
            } else {
                typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
                const s16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(sm + offV + (j & 1) * 64 + (j >> 1) * 8 * pitch));
                acc += (uint32_t)v[0] ^ (uint32_t)v[3];
            }
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (acc == 0x12345u) sink[0] = acc;
}

int main() {
    unsigned long long* out;
    uint32_t* sink;
    CHECK(hipMalloc(&out, 8));
    CHECK(hipMalloc(&sink, 4));
    const int iters = 2000;
    for (int kind = 0; kind < 2; ++kind)
        for (int pitch : {128, 144, 160, 176, 192, 208, 272, 320}) {
            unsigned long long h = 0;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64), 0, 0, out, sink, pitch, iters);
                else hipLaunchKernelGGL(k<1>, dim3(256), dim3(64), 0, 0, out, sink, pitch, iters);
                CHECK(hipDeviceSynchronize());
            }
            CHECK(hipMemcpy(&h, out, 8, hipMemcpyDeviceToHost));
            printf("%s pitch %3d B: %.1f clocks per read\n", kind == 0 ? "K  ds_read_b128       " : "V^T ds_read_b64_tr_b16", pitch, (double)h / (iters * 8.0));
        }
    return 0;
}
