// tr_read_test.hip -- pins the lane <-> element map of gfx950's ds_read_b64_tr_b16 (LDS transpose read) before the attention
// kernel relies on it.  LDS holds a row-major [KEYS][PITCH] image of u16 with value = key*256 + col.  Every lane of a 16-lane
// group b supplies the address of 4 contiguous u16: key = i>>2, cols b*16 + (i&3)*4 .. +3  (i = lane & 15).  Prints what each
// lane receives.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) short s16x4;
__global__ void k(unsigned short* out) {
    __shared__ __attribute__((aligned(16))) unsigned short lds[8 * 96];
    for (int i = threadIdx.x; i < 8 * 96; i += 64) lds[i] = (unsigned short)((i / 96) * 256 + (i % 96));
    __syncthreads();
    const int l = threadIdx.x, b = l >> 4, i = l & 15;
    const unsigned short* p = &lds[(i >> 2) * 96 + b * 16 + (i & 3) * 4];
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
    unsigned short* d;
    hipMalloc(&d, 64 * 4 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    unsigned short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int j = 0; j < 4; ++j) printf(" (k%d,c%2d)", h[l * 4 + j] >> 8, h[l * 4 + j] & 255);
        printf("\n");
    }
    return 0;
}
