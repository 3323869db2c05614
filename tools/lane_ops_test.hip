// Tuning aid / self-check for the cross-lane helpers of csrc/common.h: row16_sum (DPP butterfly inside a 16-lane row),
// rows_sum / rows_max (v_permlane32_swap + v_permlane16_swap across the four rows), wave_sum / wave_max.
// build: hipcc --offload-arch=gfx950 -O3 -I emma-x_amd/csrc -o tools/bin/lane_ops_test tools/lane_ops_test.hip
#include <cmath>
#include <cstdio>
#include <vector>

#include "common.h"

__global__ void k(const float* in, float* out) {
    const int l = threadIdx.x;
    const float a = in[l];
    out[0 * 64 + l] = row16_sum(a);
    out[1 * 64 + l] = rows_sum(a);
    out[2 * 64 + l] = rows_max(a);
    out[3 * 64 + l] = wave_sum(a);
    out[4 * 64 + l] = wave_max(a);
}

int main() {
    std::vector<float> h(64), o(5 * 64);
    for (int i = 0; i < 64; ++i) h[i] = (float)((i * 37) % 101) - 50.f;   // small integers: every sum is exact
    float *di, *dout;
    hipMalloc(&di, 64 * 4);
    hipMalloc(&dout, 5 * 64 * 4);
    hipMemcpy(di, h.data(), 64 * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(o.data(), dout, 5 * 64 * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    float tot = 0.f, mx = -1e30f;
    for (int i = 0; i < 64; ++i) { tot += h[i]; mx = fmaxf(mx, h[i]); }
    for (int l = 0; l < 64; ++l) {
        float r16 = 0.f, rs = 0.f, rm = -1e30f;
        for (int j = 0; j < 16; ++j) r16 += h[(l & ~15) + j];
        for (int r = 0; r < 4; ++r) { rs += h[(l & 15) + 16 * r]; rm = fmaxf(rm, h[(l & 15) + 16 * r]); }
        const float want[5] = {r16, rs, rm, tot, mx};
        for (int t = 0; t < 5; ++t)
            if (o[t * 64 + l] != want[t]) { if (bad < 10) printf("mismatch op %d lane %d: %g want %g\n", t, l, o[t * 64 + l], want[t]); ++bad; }
    }
    printf(bad ? "FAILED (%d)\n" : "lane ops OK\n", bad);
    return bad != 0;
}
