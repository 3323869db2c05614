// rope_vec_test.hip -- the element-wise and the 16-byte prefill RoPE + K/V append kernels of misc.hip on the same random rows: bit-for-bit
// comparison of the rotated q/k rows and of both caches (not part of the product).
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I emma-x_amd/csrc -I include tools/rope_vec_test.hip -o tools/bin/rope_vec_test
#include "../emma-x_amd/csrc/misc.hip"
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
int main() {
    const int rows = 768, Hq = 32, Hkv = 32, hd = 128, page = 64, max_pages = 16, ld = (Hq + 2 * Hkv) * hd, half = hd / 2;
    std::vector<uint16_t> h((size_t)rows * ld);
    srand(3);
    for (auto& v : h) { float f = ((rand() % 20001) - 10000) * 3e-4f; uint32_t u; memcpy(&u, &f, 4); v = (uint16_t)(u >> 16); }
    std::vector<float> cs((size_t)rows * half), sn((size_t)rows * half);
    for (int p = 0; p < rows; ++p)
        for (int d = 0; d < half; ++d) { const float a = p * powf(10000.f, -2.f * d / hd); cs[(size_t)p * half + d] = cosf(a); sn[(size_t)p * half + d] = sinf(a); }
    std::vector<int32_t> cu = {0, rows}, pt(max_pages);
    for (int i = 0; i < max_pages; ++i) pt[i] = max_pages - 1 - i;
    const size_t cache = (size_t)max_pages * Hkv * page * hd;
    uint16_t *q[2], *kc[2], *vc[2]; float *dcs, *dsn; int32_t *dcu, *dpt;
    CK(hipMalloc(&dcs, cs.size() * 4)); CK(hipMalloc(&dsn, sn.size() * 4)); CK(hipMalloc(&dcu, 8)); CK(hipMalloc(&dpt, max_pages * 4));
    CK(hipMemcpy(dcs, cs.data(), cs.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(dsn, sn.data(), sn.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcu, cu.data(), 8, hipMemcpyHostToDevice)); CK(hipMemcpy(dpt, pt.data(), max_pages * 4, hipMemcpyHostToDevice));
    std::vector<uint16_t> oq[2], ok[2], ov[2];
    for (int v = 0; v < 2; ++v) {
        CK(hipMalloc(&q[v], h.size() * 2)); CK(hipMalloc(&kc[v], cache * 2)); CK(hipMalloc(&vc[v], cache * 2));
        CK(hipMemcpy(q[v], h.data(), h.size() * 2, hipMemcpyHostToDevice)); CK(hipMemset(kc[v], 0, cache * 2)); CK(hipMemset(vc[v], 0, cache * 2));
        if (v == 0) hipLaunchKernelGGL(emmax_rope_kv_write_kernel, dim3(rows), dim3(256), 0, 0, q[v], ld, 0, Hq * hd, (Hq + Hkv) * hd, dcu, 1, dcs, dsn, kc[v], vc[v], dpt, max_pages, Hq, Hkv, hd, page);
        else hipLaunchKernelGGL(emmax_rope_kv_write_vec_kernel, dim3(rows), dim3(256), 0, 0, q[v], ld, 0, Hq * hd, (Hq + Hkv) * hd, dcu, 1, dcs, dsn, kc[v], vc[v], dpt, max_pages, Hq, Hkv, hd, page);
        CK(hipDeviceSynchronize());
        oq[v].resize(h.size()); ok[v].resize(cache); ov[v].resize(cache);
        CK(hipMemcpy(oq[v].data(), q[v], h.size() * 2, hipMemcpyDeviceToHost)); CK(hipMemcpy(ok[v].data(), kc[v], cache * 2, hipMemcpyDeviceToHost));
        CK(hipMemcpy(ov[v].data(), vc[v], cache * 2, hipMemcpyDeviceToHost));
    }
    size_t dq = 0, dk = 0, dv = 0, first = (size_t)-1;
    for (size_t i = 0; i < h.size(); ++i) if (oq[0][i] != oq[1][i]) { if (first == (size_t)-1) first = i; ++dq; }
    for (size_t i = 0; i < cache; ++i) { dk += ok[0][i] != ok[1][i]; dv += ov[0][i] != ov[1][i]; }
    printf("rope vec vs element-wise: qkv rows differing %zu of %zu, K cache %zu, V cache %zu of %zu", dq, h.size(), dk, dv, cache);
    if (dq) printf("  first at row %zu col %zu: %04x vs %04x (input %04x)", first / ld, first % ld, oq[0][first], oq[1][first], h[first]);
    printf("\n");
    return 0;
}
