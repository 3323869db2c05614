// norm.hip -- row normalisations (HBM-bound, one wave per row, 16-byte vector loads).
//   LayerNorm (eps 1e-6, affine)  = timm `Block.norm1/norm2`
//   RMSNorm                       = HF `LlamaRMSNorm.forward` (fp32 statistics, weight multiply after the normalise)
// Statistics and the normalise are done in fp32 from the bf16 row; one rounding on store.
#include "common.h"
#include "kernels.h"

namespace {

// D % 8 == 0, D <= 8 * 64 * MAXV
template <int MAXV, bool RMS>
__global__ __launch_bounds__(256) void emmax_rownorm_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y,
                                                           const bf16_t* __restrict__ w, const bf16_t* __restrict__ b,
                                                           int rows, int D, int ldx, int ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = D >> 3;
    const bf16_t* xr = x + (size_t)row * ldx;
    u32x4_t v[MAXV];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            v[i] = *(const u32x4_t*)(xr + c * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_lo(v[i][j]), bb = bf_hi(v[i][j]);
                s += a + bb;
                ss += a * a + bb * bb;
            }
        }
    }
    float mean = 0.f, rstd;
    if (RMS) {
        ss = wave_sum(ss);
        rstd = rsqrtf(ss / (float)D + eps);
    } else {
        s = wave_sum(s);
        mean = s / (float)D;
        // two-pass variance for accuracy (values are in registers)
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bf_lo(v[i][j]) - mean, bb = bf_hi(v[i][j]) - mean;
                    vs += a * a + bb * bb;
                }
            }
        }
        vs = wave_sum(vs);
        rstd = rsqrtf(vs / (float)D + eps);
    }
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            const u32x4_t wv = *(const u32x4_t*)(w + c * 8);
            u32x4_t bv = {0u, 0u, 0u, 0u};
            if (!RMS) bv = *(const u32x4_t*)(b + c * 8);
            u32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = (bf_lo(v[i][j]) - mean) * rstd, bb = (bf_hi(v[i][j]) - mean) * rstd;
                if (RMS) {
                    // HF: normalise in fp32, downcast to the activation dtype, then multiply by the weight
                    a = bf2f(f2bf(a)) * bf_lo(wv[j]);
                    bb = bf2f(f2bf(bb)) * bf_hi(wv[j]);
                } else {
                    a = a * bf_lo(wv[j]) + bf_lo(bv[j]);
                    bb = bb * bf_hi(wv[j]) + bf_hi(bv[j]);
                }
                o[j] = pack_bf16x2(a, bb);
            }
            *(u32x4_t*)(yr + c * 8) = o;
        }
    }
}

// RMSNorm of fp32 rows (the fp32 residual stream of the prefill, round 5): HF LlamaRMSNorm on an fp32 hidden state -- statistics and
// normalise in fp32, round to bf16, multiply by the weight, round.  D % 8 == 0, D <= 8 * 64 * MAXV
template <int MAXV>
__global__ __launch_bounds__(256) void emmax_rmsnorm_f32_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, const bf16_t* __restrict__ w,
                                                               int rows, int D, int ldx, int ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = D >> 3;
    const float* xr = x + (size_t)row * ldx;
    f32x8_t v[MAXV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nchunk ? ld_f32x8(xr + c * 8) : f32x8_t{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(f32x8_at(v[i], e), f32x8_at(v[i], e), ss);   // spelled out: the split-K reduce pass with the norm (gemm.hip) must round identically
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)D + eps);
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c < nchunk) {
            const u32x4_t wv = *(const u32x4_t*)(w + c * 8);
            u32x4_t o;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf2f(f2bf(f32x8_at(v[i], 2 * j) * rstd)) * bf_lo(wv[j]);
                const float bb = bf2f(f2bf(f32x8_at(v[i], 2 * j + 1) * rstd)) * bf_hi(wv[j]);
                o[j] = pack_bf16x2(a, bb);
            }
            *(u32x4_t*)(yr + c * 8) = o;
        }
    }
}

// LayerNorm statistics of every row, nothing else, the row read once.  Round 6: a wave takes R rows per pass with all of their loads issued
// before the first reduction, and walks the rows grid-stride from a grid that fits the chip -- one row per wave and one block per four rows
// (16704 blocks of ~3 us of life at 256 frames) ran at 0.85 TB/s: 160 us per launch, 98 launches per frame batch = 15 % of the ViT at B = 256.
// Per row the same operations in the same order as before: the statistics are bit-identical.
template <int MAXV, int R>
__global__ __launch_bounds__(256) void emmax_row_stats_kernel(const bf16_t* __restrict__ x, float* __restrict__ stats, int rows, int D, int ldx, float eps) {
    const int lane = threadIdx.x & 63;
    const int nwave = gridDim.x * 4;
    const int nchunk = D >> 3;
    for (int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * R; r0 < rows; r0 += nwave * R) {
        u32x4_t v[R][MAXV];
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            const bf16_t* xr = x + (size_t)min(r0 + rr, rows - 1) * ldx;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                const int c = lane + 64 * i;
                v[rr][i] = (u32x4_t){0u, 0u, 0u, 0u};
                if (c < nchunk) v[rr][i] = *(const u32x4_t*)(xr + c * 8);
            }
        }
#pragma unroll
        for (int rr = 0; rr < R; ++rr) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (lane + 64 * i < nchunk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) s += bf_lo(v[rr][i][j]) + bf_hi(v[rr][i][j]);
                }
            }
            const float mean = wave_sum(s) / (float)D;
            float vs = 0.f;
#pragma unroll
            for (int i = 0; i < MAXV; ++i) {
                if (lane + 64 * i < nchunk) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float a = bf_lo(v[rr][i][j]) - mean, bb = bf_hi(v[rr][i][j]) - mean;
                        vs += a * a + bb * bb;
                    }
                }
            }
            vs = wave_sum(vs);
            if (lane == 0 && r0 + rr < rows) *(f32x2_t*)(stats + (size_t)(r0 + rr) * 2) = (f32x2_t){mean, rsqrtf(vs / (float)D + eps)};
        }
    }
}

// one block per weight row n: W'[n, :] = bf16(W[n, :] .* gamma), ln_s[n] = sum W'[n, :], ln_c[n] = sum W[n, :] .* beta + bias[n]
__global__ __launch_bounds__(256) void emmax_ln_fold_kernel(bf16_t* __restrict__ W, int ldw, int K, const bf16_t* __restrict__ gamma,
                                                           const bf16_t* __restrict__ beta, const bf16_t* __restrict__ bias,
                                                           float* __restrict__ ln_s, float* __restrict__ ln_c) {
    const int n = blockIdx.x;
    bf16_t* w = W + (size_t)n * ldw;
    float s = 0.f, c = 0.f;
    for (int k = threadIdx.x; k < K; k += 256) {
        const float wv = bf2f(w[k]);
        const bf16_t wf = f2bf(wv * bf2f(gamma[k]));
        c += wv * (beta ? bf2f(beta[k]) : 0.f);
        s += bf2f(wf);
        w[k] = wf;
    }
    __shared__ float rs[4], rc[4];
    s = wave_sum(s);
    c = wave_sum(c);
    if ((threadIdx.x & 63) == 0) { rs[threadIdx.x >> 6] = s; rc[threadIdx.x >> 6] = c; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ln_s[n] = rs[0] + rs[1] + rs[2] + rs[3];
        ln_c[n] = rc[0] + rc[1] + rc[2] + rc[3] + (bias ? bf2f(bias[n]) : 0.f);
    }
}

}  // namespace

int launch_row_stats(const void* x, float* stats, int rows, int D, int ldx, float eps, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (D % 8 != 0 || D > 8 * 64 * 16 || ldx % 8) return -1;
    dim3 block(256);
    const int nv = cdiv(D / 8, 64);
    // R rows per wave and pass (<= 8 sixteen-byte loads per lane in flight); at most 8 blocks per CU
#define LAUNCH(MAXV, R) hipLaunchKernelGGL((emmax_row_stats_kernel<MAXV, R>), dim3(min(cdiv(rows, 4 * R), 2048)), block, 0, stream, (const bf16_t*)x, stats, rows, D, ldx, eps)
    if (nv <= 1) LAUNCH(1, 4);
    else if (nv <= 2) LAUNCH(2, 4);
    else if (nv <= 4) LAUNCH(4, 2);
    else if (nv <= 8) LAUNCH(8, 1);
    else LAUNCH(16, 1);
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_ln_fold(void* W, int ldw, int N, int K, const void* gamma, const void* beta, const void* bias, float* ln_s, float* ln_c,
                   hipStream_t stream) {
    if (N <= 0 || K <= 0 || !gamma || !ln_s || !ln_c) return -1;
    hipLaunchKernelGGL(emmax_ln_fold_kernel, dim3(N), dim3(256), 0, stream, (bf16_t*)W, ldw, K, (const bf16_t*)gamma, (const bf16_t*)beta,
                       (const bf16_t*)bias, ln_s, ln_c);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

static int launch_rownorm(bool rms, const void* x, void* y, const void* w, const void* b, int rows, int D, int ldx, int ldy,
                          float eps, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (D % 8 != 0 || D > 8 * 64 * 16) return -1;
    dim3 grid(cdiv(rows, 4)), block(256);
    const int nv = cdiv(D / 8, 64);
#define LAUNCH(MAXV)                                                                                                   \
    do {                                                                                                               \
        if (rms)                                                                                                       \
            hipLaunchKernelGGL((emmax_rownorm_kernel<MAXV, true>), grid, block, 0, stream, (const bf16_t*)x, (bf16_t*)y, \
                               (const bf16_t*)w, (const bf16_t*)b, rows, D, ldx, ldy, eps);                            \
        else                                                                                                           \
            hipLaunchKernelGGL((emmax_rownorm_kernel<MAXV, false>), grid, block, 0, stream, (const bf16_t*)x,          \
                               (bf16_t*)y, (const bf16_t*)w, (const bf16_t*)b, rows, D, ldx, ldy, eps);                \
    } while (0)
    if (nv <= 1) LAUNCH(1);
    else if (nv <= 2) LAUNCH(2);
    else if (nv <= 4) LAUNCH(4);
    else if (nv <= 8) LAUNCH(8);
    else LAUNCH(16);
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_layernorm(const void* x, void* y, const void* w, const void* b, int rows, int D, int ldx, int ldy, float eps,
                     hipStream_t stream) {
    return launch_rownorm(false, x, y, w, b, rows, D, ldx, ldy, eps, stream);
}
int launch_rmsnorm(const void* x, void* y, const void* w, int rows, int D, int ldx, int ldy, float eps, hipStream_t stream) {
    return launch_rownorm(true, x, y, w, nullptr, rows, D, ldx, ldy, eps, stream);
}
int launch_rmsnorm_f32(const float* x, void* y, const void* w, int rows, int D, int ldx, int ldy, float eps, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (D % 8 != 0 || D > 8 * 64 * 16 || ldx % 4 || ldy % 8) return -1;
    dim3 grid(cdiv(rows, 4)), block(256);
    const int nv = cdiv(D / 8, 64);
#define LAUNCH(MAXV) hipLaunchKernelGGL((emmax_rmsnorm_f32_kernel<MAXV>), grid, block, 0, stream, x, (bf16_t*)y, (const bf16_t*)w, rows, D, ldx, ldy, eps)
    if (nv <= 1) LAUNCH(1);
    else if (nv <= 2) LAUNCH(2);
    else if (nv <= 4) LAUNCH(4);
    else if (nv <= 8) LAUNCH(8);
    else LAUNCH(16);
#undef LAUNCH
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
