// gemm.hip -- bf16 MFMA GEMM with fused epilogues for the dense (MFMA-bound) stages of the hot path:
//   ViT qkv / proj / fc1 / fc2, patch-embed (im2col), projector fc1-3, LLaMA prefill qkv / o / gate-up / down.
// Replaces the torch `F.linear` calls made by timm `Attention`/`Mlp`, `PrismaticProjector.forward`
// (prismatic/extern/hf/modeling_prismatic.py:146-158) and HF `LlamaAttention`/`LlamaMLP` at q_len > 1.
//
//   C[M,N] = epi(A[M,K] . W[N,K]^T)    A, W bf16 row-major (K contiguous), fp32 accumulate.
//
// Tiling: 128x128x64 block tile, 256 threads = 4 waves in 2x2, each wave 64x64 = 4x4 v_mfma_f32_16x16x32_bf16 tiles.
// Global -> registers -> LDS staging with the next tile's loads in flight during the MFMAs of the current one.
// LDS tile = [128 rows][8 chunks of 16 B]; chunk index XOR-swizzled with (row>>1)&7 so that every ds_read_b128 lane
// group (16 lanes: 16 rows at one logical chunk) lands on 16 distinct 16-B slots of the 256-B bank row.
#include "common.h"
#include "kernels.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

__device__ __forceinline__ int lds_off(int row, int chunk) { return row * (BK * 2) + ((chunk ^ ((row >> 1) & 7)) << 4); }

template <int ACT, bool OUT_F32>
__global__ __launch_bounds__(256) void emmax_gemm_bf16_kernel(GemmParams p) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * TILE_BYTES];
    unsigned char* sA = smem;
    unsigned char* sB = smem + TILE_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, li = lane & 15;

    // XCD-aware tile order: consecutive block ids are dispatched round-robin over the 8 XCDs; give each XCD a
    // contiguous run of tiles that walk M fastest so neighbours share the W panel in that XCD's L2.
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = p.N / BN;
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid % tiles_m, tn = bid / tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const bf16_t* __restrict__ A = (const bf16_t*)p.A;
    const bf16_t* __restrict__ W = (const bf16_t*)p.W;

    // staging map: thread -> (row = tid/8 + 32*i, chunk = tid%8), i = 0..3
    const int srow = tid >> 3, schunk = tid & 7;
    u32x4_t ra[4], rb[4];
    const u32x4_t zero4 = {0u, 0u, 0u, 0u};

    auto load_tiles = [&](int k0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            const int gm = m0 + row;
            ra[i] = (gm < p.M) ? *(const u32x4_t*)(A + (size_t)gm * p.lda + k0 + schunk * 8) : zero4;
            rb[i] = *(const u32x4_t*)(W + (size_t)(n0 + row) * p.ldw + k0 + schunk * 8);
        }
    };
    auto store_tiles = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = srow + 32 * i;
            *(u32x4_t*)(sA + lds_off(row, schunk)) = ra[i];
            *(u32x4_t*)(sB + lds_off(row, schunk)) = rb[i];
        }
    };

    f32x4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = p.K / BK;
    load_tiles(0);
    for (int kt = 0; kt < nk; ++kt) {
        __syncthreads();  // previous tile fully consumed
        store_tiles();
        __syncthreads();
        if (kt + 1 < nk) load_tiles((kt + 1) * BK);  // in flight during the MFMAs below
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8_t fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                fa[i] = *(const bf16x8_t*)(sA + lds_off(wm * 64 + i * 16 + li, kk * 4 + g));
                fb[i] = *(const bf16x8_t*)(sB + lds_off(wn * 64 + i * 16 + li, kk * 4 + g));
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
        }
    }

    // ---- epilogue: C/D layout of 16x16 MFMA: col = lane&15, row = (lane>>4)*4 + reg ----
    const bf16_t* bias = (const bf16_t*)p.bias;
    const bf16_t* scale = (const bf16_t*)p.scale;
    const bf16_t* res = (const bf16_t*)p.residual;
    if (ACT == 2) {
        // SwiGLU: 16-column groups alternate (gate, up); output column = (col/32)*16 + col%16.
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int col = n0 + wn * 64 + j * 16 + li;
                const int ocol = (col >> 5) * 16 + (col & 15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = m0 + wm * 64 + i * 16 + g * 4 + r;
                    if (row < p.M) {
                        const float v = silu(acc[i][j][r]) * acc[i][j + 1][r];
                        ((bf16_t*)p.C)[(size_t)row * p.ldc + ocol] = f2bf(v);
                    }
                }
            }
        return;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + li;
        const bool col_ok = col < p.N_store;
        const float bv = bias ? bf2f(bias[col]) : 0.f;
        const float sv = scale ? bf2f(scale[col]) : 1.f;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = m0 + wm * 64 + i * 16 + g * 4 + r;
                if (row < p.M && col_ok) {
                    float v = acc[i][j][r] + bv;
                    if (ACT == 1) v = gelu_erf(v);
                    v *= sv;
                    if (res) v += bf2f(res[(size_t)row * p.ldr + col]);
                    if (OUT_F32)
                        ((float*)p.C)[(size_t)row * p.ldc + col] = v;
                    else
                        ((bf16_t*)p.C)[(size_t)row * p.ldc + col] = f2bf(v);
                }
            }
    }
}

}  // namespace

int launch_gemm(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0) return 0;
    {
        // The 256x256 kernel runs one tile per CU at ~1.3x the per-CU rate of this one: it wins whenever its last (or
        // only) round keeps >= 70 % of the 256 CUs busy, and from 112 tiles up when there is a single round (measured
        // cross-over on the prefill shapes, tools/gemm_bench.py).  EMMAX_GEMM256=0/1 forces a path.
        static const int force = getenv("EMMAX_GEMM256") ? atoi(getenv("EMMAX_GEMM256")) : -1;
        const int t = gemm256_tiles(p);
        const bool big = t <= 256 ? t >= 112 : 10 * t >= 7 * 256 * cdiv(t, 256);
        if (force == 1 || (force != 0 && big)) return launch_gemm256(p, stream);
    }
    if (p.K % BK != 0 || p.N % BN != 0 || p.K <= 0 || p.N <= 0) return -1;
    if ((p.lda % 8) || (p.ldw % 8)) return -1;
    const int tiles = cdiv(p.M, BM) * (p.N / BN);
    dim3 grid(tiles), block(256);
    if (p.act == 2) {
        if (p.out_f32) return -1;
        hipLaunchKernelGGL((emmax_gemm_bf16_kernel<2, false>), grid, block, 0, stream, p);
    } else if (p.act == 1) {
        if (p.out_f32)
            hipLaunchKernelGGL((emmax_gemm_bf16_kernel<1, true>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((emmax_gemm_bf16_kernel<1, false>), grid, block, 0, stream, p);
    } else {
        if (p.out_f32)
            hipLaunchKernelGGL((emmax_gemm_bf16_kernel<0, true>), grid, block, 0, stream, p);
        else
            hipLaunchKernelGGL((emmax_gemm_bf16_kernel<0, false>), grid, block, 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
