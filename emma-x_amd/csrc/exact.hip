// exact.hip -- the kernels only the EXACT-NUMERICS mode runs (round 6, tuning switch `exact`).
//
// The reference's CPU path is fp32 end to end (prismatic/models/vlms/prismatic.py:659-663 under BASELINE configs[0]); the default HIP path
// rounds every MFMA / dot2 operand to bf16 and sits 2.4e-2 of max|logit| away from it at full depth.  In exact mode every ACTIVATION
// between kernels is fp32, and wherever a bf16 MFMA / dot2 consumes one it is split into two bf16 terms x = hi + lo (16 mantissa bits;
// the weights are exact bf16 already) that go through the same instruction into the same fp32 accumulator (gemm.hip: GemmParams::a_hl;
// decode_ks.hip: EX).  What has no weight operand -- attention -- runs on the fp32 MFMA (v_mfma_f32_32x32x2_f32) below.
//
//   emmax_x_rows_hl_kernel     fp32 rows -> "HL rows" (per 64 elements: 64 hi bf16, 64 lo bf16), optionally through RMSNorm (HF
//                              LlamaRMSNorm on an fp32 hidden state) or LayerNorm (timm Block.norm1 / norm2) first
//   emmax_x_patch_gather       uint8 frame -> normalised fp32 -> HL im2col rows (processing_prismatic.py:136-143 + timm PatchEmbed)
//   emmax_x_assemble_tokens    fp32 token rows: patch embedding + pos_embed, cls / register prefix (timm _pos_embed)
//   emmax_x_embed_splice       fp32 residual rows of the prefill: embedding rows (exact) | fp32 patch embeddings (modeling_prismatic.py:380-385)
//   emmax_x_rope_kv_write      fp32 RoPE on q / k in place + fp32 K / V append to the paged cache (HF apply_rotary_pos_emb + cache update)
//   emmax_x_attention_kernel   flash attention on fp32 MFMA, fp32 q / k / v in, HL rows out (timm Attention / HF SDPA,
//                              modeling_prismatic.py:114-123,404-415)
#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// ---------------------------------------------------------------------------------------------------------------------
// rows: one wave per row, 8 elements per lane and pass.  MODE 1: RMSNorm (w); 2: LayerNorm (w, b).
// y: HL rows of pitch ldy bf16 elements (>= 2 * Dp); columns [D, Dp) are written as zeros.
// ---------------------------------------------------------------------------------------------------------------------
template <int MAXV, int MODE>
__global__ __launch_bounds__(256) void emmax_x_rows_hl_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, const bf16_t* __restrict__ w,
                                                             const bf16_t* __restrict__ b, int rows, int D, int Dp, int ldx, int ldy, float eps) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int nchunk = D >> 3, nchunk_p = Dp >> 3;
    const float* xr = x + (size_t)row * ldx;
    f32x8_t v[MAXV];
    float s = 0.f, ss = 0.f;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < nchunk ? ld_f32x8(xr + c * 8) : f32x8_t{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            s += f32x8_at(v[i], e);
            ss = __builtin_fmaf(f32x8_at(v[i], e), f32x8_at(v[i], e), ss);
        }
    }
    float mean = 0.f, rstd = 1.f;
    if (MODE == 1) {
        rstd = rsqrtf(wave_sum(ss) / (float)D + eps);
    } else if (MODE == 2) {
        mean = wave_sum(s) / (float)D;
        float vs = 0.f;
#pragma unroll
        for (int i = 0; i < MAXV; ++i) {
            const int c = lane + 64 * i;
            if (c < nchunk) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float a = f32x8_at(v[i], e) - mean;
                    vs = __builtin_fmaf(a, a, vs);
                }
            }
        }
        rstd = rsqrtf(wave_sum(vs) / (float)D + eps);
    }
    bf16_t* yr = y + (size_t)row * ldy;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
        const int c = lane + 64 * i;
        if (c >= nchunk_p) continue;
        u32x4_t hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
        if (c < nchunk) {
            u32x4_t wv = {0u, 0u, 0u, 0u}, bv = {0u, 0u, 0u, 0u};
            if (MODE >= 1) wv = *(const u32x4_t*)(w + c * 8);
            if (MODE == 2) bv = *(const u32x4_t*)(b + c * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float a = f32x8_at(v[i], 2 * j), bb = f32x8_at(v[i], 2 * j + 1);
                if (MODE == 1) {          // HF LlamaRMSNorm in fp32: weight * (x * rsqrt(var + eps))
                    a = bf_lo(wv[j]) * (a * rstd);
                    bb = bf_hi(wv[j]) * (bb * rstd);
                } else if (MODE == 2) {   // F.layer_norm in fp32
                    a = (a - mean) * rstd * bf_lo(wv[j]) + bf_lo(bv[j]);
                    bb = (bb - mean) * rstd * bf_hi(wv[j]) + bf_hi(bv[j]);
                }
                { const hl2_t t = split_hl2(a, bb); hi[j] = t.hi; lo[j] = t.lo; }
            }
        }
        bf16_t* o = yr + hl_col(c * 8);
        *(u32x4_t*)o = hi;
        *(u32x4_t*)(o + 64) = lo;
    }
}

// plain split of fp32 rows of any width: one thread per 8 elements (the SwiGLU product of the prefill is 11008 wide)
__global__ __launch_bounds__(256) void emmax_x_split_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int rows, int D, int Dp, int ldx, int ldy) {
    const int nc = Dp >> 3;
    const size_t total = (size_t)rows * nc;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / nc), c = (int)(i - (size_t)row * nc);
        u32x4_t hi = {0u, 0u, 0u, 0u}, lo = {0u, 0u, 0u, 0u};
        if (c * 8 < D) {
            const f32x8_t v = ld_f32x8(x + (size_t)row * ldx + c * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const hl2_t t = split_hl2(f32x8_at(v, 2 * j), f32x8_at(v, 2 * j + 1)); hi[j] = t.hi; lo[j] = t.lo; }
        }
        bf16_t* o = y + (size_t)row * ldy + hl_col(c * 8);
        *(u32x4_t*)o = hi;
        *(u32x4_t*)(o + 64) = lo;
    }
}

// HL rows -> fp32 rows, y = hi + lo (the single-kernel test entry points hand results back as fp32)
__global__ __launch_bounds__(256) void emmax_x_join_kernel(const bf16_t* __restrict__ x, float* __restrict__ y, int rows, int D, int ldx, int ldy) {
    const size_t total = (size_t)rows * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / D), c = (int)(i - (size_t)row * D);
        const bf16_t* p = x + (size_t)row * ldx + hl_col(c);
        y[(size_t)row * ldy + c] = bf2f(p[0]) + bf2f(p[64]);
    }
}

// fp32 rows -> bf16 rows (API outputs of an exact session: patch embeddings / features as the C ABI hands them out)
__global__ __launch_bounds__(256) void emmax_x_to_bf16_kernel(const float* __restrict__ x, int ldx, bf16_t* __restrict__ y, int ldy, int rows, int D) {
    const size_t total = (size_t)rows * (D >> 3);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / (D >> 3)), c = (int)(i - (size_t)row * (D >> 3));
        *(u32x4_t*)(y + (size_t)row * ldy + c * 8) = f32x8_to_bf16(ld_f32x8(x + (size_t)row * ldx + c * 8));
    }
}

// One block per (image, patch): HL row of kpad elements in conv-weight order k = c*P*P + dy*P + dx, normalised in fp32 as torchvision does
template <bool FROM_U8>
__global__ __launch_bounds__(256) void emmax_x_patch_gather_kernel(const void* __restrict__ src, bf16_t* __restrict__ out, int img, int patch, int kpad,
                                                                  int chan0, float m0, float m1, float m2, float s0, float s1, float s2) {
    const int gp = img / patch;
    const int b = blockIdx.x / (gp * gp), pidx = blockIdx.x % (gp * gp);
    const int py = pidx / gp, px = pidx % gp;
    const int pp = patch * patch;
    bf16_t* o = out + (size_t)blockIdx.x * (2 * kpad);
    for (int k = threadIdx.x; k < kpad; k += blockDim.x) {
        float v = 0.f;
        if (k < 3 * pp) {
            const int c = k / pp, rem = k - c * pp, dy = rem / patch, dx = rem - dy * patch;
            const int y = py * patch + dy, x = px * patch + dx;
            if (FROM_U8) {
                const uint8_t u = ((const uint8_t*)src)[(((size_t)b * img + y) * img + x) * 3 + c];
                const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
                v = ((float)u / 255.0f - mean) / sd;
            } else {
                v = bf2f(((const bf16_t*)src)[(((size_t)b * 6 + chan0 + c) * img + y) * img + x]);
            }
        }
        const bf16_t hi = f2bf(v);
        o[hl_col(k)] = hi;
        o[hl_col(k) + 64] = f2bf(v - bf2f(hi));
    }
}

// tokens[b, 0] = cls, tokens[b, 1..n_reg] = reg, tokens[b, n_prefix + p] = pe[b, p] + pos[p]   (fp32 rows of pitch ld)
__global__ __launch_bounds__(256) void emmax_x_assemble_tokens_kernel(const float* __restrict__ pe, const bf16_t* __restrict__ pos, const bf16_t* __restrict__ cls,
                                                                     const bf16_t* __restrict__ reg, float* __restrict__ tokens, int n_patches, int n_prefix,
                                                                     int has_cls, int D, int ld) {
    const int N = n_prefix + n_patches;
    const int b = blockIdx.x / N, t = blockIdx.x % N;
    float* o = tokens + (size_t)blockIdx.x * ld;
    for (int d = threadIdx.x; d < ld; d += blockDim.x) {
        float v = 0.f;
        if (d < D) {
            if (t < n_prefix) v = bf2f((has_cls && t == 0) ? cls[d] : reg[(size_t)(t - has_cls) * D + d]);
            else v = pe[((size_t)b * n_patches + (t - n_prefix)) * ld + d] + bf2f(pos[(size_t)(t - n_prefix) * D + d]);
        }
        o[d] = v;
    }
}

// h32[cu[b] + s] = s==0 ? E[ids[b][0]] : (s <= n_patches ? patches[b][s-1] : E[ids[b][s - n_patches]]); patches fp32 (p32) or bf16 (pbf)
__global__ __launch_bounds__(256) void emmax_x_embed_splice_kernel(const int32_t* __restrict__ ids, int P_max, const int32_t* __restrict__ cu,
                                                                  const bf16_t* __restrict__ E, const float* __restrict__ p32, const bf16_t* __restrict__ pbf,
                                                                  float* __restrict__ h32, int n_patches, int hidden, int vocab) {
    const int b = blockIdx.y, s = blockIdx.x;
    const int start = cu[b], len = cu[b + 1] - start;
    if (s >= len) return;
    float* o = h32 + (size_t)(start + s) * hidden;
    if (s >= 1 && s <= n_patches && p32) {
        const float* src = p32 + ((size_t)b * n_patches + (s - 1)) * hidden;
        for (int c = threadIdx.x; c < hidden / 4; c += blockDim.x) *(f32x4_t*)(o + c * 4) = *(const f32x4_t*)(src + c * 4);
        return;
    }
    const u32x4_t* src;
    if (s >= 1 && s <= n_patches) {
        src = (const u32x4_t*)(pbf + ((size_t)b * n_patches + (s - 1)) * hidden);
    } else {
        int id = ids[(size_t)b * P_max + (s == 0 ? 0 : s - n_patches)];
        id = min(max(id, 0), vocab - 1);
        src = (const u32x4_t*)(E + (size_t)id * hidden);
    }
    for (int c = threadIdx.x; c < hidden / 8; c += blockDim.x) {
        const f32x8_t f = bf16x8_to_f32(src[c]);
        *(f32x4_t*)(o + (size_t)c * 8) = f.lo;
        *(f32x4_t*)(o + (size_t)c * 8 + 4) = f.hi;
    }
}

// Prefill RoPE (rotate-half) on the fp32 q / k of every packed row, in place, + K / V append to the fp32 or 24-bit (kv24 > 0) cache.  One block per row.
__global__ __launch_bounds__(256) void emmax_x_rope_kv_write_kernel(float* __restrict__ qkv, int ld, int q_off, int k_off, int v_off, const int32_t* __restrict__ cu,
                                                                   int B, const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                   void* __restrict__ kcache, void* __restrict__ vcache, long long kv24,
                                                                   const int32_t* __restrict__ page_table, int max_pages, int Hq, int Hkv, int hd, int page) {
    const int row = blockIdx.x;
    int b = 0;
    while (b + 1 < B && row >= cu[b + 1]) ++b;
    const int pos = row - cu[b];
    const int half = hd >> 1;
    float* r = qkv + (size_t)row * ld;
    const float* cs = cos_t + (size_t)pos * half;
    const float* sn = sin_t + (size_t)pos * half;
    const int pg = page_table[(size_t)b * max_pages + pos / page], slot = pos % page;
    auto put = [&](void* base, size_t idx, float v) {
        if (kv24 > 0) {
            const uint32_t u = x24_bits(v);
            ((bf16_t*)base)[idx] = (bf16_t)(u >> 16);
            ((uint8_t*)base + (size_t)kv24 * 2)[idx] = (uint8_t)(u >> 8);
        } else {
            ((float*)base)[idx] = v;
        }
    };
    for (int i = threadIdx.x; i < (Hq + Hkv) * half; i += blockDim.x) {
        const int hh = i / half, d = i - hh * half;
        float* x = (hh < Hq) ? (r + q_off + hh * hd) : (r + k_off + (hh - Hq) * hd);
        const float x0 = x[d], x1 = x[d + half];
        const float c = cs[d], s = sn[d];
        // HF: q * cos + rotate_half(q) * sin, every product rounded on its own (torch fp32), then the sum
        const float y0 = __fadd_rn(__fmul_rn(x0, c), -__fmul_rn(x1, s)), y1 = __fadd_rn(__fmul_rn(x1, c), __fmul_rn(x0, s));
        x[d] = y0;
        x[d + half] = y1;
        if (hh >= Hq) {
            const size_t kc = (((size_t)pg * Hkv + (hh - Hq)) * page + slot) * hd;
            put(kcache, kc + d, y0);
            put(kcache, kc + d + half, y1);
        }
    }
    for (int i = threadIdx.x; i < Hkv * hd; i += blockDim.x) {
        const int hk = i / hd, d = i - hk * hd;
        put(vcache, (((size_t)pg * Hkv + hk) * page + slot) * hd + d, r[v_off + hk * hd + d]);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Flash attention on the fp32 MFMA (v_mfma_f32_32x32x2_f32: 32 x 32 outputs, K = 2 per issue, fp32 operands -- the products and the
// accumulation are the fp32 reference's; 157 TFLOP/s chip-wide, 1/16 of the bf16 pipe: enough for the attention of one frame).
// Swapped form, as attention.hip: S^T = K Q^T, so a lane holds 16 scores of ONE query (row statistics: one permlane32 swap), P stays in
// registers and is the B operand of O^T = V^T P^T with the MFMA's k index mapped onto the two keys the lane halves already hold.
//   block = 4 waves x 32 queries of one (sequence, head); K / V tiles of 32 keys, double-buffered in LDS (global -> registers -> LDS, one
//   barrier per tile); lane (n = lane % 32, h = lane / 32): query n, contraction indices d in [h HD/2, (h + 1) HD/2) of QK^T (16-byte LDS
//   reads of four consecutive steps), keys 8 (i / 4) + 4 h + i % 4 of the tile in score register i.
// q / k / v: fp32 rows of the packed qkv buffer; out: HL rows (the o-proj / proj GEMM's A operand).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int XA_NW = 4, XA_QB = 32 * XA_NW, XA_KT = 32;
template <int HD> struct XACfg {
    static constexpr int HH = HD / 2;                 // contraction indices per lane half
    static constexpr int PK = HD + 4;                 // K row pitch in floats: 16-byte aligned rows, odd in 16-byte chunks mod 16 -> conflict-free b128 reads
    static constexpr int DB = (HD + 31) / 32;         // 32-wide d blocks of the PV product
    static constexpr int PV = DB * 32;                // V row pitch in floats (zero padded)
    static constexpr int CPR = HD / 4;                // 16-byte chunks per K / V row
    static constexpr int NCH = (XA_KT * CPR + 255) / 256;
    static constexpr int SMEM = 2 * XA_KT * (PK + PV) * 4;
    static_assert(HH % 4 == 0, "16-byte fragment reads");
};

template <int HD>
__global__ __launch_bounds__(256) void emmax_x_attention_kernel(AttnParams p) {
    using C = XACfg<HD>;
    extern __shared__ __attribute__((aligned(16))) float xa_smem[];
    constexpr int BUF = XA_KT * (C::PK + C::PV);   // floats per buffer: K tile, then V tile
    auto sK = [&](int bf) { return xa_smem + bf * BUF; };
    auto sV = [&](int bf) { return xa_smem + bf * BUF + XA_KT * C::PK; };
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = lane & 31, h = lane >> 5;
    const int nqb = (p.max_seqlen + XA_QB - 1) / XA_QB;
    const int b = blockIdx.x / nqb, qb = blockIdx.x - b * nqb;
    const int head = blockIdx.y, hk = head / (p.Hq / p.Hkv);
    const int row0 = p.cu_seqlens[b], S = p.cu_seqlens[b + 1] - row0;
    const int q0 = qb * XA_QB;
    if (q0 >= S) return;
    const float* base = (const float*)p.qkv;
    const int q_hi = min(q0 + XA_QB, S) - 1;                                   // last query of the block
    const int n_tiles = p.causal ? (q_hi / XA_KT + 1) : (S + XA_KT - 1) / XA_KT;

    // zero padding of the V rows (head_dim 72: columns 72..95), once
    if constexpr (C::PV > HD) {
        constexpr int PADC = C::PV - HD;
        for (int i = tid; i < 2 * XA_KT * PADC; i += 256) {
            const int bf = i / (XA_KT * PADC), r = (i / PADC) % XA_KT, c = i % PADC;
            sV(bf)[r * C::PV + HD + c] = 0.f;
        }
    }
    f32x4_t kreg[C::NCH], vreg[C::NCH];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int c = tid + 256 * i;
            const int r = min(c / C::CPR, XA_KT - 1), cc = c % C::CPR;
            const int key = min(t * XA_KT + r, S - 1);                         // keys past the sequence end: the last row again (masked below)
            const float* src = base + (size_t)(row0 + key) * p.ld_qkv + hk * HD + cc * 4;
            kreg[i] = *(const f32x4_t*)(src + p.k_off);
            vreg[i] = *(const f32x4_t*)(src + p.v_off);
        }
    };
    auto store_tile = [&](int bf) {
#pragma unroll
        for (int i = 0; i < C::NCH; ++i) {
            const int c = tid + 256 * i;
            if (c < XA_KT * C::CPR) {
                const int r = c / C::CPR, cc = c % C::CPR;
                *(f32x4_t*)(sK(bf) + r * C::PK + cc * 4) = kreg[i];
                *(f32x4_t*)(sV(bf) + r * C::PV + cc * 4) = vreg[i];
            }
        }
    };

    // this lane's query: elements h HH .. + HH, pre-multiplied by scale * log2(e) (the softmax runs on exp2)
    const int qw0 = q0 + 32 * wave;                                            // first query of the wave (sequence-relative)
    const bool wave_on = qw0 < S;
    const int qpos = min(qw0 + n, S - 1);
    float q[C::HH];
    {
        const float* qp = base + (size_t)(row0 + qpos) * p.ld_qkv + p.q_off + head * HD + h * C::HH;
        const float sc = p.scale * 1.4426950408889634f;
#pragma unroll
        for (int j = 0; j < C::HH / 4; ++j) {
            const f32x4_t v = *(const f32x4_t*)(qp + 4 * j);
#pragma unroll
            for (int e = 0; e < 4; ++e) q[4 * j + e] = v[e] * sc;
        }
    }
    f32x16_t o[C::DB];
#pragma unroll
    for (int d = 0; d < C::DB; ++d)
#pragma unroll
        for (int i = 0; i < 16; ++i) o[d][i] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int t = 0; t < n_tiles; ++t) {
        const int bf = t & 1;
        if (t + 1 < n_tiles) load_tile(t + 1);
        const int k0 = t * XA_KT;
        if (wave_on && (!p.causal || k0 <= qw0 + 31)) {                        // wave-uniform
            // ---- S^T = K Q^T ----
            f32x16_t s;
#pragma unroll
            for (int i = 0; i < 16; ++i) s[i] = 0.f;
            const float* kp = sK(bf) + n * C::PK + h * C::HH;
#pragma unroll
            for (int j = 0; j < C::HH / 4; ++j) {
                const f32x4_t kf = *(const f32x4_t*)(kp + 4 * j);
#pragma unroll
                for (int e = 0; e < 4; ++e) s = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[e], q[4 * j + e], s, 0, 0, 0);
            }
            // ---- mask (only where the tile meets the sequence end / the causal diagonal), online softmax ----
            if (k0 + XA_KT > S || (p.causal && k0 + XA_KT - 1 > qw0)) {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = k0 + 8 * (i >> 2) + 4 * h + (i & 3);
                    if (key >= S || (p.causal && key > qpos)) s[i] = -INFINITY;
                }
            }
            float mt = s[0];
#pragma unroll
            for (int i = 1; i < 16; ++i) mt = fmaxf(mt, s[i]);
            {
                float a, c;
                swap_halves(mt, a, c);
                mt = fmaxf(a, c);
            }
            const float m_new = fmaxf(m_run, mt);
            const float msafe = m_new == -INFINITY ? 0.f : m_new;
            const float alpha = __builtin_amdgcn_exp2f(m_run - msafe);
            float ls = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                s[i] = __builtin_amdgcn_exp2f(s[i] - msafe);
                ls += s[i];
            }
            l_run = l_run * alpha + ls;
            m_run = m_new;
#pragma unroll
            for (int d = 0; d < C::DB; ++d)
#pragma unroll
                for (int i = 0; i < 16; ++i) o[d][i] *= alpha;
            // ---- O^T += V^T P^T: score register i = keys (8 (i / 4) + i % 4) [lanes 0-31] and that + 4 [lanes 32-63] ----
            const float* vp = sV(bf) + 4 * h * C::PV + n;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float* vr = vp + (8 * (i >> 2) + (i & 3)) * C::PV;
#pragma unroll
                for (int d = 0; d < C::DB; ++d) o[d] = __builtin_amdgcn_mfma_f32_32x32x2f32(vr[d * 32], s[i], o[d], 0, 0, 0);
            }
        }
        if (t + 1 < n_tiles) store_tile(bf ^ 1);
        __syncthreads();
    }
    if (!wave_on) return;
    {
        float a, c;
        swap_halves(l_run, a, c);
        l_run = a + c;
    }
    const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
    if (qw0 + n < S) {
        bf16_t* orow = (bf16_t*)p.out + (size_t)(row0 + qw0 + n) * p.ld_out;
#pragma unroll
        for (int d = 0; d < C::DB; ++d)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const int d0 = d * 32 + 8 * g4 + 4 * h;
                if (d0 < HD) {
                    const hl2_t t0 = split_hl2(o[d][4 * g4] * inv, o[d][4 * g4 + 1] * inv), t1 = split_hl2(o[d][4 * g4 + 2] * inv, o[d][4 * g4 + 3] * inv);
                    const u32x2_t hi = {t0.hi, t1.hi}, lo = {t0.lo, t1.lo};
                    bf16_t* dst = orow + hl_col(head * HD + d0);
                    *(u32x2_t*)dst = hi;
                    *(u32x2_t*)(dst + 64) = lo;
                }
            }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-KV decode attention over the fp32 paged cache: decode.hip's emmax_decode_attn_kernel (16 lanes per key row, online softmax per
// lane group, chunks of keys software-pipelined through two register buffers, the block's partial { o[128], m, l } per (row, head, split)
// merged by the o-proj's prologue) with fp32 q (8 elements per lane), fp32 K / V rows (two 16-byte loads per lane, key and operand) and
// fma dot products -- the arithmetic of HF's eager attention on an fp32 cache.  grid (nsplit, Hkv, B), 256 threads.
// ---------------------------------------------------------------------------------------------------------------------
template <int G, bool X24>
__global__ __launch_bounds__(256) void emmax_x_decode_attn_kernel(DecodeAttnParams p) {
    // KU keys per lane group and chunk, two chunks in flight: 32 KiB of K / V per wave at KU = 4 -- a batch-1 launch is one block per CU (one wave
    // per SIMD: registers are no constraint) and a latency chain of L / (nsplit 16 KU) round trips: KU = 2 took 8.6 us per launch, twice the
    // bf16 kernel's trips for twice its bytes
    constexpr int HD = 128, NW = 4, NT = 256, KU = G <= 2 ? 4 : 2, SP = 512, PSTRIDE = EMMAX_PSTRIDE, CHK = 4 * NW * KU;
    __shared__ int s_pages[SP];
    __shared__ float red_o[NW][G][HD];
    __shared__ float red_ml[NW][G][2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, ch = lane & 15;
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int nsplit = gridDim.x;
    const int ctx_now = p.ctx_len[b];
    const int done_word = *(p.done ? p.done + b : p.ctx_len + b);
    const int row_done = p.done ? done_word : 0;
    const int32_t* ptab = p.page_table + (size_t)b * p.max_pages;
    f32x8_t q[G];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) q[gq] = ld_f32x8((const float*)p.q + (size_t)b * p.ldq + (hk * G + gq) * HD + ch * 8);
    const int pt0 = ptab[min(tid, p.max_pages - 1)], pt1 = ptab[min(tid + NT, p.max_pages - 1)];
    __builtin_amdgcn_sched_barrier(0);
    s_pages[tid] = pt0;
    s_pages[tid + NT] = pt1;
    const int L = ctx_now + 1;
    int kps = (L + nsplit - 1) >> __builtin_ctz(nsplit);
    kps = (kps + 15) & ~15;
    const int k0 = split * kps;
    const int k1 = min(L, k0 + kps);
    const int Hq = p.Hkv * G;
    float* part = p.part + ((size_t)(b * Hq + hk * G) * nsplit + split) * PSTRIDE;
    // one split (p.o_out, batch >= 5 at 32 heads): the block owns the heads of its row, normalises them and writes the fp32 attention row itself -- in
    // place over the row's q values (read into registers above by every thread of the block; other blocks own other columns)
    float* orow = p.o_out ? (float*)p.o_out + (size_t)b * p.ldq + (size_t)hk * G * HD : nullptr;
    if (k0 >= L || row_done) {
        if (orow) {
            __syncthreads();   // (every thread holds its q values)
            for (int i = tid; i < G * HD; i += NT) orow[i] = 0.f;
            return;
        }
        for (int i = tid; i < G * PSTRIDE; i += NT) {
            const int gq = i / PSTRIDE, j = i - gq * PSTRIDE;
            part[(size_t)gq * nsplit * PSTRIDE + j] = (j == HD) ? -INFINITY : 0.f;
        }
        return;
    }
    const float* kc = (const float*)p.kcache;
    const float* vc = (const float*)p.vcache;
    const bf16_t *kh = (const bf16_t*)p.kcache, *vh = (const bf16_t*)p.vcache;                 // X24: the bf16 planes ...
    const uint8_t *ke = (const uint8_t*)p.kcache + (size_t)p.kv24 * 2, *ve = (const uint8_t*)p.vcache + (size_t)p.kv24 * 2;   // ... and the extension planes
    __syncthreads();
    float m[G], l[G], o[G][8];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        m[gq] = -INFINITY;
        l[gq] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[gq][j] = 0.f;
    }
    auto load_chunk = [&](int kb, f32x8_t (&kv)[KU], f32x8_t (&vv)[KU], bool (&ok)[KU]) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int key = kb + u * (4 * NW) + wave * 4 + kg;
            ok[u] = key < k1;
            const int kk = ok[u] ? key : k0;
            const int pg = s_pages[kk >> p.page_shift];
            const size_t off = ((((size_t)pg * p.Hkv + hk) << p.page_shift) + (kk & (p.page - 1))) * HD + ch * 8;
            if constexpr (X24) {   // 16 + 8 bytes per lane, key and operand; carried as raw bits until consume_chunk widens them
                const u32x4_t k16 = __builtin_nontemporal_load((const u32x4_t*)(kh + off)), v16 = __builtin_nontemporal_load((const u32x4_t*)(vh + off));
                const u32x2_t k8 = __builtin_nontemporal_load((const u32x2_t*)(ke + off)), v8 = __builtin_nontemporal_load((const u32x2_t*)(ve + off));
                kv[u] = {__builtin_bit_cast(f32x4_t, k16), (f32x4_t){__uint_as_float(k8[0]), __uint_as_float(k8[1]), 0.f, 0.f}};
                vv[u] = {__builtin_bit_cast(f32x4_t, v16), (f32x4_t){__uint_as_float(v8[0]), __uint_as_float(v8[1]), 0.f, 0.f}};
            } else {
                kv[u] = {__builtin_nontemporal_load((const f32x4_t*)(kc + off)), __builtin_nontemporal_load((const f32x4_t*)(kc + off + 4))};
                vv[u] = {__builtin_nontemporal_load((const f32x4_t*)(vc + off)), __builtin_nontemporal_load((const f32x4_t*)(vc + off + 4))};
            }
        }
    };
    auto consume_chunk = [&](f32x8_t (&kv)[KU], f32x8_t (&vv)[KU], const bool (&ok)[KU]) {
        if constexpr (X24) {
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                float kf[8], vf[8];
                x24_unpack8(__builtin_bit_cast(u32x4_t, kv[u].lo), (u32x2_t){__float_as_uint(kv[u].hi[0]), __float_as_uint(kv[u].hi[1])}, kf);
                x24_unpack8(__builtin_bit_cast(u32x4_t, vv[u].lo), (u32x2_t){__float_as_uint(vv[u].hi[0]), __float_as_uint(vv[u].hi[1])}, vf);
                kv[u] = {(f32x4_t){kf[0], kf[1], kf[2], kf[3]}, (f32x4_t){kf[4], kf[5], kf[6], kf[7]}};
                vv[u] = {(f32x4_t){vf[0], vf[1], vf[2], vf[3]}, (f32x4_t){vf[4], vf[5], vf[6], vf[7]}};
            }
        }
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
            float sc[KU];
            float mc = -INFINITY;
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) s = __builtin_fmaf(f32x8_at(kv[u], e), f32x8_at(q[gq], e), s);
                s = row16_sum(s);
                s = ok[u] ? s * p.scale : -INFINITY;
                sc[u] = s;
                mc = fmaxf(mc, s);
            }
            const float mn = fmaxf(m[gq], mc);
            const float msafe = (mn == -INFINITY) ? 0.f : mn;
            const float alpha = __expf(m[gq] - msafe);
            float ls = l[gq] * alpha;
#pragma unroll
            for (int j = 0; j < 8; ++j) o[gq][j] *= alpha;
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const float pw = __expf(sc[u] - msafe);
                ls += pw;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[gq][j] = __builtin_fmaf(pw, f32x8_at(vv[u], j), o[gq][j]);
            }
            l[gq] = ls;
            m[gq] = mn;
        }
    };
    {
        f32x8_t kvA[KU], vvA[KU], kvB[KU], vvB[KU];
        bool okA[KU], okB[KU];
        load_chunk(k0, kvA, vvA, okA);
        for (int kb = k0; kb < k1; kb += 2 * CHK) {
            const bool hasB = kb + CHK < k1;
            if (hasB) load_chunk(kb + CHK, kvB, vvB, okB);
            consume_chunk(kvA, vvA, okA);
            if (kb + 2 * CHK < k1) load_chunk(kb + 2 * CHK, kvA, vvA, okA);
            if (hasB) consume_chunk(kvB, vvB, okB);
        }
    }
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        const float mw = rows_max(m[gq]);
        const float msafe = (mw == -INFINITY) ? 0.f : mw;
        const float f = __expf(m[gq] - msafe);
        const float lv = rows_sum(l[gq] * f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = rows_sum(o[gq][j] * f);
            if (kg == 0) red_o[wave][gq][ch * 8 + j] = v;
        }
        if (lane == 0) {
            red_ml[wave][gq][0] = mw;
            red_ml[wave][gq][1] = lv;
        }
    }
    __syncthreads();
    if (orow) {   // out = (sum_w o_w e^(m_w - M)) * (1 / sum_w l_w e^(m_w - M)): the arithmetic of the o-proj's one-split merge (attn_merge_chunk)
        for (int i = tid; i < G * HD; i += NT) {
            const int gq = i / HD, j = i - gq * HD;
            float M = red_ml[0][gq][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, red_ml[w][gq][0]);
            const float msafe = (M == -INFINITY) ? 0.f : M;
            float v = 0.f, den = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const float f = __expf(red_ml[w][gq][0] - msafe);
                v += red_o[w][gq][j] * f;
                den += red_ml[w][gq][1] * f;
            }
            orow[i] = v * (den > 0.f ? 1.0f / den : 0.f);
        }
        return;
    }
    for (int i = tid; i < G * PSTRIDE; i += NT) {
        const int gq = i / PSTRIDE, j = i - gq * PSTRIDE;
        float M = red_ml[0][gq][0];
#pragma unroll
        for (int w = 1; w < NW; ++w) M = fmaxf(M, red_ml[w][gq][0]);
        const float msafe = (M == -INFINITY) ? 0.f : M;
        float v = 0.f;
        if (j < HD) {
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red_o[w][gq][j] * __expf(red_ml[w][gq][0] - msafe);
        } else if (j == HD) {
            v = M;
        } else if (j == HD + 1) {
#pragma unroll
            for (int w = 0; w < NW; ++w) v += red_ml[w][gq][1] * __expf(red_ml[w][gq][0] - msafe);
        }
        part[(size_t)gq * nsplit * PSTRIDE + j] = v;
    }
}

template <int HD>
int x_attention_launch(const AttnParams& p, hipStream_t stream) {
    auto kern = emmax_x_attention_kernel<HD>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, XACfg<HD>::SMEM) != hipSuccess) return -4;
        attr_done = true;
    }
    const int nqb = (p.max_seqlen + XA_QB - 1) / XA_QB;
    hipLaunchKernelGGL(kern, dim3(p.B * nqb, p.Hq), dim3(256), XACfg<HD>::SMEM, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

}  // namespace

static int x_rows(int mode, const float* x, void* y, const void* w, const void* b, int rows, int D, int Dp, int ldx, int ldy, float eps, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (D % 8 || Dp % 64 || Dp < D || Dp > 8 * 64 * 16 || ldx % 4 || ldy % 8 || ldy < 2 * Dp) return -1;
    dim3 grid(cdiv(rows, 4)), block(256);
    const int nv = cdiv(Dp / 8, 64);
#define XL(MAXV, MODE) hipLaunchKernelGGL((emmax_x_rows_hl_kernel<MAXV, MODE>), grid, block, 0, stream, x, (bf16_t*)y, (const bf16_t*)w, (const bf16_t*)b, rows, D, Dp, ldx, ldy, eps)
#define XM(MAXV) do { if (mode == 1) XL(MAXV, 1); else XL(MAXV, 2); } while (0)
    if (nv <= 1) XM(1);
    else if (nv <= 2) XM(2);
    else if (nv <= 4) XM(4);
    else if (nv <= 8) XM(8);
    else XM(16);
#undef XM
#undef XL
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_x_split_rows(const float* x, void* y_hl, int rows, int D, int Dp, int ldx, int ldy, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (D % 8 || Dp % 64 || Dp < D || ldx % 4 || ldy % 8 || ldy < 2 * Dp) return -1;
    const size_t total = (size_t)rows * (Dp / 8);
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(emmax_x_split_kernel, dim3(grid), dim3(256), 0, stream, x, (bf16_t*)y_hl, rows, D, Dp, ldx, ldy);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_x_rmsnorm(const float* x, void* y_hl, const void* w, int rows, int D, int ldx, int ldy, float eps, hipStream_t stream) {
    if (D % 64) return -1;
    return x_rows(1, x, y_hl, w, nullptr, rows, D, D, ldx, ldy, eps, stream);
}
int launch_x_layernorm(const float* x, void* y_hl, const void* w, const void* b, int rows, int D, int Dp, int ldx, int ldy, float eps, hipStream_t stream) {
    return x_rows(2, x, y_hl, w, b, rows, D, Dp, ldx, ldy, eps, stream);
}
int launch_x_to_bf16(const float* x, int ldx, void* y, int ldy, int rows, int D, hipStream_t stream) {
    if (rows <= 0) return 0;
    if (D % 8 || ldx % 4 || ldy % 8) return -1;
    const size_t total = (size_t)rows * (D / 8);
    const int grid = (int)((total + 255) / 256 < 2048 ? (total + 255) / 256 : 2048);
    hipLaunchKernelGGL(emmax_x_to_bf16_kernel, dim3(grid), dim3(256), 0, stream, x, ldx, (bf16_t*)y, ldy, rows, D);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_x_patch_gather(bool from_u8, const void* src, void* out_hl, int B, int img, int patch, int kpad, int chan0, const float* mean, const float* std,
                          hipStream_t stream) {
    if (kpad % 64) return -1;
    const int gp = img / patch;
    dim3 grid(B * gp * gp), block(256);
    if (from_u8)
        hipLaunchKernelGGL((emmax_x_patch_gather_kernel<true>), grid, block, 0, stream, src, (bf16_t*)out_hl, img, patch, kpad, chan0, mean[0], mean[1], mean[2],
                           std[0], std[1], std[2]);
    else
        hipLaunchKernelGGL((emmax_x_patch_gather_kernel<false>), grid, block, 0, stream, src, (bf16_t*)out_hl, img, patch, kpad, chan0, mean[0], mean[1], mean[2],
                           std[0], std[1], std[2]);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_x_assemble_tokens(const float* pe, const void* pos, const void* cls, const void* reg, float* tokens, int B, int n_patches, int n_prefix, int has_cls,
                             int D, int ld, hipStream_t stream) {
    hipLaunchKernelGGL(emmax_x_assemble_tokens_kernel, dim3(B * (n_prefix + n_patches)), dim3(256), 0, stream, pe, (const bf16_t*)pos, (const bf16_t*)cls,
                       (const bf16_t*)reg, tokens, n_patches, n_prefix, has_cls, D, ld);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_x_embed_splice(const int32_t* ids, int P_max, const int32_t* cu, const void* E, const float* patches32, const void* patches_bf, float* h32, int B,
                          int max_seqlen, int n_patches, int hidden, int vocab, hipStream_t stream) {
    if (hidden % 8) return -1;
    hipLaunchKernelGGL(emmax_x_embed_splice_kernel, dim3(max_seqlen, B), dim3(256), 0, stream, ids, P_max, cu, (const bf16_t*)E, patches32, (const bf16_t*)patches_bf,
                       h32, n_patches, hidden, vocab);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_x_rope_kv_write(float* qkv, int ld, int q_off, int k_off, int v_off, const int32_t* cu, int B, int total_rows, const float* cos_t, const float* sin_t,
                           void* kcache, void* vcache, long long kv24, const int32_t* page_table, int max_pages, int Hq, int Hkv, int hd, int page, hipStream_t stream) {
    if (total_rows <= 0) return 0;
    if (hd % 4 || ld % 4 || q_off % 4 || k_off % 4 || v_off % 4) return -1;
    hipLaunchKernelGGL(emmax_x_rope_kv_write_kernel, dim3(total_rows), dim3(256), 0, stream, qkv, ld, q_off, k_off, v_off, cu, B, cos_t, sin_t, kcache, vcache, kv24,
                       page_table, max_pages, Hq, Hkv, hd, page);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
// p.qkv: fp32 rows [*, ld_qkv]; p.out: HL rows of pitch ld_out bf16 elements
int launch_x_attention(const AttnParams& p, int head_dim, hipStream_t stream) {
    if (p.B <= 0 || p.max_seqlen <= 0) return 0;
    if (p.Hq % p.Hkv || p.ld_qkv % 4 || p.q_off % 4 || p.k_off % 4 || p.v_off % 4 || p.ld_out % 8) return -1;
    if (head_dim == 64) return x_attention_launch<64>(p, stream);
    if (head_dim == 72) return x_attention_launch<72>(p, stream);
    if (head_dim == 128) return x_attention_launch<128>(p, stream);
    return -1;
}

// fp32 q rows [B, ldq], fp32 K / V caches [pages][Hkv][page][128]; partials as decode.hip's kernel writes them
int launch_x_decode_attn(const DecodeAttnParams& p_in, int B, int Hq, int head_dim, int nsplit, hipStream_t stream) {
    if (head_dim != 128) return -1;
    DecodeAttnParams p = p_in;
    if (p.max_pages < 1 || p.max_pages > 512 || p.page < 1 || (p.page & (p.page - 1))) return -1;
    if (nsplit < 1 || (nsplit & (nsplit - 1)) || p.kv_stage || (p.o_out && nsplit != 1)) return -1;   // o_out (one split only): the fp32 row, else split partials for the o-proj's merge
    p.page_shift = 0;
    while ((1 << p.page_shift) < p.page) ++p.page_shift;
    dim3 grid(nsplit, p.Hkv, B), block(256);
#define XDA(GG) do { if (p.kv24 > 0) hipLaunchKernelGGL((emmax_x_decode_attn_kernel<GG, true>), grid, block, 0, stream, p); \
                     else hipLaunchKernelGGL((emmax_x_decode_attn_kernel<GG, false>), grid, block, 0, stream, p); } while (0)
    switch (Hq / p.Hkv) {
        case 1: XDA(1); break;
        case 2: XDA(2); break;
        case 4: XDA(4); break;
        case 8: XDA(8); break;
        default: return -1;
    }
#undef XDA
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_x_join_rows(const void* x_hl, float* y, int rows, int D, int ldx, int ldy, hipStream_t stream) {
    if (rows <= 0) return 0;
    const size_t total = (size_t)rows * D;
    const int grid = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(emmax_x_join_kernel, dim3(grid), dim3(256), 0, stream, (const bf16_t*)x_hl, y, rows, D, ldx, ldy);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
