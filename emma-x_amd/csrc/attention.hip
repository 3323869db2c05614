// attention.hip -- MFMA flash attention over a packed qkv buffer, used for
//   * the ViT blocks (bidirectional, N = 261 / 256, head_dim 64 / 72)   = timm `Attention` (F.scaled_dot_product_attention)
//   * the LLaMA prefill (causal, head_dim 128, GQA-generic)              = HF `LlamaAttention` at q_len > 1
// Both are called from prismatic/extern/hf/modeling_prismatic.py:121 and :404-415 in the reference.
//
// "Swapped" formulation so the softmax statistics are lane-local and P never leaves registers:
//   S^T = K . Q^T   (A = K tile from LDS, B = Q^T kept in registers)      -> lane (g,c) holds S^T[key = 16t+4g+r][q = c]
//   O^T = V^T . P^T (A = V^T tile from LDS, B = P^T = the lane's own S^T registers, converted to bf16)
// The 32-wide MFMA reduction index is free to be any bijection onto the 32 keys of a half tile as long as A and B use
// the same one; we pick  slot (g,j) -> key 16*t0 + 4g + j (j<4), 16*t1 + 4g + (j-4) (j>=4),  which is exactly what a
// lane already holds after the QK^T step.  V is transposed while it is staged into LDS (2-byte scatter), K is staged
// row-major with an odd 16-byte-slot row pitch (conflict-free ds_read_b128 over 16 rows).
// Softmax runs in fp32 with the online (running max / running sum) recurrence; masked keys get -inf.
#include "common.h"
#include "kernels.h"

namespace {

template <int HD>
struct AttnDims {
    static constexpr int HDK = (HD + 31) / 32 * 32;   // QK^T reduction length, zero padded
    static constexpr int HDV = (HD + 15) / 16 * 16;   // PV output rows, zero padded
    static constexpr int KPITCH = HDK + 8;            // bf16 elements; (KPITCH*2)/16 is odd for HDK in {64,96,128}
    static constexpr int VPITCH = 64 + 4;             // bf16 elements per V^T row (64 keys + pad)
    static constexpr int CH = HD / 8;                 // 16-byte chunks per K/V row
};

template <int HD>
__global__ __launch_bounds__(256) void emmax_attention_kernel(AttnParams p) {
    using DM = AttnDims<HD>;
    constexpr int HDK = DM::HDK, HDV = DM::HDV, KP = DM::KPITCH, VP = DM::VPITCH, CH = DM::CH;
    constexpr int NKK = HDK / 32, NDT = HDV / 16;
    __shared__ __attribute__((aligned(16))) bf16_t sK[64 * KP];
    __shared__ __attribute__((aligned(16))) bf16_t sVt[HDV * VP];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, c = lane & 15;
    const int b = blockIdx.z, h = blockIdx.y, qt = blockIdx.x;
    const int start = p.cu_seqlens[b];
    const int len = p.cu_seqlens[b + 1] - start;
    const int q_tile0 = qt * 64;
    if (q_tile0 >= len) return;
    const int hk = h / (p.Hq / p.Hkv);
    const bf16_t* __restrict__ base = (const bf16_t*)p.qkv + (size_t)start * p.ld_qkv;
    const bf16_t* qptr = base + p.q_off + h * HD;
    const bf16_t* kptr = base + p.k_off + hk * HD;
    const bf16_t* vptr = base + p.v_off + hk * HD;

    // zero the padding (columns HD..HDK of K rows, rows HD..HDV of V^T): never overwritten by the tile loads
    if (HDK > HD) {
        for (int i = tid; i < 64 * (HDK - HD); i += 256) sK[(i / (HDK - HD)) * KP + HD + (i % (HDK - HD))] = 0;
    }
    if (HDV > HD) {
        for (int i = tid; i < (HDV - HD) * VP; i += 256) sVt[HD * VP + i] = 0;
    }

    // Q^T fragments (B operand): lane (g,c) holds Q[q_base + c][kk*32 + g*8 .. +8]
    const int q_row = q_tile0 + wave * 16 + c;
    bf16x8_t qf[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) {
        const int k0 = kk * 32 + g * 8;
        u32x4_t v = {0u, 0u, 0u, 0u};
        if (q_row < len && k0 < HD) v = *(const u32x4_t*)(qptr + (size_t)q_row * p.ld_qkv + k0);
        qf[kk] = __builtin_bit_cast(bf16x8_t, v);
    }

    f32x4_t o[NDT];
#pragma unroll
    for (int i = 0; i < NDT; ++i) o[i] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    float m_run = -INFINITY, l_run = 0.f;

    int kv_end = len;
    if (p.causal) kv_end = min(len, q_tile0 + 64);
    const float scale = p.scale;

    for (int kv0 = 0; kv0 < kv_end; kv0 += 64) {
        __syncthreads();  // previous tile consumed (also orders the pad zero-fill before first use)
        // ---- stage K (row-major) and V (transposed) ----
        for (int ci = tid; ci < 64 * CH; ci += 256) {
            const int key = ci / CH, ch = ci - key * CH;
            const int kg = kv0 + key;
            u32x4_t kvv = {0u, 0u, 0u, 0u}, vv = {0u, 0u, 0u, 0u};
            if (kg < len) {
                kvv = *(const u32x4_t*)(kptr + (size_t)kg * p.ld_qkv + ch * 8);
                vv = *(const u32x4_t*)(vptr + (size_t)kg * p.ld_qkv + ch * 8);
            }
            *(u32x4_t*)(&sK[key * KP + ch * 8]) = kvv;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                sVt[(ch * 8 + 2 * j) * VP + key] = (bf16_t)(vv[j] & 0xffffu);
                sVt[(ch * 8 + 2 * j + 1) * VP + key] = (bf16_t)(vv[j] >> 16);
            }
        }
        __syncthreads();

        // ---- S^T = K . Q^T : 4 key tiles of 16 ----
        f32x4_t st[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            st[t] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8_t kf = *(const bf16x8_t*)(&sK[(t * 16 + c) * KP + kk * 32 + g * 8]);
                st[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[kk], st[t], 0, 0, 0);
            }
        }
        // ---- mask + online softmax (per query = per lane column c; rows spread over r, t and the 4 lane groups) ----
        float m_tile = -INFINITY;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = kv0 + t * 16 + g * 4 + r;
                float s = st[t][r] * scale;
                const bool ok = (key < len) && (!p.causal || key <= q_row);
                s = ok ? s : -INFINITY;
                st[t][r] = s;
                m_tile = fmaxf(m_tile, s);
            }
        m_tile = rows_max(m_tile);
        const float m_new = fmaxf(m_run, m_tile);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __expf(m_run - m_safe);   // m_run = -inf -> 0
        float psum = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float pv = __expf(st[t][r] - m_safe);
                st[t][r] = pv;
                psum += pv;
            }
        psum = rows_sum(psum);
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int i = 0; i < NDT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[i][r] *= alpha;

        // ---- O^T += V^T . P^T over the two 32-key halves ----
#pragma unroll
        for (int hh = 0; hh < 2; ++hh) {
            const int t0 = 2 * hh, t1 = 2 * hh + 1;
            u32x4_t pp;
            pp[0] = pack_bf16x2(st[t0][0], st[t0][1]);
            pp[1] = pack_bf16x2(st[t0][2], st[t0][3]);
            pp[2] = pack_bf16x2(st[t1][0], st[t1][1]);
            pp[3] = pack_bf16x2(st[t1][2], st[t1][3]);
            const bf16x8_t pf = __builtin_bit_cast(bf16x8_t, pp);
#pragma unroll
            for (int dt = 0; dt < NDT; ++dt) {
                const bf16_t* vrow = &sVt[(dt * 16 + c) * VP];
                const u32x2_t a0 = *(const u32x2_t*)(vrow + t0 * 16 + g * 4);
                const u32x2_t a1 = *(const u32x2_t*)(vrow + t1 * 16 + g * 4);
                const u32x4_t av = {a0[0], a0[1], a1[0], a1[1]};
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, av), pf, o[dt], 0, 0, 0);
            }
        }
    }

    // ---- write O[q][d]: lane (g,c) holds d = dt*16 + g*4 + r of query c ----
    if (q_row < len) {
        const float inv = l_run > 0.f ? 1.0f / l_run : 0.f;
        bf16_t* orow = (bf16_t*)p.out + (size_t)(start + q_row) * p.ld_out + h * HD;
#pragma unroll
        for (int dt = 0; dt < NDT; ++dt) {
            const int d0 = dt * 16 + g * 4;
            if (d0 < HD) {
                u32x2_t w;
                w[0] = pack_bf16x2(o[dt][0] * inv, o[dt][1] * inv);
                w[1] = pack_bf16x2(o[dt][2] * inv, o[dt][3] * inv);
                *(u32x2_t*)(orow + d0) = w;
            }
        }
    }
}

}  // namespace

int launch_attention(const AttnParams& p, int head_dim, hipStream_t stream) {
    if (p.B <= 0 || p.max_seqlen <= 0) return 0;
    if (p.Hq % p.Hkv != 0) return -1;
    if ((p.ld_qkv % 8) || (p.q_off % 8) || (p.k_off % 8) || (p.v_off % 8) || (p.ld_out % 4)) return -1;
    dim3 grid(cdiv(p.max_seqlen, 64), p.Hq, p.B), block(256);
    switch (head_dim) {
        case 64: hipLaunchKernelGGL(emmax_attention_kernel<64>, grid, block, 0, stream, p); break;
        case 72: hipLaunchKernelGGL(emmax_attention_kernel<72>, grid, block, 0, stream, p); break;
        case 128: hipLaunchKernelGGL(emmax_attention_kernel<128>, grid, block, 0, stream, p); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
