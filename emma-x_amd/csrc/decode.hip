// decode.hip -- the autoregressive decode step (HBM-bound): one token per sequence, batch B <= 8.
// Replaces HF `LlamaDecoderLayer` at q_len == 1 + one greedy step of `GenerationMixin.generate`
// (cached branch of prismatic/extern/hf/modeling_prismatic.py:325-341; loop invoked at :519 and
// prismatic/models/vlms/prismatic.py:659-663).
//
// Every projection is a weight-streaming GEMV: weights bf16 [N,K] row-major are read exactly once with 16-byte
// non-temporal loads, many loads in flight per lane, straight to VGPRs (no LDS round trip for data that is not
// shared); the B activation vectors are staged once per block in LDS (bf16) and read back with broadcast-free
// ds_read_b128; products go through v_dot2c_f32_bf16 with fp32 accumulation; rows are reduced across the wave.
// Fusions (no activation round trip through HBM beyond one bf16 vector per stage):
//   qkv    : RMSNorm prologue  -> GEMV -> RoPE (rotate-half) -> q buffer / paged K,V cache append
//   oproj  : GEMV -> + residual (in place)
//   gateup : RMSNorm prologue  -> GEMV over 16-row interleaved (gate,up) -> SiLU(gate)*up
//   down   : GEMV -> + residual (in place)
//   lmhead : final RMSNorm prologue -> GEMV -> per-block greedy argmax (logits never reach HBM)
// Attention: split-KV over the paged cache (16 lanes per key row, 4 keys per wave load), fp32 two-pass softmax inside
// a split, log-sum-exp merge of the splits in a tiny combine kernel.
// All step-varying state (positions, current tokens, done flags) is read from device memory -> hipGraph-capturable.
#include "common.h"
#include "kernels.h"

namespace {

enum { MODE_QKV = 0, MODE_RESID = 1, MODE_GATEUP = 2, MODE_LMHEAD = 3, MODE_PLAIN = 4 };

__device__ __forceinline__ u32x4_t ld_nt(const u32x4_t* p) { return __builtin_nontemporal_load(p); }

// ---------------------------------------------------------------------------------------------------------------------
// GEMV.  Block = 256 threads = 4 waves; a wave owns RPW "row slots"; a slot is one output row (RESID/LMHEAD/PLAIN) or a
// pair of rows (QKV: rows d and d+hd/2 of one head; GATEUP: gate row i and up row i).
// ---------------------------------------------------------------------------------------------------------------------
template <int B, int RPW, int MODE, bool NORM>
__global__ __launch_bounds__(256) void emmax_decode_gemv_kernel(GemvParams p) {
    constexpr int NR = (MODE == MODE_QKV || MODE == MODE_GATEUP) ? 2 * RPW : RPW;   // weight rows per wave
    constexpr int U = 8;   // 16-byte loads per row per outer iteration (8 * 64 lanes * 8 elems = 4096 elements)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4_t* xs = (u32x4_t*)smem;   // [B][KC/8] 16-byte chunks

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* __restrict__ W = (const bf16_t*)p.W;
    const int K = p.K;
    const int slot0 = (blockIdx.x * 4 + wave) * RPW;   // first slot of this wave

    // ---- weight row indices of this wave ----
    int rows[NR];
    bool slot_ok[RPW];
#pragma unroll
    for (int s = 0; s < RPW; ++s) {
        const int slot = slot0 + s;
        slot_ok[s] = slot < p.n_slots;
        const int sl = slot_ok[s] ? slot : 0;
        if (MODE == MODE_QKV) {
            const int half = p.head_dim >> 1;
            const int hb = sl / half, d = sl - hb * half;
            rows[2 * s] = hb * p.head_dim + d;
            rows[2 * s + 1] = hb * p.head_dim + d + half;
        } else if (MODE == MODE_GATEUP) {
            rows[2 * s] = (sl >> 4) * 32 + (sl & 15);
            rows[2 * s + 1] = rows[2 * s] + 16;
        } else {
            rows[s] = sl;
        }
    }

    float acc[NR][B];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[r][b] = 0.f;

    // ---- RMSNorm statistics (prologue) ----
    float rstd[B];
    if (NORM) {
        __shared__ float red[4][B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float ss = 0.f;
            const u32x4_t* xr = (const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx);
            for (int c = tid; c < (K >> 3); c += 256) {
                const u32x4_t v = xr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bf_lo(v[j]), bb = bf_hi(v[j]);
                    ss += a * a + bb * bb;
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) red[wave][b] = ss;
        }
        __syncthreads();
#pragma unroll
        for (int b = 0; b < B; ++b) rstd[b] = rsqrtf((red[0][b] + red[1][b] + red[2][b] + red[3][b]) / (float)K + p.eps);
    }

    const int KC = p.kc;   // elements per K phase (multiple of 8)
    for (int kc0 = 0; kc0 < K; kc0 += KC) {
        const int kcn = min(KC, K - kc0);   // elements in this phase
        const int nch = kcn >> 3;           // 16-byte chunks in this phase
        if (kc0 > 0) __syncthreads();       // previous phase fully consumed
        // ---- stage x[:, kc0 : kc0+kcn] into LDS (normalised if NORM) ----
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const u32x4_t* xr = (const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx + kc0);
            for (int c = tid; c < nch; c += 256) {
                u32x4_t v = xr[c];
                if (NORM) {
                    const u32x4_t wv = *((const u32x4_t*)((const bf16_t*)p.norm_w + kc0) + c);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        // HF LlamaRMSNorm: fp32 normalise -> downcast -> * weight (-> downcast)
                        const float a = bf2f(f2bf(bf_lo(v[j]) * rstd[b])) * bf_lo(wv[j]);
                        const float bb = bf2f(f2bf(bf_hi(v[j]) * rstd[b])) * bf_hi(wv[j]);
                        v[j] = pack_bf16x2(a, bb);
                    }
                }
                xs[b * (KC >> 3) + c] = v;
            }
        }
        __syncthreads();

        // ---- stream the weight rows of this wave over the phase ----
        for (int c0 = 0; c0 < nch; c0 += 64 * U) {
            u32x4_t wr[NR][U];
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const u32x4_t* wrow = (const u32x4_t*)(W + (size_t)rows[r] * p.ldw + kc0);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = c0 + u * 64 + lane;
                    wr[r][u] = (c < nch) ? ld_nt(wrow + c) : (u32x4_t){0u, 0u, 0u, 0u};
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + u * 64 + lane;
                if (c0 + u * 64 < nch) {   // wave-uniform
                    const int cc = (c < nch) ? c : 0;
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        const u32x4_t xv = xs[b * (KC >> 3) + cc];
#pragma unroll
                        for (int r = 0; r < NR; ++r) {
                            float a = acc[r][b];
                            a = dot2_bf16(wr[r][u][0], xv[0], a);
                            a = dot2_bf16(wr[r][u][1], xv[1], a);
                            a = dot2_bf16(wr[r][u][2], xv[2], a);
                            a = dot2_bf16(wr[r][u][3], xv[3], a);
                            acc[r][b] = a;
                        }
                    }
                }
            }
        }
    }

    // ---- wave reduction: every lane ends up with the full sums ----
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[r][b] = wave_sum(acc[r][b]);

    // ---- epilogues ----
    if (MODE == MODE_PLAIN) {
#pragma unroll
        for (int s = 0; s < RPW; ++s)
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (slot_ok[s] && lane == b) ((bf16_t*)p.y)[(size_t)b * p.ldy + rows[s]] = f2bf(acc[s][b]);
    } else if (MODE == MODE_RESID) {
#pragma unroll
        for (int s = 0; s < RPW; ++s)
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (slot_ok[s] && lane == b) {
                    bf16_t* hp = (bf16_t*)p.y + (size_t)b * p.ldy + rows[s];
                    *hp = f2bf(bf2f(*hp) + acc[s][b]);
                }
    } else if (MODE == MODE_GATEUP) {
#pragma unroll
        for (int s = 0; s < RPW; ++s)
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (slot_ok[s] && lane == b)
                    ((bf16_t*)p.y)[(size_t)b * p.ldy + slot0 + s] = f2bf(silu(acc[2 * s][b]) * acc[2 * s + 1][b]);
    } else if (MODE == MODE_QKV) {
        const int hd = p.head_dim, half = hd >> 1;
#pragma unroll
        for (int s = 0; s < RPW; ++s)
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (slot_ok[s] && lane == b) {
                    const int slot = slot0 + s;
                    const int hb = slot / half, d = slot - hb * half;
                    const int pos = p.ctx_len[b];
                    // linear outputs are bf16 activations in the reference; RoPE acts on those
                    const float x0 = bf2f(f2bf(acc[2 * s][b])), x1 = bf2f(f2bf(acc[2 * s + 1][b]));
                    if (hb < p.Hq + p.Hkv) {
                        const float cs = p.cos_t[(size_t)pos * half + d], sn = p.sin_t[(size_t)pos * half + d];
                        const bf16_t y0 = f2bf(x0 * cs - x1 * sn), y1 = f2bf(x1 * cs + x0 * sn);
                        if (hb < p.Hq) {
                            bf16_t* q = (bf16_t*)p.y + (size_t)b * p.ldy + hb * hd;
                            q[d] = y0;
                            q[d + half] = y1;
                        } else {
                            const int pg = p.page_table[(size_t)b * p.max_pages + pos / p.page];
                            bf16_t* kc = (bf16_t*)p.kcache + (((size_t)pg * p.Hkv + (hb - p.Hq)) * p.page + pos % p.page) * hd;
                            kc[d] = y0;
                            kc[d + half] = y1;
                        }
                    } else {
                        const int pg = p.page_table[(size_t)b * p.max_pages + pos / p.page];
                        bf16_t* vc = (bf16_t*)p.vcache + (((size_t)pg * p.Hkv + (hb - p.Hq - p.Hkv)) * p.page + pos % p.page) * hd;
                        vc[d] = f2bf(x0);
                        vc[d + half] = f2bf(x1);
                    }
                }
    } else if (MODE == MODE_LMHEAD) {
        // per-wave best over its rows, then block best; first index wins ties (torch.argmax semantics)
        __shared__ float bv[4][B];
        __shared__ int bi[4][B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float best = -INFINITY;
            int besti = 0x7fffffff;
#pragma unroll
            for (int s = 0; s < RPW; ++s)
                if (slot_ok[s]) {
                    const float v = acc[s][b];
                    if (v > best || (v == best && rows[s] < besti)) {
                        best = v;
                        besti = rows[s];
                    }
                    if (p.logits_out && lane == 0) p.logits_out[(size_t)b * p.n_slots + rows[s]] = v;
                }
            if (lane == 0) {
                bv[wave][b] = best;
                bi[wave][b] = besti;
            }
        }
        __syncthreads();
        if (tid < B) {
            float best = bv[0][tid];
            int besti = bi[0][tid];
#pragma unroll
            for (int w = 1; w < 4; ++w) {
                const float v = bv[w][tid];
                const int ii = bi[w][tid];
                if (v > best || (v == best && ii < besti)) {
                    best = v;
                    besti = ii;
                }
            }
            p.part_val[(size_t)blockIdx.x * B + tid] = best;
            p.part_idx[(size_t)blockIdx.x * B + tid] = besti;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// h[b] = E[cur_tok[b]]
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emmax_decode_embed_kernel(const int32_t* __restrict__ cur_tok, const bf16_t* __restrict__ E,
                                                                bf16_t* __restrict__ h, int hidden, int vocab) {
    const int b = blockIdx.x;
    int id = cur_tok[b];
    id = min(max(id, 0), vocab - 1);
    const u32x4_t* s = (const u32x4_t*)(E + (size_t)id * hidden);
    u32x4_t* o = (u32x4_t*)(h + (size_t)b * hidden);
    for (int c = threadIdx.x; c < hidden / 8; c += blockDim.x) o[c] = s[c];
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-KV decode attention over the paged cache.  grid (NSPLIT, Hkv, B), 256 threads.
// part[((b*Hq + h)*NSPLIT + s) * (HD+2)] = { o[0..HD) un-normalised, m, l }
// ---------------------------------------------------------------------------------------------------------------------
template <int HD, int G>
__global__ __launch_bounds__(256) void emmax_decode_attn_kernel(DecodeAttnParams p) {
    static_assert(HD == 128, "decode attention maps 16 lanes x 8 elements onto one 128-wide K/V row");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* sc = (float*)smem;                       // [G][kps] scores
    __shared__ float red_m[4][G];
    __shared__ float red_o[4][G][HD];
    __shared__ float red_l[4][G];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, ch = lane & 15;       // key group within the wave, 16-byte chunk within the row
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int nsplit = gridDim.x;
    const int L = p.ctx_len[b] + 1;                 // keys including the one appended by the qkv kernel of this step
    int kps = (L + nsplit - 1) / nsplit;
    kps = (kps + 15) & ~15;
    const int k0 = split * kps;
    const int k1 = min(L, k0 + kps);
    const int Hq = p.Hkv * G;
    float* part = p.part + ((size_t)(b * Hq + hk * G) * nsplit + split) * (HD + 2);

    if (k0 >= L) {   // empty split
        for (int i = tid; i < G * (HD + 2); i += 256) {
            const int gq = i / (HD + 2), j = i - gq * (HD + 2);
            part[(size_t)gq * nsplit * (HD + 2) + j] = (j == HD) ? -INFINITY : 0.f;
        }
        return;
    }

    const int32_t* pt = p.page_table + (size_t)b * p.max_pages;
    const bf16_t* kc = (const bf16_t*)p.kcache;
    const bf16_t* vc = (const bf16_t*)p.vcache;

    // q (already rotated, bf16) for the G heads of this kv head: lane holds elements ch*8 .. +8
    u32x4_t q[G];
#pragma unroll
    for (int gq = 0; gq < G; ++gq)
        q[gq] = *(const u32x4_t*)((const bf16_t*)p.q + (size_t)b * p.ldq + (hk * G + gq) * HD + ch * 8);

    // ---- phase A: scores ----
    float mloc[G];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) mloc[gq] = -INFINITY;
    for (int kb = k0; kb < k1; kb += 32) {
        u32x4_t kv[2];
        int key[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            key[u] = kb + u * 16 + wave * 4 + kg;
            if (key[u] < k1) {
                const int pg = pt[key[u] / p.page];
                kv[u] = *(const u32x4_t*)(kc + (((size_t)pg * p.Hkv + hk) * p.page + key[u] % p.page) * HD + ch * 8);
            } else {
                kv[u] = (u32x4_t){0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int gq = 0; gq < G; ++gq) {
                float s = 0.f;
                s = dot2_bf16(kv[u][0], q[gq][0], s);
                s = dot2_bf16(kv[u][1], q[gq][1], s);
                s = dot2_bf16(kv[u][2], q[gq][2], s);
                s = dot2_bf16(kv[u][3], q[gq][3], s);
                s += __shfl_xor(s, 1, 64);
                s += __shfl_xor(s, 2, 64);
                s += __shfl_xor(s, 4, 64);
                s += __shfl_xor(s, 8, 64);
                s *= p.scale;
                if (key[u] < k1) {
                    if (ch == 0) sc[gq * kps + (key[u] - k0)] = s;
                    mloc[gq] = fmaxf(mloc[gq], s);
                }
            }
    }
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        const float m = wave_max(mloc[gq]);
        if (lane == 0) red_m[wave][gq] = m;
    }
    __syncthreads();
    float m[G];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) m[gq] = fmaxf(fmaxf(red_m[0][gq], red_m[1][gq]), fmaxf(red_m[2][gq], red_m[3][gq]));

    // ---- phase B: o = sum_k exp(s_k - m) V_k ----
    float o[G][8], l[G];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        l[gq] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[gq][j] = 0.f;
    }
    for (int kb = k0; kb < k1; kb += 32) {
        u32x4_t vv[2];
        int key[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            key[u] = kb + u * 16 + wave * 4 + kg;
            if (key[u] < k1) {
                const int pg = pt[key[u] / p.page];
                vv[u] = *(const u32x4_t*)(vc + (((size_t)pg * p.Hkv + hk) * p.page + key[u] % p.page) * HD + ch * 8);
            } else {
                vv[u] = (u32x4_t){0u, 0u, 0u, 0u};
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (key[u] < k1) {
#pragma unroll
                for (int gq = 0; gq < G; ++gq) {
                    const float pw = __expf(sc[gq * kps + (key[u] - k0)] - m[gq]);
                    l[gq] += pw;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        o[gq][2 * j] += pw * bf_lo(vv[u][j]);
                        o[gq][2 * j + 1] += pw * bf_hi(vv[u][j]);
                    }
                }
            }
    }
    // reduce over the 4 key groups of the wave (lanes with equal ch), then over the 4 waves
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float v = o[gq][j];
            v += __shfl_xor(v, 16, 64);
            v += __shfl_xor(v, 32, 64);
            o[gq][j] = v;
        }
        float lv = l[gq];   // every lane of a key group added the same pw: count one lane per group
        lv += __shfl_xor(lv, 16, 64);
        lv += __shfl_xor(lv, 32, 64);
        if (kg == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) red_o[wave][gq][ch * 8 + j] = o[gq][j];
        }
        if (lane == 0) red_l[wave][gq] = lv;
    }
    __syncthreads();
    for (int i = tid; i < G * (HD + 2); i += 256) {
        const int gq = i / (HD + 2), j = i - gq * (HD + 2);
        float v;
        if (j < HD)
            v = red_o[0][gq][j] + red_o[1][gq][j] + red_o[2][gq][j] + red_o[3][gq][j];
        else if (j == HD)
            v = m[gq];
        else
            v = red_l[0][gq] + red_l[1][gq] + red_l[2][gq] + red_l[3][gq];
        part[(size_t)gq * nsplit * (HD + 2) + j] = v;
    }
}

// merge the split partials: out[b][h*HD + d] = sum_s o_s[d] e^{m_s - M} / sum_s l_s e^{m_s - M}
template <int HD>
__global__ __launch_bounds__(HD) void emmax_decode_attn_combine_kernel(const float* __restrict__ part, bf16_t* __restrict__ out,
                                                                      int ldo, int Hq, int nsplit) {
    const int h = blockIdx.x, b = blockIdx.y, d = threadIdx.x;
    const float* pp = part + (size_t)(b * Hq + h) * nsplit * (HD + 2);
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, pp[s * (HD + 2) + HD]);
    float num = 0.f, den = 0.f;
    for (int s = 0; s < nsplit; ++s) {
        const float ms = pp[s * (HD + 2) + HD];
        const float w = (ms == -INFINITY) ? 0.f : __expf(ms - M);
        num += pp[s * (HD + 2) + d] * w;
        den += pp[s * (HD + 2) + HD + 1] * w;
    }
    out[(size_t)b * ldo + h * HD + d] = f2bf(den > 0.f ? num / den : 0.f);
}

// ---------------------------------------------------------------------------------------------------------------------
// Finish a step: argmax over the lm-head partials, EOS / length bookkeeping, next current token.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emmax_decode_finish_kernel(FinishParams p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    __shared__ float sv[256];
    __shared__ int si[256];
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = tid; i < p.n_part; i += 256) {
        const float v = p.part_val[(size_t)i * p.B + b];
        const int ii = p.part_idx[(size_t)i * p.B + b];
        if (v > best || (v == best && ii < besti)) {
            best = v;
            besti = ii;
        }
    }
    sv[tid] = best;
    si[tid] = besti;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const float v = sv[tid + s];
            const int ii = si[tid + s];
            if (v > sv[tid] || (v == sv[tid] && ii < si[tid])) {
                sv[tid] = v;
                si[tid] = ii;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        int tok = si[0];
        if (tok == 0x7fffffff) tok = p.pad_id;
        int was_done = p.done[b];
        int n = p.n_out[b];
        if (!p.is_prefill && !was_done && n >= *p.max_new_p) {   // token budget already spent before this step
            was_done = 1;
            p.done[b] = 1;
        }
        if (!p.is_prefill && !was_done) p.ctx_len[b] += 1;   // the token consumed by this step now sits in the cache
        if (p.is_prefill)   // fresh sequence: clear the output row
            for (int i = 0; i < p.max_out; ++i) p.out_ids[(size_t)b * p.max_out + i] = p.pad_id;
        if (was_done) {
            tok = p.pad_id;
        } else {
            if (n < p.max_out) p.out_ids[(size_t)b * p.max_out + n] = tok;
            n += 1;
            p.n_out[b] = n;
            // stop on EOS, on the token budget, or when the next append would overflow the cache
            const bool budget = !p.is_prefill && n >= *p.max_new_p;
            if (tok == p.eos_id || budget || p.ctx_len[b] + 1 >= p.max_ctx) p.done[b] = 1;
        }
        p.cur_tok[b] = tok;
    }
}

__global__ void emmax_set_tokens_kernel(int32_t* cur_tok, const int32_t* toks, int B) {
    const int i = threadIdx.x;
    if (i < B) cur_tok[i] = toks[i];
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
template <int B, int RPW, int MODE, bool NORM>
static int launch_gemv_t(const GemvParams& p, hipStream_t stream) {
    const int slots_per_block = 4 * RPW;
    dim3 grid(cdiv(p.n_slots, slots_per_block)), block(256);
    const size_t smem = (size_t)B * p.kc * 2;
    auto kern = emmax_decode_gemv_kernel<B, RPW, MODE, NORM>;
    hipLaunchKernelGGL(kern, grid, block, smem, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <int MODE, bool NORM>
static int launch_gemv_mode(GemvParams p, int B, hipStream_t stream) {
    // K phase: keep B * kc * 2 bytes of activations under ~128 KiB of LDS
    const int cap = (128 * 1024 / 2 / B) & ~511;
    p.kc = p.K <= cap ? p.K : (cdiv(cdiv(p.K, cdiv(p.K, cap)), 512) * 512);
    if (NORM && p.kc != p.K) return -1;
    switch (B) {
        case 1: return launch_gemv_t<1, 2, MODE, NORM>(p, stream);
        case 2: return launch_gemv_t<2, 2, MODE, NORM>(p, stream);
        case 3: return launch_gemv_t<3, 2, MODE, NORM>(p, stream);
        case 4: return launch_gemv_t<4, 2, MODE, NORM>(p, stream);
        case 5: return launch_gemv_t<5, 2, MODE, NORM>(p, stream);
        case 6: return launch_gemv_t<6, 2, MODE, NORM>(p, stream);
        case 7: return launch_gemv_t<7, 2, MODE, NORM>(p, stream);
        case 8: return launch_gemv_t<8, 2, MODE, NORM>(p, stream);
        default: return -1;
    }
}

template <int MODE, bool NORM>
static int gemv_init_mode() {
    const int lim = 160 * 1024 - 4096;
    hipError_t e = hipSuccess;
#define SETB(BB) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_gemv_kernel<BB, 2, MODE, NORM>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    SETB(1); SETB(2); SETB(3); SETB(4); SETB(5); SETB(6); SETB(7); SETB(8);
#undef SETB
    return e == hipSuccess ? 0 : -4;
}
int decode_gemv_init() {
    static int done = -1;
    if (done == 0) return 0;
    int r = gemv_init_mode<MODE_QKV, true>();
    if (!r) r = gemv_init_mode<MODE_RESID, false>();
    if (!r) r = gemv_init_mode<MODE_GATEUP, true>();
    if (!r) r = gemv_init_mode<MODE_LMHEAD, true>();
    if (!r) r = gemv_init_mode<MODE_PLAIN, false>();
    done = r;
    return r;
}

int launch_decode_gemv(int mode, const GemvParams& p, int B, hipStream_t stream) {
    if (p.K % 8 || p.ldw % 8 || p.ldx % 8) return -1;
    switch (mode) {
        case MODE_QKV: return launch_gemv_mode<MODE_QKV, true>(p, B, stream);
        case MODE_RESID: return launch_gemv_mode<MODE_RESID, false>(p, B, stream);
        case MODE_GATEUP: return launch_gemv_mode<MODE_GATEUP, true>(p, B, stream);
        case MODE_LMHEAD: return launch_gemv_mode<MODE_LMHEAD, true>(p, B, stream);
        case MODE_PLAIN: return launch_gemv_mode<MODE_PLAIN, false>(p, B, stream);
        default: return -1;
    }
}

int launch_decode_embed(const int32_t* cur_tok, const void* E, void* h, int B, int hidden, int vocab, hipStream_t stream) {
    hipLaunchKernelGGL(emmax_decode_embed_kernel, dim3(B), dim3(256), 0, stream, cur_tok, (const bf16_t*)E, (bf16_t*)h, hidden, vocab);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int decode_attn_nsplit(int B, int Hkv) {
    int ns = 256 / (B * Hkv);
    if (ns < 1) ns = 1;
    if (ns > 16) ns = 16;
    return ns;
}

int launch_decode_attn(const DecodeAttnParams& p, int B, int Hq, int head_dim, int nsplit, int max_ctx, void* out, int ldo,
                       hipStream_t stream) {
    if (head_dim != 128) return -1;
    const int G = Hq / p.Hkv;
    int kps = cdiv(max_ctx + 1, nsplit);
    kps = (kps + 15) & ~15;
    const size_t smem = (size_t)G * kps * sizeof(float);
    dim3 grid(nsplit, p.Hkv, B), block(256);
    switch (G) {
        case 1: hipLaunchKernelGGL((emmax_decode_attn_kernel<128, 1>), grid, block, smem, stream, p); break;
        case 2: hipLaunchKernelGGL((emmax_decode_attn_kernel<128, 2>), grid, block, smem, stream, p); break;
        case 4: hipLaunchKernelGGL((emmax_decode_attn_kernel<128, 4>), grid, block, smem, stream, p); break;
        case 8: hipLaunchKernelGGL((emmax_decode_attn_kernel<128, 8>), grid, block, smem, stream, p); break;
        default: return -1;
    }
    if (hipGetLastError() != hipSuccess) return -4;
    hipLaunchKernelGGL(emmax_decode_attn_combine_kernel<128>, dim3(Hq, B), dim3(128), 0, stream, p.part, (bf16_t*)out, ldo, Hq, nsplit);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_decode_finish(const FinishParams& p, hipStream_t stream) {
    hipLaunchKernelGGL(emmax_decode_finish_kernel, dim3(p.B), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_set_tokens(int32_t* cur_tok, const int32_t* toks, int B, hipStream_t stream) {
    hipLaunchKernelGGL(emmax_set_tokens_kernel, dim3(1), dim3(64), 0, stream, cur_tok, toks, B);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
