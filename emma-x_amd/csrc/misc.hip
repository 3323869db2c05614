// misc.hip -- the HBM-bound glue kernels around the GEMMs:
//   patch gather (+ fused per-tower normalisation)   = PrismaticImageProcessor.apply_transform tail
//                                                      (processing_prismatic.py:136-143) + timm PatchEmbed im2col
//   token assembly (pos-embed add, cls/reg prefix)   = timm VisionTransformer._pos_embed (no_embed_class for DINOv2)
//   feature extraction / concat                       = modeling_prismatic.py:120-123 (drop prefix tokens, cat dim=2)
//   embedding gather + multimodal splice              = modeling_prismatic.py:380-385
//   RoPE + KV-cache append for the prefill            = HF apply_rotary_pos_emb + DynamicCache.update
#include "common.h"
#include "kernels.h"

namespace {

// One block per (image, patch). Output row = [3*P*P (+pad)] bf16 in conv-weight order k = c*P*P + dy*P + dx.
template <bool FROM_U8>
__global__ __launch_bounds__(256) void emmax_patch_gather_kernel(const void* __restrict__ src, bf16_t* __restrict__ out,
                                                                int img, int patch, int kpad, int chan0, float m0,
                                                                float m1, float m2, float s0, float s1, float s2) {
    const int gp = img / patch;
    const int b = blockIdx.x / (gp * gp), pidx = blockIdx.x % (gp * gp);
    const int py = pidx / gp, px = pidx % gp;
    const int pp = patch * patch;
    bf16_t* o = out + (size_t)blockIdx.x * kpad;
    for (int k = threadIdx.x; k < kpad; k += blockDim.x) {
        float v = 0.f;
        if (k < 3 * pp) {
            const int c = k / pp, rem = k - c * pp, dy = rem / patch, dx = rem - dy * patch;
            const int y = py * patch + dy, x = px * patch + dx;
            if (FROM_U8) {
                const uint8_t u = ((const uint8_t*)src)[(((size_t)b * img + y) * img + x) * 3 + c];
                const float mean = c == 0 ? m0 : (c == 1 ? m1 : m2), sd = c == 0 ? s0 : (c == 1 ? s1 : s2);
                v = ((float)u / 255.0f - mean) / sd;   // to_tensor then normalize, in fp32 as torchvision does
            } else {
                // pixel_values bf16 [B,6,img,img]: channels chan0..chan0+2 belong to this tower
                v = bf2f(((const bf16_t*)src)[(((size_t)b * 6 + chan0 + c) * img + y) * img + x]);
            }
        }
        o[k] = f2bf(v);
    }
}

// tokens[b, 0] = cls, tokens[b, 1..n_reg] = reg, tokens[b, n_prefix + p] = pe[b, p] + pos[p]   (pe / tokens row pitch = ld)
__global__ __launch_bounds__(256) void emmax_assemble_tokens_kernel(const bf16_t* __restrict__ pe, const bf16_t* __restrict__ pos,
                                                                   const bf16_t* __restrict__ cls, const bf16_t* __restrict__ reg,
                                                                   bf16_t* __restrict__ tokens, int n_patches, int n_prefix,
                                                                   int has_cls, int D, int ld) {
    const int N = n_prefix + n_patches;
    const int b = blockIdx.x / N, t = blockIdx.x % N;
    bf16_t* o = tokens + (size_t)blockIdx.x * ld;
    for (int d = threadIdx.x; d < D; d += blockDim.x) {
        bf16_t v;
        if (t < n_prefix) {
            v = (has_cls && t == 0) ? cls[d] : reg[(size_t)(t - has_cls) * D + d];
        } else {
            const int pi = t - n_prefix;
            v = f2bf(bf2f(pe[((size_t)b * n_patches + pi) * ld + d]) + bf2f(pos[(size_t)pi * D + d]));
        }
        o[d] = v;
    }
}

// out[(b*rows_out + r) * ld_out + col_off + d] = in[(b*rows_in + r_off + r) * ld_in + d]
__global__ __launch_bounds__(256) void emmax_copy_rows_kernel(const bf16_t* __restrict__ in, int ld_in, bf16_t* __restrict__ out,
                                                             int rows_in, int r_off, int rows_out, int D, int ld_out, int col_off) {
    const int b = blockIdx.x / rows_out, r = blockIdx.x % rows_out;
    const u32x4_t* s = (const u32x4_t*)(in + ((size_t)b * rows_in + r_off + r) * ld_in);
    u32x4_t* o = (u32x4_t*)(out + ((size_t)b * rows_out + r) * ld_out + col_off);
    for (int c = threadIdx.x; c < D / 8; c += blockDim.x) o[c] = s[c];
}

// h[cu[b] + s] = s==0 ? E[ids[b][0]] : (s <= n_patches ? patches[b][s-1] : E[ids[b][s - n_patches]])
__global__ __launch_bounds__(256) void emmax_embed_splice_kernel(const int32_t* __restrict__ ids, int P_max,
                                                                const int32_t* __restrict__ cu, const bf16_t* __restrict__ E,
                                                                const bf16_t* __restrict__ patches, bf16_t* __restrict__ h,
                                                                int n_patches, int hidden, int vocab, float* __restrict__ h32) {
    const int b = blockIdx.y, s = blockIdx.x;
    const int start = cu[b], len = cu[b + 1] - start;
    if (s >= len) return;
    const u32x4_t* src;
    if (s >= 1 && s <= n_patches) {
        src = (const u32x4_t*)(patches + ((size_t)b * n_patches + (s - 1)) * hidden);
    } else {
        int id = ids[(size_t)b * P_max + (s == 0 ? 0 : s - n_patches)];
        id = min(max(id, 0), vocab - 1);
        src = (const u32x4_t*)(E + (size_t)id * hidden);
    }
    u32x4_t* o = (u32x4_t*)(h + (size_t)(start + s) * hidden);
    for (int c = threadIdx.x; c < hidden / 8; c += blockDim.x) {
        const u32x4_t v = src[c];
        o[c] = v;
        if (h32) {   // the fp32 residual stream of the prefill starts from the same values (exact)
            const f32x8_t f = bf16x8_to_f32(v);
            float* hp = h32 + (size_t)(start + s) * hidden + (size_t)c * 8;
            *(f32x4_t*)hp = f.lo;
            *(f32x4_t*)(hp + 4) = f.hi;
        }
    }
}

// Prefill RoPE (rotate-half convention) on q,k in place + K/V append into the paged cache.
// One block per packed token row; threads over (head, pair index).
__global__ __launch_bounds__(256) void emmax_rope_kv_write_kernel(bf16_t* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                                 const int32_t* __restrict__ cu, int B,
                                                                 const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                 bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                                                                 const int32_t* __restrict__ page_table, int max_pages,
                                                                 int Hq, int Hkv, int hd, int page) {
    const int row = blockIdx.x;
    // locate the sequence of this packed row (B is small)
    int b = 0;
    while (b + 1 < B && row >= cu[b + 1]) ++b;
    const int pos = row - cu[b];
    const int half = hd >> 1;
    bf16_t* r = qkv + (size_t)row * ld;
    const float* cs = cos_t + (size_t)pos * half;
    const float* sn = sin_t + (size_t)pos * half;
    const int pg = page_table[(size_t)b * max_pages + pos / page], slot = pos % page;
    for (int i = threadIdx.x; i < (Hq + Hkv) * half; i += blockDim.x) {
        const int hh = i / half, d = i - hh * half;
        bf16_t* x = (hh < Hq) ? (r + q_off + hh * hd) : (r + k_off + (hh - Hq) * hd);
        const float x0 = bf2f(x[d]), x1 = bf2f(x[d + half]);
        const float c = cs[d], s = sn[d];
        // (the fused form hipcc chose for `x0 * c - x1 * s`, `x1 * c + x0 * s` here, written out so that the 16-byte kernel below rounds alike)
        const bf16_t y0 = f2bf(fmaf(x0, c, -(x1 * s))), y1 = f2bf(fmaf(x0, s, x1 * c));
        x[d] = y0;
        x[d + half] = y1;
        if (hh >= Hq && kcache) {   // (null: fp8 KV cache -- the quantising pass below appends the rows)
            bf16_t* kc = kcache + (((size_t)pg * Hkv + (hh - Hq)) * page + slot) * hd;
            kc[d] = y0;
            kc[d + half] = y1;
        }
    }
    for (int i = threadIdx.x; vcache && i < Hkv * hd / 8; i += blockDim.x) {
        const int hk = i / (hd / 8), ch = i - hk * (hd / 8);
        const u32x4_t v = *(const u32x4_t*)(r + v_off + hk * hd + ch * 8);
        *(u32x4_t*)(vcache + (((size_t)pg * Hkv + hk) * page + slot) * hd + ch * 8) = v;
    }
}

// The same pass with 16-byte accesses (head_dim % 16 == 0, 16-byte aligned rows): a thread owns eight consecutive pair indices
// d .. d + 7 of one head -- one 16-byte load from each half of the head, two from each table, the same fused multiply-adds per element
// (bit-identical to the element-wise kernel above, which took 13 us for the 768 rows of a one-frame prefill and 76 us for 6144).
__global__ __launch_bounds__(256) void emmax_rope_kv_write_vec_kernel(bf16_t* __restrict__ qkv, int ld, int q_off, int k_off, int v_off,
                                                                     const int32_t* __restrict__ cu, int B,
                                                                     const float* __restrict__ cos_t, const float* __restrict__ sin_t,
                                                                     bf16_t* __restrict__ kcache, bf16_t* __restrict__ vcache,
                                                                     const int32_t* __restrict__ page_table, int max_pages,
                                                                     int Hq, int Hkv, int hd, int page) {
    const int row = blockIdx.x;
    int b = 0;
    while (b + 1 < B && row >= cu[b + 1]) ++b;
    const int pos = row - cu[b];
    const int half = hd >> 1, hc = half >> 3;    // 8-pair chunks per head
    bf16_t* r = qkv + (size_t)row * ld;
    const float* cs = cos_t + (size_t)pos * half;
    const float* sn = sin_t + (size_t)pos * half;
    const int pg = page_table[(size_t)b * max_pages + pos / page], slot = pos % page;
    for (int i = threadIdx.x; i < (Hq + Hkv) * hc; i += blockDim.x) {
        const int hh = i / hc, d = (i - hh * hc) * 8;
        bf16_t* x = (hh < Hq) ? (r + q_off + hh * hd) : (r + k_off + (hh - Hq) * hd);
        const u32x4_t lo = *(const u32x4_t*)(x + d), hi = *(const u32x4_t*)(x + d + half);
        const f32x4_t c0 = *(const f32x4_t*)(cs + d), c1 = *(const f32x4_t*)(cs + d + 4);
        const f32x4_t s0 = *(const f32x4_t*)(sn + d), s1 = *(const f32x4_t*)(sn + d + 4);
        u32x4_t ylo, yhi;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float ca = j < 2 ? c0[2 * j] : c1[2 * j - 4], cb = j < 2 ? c0[2 * j + 1] : c1[2 * j - 3];
            const float sa = j < 2 ? s0[2 * j] : s1[2 * j - 4], sb = j < 2 ? s0[2 * j + 1] : s1[2 * j - 3];
            const float x0a = bf_lo(lo[j]), x0b = bf_hi(lo[j]), x1a = bf_lo(hi[j]), x1b = bf_hi(hi[j]);
            ylo[j] = pack_bf16x2(fmaf(x0a, ca, -(x1a * sa)), fmaf(x0b, cb, -(x1b * sb)));
            yhi[j] = pack_bf16x2(fmaf(x0a, sa, x1a * ca), fmaf(x0b, sb, x1b * cb));
        }
        *(u32x4_t*)(x + d) = ylo;
        *(u32x4_t*)(x + d + half) = yhi;
        if (hh >= Hq && kcache) {   // (null: fp8 KV cache -- the quantising pass below appends the rows)
            bf16_t* kc = kcache + (((size_t)pg * Hkv + (hh - Hq)) * page + slot) * hd;
            // (non-temporal: the cache rows are next read by the decode steps, a whole prefill later -- they need not displace the qkv rows the
            // attention launch is about to read)
            __builtin_nontemporal_store(ylo, (u32x4_t*)(kc + d));
            __builtin_nontemporal_store(yhi, (u32x4_t*)(kc + d + half));
        }
    }
    for (int i = threadIdx.x; vcache && i < Hkv * hd / 8; i += blockDim.x) {
        const int hk = i / (hd / 8), ch = i - hk * (hd / 8);
        const u32x4_t v = *(const u32x4_t*)(r + v_off + hk * hd + ch * 8);
        __builtin_nontemporal_store(v, (u32x4_t*)(vcache + (((size_t)pg * Hkv + hk) * page + slot) * hd + ch * 8));
    }
}

// fp8 KV cache (round 5, opt-in): the prefill's K (already rotated in place by the pass above) and V rows of every packed token as e4m3 bytes
// with one power-of-two fp32 scale per (token, kv head) row (common.h: e4m3_row_scale).  One block per token, a 16-lane group per 128-element row (head_dim 128).
// The rows are written BACK into the qkv buffer as bf16(e4m3 x scale) (round 6): the prefill attention that follows then attends
// over exactly the values every later decode step reads from the cache, as the decode step's own new key already does.
__global__ __launch_bounds__(256) void emmax_kv_quant_rows_kernel(bf16_t* __restrict__ qkv, int ld, int k_off, int v_off,
                                                                 const int32_t* __restrict__ cu, int B, uint8_t* __restrict__ k8,
                                                                 uint8_t* __restrict__ v8, float* __restrict__ kscale, float* __restrict__ vscale,
                                                                 const int32_t* __restrict__ page_table, int max_pages, int Hkv, int page) {
    constexpr int HD = 128;
    const int row = blockIdx.x;
    int b = 0;
    while (b + 1 < B && row >= cu[b + 1]) ++b;
    const int pos = row - cu[b];
    const int pg = page_table[(size_t)b * max_pages + pos / page], slot = pos % page;
    bf16_t* r = qkv + (size_t)row * ld;
    const int grp = threadIdx.x >> 4, ch = threadIdx.x & 15;
    for (int i = grp; i < 2 * Hkv; i += 16) {   // (a 16-lane group = one DPP row: the reduction never mixes an idle group in)
        const bool is_v = i >= Hkv;
        const int hk = is_v ? i - Hkv : i;
        bf16_t* src = r + (is_v ? v_off : k_off) + hk * HD + ch * 8;
        const u32x4_t v = *(const u32x4_t*)src;
        float am = 0.f;
#pragma unroll
        for (int j = 0; j < 4; ++j) am = fmaxf(am, fmaxf(fabsf(bf_lo(v[j])), fabsf(bf_hi(v[j]))));
        am = row16_max(am);
        const float sc = e4m3_row_scale(am);
        const size_t rowi = ((size_t)pg * Hkv + hk) * page + slot;
        const u32x2_t q8 = quant8_e4m3(v, 1.0f / sc);
        *(u32x2_t*)((is_v ? v8 : k8) + rowi * HD + ch * 8) = q8;
        *(u32x4_t*)src = dequant8_e4m3(q8, sc);
        if (ch == 0) (is_v ? vscale : kscale)[rowi] = sc;
    }
}

// gather the last row of every packed sequence: out[b] = in[cu[b+1]-1]
// out32 != null: the rows also widened to fp32 (the decode step's fp32 residual stream); in32 != null: the source rows are fp32 (the
// prefill's fp32 residual stream): out32 = the row, out = its bf16 rounding
__global__ __launch_bounds__(256) void emmax_gather_last_rows_kernel(const bf16_t* __restrict__ in, bf16_t* __restrict__ out,
                                                                    const int32_t* __restrict__ cu, int D, float* __restrict__ out32,
                                                                    const float* __restrict__ in32) {
    const int b = blockIdx.x;
    if (in32) {
        const float* sp = in32 + (size_t)(cu[b + 1] - 1) * D;
        for (int c = threadIdx.x; c < D / 8; c += blockDim.x) {
            const f32x8_t f = ld_f32x8(sp + (size_t)c * 8);
            *((u32x4_t*)(out + (size_t)b * D) + c) = f32x8_to_bf16(f);
            if (out32) {
                float* hp = out32 + (size_t)b * D + (size_t)c * 8;
                *(f32x4_t*)hp = f.lo;
                *(f32x4_t*)(hp + 4) = f.hi;
            }
        }
        return;
    }
    const u32x4_t* s = (const u32x4_t*)(in + (size_t)(cu[b + 1] - 1) * D);
    if (out32) {
        for (int c = threadIdx.x; c < D / 8; c += blockDim.x) {
            const f32x8_t f = bf16x8_to_f32(s[c]);
            float* hp = out32 + (size_t)b * D + (size_t)c * 8;
            *(f32x4_t*)hp = f.lo;
            *(f32x4_t*)(hp + 4) = f.hi;
        }
    }
    u32x4_t* o = (u32x4_t*)(out + (size_t)b * D);
    for (int c = threadIdx.x; c < D / 8; c += blockDim.x) o[c] = s[c];
}

// One separable pass of Pillow's antialiased bicubic resize on uint8 RGB (22-bit fixed-point taps, round-half-up, clip):
// AXIS 1: [B,H,W,3] -> [B,H,out_n,3] (horizontal);  AXIS 0: [B,H,W,3] -> [B,out_n,W,3] (vertical).
// bounds[o] = (first input index, taps), kk[o*ksize + t] = tap t.  Bit-exact with PIL.Image.resize(BICUBIC), which is what
// torchvision's TVF.resize does for the PIL images of processing_prismatic.py:136.
template <int AXIS>
__global__ __launch_bounds__(256) void emmax_resize_pass_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int B, int H,
                                                               int W, int out_n, const int32_t* __restrict__ bounds,
                                                               const int32_t* __restrict__ kk, int ksize) {
    const int OH = AXIS == 0 ? out_n : H, OW = AXIS == 1 ? out_n : W;
    const size_t total = (size_t)B * OH * OW;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int ox = (int)(i % OW), oy = (int)((i / OW) % OH), b = (int)(i / ((size_t)OW * OH));
        const int o = AXIS == 0 ? oy : ox;
        const int lo = bounds[2 * o], n = bounds[2 * o + 1];
        const int32_t* k = kk + (size_t)o * ksize;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        for (int t = 0; t < n; ++t) {
            const int sy = AXIS == 0 ? lo + t : oy, sx = AXIS == 1 ? lo + t : ox;
            const uint8_t* px = src + (((size_t)b * H + sy) * W + sx) * 3;
            const int w = k[t];
            a0 += (int)px[0] * w;
            a1 += (int)px[1] * w;
            a2 += (int)px[2] * w;
        }
        uint8_t* q = dst + (((size_t)b * OH + oy) * OW + ox) * 3;
        q[0] = (uint8_t)min(max(a0 >> 22, 0), 255);
        q[1] = (uint8_t)min(max(a1 >> 22, 0), 255);
        q[2] = (uint8_t)min(max(a2 >> 22, 0), 255);
    }
}

// device-side (re)initialisation of the per-sequence state at prefill: cu_seqlens, ctx_len, done, n_out.
// Values travel as kernel arguments, so there is no host staging buffer to race with.
__global__ void emmax_prefill_state_kernel(PrefillState st, int32_t* cu, int32_t* ctx_len, int32_t* done, int32_t* n_out, int32_t* max_new,
                                           int32_t* stop_m, int32_t* stop_after) {
    if (threadIdx.x == 0) {
        int acc = 0;
        cu[0] = 0;
        for (int b = 0; b < st.B; ++b) {
            acc += st.S[b];
            cu[b + 1] = acc;
            ctx_len[b] = st.S[b];
            done[b] = 0;
            n_out[b] = 0;
            max_new[b] = 0x7fffffff;   // no token budget until a generate / slot call sets one
            stop_m[b] = 0;
            stop_after[b] = -1;
        }
    }
}
__global__ void emmax_set_int_kernel(int32_t* p, int32_t v) { *p = v; }
__global__ void emmax_set_ints_kernel(int32_t* p, int n, int32_t v) {
    if ((int)threadIdx.x < n) p[threadIdx.x] = v;
}
// idle request slots: nothing to decode, empty context (their share of a batched step costs no K/V traffic)
__global__ void emmax_slots_idle_kernel(int n, int32_t* cur_tok, int32_t* ctx_len, int32_t* done, int32_t* n_out, int32_t pad_id) {
    const int b = threadIdx.x;
    if (b < n) {
        cur_tok[b] = pad_id;
        ctx_len[b] = 0;
        done[b] = 1;
        n_out[b] = 0;
    }
}

// one block per committed request
__global__ __launch_bounds__(256) void emmax_slots_commit_kernel(CommitParams c) {
    const int i = blockIdx.x, src = c.src[i], dst = c.slot[i], tid = threadIdx.x;
    for (int k = tid; k < c.max_out; k += 256) c.out_ids[(size_t)dst * c.max_out + k] = c.out_ids[(size_t)src * c.max_out + k];
    for (int k = tid; k < c.max_pages; k += 256) {
        const int32_t a = c.page_table[(size_t)src * c.max_pages + k], b = c.page_table[(size_t)dst * c.max_pages + k];
        c.page_table[(size_t)dst * c.max_pages + k] = a;
        c.page_table[(size_t)src * c.max_pages + k] = b;
    }
    if (tid == 0) {
        c.cur_tok[dst] = c.cur_tok[src]; c.ctx_len[dst] = c.ctx_len[src]; c.n_out[dst] = c.n_out[src]; c.max_new[dst] = c.max_new[src];
        c.stop_m[dst] = c.stop_m[src]; c.stop_after[dst] = c.stop_after[src];
        c.done[dst] = c.done[src];
        c.done[src] = 1; c.ctx_len[src] = 0;
    }
}

}  // namespace

int launch_slots_commit(const CommitParams& c, hipStream_t stream) {
    if (c.n < 1 || c.n > EMMAX_MAX_DECODE_BATCH) return -1;
    hipLaunchKernelGGL(emmax_slots_commit_kernel, dim3(c.n), dim3(256), 0, stream, c);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_prefill_state(const PrefillState& st, int32_t* cu, int32_t* ctx_len, int32_t* done, int32_t* n_out, int32_t* max_new,
                         int32_t* stop_m, int32_t* stop_after, hipStream_t stream) {
    hipLaunchKernelGGL(emmax_prefill_state_kernel, dim3(1), dim3(64), 0, stream, st, cu, ctx_len, done, n_out, max_new, stop_m, stop_after);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_set_ints(int32_t* p, int n, int32_t v, hipStream_t stream) {
    if (n < 1 || n > 64) return -1;
    hipLaunchKernelGGL(emmax_set_ints_kernel, dim3(1), dim3(64), 0, stream, p, n, v);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_slots_idle(int n, int32_t* cur_tok, int32_t* ctx_len, int32_t* done, int32_t* n_out, int32_t pad_id, hipStream_t stream) {
    if (n < 1 || n > 64) return -1;
    hipLaunchKernelGGL(emmax_slots_idle_kernel, dim3(1), dim3(64), 0, stream, n, cur_tok, ctx_len, done, n_out, pad_id);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_set_int(int32_t* p, int32_t v, hipStream_t stream) {
    hipLaunchKernelGGL(emmax_set_int_kernel, dim3(1), dim3(1), 0, stream, p, v);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_resize_bicubic_u8(const uint8_t* src, int B, int H, int W, uint8_t* dst, int OH, int OW, uint8_t* tmp, const int32_t* bounds_h,
                             const int32_t* kk_h, int ksize_h, const int32_t* bounds_v, const int32_t* kk_v, int ksize_v, hipStream_t stream) {
    // Pillow's order: horizontal first (into tmp [B,H,OW,3]), then vertical
    const uint8_t* cur = src;
    int curW = W;
    if (W != OW) {
        uint8_t* out = (H != OH) ? tmp : dst;
        const size_t total = (size_t)B * H * OW;
        hipLaunchKernelGGL(emmax_resize_pass_kernel<1>, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0, stream, cur, out, B, H, W,
                           OW, bounds_h, kk_h, ksize_h);
        cur = out;
        curW = OW;
    }
    if (H != OH) {
        const size_t total = (size_t)B * OH * curW;
        hipLaunchKernelGGL(emmax_resize_pass_kernel<0>, dim3((unsigned)min((size_t)4096, (total + 255) / 256)), dim3(256), 0, stream, cur, dst, B, H, curW,
                           OH, bounds_v, kk_v, ksize_v);
    } else if (W == OW) {
        if (hipMemcpyAsync(dst, src, (size_t)B * H * W * 3, hipMemcpyDeviceToDevice, stream) != hipSuccess) return -4;
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_patch_gather(bool from_u8, const void* src, void* out, int B, int img, int patch, int kpad, int chan0,
                        const float* mean, const float* std, hipStream_t stream) {
    const int gp = img / patch;
    dim3 grid(B * gp * gp), block(256);
    if (from_u8)
        hipLaunchKernelGGL(emmax_patch_gather_kernel<true>, grid, block, 0, stream, src, (bf16_t*)out, img, patch, kpad, chan0,
                           mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    else
        hipLaunchKernelGGL(emmax_patch_gather_kernel<false>, grid, block, 0, stream, src, (bf16_t*)out, img, patch, kpad, chan0,
                           mean[0], mean[1], mean[2], std[0], std[1], std[2]);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_assemble_tokens(const void* pe, const void* pos, const void* cls, const void* reg, void* tokens, int B,
                           int n_patches, int n_prefix, int has_cls, int D, int ld, hipStream_t stream) {
    dim3 grid(B * (n_prefix + n_patches)), block(256);
    hipLaunchKernelGGL(emmax_assemble_tokens_kernel, grid, block, 0, stream, (const bf16_t*)pe, (const bf16_t*)pos,
                       (const bf16_t*)cls, (const bf16_t*)reg, (bf16_t*)tokens, n_patches, n_prefix, has_cls, D, ld);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_copy_rows(const void* in, int ld_in, void* out, int B, int rows_in, int r_off, int rows_out, int D, int ld_out,
                     int col_off, hipStream_t stream) {
    if (D % 8 || ld_out % 8 || col_off % 8 || ld_in % 8) return -1;
    dim3 grid(B * rows_out), block(128);
    hipLaunchKernelGGL(emmax_copy_rows_kernel, grid, block, 0, stream, (const bf16_t*)in, ld_in, (bf16_t*)out, rows_in, r_off, rows_out,
                       D, ld_out, col_off);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_embed_splice(const int32_t* ids, int P_max, const int32_t* cu, const void* E, const void* patches, void* h, int B,
                        int max_seqlen, int n_patches, int hidden, int vocab, hipStream_t stream, float* h32) {
    if (hidden % 8) return -1;
    dim3 grid(max_seqlen, B), block(256);
    hipLaunchKernelGGL(emmax_embed_splice_kernel, grid, block, 0, stream, ids, P_max, cu, (const bf16_t*)E,
                       (const bf16_t*)patches, (bf16_t*)h, n_patches, hidden, vocab, h32);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_rope_kv_write(void* qkv, int ld, int q_off, int k_off, int v_off, const int32_t* cu, int B, int total_rows,
                         const float* cos_t, const float* sin_t, void* kcache, void* vcache, const int32_t* page_table,
                         int max_pages, int Hq, int Hkv, int hd, int page, hipStream_t stream) {
    dim3 grid(total_rows), block(256);
    const bool vec = (hd & 15) == 0 && ((ld | q_off | k_off | v_off) & 7) == 0 && (((size_t)qkv | (size_t)kcache | (size_t)vcache) & 15) == 0 &&
                     (((size_t)cos_t | (size_t)sin_t) & 15) == 0;
    if (vec) {
        hipLaunchKernelGGL(emmax_rope_kv_write_vec_kernel, grid, block, 0, stream, (bf16_t*)qkv, ld, q_off, k_off, v_off, cu, B, cos_t,
                           sin_t, (bf16_t*)kcache, (bf16_t*)vcache, page_table, max_pages, Hq, Hkv, hd, page);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    hipLaunchKernelGGL(emmax_rope_kv_write_kernel, grid, block, 0, stream, (bf16_t*)qkv, ld, q_off, k_off, v_off, cu, B, cos_t,
                       sin_t, (bf16_t*)kcache, (bf16_t*)vcache, page_table, max_pages, Hq, Hkv, hd, page);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_kv_quant_rows(void* qkv, int ld, int k_off, int v_off, const int32_t* cu, int B, int total_rows, void* k8, void* v8, float* kscale,
                         float* vscale, const int32_t* page_table, int max_pages, int Hkv, int hd, int page, hipStream_t stream) {
    if (hd != 128 || (ld | k_off | v_off) % 8) return -1;
    hipLaunchKernelGGL(emmax_kv_quant_rows_kernel, dim3(total_rows), dim3(256), 0, stream, (bf16_t*)qkv, ld, k_off, v_off, cu, B, (uint8_t*)k8,
                       (uint8_t*)v8, kscale, vscale, page_table, max_pages, Hkv, page);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_gather_last_rows(const void* in, void* out, const int32_t* cu, int B, int D, hipStream_t stream, float* out32, const float* in32) {
    dim3 grid(B), block(256);
    hipLaunchKernelGGL(emmax_gather_last_rows_kernel, grid, block, 0, stream, (const bf16_t*)in, (bf16_t*)out, cu, D, out32, in32);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
