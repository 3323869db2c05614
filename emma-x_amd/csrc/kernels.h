// kernels.h -- parameter blocks and host launchers of the gfx950 kernels (internal to libemmax_hip.so).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "common.h"

// ---- gemm.hip ----
struct GemmParams {
    const void* A;         // bf16 [M, lda]
    const void* W;         // bf16 [N, ldw]  (K contiguous)
    void* C;               // bf16 / f32 [M, ldc]
    const void* bias;      // bf16 [N] or null
    const void* scale;     // bf16 [N] or null (LayerScale)
    const void* residual;  // bf16 [M, ldr] or null
    int M, N, K;
    int lda, ldw, ldc, ldr;
    int res_f32;           // 1: `residual` is f32 [M, ldr] -- the fp32 residual stream of the prefill (round 5); needs out_f32 (C may alias it: every
                           //    element is read and written by the same thread).  Honoured by the direct epilogue and the split-K reduce passes
    int N_store;           // columns >= N_store are not written
    int act;               // 0 none, 1 gelu(erf), 2 swiglu over 16-col interleaved (gate,up)
    int out_f32;
    // split-K (under-filled problems with a long K, e.g. M = 768 prefill o / down, batch-1 ViT fc2): `ksplit` K slices per
    // tile write fp32 partial tiles to `ws` ([ksplit][M][N] floats), a second kernel sums them and applies the epilogue.
    // ws == null: never split.  ksplit is set by launch_gemm.
    float* ws;
    long long ws_bytes;
    int ksplit;
    // LayerNorm folded into this GEMM (the ViT qkv / fc1 projections): A holds the RAW rows x, W holds
    // W' = bf16(W .* gamma) (launch_ln_fold at finalize), and the epilogue finishes the algebra per output element
    //     y[m, n] = rstd[m] * (acc[m, n] - mean[m] * ln_s[n]) + ln_c[n],   ln_s[n] = sum_k W'[n, k],  ln_c[n] = sum_k W[n, k] beta[k] + bias[n]
    // = LN(x) W^T + bias without ever writing LN(x).  ln_stats: f32 [M][2] = (mean, rstd) per row (launch_row_stats);
    // ln_s / ln_c: f32 [N].  All three null: plain GEMM.  bias must be null when they are set (it lives in ln_c).
    const float* ln_stats;
    const float* ln_s;
    const float* ln_c;
    // RMSNorm of the RESULT rows fused into the split-K reduce pass (one-frame prefill: o-proj -> post-attention norm, down -> the next
    // layer's input norm): norm_out bf16 [M, ld_norm] = RMSNorm(C; norm_w, norm_eps), written next to C.  Only honoured on the whole-problem
    // split-K path -- ask gemm_fuses_norm(p) first, and run launch_rmsnorm yourself when it says no.
    const void* norm_w;
    void* norm_out;
    int ld_norm;
    float norm_eps;
    // exact numerics (round 6, tuning switch `exact`): A holds every activation as TWO bf16 terms x = hi + lo (16 mantissa bits), laid out
    // per 64 elements as [64 x hi][64 x lo] ("HL rows": K steps 2 j / 2 j + 1 of A are the two terms of W's K step j).  The caller passes
    // K = 2 x the real K and lda = the HL row pitch; W keeps its real K columns.  Both terms go through the SAME bf16 MFMAs into the same
    // fp32 accumulators: weights are exact bf16, products are exact in fp32, so the result carries the fp32 reference's arithmetic to
    // ~2^-17 relative per operand instead of the 2^-9 of one bf16 term.
    int a_hl;
    int dbg;               // tools only: 1 = skip the operand DMA after K step 1 (LDS + MFMA time alone), 2 = skip the stores
    long long* trace;      // tools only (tools/gemm_trace.hip): block 0 writes wall_clock64() stamps per tile phase; null in the product
};
int launch_gemm(const GemmParams& p, hipStream_t stream);                     // picks the 256x256 or the 128x128 tile geometry
int launch_gemm_geom(const GemmParams& p, int big, hipStream_t stream);       // explicit geometry (tools, tests)
int gemm_big_tiles(const GemmParams& p);
int gemm_plan_describe(const GemmParams& p, char* buf, int len);              // the plan launch_gemm would run, as text (host only); returns its kind or -1
bool gemm_fuses_norm(const GemmParams& p);                                    // launch_gemm(p) will apply p.norm_* (see GemmParams)
int launch_gemm_splitk(const GemmParams& p, int ksplit, hipStream_t stream, int big = 0);  // ksplit K slices per tile (128x128; big: 256x256) + reduce / epilogue pass (needs p.ws)

// ---- norm.hip ----
int launch_layernorm(const void* x, void* y, const void* w, const void* b, int rows, int D, int ldx, int ldy, float eps,
                     hipStream_t stream);
int launch_rmsnorm(const void* x, void* y, const void* w, int rows, int D, int ldx, int ldy, float eps, hipStream_t stream);
int launch_rmsnorm_f32(const float* x, void* y, const void* w, int rows, int D, int ldx, int ldy, float eps, hipStream_t stream);   // fp32 rows in, bf16 out
// LayerNorm statistics alone: stats f32 [rows][2] = (mean, rstd) of every bf16 row (two-pass variance from registers, as launch_layernorm)
int launch_row_stats(const void* x, float* stats, int rows, int D, int ldx, float eps, hipStream_t stream);
// fold a LayerNorm into the [N, ldw] bf16 weight of the projection that consumes it (in place): W[n, k] <- bf16(W[n, k] * gamma[k]),
// ln_s[n] = sum_k W'[n, k] (of the rounded values), ln_c[n] = sum_k W[n, k] * beta[k] + bias[n] (fp32; bias may be null)
int launch_ln_fold(void* W, int ldw, int N, int K, const void* gamma, const void* beta, const void* bias, float* ln_s, float* ln_c,
                   hipStream_t stream);

// ---- attention.hip ----
struct AttnParams {
    const void* qkv;
    void* out;
    const int32_t* cu_seqlens;
    int ld_qkv, q_off, k_off, v_off, ld_out;
    int B, max_seqlen, Hq, Hkv;
    float scale;
    int causal;
};
int launch_attention(const AttnParams& p, int head_dim, hipStream_t stream);
int launch_x_attention(const AttnParams& p, int head_dim, hipStream_t stream);   // exact.hip: fp32 q / k / v rows in, HL rows out, fp32 MFMA

// ---- exact.hip: the exact-numerics mode (tuning switch `exact`; GemmParams::a_hl) ----
// "HL rows": bf16 rows holding every element as two terms, per 64 elements [64 x hi][64 x lo]; pitch >= 2 x the padded width
int launch_x_split_rows(const float* x, void* y_hl, int rows, int D, int Dp, int ldx, int ldy, hipStream_t stream);
int launch_x_rmsnorm(const float* x, void* y_hl, const void* w, int rows, int D, int ldx, int ldy, float eps, hipStream_t stream);
int launch_x_layernorm(const float* x, void* y_hl, const void* w, const void* b, int rows, int D, int Dp, int ldx, int ldy, float eps, hipStream_t stream);
int launch_x_join_rows(const void* x_hl, float* y, int rows, int D, int ldx, int ldy, hipStream_t stream);   // y = hi + lo (tests)
int launch_x_to_bf16(const float* x, int ldx, void* y, int ldy, int rows, int D, hipStream_t stream);
int launch_x_patch_gather(bool from_u8, const void* src, void* out_hl, int B, int img, int patch, int kpad, int chan0, const float* mean, const float* std,
                          hipStream_t stream);
int launch_x_assemble_tokens(const float* pe, const void* pos, const void* cls, const void* reg, float* tokens, int B, int n_patches, int n_prefix, int has_cls,
                             int D, int ld, hipStream_t stream);
int launch_x_embed_splice(const int32_t* ids, int P_max, const int32_t* cu, const void* E, const float* patches32, const void* patches_bf, float* h32, int B,
                          int max_seqlen, int n_patches, int hidden, int vocab, hipStream_t stream);
int launch_x_rope_kv_write(float* qkv, int ld, int q_off, int k_off, int v_off, const int32_t* cu, int B, int total_rows, const float* cos_t, const float* sin_t,
                           void* kcache, void* vcache, long long kv24, const int32_t* page_table, int max_pages, int Hq, int Hkv, int hd, int page, hipStream_t stream);

// ---- misc.hip ----
int launch_patch_gather(bool from_u8, const void* src, void* out, int B, int img, int patch, int kpad, int chan0,
                        const float* mean, const float* std, hipStream_t stream);
int launch_assemble_tokens(const void* pe, const void* pos, const void* cls, const void* reg, void* tokens, int B,
                           int n_patches, int n_prefix, int has_cls, int D, int ld, hipStream_t stream);
int launch_copy_rows(const void* in, int ld_in, void* out, int B, int rows_in, int r_off, int rows_out, int D, int ld_out,
                     int col_off, hipStream_t stream);
#define EMMAX_MAX_DECODE_BATCH 64   // rows of a decode step: 17-32 = two 16-wide MFMA batch tiles, 33-64 = two halves of two (decode_kmp.hip; decode_km.hip: 16 rows); rounds 1-4: 8, round 5: 32
#define EMMAX_KMP_ROWS 32           // rows one decode_kmp.hip half stages: the down projection and the lm-head of 33-64 rows run as two launches of <= 32
#define EMMAX_MAX_STOP_IDS 16
struct PrefillState {
    int B;
    int S[EMMAX_MAX_DECODE_BATCH];
};
// per-row decode state of B fresh sequences (pointers already offset to the first row / slot)
int launch_prefill_state(const PrefillState& st, int32_t* cu, int32_t* ctx_len, int32_t* done, int32_t* n_out, int32_t* max_new,
                         int32_t* stop_m, int32_t* stop_after, hipStream_t stream);
// emmax_slots_commit: staging rows src[i] become slots slot[i]: per-row state and the output row copied, page-table rows swapped (the
// slot's old pages become the staging row's)
struct CommitParams {
    int n, max_pages, max_out;
    int slot[EMMAX_MAX_DECODE_BATCH], src[EMMAX_MAX_DECODE_BATCH];
    int32_t *cur_tok, *ctx_len, *done, *n_out, *max_new, *stop_m, *stop_after, *out_ids, *page_table;
};
int launch_slots_commit(const CommitParams& c, hipStream_t stream);
int launch_set_int(int32_t* p, int32_t v, hipStream_t stream);
int launch_set_ints(int32_t* p, int n, int32_t v, hipStream_t stream);
int launch_slots_idle(int n, int32_t* cur_tok, int32_t* ctx_len, int32_t* done, int32_t* n_out, int32_t pad_id, hipStream_t stream);
int launch_embed_splice(const int32_t* ids, int P_max, const int32_t* cu, const void* E, const void* patches, void* h, int B,
                        int max_seqlen, int n_patches, int hidden, int vocab, hipStream_t stream, float* h32 = nullptr);   // h32: the rows also as fp32
int launch_rope_kv_write(void* qkv, int ld, int q_off, int k_off, int v_off, const int32_t* cu, int B, int total_rows,
                         const float* cos_t, const float* sin_t, void* kcache, void* vcache, const int32_t* page_table,
                         int max_pages, int Hq, int Hkv, int hd, int page, hipStream_t stream);
// fp8 KV cache: quantise the prefill's K / V rows (K rotated in place by launch_rope_kv_write with null cache pointers) into e4m3 pages + row scales
int launch_kv_quant_rows(void* qkv /* rows written back as bf16(e4m3 x scale): the prefill attention sees what the cache holds */, int ld, int k_off, int v_off, const int32_t* cu, int B, int total_rows, void* k8, void* v8, float* kscale,
                         float* vscale, const int32_t* page_table, int max_pages, int Hkv, int hd, int page, hipStream_t stream);
int launch_resize_bicubic_u8(const uint8_t* src, int B, int H, int W, uint8_t* dst, int OH, int OW, uint8_t* tmp, const int32_t* bounds_h,
                             const int32_t* kk_h, int ksize_h, const int32_t* bounds_v, const int32_t* kk_v, int ksize_v, hipStream_t stream);
// out32: the rows also as fp32; in32 != null: the source rows are fp32 (`in` unused)
int launch_gather_last_rows(const void* in, void* out, const int32_t* cu, int B, int D, hipStream_t stream, float* out32 = nullptr, const float* in32 = nullptr);

// ---- tuning switches (model.hip) ----
// ONE table of named integers, read from the environment (EMMAX_<NAME>) ONCE, when the library first needs a value; after that only
// emmax_tuning_set() (include/emmax.h) changes them.  No launcher calls getenv.  Defaults are the product path; the other values
// are the A/B partners DESIGN.md's measurements quote.
struct EmmaxTune {
    int graph;           // 0: eager launch-ahead decode steps (default); 1: replay a captured hipGraph of the step
    int ks;              // 1: batch 1-2 bf16 projections on decode_ks.hip; 0: decode.hip's LDS-staged GEMV
    int ks_oproj;        // 1: ... including the o-proj with the split merge (one block per CU); 0: decode.hip's kernel
    int ks_oproj_grid;   // grid cap of that o-proj launch (256)
    int km;              // 1: batch >= 3 (and fp8) projections on decode_km.hip; 0: decode_mfma.hip
    int km_down;         // 1: ... including the two-phase down projection
    int km_roll;         // 1: decode_km.hip refills a weight register as soon as its MFMA has issued (rolling ring); 0: a tile's sixteen refills together
    int attn_deep;       // bf16 decode attention: four chunks of keys in flight per wave instead of two (the fp8-cache form always has four): -1 = from 1024 blocks (batch 32), 0 never, 1 always
    int attn_ksplit;     // causal prefill attention (head_dim 128): two key groups per block (8 waves) -- -1 = when the launch leaves <= 1 block per CU (one frame), 0 never, 1 always
    int attn_lazy;       // causal prefill attention (head_dim 128): 1 = lazy reference maximum (round 5), 0 = the running maximum of rounds 1-4 (bit-regression probes)
    int vis_streams;     // vision encode: 1 = the two towers side by side on two streams (default; batches up to 256 frames), 0 = one after the other on the caller's stream
    int attn_nw;         // waves per decode-attention block: 0 = by shape (4; 8 for the one-split form when that leaves <= 256 blocks), 4 / 8 forced
    int streamk;         // 1: stream-K work split in decode_mfma.hip; 0: whole tasks per block
    int fp8_gemv;        // -1: default routing of the batch 1-2 fp8 projections; >= 0: bit mask (1 qkv, 2 o-proj, 4 gate/up, 8 down, 16 lm-head) on the row GEMV
    int attn_nsplit;     // 0: KV splits of the decode attention chosen from (B, kv heads); > 0: forced (rounded down to 2^k, <= 16)
    int attn_direct;     // 1: with one KV split the attention launch writes the normalised bf16 row itself
    int fold_embed;      // 1: layer 0's qkv launch gathers the embedding row itself (batch 1-2, bf16)
    int mfma_xbar;       // 1: decode_mfma.hip orders the activation requests ahead of the weight head with a block barrier
    int gemm_big;        // -1: planned tile geometry; 0 / 1: all small / all big tiles, no split-K
    int gemm_splitk;     // 1: split-K for under-filled long-K GEMMs
    int gemm_sk_big;     // split-K tile geometry: -1 = planned (128 x 128, or 256 x 256 when the cost model prefers it), 0 / 1 = small / big forced (tools)
    int gemm_hybrid;     // 1: a column remainder behind whole rounds of big tiles goes through K-split small tiles when K is long (one-frame prefill gate/up)
    int gemm_normfuse;   // 1: the split-K reduce pass of the one-frame prefill o-proj / down also applies the RMSNorm that follows
    int gemm_deep;       // GEMM main loop: -1 = by geometry (256x256: staggered wave groups, 128x128: deep A ring), 0 = two stages + one barrier per step, 1 = third LDS stage for A, 3 = staggered wave groups (256x256 only)
    int gemm_dbg;        // lab: OR-ed into GemmParams::dbg (16 = the second half of the waves requests its slabs mid-step)
    int gemm_lnfuse;     // 1: LayerNorm / RMSNorm applied by the GEMM that consumes the normalised rows (no separate norm pass)
    int attn_resident;   // -1: resident ViT attention kernel where measured faster; 0: never; 2: whenever it fits (tests)
    int kv_fp8;          // 1: sessions created from now on keep the paged KV cache as e4m3 rows + one fp32 scale per (token, head) row (opt-in: half the attention bytes of a decode step, its own error line in the tests); 0: bf16
    int resid32;         // 1: prefill and decode step keep the residual stream in fp32 (GemmParams::res_f32, GemvParams::h32); 2: the decode step only; 0: bf16 rows (rounds 1-4)
    int exact;           // 1: EXACT NUMERICS (round 6) -- models finalized and sessions created from now on carry fp32 activations end to end, every bf16
                         //    MFMA / dot2 activation operand as two bf16 terms (hi + lo), fp32 attention (fp32 MFMA) over an fp32 KV cache: the
                         //    reference's fp32 CPU arithmetic to ~1e-5 of max|logit| instead of 2.4e-2.  Batch 1-2 sessions on bf16 weights; the model
                         //    keeps the ViT LayerNorms unfolded.  0: the bf16-operand product path (default)
    int epoch;           // bumped by every emmax_tuning_set: sessions drop captured graphs when it moves
};
const EmmaxTune& emmax_tune();

// ---- decode.hip ----
enum { GEMV_QKV = 0, GEMV_RESID = 1, GEMV_GATEUP = 2, GEMV_LMHEAD = 3, GEMV_PLAIN = 4 };
struct GemvParams {
    const void* x;          // bf16 [B, ldx] activations
    const void* W;          // bf16 [rows, ldw]
    void* y;                // output (mode dependent): q buffer / h (in place) / act / plain
    const void* norm_w;     // bf16 [K] RMSNorm weight (NORM modes)
    int K, ldw, ldx, ldy;
    int n_rows;             // weight rows (QKV: (Hq+2Hkv)*hd; GATEUP: 2*inter_p interleaved; LMHEAD: vocab; else output rows)
    int n_groups;           // 2-row groups (set by the launcher)
    int max_parts;          // LMHEAD: capacity of part_val / part_idx in blocks
    int kc;                 // K phase length (set by the launcher)
    int batch;              // MFMA path: runtime batch (set by the launcher)
    float eps;
    // QKV epilogue
    int head_dim, Hq, Hkv, page, max_pages;
    const int32_t* ctx_len;
    const int32_t* page_table;
    const float* cos_t;
    const float* sin_t;
    void* kcache;
    void* vcache;
    // LMHEAD epilogue
    float* part_val;
    int32_t* part_idx;
    float* logits_out;      // optional f32 [B, n_rows]
    // RESID with x = merged attention output: split partials f32 [B][Hq][nsplit][132] (null: x is a bf16 vector)
    const float* attn_part;
    int nsplit;
    void* kv_stage;         // QKV, fp8 KV cache (round 5, opt-in): the new K / V rows go to bf16 [B][Hkv][2][head_dim] instead of the paged cache;
                            //   the attention launch of the step quantises them (one scale per row) and appends them
    const float* wscale;    // per-row fp32 scales when W is an fp8 copy (MFMA path: fragment-major tiles; GEMV: rows in span order); null: bf16
    int n_pairs;            // GEMV, set by the launcher: row pairs (fp8 groups hold two)
    unsigned long long* sk_ws;   // MFMA path: stream-K granules [256 blocks][2 tiles][256] of {f32, tag} (null: whole tasks per block)
    int sk_kt8, sk_q, sk_r;      // MFMA path, set by the launcher: super-steps per task (0: whole tasks), per-block share and remainder
    unsigned int sk_magic;       // ... and ceil(2^32 / sk_kt8)
    int qk_shift;                // MFMA QKV, set by the launcher: log2(head_dim / 32)
    int x_bar;                   // MFMA path, set by the launcher: block barrier between the activation requests and the weight head
    int max_grid;           // 0: default persistent grid; > 0: cap (the K-split o-proj with the split merge runs one block per CU)
    int ks_shift;           // K-split kernel, QKV, set by its launcher: log2(head_dim / 2)
    const int32_t* x_tok;   // K-split kernel, non-null: x row b = x[x_tok[b]] (p.x = embedding table, ids clamped to x_vocab): the embed launch
    void* x_copy;           //   folded into layer 0's qkv; block 0 also copies the rows to x_copy [B, ldx] (the residual stream)
    int x_vocab;
    // fp32 residual stream (round 5, tuning switch resid32): h32 f32 [B, ldh] is the MASTER copy of the hidden rows of a decode step, the
    // bf16 rows (p.x of the NORM modes = p.y of the RESID modes) its mirror.  RESID modes (o-proj, down) add into h32 in place and store
    // bf16(sum) in the mirror; NORM modes (qkv, gate/up, lm-head) read either -- the batch 1-2 kernels the fp32 rows (free there: statistics
    // and x g in fp32, one rounding), the batch >= 3 / fp8 MFMA kernels the mirror (the fp32 rows cost them 1.5 % of a step).  Either way
    // the stream is no longer rounded after each of the 64 additions of a token: the mirror is re-derived from fp32 every time.
    float* h32;
    int ldh;
    // exact numerics (round 6, tuning switch `exact`; decode_ks.hip, batch 1-2): every activation operand enters the dot products as TWO bf16
    // terms (hi + lo of the fp32 value; the fp32 stream h32 must be set), and what the projections hand on stays fp32 -- QKV: p.y = f32 q rows
    // [B, ldy], the K / V rows go to an fp32 paged cache (kcache / vcache as float); GATEUP: p.y = f32 [B, ldy]; RESID over the SwiGLU
    // product: p.x = f32 [B, ldx]; RESID over the attention partials: the merge stays fp32 until it is split.
    int exact;
    long long kv24;         // exact numerics: > 0 = the K / V caches are 24-bit (common.h: x24) -- kcache / vcache point at the bf16 plane of kv24 elements,
                            //   the 8-bit extension plane follows it; 0 = fp32 rows
};
// where the QKV epilogues put element block (head hk, K or V) of batch row b: the paged bf16 cache, or the staging rows of the fp8 KV cache
__device__ __forceinline__ bf16_t* gemv_kv_row(const GemvParams& p, bool is_v, int b, int pg, int pos, int hk) {
    if (p.kv_stage) return (bf16_t*)p.kv_stage + (((size_t)b * p.Hkv + hk) * 2 + (is_v ? 1 : 0)) * p.head_dim;
    return (bf16_t*)(is_v ? p.vcache : p.kcache) + (((size_t)pg * p.Hkv + hk) * p.page + pos % p.page) * p.head_dim;
}
// exact numerics: element d of the (page, kv head, position) row of the fp32 / 24-bit paged cache
__device__ __forceinline__ void gemv_kv_store_x(const GemvParams& p, bool is_v, int pg, int pos, int hk, int d, float v) {
    const size_t idx = (((size_t)pg * p.Hkv + hk) * p.page + pos % p.page) * p.head_dim + d;
    void* base = is_v ? p.vcache : p.kcache;
    if (p.kv24 > 0) {
        const uint32_t u = x24_bits(v);
        ((bf16_t*)base)[idx] = (bf16_t)(u >> 16);
        ((uint8_t*)base + (size_t)p.kv24 * 2)[idx] = (uint8_t)(u >> 8);
    } else {
        ((float*)base)[idx] = v;
    }
}
int launch_decode_gemv(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out = nullptr);
// decode_ks.hip: the batch 1-2 bf16 projections with K split across the waves of a block (activation slice in registers, no
// block-wide stage); returns -2 for a shape it does not take.  launch_decode_gemv tries it first (tuning switch `ks` = 0: never).
int launch_decode_ks(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out = nullptr);
bool decode_ks_enabled();
int decode_gemv_init();   // raise the dynamic-LDS limit of every GEMV instantiation (call once, outside graph capture)
int launch_decode_embed(const int32_t* cur_tok, const void* E, void* h, int B, int hidden, int vocab, hipStream_t stream, float* h32 = nullptr);

struct DecodeAttnParams {
    const void* q;          // bf16 [B, ldq] rotated queries
    const void* kcache;     // bf16 [pages][Hkv][page][hd]
    const void* vcache;
    const int32_t* page_table;
    const int32_t* ctx_len;
    const int32_t* done;    // int32 [B] or null: finished / idle rows read no K/V (their partials are written empty)
    float* part;            // f32 [B][Hq][nsplit][132] = { o[128] un-normalised, m, l, pad }
    void* o_out;            // non-null (one split only): the block normalises its result and writes the bf16 row [B, ldq] itself
    // fp8 KV cache (kv_stage non-null): kcache / vcache are e4m3 bytes [pages][Hkv][page][hd], kscale / vscale fp32 [pages][Hkv][page]; the
    // step's new rows come from kv_stage bf16 [B][Hkv][2][hd] (written by the qkv launch): every block that needs key L - 1 quantises it
    // itself, split 0 appends it to the cache
    const void* kv_stage;
    float* kscale;
    float* vscale;
    int ldq, Hkv, page, max_pages;
    int page_shift;         // log2(page), set by the launcher
    float scale;
    long long kv24;         // exact numerics (launch_x_decode_attn): > 0 = 24-bit caches (GemvParams::kv24), 0 = fp32 rows
};
int decode_attn_nsplit(int B, int Hkv);
int launch_decode_attn(const DecodeAttnParams& p, int B, int Hq, int head_dim, int nsplit, hipStream_t stream);
int launch_x_decode_attn(const DecodeAttnParams& p, int B, int Hq, int head_dim, int nsplit, hipStream_t stream);   // exact.hip: fp32 q, fp32 cache

struct FinishParams {
    const float* part_val;
    const int32_t* part_idx;
    int n_part, B;
    int32_t* cur_tok;
    int32_t* ctx_len;
    int32_t* done;
    int32_t* n_out;
    int32_t* out_ids;
    const int32_t* max_new_p;   // device int[B]: per-row token budget
    // early exit (emmax_session_set_stop): stop_cfg = {n_trigger, n_after}; a row is done n_after tokens after it emitted
    // the trigger id sequence.  stop_m[b] = trigger ids matched so far, stop_after[b] = tokens since the match (-1: none yet)
    const int32_t* stop_ids;
    const int32_t* stop_cfg;
    int32_t* stop_m;
    int32_t* stop_after;
    int max_out, max_ctx;
    int eos_id, pad_id;
    int is_prefill;
};
int launch_decode_finish(const FinishParams& p, hipStream_t stream);
int launch_set_tokens(int32_t* cur_tok, const int32_t* toks, int B, int32_t* done, int32_t* stop_m, int32_t* stop_after, int32_t* max_new,
                      int budget, hipStream_t stream);


// ---- decode_km.hip: batch 3-16 projections with K <= 4096 on MFMA, K split across the waves, activations as register fragments ----
// p.W = the km copy (launch_repack_km: fragment-major tiles; perm 1 / 2 = the row orders that put the qkv RoPE pairs / the
// (gate, up) pairs inside one 16-row tile).  -2: shape outside the kernel, the caller falls back to launch_decode_mfma.
int launch_repack_km(const void* src, int ld, void* dst, int N, int K, int perm, int head_dim, hipStream_t stream);
int launch_decode_km(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out = nullptr);
bool decode_km_enabled();
int decode_km_init();   // raise the dynamic-LDS limit of every instantiation (call once, outside graph capture)
// decode_kmp.hip: batch 17-32 (two batch tiles per weight tile, the K slice in phases), bf16 weights, same copies; launch_decode_km routes there
int launch_decode_kmp(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out = nullptr);
int decode_kmp_init();

// ---- decode_mfma.hip: small-batch (B >= 3) projections on MFMA over the fragment-major weight copy ----
int launch_repack_fm(const void* src, int ld, void* dst, int N, int K, hipStream_t stream);
int launch_quant_fm8(const void* src, int ld, void* dst, float* scales, int N, int K, hipStream_t stream, int perm = 0, int head_dim = 0);   // fp8 e4m3 + per-row scale
bool decode_gemv_fp8_fits(int B, int K);   // the fp8 row GEMV takes this (batch, K); else the MFMA kernel serves it
int launch_quant_rm8(const void* src, int ld, void* dst, float* scales, int N, int K, hipStream_t stream);   // same values, rows in the GEMV's span order
int launch_decode_mfma(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out = nullptr);   // p.W = fragment-major copy
int decode_mfma_init();
#define EMMAX_MFMA_MIN_BATCH 3
