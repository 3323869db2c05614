/*
 * emmax.h -- C ABI of libemmax_hip.so: the MI355X-native (gfx950) Emma-X VLA forward/generate hot path.
 *
 * The reference (declare-lab/Emma-X) has NO FFI/plugin ABI: its extension point is HuggingFace Auto-class registration
 * of pure-Python nn.Modules (experiments/robot/openvla_utils.py:38-41).  This header is therefore the boundary a
 * maintainer would bind from Python (ctypes) in place of the third-party math the reference calls:
 *
 *   emmax_vision_encode*   replaces PrismaticVisionBackbone.forward + PrismaticProjector.forward
 *                          (prismatic/extern/hf/modeling_prismatic.py:114-123, 146-158; native twin
 *                          prismatic/models/backbones/vision/dinosiglip_vit.py:142-147, prismatic/util/nn_utils.py:37-53)
 *   emmax_prefill          replaces the multimodal branch of PrismaticForConditionalGeneration.forward
 *                          (modeling_prismatic.py:362-415: embed, splice [BOS]+patches+text[1:], LlamaForCausalLM prefill)
 *   emmax_decode_step      replaces the cached branch (modeling_prismatic.py:325-341) + one greedy step of
 *                          transformers GenerationMixin.generate (invoked at modeling_prismatic.py:519,
 *                          prismatic/models/vlms/prismatic.py:659-663)
 *   emmax_generate         replaces the whole greedy loop (<= max_new_tokens, EOS stop) without a host sync per token
 *   emmax_prefill_logits   replaces `forward(...).logits` (all positions), for API parity of `forward()`
 *   emmax_op_*             single-kernel entry points, used by the parity tests
 *
 * Conventions: every function returns 0 on success or a negative emmax_status; emmax_last_error() gives the message of
 * the last failure on the calling thread.  All pointers named *_dev are device pointers owned by the caller; the
 * library never allocates device memory (the caller passes arenas sized by the *_bytes queries).  `stream` is a
 * hipStream_t (NULL = default stream).  Handles are thread-compatible (one thread per handle at a time).  bf16 tensors
 * are raw uint16_t bit patterns.  No torch types cross this boundary.
 */
#ifndef EMMAX_H
#define EMMAX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever emmax_config / emmax_tower_config change layout or an entry point changes signature.
 *   1: rounds 1-2;  2: emmax_config grew `decode_fp8` (round 2, not bumped then);  3: round 4 -- emmax_config_size / emmax_tuning_*
 *   added, the lab-only entry points (persistent layer chain, in-attention split merge) removed;  4: round 5 -- emmax_session_*_ex (staging rows
 *   are asked for, the plain calls give none), decode batches / slot counts up to 64 (emmax_model_max_decode_batch). */
#define EMMAX_ABI_VERSION 5

typedef enum emmax_status {
    EMMAX_OK = 0,
    EMMAX_ERR_INVALID = -1,     /* bad argument / shape / config outside the hot path          */
    EMMAX_ERR_MISSING = -2,     /* finalize: a required weight was never bound                  */
    EMMAX_ERR_NOMEM = -3,       /* arena / workspace too small                                  */
    EMMAX_ERR_HIP = -4,         /* a HIP runtime call failed                                    */
    EMMAX_ERR_STATE = -5        /* call order violated (e.g. decode before prefill)             */
} emmax_status;

typedef enum emmax_dtype { EMMAX_BF16 = 0, EMMAX_F32 = 1, EMMAX_U8 = 2, EMMAX_I32 = 3 } emmax_dtype;

typedef void* emmax_stream;            /* hipStream_t */
typedef struct emmax_model emmax_model;
typedef struct emmax_session emmax_session;

/* One timm ViT tower as the reference instantiates it (modeling_prismatic.py:78-101). */
typedef struct emmax_tower_config {
    int32_t embed_dim, depth, num_heads, mlp_hidden;
    int32_t has_cls, n_reg, layerscale;
    int32_t patch, image_size;
    int32_t take_index;                /* block whose output is returned: depth-2 (modeling_prismatic.py:86,100) */
    float ln_eps;
    float mean[3], std[3];             /* per-tower normalisation (processing_prismatic.py:136-139)              */
} emmax_tower_config;

typedef struct emmax_config {
    emmax_tower_config tower[2];       /* [0]=featurizer (DINOv2), [1]=fused_featurizer (SigLIP)                  */
    int32_t hidden, inter, n_layers, n_heads, n_kv_heads, head_dim, vocab;
    float rms_eps, rope_theta;
    int32_t bos_id, eos_id, pad_id;
    int32_t decode_fp8;                /* 1: the decode projections stream an fp8-e4m3 (per-row scale) weight copy, de-quantised
                                          in registers (BASELINE config 5); 0: bf16 weights (the headline path)     */
} emmax_config;

const char* emmax_version(void);
const char* emmax_last_error(void);
int emmax_abi_version(void);
/* sizeof(emmax_config) as the LIBRARY was compiled: a binding whose mirror struct has another size must refuse to call
 * emmax_model_create (it would be read out of bounds).  emma-x_amd/emmax/_lib.py checks it at load time. */
int emmax_config_size(void);

/* Tuning switches: a fixed table of named integers (emma-x_amd/csrc/kernels.h, struct EmmaxTune).  The library reads the
 * environment variable EMMAX_<NAME> for each of them ONCE, the first time any value is needed; afterwards only emmax_tuning_set
 * changes them (no launcher reads the environment).  Every default is the product path; the other values are the A/B partners
 * DESIGN.md quotes.  Names: graph (1 = hipGraph replay of the decode step, BASELINE configs[4]), ks, ks_oproj, ks_oproj_grid, km,
 * km_down, km_roll, streamk, fp8_gemv, attn_nsplit, attn_direct, attn_nw, attn_deep, attn_ksplit, attn_lazy, vis_streams, fold_embed, mfma_xbar, gemm_big (2 = the 128 x 256 x 32 lab tile), gemm_splitk, gemm_sk_big,
 * gemm_hybrid, gemm_normfuse, gemm_deep, gemm_lnfuse, attn_resident, resid32 (1 = fp32 residual stream in prefill and decode, 2 = decode only,
 * 0 = bf16 rows), kv_fp8 (1 = sessions created from now on keep an e4m3 KV cache).
 * Three switches are read when an object is BUILT and frozen in it: gemm_lnfuse at emmax_model_finalize (the LayerNorm fold rewrites the ViT
 * qkv / fc1 weights in place: setting it afterwards does not change a finalized model), km at emmax_model_build_aux (which copies exist),
 * kv_fp8 at emmax_session_create (the cache format).
 * exact (round 6; 0 = off, the default): EXACT NUMERICS -- the reference's fp32 CPU arithmetic (prismatic/models/vlms/prismatic.py:659-663 under
 * BASELINE configs[0]) instead of bf16 operands: fp32 activations end to end, every activation operand of a bf16 MFMA / dot2 as TWO bf16 terms
 * hi + lo (the checkpoint's weights are exact bf16), attention on the fp32 MFMA over an fp32 KV cache.  Read at emmax_model_finalize (the ViT
 * LayerNorms stay unfolded: the fold rounds W .* gamma) and at emmax_session_create (fp32 scratch, fp32 cache = twice the KV bytes).  Exact
 * sessions run on bf16 weights, 8 rows per projection launch (1-2: decode_ks.hip's two-term dot products; 3-8: decode_km.hip with the two terms of a row
 * in the MFMA's sixteen batch columns; larger batches, up to 64 rows / slots, in chunks of 8 that each stream the weights again), slot serving included; emmax_session_exact() tells which kind a session is.  Logits sit
 * ~1e-5 of max|logit| from the fp32 restatement at full depth (default path: 2.4e-2) -- measured cost in DESIGN.md section 6.
 * Not thread-safe against concurrent launches; a session re-captures its decode graph after a change. */
int emmax_tuning_set(const char* name, int value);
int emmax_tuning_get(const char* name, int* value_out);

/* ---- model: weights ------------------------------------------------------------------------------------------------
 * bind every tensor of the HF state dict (key names: vla-scripts/extern/convert_openvla_weights_to_hf.py:74-116) as a
 * bf16 device pointer, then finalize(): all weights are re-laid-out into `arena` (kernel-native: fused QKV rows,
 * 16-row interleaved gate/up, K/N padded to tile multiples, im2col-ordered patch-embed; fp8 models: plus the e4m3 copies).
 * After finalize the bound pointers are no longer referenced and may be freed. */
int emmax_model_create(const emmax_config* cfg, emmax_model** out);
void emmax_model_destroy(emmax_model* m);
int emmax_model_bind_weight(emmax_model* m, const char* hf_key, const void* ptr_dev, int dtype,
                            const int64_t* shape, int ndim);
int64_t emmax_model_arena_bytes(const emmax_model* m);
/* rows of one decode batch / slot set this model can run: 64 (bf16 or fp8 weights) when every LLM projection is a shape the K-split
 * MFMA kernels take (K % 256 == 0 and <= 4096, N <= 32768, intermediate size % 32 (fp8: % 64) == 0 and <= 11264: LLaMA-2-7B is), else 8 (round 6;
 * round 5: 32; rounds 1-4: 8).  Exact-numerics sessions: 64 whatever the shapes (8 rows per launch, larger batches in chunks). */
int emmax_model_max_decode_batch(const emmax_model* m);
int emmax_model_finalize(emmax_model* m, void* arena_dev, int64_t arena_bytes, emmax_stream stream);
/* bf16 models: decode batches >= 3 stream the LLM projections from MFMA-fragment-major copies that a model serving batches 1-2
 * never reads.  They live in a SECOND caller-owned arena, built on demand from the finalized main arena (no bound tensors
 * needed): 13.2 GB at 7B (main arena 15.1 GB; with the tuning switch km = 0 at build time decode_mfma.hip's qkv / gate-up pair is
 * added, +9 GB).  Until it is built, emmax_prefill / emmax_slots_open with B >= 3 return EMMAX_ERR_STATE.  fp8 models keep every
 * e4m3 copy in the main arena: aux_bytes is 0 and build_aux a no-op. */
int64_t emmax_model_aux_bytes(const emmax_model* m);
int emmax_model_build_aux(emmax_model* m, void* aux_arena_dev, int64_t aux_bytes, emmax_stream stream);

/* ---- session: activations workspace + paged KV cache for up to max_batch sequences of <= max_ctx tokens ------------ */
int emmax_session_bytes(const emmax_model* m, int max_batch, int max_prompt, int max_ctx,
                        int64_t* workspace_bytes, int64_t* kv_bytes);
int emmax_session_create(emmax_model* m, int max_batch, int max_prompt, int max_ctx,
                         void* workspace_dev, int64_t workspace_bytes, void* kv_dev, int64_t kv_bytes,
                         emmax_session** out);
/* ... plus `stage_rows` STAGING rows (0 .. min(max_batch, 32)) for overlapped admissions, emmax_slots_prefill_staged below: each costs per-row
 * state and its share of the paged KV region.  The plain calls above are stage_rows = 0. */
int emmax_session_bytes_ex(const emmax_model* m, int max_batch, int max_prompt, int max_ctx, int stage_rows,
                           int64_t* workspace_bytes, int64_t* kv_bytes);
int emmax_session_create_ex(emmax_model* m, int max_batch, int max_prompt, int max_ctx, int stage_rows,
                            void* workspace_dev, int64_t workspace_bytes, void* kv_dev, int64_t kv_bytes,
                            emmax_session** out);
int emmax_session_stage_rows(const emmax_session* s);
/* 1: the session was created under the tuning switch exact = 1 (fp32 activations / fp32 KV cache, see above), 0: bf16-operand path */
int emmax_session_exact(const emmax_session* s);
void emmax_session_destroy(emmax_session* s);

/* frames_u8_dev: uint8 [B,224,224,3] RGB (normalisation fused into the patch gather);  out: bf16 [B,256,hidden].
 * EXACT-NUMERICS sessions exchange patch embeddings as FP32 rows: `patch_embeds_out_dev` here and every `patch_embeds*` argument of the
 * prefill / slot entry points below then hold float [B,256,hidden] (bf16 rows would put 2^-9 back into the input of the fp32 arithmetic). */
int emmax_vision_encode(emmax_session* s, const uint8_t* frames_u8_dev, int B, void* patch_embeds_out_dev,
                        emmax_stream stream);
/* pixel_values_dev: bf16 [B,6,224,224] as PrismaticProcessor emits (already normalised per tower). */
int emmax_vision_encode_pixels(emmax_session* s, const void* pixel_values_bf16_dev, int B, void* patch_embeds_out_dev,
                               emmax_stream stream);
/* raw concatenated tower features bf16 [B,256,D0+D1] of the last emmax_vision_encode* call (parity tests). */
int emmax_vision_features(emmax_session* s, int B, void* feats_out_dev, emmax_stream stream);

/* ids_dev: int32 [B,P_max] (row b uses the first lens_host[b] ids, ids[b][0] = BOS); patch_embeds: bf16 [B,256,hidden].
 * Builds [BOS]+patches+text[1:] per row (no padding: rows are packed), runs the decoder stack, fills the KV cache and
 * leaves the greedy first token of every row as the session's "current token". */
int emmax_prefill(emmax_session* s, const int32_t* ids_dev, const int32_t* lens_host, int B, int P_max,
                  const void* patch_embeds_dev, emmax_stream stream);
/* Language-only prefill (no image): the `pixel_values is None` branch of forward (modeling_prismatic.py:343-359). */
int emmax_prefill_text(emmax_session* s, const int32_t* ids_dev, const int32_t* lens_host, int B, int P_max, emmax_stream stream);
/* f32 logits of every prefill position, packed rows [sum_b S_b, vocab] (S_b = 256 + lens[b]); valid after prefill. */
int emmax_prefill_logits(emmax_session* s, float* logits_out_dev, emmax_stream stream);
/* f32 last-position logits [B,vocab] of the most recent prefill/decode step (parity tests; costs one extra pass). */
int emmax_last_logits(emmax_session* s, float* logits_out_dev, emmax_stream stream);

/* One greedy step for all rows: consumes each row's current token, appends KV, leaves argmax as the new current token
 * and records it in the session's output buffer.  Rows that already emitted EOS (or ran out of context) emit pad_id.
 * Reads all step-varying state from device memory, so it is hipGraph-capturable. */
int emmax_decode_step(emmax_session* s, emmax_stream stream);
/* Overwrite the current token of every row with a CALLER-supplied one: tokens_dev int32 [B] (teacher-forced scoring, external
 * sampling loops, the HF cached step `forward(input_ids[B,1], past_key_values)`).  The rows decode again whatever the engine's
 * own greedy prediction was: done flags, stop-rule state and token budgets are cleared.  EMMAX_ERR_NOMEM when a row's context
 * is full (the next append would leave the KV pages), EMMAX_ERR_STATE while request slots are open. */
int emmax_set_current_tokens(emmax_session* s, const int32_t* tokens_dev, emmax_stream stream);
/* Run up to max_new_tokens steps (including the token produced by prefill) -- eager launch-ahead by default, replays of a
 * captured hipGraph of emmax_decode_step with the tuning switch graph = 1; no host synchronisation per token; stops early once
 * all rows are done if stop_on_eos != 0.
 * out_ids_dev int32 [B,max_new_tokens] (pad_id after a row's EOS), out_lens_dev int32 [B] (tokens incl. EOS). */
int emmax_generate(emmax_session* s, int max_new_tokens, int stop_on_eos, int32_t* out_ids_dev, int32_t* out_lens_dev,
                   emmax_stream stream);

/* 1 when emmax_generate is replaying a captured hipGraph of the step (0: eager launches). */
int emmax_session_graph_active(emmax_session* s);
/* Measurement hook (bench.py `roofline`): launch decode stage `stage` (0 qkv GEMV, 1 paged attention, 2 o-proj GEMV,
 * 3 gate/up GEMV, 4 down GEMV: once per layer; 5 lm-head GEMV+argmax) `reps` sweeps on `stream`, bracketed by HIP
 * events on that stream; returns the mean duration of one launch in microseconds.  Needs a prefilled session; the
 * residual stream it leaves behind is garbage (run a new prefill afterwards). */
int emmax_profile_decode_stage(emmax_session* s, int stage, int reps, float* avg_us_out, emmax_stream stream);

/* ---- early exit + slot serving (continuous batching) -------------------------------------------------
 * The reference always decodes to EOS / max_new_tokens (prismatic.py:655-664, generate(max_new_tokens=512)) although the
 * Solver only reads the line after "POLICIES:" (policy_parser: extract_action_policies).  These entry points let a serving
 * loop stop a row as soon as its action line is complete and refill the freed row while the others keep decoding. */
/* Device-side stop rule of every later decode step: a row is done once it has emitted the id sequence trigger_ids[0..n)
 * followed by n_after more tokens (the emitted prefix is exactly the prefix of the full greedy generation).  n_trigger = 0
 * clears the rule; at most 16 ids; a mismatch restarts the match at the current token (no overlapping-prefix handling). */
int emmax_session_set_stop(emmax_session* s, const int32_t* trigger_ids_host, int n_trigger, int n_after, emmax_stream stream);
/* Turn the first n_slots rows of the session into independent, idle request slots (n_slots <= max_batch, <= emmax_model_max_decode_batch). */
int emmax_slots_open(emmax_session* s, int n_slots, emmax_stream stream);
/* Prefill ONE request into `slot` without disturbing the other slots: prompt ids (device int32[len]), its
 * [n_patches, hidden] bf16 patch embeddings (device; NULL = language-only) and its token budget.  The first generated
 * token is in place afterwards. */
int emmax_slot_prefill(emmax_session* s, int slot, const int32_t* ids_dev, int len, const void* patch_embeds_dev, int max_new,
                       emmax_stream stream);
/* Prefill n requests into the CONSECUTIVE slots slot0 .. slot0 + n - 1 in one packed pass (what a scheduler does when several
 * slots are free at once: eight one-row prefills cost ~1.6x one eight-row prefill): ids_dev int32 [n][P_max] (row i: lens_host[i]
 * ids, the rest ignored), patch_embeds_dev [n][n_patches, hidden] bf16 (NULL = language-only), max_new_host[i] the token budgets.
 * The other slots are not disturbed; every row's first generated token is in place afterwards. */
int emmax_slots_prefill(emmax_session* s, int slot0, int n, const int32_t* ids_dev, int P_max, const int32_t* lens_host,
                        const void* patch_embeds_dev, const int32_t* max_new_host, emmax_stream stream);
/* Overlapped admission.  emmax_slot_prefill and emmax_slots_prefill run on the stream the slots decode on: while a request is prefilled (16 ms for one
 * 7B row, 70 ms for eight) the other slots stand still.  The staged pair removes that stall: a session created with
 * emmax_session_create_ex holds `stage_rows` STAGING rows beside its decode rows (per-row state, output row, KV pages).
 *   emmax_slots_prefill_staged  prefills n requests into the staging rows on `stream`, which must be a stream of its own (not the
 *                               default stream, not the one the decode steps run on): the call touches nothing a decode step reads,
 *                               so decode steps may run beside it.  Arguments as emmax_slots_prefill without slot0.
 *   emmax_slots_commit          moves staged request staged_idx_host[i] (0 .. n_staged-1 of the last staged batch) into slot
 *                               slots_host[i] (idle, released) on the DECODE stream, between two steps: per-row state and the output
 *                               row are copied and the page-table rows swapped -- no K/V moves.  A staged batch may be committed
 *                               piecemeal, as slots free up.  The caller orders it after the staged prefill (event), and orders the
 *                               NEXT staged prefill after the last commit of the batch.
 * One staged batch at a time.  Host scheduler: emmax/serving.py (SlotScheduler, overlap=True). */
int emmax_slots_prefill_staged(emmax_session* s, int n, const int32_t* ids_dev, int P_max, const int32_t* lens_host,
                               const void* patch_embeds_dev, const int32_t* max_new_host, emmax_stream stream);
int emmax_slots_commit(emmax_session* s, const int32_t* staged_idx_host, const int32_t* slots_host, int n, emmax_stream stream);
/* n_steps greedy decode steps over all slots (no host synchronisation); idle / finished slots stay put. */
int emmax_slots_step(emmax_session* s, int n_steps, emmax_stream stream);
/* Copy the per-slot done flags and generated-token counts to device buffers int32[n_slots] (asynchronous on `stream`). */
int emmax_slots_state(emmax_session* s, int32_t* done_dev, int32_t* n_out_dev, emmax_stream stream);
/* Copy the first n generated ids of `slot` to a device buffer. */
int emmax_slot_output(emmax_session* s, int slot, int32_t* ids_dev, int n, emmax_stream stream);
/* Mark `slot` idle again (empty context: its share of a batched step then reads no K/V). */
int emmax_slot_release(emmax_session* s, int slot, emmax_stream stream);

/* ---- single-kernel entry points (parity tests + micro-benchmarks) -------------------------------------------------- */
/* C[M,N] = epilogue(A[M,K] @ W[N,K]^T): bf16 in, fp32 accumulate on MFMA.  K % 64 == 0, N % 128 == 0.
 * bias/scale: bf16 [N] or NULL; residual: bf16 [M,ldr] or NULL; act: 0 none, 1 exact-erf GELU, 2 SwiGLU over
 * 16-column interleaved (gate,up) groups (then C is [M,N/2]); out_f32: store fp32 instead of bf16. */
int emmax_op_gemm(const void* A_dev, int lda, const void* W_dev, int ldw, void* C_dev, int ldc, int M, int N, int K,
                  const void* bias_dev, int act, const void* scale_dev, const void* residual_dev, int ldr, int out_f32,
                  emmax_stream stream);
/* The same GEMM with `ksplit` K slices per 128x128 tile (fp32 partial tiles in ws_dev, >= ksplit*M*N*4 bytes) and a reduce +
 * epilogue pass: the path the session takes by itself for under-filled problems with a long K (prefill o / down at one
 * frame, batch-1 ViT fc2).  act in {0, 1, 2}.  ksplit = 0: the launch plan a session stage runs with this scratch -- whole
 * tiles, split-K, or whole rounds of 256x256 tiles + the remaining tile columns K-split (one-frame prefill gate/up). */
int emmax_op_gemm_splitk(const void* A_dev, int lda, const void* W_dev, int ldw, void* C_dev, int ldc, int M, int N, int K,
                         const void* bias_dev, int act, const void* scale_dev, const void* residual_dev, int ldr, int out_f32,
                         int ksplit, void* ws_dev, int64_t ws_bytes, emmax_stream stream);
/* The launch plan emmax_op_gemm / a session stage would run for this problem, as text (host only, no device work): which tile
 * geometry, row / column parts, split-K slices, and whether the split-K reduce pass also applies the RMSNorm behind the projection
 * (with_norm; one-frame prefill o-proj / down).  ws_bytes = size of the split-K scratch (0: none).  e.g. M = 768, N = 22016, K = 4096,
 * act = 2, 64 MB scratch: "hybrid cols 0..21760: big | cols 21760..22016: splitk ks=8".  has_residual = 2 (with out_f32): the residual and the
 * result are fp32 rows -- the prefill's fp32 residual stream (round 5). */
int emmax_gemm_plan(int M, int N, int K, int act, int out_f32, int has_ln, int has_residual, int with_norm, int64_t ws_bytes, char* text_out,
                    int text_len);
/* LayerNorm folded into the projection that consumes it, as the ViT qkv / fc1 stages run (timm Block: norm1 -> attn.qkv, norm2 ->
 * mlp.fc1): C[M,N] = act(LN(X; gamma, beta, eps) @ W^T + bias) without materialising LN(X).  W_dev bf16 [N, ldw] is REWRITTEN in
 * place to bf16(W .* gamma) (what emmax_model_finalize does once per model); stats_ws f32 [M][2], ln_s_ws / ln_c_ws f32 [N] are
 * caller scratch.  act in {0, 1}; K % 64 == 0, N % 128 == 0, K = the LayerNorm width. */
int emmax_op_gemm_ln(const void* X_dev, int ldx, void* W_dev, int ldw, void* C_dev, int ldc, int M, int N, int K, const void* gamma_dev,
                     const void* beta_dev, const void* bias_dev, float eps, int act, float* stats_ws_dev, float* ln_s_ws_dev,
                     float* ln_c_ws_dev, emmax_stream stream);
int emmax_op_layernorm(const void* x_dev, void* y_dev, const void* w_dev, const void* b_dev, int rows, int D, float eps,
                       emmax_stream stream);
int emmax_op_rmsnorm(const void* x_dev, void* y_dev, const void* w_dev, int rows, int D, float eps, emmax_stream stream);
/* softmax(QK^T * scale [+causal]) V over a packed qkv buffer: token t of sequence b lives at row cu_seqlens[b]+t of
 * qkv_dev (bf16, row stride ld_qkv elements); q head h at column q_off + h*head_dim, k/v head h/(Hq/Hkv) at k_off/v_off.
 * out: bf16 rows of Hq*head_dim (row stride ld_out).  head_dim in {64,72,128}. */
int emmax_op_attention(const void* qkv_dev, int ld_qkv, int q_off, int k_off, int v_off, void* out_dev, int ld_out,
                       const int32_t* cu_seqlens_dev, int B, int max_seqlen, int Hq, int Hkv, int head_dim, float scale,
                       int causal, emmax_stream stream);
/* Split-KV decode attention over a paged cache (the kernel of emmax_decode_step; HF cached attention at q_len = 1,
 * modeling_prismatic.py:325-341): row b attends to keys 0..ctx_len_dev[b] (inclusive: the key appended by this step's qkv
 * kernel sits at position ctx_len[b]).  q: bf16 [B, Hq*128] rotated queries; k/vcache: bf16 [n_pages][Hkv][page][128];
 * page_table: int32 [B][max_pages] (token t of row b lives in page page_table[b][t / page]); done_dev: int32 [B] or NULL
 * (rows flagged done read no K/V).  Writes the un-merged partials f32 [B][Hq][nsplit][132] = {o[128] un-normalised, m, l,
 * pad}; the decode step merges them in the o-proj prologue (with ONE split the session's launch normalises and writes the
 * bf16 row itself).  nsplit: power of two <= 16, or 0 = what the session picks
 * for this (B, Hkv) (returned through nsplit_out when non-NULL).  head_dim 128, page = 2^k, max_pages <= 512. */
int emmax_op_decode_attention(const void* q_dev, const void* kcache_dev, const void* vcache_dev, const int32_t* page_table_dev,
                              const int32_t* ctx_len_dev, const int32_t* done_dev, float* part_out_dev, int B, int Hq, int Hkv,
                              int page, int max_pages, int nsplit, float scale, int* nsplit_out, emmax_stream stream);
/* The ONE-split form of the same kernel, as the decode step launches it when a (row, kv head) has a single KV split (batch >= 5 at
 * 32 heads): the block holds the head's whole result, normalises it and writes o_out_dev bf16 [B, Hq*128] -- the row the o-proj
 * reads -- instead of partials (rows flagged done: zeros). */
int emmax_op_decode_attention_direct(const void* q_dev, const void* kcache_dev, const void* vcache_dev, const int32_t* page_table_dev,
                                     const int32_t* ctx_len_dev, const int32_t* done_dev, void* o_out_dev, int B, int Hq, int Hkv,
                                     int page, int max_pages, float scale, emmax_stream stream);
/* The same kernel over the opt-in fp8 KV cache (round 5, tuning switch kv_fp8 at emmax_session_create): K / V pages hold e4m3 rows
 * [pages][Hkv][page][128] with one power-of-two fp32 scale per row (the smallest with amax / scale <= 448), kscale / vscale fp32 [pages][Hkv][page].  The key of THIS step
 * (position ctx_len[b]) is NOT in the cache yet: it waits as bf16 in kv_stage [B][Hkv][2][128] (K row, V row), every block quantises
 * it itself and split 0 appends bytes + scale.  o_out != NULL: the one-split direct form (bf16 [B, Hq*128]); else partials as
 * emmax_op_decode_attention. */
int emmax_op_decode_attention_kv8(const void* q_dev, void* kcache8_dev, void* vcache8_dev, float* kscale_dev, float* vscale_dev,
                                  const void* kv_stage_dev, const int32_t* page_table_dev, const int32_t* ctx_len_dev, const int32_t* done_dev,
                                  float* partials_out_dev, void* o_out_dev, int B, int Hq, int Hkv, int page, int max_pages, int nsplit,
                                  float scale, emmax_stream stream);
/* Decode-path weight-streaming GEMV: y[b,n] = sum_k x[b,k] W[n,k]  (bf16 in, fp32 accumulate, bf16 out), B <= 8. */
int emmax_op_gemv(const void* x_dev, const void* W_dev, void* y_dev, int B, int N, int K, emmax_stream stream);

/* Pillow-exact antialiased bicubic resize of uint8 RGB frames [B,H,W,3] -> [B,OH,OW,3] on the device (the `resize-naive`
 * transform of processing_prismatic.py:136 for non-224 cameras).  bounds_* int32 [out,2] = (first input index, taps),
 * kk_* int32 [out,ksize] = 22-bit fixed-point taps as Pillow's precompute_coeffs/normalize_coeffs_8bpc produce them
 * (emmax/resize.py builds them); tmp: uint8 [B,H,OW,3] scratch, needed when both axes change. */
int emmax_op_resize_bicubic_u8(const uint8_t* src_dev, int B, int H, int W, uint8_t* dst_dev, int OH, int OW, uint8_t* tmp_dev,
                               const int32_t* bounds_h_dev, const int32_t* kk_h_dev, int ksize_h, const int32_t* bounds_v_dev,
                               const int32_t* kk_v_dev, int ksize_v, emmax_stream stream);

/* fp8 variant: quantise a row-major bf16 [N,ld] matrix to e4m3 fragment-major tiles + fp32 per-row scales (N % 16, K % 64),
 * and the matching small-batch projection (activations bf16, weights de-quantised in registers).  B <= 8: decode_mfma.hip; 9-32 rows:
 * the K-split kernels over the same tiles (decode_km.hip 9-16: K % 512 == 0 up to 4096 or the phased form above; decode_kmp.hip 17-32:
 * K % 64 == 0, K >= 512, the widest wave share <= 1408 elements), EMMAX_ERR_INVALID outside those shapes. */
int emmax_op_quant_fm8(const void* W_dev, int ld, void* W8_fm_out_dev, float* scales_out_dev, int N, int K, emmax_stream stream);
int emmax_op_gemm_small_fp8(const void* x_dev, const void* W8_fm_dev, const float* scales_dev, void* y_dev, int B, int N, int K,
                            emmax_stream stream);
/* Batch 3-32 decode projection on the K-split MFMA kernels (decode_km.hip up to 16 rows, decode_kmp.hip above; K > 4096: the phased kernel of the
 * down projection, y = W x through a zeroed residual; what emmax_decode_step runs for qkv / o-proj / gate-up /
 * lm-head at batch >= 3): W_km = emmax_op_repack_km(W row-major [N, ld]) -- fragment-major 16-row tiles, perm 0 natural row order,
 * 1 = qkv (rows d and d + head_dim/2 of a head in one tile), 2 = gate/up (gate_g and up_g in one tile; source in the 16-row
 * interleaved order of the model arena).  y bf16 [B, N] = x bf16 [B, K] W^T (perm 0).  K % 256 == 0, K <= 4096. */
int emmax_op_repack_km(const void* W_dev, int ld, void* W_km_dev, int N, int K, int perm, int head_dim, emmax_stream stream);
int emmax_op_gemm_small_km(const void* x_dev, const void* W_km_dev, void* y_dev, int B, int N, int K, emmax_stream stream);
/* fp8 rows for batch 1-2: the same e4m3 values and scales as emmax_op_quant_fm8, laid out as N rows of K bytes in the span
 * order the dot-product GEMV streams (decode.hip, emmax_quant_rm8_kernel; K % 16 == 0), and that GEMV (1 <= B <= 2). */
int emmax_op_quant_rm8(const void* W_dev, int ld, void* W8_rows_out_dev, float* scales_out_dev, int N, int K, emmax_stream stream);
int emmax_op_gemv_fp8(const void* x_dev, const void* W8_rows_dev, const float* scales_dev, void* y_dev, int B, int N, int K,
                      emmax_stream stream);
/* Small-batch decode projection on MFMA: y[b,n] = sum_k x[b,k] W[n,k], weights in the MFMA-fragment-major layout that
 * emmax_op_repack_fm produces from a row-major [N,ld] matrix (N % 16 == 0, K % 32 == 0); 1 <= B <= 8. */
int emmax_op_repack_fm(const void* W_dev, int ld, void* W_fm_out_dev, int N, int K, emmax_stream stream);
int emmax_op_gemm_small(const void* x_dev, const void* W_fm_dev, void* y_dev, int B, int N, int K, emmax_stream stream);

/* ---- exact numerics (tuning switch exact), kernel by kernel: fp32 operands in, fp32 results out (tests/test_exact_gpu.py).
 * hl_ws: device scratch for the two-term bf16 image of the activation operand, 4 bytes per (padded) element.
 *   emmax_op_x_gemm       C32[M, N] = act(A32[M, K] W[N, K]^T + bias) (+ residual32): A split into hi + lo, both through the bf16 MFMAs
 *                         (replaces F.linear on fp32 activations; act 0 / 1 GELU / 2 SwiGLU over 16-column (gate, up) groups, C32 [M, N / 2])
 *   emmax_op_x_rownorm    mode 0: y = hi + lo of x (the split alone); 1: HF LlamaRMSNorm in fp32; 2: F.layer_norm in fp32 -- y32 = the two
 *                         terms the consumer GEMM would read, joined
 *   emmax_op_x_attention  softmax(q k^T scale [causal]) v on the fp32 MFMA over packed fp32 qkv rows -> HL rows in hl_ws (pitch 2 * pad64(Hq *
 *                         head_dim)); emmax_op_x_join widens HL rows to fp32 (timm Attention / HF SDPA: modeling_prismatic.py:114-123,404-415)
 *   emmax_op_x_decode_attention  emmax_op_decode_attention over fp32 q rows and the exact-numerics paged cache (same partial layout): kv24_elems = 0:
 *                         fp32 rows [pages][Hkv][page][128] (tuning switch exact = 2); > 0: the 24-bit cache of exact = 1 -- per operand a bf16 plane of
 *                         kv24_elems elements (the top 16 bits of the fp32 value rounded to 24 bits) followed by an 8-bit extension plane */
int emmax_op_x_gemm(const float* A32_dev, int lda, const void* W_dev, int ldw, float* C32_dev, int ldc, int M, int N, int K, const void* bias_dev, int act,
                    const float* residual32_dev, int ldr, void* hl_ws_dev, void* ws_dev, int64_t ws_bytes, emmax_stream stream);
int emmax_op_x_rownorm(int mode, const float* x_dev, float* y32_dev, const void* w_dev, const void* b_dev, int rows, int D, float eps, void* hl_ws_dev,
                       emmax_stream stream);
int emmax_op_x_attention(const float* qkv32_dev, int ld_qkv, int q_off, int k_off, int v_off, const int32_t* cu_seqlens_dev, int B, int max_seqlen, int Hq,
                         int Hkv, int head_dim, float scale, int causal, void* hl_ws_dev, emmax_stream stream);
int emmax_op_x_join(const void* hl_dev, float* out32_dev, int rows, int D, emmax_stream stream);
int emmax_op_x_decode_attention(const float* q32_dev, const void* kcache_dev, const void* vcache_dev, int64_t kv24_elems, const int32_t* page_table_dev,
                                const int32_t* ctx_len_dev, const int32_t* done_dev, float* part_out_dev, int B, int Hq, int Hkv, int page, int max_pages,
                                int nsplit, float scale, emmax_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* EMMAX_H */
