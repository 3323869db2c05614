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
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

enum { MODE_QKV = 0, MODE_RESID = 1, MODE_GATEUP = 2, MODE_LMHEAD = 3, MODE_PLAIN = 4 };

__device__ __forceinline__ u32x4_t ld_nt(const u32x4_t* p) { return __builtin_nontemporal_load(p); }

constexpr int PSTRIDE = EMMAX_PSTRIDE;   // floats per attention split partial: 128 o + m + l + 2 pad (16-byte aligned rows)

// ---------------------------------------------------------------------------------------------------------------------
// GEMV.  Persistent blocks of 512 threads = 8 waves; block b owns a contiguous range of row GROUPS, its waves interleave
// inside the range.  A group is 2 weight rows: (row d, row d+hd/2) of one head for QKV, (gate_i, up_i) for GATEUP, two
// consecutive rows otherwise.  The activation prologue (RMSNorm / attention-split merge + staging into LDS) runs once
// per block; a wave keeps one block of 2 rows x 8 x 16 B of weights in flight: the block is consumed, its successor (the
// next 8 steps of the group, or the head of the next group) is requested, and only then is a finished group reduced;
// the very first block is requested before the prologue (qkv / o-proj: right behind the prologue's own loads).
// (tools/gemv_sweep.hip: this structure streams 180 MB at ~6.1 TB/s, 97 % of a read-only kernel with the same pattern.)
// ---------------------------------------------------------------------------------------------------------------------
constexpr int GW = 8;   // waves per GEMV block

// two fp8 e4m3 (the low or the high half of a dword) -> two bf16, exact
template <bool HI>
__device__ __forceinline__ uint32_t fp8x2_to_bf16x2(uint32_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, HI));
}

// row-major bf16 [N, ld] -> fp8 e4m3 (OCP) rows of K bytes in the GEMV's span order + one fp32 scale per row (amax / 448, the
// values of emmax_quant_fm8_kernel).  A row is cut into spans of 128 chunks of 8 elements (the last one shorter: nc chunks, nc
// even); the 16-byte granule l of a span holds chunk l (bytes 0-7) and chunk nc/2 + l (bytes 8-15).  One block per row; K % 16 == 0.
__global__ __launch_bounds__(256) void emmax_quant_rm8_kernel(const bf16_t* __restrict__ src, int ld, uint8_t* __restrict__ dst,
                                                             float* __restrict__ scales, int N, int K) {
    const int n = blockIdx.x, tid = threadIdx.x;
    __shared__ float red[4];
    const bf16_t* row = src + (size_t)n * ld;
    float amax = 0.f;
    for (int c = tid; c < K / 8; c += 256) {
        const u32x4_t v = *(const u32x4_t*)(row + c * 8);
#pragma unroll
        for (int j = 0; j < 4; ++j) amax = fmaxf(amax, fmaxf(fabsf(bf_lo(v[j])), fabsf(bf_hi(v[j]))));
    }
    amax = wave_max(amax);
    if ((tid & 63) == 0) red[tid >> 6] = amax;
    __syncthreads();
    amax = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float scale = amax > 0.f ? amax / 448.0f : 1.0f;
    if (tid == 0) scales[n] = scale;
    for (int c = tid; c < K / 8; c += 256) {
        const u32x4_t v = *(const u32x4_t*)(row + c * 8);
        const int sp = c >> 7, ci = c & 127, nc2 = min(128, K / 8 - sp * 128) >> 1;
        const int l = ci < nc2 ? ci : ci - nc2, half = ci < nc2 ? 0 : 1;
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[0]) / scale, bf_hi(v[0]) / scale, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[1]) / scale, bf_hi(v[1]) / scale, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[2]) / scale, bf_hi(v[2]) / scale, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[3]) / scale, bf_hi(v[3]) / scale, hi, true);
        u32x2_t w = {(uint32_t)lo, (uint32_t)hi};
        *(u32x2_t*)(dst + (size_t)n * K + sp * 1024 + l * 16 + half * 8) = w;
    }
}

// B <= 2: two resident blocks per CU (<= 128 VGPRs); larger batches keep more accumulators and run one block per CU
#ifdef DECODE_LAB_TRACE
__device__ unsigned long long g_gemv_trace[1024 * 8];   // [block][stamp]: s_memrealtime (100 MHz) of wave 0
#define GEMV_STAMP(k) do { if (threadIdx.x == 0 && blockIdx.x < 1024) g_gemv_trace[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define GEMV_STAMP(k) do { } while (0)
#endif
// F8 > 0 (B <= 2): the weights are the row-major e4m3 copy of emmax_quant_rm8_kernel above + one fp32 scale per row.  A 16-byte
// load is then 16 weights: a step covers a SPAN of up to 128 activation chunks (1024 elements), and lane l of the span holds the
// weights of chunks (l, nc/2 + l) of it (nc = chunks in the span), so that the two LDS reads of a step are each consecutive
// over the lanes, like the bf16 path's one.  Rows are half as long in bytes, so the block shape follows the matrix (the launcher
// picks): F8 = 3 -- groups of FOUR rows (two pairs) x 4 steps, the bf16 path's 16 KiB per wave in flight, for matrices with
// enough rows to give every wave a group (qkv, gate/up, lm-head); F8 = 1 / 2 -- two rows x 4 / 12 steps for the 4096-row
// matrices, whose 2048 pairs are one per wave of a one-block-per-CU grid: the WHOLE pair is requested up front (o-proj: K = 4096
// is four spans; down: K = 11008 is eleven -- 96 weight registers, which one block per CU affords).  De-quantisation is exact (e4m3 fits bf16:
// v_cvt_scalef32_pk_bf16_fp8 with scale 1, two values per instruction, ~4.5 clocks), the products accumulate in fp32 and the row
// scale multiplies the reduced sum.
template <int B, int MODE, bool NORM, bool XATTN = false, int F8 = 0>
__global__ __launch_bounds__(GW * 64, (B <= 2 && F8 != 1 && F8 != 2 ? 4 : 2)) void emmax_decode_gemv_kernel(GemvParams p) {
    GEMV_STAMP(0);
    constexpr bool FP8 = F8 > 0;
    static_assert(!FP8 || B <= 2, "the fp8 GEMV serves batch 1-2");
    constexpr int NR = F8 == 3 ? 4 : 2;            // weight rows per group
    constexpr int NP = NR / 2;                     // row PAIRS per group (the epilogues work on pairs)
    constexpr int U = F8 == 2 ? 12 : F8 ? 4 : 8;   // 16-byte loads per row per block (bf16: 8 * 64 lanes * 8 elements = 4096 elements)
    constexpr int NT = GW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    u32x4_t* xs = (u32x4_t*)smem;   // [B][KC/8 + 1] 16-byte chunks; chunk nch of a row is zero (lanes past the end of K read it)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bf16_t* __restrict__ W = (const bf16_t*)p.W;
    const int K = p.K;
    const int KC = p.kc;   // elements per K phase (multiple of 8)
    const bool multi_phase = KC < K;
    const int XS = (KC >> 3) + 1;   // chunks per staged row

    // contiguous share of the groups for this block
    const int G = gridDim.x, bid = blockIdx.x;
    const int gq = p.n_groups / G, gr = p.n_groups % G;
    const int g_lo = bid * gq + min(bid, gr), g_hi = g_lo + gq + (bid < gr ? 1 : 0);
    const int rounds = (g_hi - g_lo + GW - 1) / GW;

    auto pair_rows = [&](int g, int& r0, int& r1) {   // g: pair index (= group index on the bf16 path)
        if (MODE == MODE_QKV) {
            const int half = p.head_dim >> 1;
            const int hb = g / half, d = g - hb * half;
            r0 = hb * p.head_dim + d;
            r1 = r0 + half;
        } else if (MODE == MODE_GATEUP) {
            r0 = (g >> 4) * 32 + (g & 15);
            r1 = r0 + 16;
        } else {
            r0 = 2 * g;
            r1 = min(2 * g + 1, p.n_rows - 1);
        }
    };

    // The wave's work is a linear sequence of 8-step blocks: for every round (group) x K phase x 512-chunk block.
    // A step = one 16-byte load per lane per row.  The producer cursor runs exactly one block ahead of the consumer:
    // once the current block is consumed, the next one is requested into the same registers (sixteen loads back to back),
    // across group and phase boundaries.
    const int n_phase = (K + KC - 1) / KC;
    struct Cursor { int rd, ph, blk; };
    auto phase_nch = [&](int ph) { return min(KC, K - ph * KC) >> 3; };           // 16-byte chunks in a phase
    auto phase_nblk = [&](int ph) { return FP8 ? ((phase_nch(ph) + 127) / 128 + U - 1) / U : (phase_nch(ph) + 64 * U - 1) / (64 * U); };
    auto advance = [&](Cursor& c) {
        if (++c.blk >= phase_nblk(c.ph)) {
            c.blk = 0;
            if (++c.ph >= n_phase) { c.ph = 0; ++c.rd; }
        }
    };
    u32x4_t wr[NR][U];
    const u32x4_t* w0p = nullptr;   // row pointers of the producer's current (group, phase)
    const u32x4_t* w1p = nullptr;
    const u32x4_t* w8p[NR];         // fp8 rows
#pragma unroll
    for (int r = 0; r < NR; ++r) w8p[r] = nullptr;
    auto producer_rows = [&](const Cursor& c) {
        const int g = g_lo + c.rd * GW + wave;
        int r0, r1;
        if (FP8) {   // (single phase; ldw = bytes per row)
            const uint8_t* W8 = (const uint8_t*)p.W;
            const int gg = min(g, g_hi - 1);
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                pair_rows(min(gg * NP + i, p.n_pairs - 1), r0, r1);
                w8p[2 * i] = (const u32x4_t*)(W8 + (size_t)r0 * p.ldw);
                w8p[2 * i + 1] = (const u32x4_t*)(W8 + (size_t)r1 * p.ldw);
            }
            return;
        }
        pair_rows(min(g, g_hi - 1), r0, r1);
        w0p = (const u32x4_t*)(W + (size_t)r0 * p.ldw + c.ph * KC);
        w1p = (const u32x4_t*)(W + (size_t)r1 * p.ldw + c.ph * KC);
    };
    // FP8: chunks of span s = step u of block blk, and whether this lane holds weights of it
    auto span_nc = [&](int s) { return min(128, (K >> 3) - s * 128); };
    auto issue_step = [&](const Cursor& c, int u, bool active) {
        if constexpr (FP8) {
            const int sp = c.blk * U + u;
            const bool ok = active && lane < (span_nc(sp) >> 1);
            const int at = sp * 64 + lane;
#pragma unroll
            for (int r = 0; r < NR; ++r) wr[r][u] = ok ? ld_nt(w8p[r] + at) : (u32x4_t){0u, 0u, 0u, 0u};
        } else {
            const int ch = c.blk * 64 * U + u * 64 + lane;
            const bool ok = active && ch < phase_nch(c.ph);
            wr[0][u] = ok ? ld_nt(w0p + ch) : (u32x4_t){0u, 0u, 0u, 0u};
            wr[1][u] = ok ? ld_nt(w1p + ch) : (u32x4_t){0u, 0u, 0u, 0u};
        }
    };

    // ---- head of the weight stream: it does not depend on x.  UNCONDITIONAL loads (clamped addresses; lanes and waves past
    // the end fetch a valid row again and meet the zero chunk of x), so that hipcc can count them: the prologue's own loads
    // are requested FIRST where they are a fixed handful, and waited for with vmcnt(16) while the 64 MB first burst of the
    // chip is still landing -- loads return in order, so x queued BEHIND the burst arrived ~10 us into the launch ----
    Cursor P = {0, 0, 0}, Cc = {0, 0, 0};
    const int my_rounds = (g_lo + wave < g_hi) ? (g_hi - g_lo - wave + GW - 1) / GW : 0;   // groups this wave really owns
    auto issue_head = [&](bool counted) {
        producer_rows(P);
        const int nch0 = phase_nch(0);
        if (counted && FP8) {   // (lanes past a span's end re-read the row's last granule and meet the zero chunk of x)
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int at = min(u * 64 + lane, (nch0 >> 1) - 1);
#pragma unroll
                for (int r = 0; r < NR; ++r) wr[r][u] = ld_nt(w8p[r] + at);
            }
        } else if (counted) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ch = min(u * 64 + lane, nch0 - 1);
                wr[0][u] = ld_nt(w0p + ch);
                wr[1][u] = ld_nt(w1p + ch);
            }
        } else {   // nothing is waited for by count behind these: the predicated form (it keeps hipcc's loop shape at B = 2)
#pragma unroll
            for (int u = 0; u < U; ++u) issue_step(P, u, my_rounds > 0);
        }
        advance(P);
    };
    constexpr bool coh = false;   // activations move with plain accesses (stream ordering)
    const bool one_pass = NORM && !XATTN && !multi_phase && (K >> 3) <= NT;
    // x first pays for the qkv projection only (-0.9 us); gate/up and lm-head lose 1.5 us with it, the plain rows of the down
    // projection gain nothing.  The chained launch waits for its producer first, so there the stream always goes out ahead.
    // The o-proj prologue (XATTN: merge of the attention split partials) also goes first: with every load of a chunk's merge
    // requested up front (branch-free, split count as a template argument -- only affordable while the weight block is not
    // occupying 64 registers) it is one round trip of ~1 us; as a loop BEHIND the weight stream it was sixteen dependent
    // round trips, ~5 of the 11 us of the launch.
    // plain rows (down projection) that fit one round of four chunks per thread: activations first, see stage_x
    // (round 3: off -- the four-chunk burst next to the 64-register weight head spilled 44-84 registers at the 128-VGPR cap of
    // two blocks per CU, i.e. the wave stored its own weight head to scratch right after requesting it; the down projection this
    // was built for runs on decode_ks.hip now, this kernel only keeps shapes with K % 64 != 0 and the fp8 rows)
    const bool plain_first = false;
    // fp8: every one-pass prologue goes first -- the first burst is most of the matrix, the refills cannot go out before x is
    // staged, and x requested behind the burst arrived 7 us into a 19 us gate/up launch (tools/gemv_lab.hip)
    const bool head_first = !((one_pass && (MODE == MODE_QKV || FP8)) || XATTN || plain_first);
    if (head_first) issue_head(false);

    // chunk c (8 elements) of activation row b of a NORM mode: from the fp32 residual stream when the step keeps one (GemvParams::h32;
    // rounded to bf16 here, once per read -- decode_ks.hip, the product path at these batches, keeps statistics and x g in fp32)
    auto ld_x8 = [&](int b, int c) -> u32x4_t {
        if (p.h32) return f32x8_to_bf16(ld_f32x8(p.h32 + (size_t)b * p.ldh + (size_t)c * 8));
        return ld_act16((const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx) + c, coh);
    };
    // ---- RMSNorm statistics ----
    float rstd[B];
    // single-pass prologue: when the whole row fits one 16-byte chunk per thread, x and the norm weight are read ONCE
    // (both loads issued together), the statistics come from registers and the normalised row goes straight to LDS --
    // one L2 round trip instead of two on the critical path of every qkv / gate-up / lm-head launch
    if (one_pass) {
        __shared__ float red1p[GW][B];
        const bool mine = tid < (K >> 3);
        const int ct = min(tid, (K >> 3) - 1);
        u32x4_t xv[B];
        const u32x4_t wv = *((const u32x4_t*)p.norm_w + ct);
#pragma unroll
        for (int b = 0; b < B; ++b) xv[b] = ld_x8(b, ct);
        if (!head_first) issue_head(true);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[b][j] = mine ? xv[b][j] : 0u;
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_lo(xv[b][j]), bb = bf_hi(xv[b][j]);
                ss += a * a + bb * bb;
            }
            ss = wave_sum(ss);
            if (lane == 0) red1p[wave][b] = ss;
        }
        GEMV_STAMP(5);
        __syncthreads();
        GEMV_STAMP(6);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < GW; ++w) t += red1p[w][b];
            rstd[b] = rsqrtf(t / (float)K + p.eps);
            if (mine) {
                u32x4_t v = xv[b];
#pragma unroll
                for (int j = 0; j < 4; ++j) {   // one v_cvt_pk_bf16_f32 per rounding of a pair
                    const uint32_t r = pack_bf16x2(bf_lo(v[j]) * rstd[b], bf_hi(v[j]) * rstd[b]);
                    v[j] = pack_bf16x2(bf_lo(r) * bf_lo(wv[j]), bf_hi(r) * bf_hi(wv[j]));
                }
                xs[b * XS + tid] = v;
            }
            if (tid == NT - 1) xs[b * XS + (K >> 3)] = (u32x4_t){0u, 0u, 0u, 0u};
        }
    } else if (NORM) {
        __shared__ float red[GW][B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            float ss = 0.f;
            for (int c = tid; c < (K >> 3); c += NT) {
                const u32x4_t v = ld_x8(b, c);
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
        for (int b = 0; b < B; ++b) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < GW; ++w) t += red[w][b];
            rstd[b] = rsqrtf(t / (float)K + p.eps);
        }
    }

    // stage x[:, kc0 : kc0 + 8*nch] into LDS (normalised if NORM, merged from the attention partials if XATTN)
    auto stage_x = [&](int kc0, int nch, auto first_tag) {
        constexpr bool FIRST = decltype(first_tag)::value;   // the call in front of the main loop (no weight block live yet)
        if (!XATTN && !NORM && FIRST && plain_first) {
            // plain rows in front of the main loop (down projection: 22 KB per row): every chunk of a thread requested at once,
            // UNCONDITIONALLY (clamped index, masked at the store), and the weight head right behind them -- the wait for the
            // activations is then a counted one.  (Behind the head, with predicated loads, hipcc waited for vmcnt(0): phase
            // stamps showed the prologue of a down launch ending 6 us in, when each wave's whole 16 KiB head had landed.)
            u32x4_t v[B][4];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const u32x4_t* xr = (const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx + kc0);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[b][j] = ld_act16(xr + min(tid + j * NT, nch - 1), coh);
            }
            issue_head(true);
#pragma unroll
            for (int b = 0; b < B; ++b) {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (tid + j * NT < nch) xs[b * XS + tid + j * NT] = v[b][j];
                if (tid == NT - 1) xs[b * XS + nch] = (u32x4_t){0u, 0u, 0u, 0u};
            }
            return;
        }
        if (!XATTN && !NORM) {
            // plain rows (down projection: 22 KB per row): four loads in flight per thread, then the LDS writes
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const u32x4_t* xr = (const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx + kc0);
                for (int c0 = tid; c0 < nch; c0 += 4 * NT) {
                    u32x4_t v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = (c0 + j * NT < nch) ? ld_act16(xr + c0 + j * NT, coh) : (u32x4_t){0u, 0u, 0u, 0u};
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        if (c0 + j * NT < nch) xs[b * XS + c0 + j * NT] = v[j];
                }
                if (tid == NT - 1) xs[b * XS + nch] = (u32x4_t){0u, 0u, 0u, 0u};
            }
            return;
        }
        if (XATTN) {
            // chunk cg = head (cg>>4), elements (cg&15)*8..+8 of the split partials.  A thread's FIRST chunk is merged with
            // every load in flight at once (needs ~60 registers: before the weight block exists); the weight stream is
            // requested right behind it; any further chunk (K > 4096) takes the low-register loop.
            auto merge_at = [&](int c, int b, bool fast) {
                const int cg = (kc0 >> 3) + c;
                const float* pp = p.attn_part + (size_t)(b * p.Hq + (cg >> 4)) * p.nsplit * PSTRIDE;
                const int d0 = (cg & 15) * 8;
                // 8 splits is what decode_attn_nsplit gives at batch 1-2 for every head count up to 32
                return (fast && p.nsplit == 8) ? attn_merge_chunk<8, 4>(pp, d0) : attn_merge_chunk_loop(pp, d0, p.nsplit);
            };
            if (tid < nch) {
#pragma unroll
                for (int b = 0; b < B; ++b) xs[b * XS + tid] = merge_at(tid, b, FIRST);
            }
            if (FIRST) {
                __builtin_amdgcn_sched_barrier(0);
                issue_head(true);
                __builtin_amdgcn_sched_barrier(0);
            }
            for (int c = tid + NT; c < nch; c += NT)
#pragma unroll
                for (int b = 0; b < B; ++b) xs[b * XS + c] = merge_at(c, b, false);
            if (tid == NT - 1)
#pragma unroll
                for (int b = 0; b < B; ++b) xs[b * XS + nch] = (u32x4_t){0u, 0u, 0u, 0u};
            return;
        }
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const u32x4_t* xr = (const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx + kc0);
            for (int c = tid; c < nch; c += NT) {
                u32x4_t v;
                if (XATTN) {
                    // chunk cg = head (cg>>4), elements (cg&15)*8..+8 of the split partials
                    const int cg = (kc0 >> 3) + c;
                    const float* pp = p.attn_part + (size_t)(b * p.Hq + (cg >> 4)) * p.nsplit * PSTRIDE;
                    v = attn_merge_chunk_loop(pp, (cg & 15) * 8, p.nsplit, coh);
                } else {
                    v = NORM ? ld_x8(b, (kc0 >> 3) + c) : ld_act16(xr + c, coh);
                }
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
                xs[b * XS + c] = v;
            }
            if (tid == NT - 1) xs[b * XS + nch] = (u32x4_t){0u, 0u, 0u, 0u};
        }
    };
    if (!one_pass) stage_x(0, phase_nch(0), std::true_type{});
    GEMV_STAMP(1);
    __syncthreads();
    GEMV_STAMP(2);

    // LMHEAD: running best over this wave's rows
    float best[B];
    int besti[B];
#pragma unroll
    for (int b = 0; b < B; ++b) {
        best[b] = -INFINITY;
        besti[b] = 0x7fffffff;
    }
    float acc[NR][B];
#pragma unroll
    for (int r = 0; r < NR; ++r)
#pragma unroll
        for (int b = 0; b < B; ++b) acc[r][b] = 0.f;

    float red[NR][B];
    // Epilogue operands are fetched when a group STARTS, not when its dot products are done: the old residual values (RESID)
    // and the row's position / page id / cos-sin pair (QKV) are dependent global loads (~0.7-1.5 us from L2) that would
    // otherwise sit at the tail of every group with the wave's weight ring idle behind them.  Lane b serves batch row b.
    const int eb = lane < B ? lane : 0;
    int pre_pos = 0, pre_pg = 0;
    float pre_a[NP], pre_b[NP];                          // RESID: h[r0], h[r1];  QKV: cos, sin
    float wsc[NR];                                       // FP8: the group's row scales
#pragma unroll
    for (int i = 0; i < NP; ++i) pre_a[i] = pre_b[i] = 0.f;
#pragma unroll
    for (int r = 0; r < NR; ++r) wsc[r] = 1.f;
    if (MODE == MODE_QKV) {
        pre_pos = p.ctx_len[eb];
        pre_pg = p.page_table[(size_t)eb * p.max_pages + pre_pos / p.page];
    }
    auto prefetch_epilogue = [&](int rd) {
        const int g = g_lo + rd * GW + wave;
        if (g >= g_hi) return;
        if constexpr (FP8) {   // every lane: the lm-head compares on all of them
#pragma unroll
            for (int i = 0; i < NP; ++i) {
                int r0, r1;
                pair_rows(min(g * NP + i, p.n_pairs - 1), r0, r1);
                wsc[2 * i] = p.wscale[r0];
                wsc[2 * i + 1] = p.wscale[r1];
            }
        }
        if (lane >= B) return;
#pragma unroll
        for (int i = 0; i < NP; ++i) {
            const int pg = FP8 ? min(g * NP + i, p.n_pairs - 1) : g;
            if (MODE == MODE_RESID) {
                int r0, r1;
                pair_rows(pg, r0, r1);
                if (p.h32) {
                    pre_a[i] = p.h32[(size_t)eb * p.ldh + r0];
                    pre_b[i] = p.h32[(size_t)eb * p.ldh + r1];
                } else {
                    const bf16_t* hp = (const bf16_t*)p.y + (size_t)eb * p.ldy;
                    pre_a[i] = bf2f(ld_act_bf16(hp + r0, coh));
                    pre_b[i] = bf2f(ld_act_bf16(hp + r1, coh));
                }
            } else if (MODE == MODE_QKV) {
                const int half = p.head_dim >> 1;
                const int hb = pg / half, d = pg - hb * half;
                if (hb < p.Hq + p.Hkv) {
                    pre_a[i] = p.cos_t[(size_t)pre_pos * half + d];
                    pre_b[i] = p.sin_t[(size_t)pre_pos * half + d];
                }
            }
        }
    };
    // every wave of the block walks the same number of rounds (block-uniform barriers in the multi-phase case)
    while (Cc.rd < rounds) {
        const bool valid = Cc.rd < my_rounds;
        if (Cc.blk == 0 && Cc.ph == 0) prefetch_epilogue(Cc.rd);
        if (multi_phase && Cc.blk == 0 && (Cc.rd != 0 || Cc.ph != 0)) {   // new phase: restage x (loads keep flying)
            __syncthreads();
            stage_x(Cc.ph * KC, phase_nch(Cc.ph), std::false_type{});
            __syncthreads();
        }
        const bool p_active = P.rd < my_rounds;
        if (P.blk == 0) producer_rows(P);
        const int nch = phase_nch(Cc.ph);
        if constexpr (FP8) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int sp = Cc.blk * U + u, base = sp * 128;
                if (valid && base < nch) {   // wave-uniform
                    const int nc2 = span_nc(sp) >> 1;
                    const bool l_ok = lane < nc2;
                    const int cc0 = l_ok ? base + lane : nch, cc1 = l_ok ? base + nc2 + lane : nch;   // past the end: the zero chunk
                    u32x4_t xa[B], xb[B];
#pragma unroll
                    for (int b = 0; b < B; ++b) {
                        xa[b] = xs[b * XS + cc0];
                        xb[b] = xs[b * XS + cc1];
                    }
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        // dwords 0-1 of the granule: the 8 weights of chunk cc0, dwords 2-3: of chunk cc1 (emmax_quant_rm8_kernel)
                        uint32_t w[8];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            w[2 * j] = fp8x2_to_bf16x2<false>(wr[r][u][j]);
                            w[2 * j + 1] = fp8x2_to_bf16x2<true>(wr[r][u][j]);
                        }
#pragma unroll
                        for (int b = 0; b < B; ++b) {
                            float a = acc[r][b];
                            a = dot2_bf16(w[0], xa[b][0], a);
                            a = dot2_bf16(w[1], xa[b][1], a);
                            a = dot2_bf16(w[2], xa[b][2], a);
                            a = dot2_bf16(w[3], xa[b][3], a);
                            a = dot2_bf16(w[4], xb[b][0], a);
                            a = dot2_bf16(w[5], xb[b][1], a);
                            a = dot2_bf16(w[6], xb[b][2], a);
                            a = dot2_bf16(w[7], xb[b][3], a);
                            acc[r][b] = a;
                        }
                    }
                }
            }
        } else {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = Cc.blk * 64 * U + u * 64 + lane;
            if (valid && Cc.blk * 64 * U + u * 64 < nch) {   // wave-uniform
                const int cc = min(c, nch);   // past the end: the zero chunk
#pragma unroll
                for (int b = 0; b < B; ++b) {
                    const u32x4_t xv = xs[b * XS + cc];
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
        // refill the whole block AFTER it is consumed: sixteen requests back to back (two rows x 8 KiB, consecutive addresses).
        // This is the order hipcc picked by itself at B = 1 and it is the fast one -- the per-step interleaving it chose at
        // B = 2 (refill of step u between the dot products of steps u and u+1, counted waits) is 2 us slower per gate/up launch
#pragma unroll
        for (int u = 0; u < U; ++u) issue_step(P, u, p_active);
#ifdef DECODE_LAB_TRACE
        if (Cc.rd == 0 && Cc.ph == 0 && Cc.blk == 0) GEMV_STAMP(3);
#endif
        const bool group_done = (Cc.ph == n_phase - 1) && (Cc.blk == phase_nblk(Cc.ph) - 1);
        const int g = g_lo + Cc.rd * GW + wave;
        advance(Cc);
        advance(P);
        if (!group_done || !valid) continue;

#pragma unroll
        for (int r = 0; r < NR; ++r)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const float t = wave_sum(acc[r][b]);
                acc[r][b] = 0.f;
                red[r][b] = FP8 ? t * wsc[r] : t;
            }

#pragma unroll
        for (int i = 0; i < NP; ++i) {
        const int pg = FP8 ? g * NP + i : g;   // the pair's index
        if (FP8 && pg >= p.n_pairs) continue;
        float red0[B], red1[B];
#pragma unroll
        for (int b = 0; b < B; ++b) {
            red0[b] = red[2 * i][b];
            red1[b] = red[2 * i + 1][b];
        }
        int r0, r1;
        pair_rows(pg, r0, r1);
        if (MODE == MODE_PLAIN) {
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (lane == b) {
                    st_act_bf16((bf16_t*)p.y + (size_t)b * p.ldy + r0, f2bf(red0[b]), coh);
                    if (2 * pg + 1 < p.n_rows) st_act_bf16((bf16_t*)p.y + (size_t)b * p.ldy + r1, f2bf(red1[b]), coh);
                }
        } else if (MODE == MODE_RESID) {
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (lane == b) {
                    if (p.h32) {   // fp32 master copy of the residual stream; the bf16 rows below mirror it
                        float* hq = p.h32 + (size_t)b * p.ldh;
                        hq[r0] = pre_a[i] + red0[b];
                        if (2 * pg + 1 < p.n_rows) hq[r1] = pre_b[i] + red1[b];
                    }
                    bf16_t* hp = (bf16_t*)p.y + (size_t)b * p.ldy;
                    st_act_bf16(hp + r0, f2bf(pre_a[i] + red0[b]), coh);
                    if (2 * pg + 1 < p.n_rows) st_act_bf16(hp + r1, f2bf(pre_b[i] + red1[b]), coh);
                }
        } else if (MODE == MODE_GATEUP) {
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (lane == b) st_act_bf16((bf16_t*)p.y + (size_t)b * p.ldy + pg, f2bf(silu(red0[b]) * red1[b]), coh);
        } else if (MODE == MODE_QKV) {
            const int hd = p.head_dim, half = hd >> 1;
            const int hb = pg / half, d = pg - hb * half;
#pragma unroll
            for (int b = 0; b < B; ++b)
                if (lane == b) {
                    const int pos = pre_pos;
                    // linear outputs are bf16 activations in the reference; RoPE acts on those
                    const float x0 = bf2f(f2bf(red0[b])), x1 = bf2f(f2bf(red1[b]));
                    if (hb < p.Hq + p.Hkv) {
                        const float cs = pre_a[i], sn = pre_b[i];
                        const bf16_t y0 = f2bf(x0 * cs - x1 * sn), y1 = f2bf(x1 * cs + x0 * sn);
                        if (hb < p.Hq) {
                            bf16_t* q = (bf16_t*)p.y + (size_t)b * p.ldy + hb * hd;
                            st_act_bf16(q + d, y0, coh);
                            st_act_bf16(q + d + half, y1, coh);
                        } else {
                            const int pgid = pre_pg;
                            bf16_t* kc = gemv_kv_row(p, false, b, pgid, pos, hb - p.Hq);
                            st_act_bf16(kc + d, y0, coh);
                            st_act_bf16(kc + d + half, y1, coh);
                        }
                    } else {
                        const int pgid = pre_pg;
                        bf16_t* vc = gemv_kv_row(p, true, b, pgid, pos, hb - p.Hq - p.Hkv);
                        st_act_bf16(vc + d, f2bf(x0), coh);
                        st_act_bf16(vc + d + half, f2bf(x1), coh);
                    }
                }
        } else if (MODE == MODE_LMHEAD) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
#pragma unroll
                for (int r = 0; r < 2; ++r) {
                    const int row = r == 0 ? r0 : r1;
                    if (r == 1 && 2 * pg + 1 >= p.n_rows) continue;
                    const float v = r == 0 ? red0[b] : red1[b];
                    if (v > best[b] || (v == best[b] && row < besti[b])) {
                        best[b] = v;
                        besti[b] = row;
                    }
                    if (p.logits_out && lane == 0) p.logits_out[(size_t)b * p.n_rows + row] = v;
                }
            }
        }
        }
    }

    GEMV_STAMP(4);
    if (MODE == MODE_LMHEAD) {
        // block best; first index wins ties (torch.argmax semantics); one partial per block
        __shared__ float bv[GW][B];
        __shared__ int bi[GW][B];
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                bv[wave][b] = best[b];
                bi[wave][b] = besti[b];
            }
        }
        __syncthreads();
        if (tid < B) {
            float v0 = bv[0][tid];
            int i0 = bi[0][tid];
#pragma unroll
            for (int w = 1; w < GW; ++w) {
                const float v = bv[w][tid];
                const int ii = bi[w][tid];
                if (v > v0 || (v == v0 && ii < i0)) {
                    v0 = v;
                    i0 = ii;
                }
            }
            st_act_f32(p.part_val + (size_t)blockIdx.x * B + tid, v0, coh);
            st_act_i32(p.part_idx + (size_t)blockIdx.x * B + tid, i0, coh);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// h[b] = E[cur_tok[b]]
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emmax_decode_embed_kernel(const int32_t* __restrict__ cur_tok, const bf16_t* __restrict__ E,
                                                                bf16_t* __restrict__ h, int hidden, int vocab, float* __restrict__ h32) {
    const int b = blockIdx.x;
    int id = cur_tok[b];
    id = min(max(id, 0), vocab - 1);
    const u32x4_t* s = (const u32x4_t*)(E + (size_t)id * hidden);
    u32x4_t* o = (u32x4_t*)(h + (size_t)b * hidden);
    constexpr bool coh = false;
    if (h32) {   // fp32 residual stream: the embedding row widened (exact), beside the bf16 row
        for (int c = threadIdx.x; c < hidden / 8; c += blockDim.x) {
            const f32x8_t f = bf16x8_to_f32(s[c]);
            float* hp = h32 + (size_t)b * hidden + (size_t)c * 8;
            *(f32x4_t*)hp = f.lo;
            *(f32x4_t*)(hp + 4) = f.hi;
        }
    }
    for (int c = threadIdx.x; c < hidden / 8; c += blockDim.x) st_act16(o + c, s[c], coh);
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-KV decode attention over the paged cache.  grid (NSPLIT, Hkv, B), 256 threads.
// A 16-lane group owns one key at a time (lane = 16-byte chunk of the 128-wide row; 4 keys per wave load instruction,
// fully coalesced) and keeps its own online-softmax state (m, l, o[8 per lane]); K and V of a whole chunk of keys are
// requested before the first score is computed, so 2*KU 16-byte loads per lane are in flight.  The 16 group states of
// the block are merged through shuffles + LDS, and the block writes one partial per (row, head, split):
//   part[((b*Hq + h)*nsplit + s) * PSTRIDE] = { o[0..HD) un-normalised, m, l, pad }
// The cross-split merge is fused into the staging prologue of the o-proj GEMV (XATTN).
// ---------------------------------------------------------------------------------------------------------------------
// DIRECT (one KV split per (row, head): batch >= 5 at 32 heads): nothing to merge -- the block holds the head's whole result,
// normalises it and writes the bf16 row the o-proj reads (the arithmetic of a one-split merge), no partials, no o-proj prologue.
// (Round 3 also measured a cross-split merge INSIDE this launch for 2-8 splits -- sc1 partials, arrival counter, last arriver
// merges: 12.6 against 5.7 us per launch at B = 1; removed from the product source in round 4)
// KV8 (round 5, opt-in fp8 KV cache): K / V pages hold e4m3 rows with one fp32 scale per (token, head) row -- 8 bytes per lane and key
// instead of 16, de-quantised in registers (v_cvt_scalef32_pk_bf16_fp8 with the row's scale) right before the dot products.  The key the
// qkv launch of THIS step produced (position L - 1) waits as bf16 in p.kv_stage: every block quantises it itself (so that this step sees
// the values every later step will read back) and split 0 appends bytes + scale to the cache.
// DEEP: four chunks of keys in flight per wave instead of two (always with KV8; bf16: tuning switch attn_deep)
template <int HD, int G, bool DIRECT = false, int NW = 4, bool KV8 = false, bool DEEP = KV8>
__global__ __launch_bounds__(NW * 64) void emmax_decode_attn_kernel(DecodeAttnParams p) {
    // waves per block: 4 (8-wave blocks were measured no faster at batch 1-2, where 512 four-wave blocks already put 8 waves on a CU,
    // DESIGN.md section 6); the one-split form of batch 5-8 is 256 blocks = ONE per CU: NW = 8 there (tuning switch attn_nw, round 5)
    constexpr int NT = NW * 64;
    static_assert(HD == 128, "decode attention maps 16 lanes x 8 elements onto one 128-wide K/V row");
    constexpr int KU = G <= 2 ? 4 : 2;    // keys per lane group per chunk (block chunk = 16 * KU keys), two chunks in flight
    constexpr int SP = 512;  // page ids kept in LDS = the longest page table the launcher accepts (32 K tokens at 64 per page)
    __shared__ int s_pages[SP];
    __shared__ float red_o[NW][G][HD];
    __shared__ float red_ml[NW][G][2];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = lane >> 4, ch = lane & 15;       // key group within the wave, 16-byte chunk within the row
    const int split = blockIdx.x, hk = blockIdx.y, b = blockIdx.z;
    const int nsplit = gridDim.x;
    constexpr bool coh = false;   // activations move with plain accesses (stream ordering)
    // ONE memory round trip for everything in front of the K/V loads: the context length and the done flag (scalar loads,
    // requested first), the row's whole page table (<= SP entries, two per thread, kept in registers until all requests are
    // out, then written to LDS) and q.  (A loop that waited for each table load, then a dependent scalar load for the length,
    // then another for the flag cost three extra round trips -- half of this 7 us kernel.)
    const int ctx_now = p.ctx_len[b];
    const int done_word = *(p.done ? p.done + b : p.ctx_len + b);   // always a load, never a branch with its own wait in front of
    const int row_done = p.done ? done_word : 0;                  // the q / page-table requests
    const int32_t* ptab = p.page_table + (size_t)b * p.max_pages;
    // q (already rotated, bf16) for the G heads of this kv head: lane holds elements ch*8 .. +8
    u32x4_t q[G];
#pragma unroll
    for (int gq = 0; gq < G; ++gq)
        q[gq] = ld_act16((const u32x4_t*)((const bf16_t*)p.q + (size_t)b * p.ldq + (hk * G + gq) * HD + ch * 8), coh);
    const int pt0 = ptab[min(tid, p.max_pages - 1)], pt1 = ptab[min(tid + NT, p.max_pages - 1)];
    __builtin_amdgcn_sched_barrier(0);   // every request above is out before the first wait (hipcc sinks the q load below the LDS write otherwise)
    s_pages[tid] = pt0;
    if (NT < SP) s_pages[tid + NT] = pt1;
    const int L = ctx_now + 1;                      // keys including the one appended by the qkv kernel of this step
    int kps = (L + nsplit - 1) >> __builtin_ctz(nsplit);   // the split count is a power of two (launcher)
    kps = (kps + 15) & ~15;
    const int k0 = split * kps;
    const int k1 = min(L, k0 + kps);
    const int Hq = p.Hkv * G;
    float* part = p.part + ((size_t)(b * Hq + hk * G) * nsplit + split) * PSTRIDE;

    if (k0 >= L || row_done) {   // empty split, or a row that no longer decodes: no K/V traffic
        if constexpr (DIRECT) {   // a row that no longer decodes: zeros (what the merge of an empty partial gives)
            for (int i = tid; i < G * (HD / 8); i += NT)
                *((u32x4_t*)((bf16_t*)p.o_out + (size_t)b * p.ldq + hk * G * HD) + i) = (u32x4_t){0u, 0u, 0u, 0u};
            return;
        }
        for (int i = tid; i < G * PSTRIDE; i += NT) {
            const int gq = i / PSTRIDE, j = i - gq * PSTRIDE;
            st_act_f32(part + (size_t)gq * nsplit * PSTRIDE + j, (j == HD) ? -INFINITY : 0.f, coh);
        }
        return;
    }

    const bf16_t* kc = (const bf16_t*)p.kcache;
    const bf16_t* vc = (const bf16_t*)p.vcache;
    const uint8_t* kc8 = (const uint8_t*)p.kcache;
    const uint8_t* vc8 = (const uint8_t*)p.vcache;
    // KV8: the step's new K / V row of this kv head, requested now (one round trip with the page table), quantised below
    u32x4_t new_k = {0u, 0u, 0u, 0u}, new_v = {0u, 0u, 0u, 0u};
    if constexpr (KV8) {
        const bf16_t* st = (const bf16_t*)p.kv_stage + ((size_t)b * p.Hkv + hk) * 2 * HD + ch * 8;
        new_k = *(const u32x4_t*)st;
        new_v = *(const u32x4_t*)(st + HD);
    }

    __syncthreads();
    if constexpr (KV8) {
        // one scale per row (e4m3_row_scale of the amax) over the 16 lanes of a key group (every group of every wave holds the same row)
        auto requant = [&](u32x4_t& v, uint8_t* cache, float* scales) {
            float am = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) am = fmaxf(am, fmaxf(fabsf(bf_lo(v[j])), fabsf(bf_hi(v[j]))));
            am = row16_max(am);
            const float sc = e4m3_row_scale(am);
            const u32x2_t q8 = quant8_e4m3(v, 1.0f / sc);
            v = dequant8_e4m3(q8, sc);
            if (split == 0 && tid < 16) {   // append: bytes + scale at position L - 1
                const int pos = L - 1, pg = s_pages[pos >> p.page_shift];
                const size_t rowi = (((size_t)pg * p.Hkv + hk) << p.page_shift) + (pos & (p.page - 1));
                *(u32x2_t*)(cache + rowi * HD + ch * 8) = q8;
                if (tid == 0) scales[rowi] = sc;
            }
        };
        requant(new_k, (uint8_t*)p.kcache, p.kscale);
        requant(new_v, (uint8_t*)p.vcache, p.vscale);
    }

    float m[G], l[G], o[G][8];
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        m[gq] = -INFINITY;
        l[gq] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[gq][j] = 0.f;
    }

    // software pipeline: two register chunk buffers; the loads of chunk i+1 are issued before the scores of chunk i are
    // computed, so every wave has K/V requests in flight at all times (a 1/8 split of a 1K context is two chunks: both
    // are requested up front)
    // KV8: kv / vv carry the raw bytes in [0..1] and the row's scale in [2] until consume_chunk de-quantises them
    auto load_chunk = [&](int kb, u32x4_t (&kv)[KU], u32x4_t (&vv)[KU], bool (&ok)[KU]) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int key = kb + u * (4 * NW) + wave * 4 + kg;
            ok[u] = key < k1;
            const int kk = ok[u] ? key : k0;
            if constexpr (KV8) {
                const int pg8 = s_pages[kk >> p.page_shift];
                const size_t rowi = (((size_t)pg8 * p.Hkv + hk) << p.page_shift) + (kk & (p.page - 1));
                const u32x2_t k8 = __builtin_nontemporal_load((const u32x2_t*)(kc8 + rowi * HD + ch * 8)),   // (non-temporal: see the bf16 path below)
                              v8 = __builtin_nontemporal_load((const u32x2_t*)(vc8 + rowi * HD + ch * 8));
                kv[u] = (u32x4_t){k8[0], k8[1], __float_as_uint(p.kscale[rowi]), (uint32_t)key};
                vv[u] = (u32x4_t){v8[0], v8[1], __float_as_uint(p.vscale[rowi]), 0u};
                continue;
            }
            // the page id ALWAYS comes from LDS (the launcher rejects tables longer than SP): a select between the LDS copy and
            // the global table became a FLAT load, whose wait (vmcnt(0) lgkmcnt(0)) also drained the K/V loads in flight --
            // every key's lookup waited for the previous key's rows
            const int pg = s_pages[kk >> p.page_shift];
            const size_t off = ((((size_t)pg * p.Hkv + hk) << p.page_shift) + (kk & (p.page - 1))) * HD + ch * 8;
            // NON-TEMPORAL loads (round 5): a K / V row is read by ONE block, once per step -- as plain loads the rows were allocated in the
            // XCD's L2 and in the MALL on their way through, displacing the lines every block of the NEXT launches re-reads (activation rows,
            // norm weights, RoPE tables).  Measured, builds alternating on one box (profiles/r05_attn_kv_nt_ab.txt): this launch 5.63 ->
            // 5.42 us at batch 1, 19.9 -> 18.3 at 8 rows, 69.0 -> 61.7 at 32; the whole step 2.593 -> 2.568 ms/token, 3.26 -> 3.19, 5.37 -> 5.11 ms
            kv[u] = __builtin_nontemporal_load((const u32x4_t*)(kc + off));
            vv[u] = __builtin_nontemporal_load((const u32x4_t*)(vc + off));
            if (coh && kk == L - 1) {   // the row appended by the qkv kernel of THIS step (one kernel back): agent-scope re-read
                kv[u] = ld_act16((const u32x4_t*)(kc + off), true);
                vv[u] = ld_act16((const u32x4_t*)(vc + off), true);
            }
        }
    };
    auto consume_chunk = [&](u32x4_t (&kv)[KU], u32x4_t (&vv)[KU], const bool (&ok)[KU]) {
        if constexpr (KV8) {   // bytes -> bf16 x scale; the key of this step comes from the staging row (its cache bytes may not have landed)
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                const bool is_new = (int)kv[u][3] == L - 1;
                const u32x4_t kd = dequant8_e4m3((u32x2_t){kv[u][0], kv[u][1]}, __uint_as_float(kv[u][2]));
                const u32x4_t vd = dequant8_e4m3((u32x2_t){vv[u][0], vv[u][1]}, __uint_as_float(vv[u][2]));
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    kv[u][j] = is_new ? new_k[j] : kd[j];
                    vv[u][j] = is_new ? new_v[j] : vd[j];
                }
            }
        }
#pragma unroll
        for (int gq = 0; gq < G; ++gq) {
            float sc[KU];
            float mc = -INFINITY;
#pragma unroll
            for (int u = 0; u < KU; ++u) {
                float s = 0.f;
                s = dot2_bf16(kv[u][0], q[gq][0], s);
                s = dot2_bf16(kv[u][1], q[gq][1], s);
                s = dot2_bf16(kv[u][2], q[gq][2], s);
                s = dot2_bf16(kv[u][3], q[gq][3], s);
                s = row16_sum(s);   // the 16 lanes of a key group are one DPP row
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
                for (int j = 0; j < 4; ++j) {
                    o[gq][2 * j] += pw * bf_lo(vv[u][j]);
                    o[gq][2 * j + 1] += pw * bf_hi(vv[u][j]);
                }
            }
            l[gq] = ls;
            m[gq] = mn;
        }
    };
    if constexpr (DEEP) {
        // FOUR chunks in flight: at 8 bytes per lane and key two chunks are 8 KiB per wave -- the launch was a latency chain (one chunk
        // per ~1.1 us round trip: 18.9 us for 53 MB at batch 8); the raw bytes of four chunks take the registers two bf16 chunks did
        constexpr int CHK = (4 * NW) * KU;
        u32x4_t kvA[KU], vvA[KU], kvB[KU], vvB[KU], kvC[KU], vvC[KU], kvD[KU], vvD[KU];
        bool okA[KU], okB[KU], okC[KU], okD[KU];
        load_chunk(k0, kvA, vvA, okA);
        if (k0 + CHK < k1) load_chunk(k0 + CHK, kvB, vvB, okB);
        if (k0 + 2 * CHK < k1) load_chunk(k0 + 2 * CHK, kvC, vvC, okC);
        for (int kb = k0; kb < k1; kb += 4 * CHK) {               // every condition is block-uniform
            if (kb + 3 * CHK < k1) load_chunk(kb + 3 * CHK, kvD, vvD, okD);
            consume_chunk(kvA, vvA, okA);
            if (kb + 4 * CHK < k1) load_chunk(kb + 4 * CHK, kvA, vvA, okA);
            if (kb + CHK < k1) consume_chunk(kvB, vvB, okB);
            if (kb + 5 * CHK < k1) load_chunk(kb + 5 * CHK, kvB, vvB, okB);
            if (kb + 2 * CHK < k1) consume_chunk(kvC, vvC, okC);
            if (kb + 6 * CHK < k1) load_chunk(kb + 6 * CHK, kvC, vvC, okC);
            if (kb + 3 * CHK < k1) consume_chunk(kvD, vvD, okD);
        }
    } else {
        u32x4_t kvA[KU], vvA[KU], kvB[KU], vvB[KU];
        bool okA[KU], okB[KU];
        load_chunk(k0, kvA, vvA, okA);
        for (int kb = k0; kb < k1; kb += 2 * (4 * NW) * KU) {
            const bool hasB = kb + (4 * NW) * KU < k1;          // block-uniform
            if (hasB) load_chunk(kb + (4 * NW) * KU, kvB, vvB, okB);
            consume_chunk(kvA, vvA, okA);
            if (kb + 2 * (4 * NW) * KU < k1) load_chunk(kb + 2 * (4 * NW) * KU, kvA, vvA, okA);
            if (hasB) consume_chunk(kvB, vvB, okB);
        }
    }

    // ---- merge the 4 key groups of the wave (lanes with equal ch), then the 4 waves through LDS ----
#pragma unroll
    for (int gq = 0; gq < G; ++gq) {
        const float mw = rows_max(m[gq]);
        const float msafe = (mw == -INFINITY) ? 0.f : mw;
        const float f = __expf(m[gq] - msafe);
        // every lane of a 16-lane key group carries the same l: after the two exchanges each lane holds the wave sum
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
    auto part_value = [&](int gq, int j) {
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
        return v;
    };
    if constexpr (DIRECT) {   // this block holds the head's whole result -- normalise and write the bf16 row directly
        for (int i = tid; i < G * (HD / 8); i += NT) {
            const int gq = i / (HD / 8), c = i - gq * (HD / 8);
            const float den = part_value(gq, HD + 1);
            const float inv = den > 0.f ? 1.0f / den : 0.f;   // the arithmetic of attn_merge_chunk<1> (weight exp(m - M) = 1)
            u32x4_t v;
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(part_value(gq, c * 8 + 2 * j) * inv, part_value(gq, c * 8 + 2 * j + 1) * inv);
            *((u32x4_t*)((bf16_t*)p.o_out + (size_t)b * p.ldq + (hk * G + gq) * HD) + c) = v;
        }
        return;
    }
    for (int i = tid; i < G * PSTRIDE; i += NT) {
        const int gq = i / PSTRIDE, j = i - gq * PSTRIDE;
        st_act_f32(part + (size_t)gq * nsplit * PSTRIDE + j, part_value(gq, j), coh);
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Finish a step: argmax over the lm-head partials, EOS / length bookkeeping, next current token.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void emmax_decode_finish_kernel(FinishParams p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    constexpr bool coh = false;
    __shared__ float sv[256];
    __shared__ int si[256];
    float best = -INFINITY;
    int besti = 0x7fffffff;
    for (int i = tid; i < p.n_part; i += 256) {
        const float v = ld_act_f32(p.part_val + (size_t)i * p.B + b, coh);
        const int ii = ld_act_i32(p.part_idx + (size_t)i * p.B + b, coh);
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
        const int budget_n = p.max_new_p[b];
        if (!p.is_prefill && !was_done && n >= budget_n) {   // token budget already spent before this step
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
            const bool budget = !p.is_prefill && n >= budget_n;
            // early exit: n_after tokens after the trigger id sequence (e.g. "POLICIES:" + the 8 action-line tokens)
            bool stop = false;
            const int n_trig = p.stop_cfg[0];
            if (n_trig > 0) {
                int aft = p.stop_after[b];
                if (aft >= 0) {
                    aft += 1;
                } else {
                    int mp = p.stop_m[b];
                    mp = (tok == p.stop_ids[mp]) ? mp + 1 : ((tok == p.stop_ids[0]) ? 1 : 0);   // single-restart matcher
                    if (mp == n_trig) {
                        aft = 0;
                        mp = 0;
                    }
                    p.stop_m[b] = mp;
                }
                p.stop_after[b] = aft;
                stop = aft >= 0 && aft >= p.stop_cfg[1];
            }
            if (tok == p.eos_id || budget || stop || p.ctx_len[b] + 1 >= p.max_ctx) p.done[b] = 1;
        }
        p.cur_tok[b] = tok;
    }
}

// caller-supplied continuation (teacher forcing / the HF cached forward step): the row decodes again whatever the engine's own
// greedy prediction was -- clear the done flag and the stop-rule state, lift the token budget
__global__ void emmax_set_tokens_kernel(int32_t* cur_tok, const int32_t* toks, int B, int32_t* done, int32_t* stop_m, int32_t* stop_after,
                                        int32_t* max_new, int budget) {
    const int i = threadIdx.x;
    if (i < B) {
        cur_tok[i] = toks[i];
        done[i] = 0;
        stop_m[i] = 0;
        stop_after[i] = -1;
        max_new[i] = budget;
    }
}

}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------------------------------
// persistent grid: 2 blocks of 8 waves per CU when the staged activations allow it, never more blocks than work
static int gemv_grid(int B, size_t smem, int n_groups, int max_grid = 0) {
    int grid = (B > 2 || smem > 72 * 1024) ? 256 : 512;
    if (max_grid > 0) grid = min(grid, max_grid);
    // a partial second layer of blocks is worse than none: the CUs that hold two blocks finish ~4 us after the others (the fp8
    // qkv rows are 384 blocks' worth of four-row groups: 16.6 us with 384 blocks, 15.0 with 256; tools/gemv_lab.hip)
    if (grid == 512 && cdiv(n_groups, GW) < 512) grid = 256;
    return min(grid, cdiv(n_groups, GW));
}
// block shape of the fp8 GEMV (template argument F8): four-row groups when that still gives every wave of the 512-block grid one,
// else two rows x 12 steps when the row has more than eight spans, else two rows x 4 steps
static int gemv_fp8_shape(int n_rows, int K) {
    if (n_rows >= 4 * 512 * GW * 3 / 4) return 3;
    return K > 8 * 1024 ? 2 : 1;
}

template <int B, int MODE, bool NORM, bool XATTN = false, int F8 = 0>
static int launch_gemv_t(const GemvParams& p, hipStream_t stream, int* grid_out) {
    const size_t smem = (size_t)B * (p.kc * 2 + 16);
    int grid = gemv_grid(B, smem, p.n_groups, (F8 == 1 || F8 == 2) ? 256 : p.max_grid);
    if (MODE == MODE_LMHEAD) grid = min(grid, p.max_parts);
    if (grid_out) *grid_out = grid;
    auto kern = emmax_decode_gemv_kernel<B, MODE, NORM, XATTN, F8>;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(GW * 64), smem, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <int MODE, bool NORM, bool XATTN = false>
static int launch_gemv_mode(GemvParams p, int B, hipStream_t stream, int* grid_out) {
    // K phase: keep B * kc * 2 bytes of activations under ~128 KiB of LDS
    const int cap = (128 * 1024 / 2 / B) & ~511;
    p.kc = p.K <= cap ? p.K : (cdiv(cdiv(p.K, cdiv(p.K, cap)), 512) * 512);
    if (NORM && p.kc != p.K) return -1;
    if (MODE == MODE_QKV) p.n_groups = p.n_rows / 2;
    else if (MODE == MODE_GATEUP) p.n_groups = p.n_rows / 2;
    else p.n_groups = (p.n_rows + 1) / 2;
    p.n_pairs = p.n_groups;
    if (p.wscale) {   // fp8 rows (batch 1-2, one K phase)
        if (p.kc != p.K || p.K % 16 || B > 2) return -1;
        const int f8 = gemv_fp8_shape(p.n_rows, p.K);
        if (f8 == 3) p.n_groups = (p.n_pairs + 1) / 2;   // groups of two pairs
#define CASEF(BB, FF) if (B == BB && f8 == FF) return launch_gemv_t<BB, MODE, NORM, XATTN, FF>(p, stream, grid_out)
        CASEF(1, 1); CASEF(1, 2); CASEF(1, 3); CASEF(2, 1); CASEF(2, 2); CASEF(2, 3);
#undef CASEF
        return -1;
    }
    switch (B) {
#define CASEB(BB) case BB: return launch_gemv_t<BB, MODE, NORM, XATTN>(p, stream, grid_out)
        CASEB(1); CASEB(2); CASEB(3); CASEB(4); CASEB(5); CASEB(6); CASEB(7); CASEB(8);
#undef CASEB
        default: return -1;
    }
}

template <int MODE, bool NORM, bool XATTN = false>
static int gemv_init_mode() {
    const int lim = 160 * 1024 - 4096;
    hipError_t e = hipSuccess;
#define SETB(BB) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_gemv_kernel<BB, MODE, NORM, XATTN>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    SETB(1); SETB(2); SETB(3); SETB(4); SETB(5); SETB(6); SETB(7); SETB(8);
#undef SETB
#define SETF(BB, FF) if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_gemv_kernel<BB, MODE, NORM, XATTN, FF>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    SETF(1, 1); SETF(1, 2); SETF(1, 3); SETF(2, 1); SETF(2, 2); SETF(2, 3);
#undef SETF
    return e == hipSuccess ? 0 : -4;
}
int decode_gemv_init() {
    static int done = -1;
    if (done == 0) return 0;
    int r = gemv_init_mode<MODE_QKV, true>();
    if (!r) r = gemv_init_mode<MODE_RESID, false>();
    if (!r) r = gemv_init_mode<MODE_RESID, false, true>();
    if (!r) r = gemv_init_mode<MODE_GATEUP, true>();
    if (!r) r = gemv_init_mode<MODE_LMHEAD, true>();
    if (!r) r = gemv_init_mode<MODE_PLAIN, false>();
    done = r;
    return r;
}

// the fp8 row GEMV stages the whole activation rows in LDS (one K phase) and serves batch 1-2
bool decode_gemv_fp8_fits(int B, int K) {
    if (B < 1 || B > 2 || K <= 0 || K % 16) return false;
    return K <= ((128 * 1024 / 2 / B) & ~511);
}

int launch_quant_rm8(const void* src, int ld, void* dst, float* scales, int N, int K, hipStream_t stream) {
    if (N <= 0 || K <= 0 || K % 16 || ld % 8) return -1;
    hipLaunchKernelGGL(emmax_quant_rm8_kernel, dim3(N), dim3(256), 0, stream, (const bf16_t*)src, ld, (uint8_t*)dst, scales, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// p.wscale set: p.W is the fp8 row copy of launch_quant_rm8 (ldw = bytes per row), batch 1-2
int launch_decode_gemv(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    if (p.K % 8 || p.ldw % 8 || p.ldx % 8) return -1;
    if (decode_ks_enabled()) {   // batch 1-2, bf16, K % 64 == 0: the K-split kernel (decode_ks.hip)
        const int r = launch_decode_ks(mode, p, B, stream, grid_out);
        if (r != -2) return r;
    }
    switch (mode) {
        case MODE_QKV: return launch_gemv_mode<MODE_QKV, true>(p, B, stream, grid_out);
        case MODE_RESID:
            return p.attn_part ? launch_gemv_mode<MODE_RESID, false, true>(p, B, stream, grid_out)
                               : launch_gemv_mode<MODE_RESID, false>(p, B, stream, grid_out);
        case MODE_GATEUP: return launch_gemv_mode<MODE_GATEUP, true>(p, B, stream, grid_out);
        case MODE_LMHEAD: return launch_gemv_mode<MODE_LMHEAD, true>(p, B, stream, grid_out);
        case MODE_PLAIN: return launch_gemv_mode<MODE_PLAIN, false>(p, B, stream, grid_out);
        default: return -1;
    }
}

int launch_decode_embed(const int32_t* cur_tok, const void* E, void* h, int B, int hidden, int vocab, hipStream_t stream, float* h32) {
    hipLaunchKernelGGL(emmax_decode_embed_kernel, dim3(B), dim3(256), 0, stream, cur_tok, (const bf16_t*)E, (bf16_t*)h, hidden, vocab, h32);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// splits of the KV range per (row, kv head): ~512 blocks in flight, at most 8 partials to merge
int decode_attn_nsplit(int B, int Hkv) {
    const int forced = emmax_tune().attn_nsplit;
    if (forced > 0) {   // rounded down to a power of two (the kernel divides by shifting)
        int f = forced > 16 ? 16 : forced;
        while (f & (f - 1)) f &= f - 1;
        return f;
    }
    // batch 1-2: ~512 blocks of 4 waves (8 splits at 32 heads).  Batch >= 3: one block per CU is enough and every split less
    // halves the partials the o-proj has to merge for 8 rows -- at B = 8 (32 heads) ONE split: attention 20.5 -> 20.0 us, o-proj
    // 12.6 -> 11.0 us, step 3.373 -> 3.297 ms (4 splits: 22.3 / 15.5 us)
    int ns = (B >= 3 ? 256 : 512) / (B * Hkv);
    if (ns < 1) ns = 1;
    if (ns > 8) ns = 8;
    while (ns & (ns - 1)) ns &= ns - 1;   // power of two: the o-proj prologue merges with a branch-free unrolled loop
    return ns;
}

int launch_decode_attn(const DecodeAttnParams& p_in, int B, int Hq, int head_dim, int nsplit, hipStream_t stream) {
    if (head_dim != 128) return -1;
    DecodeAttnParams p = p_in;
    if (p.max_pages < 1 || p.max_pages > 512 || p.page < 1 || (p.page & (p.page - 1))) return -1;   // table fits the kernel's LDS copy; page = 2^k
    if (nsplit < 1 || (nsplit & (nsplit - 1))) return -1;   // the kernel divides the keys among the splits by shifting
    p.page_shift = 0;
    while ((1 << p.page_shift) < p.page) ++p.page_shift;
    if (p.o_out && nsplit != 1) return -1;   // the direct form exists for one split only (nothing to merge)
    if (p.kv_stage && (!p.kscale || !p.vscale)) return -1;   // fp8 KV cache: bytes + one scale per row
    const int G = Hq / p.Hkv;
    dim3 grid(nsplit, p.Hkv, B), block(256);
    const int nw = emmax_tune().attn_nw;
    const bool nw8 = nw == 8 || (nw == 0 && p.o_out && (long)p.Hkv * B <= 256);
    // four chunks of keys in flight (bf16 cache): measured on one box (profiles/r05_attn_deep_k32_spread_ab.txt) -- B = 32 (1024 blocks, four
    // per CU) 73.0 -> 68.5 us per launch, B = 8 / 16 (256 / 512 blocks) 20.0 -> 20.6 / 36.9 -> 37.3: on from 1024 blocks (-1 = that rule)
    const int deep_sw = emmax_tune().attn_deep;
    const bool deep = G <= 2 && (deep_sw > 0 || (deep_sw < 0 && (long)nsplit * p.Hkv * B >= 1024));   // (G >= 4: no registers for it)
    switch (G) {
#define ATTN_CASE(GG)                                                                                                   \
    case GG:                                                                                                           \
        if (p.kv_stage && p.o_out) hipLaunchKernelGGL((emmax_decode_attn_kernel<128, GG, true, 4, true>), grid, block, 0, stream, p); \
        else if (p.kv_stage) hipLaunchKernelGGL((emmax_decode_attn_kernel<128, GG, false, 4, true>), grid, block, 0, stream, p); \
        else if (p.o_out && deep) hipLaunchKernelGGL((emmax_decode_attn_kernel<128, GG, true, 4, false, true>), grid, block, 0, stream, p); \
        else if (deep) hipLaunchKernelGGL((emmax_decode_attn_kernel<128, GG, false, 4, false, true>), grid, block, 0, stream, p); \
        else if (p.o_out && nw8) hipLaunchKernelGGL((emmax_decode_attn_kernel<128, GG, true, 8>), grid, dim3(512), 0, stream, p); \
        else if (p.o_out) hipLaunchKernelGGL((emmax_decode_attn_kernel<128, GG, true>), grid, block, 0, stream, p);     \
        else hipLaunchKernelGGL((emmax_decode_attn_kernel<128, GG>), grid, block, 0, stream, p);                         \
        break
        ATTN_CASE(1); ATTN_CASE(2); ATTN_CASE(4); ATTN_CASE(8);
#undef ATTN_CASE
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_decode_finish(const FinishParams& p_in, hipStream_t stream) {
    FinishParams p = p_in;
    hipLaunchKernelGGL(emmax_decode_finish_kernel, dim3(p.B), dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_set_tokens(int32_t* cur_tok, const int32_t* toks, int B, int32_t* done, int32_t* stop_m, int32_t* stop_after, int32_t* max_new,
                      int budget, hipStream_t stream) {
    hipLaunchKernelGGL(emmax_set_tokens_kernel, dim3(1), dim3(64), 0, stream, cur_tok, toks, B, done, stop_m, stop_after, max_new, budget);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

#ifdef DECODE_LAB_TRACE
// lab builds only (tools/decode_stage_trace.py): the phase stamps of the most recent GEMV launch, [block][8]
extern "C" int emmax_debug_gemv_trace(unsigned long long* host_out, int n_words) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemv_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif

