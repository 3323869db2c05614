// decode_ks.hip -- batch 1-2 decode projections, K split across the waves of a block ("ks").
// Replaces the q_len == 1 linears of HF `LlamaDecoderLayer` + the fused stages around them (cached branch of
// prismatic/extern/hf/modeling_prismatic.py:325-341), like decode.hip's GEMV, with a structure that has NO block-wide
// activation stage and NO barrier in front of the weight stream:
//
//   * a block is 8 waves; wave w owns the K slice [w K/8, (w+1) K/8) of EVERY row of the block.  Its slice of the activation
//     row lives in registers for the whole launch (K = 4096: one 16-byte chunk per lane; the down projection's K = 11008:
//     three), loaded straight from global memory -- no LDS round trip, nobody waits for another wave's loads;
//   * weights go HBM -> VGPR with 16-byte non-temporal loads, a batch of RB rows (16 KiB per wave at K = 4096) in flight; a batch
//     is consumed completely (v_dot2c_f32_bf16 against the register-resident slice), its successor requested back to back,
//     and only then are the RB per-lane partial sums reduced -- with a TRANSPOSING butterfly (v_permlane32/16_swap, DPP
//     row_ror / half_mirror / quad_perm: ~2 VALU instructions per row instead of ~7 for one wave_sum per row) that leaves
//     row j's total in lane group j;
//   * the eight K-slice sums of a row meet in LDS once, at the end of the block (one barrier per launch), where the fused
//     epilogues run: RoPE + paged K/V append (qkv), + residual (o-proj, down), SiLU(gate) * up, greedy argmax partial;
//   * RMSNorm is folded in without a statistics pass: y_r = rstd * sum_k W[r,k] bf16(x_k g_k); every wave adds up the squares
//     of its own slice and the scalar 1/rms multiplies the reduced sums in the epilogue (one rounding fewer than HF's
//     bf16(bf16(x rstd) g): closer to the fp32 reference, and nothing waits for a row statistic before the stream starts);
//   * the o-proj's merge of the split-KV attention partials happens per lane, for the lane's own chunk only, in registers.
#include <cstdlib>

#include "common.h"
#include "kernels.h"

#ifndef KS_W_AUX
#define KS_W_AUX 2   // cache policy of the weight stream (lab: -DKS_W_AUX=n; 2 = nt, the product)
#endif
namespace {

constexpr int KS_WAVES = 8;
constexpr int KS_NT = KS_WAVES * 64;
constexpr int PSTRIDE = EMMAX_PSTRIDE;

__device__ __forceinline__ u32x4_t ks_ld_nt(const u32x4_t* p) { return __builtin_nontemporal_load(p); }

// ---- transposing reduction ------------------------------------------------------------------------------------------
// In: v[j], j < RB = this lane's partial sum of row j.  Out: the total of row ks_row_of_lane<RB>(lane) (every lane of the
// wave holds a valid total; lanes that differ only in bits the plain stages folded hold the same one).
__device__ __forceinline__ void ks_fold32(float& a, float& b) {   // a = {a.lo + a.hi, b.lo + b.hi}
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
}
__device__ __forceinline__ void ks_fold16(float& a, float& b) {   // a = {a0 + a1, b0 + b1, a2 + a3, b2 + b3} over 16-lane rows
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    a += b;
}
template <int RB>
__device__ __forceinline__ float ks_reduce_rows(float (&v)[RB], int lane) {
    static_assert(RB == 1 || RB == 2 || RB == 4 || RB == 8 || RB == 16, "batch of rows");
    // transposing stages, high to low: 32 (RB = 16), 16 (RB >= 8), 8 (RB >= 4), 4 (RB >= 2); the other stages fold plainly
    if constexpr (RB == 16) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ks_fold32(v[j], v[j + 8]);
    }
    if constexpr (RB >= 8) {
        constexpr int H = 4;
#pragma unroll
        for (int j = 0; j < H; ++j) ks_fold16(v[j], v[j + H]);
    }
    if constexpr (RB >= 4) {
        const bool hi = (lane & 8) != 0;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float keep = hi ? v[j + 2] : v[j], give = hi ? v[j] : v[j + 2];
            v[j] = keep + dpp_move<0x128>(give);   // row_ror:8 = lane ^ 8 inside a row of 16
        }
    }
    if constexpr (RB >= 2) {
        const bool hi = (lane & 4) != 0;
        const float keep = hi ? v[1] : v[0], give = hi ? v[0] : v[1];
        v[0] = keep + dpp_move<0x141>(give);       // row_half_mirror: lane ^ 7 inside 8 lanes (pairs bit-2-clear with bit-2-set lanes)
    }
    float t = v[0];
    t += dpp_move<0xB1>(t);    // quad_perm [1,0,3,2]
    t += dpp_move<0x4E>(t);    // quad_perm [2,3,0,1]
    if constexpr (RB < 2) t += dpp_move<0x141>(t);
    if constexpr (RB < 4) t += dpp_move<0x128>(t);
    if constexpr (RB < 8) { float a = t, b = t; ks_fold16(a, b); t = a; }   // {r0 + r1, r0 + r1, r2 + r3, r2 + r3}
    if constexpr (RB < 16) { float a = t, b = t; ks_fold32(a, b); t = a; }
    return t;
}
template <int RB>
__device__ __forceinline__ int ks_row_of_lane(int lane) {
    int r = 0, bit = RB >> 1;
    if constexpr (RB == 16) { r += (lane >> 5) * bit; bit >>= 1; }
    if constexpr (RB >= 8) { r += ((lane >> 4) & 1) * bit; bit >>= 1; }
    if constexpr (RB >= 4) { r += ((lane >> 3) & 1) * bit; bit >>= 1; }
    if constexpr (RB >= 2) { r += ((lane >> 2) & 1) * bit; }
    return r;
}
// the lane that stores row ks_row_of_lane(lane): first lane of its quad, plain-folded bits clear
template <int RB>
__device__ __forceinline__ bool ks_lane_stores(int lane) {
    int mask = 3;
    if constexpr (RB < 2) mask |= 4;
    if constexpr (RB < 4) mask |= 8;
    if constexpr (RB < 8) mask |= 16;
    if constexpr (RB < 16) mask |= 32;
    return (lane & mask) == 0;
}

// rows in flight per wave: 16 loads of 16 B per lane at batch 1 (64 VGPRs), 8 at batch 2 (two accumulators per row)
template <int B, int CPL> struct KsShape { static constexpr int RB = (B == 1 ? 16 : 8) / (CPL == 1 ? 1 : CPL == 2 ? 2 : 4); };

// weight registers of a wave: one flat array
template <int B> struct KsRegs { static constexpr int N = B == 1 ? 16 : 8; };

enum { XS_GLOBAL = 0, XS_ATTN = 1, XS_EMBED = 2 };   // XS_EMBED: x row b = the embedding of token x_tok[b] (layer 0's qkv, the embed launch folded in)
#ifdef DECODE_LAB_TRACE
// lab builds only (tools/ks_trace.py): s_memrealtime stamps (100 MHz) of waves 0 and 7 of every block:
// [block][wave 0 | 7][op = 0][0 entered, 1 activations in registers, 2 unused, 3 stream done, 4 barrier passed, 5 epilogue done]
__device__ unsigned long long g_ks_trace[512 * 2 * 4 * 6];
#define KS_STAMP(op, k, v) do { if ((threadIdx.x == 0 || threadIdx.x == 448) && blockIdx.x < 512 && (op) >= 0) \
    g_ks_trace[((blockIdx.x * 2 + (threadIdx.x ? 1 : 0)) * 4 + (op)) * 6 + (k)] = (v); } while (0)
#else
#define KS_STAMP(op, k, v) do { } while (0)
#endif
template <int MODE>
__device__ __forceinline__ void ks_pair_rows(const GemvParams& p, int g, int& r0, int& r1) {
    if (MODE == GEMV_QKV) {
        const int half = p.head_dim >> 1;
        const int hb = g >> p.ks_shift, d = g - hb * half;   // head_dim is a power of two (launcher)
        r0 = hb * p.head_dim + d;
        r1 = r0 + half;
    } else if (MODE == GEMV_GATEUP) {
        r0 = (g >> 4) * 32 + (g & 15);
        r1 = r0 + 16;
    } else {
        r0 = 2 * g;
        r1 = min(2 * g + 1, p.n_rows - 1);
    }
}
// block b's contiguous share of the row PAIRS
__device__ __forceinline__ void ks_share(const GemvParams& p, int& g_lo, int& npairs) {
    const int G = gridDim.x, bid = blockIdx.x;
    const int q = p.n_groups / G, r = p.n_groups % G;
    g_lo = bid * q + min(bid, r);
    npairs = min(q + (bid < r ? 1 : 0), max(p.n_groups - g_lo, 0));
}

// the matrix as a buffer: ONE descriptor, the row's byte offset in an SGPR (soffset), the lane's chunk in a 32-bit VGPR -- with
// flat pointers hipcc kept a 64-bit VGPR address per row in flight (32 registers, and spills at 16 rows).  Rows past the
// block's share get row index n_rows = the first byte past the matrix: the load returns zeros without touching memory, so a
// batch is always RB unconditional loads (counted waits, no branches).
template <int B, int MODE, int CPL>
__device__ __forceinline__ void ks_issue(const GemvParams& p, int g_lo, int nrows, int bb, const unsigned (&voff)[CPL], u32x4_t (&wr)[KsRegs<B>::N]) {
    constexpr int RB = KsShape<B, CPL>::RB;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)((unsigned)p.n_rows * (unsigned)p.ldw * 2u), 0x00020000);
#pragma unroll
    for (int jj = 0; jj < RB; ++jj) {
        const int i = bb * RB + jj;
        int r0, r1;
        ks_pair_rows<MODE>(p, g_lo + (i >> 1), r0, r1);
        // mask arithmetic, not a select: hipcc turned the select into a branch around the multiply, sixteen basic blocks per batch
        const int m = (i - nrows) >> 31;   // all ones: valid
        const int r = (((i & 1) ? r1 : r0) & m) | (p.n_rows & ~m);
        const unsigned so = (unsigned)r * (unsigned)p.ldw * 2u;
#pragma unroll
        for (int j = 0; j < CPL; ++j)
            wr[jj * CPL + j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff[j], so, KS_W_AUX));   // aux 2 = nt
    }
}

// this lane's chunks of the wave's K slice
template <int CPL>
__device__ __forceinline__ void ks_chunks(int K, int wave, int lane, int (&coff)[CPL], bool (&cok)[CPL], unsigned (&voff)[CPL]) {
    const int NCW = K >> 6;   // 16-byte chunks of a wave's K slice (K / 8 waves / 8 elements)
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = j * 64 + lane;
        cok[j] = c < NCW;
        coff[j] = wave * NCW + min(c, NCW - 1);   // lanes past the slice re-read its last chunk and meet a zero activation
        voff[j] = (unsigned)coff[j] * 16u;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// One projection.  The first batch of its weights is requested right behind the activation loads / the split merge (whose ~60
// registers the head would otherwise have to share).  part / sumsq: the launch's LDS scratch.
//   XS: where the activations come from (global row, attention split partials, embedding rows)
//   R32: the residual stream is the fp32 buffer p.h32 (GemvParams): NORM modes read it, RESID modes add into it
// (Round 3 chained four of these in one persistent launch with in-kernel mailbox hand-offs: bit-identical and 18 % slower --
// DESIGN.md section 6; that variant lives in the history, commit 61d5036, not in the product source.)
// ---------------------------------------------------------------------------------------------------------------------
//   EX: exact numerics (GemvParams::exact): the activation slice as two bf16 terms hi + lo of the fp32 value -- twice the dot2 issues per
//       weight register against the same accumulator (the stream stays HBM-bound: ~3.6 x VALU headroom per CU at batch 1), fp32 hand-offs
template <int B, int MODE, bool NORM, int XS, int CPL, bool R32, bool EX = false>
__device__ __forceinline__ void ks_run_op(const GemvParams& p, u32x4_t (&wr)[KsRegs<B>::N], float* part, float* sumsq, int rows_cap,
                                          int trace_op = -1) {
    constexpr int RB = KsShape<B, CPL>::RB;
    KS_STAMP(trace_op, 0, wall_clock64());
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int K = p.K, bid = blockIdx.x;
    int g_lo, npairs;
    ks_share(p, g_lo, npairs);
    const int nrows = 2 * npairs;

    // ---- epilogue operands of this thread's (pair, batch row), requested before anything else ----
    // wave b (< B) serves batch row b, lane i the block's pair i (launcher: npairs <= 64)
    const bool epi = wave < B && lane < npairs;
    const int eb = wave < B ? wave : 0, epair = g_lo + min(lane, max(npairs - 1, 0));
    int er0, er1;
    ks_pair_rows<MODE>(p, epair, er0, er1);
    float pre_a = 0.f, pre_b = 0.f;
    int pre_pos = 0, pre_pg = 0;
    if (epi) {
        if (MODE == GEMV_RESID) {
            if constexpr (R32) {
                const float* hp = p.h32 + (size_t)eb * p.ldh;
                pre_a = hp[er0];
                pre_b = hp[er1];
            } else {
                const bf16_t* hp = (const bf16_t*)p.y + (size_t)eb * p.ldy;
                pre_a = bf2f(hp[er0]);
                pre_b = bf2f(hp[er1]);
            }
        } else if (MODE == GEMV_QKV) {
            pre_pos = p.ctx_len[eb];
            pre_pg = p.page_table[(size_t)eb * p.max_pages + pre_pos / p.page];
            const int half = p.head_dim >> 1;
            const int hb = epair >> p.ks_shift, d = epair - hb * half;
            if (hb < p.Hq + p.Hkv) {
                pre_a = p.cos_t[(size_t)pre_pos * half + d];
                pre_b = p.sin_t[(size_t)pre_pos * half + d];
            }
        }
    }

    int coff[CPL];      // 16-byte chunk index inside a row
    bool cok[CPL];
    unsigned voff[CPL];
    ks_chunks<CPL>(K, wave, lane, coff, cok, voff);

    // ---- activations: the wave's slice, straight into registers ----
    u32x4_t xr[B][CPL];
    u32x4_t xl[EX ? B : 1][EX ? CPL : 1];   // EX: the lo terms
    if constexpr (EX && XS == XS_GLOBAL && !NORM) {
        // exact numerics, down projection: the SwiGLU product arrives as fp32 rows
        f32x8_t xq[B][CPL];
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int j = 0; j < CPL; ++j) xq[b][j] = ld_f32x8((const float*)p.x + (size_t)b * p.ldx + (size_t)coff[j] * 8);
        ks_issue<B, MODE, CPL>(p, g_lo, nrows, 0, voff, wr);
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int j = 0; j < CPL; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const hl2_t t = split_hl2(cok[j] ? f32x8_at(xq[b][j], 2 * e) : 0.f, cok[j] ? f32x8_at(xq[b][j], 2 * e + 1) : 0.f);
                    xr[b][j][e] = t.hi;
                    xl[b][j][e] = t.lo;
                }
    } else if constexpr (XS == XS_ATTN) {
        // chunk cg of the merged attention output = head (cg >> 4), elements (cg & 15) * 8 .. + 8 of the split partials; each
        // lane merges the chunks it will multiply with, every load of a chunk in flight at once
#pragma unroll
        for (int b = 0; b < B; ++b)
#pragma unroll
            for (int j = 0; j < CPL; ++j) {
                const int cg = coff[j];
                const float* pp = p.attn_part + (size_t)(b * p.Hq + (cg >> 4)) * p.nsplit * PSTRIDE;
                if constexpr (EX) {
                    float m8[8];
                    (void)(p.nsplit == 8 ? attn_merge_chunk<8, 4>(pp, (cg & 15) * 8, m8) : attn_merge_chunk_loop(pp, (cg & 15) * 8, p.nsplit, false, m8));
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const hl2_t t = split_hl2(cok[j] ? m8[2 * e] : 0.f, cok[j] ? m8[2 * e + 1] : 0.f);
                        xr[b][j][e] = t.hi;
                        xl[b][j][e] = t.lo;
                    }
                } else {
                    const u32x4_t v = p.nsplit == 8 ? attn_merge_chunk<8, 4>(pp, (cg & 15) * 8) : attn_merge_chunk_loop(pp, (cg & 15) * 8, p.nsplit);
#pragma unroll
                    for (int e = 0; e < 4; ++e) xr[b][j][e] = cok[j] ? v[e] : 0u;
                }
            }
        __builtin_amdgcn_sched_barrier(0);
        ks_issue<B, MODE, CPL>(p, g_lo, nrows, 0, voff, wr);
    } else {
        u32x4_t nw[CPL];
        if constexpr (NORM) {
#pragma unroll
            for (int j = 0; j < CPL; ++j) nw[j] = *((const u32x4_t*)p.norm_w + coff[j]);
        }
        float ss[B];
        if constexpr (R32 && NORM && XS != XS_EMBED) {
            // fp32 residual stream: the wave's slice as 8 floats per chunk; statistics and x g in fp32, ONE rounding to bf16
            f32x8_t xq[B][CPL];
#pragma unroll
            for (int b = 0; b < B; ++b)
#pragma unroll
                for (int j = 0; j < CPL; ++j) xq[b][j] = ld_f32x8(p.h32 + (size_t)b * p.ldh + (size_t)coff[j] * 8);
            ks_issue<B, MODE, CPL>(p, g_lo, nrows, 0, voff, wr);
#pragma unroll
            for (int b = 0; b < B; ++b) {
                ss[b] = 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float a = cok[j] ? f32x8_at(xq[b][j], 2 * e) : 0.f, c = cok[j] ? f32x8_at(xq[b][j], 2 * e + 1) : 0.f;
                        ss[b] += a * a + c * c;
                        if constexpr (EX) {
                            const hl2_t t = split_hl2(a * bf_lo(nw[j][e]), c * bf_hi(nw[j][e]));
                            xr[b][j][e] = t.hi;
                            xl[b][j][e] = t.lo;
                        } else {
                            xr[b][j][e] = pack_bf16x2(a * bf_lo(nw[j][e]), c * bf_hi(nw[j][e]));
                        }
                    }
            }
        } else {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                // layer 0 (XS_EMBED): the row is the embedding of the current token (the embed launch folded in); block 0 leaves a
                // copy in the residual stream for the o-proj's "+ residual"
                size_t row = (size_t)b;
                if constexpr (XS == XS_EMBED) row = (size_t)min(max(p.x_tok[b], 0), p.x_vocab - 1);
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    xr[b][j] = *((const u32x4_t*)((const bf16_t*)p.x + row * p.ldx) + coff[j]);
                    if constexpr (XS == XS_EMBED) {
                        if (blockIdx.x == 0 && cok[j]) {
                            *((u32x4_t*)((bf16_t*)p.x_copy + (size_t)b * p.ldx) + coff[j]) = xr[b][j];
                            if constexpr (R32) {   // ... and in the fp32 stream (exact)
                                const f32x8_t f = bf16x8_to_f32(xr[b][j]);
                                float* hp = p.h32 + (size_t)b * p.ldh + (size_t)coff[j] * 8;
                                *(f32x4_t*)hp = f.lo;
                                *(f32x4_t*)(hp + 4) = f.hi;
                            }
                        }
                    }
                }
            }
            // right behind the activations: they are waited for by count while the head of the stream is in flight
            ks_issue<B, MODE, CPL>(p, g_lo, nrows, 0, voff, wr);
#pragma unroll
            for (int b = 0; b < B; ++b) {
                ss[b] = 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        uint32_t v = cok[j] ? xr[b][j][e] : 0u;
                        if constexpr (NORM) {
                            const float a = bf_lo(v), c = bf_hi(v);
                            ss[b] += a * a + c * c;
                            if constexpr (EX) {   // (the embedding row of layer 0: exact bf16 values, x g in fp32, two terms)
                                const hl2_t t = split_hl2(a * bf_lo(nw[j][e]), c * bf_hi(nw[j][e]));
                                v = t.hi;
                                xl[b][j][e] = t.lo;
                            } else {
                                v = pack_bf16x2(a * bf_lo(nw[j][e]), c * bf_hi(nw[j][e]));
                            }
                        }
                        xr[b][j][e] = v;
                    }
            }
        }
        if constexpr (NORM) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const float t = wave_sum(ss[b]);
                if (lane == 0) sumsq[wave * B + b] = t;
            }
        }
    }

    KS_STAMP(trace_op, 1, wall_clock64());
    // ---- main loop: consume a batch of RB rows, request the next one, then reduce ----
    const int nbatch = (nrows + RB - 1) / RB;
    for (int bb = 0; bb < nbatch; ++bb) {
        float acc[B][RB];
#pragma unroll
        for (int jj = 0; jj < RB; ++jj)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                float a = 0.f;
#pragma unroll
                for (int j = 0; j < CPL; ++j) {
                    a = dot2_bf16(wr[jj * CPL + j][0], xr[b][j][0], a);
                    a = dot2_bf16(wr[jj * CPL + j][1], xr[b][j][1], a);
                    a = dot2_bf16(wr[jj * CPL + j][2], xr[b][j][2], a);
                    a = dot2_bf16(wr[jj * CPL + j][3], xr[b][j][3], a);
                    if constexpr (EX) {
                        a = dot2_bf16(wr[jj * CPL + j][0], xl[b][j][0], a);
                        a = dot2_bf16(wr[jj * CPL + j][1], xl[b][j][1], a);
                        a = dot2_bf16(wr[jj * CPL + j][2], xl[b][j][2], a);
                        a = dot2_bf16(wr[jj * CPL + j][3], xl[b][j][3], a);
                    }
                }
                acc[b][jj] = a;
            }
        // the refill reuses the registers the dot products just freed: pinned, or hipcc hoists it into fresh ones (and spills)
        __builtin_amdgcn_sched_barrier(0);
        // (past the last batch every row is out of range: RB free loads of zeros -- a branch here cost hipcc its register
        // assignment: the refill landed in fresh registers and the kernel spilled)
        ks_issue<B, MODE, CPL>(p, g_lo, nrows, bb + 1, voff, wr);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int b = 0; b < B; ++b) {
            const float t = ks_reduce_rows<RB>(acc[b], lane);
            if (ks_lane_stores<RB>(lane)) part[(wave * B + b) * rows_cap + bb * RB + ks_row_of_lane<RB>(lane)] = t;
        }
    }
    KS_STAMP(trace_op, 3, wall_clock64());
    __syncthreads();
    KS_STAMP(trace_op, 4, wall_clock64());

    // ---- epilogue: wave b, lane i = (batch row b, pair i) ----
    float red0 = 0.f, red1 = 0.f;
    if (epi) {
#pragma unroll
        for (int w = 0; w < KS_WAVES; ++w) {
            red0 += part[(w * B + eb) * rows_cap + 2 * lane];
            red1 += part[(w * B + eb) * rows_cap + 2 * lane + 1];
        }
        if constexpr (NORM) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < KS_WAVES; ++w) t += sumsq[w * B + eb];
            const float rstd = rsqrtf(t / (float)K + p.eps);
            red0 *= rstd;
            red1 *= rstd;
        }
    }
    const bool has1 = MODE == GEMV_QKV || MODE == GEMV_GATEUP || 2 * epair + 1 < p.n_rows;   // the pair's second row exists
    if (MODE == GEMV_PLAIN) {
        if (epi) {
            bf16_t* yp = (bf16_t*)p.y + (size_t)eb * p.ldy;
            yp[er0] = f2bf(red0);
            if (has1) yp[er1] = f2bf(red1);
        }
    } else if (MODE == GEMV_RESID) {
        if (epi) {
            if constexpr (R32) {   // the fp32 stream is the master; the bf16 rows below mirror it (batch >= 3 / fp8 kernels read those)
                float* hq = p.h32 + (size_t)eb * p.ldh;
                hq[er0] = pre_a + red0;
                if (has1) hq[er1] = pre_b + red1;
            }
            const uint32_t hv = pack_bf16x2(pre_a + red0, pre_b + red1);
            bf16_t* hp = (bf16_t*)p.y + (size_t)eb * p.ldy;
            hp[er0] = (bf16_t)(hv & 0xffffu);
            if (has1) hp[er1] = (bf16_t)(hv >> 16);
        }
    } else if (MODE == GEMV_GATEUP) {
        if constexpr (EX) {
            if (epi) ((float*)p.y)[(size_t)eb * p.ldy + epair] = silu_precise(red0) * red1;
        } else {
            const float a = silu(red0) * red1;
            if (epi) ((bf16_t*)p.y)[(size_t)eb * p.ldy + epair] = f2bf(a);
        }
    } else if (MODE == GEMV_QKV && EX) {
        if (epi) {   // exact numerics: nothing is rounded to bf16 -- fp32 RoPE (every product rounded on its own, as torch does), fp32 q, fp32 cache rows
            const int hd = p.head_dim, half = hd >> 1;
            const int hb = epair >> p.ks_shift, d = epair - hb * half;
            if (hb < p.Hq + p.Hkv) {
                const float y0 = __fadd_rn(__fmul_rn(red0, pre_a), -__fmul_rn(red1, pre_b)), y1 = __fadd_rn(__fmul_rn(red1, pre_a), __fmul_rn(red0, pre_b));
                if (hb < p.Hq) {
                    float* dst = (float*)p.y + (size_t)eb * p.ldy + hb * hd;
                    dst[d] = y0;
                    dst[d + half] = y1;
                } else {
                    gemv_kv_store_x(p, false, pre_pg, pre_pos, hb - p.Hq, d, y0);
                    gemv_kv_store_x(p, false, pre_pg, pre_pos, hb - p.Hq, d + half, y1);
                }
            } else {
                gemv_kv_store_x(p, true, pre_pg, pre_pos, hb - p.Hq - p.Hkv, d, red0);
                gemv_kv_store_x(p, true, pre_pg, pre_pos, hb - p.Hq - p.Hkv, d + half, red1);
            }
        }
    } else if (MODE == GEMV_QKV) {
        if (epi) {
            const int hd = p.head_dim, half = hd >> 1;
            const int hb = epair >> p.ks_shift, d = epair - hb * half;
            // linear outputs are bf16 activations in the reference; RoPE acts on those
            const float x0 = bf2f(f2bf(red0)), x1 = bf2f(f2bf(red1));
            if (hb < p.Hq + p.Hkv) {
                const bf16_t y0 = f2bf(x0 * pre_a - x1 * pre_b), y1 = f2bf(x1 * pre_a + x0 * pre_b);
                if (hb < p.Hq) {
                    bf16_t* q = (bf16_t*)p.y + (size_t)eb * p.ldy + hb * hd;
                    q[d] = y0;
                    q[d + half] = y1;
                } else {
                    bf16_t* kc = gemv_kv_row(p, false, eb, pre_pg, pre_pos, hb - p.Hq);
                    kc[d] = y0;
                    kc[d + half] = y1;
                }
            } else {
                bf16_t* vc = gemv_kv_row(p, true, eb, pre_pg, pre_pos, hb - p.Hq - p.Hkv);
                vc[d] = f2bf(x0);
                vc[d + half] = f2bf(x1);
            }
        }
    } else if (MODE == GEMV_LMHEAD) {
        if (wave < B) {   // whole waves: the argmax below is a wave reduction
            float bv = -INFINITY;
            int bi = 0x7fffffff;
            if (epi) {
                bv = red0;
                bi = er0;
                if (has1 && red1 > bv) { bv = red1; bi = er1; }   // er1 > er0: the first index wins ties (torch.argmax)
                if (p.logits_out) {
                    p.logits_out[(size_t)eb * p.n_rows + er0] = red0;
                    if (has1) p.logits_out[(size_t)eb * p.n_rows + er1] = red1;
                }
            }
            const float m = wave_max(bv);
            // vocabulary indices are exact in fp32: the smallest index among the maxima is a wave_max of its negation
            const float mi = wave_max((epi && bv == m) ? -(float)bi : -INFINITY);
            if (lane == 0) {
                p.part_val[(size_t)bid * B + eb] = m;
                p.part_idx[(size_t)bid * B + eb] = mi == -INFINITY ? 0x7fffffff : (int)(-mi);
            }
        }
    }
    KS_STAMP(trace_op, 5, wall_clock64());
}

// ---------------------------------------------------------------------------------------------------------------------
// Stand-alone launch of one projection.  grid: 2 blocks per CU; block b owns a contiguous range of row PAIRS (the epilogues
// work on pairs: (d, d + hd/2) of a head for QKV, (gate_i, up_i) for GATEUP, two consecutive rows otherwise).
// Dynamic LDS: float part[8 waves][B][rows_cap] + float sumsq[8][B].
// ---------------------------------------------------------------------------------------------------------------------
template <int B, int MODE, bool NORM, int XS, int CPL, bool R32, bool EX = false>
__global__ __launch_bounds__(KS_NT, 4) void emmax_decode_ks_kernel(GemvParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char ks_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rows_cap = p.kc;              // launcher: LDS rows per (wave, batch row), a multiple of RB
    float* part = (float*)ks_smem;          // [KS_WAVES][B][rows_cap]
    float* sumsq = part + KS_WAVES * B * rows_cap;   // [KS_WAVES][B]
    u32x4_t wr[KsRegs<B>::N];
    ks_run_op<B, MODE, NORM, XS, CPL, R32, EX>(p, wr, part, sumsq, rows_cap, 0);
}

template <int B, int MODE, bool NORM, int XS, int CPL, bool R32, bool EX = false>
int ks_launch_r(GemvParams p, hipStream_t stream, int* grid_out) {
    constexpr int RB = KsShape<B, CPL>::RB;
    int grid = min(512, p.n_groups);
    if (p.max_grid > 0) grid = min(grid, p.max_grid);
    if (MODE == GEMV_LMHEAD) grid = min(grid, p.max_parts);
    if (grid < 1) return -2;
    const int pairs_max = cdiv(p.n_groups, grid);
    if (pairs_max > 64) return -2;   // the epilogue maps one lane to a pair
    p.kc = cdiv(2 * pairs_max, RB) * RB;
    const size_t smem = (size_t)(KS_WAVES * B * p.kc + KS_WAVES * B) * sizeof(float);
    if (grid_out) *grid_out = grid;
    hipLaunchKernelGGL((emmax_decode_ks_kernel<B, MODE, NORM, XS, CPL, R32, EX>), dim3(grid), dim3(KS_NT), smem, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
// the fp32 residual stream concerns the modes that read (NORM) or update (RESID) the hidden rows; the embedding gather exists for qkv only
template <int B, int MODE, bool NORM, bool XATTN, int CPL>
int ks_launch_t(const GemvParams& p, hipStream_t stream, int* grid_out) {
    constexpr bool touches_h = NORM || MODE == GEMV_RESID;
    if (p.exact) {   // exact numerics: the projections of a decode step only, always on the fp32 stream
        if constexpr (MODE == GEMV_PLAIN) return -2;
        else {
            if (!p.h32) return -2;
            if constexpr (MODE == GEMV_QKV) {
                if (p.x_tok) return ks_launch_r<B, MODE, NORM, XS_EMBED, CPL, true, true>(p, stream, grid_out);
            }
            if (p.x_tok) return -2;
            return ks_launch_r<B, MODE, NORM, XATTN ? XS_ATTN : XS_GLOBAL, CPL, true, true>(p, stream, grid_out);
        }
    }
    if constexpr (MODE == GEMV_QKV) {
        if (p.x_tok) return p.h32 ? ks_launch_r<B, MODE, NORM, XS_EMBED, CPL, true>(p, stream, grid_out) : ks_launch_r<B, MODE, NORM, XS_EMBED, CPL, false>(p, stream, grid_out);
    } else if (p.x_tok) return -2;
    if constexpr (touches_h) {
        if (p.h32) return ks_launch_r<B, MODE, NORM, XATTN ? XS_ATTN : XS_GLOBAL, CPL, true>(p, stream, grid_out);
    } else if (p.h32) return -2;
    return ks_launch_r<B, MODE, NORM, XATTN ? XS_ATTN : XS_GLOBAL, CPL, false>(p, stream, grid_out);
}

// pairs / shift of a mode
template <int MODE>
int ks_prepare(GemvParams& p) {
    if (MODE == GEMV_QKV || MODE == GEMV_GATEUP) p.n_groups = p.n_rows / 2;
    else p.n_groups = (p.n_rows + 1) / 2;
    p.n_pairs = p.n_groups;
    if (MODE == GEMV_QKV) {
        p.ks_shift = 0;
        while ((2 << p.ks_shift) < p.head_dim) ++p.ks_shift;
        if ((2 << p.ks_shift) != p.head_dim) return -2;
    }
    return 0;
}

template <int MODE, bool NORM, bool XATTN>
int ks_launch_mode(GemvParams p, int B, hipStream_t stream, int* grid_out) {
    if (ks_prepare<MODE>(p)) return -2;
    const int cpl = cdiv(p.K >> 6, 64);
#define KS_CASE(BB, CC) if (B == BB && cpl == CC) return ks_launch_t<BB, MODE, NORM, XATTN, CC>(p, stream, grid_out)
    KS_CASE(1, 1); KS_CASE(1, 2); KS_CASE(1, 3); KS_CASE(2, 1); KS_CASE(2, 2); KS_CASE(2, 3);
#undef KS_CASE
    return -2;
}

}  // namespace

// tuning switch `ks` = 0 keeps the batch 1-2 bf16 projections on decode.hip's LDS-staged GEMV (the A/B partner)
bool decode_ks_enabled() { return emmax_tune().ks != 0; }

// -2: shape outside this kernel (the caller falls back to launch_decode_gemv's LDS-staged kernel); bf16 weights, batch 1-2,
// plain stream ordering, K a multiple of 64 and at most 12288 (three 16-byte chunks per lane)
int launch_decode_ks(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    if (B < 1 || B > 2 || p.wscale) return -2;
    if (p.K % 64 || p.K > 64 * 64 * 3 || p.ldw % 8 || p.ldx % 8 || p.K < 64) return -2;
    switch (mode) {
        case GEMV_QKV: return ks_launch_mode<GEMV_QKV, true, false>(p, B, stream, grid_out);
        case GEMV_RESID:
            if (p.attn_part && (p.K != p.Hq * 128)) return -2;   // the merge maps 16 chunks to a 128-wide head
            // The o-proj with the split merge in its prologue: every WAVE merges the chunks of its own four heads (24 loads in
            // flight per lane, no LDS stage, no barrier).  With 512 blocks that is twice the L2 reads of decode.hip's 256-block
            // LDS-staged merge and slower (12.2 against 10.1 us at B = 1, 7B); with ONE block per CU it is the faster one
            // (9.3 us: step 2.624 -> 2.594 ms/token).  Tuning switches ks_oproj = 0: decode.hip's kernel; ks_oproj_grid.
            if (p.attn_part) {
                if (!emmax_tune().ks_oproj) return -2;
                GemvParams q = p;
                q.max_grid = emmax_tune().ks_oproj_grid;
                return ks_launch_mode<GEMV_RESID, false, true>(q, B, stream, grid_out);
            }
            return ks_launch_mode<GEMV_RESID, false, false>(p, B, stream, grid_out);
        case GEMV_GATEUP: return ks_launch_mode<GEMV_GATEUP, true, false>(p, B, stream, grid_out);
        case GEMV_LMHEAD: return ks_launch_mode<GEMV_LMHEAD, true, false>(p, B, stream, grid_out);
        case GEMV_PLAIN: return ks_launch_mode<GEMV_PLAIN, false, false>(p, B, stream, grid_out);
        default: return -2;
    }
}

#ifdef DECODE_LAB_TRACE
extern "C" int emmax_debug_ks_trace(unsigned long long* host_out, int n_words) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_ks_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
