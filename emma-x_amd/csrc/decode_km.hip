// decode_km.hip -- batch 3-16 decode projections with K <= 4096 on MFMA, K split across the waves of a block ("km"; round 5: rows 9-16 --
// the batch is the 16-wide N side of the MFMA -- through the template argument NB; batches 17-32 are routed to decode_kmp.hip).
// Replaces the q_len == 1 linears of HF `LlamaDecoderLayer` at small batch (cached branch of
// prismatic/extern/hf/modeling_prismatic.py:325-341; the configs[2] per-GPU shard), like decode_mfma.hip, with the structure
// that made the batch-1 kernel of decode_ks.hip faster:
//
//   * one block of 8 waves per CU; wave w owns the k-steps [w KT/8, (w+1) KT/8) of EVERY 16-row tile of the block.  Its slice of
//     the activations lives in registers for the whole launch, already in MFMA B-operand form (16 fragments of 8 bf16 per lane:
//     batch row = lane & 15, rows past the batch are zero) -- no LDS stage, no barrier in front of the weight stream, no LDS
//     read per MFMA;
//   * weights stream from a fragment-major copy (one 1 KiB tile = the A operand of one v_mfma_f32_16x16x32_bf16) with
//     non-temporal buffer loads, the tile in an SGPR offset: TWO tiles (32 KiB) per wave in flight, i.e. 64 MB on the chip like
//     the batch-1 kernel (decode_mfma.hip keeps 16 MB in flight: at ~3 us of loaded latency that alone bounds it near 5.3 TB/s);
//     every load is unconditional (tiles / steps past the end carry an out-of-range offset and return zeros for free), so the
//     waits are counted;
//   * the eight K-slice partial tiles meet in LDS once, at the end of the block, where the fused epilogues run;
//   * the pairs the epilogues need live INSIDE a tile: the qkv and gate/up copies are row-permuted when they are built (rows
//     0-7 of a tile = elements d .. d+7 of a head / gate rows, rows 8-15 = d+64 .. d+71 / the matching up rows), so the work
//     unit is ONE tile for every matrix: 768 qkv tiles are 3 per block (as 384 two-tile tasks they were 1.5 and needed the
//     stream-K split of decode_mfma.hip);
//   * RMSNorm folded in as in decode_ks.hip: y = rstd * W (x .* g), the sum of squares per wave slice, 1/rms in the epilogue.
// The down projection (K = 11008: 43 k-steps per wave would need 172 activation registers) has its own two-phase kernel below
// (emmax_decode_kmd_kernel).
#include <cstdlib>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int KM_WAVES = 8;
constexpr int KM_NT = KM_WAVES * 64;
constexpr int KM_STEPS = 16;      // bf16 k-steps (32 elements) per tile and wave: K <= 8 * 16 * 32 = 4096
constexpr int KM_MAX_TILES = 8;   // tiles per block the LDS partial sums hold
constexpr int PSTRIDE = EMMAX_PSTRIDE;

// two fp8 e4m3 pairs (the low / high half of a dword) -> two bf16, exact
template <bool HI>
__device__ __forceinline__ uint32_t km_fp8x2(uint32_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, HI));
}
__device__ __forceinline__ bf16x8_t km_fp8x8(uint32_t lo, uint32_t hi) {
    const u32x4_t v = {km_fp8x2<false>(lo), km_fp8x2<true>(lo), km_fp8x2<false>(hi), km_fp8x2<true>(hi)};
    return __builtin_bit_cast(bf16x8_t, v);
}

// row-major [N, ld] -> fragment-major tiles in the km row order (N % 16 == 0, K % 32 == 0)
__global__ __launch_bounds__(256) void emmax_repack_km_kernel(const bf16_t* __restrict__ src, int ld, u32x4_t* __restrict__ dst, int N, int K,
                                                             int perm, int head_dim) {
    const size_t total = (size_t)N * K / 8;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (size_t)gridDim.x * 256) {
        const int lane = (int)(c & 63);
        const size_t tile = c >> 6;
        const int kt = (int)(tile % (K / 32)), nt = (int)(tile / (K / 32));
        const int n = km_src_row(perm, head_dim, nt, lane & 15), k = kt * 32 + (lane >> 4) * 8;
        dst[c] = *(const u32x4_t*)(src + (size_t)n * ld + k);
    }
}

// o-proj prologue: merge the NS split partials of head (cg >> 4) for every batch row into the staged rows [B][K] bf16, two rows per
// iteration so that the loads of both are in flight together (branch-free merge: attn_merge_chunk, common.h)
template <int NS>
__device__ __forceinline__ void km_merge_rows(const float* __restrict__ attn_part, unsigned char* dst, int pitch, int B, int Hq, int cg) {
    const float* pp0 = attn_part + (size_t)(cg >> 4) * NS * PSTRIDE;
    const size_t row = (size_t)Hq * NS * PSTRIDE;
    const int d0 = (cg & 15) * 8;
    int b = 0;
    for (; b + 1 < B; b += 2) {
        const u32x4_t v0 = attn_merge_chunk<NS>(pp0 + (size_t)b * row, d0);
        const u32x4_t v1 = attn_merge_chunk<NS>(pp0 + (size_t)(b + 1) * row, d0);
        *(u32x4_t*)(dst + (size_t)b * pitch) = v0;
        *(u32x4_t*)(dst + (size_t)(b + 1) * pitch) = v1;
    }
    if (b < B) *(u32x4_t*)(dst + (size_t)b * pitch) = attn_merge_chunk<NS>(pp0 + (size_t)b * row, d0);
}

// ---------------------------------------------------------------------------------------------------------------------
// grid: one block per CU; block b owns a contiguous range of tiles (launcher: at most KM_MAX_TILES).
// Dynamic LDS: float part[8 waves][tiles_cap][64][4] + float sumsq[8][16] + the activation staging (XATTN: the merged rows, bf16
// [B][K]; else one window of 8 row slices per wave).
// FP8: the tiles are e4m3 (16 rows x 64 k per KiB, decode_mfma.hip's emmax_quant_fm8_kernel layout) + one fp32 scale per row,
// de-quantised in registers (exact), two MFMAs per load.
// ---------------------------------------------------------------------------------------------------------------------
// (v_mfma_f32_16x16x32_fp8_fp8 on e4m3-quantised activations was measured in round 3: no faster -- the kernel is memory-bound by 14x --
// and 1.0e-1 of max|logit| from the oracle; removed from the product source)
// R32 (RESID modes): the residual stream's master copy is the fp32 buffer p.h32 (GemvParams) -- the epilogue adds into it and mirrors
// the sum into the bf16 rows p.y, which is what the NORM modes of this file read (fetching the fp32 rows instead cost the batch-8
// step 1.5 %: twice the prologue bytes in front of every qkv / gate-up / lm-head launch; the rounding that matters -- the one that
// used to accumulate over the 64 additions of a token -- is gone either way)
// NB: batch rows the prologue stages (8: batch <= 8; 16: batch 9-16 -- round 5: the batch is the 16-wide N side of the MFMA, columns 8-15
// were zeros until then).  LDS (not XATTN): ONE region per wave -- its activation window [NB][XPITCH] during the prologue, its partial
// tiles [tiles_cap][64][4] afterwards (the window is dead once the wave holds its fragments; nobody else touches the region before
// the block's only barrier) -- then sumsq[8][16]: 8 x 16.25 KiB at NB = 16, where window + partial tiles side by side would not fit.
// ROLL (tuning switch km_roll): a weight register is refilled with its step of the tile two ahead as soon as its MFMA has issued (32 KiB
// per wave in flight at all times) instead of all sixteen once the tile is done (16-32 KiB)
// EX (exact numerics, round 6, batch 3-8; NB = 16): the activations arrive as fp32 rows and enter the MFMA as two bf16 terms -- the hi terms of row b in
// batch column b, the lo terms in column b + 8: the eight columns a batch <= 8 leaves empty carry the second term, so the weight stream, the
// MFMA count and the register budget are those of the plain kernel; the epilogue adds columns b and b + 8 and hands fp32 results on (fp32 q rows,
// fp32 RoPE with torch's per-product rounding, the 24-bit / fp32 cache, the SwiGLU product in fp32) as decode_ks.hip's EX form does at batch 1-2.
template <int MODE, bool NORM, bool XATTN, bool FP8, bool R32, int NB, bool ROLL, bool EX = false>
__global__ __launch_bounds__(KM_NT, 2) void emmax_decode_km_kernel(GemvParams p) {
    static_assert(!EX || (NB == 16 && !FP8 && !ROLL), "exact numerics: sixteen window rows = eight batch rows x two terms, bf16 weights");
    extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, c16 = lane & 15;
    const int B = p.batch, K = p.K;
    constexpr int KS = FP8 ? 64 : 32;             // elements per load step
    constexpr int NSTEP = FP8 ? KM_STEPS / 2 : KM_STEPS;   // load steps per tile and wave
    const int KT = K / KS;                        // load steps of a whole row
    const int KTW = KT / KM_WAVES;                // ... of this wave's slice (launcher: K % (8 * KS) == 0, KTW <= NSTEP)
    const int tiles_cap = p.kc;                   // launcher
    constexpr int XPITCH = KM_STEPS * 64 + 16;    // bytes per staged row slice (+16: the four k-groups of a fragment read hit different banks)
    // bytes of a wave's region: XATTN: its partial tiles only (the merged rows [B][K] bf16 follow the regions and sumsq)
    const int wreg = XATTN ? tiles_cap * 1024 : max(NB * XPITCH, tiles_cap * 1024);
    auto part_of = [&](int w) { return (float*)(km_smem + (size_t)w * wreg); };   // [tiles_cap][64][4]
    float* sumsq = (float*)(km_smem + (size_t)KM_WAVES * wreg);      // [KM_WAVES][16]
    unsigned char* xlds = (unsigned char*)(sumsq + KM_WAVES * 16);   // XATTN: the merged rows [B][K] bf16 (EX: [16][K], rows b / b + 8 = the two terms)
    auto col_live = [&](int c) { return EX ? (c & 7) < B : c < B; };   // batch column c of the MFMA carries a row

    const int G = gridDim.x, bid = blockIdx.x;
    const int n_tiles = p.n_groups;
    const int tq = n_tiles / G, tr = n_tiles % G;
    const int t_lo = bid * tq + min(bid, tr), ntb = tq + (bid < tr ? 1 : 0);

    // ---- epilogue operands of this thread's (tile, lane) slot, requested before anything else ----
    // thread (tl = tid >> 6, l = tid & 63) finalises tile tl of the block: rows 4 (l >> 4) + j, batch column l & 15
    const int e_tl = tid >> 6, e_l = lane, e_c = e_l & 15, e_rq = e_l >> 4;
    const bool e_pairs = MODE == GEMV_QKV || MODE == GEMV_GATEUP;          // rows r, r + 8 of a tile belong together
    const bool e_on = e_tl < ntb && e_c < B && (!EX || e_c < 8) && (!e_pairs || e_rq < 2);
    const int e_tile = t_lo + min(e_tl, max(ntb - 1, 0));
    float pre_a[4] = {0.f, 0.f, 0.f, 0.f}, pre_b[4] = {0.f, 0.f, 0.f, 0.f};
    int pre_pos = 0, pre_pg = 0;
    if (e_on) {
        if (MODE == GEMV_RESID) {
            if constexpr (R32) {
                const f32x4_t hv = *(const f32x4_t*)(p.h32 + (size_t)e_c * p.ldh + e_tile * 16 + 4 * e_rq);
#pragma unroll
                for (int j = 0; j < 4; ++j) pre_a[j] = hv[j];
            } else {
                const bf16_t* hp = (const bf16_t*)p.y + (size_t)e_c * p.ldy + e_tile * 16 + 4 * e_rq;
#pragma unroll
                for (int j = 0; j < 4; ++j) pre_a[j] = bf2f(hp[j]);
            }
        } else if (MODE == GEMV_QKV) {
            pre_pos = p.ctx_len[e_c];
            pre_pg = p.page_table[(size_t)e_c * p.max_pages + pre_pos / p.page];
            const int hd = p.head_dim, half = hd >> 1, tph = hd / 16;
            const int hb = e_tile / tph, d0 = 8 * (e_tile - hb * tph) + 4 * e_rq;
            if (hb < p.Hq + p.Hkv) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pre_a[j] = p.cos_t[(size_t)pre_pos * half + d0 + j];
                    pre_b[j] = p.sin_t[(size_t)pre_pos * half + d0 + j];
                }
            }
        }
    }

    // ---- weight stream: buffer loads, the (tile, step) offset in an SGPR, the lane's 16 bytes in the VGPR offset ----
    const unsigned w_bytes = (unsigned)((size_t)p.n_groups * 16 * (size_t)K * (FP8 ? 1 : 2));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)w_bytes, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    u32x4_t wa[NSTEP], wb[NSTEP];   // two tiles in flight
    auto issue_one = [&](u32x4_t (&w)[NSTEP], int tl, int s) {   // step s of tile tl of the block (uniform); past the end: out of range = zeros, no traffic
        const int ok = (tl < ntb && s < KTW) ? -1 : 0;
        const unsigned so = ((unsigned)(((t_lo + tl) * KT + wave * KTW + s) * 1024) & (unsigned)ok) | (w_bytes & (unsigned)~ok);
        w[s] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, so, 2));   // aux 2 = nt
    };
    auto issue = [&](u32x4_t (&w)[NSTEP], int tl) {
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) issue_one(w, tl, s);
    };

    // ---- activations: this wave's K slice as MFMA B fragments (batch row = lane & 15, k = 8 (lane >> 4) .. + 8 of a 32-step) ----
    // Fetched ROW-shaped -- one contiguous 1 KiB request per batch row, lane l the 8 elements [8 l, 8 l + 8) of the slice -- and
    // turned into fragments through a wave-private LDS window (no barrier: only this wave writes and reads it).  Fetched
    // fragment-shaped (every instruction 16 bytes from each of B rows x 4 k-groups) the 32 requests of a wave touched 4x the cache
    // lines and held the CU's address path for ~3 us: qkv 28.8 us against 22.4 for decode_mfma.hip at B = 8.
    bf16x8_t xf[KM_STEPS];
    const int ksl = K / KM_WAVES / 32;   // 32-element fragments in the wave's slice
    if constexpr (XATTN) {
        // o-proj: the block merges the attention split partials once (every thread one 8-element chunk per row), through LDS
        issue(wa, 0);
        const int nch = K >> 3;
        for (int c = tid; c < nch; c += KM_NT) {
            unsigned char* dst = xlds + (size_t)c * 16;
            if constexpr (EX) {   // the merged chunk in fp32, split into its two terms: rows b and b + 8
                for (int b = 0; b < B; ++b) {
                    float m8[8];
                    (void)attn_merge_chunk_loop(p.attn_part + (size_t)(b * p.Hq + (c >> 4)) * p.nsplit * PSTRIDE, (c & 15) * 8, p.nsplit, false, m8);
                    u32x4_t hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const hl2_t t = split_hl2(m8[2 * e], m8[2 * e + 1]);
                        hi[e] = t.hi;
                        lo[e] = t.lo;
                    }
                    *(u32x4_t*)(dst + (size_t)b * K * 2) = hi;
                    *(u32x4_t*)(dst + (size_t)(b + 8) * K * 2) = lo;
                }
                continue;
            }
            switch (p.nsplit) {
                case 1: km_merge_rows<1>(p.attn_part, dst, K * 2, B, p.Hq, c); break;
                case 2: km_merge_rows<2>(p.attn_part, dst, K * 2, B, p.Hq, c); break;
                case 4: km_merge_rows<4>(p.attn_part, dst, K * 2, B, p.Hq, c); break;
                case 8: km_merge_rows<8>(p.attn_part, dst, K * 2, B, p.Hq, c); break;
                default:
                    for (int b = 0; b < B; ++b)
                        *(u32x4_t*)(dst + (size_t)b * K * 2) =
                            attn_merge_chunk_loop(p.attn_part + (size_t)(b * p.Hq + (c >> 4)) * p.nsplit * PSTRIDE, (c & 15) * 8, p.nsplit);
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < KM_STEPS; ++s) {
            const int k0 = (wave * ksl + s) * 32 + g4 * 8;
            const u32x4_t v = (s < ksl && col_live(c16)) ? *(const u32x4_t*)(xlds + ((size_t)c16 * K + k0) * 2) : (u32x4_t){0u, 0u, 0u, 0u};
            xf[s] = __builtin_bit_cast(bf16x8_t, v);
        }
    } else {
        unsigned char* xw = km_smem + (size_t)wave * wreg;                           // this wave's window: [NB rows][XPITCH]
        const bool mine = lane < ksl * 4;                                            // 16-byte chunks of the slice
        const int ch = min(lane, ksl * 4 - 1);
        if constexpr (EX) {
            // fp32 rows (NORM: the residual stream h32; else the fp32 operand p.x): lane l holds elements [8 l, 8 l + 8) of the wave's slice of every
            // batch row; statistics and x g in fp32, then the two terms into window rows b (hi) and b + 8 (lo)
            const float* src = NORM ? (const float*)p.h32 : (const float*)p.x;
            const size_t ld = NORM ? (size_t)p.ldh : (size_t)p.ldx;
            f32x4_t xq[8][2];
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float* rp = src + (size_t)min(b, B - 1) * ld + (size_t)wave * ksl * 32 + (size_t)ch * 8;
                xq[b][0] = *(const f32x4_t*)rp;
                xq[b][1] = *(const f32x4_t*)(rp + 4);
            }
            u32x4_t nwv = {0u, 0u, 0u, 0u};
            if constexpr (NORM) nwv = *((const u32x4_t*)((const bf16_t*)p.norm_w + wave * ksl * 32) + ch);
            issue(wa, 0);
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                float ss = 0.f;
                u32x4_t hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float a = (mine && b < B) ? xq[b][e >> 1][2 * (e & 1)] : 0.f, c = (mine && b < B) ? xq[b][e >> 1][2 * (e & 1) + 1] : 0.f;
                    if constexpr (NORM) {
                        ss += a * a + c * c;
                        a *= bf_lo(nwv[e]);
                        c *= bf_hi(nwv[e]);
                    }
                    const hl2_t t = split_hl2(a, c);
                    hi[e] = t.hi;
                    lo[e] = t.lo;
                }
                if constexpr (NORM) {
                    const float t = wave_sum(ss);
                    if (lane == 0) sumsq[wave * 16 + b] = t;
                }
                *(u32x4_t*)(xw + (size_t)b * XPITCH + (size_t)lane * 16) = hi;
                *(u32x4_t*)(xw + (size_t)(b + 8) * XPITCH + (size_t)lane * 16) = lo;
            }
        } else {
        u32x4_t xr[NB];
#pragma unroll
        for (int b = 0; b < NB; ++b) xr[b] = *((const u32x4_t*)((const bf16_t*)p.x + (size_t)min(b, B - 1) * p.ldx + wave * ksl * 32) + ch);
        u32x4_t nwv = {0u, 0u, 0u, 0u};
        if constexpr (NORM) nwv = *((const u32x4_t*)((const bf16_t*)p.norm_w + wave * ksl * 32) + ch);
        issue(wa, 0);   // behind the activation requests: those are waited for by count while the first tile is in flight
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float ss = 0.f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t v = (mine && b < B) ? xr[b][e] : 0u;
                if constexpr (NORM) {
                    const float a = bf_lo(v), c = bf_hi(v);
                    ss += a * a + c * c;
                    v = pack_bf16x2(a * bf_lo(nwv[e]), c * bf_hi(nwv[e]));
                }
                xr[b][e] = v;
            }
            if constexpr (NORM) {
                const float t = wave_sum(ss);
                if (lane == 0) sumsq[wave * 16 + b] = t;
            }
            *(u32x4_t*)(xw + (size_t)b * XPITCH + (size_t)lane * 16) = xr[b];   // lanes past the slice write zeros (never read)
        }
        }
        // wave-private: the reads below only need this wave's own LDS writes to have landed (hipcc's lgkmcnt), no barrier
#pragma unroll
        for (int s = 0; s < KM_STEPS; ++s) {
            const u32x4_t v = (s < ksl && col_live(c16)) ? *(const u32x4_t*)(xw + (size_t)c16 * XPITCH + (size_t)(s * 4 + g4) * 16) : (u32x4_t){0u, 0u, 0u, 0u};
            xf[s] = __builtin_bit_cast(bf16x8_t, v);
        }
    }
    issue(wb, 1);   // the second tile once the prologue's temporaries are dead (all of them next to two tiles would not fit 256 VGPRs)

    // ---- main loop: two tiles per trip (register sets a / b); a set is consumed MFMA by MFMA and refilled with the tile two ahead ----
    auto run_tile = [&](u32x4_t (&w)[NSTEP], int tl) {
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if constexpr (FP8) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(km_fp8x8(w[s][0], w[s][1]), xf[2 * s], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(km_fp8x8(w[s][2], w[s][3]), xf[2 * s + 1], acc, 0, 0, 0);
            } else {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[s]), xf[s], acc, 0, 0, 0);
            }
            if constexpr (ROLL) issue_one(w, tl + 2, s);
        }
        if constexpr (!ROLL) {
            __builtin_amdgcn_sched_barrier(0);
            issue(w, tl + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (tl < ntb) *(f32x4_t*)(part_of(wave) + ((size_t)tl * 64 + lane) * 4) = acc;
    };
    for (int tl = 0; tl < ntb; tl += 2) {
        run_tile(wa, tl);
        run_tile(wb, tl + 1);
    }
    __syncthreads();

    // ---- epilogue: thread (tl, l): rows 4 (l >> 4) + j of tile tl, batch column l & 15 ----
    float v[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};   // u: the partner rows r + 8 (qkv, gate/up)
    if (e_on) {
#pragma unroll
        for (int w = 0; w < KM_WAVES; ++w) {
            const f32x4_t a = *(const f32x4_t*)(part_of(w) + ((size_t)e_tl * 64 + e_l) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += a[j];
            if (e_pairs) {
                const f32x4_t b2 = *(const f32x4_t*)(part_of(w) + ((size_t)e_tl * 64 + e_l + 32) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) u[j] += b2[j];
            }
        }
        if constexpr (EX) {   // the lo terms' products: batch column e_c + 8 of the same rows
            float vl[4] = {0.f, 0.f, 0.f, 0.f}, ul[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < KM_WAVES; ++w) {
                const f32x4_t a = *(const f32x4_t*)(part_of(w) + ((size_t)e_tl * 64 + e_l + 8) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) vl[j] += a[j];
                if (e_pairs) {
                    const f32x4_t b2 = *(const f32x4_t*)(part_of(w) + ((size_t)e_tl * 64 + e_l + 40) * 4);
#pragma unroll
                    for (int j = 0; j < 4; ++j) ul[j] += b2[j];
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) { v[j] += vl[j]; u[j] += ul[j]; }
        }
        float sc = 1.f;
        if constexpr (NORM) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < KM_WAVES; ++w) t += sumsq[w * 16 + e_c];
            sc = rsqrtf(t / (float)K + p.eps);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float sv = sc, su = sc;
            if constexpr (FP8) {
                sv *= p.wscale[e_tile * 16 + 4 * e_rq + j];
                if (e_pairs) su *= p.wscale[e_tile * 16 + 8 + 4 * e_rq + j];
            }
            v[j] *= sv;
            u[j] *= su;
        }
    }
    float best = -INFINITY;
    int besti = 0x7fffffff;
    if (e_on) {
        const int row0 = e_tile * 16 + 4 * e_rq;   // natural-order matrices
        if (MODE == GEMV_PLAIN) {
            bf16_t* yp = (bf16_t*)p.y + (size_t)e_c * p.ldy + row0;
#pragma unroll
            for (int j = 0; j < 4; ++j) yp[j] = f2bf(v[j]);
        } else if (MODE == GEMV_RESID) {
            if constexpr (R32)
                *(f32x4_t*)(p.h32 + (size_t)e_c * p.ldh + row0) = (f32x4_t){pre_a[0] + v[0], pre_a[1] + v[1], pre_a[2] + v[2], pre_a[3] + v[3]};
            bf16_t* hp = (bf16_t*)p.y + (size_t)e_c * p.ldy + row0;
#pragma unroll
            for (int j = 0; j < 4; ++j) hp[j] = f2bf(pre_a[j] + v[j]);
        } else if (MODE == GEMV_GATEUP && EX) {
            float* yp = (float*)p.y + (size_t)e_c * p.ldy + 8 * e_tile + 4 * e_rq;
#pragma unroll
            for (int j = 0; j < 4; ++j) yp[j] = silu_precise(v[j]) * u[j];
        } else if (MODE == GEMV_GATEUP) {
            bf16_t* yp = (bf16_t*)p.y + (size_t)e_c * p.ldy + 8 * e_tile + 4 * e_rq;
#pragma unroll
            for (int j = 0; j < 4; ++j) yp[j] = f2bf(silu(v[j]) * u[j]);
        } else if (MODE == GEMV_QKV && EX) {
            // nothing is rounded to bf16: fp32 RoPE (every product rounded on its own, as torch does), fp32 q rows, 24-bit / fp32 cache rows
            const int hd = p.head_dim, half = hd >> 1, tph = hd / 16;
            const int hb = e_tile / tph, d0 = 8 * (e_tile - hb * tph) + 4 * e_rq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = d0 + j;
                if (hb < p.Hq + p.Hkv) {
                    const float y0 = __fadd_rn(__fmul_rn(v[j], pre_a[j]), -__fmul_rn(u[j], pre_b[j]));
                    const float y1 = __fadd_rn(__fmul_rn(u[j], pre_a[j]), __fmul_rn(v[j], pre_b[j]));
                    if (hb < p.Hq) {
                        float* q = (float*)p.y + (size_t)e_c * p.ldy + hb * hd;
                        q[d] = y0;
                        q[d + half] = y1;
                    } else {
                        gemv_kv_store_x(p, false, pre_pg, pre_pos, hb - p.Hq, d, y0);
                        gemv_kv_store_x(p, false, pre_pg, pre_pos, hb - p.Hq, d + half, y1);
                    }
                } else {
                    gemv_kv_store_x(p, true, pre_pg, pre_pos, hb - p.Hq - p.Hkv, d, v[j]);
                    gemv_kv_store_x(p, true, pre_pg, pre_pos, hb - p.Hq - p.Hkv, d + half, u[j]);
                }
            }
        } else if (MODE == GEMV_QKV) {
            const int hd = p.head_dim, half = hd >> 1, tph = hd / 16;
            const int hb = e_tile / tph, d0 = 8 * (e_tile - hb * tph) + 4 * e_rq;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int d = d0 + j;
                // linear outputs are bf16 activations in the reference; RoPE acts on those
                const float x0 = bf2f(f2bf(v[j])), x1 = bf2f(f2bf(u[j]));
                if (hb < p.Hq + p.Hkv) {
                    const bf16_t y0 = f2bf(x0 * pre_a[j] - x1 * pre_b[j]), y1 = f2bf(x1 * pre_a[j] + x0 * pre_b[j]);
                    if (hb < p.Hq) {
                        bf16_t* q = (bf16_t*)p.y + (size_t)e_c * p.ldy + hb * hd;
                        q[d] = y0;
                        q[d + half] = y1;
                    } else {
                        bf16_t* kc = gemv_kv_row(p, false, e_c, pre_pg, pre_pos, hb - p.Hq);
                        kc[d] = y0;
                        kc[d + half] = y1;
                    }
                } else {
                    bf16_t* vc = gemv_kv_row(p, true, e_c, pre_pg, pre_pos, hb - p.Hq - p.Hkv);
                    vc[d] = f2bf(x0);
                    vc[d + half] = f2bf(x1);
                }
            }
        } else if (MODE == GEMV_LMHEAD) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int row = row0 + j;
                if (row < p.n_rows) {
                    if (v[j] > best) { best = v[j]; besti = row; }   // rows ascend: the first index wins ties
                    if (p.logits_out) p.logits_out[(size_t)e_c * p.n_rows + row] = v[j];
                }
            }
        }
    }
    if (MODE == GEMV_LMHEAD) {
        // block best per batch column: the 32 slots (tile, row quarter) of a column through LDS; first index wins ties
        __syncthreads();                      // every thread has read its partial sums
        float* bv = part_of(0);               // [512] + [512]: 4 KiB of wave 0's region (>= its 8.1 KiB window)
        int* bi = (int*)(bv + KM_NT);         // [512]
        bv[tid] = best;
        bi[tid] = besti;
        __syncthreads();
        if (tid < B) {
            float v0 = -INFINITY;
            int i0 = 0x7fffffff;
            for (int tl = 0; tl < KM_WAVES; ++tl)
                for (int rq = 0; rq < 4; ++rq) {
                    const int e = tl * 64 + rq * 16 + tid;
                    const float x = bv[e];
                    const int ii = bi[e];
                    if (x > v0 || (x == v0 && ii < i0)) { v0 = x; i0 = ii; }
                }
            p.part_val[(size_t)bid * B + tid] = v0;
            p.part_idx[(size_t)bid * B + tid] = i0;
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The down projection at batch 3-16 (K = 11008: 43 k-steps per wave -- 172 registers of activation fragments, or 176 KB of LDS for
// eight rows, neither exists).  One 16-row tile per block; the wave's K slice runs in PHASES of FR fragments (NB = 8 rows: two phases
// of 22; NB = 16 rows, round 5: four of 12): the row slices of a phase go through the wave-private LDS window (row-shaped requests,
// no barrier), the fragments are read just in time, one ds_read_b128 per MFMA; the row slices of the next two phases wait in two
// register sets and replace the window's contents when the phase's MFMAs are done.  Weights: the first phase's steps in flight from
// the start (22 / 12 KiB per wave), every register refilled with the next phase's step as soon as its MFMA has issued.
// y = h + W x in place (+ the per-row scale with fp8 weights).
// ---------------------------------------------------------------------------------------------------------------------
template <int NB> struct KdShape {
    static constexpr int FR = NB == 8 ? 22 : 12;      // fragments (32 elements) per phase and wave
    static constexpr int NPH = NB == 8 ? 2 : 4;       // phases: K <= 8 waves x NPH x FR x 32 = 11264 / 12288
    static constexpr int XP = FR * 64 + 16;           // bytes per staged row slice
    static constexpr int CH = (FR * 4 + 63) / 64;     // 16-byte chunks per lane and row of a phase
};
// EX (exact numerics, batch 3-8; NB = 16): p.x holds the fp32 SwiGLU product; window rows b / b + 8 = the two bf16 terms of batch row b (see
// emmax_decode_km_kernel); the row registers hold the fp32 chunk as two quads (r[2 b], r[2 b + 1]).
template <bool FP8, bool R32, int NB, bool EX = false>
__global__ __launch_bounds__(KM_NT, 2) void emmax_decode_kmd_kernel(GemvParams p) {
    static_assert(!EX || (NB == 16 && !FP8 && R32), "exact numerics: eight batch rows x two terms, bf16 weights, fp32 residual stream");
    extern __shared__ __attribute__((aligned(16))) unsigned char km_smem[];
    using S = KdShape<NB>;
    constexpr int FR = S::FR, NPH = S::NPH, XP = S::XP, CH = S::CH;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g4 = lane >> 4, c16 = lane & 15;
    const int B = p.batch, K = p.K;
    constexpr int KS = FP8 ? 64 : 32, FPS = FP8 ? 2 : 1;   // elements / fragments per load step
    constexpr int NST = FR / FPS;                            // load steps per phase
    const int KT = K / KS;
    const int kq = KT / KM_WAVES, kr = KT % KM_WAVES;
    const int k_lo = wave * kq + min(wave, kr), k_n = kq + (wave < kr ? 1 : 0);   // this wave's load steps (launcher: k_n <= NPH NST)
    float* part = (float*)km_smem;                                                 // [KM_WAVES][64][4]
    unsigned char* xw = km_smem + KM_WAVES * 1024 + (size_t)wave * NB * XP;        // this wave's window
    const int tile = blockIdx.x;

    // epilogue operands of thread (lane l of wave 0): rows 4 (l >> 4) + j, batch column l & 15
    const bool e_on = tid < 64 && c16 < B && (!EX || c16 < 8);
    float pre[4] = {0.f, 0.f, 0.f, 0.f};
    if (e_on) {
        if constexpr (R32) {
            const f32x4_t hv = *(const f32x4_t*)(p.h32 + (size_t)c16 * p.ldh + tile * 16 + 4 * g4);
#pragma unroll
            for (int j = 0; j < 4; ++j) pre[j] = hv[j];
        } else {
            const bf16_t* hp = (const bf16_t*)p.y + (size_t)c16 * p.ldy + tile * 16 + 4 * g4;
#pragma unroll
            for (int j = 0; j < 4; ++j) pre[j] = bf2f(hp[j]);
        }
    }

    // ---- activation row slices of a phase: lane l holds chunks l (and l + 64) of the phase's slice, for every batch row ----
    auto phase_steps = [&](int ph) { return max(0, min(NST, k_n - ph * NST)); };
    auto load_rows = [&](int ph, u32x4_t (&r)[NB][CH]) {
        const int nch = phase_steps(ph) * (KS / 8);             // 16-byte chunks in the slice
        if constexpr (EX) {
            static_assert(!EX || CH == 1, "one chunk per lane and row");
            const float* b32 = (const float*)p.x + (size_t)(k_lo + ph * NST) * KS + (size_t)min(lane, max(nch - 1, 0)) * 8;
#pragma unroll
            for (int b = 0; b < 8; ++b) {
                const float* rp = b32 + (size_t)min(b, B - 1) * p.ldx;
                r[2 * b][0] = *(const u32x4_t*)rp;
                r[2 * b + 1][0] = *(const u32x4_t*)(rp + 4);
            }
            return;
        }
        const bf16_t* base = (const bf16_t*)p.x + (size_t)(k_lo + ph * NST) * KS;
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int h = 0; h < CH; ++h) {
                const int c = min(lane + 64 * h, max(nch - 1, 0));
                r[b][h] = *((const u32x4_t*)(base + (size_t)min(b, B - 1) * p.ldx) + c);
            }
    };
    auto store_rows = [&](int ph, const u32x4_t (&r)[NB][CH]) {
        const int nch = phase_steps(ph) * (KS / 8);
        if constexpr (EX) {
            if (lane < FR * 4) {
#pragma unroll
                for (int b = 0; b < 8; ++b) {
                    const bool live = lane < nch && b < B;
                    u32x4_t hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const u32x4_t& q = r[2 * b + (e >> 1)][0];
                        const hl2_t t = split_hl2(live ? __uint_as_float(q[2 * (e & 1)]) : 0.f, live ? __uint_as_float(q[2 * (e & 1) + 1]) : 0.f);
                        hi[e] = t.hi;
                        lo[e] = t.lo;
                    }
                    *(u32x4_t*)(xw + (size_t)b * XP + (size_t)lane * 16) = hi;
                    *(u32x4_t*)(xw + (size_t)(b + 8) * XP + (size_t)lane * 16) = lo;
                }
            }
            return;
        }
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int h = 0; h < CH; ++h) {
                const int c = lane + 64 * h;
                if (c < FR * 4) {   // chunks past the slice / rows past the batch: zeros
                    const bool live = c < nch && b < B;
                    *(u32x4_t*)(xw + (size_t)b * XP + (size_t)c * 16) = live ? r[b][h] : (u32x4_t){0u, 0u, 0u, 0u};
                }
            }
    };
    u32x4_t xa[NB][CH], xb[NB][CH];   // even / odd phases
    load_rows(0, xa);
    load_rows(1, xb);

    // ---- weights: phase 0 in flight now, later phases refilled register by register ----
    const unsigned w_bytes = (unsigned)((size_t)p.n_groups * 16 * (size_t)K * (FP8 ? 1 : 2));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)w_bytes, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    u32x4_t w[NST];
    auto issue_w = [&](int ph, int s) {   // load step s of phase ph (past the wave's slice: out of range = zeros, no traffic)
        const int ok = (ph * NST + s < k_n) ? -1 : 0;
        const unsigned so = ((unsigned)((tile * KT + k_lo + ph * NST + s) * 1024) & (unsigned)ok) | (w_bytes & (unsigned)~ok);
        w[s] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, so, 2));
    };
#pragma unroll
    for (int s = 0; s < NST; ++s) issue_w(0, s);
    store_rows(0, xa);   // waits for the phase-0 rows only (counted: the phase-1 rows and the weights stay in flight)

    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
    // batch column = lane & 15; columns past the batch are zero (their window rows hold zeros; NB = 8: columns 8-15 have no row at all)
    const unsigned char* xcol = xw + (size_t)min(c16, NB - 1) * XP + (size_t)g4 * 16;
    auto frag = [&](int f) {
        const u32x4_t v = *(const u32x4_t*)(xcol + (size_t)f * 64);
        return __builtin_bit_cast(bf16x8_t, (EX ? (c16 & 7) < B : c16 < B) ? v : (u32x4_t){0u, 0u, 0u, 0u});
    };
    auto mfma_step = [&](int s) {
        if constexpr (FP8) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(km_fp8x8(w[s][0], w[s][1]), frag(2 * s), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(km_fp8x8(w[s][2], w[s][3]), frag(2 * s + 1), acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, w[s]), frag(s), acc, 0, 0, 0);
        }
    };
#pragma unroll
    for (int ph = 0; ph < NPH; ++ph) {
        // the register set this phase's rows came from is free since they went into the window: request the rows two phases ahead
        if (ph + 2 < NPH) { if (ph & 1) load_rows(ph + 2, xb); else load_rows(ph + 2, xa); }
#pragma unroll
        for (int s = 0; s < NST; ++s) {
            mfma_step(s);
            if (ph + 1 < NPH) issue_w(ph + 1, s);   // the register's step of the next phase
        }
        // same wave, in order behind the fragment reads above: the next phase's rows replace this phase's
        if (ph + 1 < NPH) { if ((ph + 1) & 1) store_rows(ph + 1, xb); else store_rows(ph + 1, xa); }
    }

    *(f32x4_t*)(part + ((size_t)wave * 64 + lane) * 4) = acc;
    __syncthreads();
    if (e_on) {
        float v[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int wv = 0; wv < KM_WAVES; ++wv) {
            const f32x4_t a = *(const f32x4_t*)(part + ((size_t)wv * 64 + lane) * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += a[j];
        }
        if constexpr (EX) {   // the lo terms' products: batch column c16 + 8
            float vl[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int wv = 0; wv < KM_WAVES; ++wv) {
                const f32x4_t a = *(const f32x4_t*)(part + ((size_t)wv * 64 + lane + 8) * 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) vl[j] += a[j];
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] += vl[j];
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if constexpr (FP8) v[j] *= p.wscale[tile * 16 + 4 * g4 + j];
            v[j] += pre[j];
        }
        if constexpr (R32) *(f32x4_t*)(p.h32 + (size_t)c16 * p.ldh + tile * 16 + 4 * g4) = (f32x4_t){v[0], v[1], v[2], v[3]};
        bf16_t* hp = (bf16_t*)p.y + (size_t)c16 * p.ldy + tile * 16 + 4 * g4;
#pragma unroll
        for (int j = 0; j < 4; ++j) hp[j] = f2bf(v[j]);
    }
}

template <bool FP8, int NB, bool EX = false>
int kmd_launch(GemvParams p, int B, hipStream_t stream) {
    using S = KdShape<NB>;
    constexpr int KS = FP8 ? 64 : 32, FPS = FP8 ? 2 : 1;
    if (p.K % KS || p.n_rows % 16 || p.attn_part) return -2;
    p.batch = B;
    p.n_groups = p.n_rows / 16;
    if (cdiv(p.K / KS, KM_WAVES) > S::NPH * (S::FR / FPS)) return -2;   // NPH phases of FR fragments per wave
    if (p.K / KS < KM_WAVES) return -2;
    const size_t smem = (size_t)KM_WAVES * 1024 + (size_t)KM_WAVES * NB * S::XP;
    if constexpr (EX) {
        if (!p.h32 || B > 8) return -2;
        hipLaunchKernelGGL((emmax_decode_kmd_kernel<false, true, 16, true>), dim3(p.n_groups), dim3(KM_NT), smem, stream, p);
    } else {
        if (p.h32) hipLaunchKernelGGL((emmax_decode_kmd_kernel<FP8, true, NB>), dim3(p.n_groups), dim3(KM_NT), smem, stream, p);
        else hipLaunchKernelGGL((emmax_decode_kmd_kernel<FP8, false, NB>), dim3(p.n_groups), dim3(KM_NT), smem, stream, p);
    }
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <int MODE, bool NORM, bool XATTN, bool FP8, int NB, bool ROLL, bool EX = false>
int km_launch_nb(GemvParams p, int B, hipStream_t stream, int* grid_out) {
    constexpr int KS = FP8 ? 64 : 32;
    if (p.K % (KM_WAVES * KS) || p.K > KM_WAVES * KM_STEPS * 32 || p.n_rows % 16) return -2;
    if (MODE == GEMV_QKV && (p.head_dim % 16 || p.head_dim < 16)) return -2;
    p.batch = B;
    p.n_groups = p.n_rows / 16;   // tiles
    int grid = min(256, p.n_groups);
    if (MODE == GEMV_LMHEAD) grid = min(grid, p.max_parts);
    if (grid < 1) return -2;
    p.kc = cdiv(p.n_groups, grid);
    if (p.kc > KM_MAX_TILES) return -2;
    if (p.kc & 1) p.kc += 1;   // the loop runs tiles in pairs
    // one region per wave (kernel: window [NB][XPITCH] overlaid by the partial tiles) + sumsq (+ XATTN: the merged rows)
    const size_t wreg = XATTN ? (size_t)p.kc * 1024 : std::max((size_t)NB * (KM_STEPS * 64 + 16), (size_t)p.kc * 1024);
    const size_t smem = (size_t)KM_WAVES * wreg + KM_WAVES * 16 * 4 + (XATTN ? (size_t)(EX ? 16 : B) * p.K * 2 : 0);
    if (smem > 150 * 1024) return -2;
    if (grid_out) *grid_out = grid;
    if constexpr (EX) {   // exact numerics: fp32 rows in (h32 / p.x / the split partials), two terms per batch row, batch <= 8
        if (B > 8 || !p.h32) return -2;
        hipLaunchKernelGGL((emmax_decode_km_kernel<MODE, NORM, XATTN, false, MODE == GEMV_RESID, 16, false, true>), dim3(grid), dim3(KM_NT), smem, stream, p);
        return hipGetLastError() == hipSuccess ? 0 : -4;
    }
    if constexpr (MODE == GEMV_RESID) {   // (NORM modes read the bf16 mirror: h32 is ignored there)
        if (p.h32) {
            hipLaunchKernelGGL((emmax_decode_km_kernel<MODE, NORM, XATTN, FP8, true, NB, ROLL>), dim3(grid), dim3(KM_NT), smem, stream, p);
            return hipGetLastError() == hipSuccess ? 0 : -4;
        }
    }
    hipLaunchKernelGGL((emmax_decode_km_kernel<MODE, NORM, XATTN, FP8, false, NB, ROLL>), dim3(grid), dim3(KM_NT), smem, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
template <int MODE, bool NORM, bool XATTN, bool FP8>
int km_launch_t(const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    if (emmax_tune().km_roll)
        return B <= 8 ? km_launch_nb<MODE, NORM, XATTN, FP8, 8, true>(p, B, stream, grid_out) : km_launch_nb<MODE, NORM, XATTN, FP8, 16, true>(p, B, stream, grid_out);
    return B <= 8 ? km_launch_nb<MODE, NORM, XATTN, FP8, 8, false>(p, B, stream, grid_out) : km_launch_nb<MODE, NORM, XATTN, FP8, 16, false>(p, B, stream, grid_out);
}

// exact numerics (GemvParams::exact), batch 3-8: the EX forms
int km_launch_mode_x(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    switch (mode) {
        case GEMV_QKV: return km_launch_nb<GEMV_QKV, true, false, false, 16, false, true>(p, B, stream, grid_out);
        case GEMV_RESID:
            if (p.attn_part && p.K != p.Hq * 128) return -2;
            return p.attn_part ? km_launch_nb<GEMV_RESID, false, true, false, 16, false, true>(p, B, stream, grid_out)
                               : km_launch_nb<GEMV_RESID, false, false, false, 16, false, true>(p, B, stream, grid_out);
        case GEMV_GATEUP: return km_launch_nb<GEMV_GATEUP, true, false, false, 16, false, true>(p, B, stream, grid_out);
        case GEMV_LMHEAD: return km_launch_nb<GEMV_LMHEAD, true, false, false, 16, false, true>(p, B, stream, grid_out);
        default: return -2;
    }
}

template <bool FP8>
int km_launch_mode(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    switch (mode) {
        case GEMV_QKV: return km_launch_t<GEMV_QKV, true, false, FP8>(p, B, stream, grid_out);
        case GEMV_RESID:
            if (p.attn_part && p.K != p.Hq * 128) return -2;
            if constexpr (!FP8) {
                // the bf16 o-proj with the split merge stays on decode_mfma.hip (12.5 against 15.4 us at B = 8: the merge of 8 rows
                // queues behind this kernel's 16 KiB weight heads); with fp8 weights this kernel is the faster one (10.3 against 10.8)
                if (p.attn_part) return -2;
                return km_launch_t<GEMV_RESID, false, false, FP8>(p, B, stream, grid_out);
            } else {
                return p.attn_part ? km_launch_t<GEMV_RESID, false, true, FP8>(p, B, stream, grid_out) : km_launch_t<GEMV_RESID, false, false, FP8>(p, B, stream, grid_out);
            }
        case GEMV_GATEUP: return km_launch_t<GEMV_GATEUP, true, false, FP8>(p, B, stream, grid_out);
        case GEMV_LMHEAD: return km_launch_t<GEMV_LMHEAD, true, false, FP8>(p, B, stream, grid_out);
        case GEMV_PLAIN: return km_launch_t<GEMV_PLAIN, false, false, FP8>(p, B, stream, grid_out);
        default: return -2;
    }
}

}  // namespace

// raise the dynamic-LDS limit of every instantiation (call once, outside graph capture)
int decode_km_init() {
    static int done = -1;
    if (done == 0) return 0;
    const int lim = 150 * 1024;
    hipError_t e = hipSuccess;
#define KM_SET4(M, N_, X, F, R, NB_, RL) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_km_kernel<M, N_, X, F, R, NB_, RL>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
#define KM_SET3(M, N_, X, F, R, NB_) KM_SET4(M, N_, X, F, R, NB_, false); KM_SET4(M, N_, X, F, R, NB_, true)
#define KM_SET2(M, N_, X, F, R) KM_SET3(M, N_, X, F, R, 8); KM_SET3(M, N_, X, F, R, 16)
#define KM_SET1(M, N_, X, F) KM_SET2(M, N_, X, F, false); if (M == GEMV_RESID) KM_SET2(M, N_, X, F, true)
#define KM_SET(M, N_, X) KM_SET1(M, N_, X, false); KM_SET1(M, N_, X, true)
    KM_SET(GEMV_QKV, true, false); KM_SET1(GEMV_RESID, false, true, true); KM_SET(GEMV_RESID, false, false); KM_SET(GEMV_GATEUP, true, false);
    KM_SET(GEMV_LMHEAD, true, false); KM_SET(GEMV_PLAIN, false, false);
#undef KM_SET
#undef KM_SET1
#undef KM_SET2
#undef KM_SET3
#undef KM_SET4
#define KD_SET(F, R, NB_) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_kmd_kernel<F, R, NB_>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    KD_SET(false, false, 8); KD_SET(true, false, 8); KD_SET(false, true, 8); KD_SET(true, true, 8);
    KD_SET(false, false, 16); KD_SET(true, false, 16); KD_SET(false, true, 16); KD_SET(true, true, 16);
#undef KD_SET
#define KX_SET(M, N_, X) \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_km_kernel<M, N_, X, false, M == GEMV_RESID, 16, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    KX_SET(GEMV_QKV, true, false); KX_SET(GEMV_RESID, false, true); KX_SET(GEMV_RESID, false, false); KX_SET(GEMV_GATEUP, true, false); KX_SET(GEMV_LMHEAD, true, false);
#undef KX_SET
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_kmd_kernel<false, true, 16, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim);
    done = (e == hipSuccess) ? 0 : -4;
    return done;
}

// tuning switch `km` = 0 keeps every batch >= 3 projection on decode_mfma.hip (the A/B partner)
bool decode_km_enabled() { return emmax_tune().km != 0; }

// perm: 0 natural row order, 1 qkv (head_dim given), 2 gate/up (source in decode.hip's 16-row interleaved order)
int launch_repack_km(const void* src, int ld, void* dst, int N, int K, int perm, int head_dim, hipStream_t stream) {
    if (N % 16 || K % 32 || ld % 8) return -1;
    if (perm == 1 && (head_dim % 16 || N % head_dim)) return -1;
    if (perm == 2 && N % 32) return -1;
    hipLaunchKernelGGL(emmax_repack_km_kernel, dim3(2048), dim3(256), 0, stream, (const bf16_t*)src, ld, (u32x4_t*)dst, N, K, perm, head_dim);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// p.W: the km copy of the matrix (launch_repack_km; fp8: decode_mfma.hip's e4m3 tiles of the permuted rows + p.wscale in the same
// row order).  -2: shape outside this kernel (K % 256, K > 4096, more than 8 tiles per block) -- the caller uses decode_mfma.hip.
int launch_decode_km(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    if (B < 1 || B > EMMAX_MAX_DECODE_BATCH) return -2;
    if (B > 16) return launch_decode_kmp(mode, p, B, stream, grid_out);   // two batch tiles: decode_kmp.hip
    if (decode_km_init() != 0) return -4;
    if (p.exact) {   // exact numerics: batch <= 8 (two terms per batch row in the sixteen MFMA columns), bf16 weights
        if (B > 8 || p.wscale) return -2;
        // RESID launches without split partials (fp32 rows in p.x): the o-proj behind a one-split attention launch on the K-split kernel when its K fits,
        // the down projection (and any other K) on the phased kernel
        if (mode == GEMV_RESID && !p.attn_part && (p.K % (KM_WAVES * 32) || p.K > KM_WAVES * KM_STEPS * 32)) return kmd_launch<false, 16, true>(p, B, stream);
        return km_launch_mode_x(mode, p, B, stream, grid_out);
    }
    if (mode == GEMV_RESID && !p.attn_part && p.K > KM_WAVES * KM_STEPS * 32) {   // the down projection: two K phases (natural row order copy)
        if (!emmax_tune().km_down) return -2;   // A/B partner: decode_mfma.hip
        if (B <= 8) return p.wscale ? kmd_launch<true, 8>(p, B, stream) : kmd_launch<false, 8>(p, B, stream);
        return p.wscale ? kmd_launch<true, 16>(p, B, stream) : kmd_launch<false, 16>(p, B, stream);
    }
    return p.wscale ? km_launch_mode<true>(mode, p, B, stream, grid_out) : km_launch_mode<false>(mode, p, B, stream, grid_out);
}
