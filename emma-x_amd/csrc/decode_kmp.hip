// decode_kmp.hip -- batch 17-32 decode projections with K <= 4096 on MFMA: decode_km.hip's K split across the waves with TWO 16-wide
// batch tiles per weight tile ("kmp": K split, MFMA, Phased).  Replaces the q_len == 1 linears of HF `LlamaDecoderLayer` at batch 17-32
// (cached branch of prismatic/extern/hf/modeling_prismatic.py:325-341; SURVEY.md 8(f)4: fleets of cameras -- the reference is bs = 1,
// prismatic/models/vlms/prismatic.py:627-696).  VERDICT r04 next #5.
//
// decode_km.hip keeps a wave's whole K slice of the activations in registers as MFMA B fragments: 64 registers per 16 batch rows.  Two
// batch tiles would take 128, next to 128 for the weights in flight -- the register file is 256.  Here the slice runs in PHASES of
// PH = 4 k-steps (128 elements): the rows of a phase go through the wave-private LDS window ([32 rows][272 B], row-shaped requests, no
// barrier), its 4 x 2 fragments are read ONCE into 32 registers and serve every weight tile of the block; the accumulators of all tiles
// of the block (at most TMAX) x 2 batch tiles stay in registers across the phases; the rows of the next phase are requested while this
// one computes.  The loop nest is (phase, tile, step) and the weight registers form a ring over it: the register of unit (phase, tile,
// step) is refilled with (phase + LOOK, tile, step) as soon as its two MFMAs have issued, so TMAX x PH x LOOK KiB per wave stay in
// flight whatever the tile count -- three shapes: TMAX = 6 tiles x 1 phase of lookahead (gate/up, lm-head), 3 x 2 (qkv), 1 x 4 (o-proj:
// the whole slice at once).  Same weight copies as decode_km.hip (launch_repack_km; with fp8 weights the e4m3 tiles of 16 rows x 64 k
// and a scale per row: a load step is then two fragments, converted to bf16 in registers, and the lookahead counts twice the phases
// for the same bytes in flight), same fused prologues / epilogues, same arithmetic (y = rstd * W (x .* g), the eight K-slice partial
// tiles meet in LDS at the end of the block).  The K slices of the waves are whole load steps and may differ by one (the fp8 down
// projection: 172 steps over 8 waves).
#include <cstdlib>
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int KP_WAVES = 8;
constexpr int KP_THREADS = KP_WAVES * 64;
constexpr int KP_PH = 4;                    // k-steps (32 elements) per phase
// phases NPH: 4 (K <= 8 waves x 4 x 4 x 32 = 4096) or 11 (the down projection, K <= 11264: one tile per block, TMAX = 1)
constexpr int KP_ROWS = 32;                 // staged batch rows = two MFMA batch tiles
constexpr int KP_XP = KP_PH * 64 + 16;      // window pitch in bytes (+16: the rows of a fragment read start on different bank slots)
constexpr int KP_RPL = KP_ROWS * KP_PH * 4 / 64;   // 16-byte chunks per lane and phase: rows (lane >> 4) + 4 j, chunk lane & 15
#ifndef KP_XD_LONG
#define KP_XD_LONG 2
#endif
#ifndef KP_AUX2
#define KP_AUX2 0
#endif
#ifndef KP_HALF_SYNC
#define KP_HALF_SYNC 1
#endif

// phases of weights in flight (fp8 tiles carry 64 k per KiB: twice the phases for the same bytes)
template <int TMAX, bool FP8> struct KpLook { static constexpr int L = (TMAX >= 4 ? 1 : TMAX >= 2 ? 2 : 4) * (FP8 ? 2 : 1); };

// two fp8 e4m3 pairs (the low / high half of a dword) -> two bf16, exact (as decode_km.hip)
template <bool HI>
__device__ __forceinline__ uint32_t kp_fp8x2(uint32_t v) {
    return __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(v, 1.0f, HI));
}
__device__ __forceinline__ bf16x8_t kp_fp8x8(uint32_t lo, uint32_t hi) {
    const u32x4_t v = {kp_fp8x2<false>(lo), kp_fp8x2<true>(lo), kp_fp8x2<false>(hi), kp_fp8x2<true>(hi)};
    return __builtin_bit_cast(bf16x8_t, v);
}

// activation row sets in flight (registers): the eleven-phase down projection keeps two
template <int TMAX, int NPH> struct KpRowSets { static constexpr int N = (NPH > 4 && TMAX == 1) ? KP_XD_LONG : 1; };

// NH = 2 (round 6, batches of 33-64 rows): the block's eight waves are two HALVES of four -- half h stages rows 32 h .. 32 h + 31 and owns their two batch
// tiles, its four waves split K four ways (twice the phases per wave); both halves stream the block's weight tiles (the second request is an L2 hit: HBM
// reads them once), so the registers of a wave are those of the 32-row kernel.  Epilogue slots (tile, half) are dealt to the eight waves in passes.
template <int MODE, bool NORM, bool R32, int TMAX, int NPH, bool FP8, int NH = 1>
__global__ __launch_bounds__(KP_THREADS, 2) void emmax_decode_kmp_kernel(GemvParams p) {
    static_assert(NH == 1 || (NH == 2 && MODE != GEMV_LMHEAD), "halves: 1 or 2; the lm-head of 33-64 rows runs as two launches");
    constexpr int KW = KP_WAVES / NH;          // K slices = waves of a half
    constexpr int KP_NPH = NPH;
    extern __shared__ __attribute__((aligned(16))) unsigned char kp_smem[];
    constexpr int LOOK = KpLook<TMAX, FP8>::L;
    constexpr int KS = FP8 ? 64 : 32;                  // elements per load step (one 1 KiB tile)
    constexpr int PHS = KP_PH * 32 / KS;               // load steps per phase
    constexpr int RING = TMAX * PHS * LOOK;            // weight registers (16-byte loads of 1 KiB tiles) of a wave
    constexpr int WREG = TMAX * 2048 > KP_ROWS * KP_XP ? TMAX * 2048 : KP_ROWS * KP_XP;   // a wave's LDS region: window, later its partial tiles
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wk = wave % KW, half = wave / KW, row0 = 32 * half;   // K slice, half, first batch row of the half
    const int g4 = lane >> 4, c16 = lane & 15;
    const int B = p.batch, K = p.K;
    const int KT = K / KS;                             // load steps of a row
    const int kq = KT / KW, kr = KT % KW;
    const int k_lo = wk * kq + min(wk, kr), k_n = kq + (wk < kr ? 1 : 0);   // this wave's load steps (launcher: k_n <= NPH PHS)
    unsigned char* xw = kp_smem + (size_t)wave * WREG;                     // this wave's window [32][KP_XP]
    auto part_of = [&](int w) { return (float*)(kp_smem + (size_t)w * WREG); };   // [TMAX][2][64][4]
    float* sumsq = (float*)(kp_smem + (size_t)KP_WAVES * WREG);            // [KP_WAVES][32]

    const int G = gridDim.x, bid = blockIdx.x;
    const int n_tiles = p.n_groups;
    const int tq = n_tiles / G, tr = n_tiles % G;
    const int t_lo = bid * tq + min(bid, tr), ntb = tq + (bid < tr ? 1 : 0);   // launcher: at most TMAX

    // ---- epilogue operands of this thread's slots, requested before anything else: thread (tl = tid >> 6, lane) finalises tile tl of
    // the block, rows 4 (lane >> 4) + j, batch columns (lane & 15) and 16 + (lane & 15) ----
    // epilogue slots: slot s = (tile s / NH, half s % NH) of the block; wave w serves slots w, w + 8, ... (NPASS passes; the modes with operands to
    // prefetch -- the residual rows, the RoPE tables -- fit one pass)
    constexpr int NPASS = (TMAX * NH + KP_WAVES - 1) / KP_WAVES;
    static_assert(NPASS == 1 || (MODE != GEMV_RESID && MODE != GEMV_QKV), "two epilogue passes: no prefetched operands");
    const int e_slot = tid >> 6, e_rq = lane >> 4;
    const int e_tl = e_slot / NH, e_hf = e_slot % NH;
    constexpr bool e_pairs = MODE == GEMV_QKV || MODE == GEMV_GATEUP;      // rows r, r + 8 of a tile belong together
    const int e_tile = t_lo + min(e_tl, max(ntb - 1, 0));
    bool e_on[2];
    float pre_a[2][4], pre_b[2][4];
    int pre_pos[2] = {0, 0}, pre_pg[2] = {0, 0};
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int e_c = 32 * e_hf + nt * 16 + c16;
        e_on[nt] = e_tl < ntb && e_tl < TMAX && e_c < B && (!e_pairs || e_rq < 2);
#pragma unroll
        for (int j = 0; j < 4; ++j) pre_a[nt][j] = pre_b[nt][j] = 0.f;
        if (e_on[nt]) {
            if (MODE == GEMV_RESID) {
                if constexpr (R32) {
                    const f32x4_t hv = *(const f32x4_t*)(p.h32 + (size_t)e_c * p.ldh + e_tile * 16 + 4 * e_rq);
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre_a[nt][j] = hv[j];
                } else {
                    const bf16_t* hp = (const bf16_t*)p.y + (size_t)e_c * p.ldy + e_tile * 16 + 4 * e_rq;
#pragma unroll
                    for (int j = 0; j < 4; ++j) pre_a[nt][j] = bf2f(hp[j]);
                }
            } else if (MODE == GEMV_QKV) {
                pre_pos[nt] = p.ctx_len[e_c];
                pre_pg[nt] = p.page_table[(size_t)e_c * p.max_pages + pre_pos[nt] / p.page];
                const int hd = p.head_dim, half = hd >> 1, tph = hd / 16;
                const int hb = e_tile / tph, d0 = 8 * (e_tile - hb * tph) + 4 * e_rq;
                if (hb < p.Hq + p.Hkv) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        pre_a[nt][j] = p.cos_t[(size_t)pre_pos[nt] * half + d0 + j];
                        pre_b[nt][j] = p.sin_t[(size_t)pre_pos[nt] * half + d0 + j];
                    }
                }
            }
        }
    }

    // ---- weight stream: buffer loads, the (tile, step) offset in an SGPR, the lane's 16 bytes in the VGPR offset ----
    const unsigned w_bytes = (unsigned)((size_t)p.n_groups * 16 * (size_t)K * (FP8 ? 1 : 2));
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.W, 0, (int)w_bytes, 0x00020000);
    const unsigned voff = (unsigned)lane * 16u;
    u32x4_t w[RING];
    // unit (ph, tl, s) lives in register ((ph % LOOK) * TMAX + tl) * PHS + s; past the block's tiles / the slice: out of range = zeros, no traffic
    // (measured and removed, profiles/r06_kmp_rot_ab.txt: blocks walking their phases in ROTATED order -- so that the CUs of an XCD would not ask L2 for the same
    // row chunks at the same time -- is slower, qkv 23.0 -> 26.8 us, gate/up 33.6 -> 36.3 at 32 rows: the simultaneous requests for the same rows are what L2 merges)
    auto issue_w = [&](int ph, int tl, int s) {
        const int ks = ph * PHS + s;
        const int ok = (ph < KP_NPH && tl < ntb && ks < k_n) ? -1 : 0;
        const unsigned so = ((unsigned)(((t_lo + tl) * KT + k_lo + ks) * 1024) & (unsigned)ok) | (w_bytes & (unsigned)~ok);
        // aux 2 = nt (streamed once); NH = 2: the other half of the block requests the same tile -- default policy, so that the second request hits L2
        // (with nt both went to HBM: gate/up at 64 rows 61.8 us = twice the 32-row launch)
        w[((ph % LOOK) * TMAX + tl) * PHS + s] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(wrsrc, voff, so, NH == 2 ? KP_AUX2 : 2));
    };

    // ---- activations of a phase: lane l holds chunk l & 15 (8 elements) of rows (l >> 4) + 4 j of the phase's 128-element slice ----
    const int r0 = lane >> 4;
    // XD row sets: the rows of phase ph wait in set ph % XD and were requested XD phases before they enter the window (the down projection's eleven
    // short phases -- one tile, 8 MFMAs each -- stalled on every phase's rows at XD = 1: an L2 round trip per phase)
    constexpr int XD = KpRowSets<TMAX, NPH>::N;
    u32x4_t xr[XD][KP_RPL], nwv[XD];
#pragma unroll
    for (int d = 0; d < XD; ++d) nwv[d] = (u32x4_t){0u, 0u, 0u, 0u};
    float ssl[KP_RPL];   // NORM: this lane's running sum of squares of rows r0 + 4 j
#pragma unroll
    for (int j = 0; j < KP_RPL; ++j) ssl[j] = 0.f;
    auto phase_live = [&](int ph) { return ph * KP_PH * 4 + c16 < k_n * (KS / 8); };   // the lane's chunk lies inside the wave's slice
    // rows as buffer loads: one descriptor, a 32-bit lane offset per row (8 registers instead of 8 x 64-bit addresses per phase -- the
    // row requests of the three later phases had their addresses precomputed and held), the phase in the scalar offset; a chunk past the
    // slice reads the neighbouring slice or, past the matrix, zeros: it is masked at the store either way
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)B * (unsigned)p.ldx * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t nrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.norm_w, 0, NORM ? K * 2 : 0, 0x00020000);
    unsigned xvoff[KP_RPL];
#pragma unroll
    for (int j = 0; j < KP_RPL; ++j) xvoff[j] = (unsigned)min(row0 + r0 + 4 * j, B - 1) * (unsigned)p.ldx * 2u + (unsigned)c16 * 16u;
    auto load_rows = [&](int ph, auto SET) {
        constexpr int st = decltype(SET)::value;
        const unsigned so = (unsigned)(k_lo * KS + ph * KP_PH * 32) * 2u;   // first byte of the phase inside a row
#pragma unroll
        for (int j = 0; j < KP_RPL; ++j) xr[st][j] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(xrsrc, xvoff[j], so, 0));
        if constexpr (NORM) nwv[st] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(nrsrc, (unsigned)c16 * 16u, so, 0));
    };
    auto store_rows = [&](int ph, auto SET) {   // (NORM: x .* g rounded to bf16, squares added to the lane's row sums)
        constexpr int st = decltype(SET)::value;
        const bool live = phase_live(ph);
#pragma unroll
        for (int j = 0; j < KP_RPL; ++j) {
            u32x4_t o;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                uint32_t v = (live && row0 + r0 + 4 * j < B) ? xr[st][j][e] : 0u;
                if constexpr (NORM) {
                    const float a = bf_lo(v), c = bf_hi(v);
                    ssl[j] += a * a + c * c;
                    v = pack_bf16x2(a * bf_lo(nwv[st][e]), c * bf_hi(nwv[st][e]));
                }
                o[e] = v;
            }
            *(u32x4_t*)(xw + (size_t)(r0 + 4 * j) * KP_XP + (size_t)c16 * 16) = o;
            // (an opaque use: the sums are only read after the last phase, and left alone hipcc SINKS every square there -- keeping the
            // fp32 operands of all four phases, 256 values per lane, in registers and scratch until then)
            if constexpr (NORM) asm volatile("" : "+v"(ssl[j]));
        }
    };

#define KP_SET(PHV) std::integral_constant<int, (PHV) % XD>{}
    load_rows(0, KP_SET(0));
    if constexpr (XD > 1 && 1 < KP_NPH) load_rows(1, KP_SET(1));
    if constexpr (XD > 2 && 2 < KP_NPH) load_rows(2, KP_SET(2));
    static_assert(XD <= 3, "row sets are spelled out");
    // the first LOOK phases of weights, behind the activation requests (those are waited for by count)
#pragma unroll
    for (int ph = 0; ph < LOOK; ++ph)
#pragma unroll
        for (int tl = 0; tl < TMAX; ++tl)
#pragma unroll
            for (int s = 0; s < PHS; ++s) issue_w(ph, tl, s);
    __builtin_amdgcn_sched_barrier(0);
    store_rows(0, KP_SET(0));
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (XD < KP_NPH) load_rows(XD, KP_SET(XD));   // set 0 is free again
    __builtin_amdgcn_sched_barrier(0);

    f32x4_t acc[TMAX][2];
#pragma unroll
    for (int tl = 0; tl < TMAX; ++tl) acc[tl][0] = acc[tl][1] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    // (the phases are spelled out with a compile-time phase: left to `#pragma unroll`, hipcc kept the loop rolled in some instantiations
    // and the ring index became dynamic -- the weight registers went to scratch; as a lambda calling the row lambdas, the captured
    // row / sum registers were materialised in a stack frame)
#define KP_RUN_PHASE(PHV)                                                                                               \
    {                                                                                                                   \
        constexpr int ph = PHV;                                                                                         \
        /* NH = 2: the two halves request the same weight tiles -- kept within a phase of each other (one barrier per phase), \
           so that the second request finds the line in flight or present in L2 instead of reading HBM again (PMC: gate/up  \
           347 -> 201 MB per launch, 55 -> 44 us; qkv 183 -> 107 MB, 37 -> 35 us).  Not for the one-tile o-proj (14.8 -> 16.0 us) \
           nor with fp8 tiles (half the bytes: the launch is bound by the L2 -> CU path either way, the barrier only costs) */ \
        if constexpr (NH == 2 && KP_HALF_SYNC && !FP8 && TMAX >= 3) __syncthreads();                                    \
        /* the phase's fragments: batch tile nt, k-step s: row 16 nt + c16, chunk 4 s + g4 (wave-private window: only this  \
           wave's own LDS writes have to have landed, no barrier) */                                                    \
        bf16x8_t xf[KP_PH][2];                                                                                          \
        _Pragma("unroll") for (int s = 0; s < KP_PH; ++s)                                                               \
            _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                            \
                xf[s][nt] = *(const bf16x8_t*)(xw + (size_t)(16 * nt + c16) * KP_XP + (size_t)(4 * s + g4) * 16);        \
        /* the window is free again (same wave, LDS in order): the next phase's rows replace this one's, the rows after them \
           are requested (pinned: left alone, the scheduler gathers the row requests of every phase at the top: 128 registers) */ \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        if constexpr (ph + 1 < KP_NPH) {                                                                                \
            store_rows(ph + 1, KP_SET(ph + 1));                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                                                          \
            if constexpr (ph + 1 + XD < KP_NPH) load_rows(ph + 1 + XD, KP_SET(ph + 1 + XD));                            \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
        _Pragma("unroll") for (int tl = 0; tl < TMAX; ++tl) {                                                           \
            if (TMAX <= 3 || tl < ntb) { /* block-uniform; small shapes run straight-line (a missing tile: zero weights) */ \
                _Pragma("unroll") for (int s = 0; s < PHS; ++s) {                                                       \
                    constexpr int slot0 = (ph % LOOK) * TMAX * PHS;                                                     \
                    const u32x4_t wv = w[slot0 + tl * PHS + s];                                                         \
                    if constexpr (FP8) { /* a 16 x 64 e4m3 tile: two bf16 fragments */                                  \
                        const bf16x8_t wlo = kp_fp8x8(wv[0], wv[1]), whi = kp_fp8x8(wv[2], wv[3]);                      \
                        acc[tl][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xf[2 * s][0], acc[tl][0], 0, 0, 0);   \
                        acc[tl][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wlo, xf[2 * s][1], acc[tl][1], 0, 0, 0);   \
                        acc[tl][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xf[2 * s + 1][0], acc[tl][0], 0, 0, 0); \
                        acc[tl][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(whi, xf[2 * s + 1][1], acc[tl][1], 0, 0, 0); \
                    } else {                                                                                            \
                        const bf16x8_t wf = __builtin_bit_cast(bf16x8_t, wv);                                           \
                        acc[tl][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[s][0], acc[tl][0], 0, 0, 0);        \
                        acc[tl][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf[s][1], acc[tl][1], 0, 0, 0);        \
                    }                                                                                                   \
                    issue_w(ph + LOOK, tl, s); /* the register's unit LOOK phases ahead */                              \
                }                                                                                                       \
            }                                                                                                           \
        }                                                                                                               \
        __builtin_amdgcn_sched_barrier(0);                                                                              \
    }
#define KP_PHASE_IF(PHV) if constexpr (PHV < KP_NPH) KP_RUN_PHASE(PHV)
    KP_PHASE_IF(0) KP_PHASE_IF(1) KP_PHASE_IF(2) KP_PHASE_IF(3) KP_PHASE_IF(4) KP_PHASE_IF(5)
    KP_PHASE_IF(6) KP_PHASE_IF(7) KP_PHASE_IF(8) KP_PHASE_IF(9) KP_PHASE_IF(10)
#undef KP_PHASE_IF
#undef KP_RUN_PHASE
#undef KP_SET
    static_assert(KP_NPH <= 11, "phases are spelled out");

    // ---- the wave's partial tiles into its region (the window is dead), row sums of squares, one barrier ----
#pragma unroll
    for (int tl = 0; tl < TMAX; ++tl)
        if (tl < ntb) {
            *(f32x4_t*)(part_of(wave) + ((size_t)(tl * 2 + 0) * 64 + lane) * 4) = acc[tl][0];
            *(f32x4_t*)(part_of(wave) + ((size_t)(tl * 2 + 1) * 64 + lane) * 4) = acc[tl][1];
        }
    if constexpr (NORM) {
#pragma unroll
        for (int j = 0; j < KP_RPL; ++j) {
            const float t = row16_sum(ssl[j]);   // the 16 lanes of a DPP row hold the 16 chunks of row r0 + 4 j
            if (c16 == 0) sumsq[wave * 32 + r0 + 4 * j] = t;
        }
    }
    __syncthreads();

    // ---- epilogue: thread (tl, lane): rows 4 (lane >> 4) + j of tile tl, batch columns c16 and 16 + c16 ----
    float best[2] = {-INFINITY, -INFINITY};
    int besti[2] = {0x7fffffff, 0x7fffffff};
#pragma unroll
    for (int ps = 0; ps < NPASS; ++ps) {
        const int q_slot = e_slot + KP_WAVES * ps, q_tl = q_slot / NH, q_hf = q_slot % NH;
        const int q_tile = t_lo + min(q_tl, max(ntb - 1, 0));
        bool q_on[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
            q_on[nt] = ps == 0 ? e_on[nt] : (q_tl < ntb && q_tl < TMAX && 32 * q_hf + nt * 16 + c16 < B && (!e_pairs || e_rq < 2));
    #pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            const int e_c = 32 * q_hf + nt * 16 + c16, e_cl = nt * 16 + c16;   // batch row; column inside the half
            float v[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};   // u: the partner rows r + 8 (qkv, gate/up)
            if (!q_on[nt]) continue;
    #pragma unroll
            for (int wv = 0; wv < KW; ++wv) {
                const f32x4_t a = *(const f32x4_t*)(part_of(q_hf * KW + wv) + ((size_t)(q_tl * 2 + nt) * 64 + lane) * 4);
    #pragma unroll
                for (int j = 0; j < 4; ++j) v[j] += a[j];
                if (e_pairs) {
                    const f32x4_t b2 = *(const f32x4_t*)(part_of(q_hf * KW + wv) + ((size_t)(q_tl * 2 + nt) * 64 + lane + 32) * 4);
    #pragma unroll
                    for (int j = 0; j < 4; ++j) u[j] += b2[j];
                }
            }
            if constexpr (NORM) {
                float t = 0.f;
    #pragma unroll
                for (int wv = 0; wv < KW; ++wv) t += sumsq[(q_hf * KW + wv) * 32 + e_cl];
                const float sc = rsqrtf(t / (float)K + p.eps);
    #pragma unroll
                for (int j = 0; j < 4; ++j) { v[j] *= sc; u[j] *= sc; }
            }
            if constexpr (FP8) {   // per-row weight scales (km row order)
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] *= p.wscale[q_tile * 16 + 4 * e_rq + j];
                    if (e_pairs) u[j] *= p.wscale[q_tile * 16 + 8 + 4 * e_rq + j];
                }
            }
            const int row0 = q_tile * 16 + 4 * e_rq;   // natural-order matrices
            if (MODE == GEMV_PLAIN) {
                bf16_t* yp = (bf16_t*)p.y + (size_t)e_c * p.ldy + row0;
    #pragma unroll
                for (int j = 0; j < 4; ++j) yp[j] = f2bf(v[j]);
            } else if (MODE == GEMV_RESID) {
                if constexpr (R32)
                    *(f32x4_t*)(p.h32 + (size_t)e_c * p.ldh + row0) = (f32x4_t){pre_a[nt][0] + v[0], pre_a[nt][1] + v[1], pre_a[nt][2] + v[2], pre_a[nt][3] + v[3]};
                bf16_t* hp = (bf16_t*)p.y + (size_t)e_c * p.ldy + row0;
    #pragma unroll
                for (int j = 0; j < 4; ++j) hp[j] = f2bf(pre_a[nt][j] + v[j]);
            } else if (MODE == GEMV_GATEUP) {
                bf16_t* yp = (bf16_t*)p.y + (size_t)e_c * p.ldy + 8 * q_tile + 4 * e_rq;
    #pragma unroll
                for (int j = 0; j < 4; ++j) yp[j] = f2bf(silu(v[j]) * u[j]);
            } else if (MODE == GEMV_QKV) {
                const int hd = p.head_dim, half = hd >> 1, tph = hd / 16;
                const int hb = q_tile / tph, d0 = 8 * (q_tile - hb * tph) + 4 * e_rq;
                const int pos = pre_pos[nt], pg = pre_pg[nt];
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int d = d0 + j;
                    // linear outputs are bf16 activations in the reference; RoPE acts on those
                    const float x0 = bf2f(f2bf(v[j])), x1 = bf2f(f2bf(u[j]));
                    if (hb < p.Hq + p.Hkv) {
                        const bf16_t y0 = f2bf(x0 * pre_a[nt][j] - x1 * pre_b[nt][j]), y1 = f2bf(x1 * pre_a[nt][j] + x0 * pre_b[nt][j]);
                        if (hb < p.Hq) {
                            bf16_t* q = (bf16_t*)p.y + (size_t)e_c * p.ldy + hb * hd;
                            q[d] = y0;
                            q[d + half] = y1;
                        } else {
                            bf16_t* kc = gemv_kv_row(p, false, e_c, pg, pos, hb - p.Hq);
                            kc[d] = y0;
                            kc[d + half] = y1;
                        }
                    } else {
                        bf16_t* vc = gemv_kv_row(p, true, e_c, pg, pos, hb - p.Hq - p.Hkv);
                        vc[d] = f2bf(x0);
                        vc[d + half] = f2bf(x1);
                    }
                }
            } else if (MODE == GEMV_LMHEAD) {
    #pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int row = row0 + j;
                    if (row < p.n_rows) {
                        if (v[j] > best[nt]) { best[nt] = v[j]; besti[nt] = row; }   // rows ascend: the first index wins ties
                        if (p.logits_out) p.logits_out[(size_t)e_c * p.n_rows + row] = v[j];
                    }
                }
            }
        }
    }
    if (MODE == GEMV_LMHEAD) {
        // block best per batch column: the (tile, row quarter) slots of a column through LDS; first index wins ties
        __syncthreads();                      // every thread has read its partial sums
        float* bv = part_of(0);               // [2][512] (wave 0's region: >= 8.5 KiB)
        int* bi = (int*)(bv + 2 * KP_THREADS);
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            bv[nt * KP_THREADS + tid] = best[nt];
            bi[nt * KP_THREADS + tid] = besti[nt];
        }
        __syncthreads();
        if (tid < B) {
            const int nt = tid >> 4, c = tid & 15;
            float v0 = -INFINITY;
            int i0 = 0x7fffffff;
            for (int tl = 0; tl < KP_WAVES; ++tl)
                for (int rq = 0; rq < 4; ++rq) {
                    const int e = nt * KP_THREADS + tl * 64 + rq * 16 + c;
                    const float x = bv[e];
                    const int ii = bi[e];
                    if (x > v0 || (x == v0 && ii < i0)) { v0 = x; i0 = ii; }
                }
            p.part_val[(size_t)bid * B + tid] = v0;
            p.part_idx[(size_t)bid * B + tid] = i0;
        }
    }
}

template <int TMAX> constexpr size_t kp_smem_bytes() {
    return (size_t)KP_WAVES * (TMAX * 2048 > KP_ROWS * KP_XP ? TMAX * 2048 : KP_ROWS * KP_XP) + KP_WAVES * 32 * 4;
}

template <int MODE, bool NORM, bool R32, int TMAX, int NPH, bool FP8, int NH = 1>
int kp_launch_r(const GemvParams& p, int grid, hipStream_t stream) {
    auto kern = emmax_decode_kmp_kernel<MODE, NORM, R32, TMAX, NPH, FP8, NH>;
    static bool attr_done = false;   // per instantiation (first call: outside graph capture -- decode_kmp_init)
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kp_smem_bytes<TMAX>()) != hipSuccess) return -4;
        attr_done = true;
    }
    if (grid > 0) hipLaunchKernelGGL(kern, dim3(grid), dim3(KP_THREADS), kp_smem_bytes<TMAX>(), stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <int MODE, bool NORM, int TMAX, int NPH, bool FP8, int NH = 1>
int kp_launch_tf(const GemvParams& p, int grid, bool r32, bool init_only, hipStream_t stream) {
    if constexpr (MODE == GEMV_RESID) {   // (NORM modes read the bf16 mirror of the fp32 residual stream, as in decode_km.hip)
        if (init_only || r32) {
            const int r = kp_launch_r<MODE, NORM, true, TMAX, NPH, FP8, NH>(p, init_only ? 0 : grid, stream);
            if (r || !init_only) return r;
        }
    }
    return kp_launch_r<MODE, NORM, false, TMAX, NPH, FP8, NH>(p, init_only ? 0 : grid, stream);
}
template <int MODE, bool NORM, int TMAX, int NPH = 4, int NH = 1>
int kp_launch_tm(const GemvParams& p, int grid, bool r32, bool init_only, hipStream_t stream) {
    if (init_only || p.wscale) {
        const int r = kp_launch_tf<MODE, NORM, TMAX, NPH, true, NH>(p, grid, r32, init_only, stream);
        if (r || !init_only) return r;
    }
    return kp_launch_tf<MODE, NORM, TMAX, NPH, false, NH>(p, grid, r32, init_only, stream);
}

template <int MODE, bool NORM>
int kp_launch_t(GemvParams p, int B, hipStream_t stream, int* grid_out, bool init_only) {
    if (init_only) {
        int r = kp_launch_tm<MODE, NORM, 1>(p, 0, false, true, stream);
        if (!r) r = kp_launch_tm<MODE, NORM, 3>(p, 0, false, true, stream);
        if (!r) r = kp_launch_tm<MODE, NORM, 6>(p, 0, false, true, stream);
        if constexpr (MODE == GEMV_RESID || MODE == GEMV_PLAIN) {
            if (!r) r = kp_launch_tm<MODE, NORM, 1, 11>(p, 0, false, true, stream);
        }
        // 33-64 rows: two halves of four waves, eight phases (K <= 4096)
        if constexpr (MODE == GEMV_QKV) { if (!r) r = kp_launch_tm<MODE, NORM, 3, 8, 2>(p, 0, false, true, stream); }
        if constexpr (MODE == GEMV_RESID) { if (!r) r = kp_launch_tm<MODE, NORM, 1, 8, 2>(p, 0, false, true, stream); }
        if constexpr (MODE == GEMV_GATEUP) { if (!r) r = kp_launch_tm<MODE, NORM, 6, 8, 2>(p, 0, false, true, stream); }
        return r;
    }
    if (B > 32) {
        // 33-64 rows (NH = 2): qkv (3 tiles per block), the o-proj (1), gate/up (6) with K within eight phases of a four-way split; the down projection and
        // the lm-head run as two launches of <= 32 rows (model.hip)
        const int ks2 = p.wscale ? 64 : 32;
        if (p.K % ks2 || p.K / ks2 < 4 || cdiv(p.K / ks2, 4) * ks2 > 8 * KP_PH * 32 || p.n_rows % 16 || p.attn_part) return -2;
        if (MODE == GEMV_QKV && (p.head_dim % 16 || p.head_dim < 16)) return -2;
        p.batch = B;
        p.n_groups = p.n_rows / 16;
        const int grid2 = p.n_groups < 256 ? p.n_groups : 256;
        const int tpb2 = cdiv(p.n_groups, grid2);
        if (grid_out) *grid_out = grid2;
        const bool r32b = MODE == GEMV_RESID && p.h32 != nullptr;
        if constexpr (MODE == GEMV_QKV) { if (tpb2 <= 3) return kp_launch_tm<MODE, NORM, 3, 8, 2>(p, grid2, r32b, false, stream); }
        if constexpr (MODE == GEMV_RESID) { if (tpb2 <= 1) return kp_launch_tm<MODE, NORM, 1, 8, 2>(p, grid2, r32b, false, stream); }
        if constexpr (MODE == GEMV_GATEUP) { if (tpb2 <= 6) return kp_launch_tm<MODE, NORM, 6, 8, 2>(p, grid2, r32b, false, stream); }
        return -2;
    }
    // K in whole load steps (32 elements; 64 with fp8 tiles), at least one per wave, a wave's share within eleven phases
    const int ks_el = p.wscale ? 64 : 32;
    if (p.K % ks_el || p.K / ks_el < KP_WAVES || cdiv(p.K / ks_el, KP_WAVES) * ks_el > 11 * KP_PH * 32 || p.n_rows % 16 || p.attn_part) return -2;
    if (MODE == GEMV_QKV && (p.head_dim % 16 || p.head_dim < 16)) return -2;
    p.batch = B;
    p.n_groups = p.n_rows / 16;   // tiles
    // one block per CU while that leaves at most 6 tiles per block; beyond (lm-head: 2008 tiles) more blocks of 6
    const int grid = cdiv(p.n_groups, 256) <= 6 ? (p.n_groups < 256 ? p.n_groups : 256) : cdiv(p.n_groups, 6);
    if (grid < 1 || (MODE == GEMV_LMHEAD && grid > p.max_parts)) return -2;
    if (grid_out) *grid_out = grid;
    const int tpb = cdiv(p.n_groups, grid);
    const bool r32 = MODE == GEMV_RESID && p.h32 != nullptr;
    if (cdiv(p.K / ks_el, KP_WAVES) * ks_el > 4 * KP_PH * 32) {   // the down projection: eleven phases, one tile per block
        if constexpr (MODE == GEMV_RESID || MODE == GEMV_PLAIN) {
            if (tpb <= 1) return kp_launch_tm<MODE, NORM, 1, 11>(p, grid, r32, false, stream);
        }
        return -2;
    }
    if (tpb <= 1) return kp_launch_tm<MODE, NORM, 1>(p, grid, r32, false, stream);
    if (tpb <= 3) return kp_launch_tm<MODE, NORM, 3>(p, grid, r32, false, stream);
    return kp_launch_tm<MODE, NORM, 6>(p, grid, r32, false, stream);
}

int kp_dispatch(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out, bool init_only) {
    switch (mode) {
        case GEMV_QKV: return kp_launch_t<GEMV_QKV, true>(p, B, stream, grid_out, init_only);
        case GEMV_RESID: return kp_launch_t<GEMV_RESID, false>(p, B, stream, grid_out, init_only);
        case GEMV_GATEUP: return kp_launch_t<GEMV_GATEUP, true>(p, B, stream, grid_out, init_only);
        case GEMV_LMHEAD: return kp_launch_t<GEMV_LMHEAD, true>(p, B, stream, grid_out, init_only);
        case GEMV_PLAIN: return kp_launch_t<GEMV_PLAIN, false>(p, B, stream, grid_out, init_only);
        default: return -2;
    }
}

}  // namespace

// raise the dynamic-LDS limit of every instantiation (call once, outside graph capture)
int decode_kmp_init() {
    static int done = -1;
    if (done == 0) return 0;
    GemvParams p = {};
    int r = 0;
    for (int mode = GEMV_QKV; mode <= GEMV_PLAIN && r == 0; ++mode) r = kp_dispatch(mode, p, 32, nullptr, nullptr, true);
    done = r == 0 ? 0 : -4;
    return done;
}

// p.W: the km copy of the matrix (launch_repack_km; fp8: decode_mfma.hip's e4m3 tiles of the permuted rows + p.wscale in the same row
// order).  -2: shape outside this kernel (K not in whole load steps, a wave's share beyond eleven phases, the o-proj's split merge)
int launch_decode_kmp(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    if (B < 17 || B > 64) return -2;
    if (decode_kmp_init() != 0) return -4;
    return kp_dispatch(mode, p, B, stream, grid_out, false);
}
