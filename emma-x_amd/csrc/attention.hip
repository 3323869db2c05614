// attention.hip -- MFMA flash attention over a packed qkv buffer, used for
//   * the ViT blocks (bidirectional, N = 261 / 256, head_dim 64 / 72)   = timm `Attention` (F.scaled_dot_product_attention)
//   * the LLaMA prefill (causal, head_dim 128, GQA-generic)              = HF `LlamaAttention` at q_len > 1
// Both are called from prismatic/extern/hf/modeling_prismatic.py:121 and :404-415 in the reference.
//
// Structure (gfx950): a block = 3 or 4 waves = 96 / 128 queries of one (sequence, head); wave w owns 32 consecutive queries and
// keeps their Q^T, running softmax statistics and whole O^T accumulator in registers (the kernel template also builds with two
// query blocks per wave, 8-wave blocks and 3-4 deep LDS rings; none of them measured faster -- see launch_attention).  The ViT
// sequences are only 4-5 key tiles long, so a block is mostly prologue (Q / first tile from HBM) and epilogue: blocks are kept
// light -- 3-4 share a CU and one block's memory latency hides under another's math (8-wave blocks, one per CU, ran at 16 %
// MFMA utilisation).  The grid is 1-D and XCD-aware: the query chunks of a head, neighbouring heads and (with many sequences)
// all heads of a sequence run on ONE XCD, back to back, so the 128-byte lines that neighbouring heads share in the packed qkv rows (head pitch 128 / 144 B) and the K / V
// re-read by the other query chunks are served by that XCD's L2: HBM traffic measured = the qkv buffer once + the output
// (profiles/r02_pmc_attention.json).  Keys / values stream through LDS in tiles of 64, double buffered, by LDS-DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass): the DMA of tile j+1 is issued before the math of tile j
// and drained at the one barrier that ends the tile.  A wave-level DMA writes 64 consecutive 16-byte slots; the row padding
// slots are simply lanes that stay inactive (zeroed once).  Both images are ROW-major, exactly as they sit in the qkv buffer
// -- no transpose on the way in (bank conflicts measured: 0):
//   S^T = K . Q^T   v_mfma_f32_32x32x16_bf16, A = K rows via ds_read_b128 (odd 16-byte row pitch: conflict-free), B = Q^T from
//                   registers.  Lane (q = lane & 31, hi = lane >> 5) receives the scores of query q against keys
//                   crow(r, hi) = (r & 3) + 8 (r >> 2) + 4 hi, r = 0..15, of a 32-key group: a whole P row is lane-local up to
//                   ONE v_permlane32_swap (row max); the row sum stays per lane until the epilogue.
//   O^T = V^T . P^T the same MFMA; B = P^T = the lane's own score registers converted to bf16 (the 16-wide reduction index may
//                   be any bijection onto 16 keys as long as A uses the same one, so P never moves between lanes); A = V^T
//                   fetched from the row-major V image with ds_read_b64_tr_b16, the LDS transpose read: lane i of a 16-lane
//                   group supplies the address of 4 contiguous d of key i >> 2 and receives 4 consecutive keys of column i
//                   (map pinned on hardware by tools/tr_read_test.hip).  V row pitch = 64 B x odd: the four key rows of a read
//                   land in different bank quarters.
// Softmax in fp32 with the online (running max / running sum) recurrence, exp2 with the scale folded in; masked keys get -inf;
// mask code runs only on tiles that touch the sequence end or the causal diagonal.  Causal blocks stop at their diagonal and a
// wave skips the tiles past its own 32 queries.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

typedef __attribute__((ext_vector_type(16))) float f32x16_t;
typedef __attribute__((ext_vector_type(4))) short s16x4_t;

template <int HD_, int NW_, int QB_, int NBUF_, int BPC_>
struct AttnCfg {
    static constexpr int HD = HD_;
    static constexpr int HDK = (HD + 15) / 16 * 16;     // QK^T reduction length (k steps of 16), zero padded
    static constexpr int NKK = HDK / 16;
    static constexpr int NDB = (HD + 31) / 32;          // O^T row blocks of 32
    static constexpr int KP = HDK + 8;                  // K row pitch, bf16 elements: (2 KP) / 16 is odd for HDK in {64, 80, 128}
    static constexpr int VP = (NDB * 32 <= 96) ? 96 : 160;   // V row pitch: 192 B / 320 B = 64 B x odd, >= 2 * 32 * NDB bytes
    static constexpr int CH = HD / 8;                   // 16-byte chunks per K / V row
    static constexpr int TILE = 64;
    static constexpr int NW = NW_;                      // waves per block
    static constexpr int NT = NW * 64;
    static constexpr int QB = QB_;                      // 32-query blocks per wave
    static constexpr int QCH = NW * 32 * QB;            // queries per block
    static constexpr int NBUF = NBUF_;                  // LDS ring depth: NBUF - 1 tiles of DMA in flight
    static constexpr int KS = KP / 8, VS = VP / 8;      // 16-byte slots per LDS row
    static constexpr int NKI = (TILE * KS + 63) / 64, NVI = (TILE * VS + 63) / 64;   // 1 KiB DMA instructions per tile
    static constexpr bool GS = HD == 72 || (HD == 128 && QB == 2);   // softmax step = one 32-key group instead of the 64-key tile (registers)
    static constexpr int MINW = BPC_ * NW / 4;          // waves per SIMD the register budget is planned for (BPC blocks per CU)
};

// 16 bytes per lane, HBM/L2 -> LDS, no VGPR in between: lane i's data lands at lds_dst + 16 i (lds_dst wave-uniform).
// Issued as asm on purpose: for the builtin form hipcc tracks "an LDS-DMA write is in flight" and puts s_waitcnt vmcnt(0) in
// front of the next LDS read it cannot prove disjoint -- here the first V^T read of EVERY tile, which drained the tiles being
// prefetched into the other ring buffers.  The ring's ordering is carried by the counted waits + barriers in the kernel.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned int lds_dst) {
    const unsigned int m0v = __builtin_amdgcn_readfirstlane(lds_dst);
    asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(gsrc), "s"(m0v) : "m0", "memory");
}

__device__ __forceinline__ float half_max(float v) {   // max over the two half-waves (lanes l and l ^ 32)
    float a, b;
    swap_halves(v, a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float half_sum(float v) {
    float a, b;
    swap_halves(v, a, b);
    return a + b;
}

template <class DM>
__global__ __launch_bounds__(DM::NT, DM::MINW) void emmax_attention_kernel(AttnParams p, int nchunk, int per_xcd) {
    constexpr int HD = DM::HD, NBUF = DM::NBUF;
    constexpr int NKK = DM::NKK, NDB = DM::NDB, KP = DM::KP, VP = DM::VP, CH = DM::CH, TILE = DM::TILE, NT = DM::NT, NW = DM::NW;
    constexpr int QB = DM::QB, QCH = DM::QCH, KS = DM::KS, VS = DM::VS, NKI = DM::NKI, NVI = DM::NVI;
    constexpr bool GS = DM::GS;
    __shared__ __attribute__((aligned(16))) bf16_t sK[NBUF][TILE * KP];
    __shared__ __attribute__((aligned(16))) bf16_t sV[NBUF][TILE * VP];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int ql = lane & 31, hi = lane >> 5;
    // XCD-aware work map: linear block id L runs on XCD L % 8.  The (sequence, head) items are dealt out in contiguous runs of
    // `per_xcd`, so neighbouring heads of a sequence -- and, when there are many sequences, whole sequences -- stay on one XCD;
    // an item's query chunks occupy consecutive slots of that XCD, heavy (late) causal chunks first
    const int L = (int)blockIdx.x;
    const int slot = L >> 3;
    const int item = (L & 7) * per_xcd + slot / nchunk;
    if (item >= p.B * p.Hq) return;
    const int b = item / p.Hq, h = item - b * p.Hq;
    const int qc = nchunk - 1 - slot % nchunk;
    const int start = p.cu_seqlens[b];
    const int len = p.cu_seqlens[b + 1] - start;
    const int q_blk0 = qc * QCH;
    if (q_blk0 >= len) return;
    const int hk = h / (p.Hq / p.Hkv);
    const bf16_t* __restrict__ base = (const bf16_t*)p.qkv + (size_t)start * p.ld_qkv;
    const bf16_t* qptr = base + p.q_off + h * HD;
    const bf16_t* kptr = base + p.k_off + hk * HD;
    const bf16_t* vptr = base + p.v_off + hk * HD;

    // zero the padding once: K columns HD..HDK (HD = 72) enter the QK^T reduction; V columns past HD only feed discarded O^T rows
    // but must be finite.  The tile DMA never touches them (those lanes stay inactive), so nothing orders against it.
    {
        constexpr int KZ = (KP - HD) / 8, VZ = (VP - HD) / 8;    // 16-byte slots of padding per row (HD % 8 == 0)
        const u32x4_t z = {0u, 0u, 0u, 0u};
        for (int i = tid; i < NBUF * TILE * KZ; i += NT) *(u32x4_t*)(&sK[0][0] + (i / KZ) * KP + HD + (i % KZ) * 8) = z;
        for (int i = tid; i < NBUF * TILE * VZ; i += NT) *(u32x4_t*)(&sV[0][0] + (i / VZ) * VP + HD + (i % VZ) * 8) = z;
    }

    // Q^T fragments (B operand): lane (ql, hi) holds Q[q_row][16 kk + 8 hi .. +8] of each of its QB query blocks
    const int q_w0 = q_blk0 + wave * 32 * QB;
    const bool wave_live = q_w0 < len;
    bf16x8_t qf[QB][NKK];
#pragma unroll
    for (int qb = 0; qb < QB; ++qb)
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            const int q_row = q_w0 + qb * 32 + ql;
            const int k0 = kk * 16 + hi * 8;
            u32x4_t v = {0u, 0u, 0u, 0u};
            if (q_row < len && k0 < HD) v = *(const u32x4_t*)(qptr + (size_t)q_row * p.ld_qkv + k0);
            qf[qb][kk] = __builtin_bit_cast(bf16x8_t, v);
        }

    f32x16_t o[QB][NDB];
    float m_run[QB], l_run[QB];   // l_run: this lane's share of the row sum (its 16 keys of every 32-key group)
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        m_run[qb] = -INFINITY;
        l_run[qb] = 0.f;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[qb][i][r] = 0.f;
    }

    const int kv_end = p.causal ? min(len, q_blk0 + QCH) : len;                 // block
    const int kv_end_w = p.causal ? min(len, q_w0 + 32 * QB) : len;             // this wave
    const float c = p.scale * 1.44269504088896340736f;                          // exp(x * scale) = exp2(x * c)
    const int ntile = (kv_end + TILE - 1) / TILE;

    // ---- tile staging by LDS-DMA: instruction i of a tile fills the 64 slots 64 i .. 64 i + 63 of the row-major image; slot n is
    // row n / KS, chunk n % KS; padding chunks are inactive lanes; rows past the sequence end re-read its last row (finite, and
    // masked where it matters), so every data slot of a tile is rewritten.  Every wave issues exactly PT instructions per tile
    // (the surplus slots of the last round re-issue instructions 0.. of the same tile: same data to the same place), so the
    // waits can count: vmcnt(n PT) = "all but the n youngest tiles have landed".  Everything that does not depend on the tile is
    // computed once: per instruction the lane's row, its source pointer at row 0 and its LDS slab. ----
    constexpr int TOT = NKI + NVI, PT = (TOT + NW - 1) / NW;
    // per instruction: the lane's row inside the tile (-1: padding slot, the lane stays inactive) and its source byte offset
    // from `base` at row 0 (head column + chunk)
    const int k_col0 = p.k_off + hk * HD, v_col0 = p.v_off + hk * HD;   // locals: selecting between two FIELDS of the by-value
                                                                      // argument struct at run time made hipcc copy it to scratch
    auto slot_of = [&](int u, int& row, unsigned int& off) {
        int idx = wave + u * NW;
        idx = idx >= TOT ? idx - TOT : idx;
        const bool isk = idx < NKI;
        const int n = (isk ? idx : idx - NKI) * 64 + lane;
        const int rk = n / KS, rv = n / VS;
        const int r = isk ? rk : rv, ch = isk ? n - rk * KS : n - rv * VS;
        row = (ch < CH && r < TILE) ? r : -1;
        off = (unsigned int)(((isk ? k_col0 : v_col0) + ch * 8) * 2);
    };
    constexpr bool PRE = PT <= 12;            // kept in registers; beyond that (2-wave blocks: 19) hipcc indexes them from scratch
    int d_row[PRE ? PT : 1];
    unsigned int d_off[PRE ? PT : 1];
    if (PRE) {
#pragma unroll
        for (int u = 0; u < PT; ++u) slot_of(u, d_row[PRE ? u : 0], d_off[PRE ? u : 0]);
    }
    const unsigned int ldsK0 = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) void*)&sK[0][0];
    const unsigned int ldsV0 = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) void*)&sV[0][0];
    const unsigned int row_bytes = (unsigned int)p.ld_qkv * 2u;
    auto dma_tile = [&](int kv0, int buf) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            int idx = wave + u * NW;                       // scalar: which 1 KiB slab of which image
            idx = idx >= TOT ? idx - TOT : idx;
            const bool isk = idx < NKI;
            const unsigned int slab = isk ? ldsK0 + buf * (TILE * KP * 2) + idx * 1024 : ldsV0 + buf * (TILE * VP * 2) + (idx - NKI) * 1024;
            int row;
            unsigned int off;
            if (PRE) { row = d_row[PRE ? u : 0]; off = d_off[PRE ? u : 0]; }
            else slot_of(u, row, off);
            const int kg = min(kv0 + row, len - 1);
            if (row >= 0) glds16((const char*)base + ((size_t)(unsigned int)kg * row_bytes + off), slab);
        }
    };
    auto wait_all_but = [&](int younger) {   // wave-uniform
        if (younger <= 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
    };
    static_assert(NBUF >= 2 && NBUF <= 4 && 2 * PT <= 63, "ring depth / wait counter range");

#pragma unroll
    for (int t = 0; t < NBUF - 1; ++t)
        if (t < ntile) dma_tile(t * TILE, t);
    // a wait the COMPILER sees (vmcnt(0), other counters untouched): it now knows the Q loads have returned.  With only the asm
    // waits it keeps "Q may be pending" alive around the loop and emits its own vmcnt(0) before the first MFMA of every tile --
    // which, unknown to it, also drains the tiles the asm DMA is prefetching.
    __builtin_amdgcn_s_waitcnt(0x0F70);
    __syncthreads();      // tile 0 (and its successors) and the zero fill are in place

    // lane-constant LDS offsets
    const int k_off = ql * KP + hi * 8;                                        // K fragment: row ql of a 32-key group, + 16 kk
    const int i16 = lane & 15;
    const int v_off = (4 * hi + (i16 >> 2)) * VP + 16 * ((lane >> 4) & 1) + (i16 & 3) * 4;   // + key base * VP + 32 db

    // ---- one softmax step: NG groups of 32 keys starting at key ks0 (Kg / Vg = their rows in the LDS images).  EDGE steps
    // (sequence end / causal diagonal) mask; the others carry no mask code at all ----
    constexpr int NGMAX = GS ? 1 : 2;
    auto step = [&](auto edge_tag, auto ng_tag, int ks0, const bf16_t* __restrict__ Kg, const bf16_t* __restrict__ Vg) {
        constexpr bool EDGE = decltype(edge_tag)::value;
        constexpr int NG = decltype(ng_tag)::value;
        // S^T = K . Q^T ; one K fragment read feeds the MFMAs of all QB query blocks
        f32x16_t st[QB][NG];
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int kk = 0; kk < NKK; ++kk) {
                const bf16x8_t kf = *(const bf16x8_t*)(&Kg[gi * 32 * KP + k_off + kk * 16]);
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    st[qb][gi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[qb][kk], kk == 0 ? zero : st[qb][gi], 0, 0, 0);
                }
            }
        // online softmax per query block; P^T packed to bf16
        u32x4_t pf[QB][NG][2];
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            if (EDGE) {
                // key of register r: ks0 + 32 gi + (r & 3) + 8 (r >> 2) + 4 hi; visible iff below lim (sequence end, causal diagonal)
                const int q_row = q_w0 + qb * 32 + ql;
                const int lim = (p.causal ? min(len, q_row + 1) : len) - ks0 - 4 * hi;
#pragma unroll
                for (int gi = 0; gi < NG; ++gi)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        st[qb][gi][r] = (gi * 32 + (r & 3) + 8 * (r >> 2) < lim) ? st[qb][gi][r] : -INFINITY;
            }
            float m_tile = st[qb][0][0];
#pragma unroll
            for (int gi = 0; gi < NG; ++gi)
#pragma unroll
                for (int r = 0; r < 16; ++r) m_tile = fmaxf(m_tile, st[qb][gi][r]);
            m_tile = half_max(m_tile);
            const float m_new = fmaxf(m_run[qb], m_tile);     // finite from the first step on: key 0 is visible to every query
            const float alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
            const float mc = m_new * c;
            float psum = 0.f;
#pragma unroll
            for (int gi = 0; gi < NG; ++gi)
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qb][gi][8 * mm + 2 * t], c, -mc));
                        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[qb][gi][8 * mm + 2 * t + 1], c, -mc));
                        psum += p0 + p1;
                        pf[qb][gi][mm][t] = pack_bf16x2(p0, p1);
                    }
            l_run[qb] = __builtin_fmaf(l_run[qb], alpha, psum);
            m_run[qb] = m_new;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
        }
        // O^T += V^T . P^T : k steps of 16 keys; one V^T fragment (two transpose reads) feeds all QB query blocks
#pragma unroll
        for (int gi = 0; gi < NG; ++gi)
#pragma unroll
            for (int mm = 0; mm < 2; ++mm) {
                const bf16_t* vb = Vg + (gi * 32 + mm * 16) * VP + v_off;
#pragma unroll
                for (int db = 0; db < NDB; ++db) {
                    typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
                    const s16x4_t a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + db * 32));
                    const s16x4_t a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + 8 * VP + db * 32));
                    const u32x2_t w0 = __builtin_bit_cast(u32x2_t, a0), w1 = __builtin_bit_cast(u32x2_t, a1);
                    const u32x4_t av = {w0[0], w0[1], w1[0], w1[1]};
#pragma unroll
                    for (int qb = 0; qb < QB; ++qb)
                        o[qb][db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av),
                                                                            __builtin_bit_cast(bf16x8_t, pf[qb][gi][mm]), o[qb][db], 0, 0, 0);
                }
            }
    };

    // ---- tile loops.  A wave computes the tiles its own queries can see (all of them unless causal) and only keeps the ring and
    // the barriers going for the rest.  Steps that need the mask (sequence end, causal diagonal) are always the LAST ones of a
    // wave's key range, so the tiles are walked in three consecutive loops -- mask-free tiles, tiles with the mask, ring-only
    // tiles -- each with ONE call site of the step: with the masked and the unmasked step as the two arms of a branch inside one
    // loop, hipcc kept the O^T accumulators in different registers per arm and copied all of them (24-48 v_mov per step) ----
    const int ntile_w = wave_live ? (kv_end_w + TILE - 1) / TILE : 0;
    constexpr int STEP = 32 * NGMAX;
    using NGfull = std::integral_constant<int, NGMAX>;
    using NGone = std::integral_constant<int, 1>;
    auto step_is_edge = [&](int ks0) { return (ks0 + STEP > len) || (p.causal && ks0 + STEP - 1 > q_w0); };
    auto steps_of = [&](int kv0) { return min(TILE / STEP, (kv_end_w - kv0 + STEP - 1) / STEP); };
    int j_plain = 0;   // leading tiles without a masked step
    while (j_plain < ntile_w && !step_is_edge(j_plain * TILE + (steps_of(j_plain * TILE) - 1) * STEP)) ++j_plain;
    auto ring = [&](int j) {
        // tile j + NBUF - 1 goes into the buffer tile j - 1 was read from (everybody passed the barrier that ended it)
        if (j + NBUF - 1 < ntile) dma_tile((j + NBUF - 1) * TILE, (j + NBUF - 1) % NBUF);
    };
    auto tile_end = [&](int j) {
        wait_all_but(min(NBUF - 2, ntile - 2 - j));        // tile j + 1 has landed (younger ones may still fly) ...
        __syncthreads();                                    // ... and everybody is done reading tile j
    };
    int j = 0;
    for (; j < j_plain; ++j) {
        ring(j);
        const int kv0 = j * TILE, buf = j % NBUF, nst = steps_of(kv0);
        for (int sidx = 0; sidx < nst; ++sidx)
            step(std::false_type{}, NGfull{}, kv0 + sidx * STEP, &sK[buf][sidx * STEP * KP], &sV[buf][sidx * STEP * VP]);
        tile_end(j);
    }
    for (; j < ntile_w; ++j) {
        ring(j);
        const int kv0 = j * TILE, buf = j % NBUF, nst = steps_of(kv0);
        for (int sidx = 0; sidx < nst; ++sidx) {
            const int ks0 = kv0 + sidx * STEP;
            if (NGMAX == 2 && kv_end_w - ks0 <= 32)   // a tail of <= 32 keys (DINOv2: 261 = 4 x 64 + 5; causal diagonal): half a step
                step(std::true_type{}, NGone{}, ks0, &sK[buf][sidx * STEP * KP], &sV[buf][sidx * STEP * VP]);
            else
                step(std::true_type{}, NGfull{}, ks0, &sK[buf][sidx * STEP * KP], &sV[buf][sidx * STEP * VP]);
        }
        tile_end(j);
    }
    for (; j < ntile; ++j) {
        ring(j);
        tile_end(j);
    }

    // ---- write O[q][d]: lane (ql, hi) holds d = 32 db + 8 a + 4 hi + (0..3) in o[db][4 a .. 4 a + 3] ----
#pragma unroll
    for (int qb = 0; qb < QB; ++qb) {
        const float l_tot = half_sum(l_run[qb]);
        const int q_row = q_w0 + qb * 32 + ql;
        if (q_row < len) {
            const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
            bf16_t* orow = (bf16_t*)p.out + (size_t)(start + q_row) * p.ld_out + h * HD;
#pragma unroll
            for (int db = 0; db < NDB; ++db)
#pragma unroll
                for (int a = 0; a < 4; ++a) {
                    const int d0 = db * 32 + a * 8 + hi * 4;
                    if (d0 < HD) {
                        u32x2_t w;
                        w[0] = pack_bf16x2(o[qb][db][4 * a] * inv, o[qb][db][4 * a + 1] * inv);
                        w[1] = pack_bf16x2(o[qb][db][4 * a + 2] * inv, o[qb][db][4 * a + 3] * inv);
                        *(u32x2_t*)(orow + d0) = w;
                    }
                }
        }
    }
}

}  // namespace

template <class DM>
static int launch_attention_t(const AttnParams& p, hipStream_t stream) {
    const int nchunk = cdiv(p.max_seqlen, DM::QCH);
    const int per_xcd = cdiv(p.B * p.Hq, 8);     // (sequence, head) items per XCD
    const int grid = 8 * per_xcd * nchunk;
    hipLaunchKernelGGL(emmax_attention_kernel<DM>, dim3(grid), dim3(DM::NT), 0, stream, p, nchunk, per_xcd);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_attention(const AttnParams& p, int head_dim, hipStream_t stream) {
    if (p.B <= 0 || p.max_seqlen <= 0) return 0;
    if (p.Hq % p.Hkv != 0) return -1;
    if ((p.ld_qkv % 8) || (p.q_off % 8) || (p.k_off % 8) || (p.v_off % 8) || (p.ld_out % 4)) return -1;
    // Block shape: 4 waves (128 queries) unless 3 waves (96) waste fewer padded query rows -- DINOv2's 261 tokens are 3 x 96 - 27
    // against 3 x 128 - 123.  One 32-query block per wave and 2-4 light blocks per CU measured best on all three head sizes
    // (tools/attn_probe.py; two query blocks per wave halve the LDS fragment reads but need ~250 registers and lose to spills /
    // occupancy; deeper LDS rings and 8-wave blocks changed nothing: the loop is bound by its VALU softmax work, not by DMA latency).
    const bool three = cdiv(p.max_seqlen, 96) * 96 < cdiv(p.max_seqlen, 128) * 128;
    //                                            HD  NW QB NBUF blocks/CU
    switch (head_dim) {
        case 64: return three ? launch_attention_t<AttnCfg<64, 3, 1, 2, 4>>(p, stream) : launch_attention_t<AttnCfg<64, 4, 1, 2, 3>>(p, stream);
        case 72: return three ? launch_attention_t<AttnCfg<72, 3, 1, 2, 4>>(p, stream) : launch_attention_t<AttnCfg<72, 4, 1, 2, 3>>(p, stream);
        case 128: return launch_attention_t<AttnCfg<128, 4, 1, 2, 2>>(p, stream);
        default: return -1;
    }
}
