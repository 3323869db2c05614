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

template <int HD_, int NW_, int QB_, int NBUF_, int BPC_, int KSPL_ = 1, bool LAZY_ = true>
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
    // KSPL = 2 (round 5, the under-filled causal prefill): the block's waves form two KEY groups over the same NW / 2 query waves --
    // wave w and wave w + NW / 2 hold the same 32 queries, group g takes the 32-key half g of every 64-key tile, and the two
    // (m, l, O^T) states meet through LDS after the last tile.  One frame's causal prefill is 6 chunks x 32 heads = 192 blocks
    // on 256 CUs and lasts as long as its heaviest block (12 tiles behind one another at one wave per SIMD); with the key groups
    // that chain is half as long per wave and a SIMD holds two waves whose LDS / DMA waits cover each other.
    static constexpr int KSPL = KSPL_;
    static constexpr int NQW = NW / KSPL;               // query waves
    static constexpr int QCH = NQW * 32 * QB;           // queries per block
    static constexpr int NBUF = NBUF_;                  // LDS ring depth: NBUF - 1 tiles of DMA in flight
    static constexpr int KS = KP / 8, VS = VP / 8;      // 16-byte slots per LDS row
    static constexpr int NKI = (TILE * KS + 63) / 64, NVI = (TILE * VS + 63) / 64;   // 1 KiB DMA instructions per tile
    static constexpr bool GS = HD == 72 || (HD == 128 && QB == 2) || KSPL == 2;   // softmax step = one 32-key group instead of the 64-key tile (registers; KSPL: the wave's half)
    static_assert(KSPL == 1 || (KSPL == 2 && NW % 2 == 0 && QB == 1), "key groups: two, one query block per wave");
    static constexpr bool LZ = HD == 128 && QB == 1 && LAZY_;   // lazy reference maximum (round 5; the ViT shapes of this kernel are 4-5 tiles long: mostly first steps)
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

// The same with the LDS base left in M0 by the caller (`glds_base`) and the 1 KiB slab chosen by the instruction's IMMEDIATE offset,
// which the hardware adds to BOTH addresses (so the global address is pre-decremented by it).  Rewriting M0 between two LDS-DMA
// instructions serialises them on the first one's data return when the data comes from HBM (tools/dma_issue_rate.hip: 64
// instructions per CU land in 9.5 k clocks with M0 per instruction, in 2.1 k with one M0 per eight); eight slabs,
// immediates -4096 .. 3072, share one M0 = their base + 4096.
__device__ __forceinline__ void glds_base(unsigned int lds_base) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(__builtin_amdgcn_readfirstlane(lds_base)) : "m0", "memory");
}
template <int J>   // slab J of the eight
__device__ __forceinline__ void glds16_slab(const char* gsrc) {
    asm volatile("global_load_lds_dwordx4 %0, off offset:%1" ::"v"(gsrc - (J - 4) * 1024), "n"((J - 4) * 1024) : "memory");
}

// slab j (0..7) with j known only after unrolling: the immediate must be a literal, so the eight forms are spelled out
__device__ __forceinline__ void glds16_slab_j(int j, const char* gsrc) {
    switch (j) {
        case 0: glds16_slab<0>(gsrc); break;
        case 1: glds16_slab<1>(gsrc); break;
        case 2: glds16_slab<2>(gsrc); break;
        case 3: glds16_slab<3>(gsrc); break;
        case 4: glds16_slab<4>(gsrc); break;
        case 5: glds16_slab<5>(gsrc); break;
        case 6: glds16_slab<6>(gsrc); break;
        default: glds16_slab<7>(gsrc); break;
    }
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
    constexpr int KSPL = DM::KSPL, NQW = DM::NQW;
    constexpr bool LZ = DM::LZ;
    // one ring buffer = the K image (NKI slabs of 1 KiB), the V image (NVI slabs) and the surplus slabs of the last wave's share (see
    // dma_tile), contiguous: slab i of a tile sits 1024 i bytes into its buffer, so a wave's consecutive slabs share one M0
    constexpr int PTW = (NKI + NVI + NW - 1) / NW, BUFE = NW * PTW * 512;
    static_assert(TILE * KP * 2 == NKI * 1024 && TILE * VP * 2 == NVI * 1024, "the images are whole 1 KiB slabs");
    __shared__ __attribute__((aligned(1024))) bf16_t sKV[NBUF][BUFE];
    auto sK = [&](int buf) -> bf16_t* { return &sKV[buf][0]; };
    auto sV = [&](int buf) -> bf16_t* { return &sKV[buf][TILE * KP]; };
    // KSPL: key group 1 hands its state to group 0 here: [query wave][16 NDB + 2 registers][lane] fp32
    constexpr int MREG = 16 * DM::NDB + 2;
    __shared__ float sM[KSPL == 2 ? NQW * MREG * 64 : 1];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave: an SGPR
    const int ql = lane & 31, hi = lane >> 5;
    const int wq = KSPL == 2 ? (wave >= NQW ? wave - NQW : wave) : wave;    // query wave
    const int kg = KSPL == 2 ? (wave >= NQW ? 1 : 0) : 0;                   // key group
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
        for (int i = tid; i < NBUF * TILE * KZ; i += NT) *(u32x4_t*)(sK(i / (TILE * KZ)) + (i / KZ % TILE) * KP + HD + (i % KZ) * 8) = z;
        for (int i = tid; i < NBUF * TILE * VZ; i += NT) *(u32x4_t*)(sV(i / (TILE * VZ)) + (i / VZ % TILE) * VP + HD + (i % VZ) * 8) = z;
    }

    // Q^T fragments (B operand): lane (ql, hi) holds Q[q_row][16 kk + 8 hi .. +8] of each of its QB query blocks
    const int q_w0 = q_blk0 + wq * 32 * QB;
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

    // ---- tile staging by LDS-DMA: slab i of a tile fills the 64 slots 64 i .. 64 i + 63 of the row-major images (K: slabs 0 .. NKI - 1,
    // V: the NVI behind them); slot n is row n / KS, chunk n % KS; padding chunks are inactive lanes; rows past the sequence end re-read
    // its last row (finite, and masked where it matters), so every data slot of a tile is rewritten.  Wave w issues the PT CONSECUTIVE
    // slabs w PT .. w PT + PT - 1 of every tile (round 5; it had been slabs w, w + NW, ..): they lie within 8 KiB of each other, so one
    // M0 serves eight of them through the instruction's immediate offset -- rewriting M0 between two LDS-DMA instructions serialises
    // them on the first one's data return (tools/dma_issue_rate.hip) and the issuing wave stands still meanwhile.  The last wave's
    // surplus slabs (NW PT - TOT of them) land in spare LDS behind the images, so every wave issues exactly PT instructions per tile
    // and the waits can count: vmcnt(n PT) = "all but the n youngest tiles have landed".  Everything that does not depend on the tile
    // is computed once: per instruction the lane's row and its source byte offset at row 0. ----
    constexpr int TOT = NKI + NVI, PT = PTW;
    // per instruction: the lane's row inside the tile (-1: padding slot, the lane stays inactive) and its source byte offset
    // from `base` at row 0 (head column + chunk)
    const int k_col0 = p.k_off + hk * HD, v_col0 = p.v_off + hk * HD;   // locals: selecting between two FIELDS of the by-value
                                                                      // argument struct at run time made hipcc copy it to scratch
    auto slot_of = [&](int u, int& row, unsigned int& off) {
        const int idx = wave * PT + u;
        const bool isk = idx < NKI;
        const int n = (isk ? idx : idx - NKI) * 64 + lane;
        const int rk = n / KS, rv = n / VS;
        const int r = isk ? rk : rv, ch = isk ? n - rk * KS : n - rv * VS;
        row = (ch < CH && r < TILE) ? r : -1;
        off = (unsigned int)(((isk ? k_col0 : v_col0) + ch * 8) * 2);
        if (idx >= TOT) { row = 0; off = (unsigned int)(k_col0 * 2); }      // surplus slab: any readable 16 bytes, into the spare LDS
    };
    constexpr bool PRE = PT <= 12;            // kept in registers; beyond that (2-wave blocks: 19) hipcc indexes them from scratch
    unsigned int d_pk[PRE ? PT : 1];          // (row + 1) << 20 | off: one register per instruction (off < 2^20: a qkv row is a few KiB)
    if (PRE) {
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            int row;
            unsigned int off;
            slot_of(u, row, off);
            d_pk[PRE ? u : 0] = ((unsigned int)(row + 1) << 20) | off;
        }
    }
    const unsigned int ldsKV0 = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) void*)&sKV[0][0];
    const unsigned int row_bytes = (unsigned int)p.ld_qkv * 2u;
    auto dma_tile = [&](int kv0, int buf) {
        const unsigned int slab0 = ldsKV0 + buf * (BUFE * 2) + wave * (PT * 1024);
#pragma unroll
        for (int u = 0; u < PT; ++u) {
            if (u % 8 == 0) glds_base(slab0 + u * 1024 + 4096);
            int row;
            unsigned int off;
            if (PRE) { row = (int)(d_pk[PRE ? u : 0] >> 20) - 1; off = d_pk[PRE ? u : 0] & 0xFFFFFu; }
            else slot_of(u, row, off);
            const int krow = min(kv0 + row, len - 1);
            if (row >= 0) glds16_slab_j(u % 8, (const char*)base + ((size_t)(unsigned int)krow * row_bytes + off));
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
        // (round 5, KSPL form: requesting all K and V^T fragments of a step ahead of its MFMA chain -- pinned with sched_group_barrier, 212
        // registers -- was measured and not kept: 20.4 against 20.1 us; with two waves per SIMD the partner wave covers the LDS round trips)
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
            float alpha = 1.0f, mc;
            if constexpr (LZ) {
                // lazy reference maximum, as in the resident kernel below: m_run moves only when the step's maximum exceeds it by more
                // than 2^8 in the exp2 domain (always at a wave's first step), and the rescale of l and the 64 O^T registers runs only
                // when some row of the wave moved -- multiplied in place by asm, so the wave-uniform branch leaves hipcc nothing to copy
                const bool move = m_tile * c > m_run[qb] * c + 8.0f;   // finite m_tile: every step a wave runs holds a key each of its rows sees
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const float m_new = move ? m_tile : m_run[qb];
                    alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);   // 1 for the rows that stay
                    l_run[qb] *= alpha;
                    m_run[qb] = m_new;
#pragma unroll
                    for (int i = 0; i < NDB; ++i)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            // the FIRST multiply carries its own hazard guard: alpha comes from v_exp_f32, and hipcc's hazard recogniser does not look
                            // inside an asm statement (cdna_hip_programming.md 5.7) -- scheduled right behind the transcendental (round 6, once the
                            // packed softmax arithmetic had removed what used to sit between them) it read the register one wait state early: NaN in
                            // accumulator register 0 of some rows.  s_nop 1 = the two wait states of the trans-use hazard
                            if (i == 0 && r == 0) asm volatile("s_nop 1\n\tv_mul_f32 %0, %0, %1" : "+v"(o[qb][i][r]) : "v"(alpha));
                            else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(o[qb][i][r]) : "v"(alpha));
                        }
                }
                mc = m_run[qb] * c;
            } else {
                const float m_new = fmaxf(m_run[qb], m_tile);     // finite from the first step on: key 0 is visible to every query
                alpha = __builtin_amdgcn_exp2f((m_run[qb] - m_new) * c);
                mc = m_new * c;
                m_run[qb] = m_new;
            }
            // (round 6: the exp2 arguments and the row-sum adds as 2-vectors -- v_pk_fma_f32 / v_pk_add_f32, two scores per issue slot -- were built and
            // measured: the VALU count of a 32-key group drops from ~70 to ~54, the launches do not get faster (ViT B = 256 103.3 against 103.8 ms,
            // prefill equal; profiles/r06_attn_packed_ab.txt): the step is bound by its 16 quarter-rate v_exp_f32 and the MFMAs, not by issue slots)
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
            if constexpr (LZ) l_run[qb] += psum;
            else {
                l_run[qb] = __builtin_fmaf(l_run[qb], alpha, psum);
#pragma unroll
                for (int i = 0; i < NDB; ++i)
#pragma unroll
                    for (int r = 0; r < 16; ++r) o[qb][i][r] *= alpha;
            }
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
    // (KSPL: the tiles in which this wave's half -- keys 64 j + 32 kg .. + 32 -- starts below the wave's last visible key)
    const int ntile_w = !wave_live ? 0 : KSPL == 2 ? (kv_end_w > 32 * kg ? (kv_end_w - 32 * kg + TILE - 1) / TILE : 0) : (kv_end_w + TILE - 1) / TILE;
    constexpr int STEP = 32 * NGMAX;
    using NGfull = std::integral_constant<int, NGMAX>;
    using NGone = std::integral_constant<int, 1>;
    auto step_is_edge = [&](int ks0) { return (ks0 + STEP > len) || (p.causal && ks0 + STEP - 1 > q_w0); };
    auto steps_of = [&](int kv0) { return min(TILE / STEP, (kv_end_w - kv0 + STEP - 1) / STEP); };
    int j_plain = 0;   // leading tiles without a masked step
    if (KSPL == 2)
        while (j_plain < ntile_w && !step_is_edge(j_plain * TILE + kg * 32)) ++j_plain;
    else
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
        if (KSPL == 2) step(std::false_type{}, NGone{}, kv0 + kg * 32, sK(buf) + kg * 32 * KP, sV(buf) + kg * 32 * VP);
        else
        for (int sidx = 0; sidx < nst; ++sidx)
            step(std::false_type{}, NGfull{}, kv0 + sidx * STEP, sK(buf) + sidx * STEP * KP, sV(buf) + sidx * STEP * VP);
        tile_end(j);
    }
    for (; j < ntile_w; ++j) {
        ring(j);
        const int kv0 = j * TILE, buf = j % NBUF, nst = steps_of(kv0);
        if (KSPL == 2) step(std::true_type{}, NGone{}, kv0 + kg * 32, sK(buf) + kg * 32 * KP, sV(buf) + kg * 32 * VP);
        else
        for (int sidx = 0; sidx < nst; ++sidx) {
            const int ks0 = kv0 + sidx * STEP;
            if (NGMAX == 2 && kv_end_w - ks0 <= 32)   // a tail of <= 32 keys (DINOv2: 261 = 4 x 64 + 5; causal diagonal): half a step
                step(std::true_type{}, NGone{}, ks0, sK(buf) + sidx * STEP * KP, sV(buf) + sidx * STEP * VP);
            else
                step(std::true_type{}, NGfull{}, ks0, sK(buf) + sidx * STEP * KP, sV(buf) + sidx * STEP * VP);
        }
        tile_end(j);
    }
    for (; j < ntile; ++j) {
        ring(j);
        tile_end(j);
    }

    // ---- KSPL: the two key groups of a query wave meet.  Group 1 leaves (O^T, m, l) in LDS -- the tile images are dead: the last
    // tile_end was a barrier behind everybody's last read -- and group 0 folds it into its own state with the usual rescale.  A
    // group-1 wave that saw no key at all (the first 32 queries of a sequence) hands over m = -inf, l = 0, O = 0: weight exp2(-inf) = 0 ----
    if constexpr (KSPL == 2) {
        float* mine = sM + (wq * MREG) * 64 + lane;
        if (kg == 1) {
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) mine[(i * 16 + r) * 64] = o[0][i][r];
            mine[(16 * NDB) * 64] = m_run[0];
            mine[(16 * NDB + 1) * 64] = l_run[0];
        }
        __syncthreads();
        if (kg == 1) return;
        const float m1 = mine[(16 * NDB) * 64], l1 = mine[(16 * NDB + 1) * 64];
        const float m = fmaxf(m_run[0], m1);
        const float a0 = __builtin_amdgcn_exp2f((m_run[0] - m) * c), a1 = __builtin_amdgcn_exp2f((m1 - m) * c);
        l_run[0] = l_run[0] * a0 + l1 * a1;
#pragma unroll
        for (int i = 0; i < NDB; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[0][i][r] = o[0][i][r] * a0 + mine[(i * 16 + r) * 64] * a1;
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

// ---------------------------------------------------------------------------------------------------------------------
// Resident form for short non-causal sequences (the ViT towers: 261 / 256 tokens, thousands of (sequence, head) items).  One
// persistent block per CU: NW query waves (32 queries each) + NP producer waves; an item's WHOLE K and V live in LDS, double
// buffered.  While the query waves free-run over all keys of item i (no ring, no per-tile barrier), the producer waves request
// every slab of item i + 1 by LDS-DMA and the query waves' Q fragments of item i + 1 are in flight; one barrier per item.
// What the phase stamps (tools/attn_lab.hip, ATTN_LAB_TRACE) and micro-benchmarks behind this shape say:
//   * one item per block at a time with the load phase exposed (two blocks per CU, single buffered): 60 % of an item's time was
//     descriptor loads + DMA issue + HBM latency + two barriers;
//   * the vector-memory front end of a CU takes 50-100 clocks per row-strided 1 KiB request and a wave stalls while it issues:
//     requests issued by the query waves -- in one burst or one per softmax step -- add their issue time to the compute time,
//     issued by dedicated waves they do not (one producer wave alone is too slow: ~600 clocks per request against HBM, two are
//     enough);
//   * rewriting M0 between two LDS-DMA instructions costs (tools/dma_issue_rate.hip), so eight slabs share one M0 and are
//     selected by the immediate offset;
//   * MFMA and VALU instructions of one SIMD do NOT overlap, not even from different waves (tools/mfma_valu_overlap.hip: an
//     MFMA wave and a VALU wave on one SIMD take the sum of their times): a step costs its MFMA clocks PLUS its VALU clocks,
//     more waves per SIMD only hide LDS / memory latency.  Hence the lazy rescale and the read pipelining in the step below.
//   head_dim 64: 128-byte rows without padding, bank conflicts removed by an XOR swizzle of the 16-byte chunks applied to the
//                SOURCE address of the DMA (its LDS image is lane-linear): K chunk c of row r holds logical chunk
//                c ^ ((r >> 1) & 7) (16 rows of one ds_read_b128 pass -> 64 distinct banks), V chunk c holds
//                c ^ (((r >> 1) & 1) << 2) (the 4 key rows x 64 B of a transpose-read pass -> 64 distinct banks).
//   head_dim 72: K rows of 144 B (9 chunks: odd, conflict free), the k padding 72..79 of the QK^T reduction reads the next
//                row's first chunk against Q fragments that are ZERO there; V rows of 160 B (9 chunks + one zero chunk), the
//                third 32-row block of O^T reads 32 B into the next row: finite values feeding rows that are never stored.
// Rows an item does not have keep finite leftovers (LDS is zeroed once per block) and are masked in the last step.
template <int HD_, int NW_, int NP_>
struct ResCfg {
    static constexpr int HD = HD_, NW = NW_, NP = NP_, NT = (NW + NP) * 64, ROWS = NW * 32;   // NW query waves + NP producer waves
    static constexpr int HDK = (HD + 15) / 16 * 16, NKK = HDK / 16, NDB = (HD + 31) / 32, CH = HD / 8;
    static constexpr bool SWZ = HD == 64;
    static constexpr int KPB = SWZ ? 128 : HD * 2, VPB = SWZ ? 128 : 160;      // bytes per LDS row
    static constexpr int KCH = KPB / 16, VCH = VPB / 16;                        // 16-byte slots per row
    static constexpr int K_BYTES = ROWS * KPB, V_BYTES = ROWS * VPB, BUF = V_BYTES + K_BYTES;   // V first: its overshoot lands in K
    static constexpr int NKI = (ROWS * KCH + 63) / 64, NVI = (ROWS * VCH + 63) / 64, TOT = NKI + NVI, PT = (TOT + NW - 1) / NW;
    static constexpr int SMEM = 2 * BUF + 64;
    static constexpr int MINW = (NW + NP + 3) / 4;                              // one block per CU
    static_assert(SMEM <= 160 * 1024 && HD % 8 == 0 && (HD == 64 || HD == 72), "LDS budget; layouts above");
};

#ifdef ATTN_LAB_TRACE
__device__ unsigned long long g_attn_trace[16 * 12 * 8 * 8];   // [block L < 16][wave < 12][item < 8][stamp < 8]
#define ATTN_STAMP(k) do { if (L < 16 && tr_it < 8 && lane == 0) g_attn_trace[((L * 12 + wave) * 8 + tr_it) * 8 + (k)] = __builtin_readcyclecounter(); } while (0)
#else
#define ATTN_STAMP(k) do { } while (0)
#endif

template <class RC>
__global__ __launch_bounds__(RC::NT, RC::MINW) void emmax_attention_resident_kernel(AttnParams p, int per_xcd, int gx) {
    constexpr int HD = RC::HD, NW = RC::NW, NT = RC::NT, ROWS = RC::ROWS, NKK = RC::NKK, NDB = RC::NDB, CH = RC::CH;
    constexpr int KPB = RC::KPB, VPB = RC::VPB, KCH = RC::KCH, VCH = RC::VCH, NKI = RC::NKI, TOT = RC::TOT, PT = RC::PT;
    constexpr bool SWZ = RC::SWZ;
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ql = lane & 31, hi = lane >> 5, i16 = lane & 15, b4 = (lane >> 4) & 1;
    const int L = (int)blockIdx.x, xcd = L & 7, kb = L >> 3;      // block L runs on XCD L % 8; items in contiguous runs per XCD
    const int n_items = p.B * p.Hq;
#ifdef ATTN_LAB_TRACE
    int tr_it = 0;
#endif

    struct Item { int valid, h, start, len; };
    auto item_at = [&](int it) {
        Item w = {0, 0, 0, 0};
        const int item = xcd * per_xcd + it;
        if (it < per_xcd && item < n_items) {
            const int b = item / p.Hq;
            w.valid = 1;
            w.h = item - b * p.Hq;
            w.start = p.cu_seqlens[b];
            w.len = max(0, min(p.cu_seqlens[b + 1] - w.start, ROWS));
        }
        return w;
    };
    Item cur = item_at(kb);
    if (!cur.valid) return;

    for (int i = tid * 16; i < RC::SMEM; i += NT * 16) *(u32x4_t*)(smem + i) = (u32x4_t){0u, 0u, 0u, 0u};

    const float c = p.scale * 1.44269504088896340736f;
    const unsigned int lds0 = (unsigned int)(uintptr_t)(__attribute__((address_space(3))) void*)smem;
    const unsigned int row_bytes = (unsigned int)p.ld_qkv * 2u;
    // lane-constant fragment offsets (bytes)
    const int kx = SWZ ? ((hi ^ ((ql >> 1) & 7)) << 4) : (hi << 4);          // K chunk 2 kk + hi: SWZ -> kx ^ (kk << 5), else kx + (kk << 5)
    const int k_row = RC::V_BYTES + ql * KPB;
    const int v_row = (4 * hi + (i16 >> 2)) * VPB;                            // + key base * VPB; second read + 8 VPB
    // V chunk 4 db + 2 b4 + ((i16 & 3) >> 1), byte 8 (i16 & 1) inside it; SWZ flips the db bit on rows with (row >> 1) & 1
    const int v_in = (b4 << 5) + ((i16 & 3) << 3);
    const int v_flip = SWZ ? (((i16 >> 3) & 1) << 6) : 0;

    // requesting an item: its Q^T fragments (B operand: lane (ql, hi) holds Q[q_row][16 kk + 8 hi .. +8]; unconditional clamped
    // loads, masked by `mask_q` once they have landed -- masking here would make the compiler wait for them, i.e. for everything
    // requested behind them) by the query waves, its K / V slabs by the producer waves
    auto request_q = [&](const Item& w, bf16x8_t (&q)[NKK]) {
        if (w.len <= 0) return;
        const bf16_t* base = (const bf16_t*)p.qkv + (size_t)w.start * p.ld_qkv;
        const int q_row = wave * 32 + ql;
        const bf16_t* qp = base + p.q_off + w.h * HD + (size_t)min(q_row, w.len - 1) * p.ld_qkv;
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) q[kk] = __builtin_bit_cast(bf16x8_t, *(const u32x4_t*)(qp + min(kk * 16 + hi * 8, HD - 8)));
    };
    // DMA instructions 8 grp .. 8 grp + 7 of the K (isk) or V image of item w: instruction il fills the 64 slots 64 il ..; one M0
    // per group, the slab by immediate offset
    auto request_kv8 = [&](const Item& w, int buf, bool isk, int grp) {
        const int xch = isk ? KCH : VCH, ni = isk ? NKI : RC::NVI;
        const bf16_t* kvb = (const bf16_t*)p.qkv + (size_t)w.start * p.ld_qkv + (w.h / (p.Hq / p.Hkv)) * HD;
        // M0 is written once for the eight requests below, in separate asm statements (their per-lane predicates differ, so they
        // cannot be one statement as in gemm.hip): nothing between them may touch M0.  What sits between them is scalar / vector
        // address arithmetic only (no dynamic register indexing, no LDS-DMA builtin); the scheduling fences keep unrelated code
        // from being moved into the group.
        __builtin_amdgcn_sched_barrier(0);
        glds_base(lds0 + buf * RC::BUF + (isk ? RC::V_BYTES : 0) + grp * 8192 + 4096);
        auto one = [&](auto jtag) {
            constexpr int J = decltype(jtag)::value;
            const int il = grp * 8 + J;
            if (il >= ni || (il * 64) / xch >= w.len) return;          // past the image / the slab's first row does not exist (scalar)
            const int n = il * 64 + lane;
            const int row = n / xch, pc = n - row * xch;
            int lc = pc;
            if (SWZ) lc = isk ? pc ^ ((row >> 1) & 7) : pc ^ (((row >> 1) & 1) << 2);
            const unsigned int off = (unsigned int)(((isk ? p.k_off : p.v_off) + lc * 8) * 2);
            const int rg = min(row, w.len - 1);
            if (pc < CH && row < ROWS) glds16_slab<J>((const char*)kvb + ((size_t)(unsigned int)rg * row_bytes + off));
        };
        one(std::integral_constant<int, 0>{}); one(std::integral_constant<int, 1>{}); one(std::integral_constant<int, 2>{});
        one(std::integral_constant<int, 3>{}); one(std::integral_constant<int, 4>{}); one(std::integral_constant<int, 5>{});
        one(std::integral_constant<int, 6>{}); one(std::integral_constant<int, 7>{});
        __builtin_amdgcn_sched_barrier(0);
    };
    auto request_kv_all = [&](const Item& w, int buf) {   // producer waves: the groups of eight dealt round robin
        if (w.len <= 0) return;
        constexpr int GK = (NKI + 7) / 8, GV = (RC::NVI + 7) / 8;
        for (int gi = RC::NP ? wave - NW : wave; gi < GK + GV; gi += (RC::NP ? RC::NP : NW)) {   // no producer waves: everybody
            if (gi < GK) request_kv8(w, buf, true, gi);
            else request_kv8(w, buf, false, gi - GK);
        }
    };
    auto mask_q = [&](const Item& w, const bf16x8_t (&raw)[NKK], bf16x8_t (&q)[NKK]) {   // rows past the end and the k padding are zero
#pragma unroll
        for (int kk = 0; kk < NKK; ++kk) {
            u32x4_t v = __builtin_bit_cast(u32x4_t, raw[kk]);
            if (wave * 32 + ql >= w.len || kk * 16 + hi * 8 >= HD) v = (u32x4_t){0u, 0u, 0u, 0u};
            q[kk] = __builtin_bit_cast(bf16x8_t, v);
        }
    };
    bf16x8_t qf[NKK], qn[NKK];
#pragma unroll
    for (int kk = 0; kk < NKK; ++kk) qn[kk] = __builtin_bit_cast(bf16x8_t, (u32x4_t){0u, 0u, 0u, 0u});
    __syncthreads();                      // the zero fill is complete
    const bool producer = RC::NP > 0 && wave >= NW, issuer = RC::NP == 0 || producer;
    if (producer) __builtin_amdgcn_s_setprio(3);   // its few instructions go ahead of the query waves sharing its SIMD
    if (!producer) request_q(cur, qn);
    if (issuer) request_kv_all(cur, 0);
    Item nxt = item_at(kb + gx);
    int it2 = kb + 2 * gx;                // the item after the next: its descriptor loads are issued one item ahead, raw
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), visible to the compiler: Q and this wave's slabs have landed
    __syncthreads();                      // ... everybody's slabs
    mask_q(cur, qn, qf);
    int buf = 0;
    while (true) {
        ATTN_STAMP(0);
        if (nxt.valid && !producer) request_q(nxt, qn);   // lands while this item is computed
        if (RC::NP > 0) __syncthreads();                  // the Q requests are in the memory queue ahead of the slabs
        if (nxt.valid && issuer)                          // everybody left that buffer at the barrier that ended the last item
            request_kv_all(nxt, buf ^ 1);
        const int item2 = xcd * per_xcd + it2;
        const bool valid2 = it2 < per_xcd && item2 < n_items;
        const int b2 = valid2 ? item2 / p.Hq : 0;
        const int raw_lo = p.cu_seqlens[b2], raw_hi = p.cu_seqlens[b2 + 1];   // used after the barrier that ends this item
        ATTN_STAMP(1);
        const int len = cur.len;
        if (wave * 32 < len && !producer) {
            const unsigned char* sB = smem + buf * RC::BUF;
            f32x16_t o[NDB];
            float m_run = -INFINITY, l_run = 0.f;
#pragma unroll
            for (int i = 0; i < NDB; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[i][r] = 0.f;

            // One softmax step over the 32 keys of group g (the ring kernel's algebra with NG = 1), software-pipelined on the LDS
            // reads: MFMA and VALU instructions of one SIMD do not overlap (tools/mfma_valu_overlap.hip: an MFMA wave and a VALU
            // wave sharing a SIMD take the SUM of their times), so what is left to hide is LDS latency -- the V^T fragments of the
            // step are requested before its QK^T MFMAs, the K fragments of the NEXT step right behind them, and both land under
            // the softmax arithmetic.
            typedef __attribute__((address_space(3))) s16x4_t lds_s16x4;
            constexpr bool KPRE = NDB <= 2;   // head_dim 72 (three row blocks, 20 more registers of Q / K fragments) has no registers for it
            bf16x8_t kf[NKK];
            auto read_k = [&](int g) {
                const unsigned char* Kg = sB + g * 32 * KPB + k_row;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) kf[kk] = *(const bf16x8_t*)(Kg + (SWZ ? (kx ^ (kk << 5)) : (kx + (kk << 5))));
            };
            auto step = [&](auto edge_tag, int g, bool prefetch_k) {
                constexpr bool EDGE = decltype(edge_tag)::value;
                // V^T fragments of this step: two transpose reads each (head_dim 72: read right before their MFMAs instead)
                constexpr int VPRE = NDB <= 2 ? 2 : 0;
                if (!KPRE) read_k(g);
                s16x4_t va[2][NDB][2];
                auto read_v = [&](int mm) {
                    const unsigned char* vb = sB + (g * 32 + mm * 16) * VPB + v_row;
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const int vo = ((db << 6) ^ v_flip) + v_in;
                        va[mm][db][0] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + vo));
                        va[mm][db][1] = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(vb + 8 * VPB + vo));
                    }
                };
#pragma unroll
                for (int mm = 0; mm < VPRE; ++mm) read_v(mm);
                f32x16_t st;
#pragma unroll
                for (int kk = 0; kk < NKK; ++kk) {
                    const f32x16_t zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                    st = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[kk], qf[kk], kk == 0 ? zero : st, 0, 0, 0);
                }
                if (KPRE && prefetch_k) read_k(g + 1);
                if (EDGE) {
                    const int lim = len - g * 32 - 4 * hi;   // key of register r: 32 g + (r & 3) + 8 (r >> 2) + 4 hi
#pragma unroll
                    for (int r = 0; r < 16; ++r) st[r] = ((r & 3) + 8 * (r >> 2) < lim) ? st[r] : -INFINITY;
                }
                float m_tile = st[0];
#pragma unroll
                for (int r = 1; r < 16; ++r) m_tile = fmaxf(m_tile, st[r]);
                m_tile = half_max(m_tile);
                // LAZY reference maximum: m_run only moves when the step's maximum exceeds it by more than 2^LAZY in the exp2
                // domain (always at the first step: m_run = -inf).  P then lies in (0, 2^LAZY] instead of (0, 1] -- the same
                // relative precision in bf16, fp32 sums far from overflow -- and numerator and denominator use the same reference,
                // so the quotient is unchanged; what it buys: the rescale of the O^T accumulators (NDB x 16 multiplies, a fifth
                // of the step's VALU work) runs only when some row of the wave really moved (wave-uniform branch; not for
                // head_dim 72, where the two register copies of the accumulators the branch makes hipcc keep do not fit)
                constexpr float LAZY = 8.0f;
                const bool move = m_tile * c > m_run * c + LAZY;   // finite m_tile: key 0 is visible to every query
                if (__builtin_amdgcn_ballot_w64(move) != 0) {
                    const float m_new = move ? m_tile : m_run;
                    const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * c);   // 1 for the rows that stay
                    l_run *= alpha;
                    m_run = m_new;
                    if (NDB > 2) {
                        // head_dim 72 (168 registers at three waves per SIMD): as C++ the branch made hipcc keep a second copy of the 48
                        // accumulator registers and spill; multiplied IN PLACE by asm statements there is nothing to copy
#pragma unroll
                        for (int i = 0; i < NDB; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) {
                                if (i == 0 && r == 0) asm volatile("s_nop 1\n\tv_mul_f32 %0, %0, %1" : "+v"(o[i][r]) : "v"(alpha));   // trans-use hazard guard, see the ring kernel
                                else asm volatile("v_mul_f32 %0, %0, %1" : "+v"(o[i][r]) : "v"(alpha));
                            }
                    } else {
#pragma unroll
                        for (int i = 0; i < NDB; ++i)
#pragma unroll
                            for (int r = 0; r < 16; ++r) o[i][r] *= alpha;
                    }
                }
                const float mc = m_run * c;
                float psum = 0.f;
                u32x4_t pf[2];
#pragma unroll
                for (int mm = 0; mm < 2; ++mm)
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        const float p0 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[8 * mm + 2 * t], c, -mc));
                        const float p1 = __builtin_amdgcn_exp2f(__builtin_fmaf(st[8 * mm + 2 * t + 1], c, -mc));
                        psum += p0 + p1;
                        pf[mm][t] = pack_bf16x2(p0, p1);
                    }
                l_run += psum;
#pragma unroll
                for (int mm = 0; mm < 2; ++mm) {
                    if (mm >= VPRE) read_v(mm);
#pragma unroll
                    for (int db = 0; db < NDB; ++db) {
                        const u32x2_t w0 = __builtin_bit_cast(u32x2_t, va[mm][db][0]), w1 = __builtin_bit_cast(u32x2_t, va[mm][db][1]);
                        const u32x4_t av = {w0[0], w0[1], w1[0], w1[1]};
                        o[db] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, av), __builtin_bit_cast(bf16x8_t, pf[mm]), o[db], 0, 0, 0);
                    }
                }
            };
            const int n_full = len >> 5, n_steps = n_full + ((len & 31) ? 1 : 0);
#ifdef ATTN_LAB_KO_COMPUTE
            if (len < 0) step(std::true_type{}, 0, false);
#else
            if (KPRE) read_k(0);
            for (int g = 0; g < n_full; ++g) step(std::false_type{}, g, g + 1 < n_steps);
            if (len & 31) step(std::true_type{}, n_full, false);
#endif
            ATTN_STAMP(2);

            // ---- write O[q][d]: lane (ql, hi) holds d = 32 db + 8 a + 4 hi + (0..3) in o[db][4 a .. 4 a + 3] ----
            const float l_tot = half_sum(l_run);
            const int q_row = wave * 32 + ql;
            if (q_row < len) {
                const float inv = l_tot > 0.f ? 1.0f / l_tot : 0.f;
                bf16_t* orow = (bf16_t*)p.out + (size_t)(cur.start + q_row) * p.ld_out + cur.h * HD;
#pragma unroll
                for (int db = 0; db < NDB; ++db)
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int d0 = db * 32 + a * 8 + hi * 4;
                        if (d0 < HD) {
                            u32x2_t w;
                            w[0] = pack_bf16x2(o[db][4 * a] * inv, o[db][4 * a + 1] * inv);
                            w[1] = pack_bf16x2(o[db][4 * a + 2] * inv, o[db][4 * a + 3] * inv);
                            *(u32x2_t*)(orow + d0) = w;
                        }
                    }
            }
        }
        ATTN_STAMP(3);
        if (!nxt.valid) break;
        __builtin_amdgcn_s_waitcnt(0x0F70);   // the next item (and its Q) has landed ...
        ATTN_STAMP(4);
        __syncthreads();                      // ... for everybody, and everybody is done reading this one
        ATTN_STAMP(5);
#ifdef ATTN_LAB_TRACE
        ++tr_it;
#endif
        mask_q(nxt, qn, qf);
        cur = nxt;
        nxt.valid = valid2;
        nxt.h = item2 - b2 * p.Hq;
        nxt.start = raw_lo;
        nxt.len = max(0, min(raw_hi - raw_lo, ROWS));
        it2 += gx;
        buf ^= 1;
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

template <class RC>
static int launch_attention_resident(const AttnParams& p, hipStream_t stream) {
    auto kern = emmax_attention_resident_kernel<RC>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, RC::SMEM) != hipSuccess) return -4;
        attr_done = true;
    }
    const int per_xcd = cdiv(p.B * p.Hq, 8);               // (sequence, head) items per XCD
    const int gx = per_xcd < 32 ? per_xcd : 32;            // blocks per XCD: persistent, one per CU
    hipLaunchKernelGGL(kern, dim3(8 * gx), dim3(RC::NT), RC::SMEM, stream, p, per_xcd, gx);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int launch_attention(const AttnParams& p, int head_dim, hipStream_t stream) {
    if (p.B <= 0 || p.max_seqlen <= 0) return 0;
    if (p.Hq % p.Hkv != 0) return -1;
    if ((p.ld_qkv % 8) || (p.q_off % 8) || (p.k_off % 8) || (p.v_off % 8) || (p.ld_out % 4)) return -1;
    if (p.ld_qkv >= (1 << 19)) return -1;   // the ring kernel packs a lane's source byte offset inside a row into 20 bits (d_pk)
    const int force = emmax_tune().attn_resident;   // -1 default; 0 never; 2 = whenever it fits (tests)
    const bool no_resident = force == 0;
    // short non-causal sequences with enough (sequence, head) items to keep persistent blocks busy: the resident form (see
    // above).  Thresholds from tools/attn_lab.hip and the in-situ kernel trace: head_dim 64 wins from 32 frames on (190 vs 229 us
    // at 256 frames), head_dim 72 -- no room for the pipelined step -- only at the largest batches (174 vs 182 us in situ)
    const int items = p.B * p.Hq;
    if (!p.causal && !no_resident) {
        const bool big64 = force == 2 ? items >= 8 : items >= 512, big72 = force == 2 ? items >= 8 : items >= 2048;
        if (head_dim == 64 && big64 && p.max_seqlen <= 256) return launch_attention_resident<ResCfg<64, 8, 2>>(p, stream);
        if (head_dim == 64 && big64 && p.max_seqlen <= 288) return launch_attention_resident<ResCfg<64, 9, 2>>(p, stream);
        if (head_dim == 72 && big72 && p.max_seqlen <= 256) return launch_attention_resident<ResCfg<72, 8, 2>>(p, stream);
    }
    // Block shape: 4 waves (128 queries) unless 3 waves (96) waste fewer padded query rows -- DINOv2's 261 tokens are 3 x 96 - 27
    // against 3 x 128 - 123.  One 32-query block per wave and 2-4 light blocks per CU measured best on all three head sizes
    // (tools/attn_probe.py; two query blocks per wave halve the LDS fragment reads but need ~250 registers and lose to spills /
    // occupancy; deeper LDS rings and 8-wave blocks changed nothing: the loop is bound by its VALU softmax work, not by DMA latency).
    const bool three = cdiv(p.max_seqlen, 96) * 96 < cdiv(p.max_seqlen, 128) * 128;
    //                                            HD  NW QB NBUF blocks/CU
    switch (head_dim) {
        case 64: return three ? launch_attention_t<AttnCfg<64, 3, 1, 2, 4>>(p, stream) : launch_attention_t<AttnCfg<64, 4, 1, 2, 3>>(p, stream);
        case 72: return three ? launch_attention_t<AttnCfg<72, 3, 1, 2, 4>>(p, stream) : launch_attention_t<AttnCfg<72, 4, 1, 2, 3>>(p, stream);
        case 128: {
            // an under-filled causal launch (one frame's prefill: 6 chunks x 32 heads = 192 blocks) lasts as long as its heaviest block:
            // two key groups per block halve that chain (AttnCfg::KSPL; tuning switch attn_ksplit: -1 = when the grid leaves at most one
            // block per CU, 0 never, 1 every causal head_dim-128 launch).  Measured: profiles/r05_attn_ksplit_ab.txt
            const int sw = emmax_tune().attn_ksplit;
            const int blocks = 8 * cdiv(p.B * p.Hq, 8) * cdiv(p.max_seqlen, 128);
            if (p.causal && (sw == 1 || (sw < 0 && blocks <= 256))) return launch_attention_t<AttnCfg<128, 8, 1, 2, 1, 2>>(p, stream);
            // attn_lazy = 0 (tools/regress_bits.py): the running maximum of rounds 1-4 -- with attn_ksplit = 0 as well this launch reproduces the
            // round-4 kernel bit for bit
            if (emmax_tune().attn_lazy == 0) return launch_attention_t<AttnCfg<128, 4, 1, 2, 2, 1, false>>(p, stream);
            return launch_attention_t<AttnCfg<128, 4, 1, 2, 2>>(p, stream);
        }
        default: return -1;
    }
}
