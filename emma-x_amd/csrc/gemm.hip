// gemm.hip -- bf16 MFMA GEMM with fused epilogues for the dense (MFMA-bound) stages of the hot path:
//   ViT qkv / proj / fc1 / fc2, patch-embed (im2col), projector fc1-3, LLaMA prefill qkv / o / gate-up / down.
// Replaces the torch `F.linear` calls made by timm `Attention`/`Mlp`, `PrismaticProjector.forward`
// (prismatic/extern/hf/modeling_prismatic.py:146-158) and HF `LlamaAttention`/`LlamaMLP` at q_len > 1.
//
//   C[M,N] = epi(A[M,K] . W[N,K]^T)    A, W bf16 row-major (K contiguous), fp32 accumulate.
//   epilogues: + bias, exact-erf GELU, x LayerScale, + residual, SwiGLU over 16-column (gate, up) groups, fp32 output.
//
// One kernel template, two tile geometries (Geom<WMW, WNW, MT, NT>: WMW x WNW waves, each MT x NT MFMA tiles of 16x16):
//   big    256x256x64, 8 waves as 2(M) x 4(N), wave tile 128x64 (128 accumulator registers), one block per CU, 128 KiB LDS
//   small  128x128x64, 4 waves as 2 x 2,      wave tile  64x64, two blocks per CU, 64 KiB LDS each
// `launch_gemm` plans with a measured cost model: all big, all small, whole rounds of big tiles followed by the remaining
// rows in small tiles (tile quantisation -- e.g. 1044 big tiles = 4.08 rounds -- is the main loss left in these GEMMs), a
// separate small-tile launch for a half-empty last tile column, or split-K (K slices per small tile into an fp32 scratch +
// a reduce / epilogue pass) when a long-K problem has fewer tiles than the chip has CUs.
//
// Persistent blocks walk a banded, XCD-aware tile order: block ids go round-robin over the 8 XCDs; each XCD owns a
// contiguous run of tiles, walked in bands of 4 tile columns (column fastest), so the blocks resident in one XCD form a
// compact patch that shares a few A and W panels in that XCD's 4 MiB L2 instead of streaming one A panel per block.
// Staging: HBM -> LDS directly with global_load_lds_dwordx4 (no staging registers, no ds_write pass), two LDS stages:
// the DMA of K step t+1 is issued before the MFMAs of K step t and drained (vmcnt(0)) at the single barrier that ends the
// step.  (A ring of four 32-deep stages with counted vmcnt was measured 15 % slower: twice the barriers for the same work.)
// Fragments: ds_read_b128 two row tiles ahead of the MFMAs that consume them (A through a 4-slot register ring, the B
// fragments of the second k half during the first), pinned with sched_group_barrier -- left alone, the scheduler sinks
// every read to just above its first use.  (With a global_load_lds in flight hipcc models the LDS counter as unordered and
// emits lgkmcnt(0) for every wait; inline-asm reads with hand-counted lgkmcnt(N) were tried and measured neutral, +-4 %:
// the loop is bound by LDS and L2->LDS bandwidth, not by read latency.)
// LDS image: operand tile = [rows][8 chunks of 16 B], rows 128 B apart.  A wave-level DMA writes 1 KiB = 8 whole rows in
// lane order, so the image itself is linear; the bank-conflict swizzle is applied to the SOURCE address instead: physical
// chunk c of row r holds logical chunk c ^ ((r >> 1) & 7).  A ds_read_b128 fragment read (16 lanes = 16 rows at one
// logical chunk) then touches 16 distinct 16-byte slots of the 256-byte bank row.
// MFMA operands are swapped (W fragment first): the accumulator of a lane then holds 4 CONSECUTIVE output columns of one
// output row.  bf16 results leave through a wave-private LDS window as whole rows (16 bytes per lane, 128 contiguous bytes
// per row): 2.4x the chip-wide store throughput of accumulator-layout stores, which had made the epilogue of a K = 1024
// tile a third of its time.  The next tile's first K step is requested before the epilogue, so its latency (and the block
// re-launch a one-tile-per-block grid would pay) hides under the stores.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

constexpr int BK = 64;
constexpr int BAND = 4;   // tile columns per band of the tile order

template <int WMW_, int WNW_, int MT_, int NT_>
struct Geom {
    static constexpr int WMW = WMW_, WNW = WNW_, MT = MT_, NT = NT_;
    static constexpr int NW = WMW * WNW;                    // waves per block
    static constexpr int BM = WMW * MT * 16, BN = WNW * NT * 16;
    static constexpr int OPA = BM * BK * 2, OPB = BN * BK * 2;   // operand tile bytes
    static constexpr int STAGE = OPA + OPB;
    static constexpr int SMEM = 2 * STAGE;                        // two stages of A and W
    static constexpr int SMEM_A3 = 2 * STAGE + OPA;              // three A stages + two W stages (big: all 160 KiB of the CU)
    static constexpr int SA = BM / 8 / NW, SB = BN / 8 / NW;      // 1 KiB DMA slabs (8 rows) per wave and operand
    static constexpr int WIN = 8192;                              // epilogue window per wave: 64 rows x 128 B
    static_assert(NT == 4 && MT % 4 == 0 && NT <= MT, "epilogue / fragment pipeline assume 64-column wave tiles");
    static_assert(BM % (8 * NW) == 0 && BN % (8 * NW) == 0 && NW * WIN <= STAGE, "DMA slabs / epilogue windows must fit");
};
using GeomBig = Geom<2, 4, 8, 4>;
using GeomSmall = Geom<2, 2, 4, 4>;

// LDS-DMA, 16 bytes per lane: lane i's data lands at M0 + immediate + 16 i.  The LDS base goes to M0 ONCE per group of
// slabs; the slab inside the group is chosen by the instruction's immediate offset, which the hardware adds to
// BOTH addresses.  The global address is a wave-uniform 64-bit base in SGPRs (tile origin + K offset, advanced by scalar adds)
// plus a per-lane 32-bit offset that does not change along K: no vector address arithmetic per request (as flat 64-bit
// pointers every request cost two v_lshl_add_u64 -- and VALU instructions are not free next to the MFMAs of the same SIMD).
// The lane offset carries the immediate's compensation and a bias DMA_BIAS >= 3072 so that it stays non-negative when an edge
// tile clamps its rows; the scalar base is lowered by the same bias.  Issued as asm: hipcc's builtin form rewrites M0 before
// every instruction and only takes flat pointers.
constexpr int DMA_BIAS = 4096;
// ONE asm statement per group: M0 is compiler-reserved and not preserved between statements, so the write of M0 and the four
// requests that read it must not be separable by anything hipcc might schedule in between
__device__ __forceinline__ void glds16_group4(unsigned int lds_base, unsigned int v0, unsigned int v1, unsigned int v2, unsigned int v3,
                                              unsigned long long sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %5 offset:0\n\t"
                 "global_load_lds_dwordx4 %2, %5 offset:1024\n\t"
                 "global_load_lds_dwordx4 %3, %5 offset:2048\n\t"
                 "global_load_lds_dwordx4 %4, %5 offset:3072"
                 ::"s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(v0), "v"(v1), "v"(v2), "v"(v3), "s"(sbase)
                 : "m0", "memory");
}

// bf16 epilogue through LDS: a wave parks 64 rows of its result (64 columns; SwiGLU: 32) in its private 8 KiB window
// (16-byte chunks XOR-swizzled with the row, so both the 8-byte writes in accumulator layout and the 16-byte reads in row
// layout are conflict-free) and stores them back as whole rows.
template <class G, int ACT, bool LN = false>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmParams& p, const f32x4_t (&acc)[G::MT][G::NT], int m0, int n0, int wm, int wn,
                                                     int g, int li, int lane, unsigned char* wl) {
    // the window addresses below depend on the lane only: left alone, hipcc computes them once per kernel, cannot keep them through the
    // 256-register main loop and spills them to scratch (with a vmcnt(0) per reload inside this epilogue).  Opaque copies of the
    // lane indices make them a handful of VALU instructions per tile instead.
    asm volatile("" : "+v"(lane), "+v"(li), "+v"(g));
    constexpr int WC = (ACT == 2) ? 32 : 64;   // output columns of this wave
    constexpr int CH = WC / 8;                  // 16-byte chunks per staged row
    constexpr int NG = WC / 16;                 // 16-column output groups
    const bf16_t* bias = (const bf16_t*)p.bias;
    const bf16_t* scale = (const bf16_t*)p.scale;
    const bf16_t* res = (const bf16_t*)p.residual;
    const int n_ok = (ACT == 2) ? p.N / 2 : min(p.N, p.N_store);
    const int ocol0 = (ACT == 2) ? ((n0 + wn * 64) >> 1) : (n0 + wn * 64);
    float bv[NG][4], sv[NG][4], cs[LN ? NG : 1][4];   // LN: bv = ln_c (fp32), cs = ln_s
    // a lane's four columns of a group are consecutive: one 8- / 16-byte load per operand and group (were four 2- / 4-byte loads)
    // (LN form: ln_c / ln_s take 16-byte loads -- the model's arena keeps them aligned, emmax_op_gemm_ln takes caller workspaces: ADVICE r04)
    const bool vec_col = (n_ok & 3) == 0 && (((uintptr_t)bias | (uintptr_t)scale) & 7) == 0 &&
                         (!LN || ((((uintptr_t)p.ln_c | (uintptr_t)p.ln_s) & 15) == 0));
#pragma unroll
    for (int jo = 0; jo < NG; ++jo) {
        const int col4 = ocol0 + jo * 16 + g * 4;
        if (vec_col) {
            const int cc = min(col4, n_ok - 4);            // (columns past the edge: clamped, never stored)
            if constexpr (LN) {
                const f32x4_t c4 = *(const f32x4_t*)(p.ln_c + cc), s4 = *(const f32x4_t*)(p.ln_s + cc);
#pragma unroll
                for (int r = 0; r < 4; ++r) { bv[jo][r] = c4[r]; cs[jo][r] = s4[r]; sv[jo][r] = 1.f; }
            } else {
                u32x2_t b2 = {0u, 0u}, s2 = {0x3f803f80u, 0x3f803f80u};   // bf16 1.0 pairs
                if (ACT != 2 && bias) b2 = *(const u32x2_t*)(bias + cc);
                if (ACT != 2 && scale) s2 = *(const u32x2_t*)(scale + cc);
                bv[jo][0] = bf_lo(b2[0]); bv[jo][1] = bf_hi(b2[0]); bv[jo][2] = bf_lo(b2[1]); bv[jo][3] = bf_hi(b2[1]);
                sv[jo][0] = bf_lo(s2[0]); sv[jo][1] = bf_hi(s2[0]); sv[jo][2] = bf_lo(s2[1]); sv[jo][3] = bf_hi(s2[1]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int col = col4 + r;
                if constexpr (LN) {
                    bv[jo][r] = col < n_ok ? p.ln_c[col] : 0.f;
                    cs[jo][r] = col < n_ok ? p.ln_s[col] : 0.f;
                } else {
                    bv[jo][r] = (ACT != 2 && bias && col < n_ok) ? bf2f(bias[col]) : 0.f;
                }
                sv[jo][r] = (!LN && ACT != 2 && scale && col < n_ok) ? bf2f(scale[col]) : 1.f;   // (LN form: no LayerScale, checked by the launcher)
            }
        }
    }
    // ---- the rows.  The residual is always the vectorised form here (8 bytes per lane and column group, requested ahead; any other
    // residual layout takes the direct epilogue): with a second, element-wise form chosen per (row tile, column group) hipcc had
    // interleaved both behind 64 branches per tile.  Rows / columns of an edge tile that lie outside C: requests clamped, stores
    // masked -- and the window reads unconditional AHEAD of the stores: inside the row check they were serialised (ds_read,
    // lgkmcnt(0), store: eight LDS round trips per half).
    const bool has_res = !LN && ACT != 2 && res != nullptr;   // (8-byte requests possible: staged_ok in the kernel; LN form: no residual, checked by the launcher)
    {
        constexpr int NH = G::MT / 4;
        u32x2_t rres[4][NG];       // residual: one 64-row half at a time (both halves at once: 64 registers, spilled)
        f32x2_t stv[G::MT];   // LN: (mean, rstd) of every output row of the wave, all requested up front (8 bytes per row)
        // residual rows of a 64-row half are requested TOGETHER, ahead of the arithmetic (loaded where they are used, each of the 16
        // residual loads of a half was followed by its own vmcnt(0)), and the SECOND half's as soon as the arithmetic of the first
        // has consumed the registers, i.e. BEFORE the first half's window reads and stores: requested behind the stores, their wait
        // also waited for the stores' acknowledgements (the counter is in order)
        auto request_rows = [&](int h, int ii, u32x2_t (&rr)[NG]) {
            const int rowc = min(m0 + wm * (G::MT * 16) + (4 * h + ii) * 16 + li, p.M - 1);   // (rows / columns past the edge: clamped, never stored)
            if (has_res) {   // (rres is only ever read under the same condition)
#pragma unroll
                for (int jo = 0; jo < NG; ++jo) rr[jo] = *(const u32x2_t*)(res + (size_t)rowc * p.ldr + min(ocol0 + jo * 16 + g * 4, n_ok - 4));
            }
        };
#pragma unroll
        for (int i = 0; i < G::MT; ++i) {
            if constexpr (LN) stv[i] = *(const f32x2_t*)(p.ln_stats + (size_t)min(m0 + wm * (G::MT * 16) + i * 16 + li, p.M - 1) * 2);
            else stv[i] = (f32x2_t){0.f, 1.f};
        }
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) request_rows(0, ii, rres[ii]);
#pragma unroll
        for (int h = 0; h < NH; ++h) {
#pragma unroll
            for (int ii = 0; ii < 4; ++ii) {
                const int i = 4 * h + ii, rl = ii * 16 + li;          // row inside the 64-row window
                const int row = m0 + wm * (G::MT * 16) + i * 16 + li;
                const f32x2_t st = stv[i];                             // LN: (mean, rstd) of this output row
#pragma unroll
                for (int jo = 0; jo < NG; ++jo) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        if (ACT == 2) {
                            v[r] = silu(acc[i][2 * jo][r]) * acc[i][2 * jo + 1][r];
                        } else {
                            if constexpr (LN) v[r] = (acc[i][jo][r] - st[0] * cs[jo][r]) * st[1] + bv[jo][r];
                            else v[r] = acc[i][jo][r] + bv[jo][r];
                        }
                    }
                    if (ACT == 1) {   // GELU two values per packed-fp32 instruction
                        const f32x2_t g0 = gelu_erf2((f32x2_t){v[0], v[1]}), g1 = gelu_erf2((f32x2_t){v[2], v[3]});
                        v[0] = g0[0]; v[1] = g0[1]; v[2] = g1[0]; v[3] = g1[1];
                    }
                    if constexpr (!LN) {
                        if (ACT != 2) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) v[r] *= sv[jo][r];
                        }
                    }
                    if (has_res) {
                        const u32x2_t rv = rres[ii][jo];
                        v[0] += bf_lo(rv[0]); v[1] += bf_hi(rv[0]); v[2] += bf_lo(rv[1]); v[3] += bf_hi(rv[1]);
                    }
                    const int c = jo * 2 + (g >> 1);                  // 16-byte chunk of the row, half g & 1
                    *(u32x2_t*)(wl + rl * (WC * 2) + ((c ^ (rl & (CH - 1))) << 4) + (g & 1) * 8) =
                        (u32x2_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                }
            }
            if (h + 1 < NH) {   // the next half's residual rows: the registers are free, and the request stays AHEAD of this half's stores
#pragma unroll
                for (int ii = 0; ii < 4; ++ii) request_rows(h + 1, ii, rres[ii]);
            }
            // whole rows back out: lane -> (row, chunk); LDS executes a wave's accesses in order, no barrier needed
            u32x4_t val[CH];
#pragma unroll
            for (int itr = 0; itr < CH; ++itr) {
                const int idx = itr * 64 + lane;
                const int rr = idx / CH, cl = idx % CH;
                val[itr] = *(const u32x4_t*)(wl + rr * (WC * 2) + ((cl ^ (rr & (CH - 1))) << 4));
            }
#pragma unroll
            for (int itr = 0; itr < CH; ++itr) {
                const int idx = itr * 64 + lane;
                const int rr = idx / CH, cl = idx % CH;
                const int row = m0 + wm * (G::MT * 16) + h * 64 + rr, col = ocol0 + cl * 8;
                bf16_t* dst = (bf16_t*)p.C + (size_t)row * p.ldc + col;
                // (the launcher routes a column count that is not a multiple of 8 to the direct epilogue)
                if (row < p.M && col + 7 < n_ok) *(u32x4_t*)dst = val[itr];
            }
        }
    }
}

// direct epilogue (fp32 output, or a C whose rows are not 16-byte aligned): 8 / 16 bytes per lane in accumulator layout
template <class G, int ACT, bool OUT_F32>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, const f32x4_t (&acc)[G::MT][G::NT], int m0, int n0, int wm, int wn, int g, int li,
                                              size_t c_off = 0) {   // c_off: element offset into C (split-K partial slab)
    // ---- epilogue.  Swapped-operand C/D layout: lane (g, li) holds output row m = li of the row tile and the four
    // consecutive columns n = 4 g + r (r = register) of the column tile ----
    const bf16_t* bias = (const bf16_t*)p.bias;
    const bf16_t* scale = (const bf16_t*)p.scale;
    const bf16_t* res = (const bf16_t*)p.residual;
    const int n_ok = min(p.N, p.N_store);
    if (ACT == 2) {
        // SwiGLU: 16-column groups alternate (gate, up); output column = (col/32)*16 + col%16.
        const bool vec = (p.ldc & 3) == 0;
#pragma unroll
        for (int i = 0; i < G::MT; ++i) {
            const int row = m0 + wm * (G::MT * 16) + i * 16 + li;
#pragma unroll
            for (int j = 0; j < 4; j += 2) {
                const int col = n0 + wn * 64 + j * 16 + g * 4;
                const int ocol = (col >> 5) * 16 + (col & 15);
                if (row < p.M && col < p.N) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = (OUT_F32 ? silu_precise(acc[i][j][r]) : silu(acc[i][j][r])) * acc[i][j + 1][r];
                    if (OUT_F32) {   // exact numerics: the product stays fp32 (split into two bf16 terms by the consumer's pass)
                        float* d32 = (float*)p.C + (size_t)row * p.ldc + ocol;
                        if (vec) *(f32x4_t*)d32 = (f32x4_t){v[0], v[1], v[2], v[3]};
                        else {
#pragma unroll
                            for (int r = 0; r < 4; ++r) d32[r] = v[r];
                        }
                        continue;
                    }
                    bf16_t* dst = (bf16_t*)p.C + (size_t)row * p.ldc + ocol;
                    if (vec) {
                        *(u32x2_t*)dst = (u32x2_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r) dst[r] = f2bf(v[r]);
                    }
                }
            }
        }
        return;
    }
    const bool vec_c = (p.ldc & 3) == 0, vec_r = (p.ldr & 3) == 0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int col = n0 + wn * 64 + j * 16 + g * 4;
        float bv[4], sv[4], cs[4];
        const bool ln = p.ln_stats != nullptr;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            bv[r] = (col + r < n_ok) ? (ln ? p.ln_c[col + r] : bias ? bf2f(bias[col + r]) : 0.f) : 0.f;
            cs[r] = (ln && col + r < n_ok) ? p.ln_s[col + r] : 0.f;
            sv[r] = (scale && col + r < n_ok) ? bf2f(scale[col + r]) : 1.f;
        }
        const bool full = col + 3 < n_ok;
#pragma unroll
        for (int i = 0; i < G::MT; ++i) {
            const int row = m0 + wm * (G::MT * 16) + i * 16 + li;
            if (row >= p.M || col >= n_ok) continue;
            f32x2_t st = {0.f, 1.f};
            if (ln) st = *(const f32x2_t*)(p.ln_stats + (size_t)row * 2);
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[r] = (acc[i][j][r] - st[0] * cs[r]) * st[1] + bv[r];
                if (ACT == 1) v[r] = gelu_erf(v[r]);   // (2.8e-7 absolute: also what the exact-numerics path uses)
                v[r] *= sv[r];
            }
            if (OUT_F32 && res && p.res_f32) {   // fp32 residual stream (C may alias it)
                const float* rq = (const float*)p.residual + (size_t)row * p.ldr + col;
                if (full && vec_r) {
                    const f32x4_t rv = *(const f32x4_t*)rq;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] += rv[r];
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < n_ok) v[r] += rq[r];
                }
            } else if (res) {
                const bf16_t* rp = res + (size_t)row * p.ldr + col;
                if (full && vec_r) {
                    const u32x2_t rv = *(const u32x2_t*)rp;
                    v[0] += bf_lo(rv[0]); v[1] += bf_hi(rv[0]); v[2] += bf_lo(rv[1]); v[3] += bf_hi(rv[1]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < n_ok) v[r] += bf2f(rp[r]);
                }
            }
            if (OUT_F32) {
                float* dst = (float*)p.C + c_off + (size_t)row * p.ldc + col;
                if (full && vec_c) {
                    *(f32x4_t*)dst = (f32x4_t){v[0], v[1], v[2], v[3]};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < n_ok) dst[r] = v[r];
                }
            } else {
                bf16_t* dst = (bf16_t*)p.C + (size_t)row * p.ldc + col;
                if (full && vec_c) {
                    *(u32x2_t*)dst = (u32x2_t){pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3])};
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (col + r < n_ok) dst[r] = f2bf(v[r]);
                }
            }
        }
    }
}

// DEEP = 1 / 2: THREE stages of the A / of the W operand, two of the other (LDS image [A0 | W0 | A1 | A2 | W1] / [A0 | W0 | A1 | W1 | W2]:
// all 160 KiB of the CU for the 256 x 256 tile).  One operand of these GEMMs streams from HBM / MALL once per band of tiles -- the
// activations of a ViT batch (137 MB against 2-9 MB of weights), the weights of a LLaMA prefill (100-180 MB against 6-50 MB of rows)
// -- while the panels of the other are re-read by every tile of the band and sit in the XCD's L2.  The streaming operand's request
// of K step t + 2 goes out during step t and is waited for by COUNT (vmcnt(4): its four slab requests stay in flight across the
// step barrier, a bare s_barrier); the resident operand keeps one step of lead.  Measured (round 4, interleaved A/B on one box):
// ViT shapes +3 .. 8 % with DEEP = 1 and -3 % with it on the LLaMA shapes, where the weights are the stream.
// Two waves per SIMD in both geometries (one 8-wave block or two 4-wave blocks per CU), stated to the compiler: left to the
// block size alone, the 4-wave kernels may take up to 512 registers, and the LN + GELU instantiation took 260 -- ONE block per CU.
template <class G, int ACT, bool OUT_F32, bool LN = false, int DEEP = 0>
__global__ __launch_bounds__(G::NW * 64, 2) void emmax_gemm_bf16_kernel(GemmParams p) {
    constexpr int BM = G::BM, BN = G::BN, MT = G::MT, NT = G::NT, STAGE_BYTES = G::STAGE, OP_BYTES = G::OPA;
    static_assert(G::OPA == G::OPB, "the stage slots of A and W are interchangeable");
    constexpr int NSA = DEEP == 1 ? 3 : 2, NSW = DEEP >= 2 ? 3 : 2;
    constexpr bool PP = DEEP == 3;   // staggered wave groups (below)
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    // byte offsets of the stages: A slot 0 and W slot 0 form "stage 0", everything behind them (64 KiB at least) is free while a
    // tile's epilogue runs (the next tile's K step 0 is already in flight into stage 0)
    auto a_slot = [](int i) { return i == 0 ? 0 : STAGE_BYTES + (i - 1) * OP_BYTES; };
    auto w_slot = [](int j) { return j == 0 ? OP_BYTES : STAGE_BYTES + (NSA - 1 + j - 1) * OP_BYTES; };

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / G::WNW, wn = wave % G::WNW;
    const int g = lane >> 4, li = lane & 15;

    // ---- persistent tile loop.  Block b lives on XCD b & 7 (round-robin dispatch); that XCD owns a contiguous run of the
    // banded tile order and its gridDim/8 blocks walk it side by side ----
    // split-K: the work units are (K slice, tile), slice-major, so concurrently resident blocks work on the same K range of
    // neighbouring tiles (shared panels) and the slices of one tile never meet inside this kernel
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int ntile = tiles_m * tiles_n;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int nwg = ntile * ks;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int rq = nwg >> 3, rr = nwg & 7;
    const int run0 = xcd < rr ? xcd * (rq + 1) : rr * (rq + 1) + (xcd - rr) * rq, run_n = rq + (xcd < rr ? 1 : 0);
    const int nk_all = (p.dbg & 4) ? 1 : p.K / BK;   // dbg 4: one K step per tile (store path alone)
    const int nk_slice = (nk_all + ks - 1) / ks;
    int kidx = 0, kbeg = 0, nk = nk_all;            // K slice of the current unit: steps kbeg .. kbeg + nk
    auto tile_origin = [&](int it, int& m0, int& n0) {
        const int unit = run0 + it;
        kidx = unit / ntile;
        kbeg = kidx * nk_slice;
        nk = min(nk_slice, nk_all - kbeg);
        const int bid = unit - kidx * ntile;
        const int band = bid / (BAND * tiles_m), in_band = bid - band * (BAND * tiles_m);
        const int bw = min(BAND, tiles_n - band * BAND);
        const int tm = in_band / bw, tn = band * BAND + (in_band - tm * bw);
        m0 = tm * BM;
        n0 = tn * BN;
    };

    // ---- DMA sources: wave w fills 1 KiB slabs (8 rows each) SA*w .. of the A tile and SB*w .. of the W tile ----
    unsigned int offA_l[G::SA], offB_l[G::SB];        // per lane: (row - tile origin) * ld * 2 + swizzled chunk - slab immediate + bias
    unsigned long long baseA = 0, baseB = 0;          // wave-uniform: tile origin - bias (K offset added per step)
    const unsigned int lds0 = __builtin_amdgcn_readfirstlane((unsigned int)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
    auto set_sources = [&](int m0, int n0) {
        // logical chunk landing in physical chunk lane&7: (row >> 1) & 7 with row = 8 * slab + (lane >> 3)
        baseA = (unsigned long long)(uintptr_t)((const bf16_t*)p.A + (size_t)m0 * p.lda) - DMA_BIAS;
        baseB = (unsigned long long)(uintptr_t)((const bf16_t*)p.W + (size_t)n0 * p.ldw) - DMA_BIAS;
#pragma unroll
        for (int j = 0; j < G::SA; ++j) {
            const int row = (wave * G::SA + j) * 8 + (lane >> 3);
            const int gm = min(m0 + row, p.M - 1) - m0;   // edge tiles re-read the last row (never stored)
            offA_l[j] = (unsigned int)gm * (unsigned int)(p.lda * 2) + ((lane & 7) ^ ((row >> 1) & 7)) * 16 + (DMA_BIAS - j * 1024);
        }
#pragma unroll
        for (int j = 0; j < G::SB; ++j) {
            const int row = (wave * G::SB + j) * 8 + (lane >> 3);
            const int gn = min(n0 + row, p.N - 1) - n0;
            offB_l[j] = (unsigned int)gn * (unsigned int)(p.ldw * 2) + ((lane & 7) ^ ((row >> 1) & 7)) * 16 + (DMA_BIAS - j * 1024);
        }
    };
    static_assert(G::SA == 4 && G::SB == 4, "a wave's slabs of one operand share one M0 (immediates 0 .. 3072)");
    // (round 5: the non-temporal hint on these requests -- `global_load_lds_dwordx4 ... nt` on the A or on the W operand -- was measured and
    // not kept: ViT shapes +-1 %, LLaMA shapes -3 .. -16 %; the panels ARE shared through the XCD's L2.  profiles/r05_gemm_nt_ab.txt,
    // commit 62b3d01 has the switch)
    auto issueA = [&](int kt, int slot) {   // K step kt of the current tile into A slot `slot`
        glds16_group4(lds0 + a_slot(slot) + wave * G::SA * 1024, offA_l[0], offA_l[1], offA_l[2], offA_l[3],
                      baseA + (unsigned long long)(kbeg + kt) * (BK * 2));
    };
    // a_hl (exact numerics, GemmParams): K steps 2 j and 2 j + 1 of A hold the hi and the lo bf16 term of the SAME 64 activations -- both
    // meet W's K step j (a scalar shift; the weights are exact bf16 and are fetched into LDS once per term from the XCD's L2)
    auto issueW = [&](int kt, int slot) {
        glds16_group4(lds0 + w_slot(slot) + wave * G::SB * 1024, offB_l[0], offB_l[1], offB_l[2], offB_l[3],
                      baseB + (unsigned long long)((kbeg + kt) >> p.a_hl) * (BK * 2));
    };

    // ---- fragment read offsets: row base + swizzled chunk; (row >> 1) & 7 == (li >> 1) & 7 for every 16-row tile ----
    const int swz = (li >> 1) & 7;
    const int c0 = (g ^ swz) << 4;            // k half 0: logical chunk g;  half 1: logical chunk 4 + g == c0 ^ 64
    const int offA = (wm * (MT * 16) + li) * 128, offB = (wn * (NT * 16) + li) * 128;
    // position s = kk * MT + i of the K step: k half kk, row tile i
    auto ldA = [&](const unsigned char* st, int s_) { return *(const bf16x8_t*)(st + offA + (s_ % MT) * 2048 + ((s_ / MT) ? (c0 ^ 64) : c0)); };
    auto ldB = [&](const unsigned char* st, int kk, int j) { return *(const bf16x8_t*)(st + offB + j * 2048 + (kk ? (c0 ^ 64) : c0)); };

    const bool staged_ok = (p.ldc & 7) == 0 && (((size_t)p.C) & 15) == 0 && !(p.dbg & 8) &&
                           (ACT == 2 ? (p.N & 15) == 0 : (min(p.N, p.N_store) & 7) == 0) &&   // 16-byte row-layout stores possible
                           (!p.residual || ((p.ldr & 3) == 0 && (((size_t)p.residual) & 7) == 0));   // ... and 8-byte residual requests
    int it = blockIdx.x >> 3;
    if (it >= run_n) return;
#ifdef GEMM_LAB_TRACE
    int ntrace = 0;
#define GEMM_STAMP() do { if (p.trace && blockIdx.x == 0 && tid == 0) p.trace[ntrace++] = wall_clock64(); } while (0)
#else
#define GEMM_STAMP() do { } while (0)
#endif
    int m0, n0;
    tile_origin(it, m0, n0);
    set_sources(m0, n0);
    issueA(0, 0);
    issueW(0, 0);
    while (true) {
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // K step 0 has landed and the previous tile's stores have retired.  As a BUILTIN (vmcnt(0), expcnt / lgkmcnt untouched): hipcc does
        // not read the counters an asm statement waits for, believed the epilogue's stores still pending and drained the queue -- the
        // freshly issued requests of the next steps with it -- in the middle of every tile's first K step
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        GEMM_STAMP();
        if constexpr (PP) {
            // ---- DEEP = 3, lab: the two wave groups of the block -- waves 0 .. NW/2-1 (tile rows 0 .. BM/2) and their SIMD partners
            // NW/2 .. NW-1 -- run HALF A K STEP APART, with a bare s_barrier per half step, so that one group's DMA issue and first
            // fragment reads fall under the other's MFMAs.  A halves are private to a group (two stages, requested by the group at
            // the start of its own step, one step ahead); W is shared by both groups for three half steps, hence THREE W slots: the
            // request of step kt + 2 goes out as soon as step kt - 1's last reader has passed its end barrier.  Counted waits:
            //   group 0, end of step kt:  all but W(kt + 2) landed;   group 1, middle of step kt: its W(kt + 1) rows landed (group 0
            //   starts step kt + 1 behind that barrier), end of step kt: its A(kt + 1) landed.
            auto run_steps = [&](auto grp_tag) {
                constexpr int GRP = decltype(grp_tag)::value;
                auto bar = [] { asm volatile("s_barrier" ::: "memory"); };
                if (nk > 1) issueW(1, 1);
                if (GRP == 1) bar();                              // half step 0: group 0 alone
                int sw = 0;
                for (int kt = 0; kt < nk; ++kt) {
                    const int sw1 = sw + 1 == 3 ? 0 : sw + 1, sw2 = sw1 + 1 == 3 ? 0 : sw1 + 1;
                    int offsA = a_slot(kt & 1), offsW = w_slot(sw);
                    asm volatile("" : "+s"(offsA), "+s"(offsW));
                    const unsigned char* stA = smem + offsA;
                    const unsigned char* stW = smem + offsW;
                    const bool n1 = kt + 1 < nk, n2 = kt + 2 < nk;
                    if (n1) issueA(kt + 1, (kt + 1) & 1);
                    if (GRP == 1 && n2) issueW(kt + 2, sw2);
                    bf16x8_t fb[2][NT], fa[4];
#pragma unroll
                    for (int j = 0; j < NT; ++j) fb[0][j] = ldB(stW, 0, j);
                    fa[0] = ldA(stA, 0);
                    fa[1] = ldA(stA, 1);
                    __builtin_amdgcn_sched_group_barrier(0x100, NT + 2, 0);
#pragma unroll
                    for (int s_ = 0; s_ < 2 * MT; ++s_) {
                        const int kk = s_ / MT, i = s_ % MT;
                        if (s_ == MT) {                           // ---- middle of the step
                            if (GRP == 1) {
                                if (n1 && n2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
                                else if (n1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                            }
                            bar();
                            if (GRP == 0 && n2) issueW(kt + 2, sw2);
                        }
                        if (s_ + 2 < 2 * MT) fa[(s_ + 2) & 3] = ldA(stA, s_ + 2);
                        if (s_ < NT) fb[1][s_] = ldB(stW, 1, s_);
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[s_ & 3], acc[i][j], 0, 0, 0);
                        if (s_ < NT && s_ + 2 < 2 * MT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        else if (s_ < NT || s_ + 2 < 2 * MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                    }
                    if (n2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    sw = sw1;
                }
                if (GRP == 0) bar();                              // half step 2 nk: group 1 alone
            };
            if (__builtin_amdgcn_readfirstlane(wave) >= G::NW / 2) run_steps(std::integral_constant<int, 1>{});
            else run_steps(std::integral_constant<int, 0>{});
        } else {
        if (nk > 1) {                                        // (slot 1 was part of the epilogue window until this barrier)
            if (DEEP == 1) issueA(1, 1);
            if (DEEP == 2) issueW(1, 1);
        }
        int sd = 0;                                          // slot of the three-stage operand for the current K step (the other: kt & 1)
        for (int kt = 0; kt < nk; ++kt) {
            const int sd1 = sd + 1 == 3 ? 0 : sd + 1, sd2 = sd1 + 1 == 3 ? 0 : sd1 + 1;
            // (the slot offsets stay scalar, opaque values: hipcc otherwise keeps a per-lane fragment address for every slot in
            // registers across the loop -- ten VGPRs the 256-register main loop does not have)
            int offsA = a_slot(DEEP == 1 ? sd : (kt & 1)), offsW = w_slot(DEEP == 2 ? sd : (kt & 1));
            asm volatile("" : "+s"(offsA), "+s"(offsW));
            const unsigned char* stA = smem + offsA;
            const unsigned char* stW = smem + offsW;
            const bool dma_on = !((p.dbg & 1) && kt > 0);
            auto issue_step = [&]() {   // the resident operand of step kt + 1 first, then the streaming one of step kt + 2 (stays in flight)
                if (!dma_on) return;
                if (DEEP == 1) {
                    if (kt + 1 < nk) issueW(kt + 1, (kt + 1) & 1);
                    if (kt + 2 < nk) issueA(kt + 2, sd2);
                } else if (DEEP == 2) {
                    if (kt + 1 < nk) issueA(kt + 1, (kt + 1) & 1);
                    if (kt + 2 < nk) issueW(kt + 2, sd2);
                } else if (kt + 1 < nk) {
                    issueA(kt + 1, (kt + 1) & 1);
                    issueW(kt + 1, (kt + 1) & 1);
                }
            };
            issue_step();
            bf16x8_t fb[2][NT], fa[4];
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[0][j] = ldB(stW, 0, j);
            fa[0] = ldA(stA, 0);
            fa[1] = ldA(stA, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, NT + 2, 0);
#pragma unroll
            for (int s_ = 0; s_ < 2 * MT; ++s_) {
                const int kk = s_ / MT, i = s_ % MT;
                if (s_ + 2 < 2 * MT) fa[(s_ + 2) & 3] = ldA(stA, s_ + 2);
                if (s_ < NT) fb[1][s_] = ldB(stW, 1, s_);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[kk][j], fa[s_ & 3], acc[i][j], 0, 0, 0);
                // pin the interleave: this position's DS reads, then its NT MFMAs
                if (s_ < NT && s_ + 2 < 2 * MT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                else if (s_ < NT || s_ + 2 < 2 * MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
            }
            if (DEEP) {
                // everything but the streaming operand's request of step kt + 2 (this wave's last four) has landed; the barrier carries no memory wait
                if (kt + 2 < nk && dma_on) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage has landed ...
                __syncthreads();                                    // ... and everybody is done reading this one
            }
            sd = sd1;
        }
        }
        // the next tile's first K step is requested BEFORE this tile's epilogue: its HBM/L2 latency (and the block
        // re-launch a one-tile-per-block grid would pay) hides under the stores
        GEMM_STAMP();
        const int cm0 = m0, cn0 = n0;
        const size_t c_off = ks > 1 ? (size_t)kidx * p.M * p.ldc : 0;
        it += per_xcd;
        const bool more = it < run_n;
        if (more) {
            tile_origin(it, m0, n0);
            set_sources(m0, n0);
            issueA(0, 0);
            issueW(0, 0);
        }
        GEMM_STAMP();
        if (!(p.dbg & 2)) {
            // everything behind stage 0 is free between the last K step of this tile and the second DMA of the next one
            if (!OUT_F32 && staged_ok) gemm_epilogue_staged<G, ACT, LN>(p, acc, cm0, cn0, wm, wn, g, li, lane, smem + STAGE_BYTES + wave * G::WIN);
            else gemm_epilogue<G, ACT, OUT_F32>(p, acc, cm0, cn0, wm, wn, g, li, c_off);
        }
        GEMM_STAMP();
        if (!more) break;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// 128 x 256 x 32 tile, TWO independent 4-wave blocks per CU (round 5; VERDICT r04 next #1a).
// The 256 x 256 tile keeps both waves of a SIMD in ONE block: they meet the same barriers, issue their DMA requests at the same time and
// run their epilogues together -- for ~9 of the ~32 us of a K = 1024 tile no MFMA of the CU runs.  Here a block is four waves side by
// side (wave tile 128 x 64: the same 128 accumulator registers and the same 0.375 fragment reads per MFMA as the big tile), one per
// SIMD, and TWO blocks share the CU: whatever stalls one block -- a step barrier, DMA issue, LDS latency, the whole epilogue and the
// tile-top wait -- leaves the SIMDs to the other block's MFMAs, and the two blocks' tile boundaries drift apart by themselves.
// K steps are 32 deep (64-byte rows) so that THREE stages of (8 + 16) KiB fit twice into the CU's 160 KiB: the request of K step t + 2
// goes out at the top of step t (into the stage step t - 1 read), ONE bare s_barrier per step, waits counted (vmcnt(6): the six
// requests of step t + 1 stay in flight).  Price: 0.0117 instead of 0.0078 L2 -> LDS bytes per FLOP.
// LDS image of a stage: A rows 0..127 then W rows 0..255, 64 bytes each = 4 chunks of 16 bytes; a wave-level DMA writes 1 KiB = 16 whole
// rows in lane order, the swizzle is applied to the SOURCE address: physical chunk c of row r holds logical chunk c ^ F((r >> 2) & 3),
// F = (0, 2, 3, 1).  A ds_read_b128 fragment read (lane (g, li): row li of a 16-row tile, logical chunk g) is serviced in four groups
// of 16 lanes {g: li 0-3, 12-15; g ^ 1: li 4-11}: per row residue mod 4 the four lanes see chunks F(0), F(3), 1 ^ F(1), 1 ^ F(2)
// (resp. F(1), F(2), 1 ^ F(0), 1 ^ F(3)) -- all different with this F: sixteen distinct 16-byte slots of the 256-byte bank row.
// ---------------------------------------------------------------------------------------------------------------------
using GeomK32 = Geom<1, 4, 8, 4>;   // (for the epilogues: 4 waves side by side, MT x NT = 8 x 4 MFMA tiles per wave)
constexpr int K32_BK = 32;
constexpr int K32_OPA = 128 * 64, K32_OPB = 256 * 64, K32_STAGE = K32_OPA + K32_OPB, K32_NS = 3, K32_SMEM = K32_NS * K32_STAGE;
static_assert(K32_STAGE + GeomK32::NW * GeomK32::WIN <= K32_SMEM, "the epilogue windows live behind stage 0");

__device__ __forceinline__ void glds16_group2(unsigned int lds_base, unsigned int v0, unsigned int v1, unsigned long long sbase) {
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\t"
                 "global_load_lds_dwordx4 %1, %3 offset:0\n\t"
                 "global_load_lds_dwordx4 %2, %3 offset:1024"
                 ::"s"(__builtin_amdgcn_readfirstlane(lds_base)), "v"(v0), "v"(v1), "s"(sbase)
                 : "m0", "memory");
}
__device__ __forceinline__ int k32_swz(int row) { return (0x1320 >> (((row >> 2) & 3) * 4)) & 3; }

template <int ACT, bool OUT_F32, bool LN>
__global__ __launch_bounds__(256, 2) void emmax_gemm_k32_kernel(GemmParams p) {
    using G = GeomK32;
    constexpr int BM = G::BM, BN = G::BN, MT = G::MT, NT = G::NT;
    static_assert(BM == 128 && BN == 256, "stage image");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = 0, wn = wave;
    const int g = lane >> 4, li = lane & 15;

    // ---- persistent tile walk: as emmax_gemm_bf16_kernel (banded, XCD-aware; split-K units slice-major) ----
    const int tiles_m = (p.M + BM - 1) / BM, tiles_n = (p.N + BN - 1) / BN;
    const int ntile = tiles_m * tiles_n;
    const int ks = p.ksplit > 1 ? p.ksplit : 1;
    const int nwg = ntile * ks;
    const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;
    const int rq = nwg >> 3, rr = nwg & 7;
    const int run0 = xcd < rr ? xcd * (rq + 1) : rr * (rq + 1) + (xcd - rr) * rq, run_n = rq + (xcd < rr ? 1 : 0);
    const int nk_all = p.K / K32_BK;
    const int nk_slice = (nk_all + ks - 1) / ks;
    int kidx = 0, kbeg = 0, nk = nk_all;
    auto tile_origin = [&](int it, int& m0, int& n0) {
        const int unit = run0 + it;
        kidx = unit / ntile;
        kbeg = kidx * nk_slice;
        nk = min(nk_slice, nk_all - kbeg);
        const int bid = unit - kidx * ntile;
        const int band = bid / (BAND * tiles_m), in_band = bid - band * (BAND * tiles_m);
        const int bw = min(BAND, tiles_n - band * BAND);
        const int tm = in_band / bw, tn = band * BAND + (in_band - tm * bw);
        m0 = tm * BM;
        n0 = tn * BN;
    };

    // ---- DMA sources: wave w fills slabs (16 rows of 64 bytes) 2 w, 2 w + 1 of the A image and 4 w .. 4 w + 3 of the W image ----
    unsigned int offA_l[2], offB_l[4];
    unsigned long long baseA = 0, baseB = 0;
    const unsigned int lds0 = __builtin_amdgcn_readfirstlane((unsigned int)(uintptr_t)(__attribute__((address_space(3))) void*)smem);
    auto set_sources = [&](int m0, int n0) {
        baseA = (unsigned long long)(uintptr_t)((const bf16_t*)p.A + (size_t)m0 * p.lda) - DMA_BIAS;
        baseB = (unsigned long long)(uintptr_t)((const bf16_t*)p.W + (size_t)n0 * p.ldw) - DMA_BIAS;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int row = (wave * 2 + j) * 16 + (lane >> 2);
            const int gm = min(m0 + row, p.M - 1) - m0;   // edge tiles re-read the last row (never stored)
            offA_l[j] = (unsigned int)gm * (unsigned int)(p.lda * 2) + ((lane & 3) ^ k32_swz(row)) * 16 + (DMA_BIAS - j * 1024);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int row = (wave * 4 + j) * 16 + (lane >> 2);
            const int gn = min(n0 + row, p.N - 1) - n0;
            offB_l[j] = (unsigned int)gn * (unsigned int)(p.ldw * 2) + ((lane & 3) ^ k32_swz(row)) * 16 + (DMA_BIAS - j * 1024);
        }
    };
    auto issue = [&](int kt, int slot) {   // K step kt of the current tile into stage `slot`: this wave's six slabs
        const unsigned long long ko = (unsigned long long)(kbeg + kt) * (K32_BK * 2);
        glds16_group2(lds0 + slot * K32_STAGE + wave * 2048, offA_l[0], offA_l[1], baseA + ko);
        glds16_group4(lds0 + slot * K32_STAGE + K32_OPA + wave * 4096, offB_l[0], offB_l[1], offB_l[2], offB_l[3], baseB + ko);
    };
    // lab (gemm_dbg bit 32): the same six requests in three pairs, issued BETWEEN the MFMA groups of the step instead of at its top
    auto issue_part = [&](int kt, int slot, int part) {
        const unsigned long long ko = (unsigned long long)(kbeg + kt) * (K32_BK * 2);
        if (part == 0) glds16_group2(lds0 + slot * K32_STAGE + wave * 2048, offA_l[0], offA_l[1], baseA + ko);
        else if (part == 1) glds16_group2(lds0 + slot * K32_STAGE + K32_OPA + wave * 4096, offB_l[0], offB_l[1], baseB + ko);
        else glds16_group2(lds0 + slot * K32_STAGE + K32_OPA + wave * 4096 + 2048, offB_l[2] + 2048, offB_l[3] + 2048, baseB + ko);
    };
    const bool spread = (p.dbg & 32) != 0;

    // ---- fragment reads: lane (g, li) = row li of a 16-row tile, logical chunk g ----
    const int c0 = (g ^ k32_swz(li)) << 4;
    const int offA = li * 64 + c0, offB = K32_OPA + (wn * (NT * 16) + li) * 64 + c0;
    auto ldA = [&](const unsigned char* st, int i) { return *(const bf16x8_t*)(st + offA + i * 1024); };
    auto ldB = [&](const unsigned char* st, int j) { return *(const bf16x8_t*)(st + offB + j * 1024); };

    const bool staged_ok = (p.ldc & 7) == 0 && (((size_t)p.C) & 15) == 0 &&
                           (ACT == 2 ? (p.N & 15) == 0 : (min(p.N, p.N_store) & 7) == 0) &&
                           (!p.residual || ((p.ldr & 3) == 0 && (((size_t)p.residual) & 7) == 0));
    int it = blockIdx.x >> 3;
    if (it >= run_n) return;
    int m0, n0;
    tile_origin(it, m0, n0);
    set_sources(m0, n0);
    issue(0, 0);
    auto bar = [] { asm volatile("s_barrier" ::: "memory"); };
    while (true) {
        f32x4_t acc[MT][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        // K step 0 has landed, the previous tile's stores have retired (the builtin: see emmax_gemm_bf16_kernel), and behind the barrier
        // every wave has left its epilogue window: stages 1 and 2 may be written again
        __builtin_amdgcn_s_waitcnt(0x0F70);
        __syncthreads();
        if (nk > 1) issue(1, 1);
        int st_i = 0;   // stage of the current K step
        for (int kt = 0; kt < nk; ++kt) {
            const int st_2 = st_i >= 1 ? st_i - 1 : 2;   // (st_i + 2) % 3: the stage step kt - 1 read
            if (kt > 0) {
                // this wave's requests of step kt have landed (those of step kt + 1 stay in flight); behind the barrier everybody's have,
                // and everybody is done reading step kt - 1
                if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                bar();
            }
            if (kt + 2 < nk && !spread) issue(kt + 2, st_2);
            int offs = st_i * K32_STAGE;
            asm volatile("" : "+s"(offs));
            const unsigned char* st = smem + offs;
            bf16x8_t fb[NT], fa[4];
#pragma unroll
            for (int j = 0; j < NT; ++j) fb[j] = ldB(st, j);
            fa[0] = ldA(st, 0);
            fa[1] = ldA(st, 1);
            __builtin_amdgcn_sched_group_barrier(0x100, NT + 2, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                if (i + 2 < MT) fa[(i + 2) & 3] = ldA(st, i + 2);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fb[j], fa[i & 3], acc[i][j], 0, 0, 0);
                if (i + 2 < MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, NT, 0);
                if (spread && kt + 2 < nk && (i == 1 || i == 3 || i == 5)) issue_part(kt + 2, st_2, i >> 1);
            }
            st_i = st_i == 2 ? 0 : st_i + 1;
        }
        // every wave is done reading the last stage (it may be stage 0, which the next tile's first request overwrites, or lie under an
        // epilogue window)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        bar();
        const int cm0 = m0, cn0 = n0;
        const size_t c_off = ks > 1 ? (size_t)kidx * p.M * p.ldc : 0;
        it += per_xcd;
        const bool more = it < run_n;
        if (more) {   // the next tile's first K step is requested BEFORE this tile's epilogue
            tile_origin(it, m0, n0);
            set_sources(m0, n0);
            issue(0, 0);
        }
        if (!OUT_F32 && staged_ok) gemm_epilogue_staged<G, ACT, LN>(p, acc, cm0, cn0, wm, wn, g, li, lane, smem + K32_STAGE + wave * G::WIN);
        else gemm_epilogue<G, ACT, OUT_F32>(p, acc, cm0, cn0, wm, wn, g, li, c_off);
        if (!more) break;
    }
}

template <int ACT, bool OUT_F32, bool LN = false>
int launch_k32_t(const GemmParams& p, hipStream_t stream) {
    auto kern = emmax_gemm_k32_kernel<ACT, OUT_F32, LN>;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, K32_SMEM) != hipSuccess) return -4;
        attr_done = true;
    }
    const int tiles = cdiv(p.M, GeomK32::BM) * cdiv(p.N, GeomK32::BN) * (p.ksplit > 1 ? p.ksplit : 1);
    const int resident = 512;   // two blocks per CU
    const int grid = tiles < resident ? (tiles + 7) / 8 * 8 : resident;
    GemmParams q = p;
    q.dbg |= emmax_tune().gemm_dbg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), K32_SMEM, stream, q);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}
int launch_k32(const GemmParams& p, hipStream_t stream) {
    if (p.K % K32_BK) return -1;
    if (p.act == 2) return p.out_f32 ? -1 : launch_k32_t<2, false>(p, stream);
    if (p.ln_stats && !p.out_f32) return p.act == 1 ? launch_k32_t<1, false, true>(p, stream) : launch_k32_t<0, false, true>(p, stream);
    if (p.act == 1) return p.out_f32 ? launch_k32_t<1, true>(p, stream) : launch_k32_t<1, false>(p, stream);
    return p.out_f32 ? launch_k32_t<0, true>(p, stream) : launch_k32_t<0, false>(p, stream);
}

// split-K second pass: C[m, n..n+8) = epi(sum over slices of ws[s][m][..)); one thread per 8 OUTPUT columns, whole rows in order.
// ACT = 2 (SwiGLU): the partial tiles hold the interleaved (gate, up) 16-column groups of the weight order; output column o of
// group o / 16 pairs partial columns (o / 16) * 32 + o % 16 and + 16.  Eight consecutive output columns never straddle a group.
template <int ACT>
__global__ __launch_bounds__(256) void emmax_splitk_reduce_kernel(GemmParams p) {
    const int n_out = ACT == 2 ? p.N >> 1 : p.N;
    const int nc = n_out >> 3;   // 8-column groups per output row (N % 128 == 0)
    const size_t total = (size_t)p.M * nc, slab = (size_t)p.M * p.N;
    const bf16_t* bias = (const bf16_t*)p.bias;
    const bf16_t* scale = (const bf16_t*)p.scale;
    const bf16_t* res = (const bf16_t*)p.residual;
    const int n_ok = ACT == 2 ? n_out : min(p.N, p.N_store);
    const bool vec_c = !p.out_f32 && (p.ldc & 7) == 0 && (((size_t)p.C) & 15) == 0;
    const bool vec_c32 = p.out_f32 && (p.ldc & 3) == 0 && (((size_t)p.C) & 15) == 0;
    const bool vec_r = res && !p.res_f32 && (p.ldr & 7) == 0 && (((size_t)res) & 15) == 0;
    const bool vec_r32 = res && p.res_f32 && (p.ldr & 3) == 0 && (((size_t)res) & 15) == 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int row = (int)(i / nc), col = (int)(i - (size_t)row * nc) * 8;
        const int pcol = ACT == 2 ? (col >> 4) * 32 + (col & 15) : col;
        const float* src = p.ws + (size_t)row * p.N + pcol;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, u[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        for (int s = 0; s < p.ksplit; ++s) {
            const f32x4_t a = *(const f32x4_t*)(src + s * slab), b = *(const f32x4_t*)(src + s * slab + 4);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                v[e] += a[e];
                v[4 + e] += b[e];
            }
            if (ACT == 2) {
                const f32x4_t c = *(const f32x4_t*)(src + s * slab + 16), d = *(const f32x4_t*)(src + s * slab + 20);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    u[e] += c[e];
                    u[4 + e] += d[e];
                }
            }
        }
        const bool full = col + 7 < n_ok;
        u32x4_t rv = {0u, 0u, 0u, 0u};
        if (ACT != 2 && vec_r && full) rv = *(const u32x4_t*)(res + (size_t)row * p.ldr + col);
        f32x4_t rf0 = {0.f, 0.f, 0.f, 0.f}, rf1 = {0.f, 0.f, 0.f, 0.f};   // fp32 residual stream
        if (ACT != 2 && vec_r32 && full) {
            const float* rq = (const float*)p.residual + (size_t)row * p.ldr + col;
            rf0 = *(const f32x4_t*)rq;
            rf1 = *(const f32x4_t*)(rq + 4);
        }
        float x[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            if (ACT == 2) {
                x[e] = silu(v[e]) * u[e];
            } else {
                x[e] = v[e] + ((bias && col + e < n_ok) ? bf2f(bias[col + e]) : 0.f);
                if (ACT == 1) x[e] = gelu_erf(x[e]);
                if (scale && col + e < n_ok) x[e] *= bf2f(scale[col + e]);
                if (res) {
                    if (vec_r && full) x[e] += (e & 1) ? bf_hi(rv[e >> 1]) : bf_lo(rv[e >> 1]);
                    else if (vec_r32 && full) x[e] += e < 4 ? rf0[e & 3] : rf1[e & 3];
                    else if (col + e < n_ok) x[e] += p.res_f32 ? ((const float*)p.residual)[(size_t)row * p.ldr + col + e] : bf2f(res[(size_t)row * p.ldr + col + e]);
                }
            }
        }
        if (vec_c && full) {
            *(u32x4_t*)((bf16_t*)p.C + (size_t)row * p.ldc + col) =
                (u32x4_t){pack_bf16x2(x[0], x[1]), pack_bf16x2(x[2], x[3]), pack_bf16x2(x[4], x[5]), pack_bf16x2(x[6], x[7])};
        } else if (vec_c32 && full) {
            float* cp = (float*)p.C + (size_t)row * p.ldc + col;
            *(f32x4_t*)cp = (f32x4_t){x[0], x[1], x[2], x[3]};
            *(f32x4_t*)(cp + 4) = (f32x4_t){x[4], x[5], x[6], x[7]};
        } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                if (col + e >= n_ok) continue;
                if (p.out_f32) ((float*)p.C)[(size_t)row * p.ldc + col + e] = x[e];
                else ((bf16_t*)p.C)[(size_t)row * p.ldc + col + e] = f2bf(x[e]);
            }
        }
    }
}

// The same second pass with the RMSNorm that consumes its result (HF LlamaRMSNorm: fp32 statistics of the bf16 row, normalise, round,
// multiply by the weight): ONE WAVE PER ROW with the lane -> chunk map, the accumulation order and the arithmetic of
// emmax_rownorm_kernel<8, true> (norm.hip), so C and norm_out are bit-identical to the reduce pass followed by that kernel -- the row is
// summed, finished (bias / LayerScale / residual), rounded and stored, and its rounded values stay in registers for the statistics.
// One launch and one read of the row less per o-proj / down projection of a one-frame prefill.  N = 4096 (512 NV), act 0, bf16 C.
// F32 (round 5): the fp32 residual stream of the prefill -- `residual` and C are f32 rows (C may alias the residual), the statistics and
// the normalise take the fp32 sums (HF's arithmetic on an fp32 hidden state: normalise, round to bf16, multiply by the weight, round);
// nothing of the stream is rounded to bf16 except the normalised row the next GEMM consumes.
template <int NV, bool F32>   // 16-byte chunks per lane: N = 512 NV exactly, so that no access is predicated and a slice's 2 NV loads go out together
__global__ __launch_bounds__(256) void emmax_splitk_reduce_norm_kernel(GemmParams p) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= p.M) return;
    const size_t slab = (size_t)p.M * p.N;
    const bf16_t* bias = (const bf16_t*)p.bias;
    const bf16_t* scale = (const bf16_t*)p.scale;
    const bf16_t* res = (const bf16_t*)p.residual;
    const float* res32 = (const float*)p.residual;
    const float* src = p.ws + (size_t)row * p.N + lane * 8;
    u32x4_t rv[F32 ? 1 : NV];
    f32x8_t rq[F32 ? NV : 1];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        if constexpr (F32) rq[i] = p.residual ? ld_f32x8(res32 + (size_t)row * p.ldr + (lane + 64 * i) * 8) : f32x8_t{{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        else rv[i] = res ? *(const u32x4_t*)(res + (size_t)row * p.ldr + (lane + 64 * i) * 8) : (u32x4_t){0u, 0u, 0u, 0u};
    }
    float a8[NV][8];
#pragma unroll
    for (int i = 0; i < NV; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) a8[i][e] = 0.f;
    for (int sl = 0; sl < p.ksplit; ++sl) {
        f32x4_t a[NV], b[NV];
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            a[i] = *(const f32x4_t*)(src + sl * slab + i * 512);
            b[i] = *(const f32x4_t*)(src + sl * slab + i * 512 + 4);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                a8[i][e] += a[i][e];
                a8[i][4 + e] += b[i][e];
            }
    }
    u32x4_t v[F32 ? 1 : NV];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int col = (lane + 64 * i) * 8;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float x = a8[i][e] + (bias ? bf2f(bias[col + e]) : 0.f);
            if (scale) x *= bf2f(scale[col + e]);
            if constexpr (F32) x += f32x8_at(rq[i], e);
            else if (res) x += (e & 1) ? bf_hi(rv[i][e >> 1]) : bf_lo(rv[i][e >> 1]);
            a8[i][e] = x;
        }
        if constexpr (F32) {
            float* cp = (float*)p.C + (size_t)row * p.ldc + col;
            *(f32x4_t*)cp = (f32x4_t){a8[i][0], a8[i][1], a8[i][2], a8[i][3]};
            *(f32x4_t*)(cp + 4) = (f32x4_t){a8[i][4], a8[i][5], a8[i][6], a8[i][7]};
#pragma unroll
            for (int e = 0; e < 8; ++e) ss = __builtin_fmaf(a8[i][e], a8[i][e], ss);   // spelled out: emmax_rmsnorm_f32_kernel must round identically
        } else {
            v[i] = (u32x4_t){pack_bf16x2(a8[i][0], a8[i][1]), pack_bf16x2(a8[i][2], a8[i][3]), pack_bf16x2(a8[i][4], a8[i][5]), pack_bf16x2(a8[i][6], a8[i][7])};
            *(u32x4_t*)((bf16_t*)p.C + (size_t)row * p.ldc + col) = v[i];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = bf_lo(v[i][j]), bb = bf_hi(v[i][j]);
                ss += a * a + bb * bb;
            }
        }
    }
    ss = wave_sum(ss);
    const float rstd = rsqrtf(ss / (float)p.N + p.norm_eps);
    const bf16_t* w = (const bf16_t*)p.norm_w;
    bf16_t* yr = (bf16_t*)p.norm_out + (size_t)row * p.ld_norm;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const int c = lane + 64 * i;
        const u32x4_t wv = *(const u32x4_t*)(w + c * 8);
        u32x4_t o;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            float a, bb;
            if constexpr (F32) { a = a8[i][2 * j] * rstd; bb = a8[i][2 * j + 1] * rstd; }
            else { a = bf_lo(v[i][j]) * rstd; bb = bf_hi(v[i][j]) * rstd; }
            a = bf2f(f2bf(a)) * bf_lo(wv[j]);
            bb = bf2f(f2bf(bb)) * bf_hi(wv[j]);
            o[j] = pack_bf16x2(a, bb);
        }
        *(u32x4_t*)(yr + c * 8) = o;
    }
}

template <class G, int ACT, bool OUT_F32, bool LN = false, int DEEP = 0>
int launch_t(const GemmParams& p, hipStream_t stream) {
    if constexpr (DEEP == 0) {
        // Main-loop variant by geometry (tuning switch gemm_deep = -1; 0 = two stages of both operands, one barrier per K step, as in
        // rounds 1-3; 1 = deep A ring; 3 = staggered wave groups).  Measured, interleaved A/B per shape on one box
        // (profiles/r04_gemm_deep_ab.txt, r04_gemm_pp_ab.txt): the 256 x 256 tile gains 1-14 % on EVERY shape with the staggered
        // schedule (8192^3 1385 -> 1493 TFLOP/s, ViT shapes +7 .. 14 %, LLaMA prefill B = 8 +1 .. 4 %), more than with the deep A ring
        // (ViT +3 .. 10 %, LLaMA -4 %); the 128 x 128 tile (two blocks of four waves per CU: nothing to stagger) gains 25-30 % per launch
        // with the deep A ring.
        int deep = emmax_tune().gemm_deep;
        constexpr bool big = std::is_same<G, GeomBig>::value;
        if (deep < 0) deep = big ? 3 : 1;
        // (round 5: fp32 results take the staggered loop too -- the direct epilogue needs no LDS window; they had been left on the
        // two-stage loop, which cost the fp32 residual stream of an eight-frame prefill 4 %)
        if (deep == 3 && !big) deep = 1;
        if (deep == 1) return launch_t<G, ACT, OUT_F32, LN, 1>(p, stream);
        if constexpr (big) {
            if (deep == 3) return launch_t<G, ACT, OUT_F32, LN, 3>(p, stream);
        }
    }
    auto kern = emmax_gemm_bf16_kernel<G, ACT, OUT_F32, LN, DEEP>;
    constexpr int SMEM = DEEP ? G::SMEM_A3 : G::SMEM;
    static_assert(DEEP != 2, "the deep W ring (round 4: never faster than the deep A ring) is not instantiated");
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) return -4;
        attr_done = true;
    }
    // persistent: as many blocks as fit the chip at once (one big / two small per CU), a multiple of the 8 XCDs
    const int tiles = cdiv(p.M, G::BM) * cdiv(p.N, G::BN) * (p.ksplit > 1 ? p.ksplit : 1);
    const int resident = 256 * (160 * 1024 / SMEM);
    const int grid = tiles < resident ? (tiles + 7) / 8 * 8 : resident;
    GemmParams q = p;
    q.dbg |= emmax_tune().gemm_dbg;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(G::NW * 64), SMEM, stream, q);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <class G>
int launch_geom(const GemmParams& p, hipStream_t stream) {
    if (p.act == 2) return p.out_f32 ? launch_t<G, 2, true>(p, stream) : launch_t<G, 2, false>(p, stream);
    if (p.ln_stats && !p.out_f32)   // LayerNorm folded in: the staged epilogue's LN form (an unaligned C falls to the direct epilogue, which reads p.ln_*)
        return p.act == 1 ? launch_t<G, 1, false, true>(p, stream) : launch_t<G, 0, false, true>(p, stream);
    if (p.act == 1) return p.out_f32 ? launch_t<G, 1, true>(p, stream) : launch_t<G, 1, false>(p, stream);
    return p.out_f32 ? launch_t<G, 0, true>(p, stream) : launch_t<G, 0, false>(p, stream);
}

}  // namespace

// number of 256x256 tiles of the problem
int gemm_big_tiles(const GemmParams& p) { return cdiv(p.M, GeomBig::BM) * cdiv(p.N, GeomBig::BN); }

int launch_gemm_geom(const GemmParams& p, int big, hipStream_t stream) {
    if (p.M <= 0) return 0;
    if (p.K % BK != 0 || p.N % 128 != 0 || p.K <= 0 || p.N <= 0) return -1;
    if ((p.lda % 8) || (p.ldw % 8)) return -1;
    if (p.ln_stats && (!p.ln_s || !p.ln_c || p.bias || p.scale || p.residual || p.act == 2 || p.ksplit > 1)) return -1;
    if (p.res_f32 && (!p.out_f32 || !p.residual || p.act == 2)) return -1;   // the fp32 residual stream yields an fp32 result
    if (p.a_hl && (p.a_hl != 1 || p.K % (2 * BK) || p.ln_stats || big == 2)) return -1;   // (hi, lo) pairs of K steps; no folded LayerNorm, no 32-deep tile
    if (big == 2) return launch_k32(p, stream);   // 128 x 256 x 32, two blocks per CU
    return big ? launch_geom<GeomBig>(p, stream) : launch_geom<GeomSmall>(p, stream);
}

// ---- launch plan ------------------------------------------------------------------------------------------------------
// Cost model in units of one full round of the big geometry (256 tiles of 256x256, one per CU), calibrated with
// tools/gemm_bench.py on the ViT / prefill shapes (predicted vs measured small/big time ratios agree within 3 %):
//   big   : whole rounds, a partly filled round costs a full one (one tile per CU);
//   small : 512 tiles (= 128 big tiles of work) per round at 0.63 -- i.e. 0.79x the per-CU rate of the big geometry -- and
//           0.45 for a last round that leaves at most one block per CU.
// Tile quantisation is the main loss left in these GEMMs (e.g. 1044 big tiles = 4.08 rounds), so besides "all big" and
// "all small" the plan may split the rows: whole rounds of big tiles first, the remaining rows as a second launch of the
// finer small tiles.
static double cost_big(long tiles) { return tiles ? (double)((tiles + 255) / 256) : 0.0; }
static double cost_small(long tiles) {
    if (!tiles) return 0.0;
    const long r = (tiles + 511) / 512;
    return (double)(r - 1) * 0.63 + ((tiles - (r - 1) * 512) <= 256 ? 0.45 : 0.63);
}

// slices for launch_gemm_splitk, or 0 when splitting does not pay
static int splitk_plan(const GemmParams& p, int* big_out = nullptr) {
    if (big_out) *big_out = 0;
    if (!p.ws || p.ln_stats || (p.act == 2 && p.out_f32)) return 0;
    const int nk = p.K / BK;
    // 256 x 256 tiles with floor(256 / tiles) slices (round 5, tools/gemm_sk_sweep.py, profiles/r05_gemm_sk_sweep.txt): for a VERY long K
    // (the down projection, 172 steps) and at most half a round of big tiles, one slice of a big tile per CU beats the small tiles --
    // M = 768: 78.7 against 86.1 us (5 slices of 48 tiles), M = 1536: 138 against 151 us (2 slices of 96 tiles; that shape had no
    // split-K plan at all: 384 small tiles); at K = 4096 (o-proj) the small tiles stay ahead (44 against 49 us), from 144 tiles on the
    // plans are within noise of each other; 64 / 80 tiles (M = 1024 / 1280): 92 / 109 against 99 / 146 us.  The fp32 partials of a slice cost the same per tile in both geometries; what the big tile
    // buys is its K-step rate (section 6), what it costs is slices of uneven value: 240 of 256 CUs busy.
    {
        const long tb = (long)cdiv(p.M, GeomBig::BM) * cdiv(p.N, GeomBig::BN);
        int ks = tb >= 48 && tb <= 128 ? (int)(256 / tb) : 0;   // (below 48 big tiles the small ones win: 16 tiles 36 against 44 us, 32 tiles 51 against 55)
        if (ks > 8) ks = 8;
        while (ks >= 2 && (long long)ks * p.M * p.N * 4 > p.ws_bytes) --ks;
        if (nk >= 128 && ks >= 2 && nk / ks >= 16 && p.act != 2 && emmax_tune().gemm_sk_big != 0) {
            if (big_out) *big_out = 1;
            return ks;
        }
    }
    const long ts = (long)cdiv(p.M, GeomSmall::BM) * cdiv(p.N, GeomSmall::BN);
    if (nk < 32 || ts > 224) return 0;              // K >= 2048, at most ~1 block per CU without the split
    int ks = (int)(512 / ts);
    if (ks > nk / 8) ks = nk / 8;
    if (ks > 8) ks = 8;
    while (ks >= 2 && (long long)ks * p.M * p.N * 4 > p.ws_bytes) --ks;
    return ks >= 2 ? ks : 0;
}

// rows [r0, r0 + rows) of the problem as one launch of the given geometry
static int launch_rows(const GemmParams& p, size_t r0, int rows, int big, hipStream_t stream) {
    if (rows <= 0) return 0;
    GemmParams q = p;
    q.M = rows;
    q.A = (const bf16_t*)p.A + r0 * p.lda;
    q.C = p.out_f32 ? (void*)((float*)p.C + r0 * p.ldc) : (void*)((bf16_t*)p.C + r0 * p.ldc);
    if (p.residual) q.residual = p.res_f32 ? (const void*)((const float*)p.residual + r0 * p.ldr) : (const void*)((const bf16_t*)p.residual + r0 * p.ldr);
    if (p.ln_stats) q.ln_stats = p.ln_stats + r0 * 2;
    return launch_gemm_geom(q, big, stream);
}

// plan over the rows of one column range: all big, all small, or m1 big tile rows + the rest in small tiles
static double plan_rows(int M, int N, long* m1_out) {
    const long tm = cdiv(M, GeomBig::BM), tn = cdiv(N, GeomBig::BN), sn = cdiv(N, GeomSmall::BN);
    const double all_big = cost_big(tm * tn), all_small = cost_small((long)cdiv(M, GeomSmall::BM) * sn);
    double best = all_big <= all_small ? all_big : all_small;
    long best_m1 = all_big <= all_small ? tm : 0;
    for (long m1 = 1; m1 < tm; ++m1) {   // + one kernel boundary
        const double c = cost_big(m1 * tn) + cost_small((long)cdiv(M - (int)(m1 * GeomBig::BM), GeomSmall::BM) * sn) + 0.03;
        if (c < best - 1e-9) {
            best = c;
            best_m1 = m1;
        }
    }
    *m1_out = best_m1;
    return best;
}

static int launch_planned_rows(const GemmParams& p, long m1, hipStream_t stream) {
    const long tm = cdiv(p.M, GeomBig::BM);
    if (m1 >= tm) return launch_gemm_geom(p, 1, stream);
    if (m1 <= 0) return launch_gemm_geom(p, 0, stream);
    const size_t r0 = (size_t)m1 * GeomBig::BM;
    const int r = launch_rows(p, 0, (int)r0, 1, stream);
    return r ? r : launch_rows(p, r0, p.M - (int)r0, 0, stream);
}

// Split-K for under-filled problems with a long K: fewer small tiles than CUs-and-a-half means one 4-wave block per CU
// grinding through K alone (M = 768 prefill o / down: 192 tiles, 64-172 K steps at ~0.75 us; batch-1 ViT fc2: 24 tiles).
// ks slices per tile fill the chip (<= 512 resident blocks), each >= 8 K steps; the partial tiles meet in a second pass.
// the reduce pass can apply p.norm_*: whole rows of 4096 bf16 columns (the LLaMA-2-7B hidden size; other widths keep the separate norm), 16-byte accesses
static bool splitk_norm_ok(const GemmParams& p) {
    // (out_f32: only as the fp32 residual stream -- fp32 residual in, fp32 C out)
    return p.norm_out && p.norm_w && p.act == 0 && (!p.out_f32 || p.res_f32) && (!p.res_f32 || p.out_f32) && !p.ln_stats && p.N == 4096 &&
           p.N_store >= p.N && (p.ldc & 7) == 0 && (p.ld_norm & 7) == 0 && (((size_t)p.C | (size_t)p.norm_out | (size_t)p.norm_w) & 15) == 0 &&
           (!p.residual || ((p.ldr & 7) == 0 && (((size_t)p.residual) & 15) == 0));
}

int launch_gemm_splitk(const GemmParams& p, int ks, hipStream_t stream, int big) {
    if (ks < 2 || !p.ws || p.K % BK || p.N % 128 || p.ln_stats) return -1;
    if (p.act == 2 && (p.out_f32 || (p.N & 31))) return -1;
    if ((long long)ks * p.M * p.N * 4 > p.ws_bytes) return -1;
    if (p.K / BK < ks) return -1;
    // (the tools' geometry override gemm_sk_big is applied where the plan is made -- plan_gemm, kind SPLITK -- not here: the HYBRID column
    // remainder is planned for small tiles and its slice count was chosen for them)
    GemmParams a = p;
    a.ksplit = ks;
    a.C = p.ws; a.ldc = p.N; a.N_store = p.N; a.out_f32 = 1; a.act = 0;
    a.bias = nullptr; a.scale = nullptr; a.residual = nullptr; a.res_f32 = 0;
    int r = launch_gemm_geom(a, big, stream);
    if (r) return r;
    GemmParams b = p;
    b.ksplit = ks;
    const size_t groups = (size_t)p.M * ((p.act == 2 ? p.N >> 1 : p.N) >> 3);
    const int grid = (int)((groups + 255) / 256 < 2048 ? (groups + 255) / 256 : 2048);
    if (p.norm_out) {
        if (!splitk_norm_ok(p)) return -1;
        if (p.res_f32) hipLaunchKernelGGL((emmax_splitk_reduce_norm_kernel<8, true>), dim3((p.M + 3) / 4), dim3(256), 0, stream, b);
        else hipLaunchKernelGGL((emmax_splitk_reduce_norm_kernel<8, false>), dim3((p.M + 3) / 4), dim3(256), 0, stream, b);
    } else if (p.act == 2) hipLaunchKernelGGL(emmax_splitk_reduce_kernel<2>, dim3(grid), dim3(256), 0, stream, b);
    else if (p.act == 1) hipLaunchKernelGGL(emmax_splitk_reduce_kernel<1>, dim3(grid), dim3(256), 0, stream, b);
    else hipLaunchKernelGGL(emmax_splitk_reduce_kernel<0>, dim3(grid), dim3(256), 0, stream, b);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

// Column remainder through split-K (round 4, late).  A short, wide problem whose big tiles are a few more than whole rounds -- the
// one-frame prefill gate/up: 3 x 86 = 258 tiles = one round + 2 tiles, i.e. a second round for 1 % of the work (or 1032 small
// tiles, 2.02 rounds) -- cannot use the row split (three tile rows).  Instead: columns [0, n1) = as many whole tile columns as fit
// the whole rounds (85 x 3 = 255 tiles, one round), and the remaining columns as K-split 128 x 128 tiles that fill the chip for a
// few K steps each + the reduce / epilogue pass (SwiGLU included).  Returns the slices (0: not applicable) and n1.
static int hybrid_cols_plan(const GemmParams& p, int* n1_out, double* cost_out) {
    if (!p.ws || p.ln_stats || p.out_f32) return 0;
    const int nk = p.K / BK;
    const long tm = cdiv(p.M, GeomBig::BM), tn = cdiv(p.N, GeomBig::BN), tiles = tm * tn;
    const long rounds = tiles / 256;
    if (nk < 32 || rounds < 1 || tiles == rounds * 256 || tm > 256) return 0;
    const long c1 = rounds * 256 / tm;
    if (c1 <= 0 || c1 >= tn) return 0;
    const int n1 = (int)c1 * GeomBig::BN, nr = p.N - n1;
    const long ts = (long)cdiv(p.M, GeomSmall::BM) * cdiv(nr, GeomSmall::BN);
    if (ts > 224) return 0;
    int ks = (int)(512 / ts);
    if (ks > nk / 8) ks = nk / 8;
    if (ks > 8) ks = 8;
    while (ks >= 2 && (long long)ks * p.M * nr * 4 > p.ws_bytes) --ks;
    if (ks < 2) return 0;
    // cost in rounds of the big geometry: one round = 1.44 us per K step + ~9 us; a small block runs a K step in ~0.65 us alone
    // on its CU, ~0.91 us next to a second one; + fp32 tile stores and ramp, two kernel boundaries, the reduce pass at ~3 TB/s
    const double round_us = 1.44 * nk + 9.0;
    const double right_us = (double)cdiv(nk, ks) * (ts * ks > 256 ? 0.91 : 0.65) + 8.0 + 8.0 + (double)ks * p.M * nr * 4.0 / 3.0e6;
    long m1 = 0;
    *cost_out = plan_rows(p.M, n1, &m1) + right_us / round_us;
    *n1_out = n1;
    return ks;
}

bool gemm_fuses_norm(const GemmParams& p) {
    if (p.M <= 0 || emmax_tune().gemm_big >= 0 || emmax_tune().gemm_splitk == 0 || emmax_tune().gemm_normfuse == 0) return false;
    return splitk_norm_ok(p) && splitk_plan(p) != 0;
}

// The launch plan of launch_gemm, as data (gemm_plan_describe prints it: the CPU tests pin the plans of the shapes of the hot path).
//   GEOM    one geometry, forced by the tuning switch gemm_big
//   SPLITK  the whole problem K-split on 128 x 128 tiles (big: 256 x 256, for a very long K) + reduce pass (ks slices)
//   HYBRID  columns [0, n1): rows plan m1 (big tile rows, the rest small); columns [n1, N): K-split (ks) + reduce pass
//   COLS    columns [0, n1): rows plan m1; columns [n1, N): one all-small launch (half-empty last tile column)
//   ROWS    m1 big tile rows, the remaining rows in small tiles (m1 = all: one big launch; 0: one small launch)
struct GemmPlan { enum Kind { GEOM, SPLITK, HYBRID, COLS, ROWS } kind; int big, ks, n1; long m1; };

static GemmPlan plan_gemm(const GemmParams& p) {
    GemmPlan pl = {GemmPlan::ROWS, 0, 0, 0, 0};
    const int force = emmax_tune().gemm_big;   // 0 / 1: one geometry, no split
    if (force >= 0) { pl.kind = GemmPlan::GEOM; pl.big = force > 2 ? 1 : force; return pl; }   // 0 small, 1 big, 2 = 128 x 256 x 32
    const bool no_splitk = emmax_tune().gemm_splitk == 0;
    int sk_big = 0;
    if (const int ks = no_splitk ? 0 : splitk_plan(p, &sk_big)) {
        pl.kind = GemmPlan::SPLITK; pl.ks = ks; pl.big = sk_big;
        if (emmax_tune().gemm_sk_big >= 0 && (emmax_tune().gemm_sk_big == 0 || p.K / BK / ks >= 16)) pl.big = emmax_tune().gemm_sk_big;   // tools: geometry forced (big: >= 16 K steps per slice)
        return pl;
    }
    long m1 = 0;
    const double whole = plan_rows(p.M, p.N, &m1);
    if (!no_splitk && emmax_tune().gemm_hybrid != 0) {
        int hn1 = 0;
        double hcost = 0.0;
        const int hks = hybrid_cols_plan(p, &hn1, &hcost);
        if (hks && hcost < whole - 0.02) {
            pl.kind = GemmPlan::HYBRID; pl.ks = hks; pl.n1 = hn1;
            plan_rows(p.M, hn1, &pl.m1);
            return pl;
        }
    }
    // a half-empty last tile column (N = 1152, 3456: 4.5 / 13.5 big tiles wide) can go to the small geometry instead:
    // columns [0, n1) planned as above + columns [n1, N) as one all-small launch
    const int n1 = p.N / GeomBig::BN * GeomBig::BN;
    if (p.act != 2 && n1 > 0 && n1 < p.N) {
        long m1a = 0;
        const double left = plan_rows(p.M, n1, &m1a);
        // the narrow launch re-reads ALL of A for 128 columns of output: at long K it is bound by that stream, not by its tiles
        // (SigLIP fc2, K = 4352: 570 MB in 144 us measured against 67 us modelled -- the split was 7.6 % behind all-big,
        // `gemm_bench --ab gemm_big=-1,0,1`, round 4).  One big round takes ~1.44 us per K step + ~9 us (section 6); A streams at ~4.5 TB/s.
        const double round_us = 1.44 * (p.K / BK) + 9.0, stream_us = (double)p.M * p.K * 2.0 / 4.5e6;
        double right = cost_small((long)cdiv(p.M, GeomSmall::BM) * cdiv(p.N - n1, GeomSmall::BN));
        if (stream_us / round_us > right) right = stream_us / round_us;
        right += 0.03;
        if (left + right < whole - 1e-9) {
            pl.kind = GemmPlan::COLS; pl.n1 = n1; pl.m1 = m1a;
            return pl;
        }
    }
    pl.m1 = m1;
    return pl;
}

int gemm_plan_describe(const GemmParams& p, char* buf, int len) {
    if (p.M <= 0 || p.K % BK != 0 || p.N % 128 != 0 || p.K <= 0 || p.N <= 0) return -1;
    const GemmPlan pl = plan_gemm(p);
    const long tm = cdiv(p.M, GeomBig::BM);
    auto rows = [&](long m1, char* o, int n) {
        if (m1 >= tm) snprintf(o, n, "big");
        else if (m1 <= 0) snprintf(o, n, "small");
        else snprintf(o, n, "big rows 0..%ld + small rows %ld..%d", m1 * GeomBig::BM, m1 * GeomBig::BM, p.M);
    };
    char r[96];
    switch (pl.kind) {
        case GemmPlan::GEOM: snprintf(buf, len, "forced %s", pl.big == 2 ? "k32" : pl.big ? "big" : "small"); break;
        case GemmPlan::SPLITK: snprintf(buf, len, "splitk%s ks=%d%s", pl.big ? " big" : "", pl.ks, gemm_fuses_norm(p) ? " +norm" : ""); break;
        case GemmPlan::HYBRID: rows(pl.m1, r, sizeof r); snprintf(buf, len, "hybrid cols 0..%d: %s | cols %d..%d: splitk ks=%d", pl.n1, r, pl.n1, p.N, pl.ks); break;
        case GemmPlan::COLS: rows(pl.m1, r, sizeof r); snprintf(buf, len, "cols 0..%d: %s | cols %d..%d: small", pl.n1, r, pl.n1, p.N); break;
        case GemmPlan::ROWS: rows(pl.m1, r, sizeof r); snprintf(buf, len, "%s", r); break;
    }
    return (int)pl.kind;
}

int launch_gemm(const GemmParams& p, hipStream_t stream) {
    if (p.M <= 0) return 0;
    if (p.norm_out && !gemm_fuses_norm(p)) return -1;   // (the caller asks first)
    const GemmPlan pl = plan_gemm(p);
    if (pl.kind == GemmPlan::GEOM) return launch_gemm_geom(p, pl.big, stream);
    if (pl.kind == GemmPlan::SPLITK) return launch_gemm_splitk(p, pl.ks, stream, pl.big);
    if (pl.kind == GemmPlan::ROWS) return launch_planned_rows(p, pl.m1, stream);
    // column parts: a = columns [0, n1), b = columns [n1, N)
    const int n1 = pl.n1;
    GemmParams a = p, b = p;
    a.N = n1;
    a.N_store = p.N_store < n1 ? p.N_store : n1;
    b.N = p.N - n1;
    b.N_store = p.N_store > n1 ? p.N_store - n1 : 0;
    b.W = (const bf16_t*)p.W + (size_t)n1 * p.ldw;
    const int c1 = p.act == 2 ? n1 / 2 : n1;
    b.C = p.out_f32 ? (void*)((float*)p.C + c1) : (void*)((bf16_t*)p.C + c1);
    if (p.bias) b.bias = (const bf16_t*)p.bias + n1;
    if (p.scale) b.scale = (const bf16_t*)p.scale + n1;
    if (p.residual) b.residual = p.res_f32 ? (const void*)((const float*)p.residual + n1) : (const void*)((const bf16_t*)p.residual + n1);
    if (p.ln_stats) { b.ln_s = p.ln_s + n1; b.ln_c = p.ln_c + n1; }
    const int r = launch_planned_rows(a, pl.m1, stream);
    if (r) return r;
    if (pl.kind == GemmPlan::HYBRID) return (p.act == 2 || b.N_store > 0) ? launch_gemm_splitk(b, pl.ks, stream) : 0;
    return b.N_store > 0 ? launch_gemm_geom(b, 0, stream) : 0;
}
