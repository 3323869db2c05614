// decode_mfma.hip -- small-batch (3 <= B <= 8) decode projections on the matrix cores.
//
// At B >= 3 the per-lane dot-product GEMV of decode.hip stops being HBM-bound (B accumulators per row, B LDS reads per
// weight chunk, B wave reductions per row), so the batch is treated as what it is: a [B,K] x [K,N] GEMM with a tiny M.
// Weights are read from a second, MFMA-FRAGMENT-MAJOR copy built at finalize:
//     fm[((n/16)*(K/32) + k/32) * 64 + lane] = 8 bf16 = W[16*(n/16) + (lane&15)][32*(k/32) + 8*(lane>>4) .. +8]
// i.e. one wave load instruction = one contiguous 1 KiB tile = the A operand of one v_mfma_f32_16x16x32_bf16, straight
// from HBM to VGPRs (nt), no LDS round trip, no cross-lane reduction (tools/gemv_sweep.hip: 5.7 TB/s at B=8 vs 4.3 TB/s
// for MFMA over the row-major copy and 3.5 TB/s for the dot2 path).  The B activation rows (+1 zero row for the 16-B
// padding columns) sit in LDS with an odd 16-byte-slot pitch and are the B operand.
// A block (8 waves) owns one task (1 or 2 row tiles) at a time and splits K 8 ways; the partial tiles meet in LDS.
// Same fused prologues / epilogues as the GEMV: RMSNorm, attention split merge | RoPE + paged K/V append, +residual,
// SiLU*mul, greedy argmax.
#include <type_traits>

#include "common.h"
#include "kernels.h"

namespace {

enum { MODE_QKV = 0, MODE_RESID = 1, MODE_GATEUP = 2, MODE_LMHEAD = 3, MODE_PLAIN = 4 };
constexpr int PSTRIDE = EMMAX_PSTRIDE;
constexpr int MFMA_MAX_B = 8;   // this kernel stages at most 8 batch rows (batch 9-16 run on decode_km.hip only)
constexpr int GW = 8;
__device__ __forceinline__ u32x4_t ld_nt(const u32x4_t* p) { return __builtin_nontemporal_load(p); }

// row-major [N, ld] -> fragment-major tiles (N % 16 == 0, K % 32 == 0)
__global__ __launch_bounds__(256) void emmax_repack_fm_kernel(const bf16_t* __restrict__ src, int ld, u32x4_t* __restrict__ dst, int N, int K) {
    const size_t total = (size_t)N * K / 8;
    for (size_t c = (size_t)blockIdx.x * 256 + threadIdx.x; c < total; c += (size_t)gridDim.x * 256) {
        const int lane = (int)(c & 63);
        const size_t tile = c >> 6;
        const int kt = (int)(tile % (K / 32)), nt = (int)(tile / (K / 32));
        const int n = nt * 16 + (lane & 15), k = kt * 32 + (lane >> 4) * 8;
        dst[c] = *(const u32x4_t*)(src + (size_t)n * ld + k);
    }
}

// row-major bf16 [N, ld] -> fp8 e4m3 (OCP) fragment-major tiles + one fp32 scale per row (amax / 448).
//   fm8[((n/16)*(K/64) + k/64) * 64 + lane] = 16 bytes: W8[row][64*(k/64) + 8g .. +8] ++ W8[row][64*(k/64) + 32 + 8g .. +8]
//   with row = 16*(n/16) + (lane & 15), g = lane >> 4: one 16-byte load feeds two v_mfma_f32_16x16x32_bf16 k-steps.
// One block per row.  N % 16 == 0, K % 64 == 0.
// perm / head_dim: row order of the copy (km_src_row, common.h; 0 = natural): dst row n holds source row km_src_row(.., n / 16, n % 16)
__global__ __launch_bounds__(256) void emmax_quant_fm8_kernel(const bf16_t* __restrict__ src, int ld, uint8_t* __restrict__ dst,
                                                             float* __restrict__ scales, int N, int K, int perm, int head_dim) {
    const int n = blockIdx.x, tid = threadIdx.x;
    __shared__ float red[4];
    const bf16_t* row = src + (size_t)km_src_row(perm, head_dim, n >> 4, n & 15) * ld;
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
    const int nt = n >> 4, i16 = n & 15, KT2 = K / 64;
    for (int c = tid; c < K / 8; c += 256) {       // one 8-element (8-byte) group per thread
        const u32x4_t v = *(const u32x4_t*)(row + c * 8);
        const int k = c * 8, kt2 = k >> 6, kin = k & 63, half = kin >> 5, g = (kin & 31) >> 3;
        int lo = 0, hi = 0;
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[0]) / scale, bf_hi(v[0]) / scale, lo, false);
        lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[1]) / scale, bf_hi(v[1]) / scale, lo, true);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[2]) / scale, bf_hi(v[2]) / scale, hi, false);
        hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[3]) / scale, bf_hi(v[3]) / scale, hi, true);
        u32x2_t w = {(uint32_t)lo, (uint32_t)hi};
        *(u32x2_t*)(dst + (((size_t)nt * KT2 + kt2) * 64 + g * 16 + i16) * 16 + half * 8) = w;
    }
}

// 8 fp8 e4m3 (two dwords) -> 8 bf16 (exact: e4m3 fits bf16): v_cvt_scalef32_pk_bf16_fp8 with scale 1, two values per
// instruction.  (Through v_cvt_pk_f32_fp8 + repacking it was three to four instructions per pair, and VALU work is not free
// next to the MFMAs -- MFMA and VALU instructions of one SIMD do not overlap, tools/mfma_valu_overlap.hip -- the fp8
// projections were bound by this conversion, not by HBM.)
__device__ __forceinline__ bf16x8_t fp8x8_to_bf16x8(uint32_t a, uint32_t b) {
    typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2v;
    const bf16x2v p0 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a, 1.0f, false), p1 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(a, 1.0f, true);
    const bf16x2v p2 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(b, 1.0f, false), p3 = __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(b, 1.0f, true);
    u32x4_t r;
    r[0] = __builtin_bit_cast(uint32_t, p0);
    r[1] = __builtin_bit_cast(uint32_t, p1);
    r[2] = __builtin_bit_cast(uint32_t, p2);
    r[3] = __builtin_bit_cast(uint32_t, p3);
    return __builtin_bit_cast(bf16x8_t, r);
}

// o-proj prologue: merge the NS split partials of head (cg >> 4) for every batch row, two rows per iteration so the loads
// of both are in flight together; dst = LDS address of chunk c of row 0
template <int NS>
__device__ __forceinline__ void stage_attn_rows(const float* __restrict__ attn_part, unsigned char* dst, int pitch, int B, int Hq, int cg) {
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

// FP8: weights are the fp8 fragment-major copy (1 KiB tile = 16 rows x 64 k), de-quantised to bf16 in registers
// (exact), per-row scale applied to the fp32 result; activations stay bf16.  A "k-step" is then 64 elements.
#ifdef DECODE_LAB_TRACE
__device__ unsigned long long g_dec_trace[256 * 8];   // [block][stamp]: s_memrealtime (100 MHz) of wave 0
#define DEC_STAMP(k) do { if (threadIdx.x == 0) g_dec_trace[blockIdx.x * 8 + (k)] = wall_clock64(); } while (0)
#else
#define DEC_STAMP(k) do { } while (0)
#endif
template <int MODE, bool NORM, bool XATTN, bool FP8 = false>
__global__ __launch_bounds__(GW * 64, 2) void emmax_decode_mfma_kernel(GemvParams p) {
    DEC_STAMP(0);
    constexpr int TILES = (MODE == MODE_QKV || MODE == MODE_GATEUP) ? 2 : 1;
    // k-steps per ring block: 8 KiB of weights in flight per wave (64 KiB per CU, 16 MiB on the chip) is the measured optimum
    // once the ring really rolls -- 16 KiB per wave costs 1-3 us per launch, 32 KiB up to 20 us (the first burst alone is then
    // 64 MB: everything else, the activations of the prologue included, queues behind it)
    constexpr int U = TILES == 2 ? 4 : 8;
    constexpr int NT = GW * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int B = p.batch;
    const int K = p.K, KC = p.kc;
    {   // every kernel argument the start-up path needs in ONE scalar-load round (hipcc otherwise fetches them where they are
        // first used: five dependent round trips, ~1 us, in front of the first weight request)
        const int a0 = p.n_groups, a1 = p.sk_kt8, a2 = p.sk_q, a3 = p.sk_r, a4 = p.qk_shift, a5 = p.head_dim;
        const unsigned a6 = p.sk_magic;
        const void* a7 = p.W;
        const void* a8 = p.x;
        const void* a9 = p.norm_w;
        asm volatile("" ::"s"(B), "s"(K), "s"(KC), "s"(a0), "s"(a1), "s"(a2), "s"(a3), "s"(a4), "s"(a5), "s"(a6), "s"(a7), "s"(a8), "s"(a9));
    }
    const int pitch = KC * 2 + 16;                               // bytes per staged x row
    constexpr int RSZ = GW * TILES * 256;                       // floats of one reduction buffer
    float* red = (float*)(smem + (size_t)(B + 1) * pitch);      // [2][GW][TILES][4][64], double-buffered (deferred reduction)
    float* srstd = red + 2 * RSZ;                               // [B]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g4 = lane >> 4, c16 = lane & 15;
    constexpr int KS = FP8 ? 64 : 32;                            // elements per k-step (one 16-byte load per lane)
    const int KT = K / KS;                                       // k-steps of the whole row
    const int n_tasks = p.n_groups;
    const int G = gridDim.x, bid = blockIdx.x;
    const int n_phase = (K + KC - 1) / KC;
    // Work split.  A task's K range is KT8 = KT / 8 super-steps (one k-step per wave).  Plain split: whole tasks per block --
    // 384 qkv tasks on 256 blocks means half the blocks stream twice as much as the others.  STREAM-K split (single K phase,
    // n_tasks >= grid): every block owns the same number of super-steps of the linearised (task, super-step) sequence, so a
    // task may straddle two consecutive blocks.  The block that holds the END of the task's K range (it is the first thing
    // that block does) publishes its partial tile sums as data-tagged 8-byte granules {f32, tag} (agent-scope write-through
    // stores, no flag, no fence); the block that holds the START adds them to its own sums, clears the granules for the next
    // launch and runs the epilogue.  It takes that task just before its last whole task, so the wait (if any) and the
    // granule round trip hide under the weight stream instead of sitting at the tail of the block.
    // The launcher decides (sk_kt8 > 0: stream-K) and precomputes the per-block share q, r and the magic multiplier of the
    // division by KT8, so the start-up path has no integer division.
    const bool sk = p.sk_kt8 > 0;
    const int KT8 = sk ? p.sk_kt8 : 1;
    auto divk = [&](int n) { return (int)__umulhi((unsigned)n, p.sk_magic); };   // n / KT8, exact for n < 2^32 / KT8
    const int u_lo = bid * p.sk_q + min(bid, p.sk_r);            // sk: super-step range; else task range
    const int u_hi = u_lo + p.sk_q + (bid < p.sk_r ? 1 : 0);
    const int t_first = sk ? divk(u_lo) : u_lo;
    const int n_seg = u_hi <= u_lo ? 0 : (sk ? divk(u_hi - 1) - t_first + 1 : u_hi - u_lo);
    const bool head_part = sk && u_lo != t_first * KT8, tail_part = sk && u_hi != divk(u_hi) * KT8;
    // processing order: natural, except that a partial tail segment swaps places with the whole task before it
    const bool swap_tail = n_seg >= 2 && tail_part && !(n_seg == 2 && head_part);
    struct Seg { int t, gb, ge; };   // task, super-step range [gb, ge) of its K range
    auto seg_of = [&](int i) {
        Seg sg;
        if (!sk) {
            sg.t = min(u_lo + i, n_tasks - 1); sg.gb = 0; sg.ge = KT8;
            return sg;
        }
        int j = i;
        if (swap_tail) j = (i == n_seg - 2) ? n_seg - 1 : ((i == n_seg - 1) ? n_seg - 2 : i);
        j = min(j, n_seg - 1);
        sg.t = t_first + j;
        sg.gb = (j == 0) ? u_lo - t_first * KT8 : 0;
        sg.ge = min(KT8, u_hi - sg.t * KT8);
        return sg;
    };
    const u32x4_t* __restrict__ Wfm = (const u32x4_t*)p.W;

    // tile index (16-row units) of tile tt of task t
    auto tile_of = [&](int t, int tt) {
        if (MODE == MODE_QKV) {
            const int halfb = 1 << p.qk_shift;                               // head_dim / 32: 8 tiles per head, 4 low + 4 high
            const int hb = t >> p.qk_shift, db = t & (halfb - 1);
            return hb * 2 * halfb + db + tt * halfb;
        }
        return t * TILES + tt;
    };
    // this wave's k-step range inside phase ph
    auto slice = [&](const Seg& sg, int ph, int& k_lo, int& k_n) {
        if (sk) {   // every wave takes (ge - gb) consecutive k-steps of the segment
            k_n = sg.ge - sg.gb;
            k_lo = sg.gb * GW + wave * k_n;
            return;
        }
        const int kt0 = ph * (KC / KS), ktn = min(KC, K - ph * KC) / KS;
        const int q = ktn / GW, r = ktn % GW;
        k_lo = kt0 + wave * q + min(wave, r);
        k_n = q + (wave < r ? 1 : 0);
    };

    // The wave's work is a linear sequence of U-step blocks: for every task x K phase x block of its k-slice.  The producer
    // cursor runs exactly one block ahead of the consumer: right after the MFMAs of step u are issued, step u of the next
    // block is requested into the same registers (rolling ring, TILES*U KiB in flight per wave at every instant, across
    // task and phase boundaries and underneath the block barriers of the reduction).
    struct Cursor { int i, ph, kb; Seg sg; };   // segment index (processing order), K phase, U-block; sg = seg_of(i)
    // block-uniform trip count: the longest k-slice of the phase
    auto nkb_of = [&](const Seg& sg, int ph) {
        if (sk) return (sg.ge - sg.gb + U - 1) / U;
        const int ktn = min(KC, K - ph * KC) / KS;
        return (ktn / GW + (ktn % GW ? 1 : 0) + U - 1) / U;
    };
    auto advance = [&](Cursor& c) {
        if (++c.kb >= nkb_of(c.sg, c.ph)) {
            c.kb = 0;
            if (++c.ph >= n_phase) {
                c.ph = 0;
                ++c.i;
                c.sg = seg_of(c.i);
            }
        }
    };
    u32x4_t wr[TILES][U];
    const u32x4_t* wbase[TILES];   // producer: tile base + slice start of its current (task, phase)
    int p_n = 0;                   // producer: k-steps in its slice
    auto producer_setup = [&](const Cursor& c) {
        int k_lo;
        slice(c.sg, c.ph, k_lo, p_n);
#pragma unroll
        for (int tt = 0; tt < TILES; ++tt) wbase[tt] = Wfm + ((size_t)tile_of(c.sg.t, tt) * KT + k_lo) * 64 + lane;
    };
    // UNCONDITIONAL loads (callers guarantee c.i < n_seg): a predicated load leaves hipcc's s_waitcnt bookkeeping one load
    // short at the merge point, every MFMA then waits for vmcnt(0) -- i.e. for the refill issued one step earlier -- and the
    // "ring" degenerates to one k-step in flight per wave (a latency-bound kernel: 0.84 us per step, 4.9 TB/s).  Steps past the
    // end of the slice re-read its last step (an L2 hit) and are never consumed.
    auto issue_step = [&](const Cursor& c, int u) {
        const int k = min(c.kb * U + u, p_n - 1);
#pragma unroll
        for (int tt = 0; tt < TILES; ++tt) wr[tt][u] = ld_nt(wbase[tt] + (size_t)k * 64);
    };

    if (n_seg == 0) return;   // grid <= n_groups: cannot happen; keeps the unconditional loads below in bounds
    Cursor P = {0, 0, 0, seg_of(0)}, C = P;
    // head of the stream: requested before the prologue computes anything -- but AFTER the prologue's own loads where those
    // are a fixed handful (one-pass): loads return in order, so x queued behind 16 KiB of weights per wave would keep the
    // whole prologue waiting for HBM
    auto issue_head = [&]() {
        // x_bar: every wave of the block (one block per CU) has REQUESTED its activation loads before any wave requests weights --
        // the CU's vector-memory path serves requests in arrival order, so activations queued behind other waves' 8 KiB heads
        // arrived 3-4 us into the launch (a bare s_barrier: it orders the requests, it does not wait for data)
        if (p.x_bar) __builtin_amdgcn_s_barrier();
        producer_setup(P);
#pragma unroll
        for (int u = 0; u < U; ++u) issue_step(P, u);
        advance(P);
    };

    // ---- prologue: x (B rows + one zero row) into LDS.  Every variant issues all of a thread's loads before the first
    // use: B sequential L2 round trips (one per row) used to cost 12-16 us of every launch at B = 8 ----
    // one pass (norm'd projections, K/8 <= 512): thread c keeps chunk c of every row in registers, the statistics come
    // from those registers and the normalised rows go straight to LDS
    const bool one_pass = NORM && !XATTN && n_phase == 1 && (K >> 3) <= NT;
    if (one_pass) {
        __shared__ __attribute__((aligned(16))) float rsum1[GW][MFMA_MAX_B];
        const bool mine = tid < (K >> 3);
        u32x4_t xv[MFMA_MAX_B];
        // unconditional (clamped) loads, masked afterwards: hipcc can then count them and waits for exactly these nine loads
        // -- behind predicated loads it waited for the whole first ring as well (the prologue ended when 32 MB of weights had
        // landed, ~10 us into the launch, with HBM idle for half of that)
        const int ct = min(tid, (K >> 3) - 1);
        u32x4_t wv = *((const u32x4_t*)p.norm_w + ct);
#pragma unroll
        for (int b = 0; b < MFMA_MAX_B; ++b)
            xv[b] = *((const u32x4_t*)((const bf16_t*)p.x + (size_t)min(b, B - 1) * p.ldx) + ct);
        issue_head();
#pragma unroll
        for (int b = 0; b < MFMA_MAX_B; ++b) {
            if (b >= B) break;   // block-uniform: rows the batch does not have cost nothing (fp8 runs this kernel at B = 1 too)
            float ss = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                xv[b][j] = mine ? xv[b][j] : 0u;
                const float a = bf_lo(xv[b][j]), bb = bf_hi(xv[b][j]);
                ss += a * a + bb * bb;
            }
            ss = wave_sum(ss);
            if (lane == 0) rsum1[wave][b] = ss;
        }
        __syncthreads();
        // the 8 x 8 wave partials as sixteen broadcast 16-byte LDS reads per thread (64 scalar reads cost ~2 us)
        static_assert(MFMA_MAX_B == 8, "row statistics are read as two float4 per wave");
        float tot[MFMA_MAX_B];
#pragma unroll
        for (int b = 0; b < MFMA_MAX_B; ++b) tot[b] = 0.f;
#pragma unroll
        for (int w = 0; w < GW; ++w) {
            const f32x4_t lo = *(const f32x4_t*)&rsum1[w][0], hi = *(const f32x4_t*)&rsum1[w][4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                tot[j] += lo[j];
                tot[4 + j] += hi[j];
            }
        }
#pragma unroll
        for (int b = 0; b < MFMA_MAX_B; ++b) {
            if (b < B) {   // block-uniform: a scalar branch, no exec-mask juggling per row
                const float rs = rsqrtf(tot[b] / (float)K + p.eps);
                u32x4_t v = xv[b];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    // HF LlamaRMSNorm: fp32 normalise -> downcast -> * weight (-> downcast); both halves of a pair go
                    // through ONE v_cvt_pk_bf16_f32 per rounding
                    const uint32_t r = pack_bf16x2(bf_lo(v[j]) * rs, bf_hi(v[j]) * rs);
                    v[j] = pack_bf16x2(bf_lo(r) * bf_lo(wv[j]), bf_hi(r) * bf_hi(wv[j]));
                }
                if (mine) *(u32x4_t*)(smem + (size_t)b * pitch + (size_t)tid * 16) = v;
            }
        }
        if (mine) *(u32x4_t*)(smem + (size_t)B * pitch + (size_t)tid * 16) = (u32x4_t){0u, 0u, 0u, 0u};
    } else if (NORM) {
        __shared__ float rsum[GW][MFMA_MAX_B];
        for (int b = 0; b < B; ++b) {
            float ss = 0.f;
            const u32x4_t* xr = (const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx);
            for (int c = tid; c < (K >> 3); c += NT) {
                const u32x4_t v = xr[c];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float a = bf_lo(v[j]), bb = bf_hi(v[j]);
                    ss += a * a + bb * bb;
                }
            }
            ss = wave_sum(ss);
            if (lane == 0) rsum[wave][b] = ss;
        }
        __syncthreads();
        if (tid < B) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < GW; ++w) t += rsum[w][tid];
            srstd[tid] = rsqrtf(t / (float)K + p.eps);
        }
        __syncthreads();
    }

    auto stage_x = [&](int ph, auto head_tag) {
        constexpr bool HEAD = decltype(head_tag)::value;   // also start the weight stream (first phase)
        const int kc0 = ph * KC, nch = min(KC, K - kc0) >> 3;
        if (XATTN) {
            if (HEAD) issue_head();
            // x = merged attention split partials; chunk cg = head (cg >> 4), elements (cg & 15) * 8 .. +8
            for (int c = tid; c < nch; c += NT) {
                const int cg = (kc0 >> 3) + c;
                unsigned char* dst = smem + (size_t)c * 16;
                switch (p.nsplit) {
                    case 1: stage_attn_rows<1>(p.attn_part, dst, pitch, B, p.Hq, cg); break;
                    case 2: stage_attn_rows<2>(p.attn_part, dst, pitch, B, p.Hq, cg); break;
                    case 4: stage_attn_rows<4>(p.attn_part, dst, pitch, B, p.Hq, cg); break;
                    case 8: stage_attn_rows<8>(p.attn_part, dst, pitch, B, p.Hq, cg); break;
                    default:
                        for (int b = 0; b < B; ++b)
                            *(u32x4_t*)(dst + (size_t)b * pitch) =
                                attn_merge_chunk_loop(p.attn_part + (size_t)(b * p.Hq + (cg >> 4)) * p.nsplit * PSTRIDE, (cg & 15) * 8, p.nsplit);
                }
                *(u32x4_t*)(dst + (size_t)B * pitch) = (u32x4_t){0u, 0u, 0u, 0u};
            }
        } else if (!NORM) {
            // flattened (row, chunk) space, SX unconditional (clamped) loads in flight per thread, masked at the store: one L2
            // round trip stages a whole phase of the down projection at B = 8 (688 chunks x 8 rows = 10.75 per thread); the
            // predicated four-at-a-time form took three, each behind a full vmcnt(0).  HEAD (first phase): the weight stream
            // starts after these loads are queued, not in front of them
            constexpr int SX = 12;
            for (int c = tid; c < nch; c += NT) *(u32x4_t*)(smem + (size_t)B * pitch + (size_t)c * 16) = (u32x4_t){0u, 0u, 0u, 0u};
            const int total = B * nch;
            const float inv = 1.0f / (float)nch;
            for (int base = 0; base < total; base += SX * NT) {   // block-uniform
                u32x4_t v[SX];
                int off[SX];
#pragma unroll
                for (int j = 0; j < SX; ++j) {
                    const int i = base + tid + j * NT, ic = min(i, total - 1);
                    int b = (int)(((float)ic + 0.5f) * inv), c = ic - b * nch;   // float quotient, one step of correction
                    if (c < 0) { --b; c += nch; }
                    if (c >= nch) { ++b; c -= nch; }
                    off[j] = i < total ? (int)(b * pitch + c * 16) : -1;
                    v[j] = *((const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx + kc0) + c);
                }
                if (HEAD && base == 0) { DEC_STAMP(6); issue_head(); DEC_STAMP(7); }
#pragma unroll
                for (int j = 0; j < SX; ++j)
                    if (off[j] >= 0) *(u32x4_t*)(smem + off[j]) = v[j];
            }
        } else {
            if (HEAD) issue_head();
            // flattened (row, chunk) space, four loads in flight per thread
            const int total = (B + 1) * nch;
            for (int i0 = tid; i0 < total; i0 += 4 * NT) {
                u32x4_t v[4];
                int off[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int i = i0 + j * NT;
                    const int b = i / nch, c = i - b * nch;
                    off[j] = i < total ? (int)(b * pitch + c * 16) : -1;
                    v[j] = (i < total && b < B) ? *((const u32x4_t*)((const bf16_t*)p.x + (size_t)b * p.ldx + kc0) + c) : (u32x4_t){0u, 0u, 0u, 0u};
                    if (NORM && i < total && b < B) {
                        const u32x4_t wv = *((const u32x4_t*)((const bf16_t*)p.norm_w + kc0) + c);
                        const float rs = srstd[b];
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const float a = bf2f(f2bf(bf_lo(v[j][q]) * rs)) * bf_lo(wv[q]);
                            const float bb = bf2f(f2bf(bf_hi(v[j][q]) * rs)) * bf_hi(wv[q]);
                            v[j][q] = pack_bf16x2(a, bb);
                        }
                    }
                }
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (off[j] >= 0) *(u32x4_t*)(smem + off[j]) = v[j];
            }
        }
    };
    if (!one_pass) stage_x(0, std::true_type{});
    DEC_STAMP(1);
    __syncthreads();
    DEC_STAMP(2);

    // LMHEAD: per-thread running best of the (row slot, batch) it finalises
    float best = -INFINITY;
    int besti = 0x7fffffff;

    // ---- deferred cross-wave reduction + epilogue of a finished segment (waves 0-3; rb = its reduction buffer) ----
    // Epilogue operands that have to be LOADED (RoPE cos / sin, the old residual value) are fetched when the segment's
    // partial sums are written, one U-block before they are used; position and page id once per launch.
    const int ec = tid & 15;   // batch column this thread finalises (tid < 256)
    int pre_pos = 0, pre_pg = 0;
    float pf_a = 0.f, pf_b = 0.f;
    if (MODE == MODE_QKV && tid < 256 && ec < B) {
        pre_pos = p.ctx_len[ec];
        pre_pg = p.page_table[(size_t)ec * p.max_pages + pre_pos / p.page];
    }
    auto prefetch_epilogue = [&](const Seg& cs) {
        if (tid >= 256 || ec >= B || cs.gb != 0) return;   // segments that only publish partial sums have no epilogue
        const int row_in = 4 * ((tid & 63) >> 4) + (tid >> 6);
        if (MODE == MODE_RESID) {
            pf_a = p.h32 ? p.h32[(size_t)ec * p.ldh + cs.t * 16 + row_in] : bf2f(((const bf16_t*)p.y)[(size_t)ec * p.ldy + cs.t * 16 + row_in]);
        } else if (MODE == MODE_QKV) {
            const int half = p.head_dim >> 1;
            const int hb = cs.t >> p.qk_shift, d = (cs.t & ((1 << p.qk_shift) - 1)) * 16 + row_in;
            if (hb < p.Hq + p.Hkv) {
                pf_a = p.cos_t[(size_t)pre_pos * half + d];
                pf_b = p.sin_t[(size_t)pre_pos * half + d];
            }
        }
    };
    auto reduce_epilogue = [&](const Seg& cs, const float* rb) {
        const int t = cs.t;
        if (tid < 256) {
            const int l = tid & 63, r = tid >> 6;
            const int c = l & 15, row_in = 4 * (l >> 4) + r;       // batch column, row inside the tile
            float v[TILES];
#pragma unroll
            for (int tt = 0; tt < TILES; ++tt) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < GW; ++w) s += rb[((w * TILES + tt) * 4 + r) * 64 + l];
                v[tt] = s;
            }
            bool finalise = true;
            if (sk && cs.ge - cs.gb < KT8) {   // a task shared with a neighbour block (see the work split above)
                constexpr unsigned long long TAG = 0x7a6b5c4dull << 32;
                if (cs.gb > 0) {   // END of the K range: publish, the previous block finalises
                    unsigned long long* gr = p.sk_ws + ((size_t)(bid - 1) * TILES) * 256 + tid;
#pragma unroll
                    for (int tt = 0; tt < TILES; ++tt)
                        __hip_atomic_store(gr + tt * 256, TAG | (unsigned long long)__float_as_uint(v[tt]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    finalise = false;
                } else {           // START of the K range: add the next block's partial sums, clear the granules
                    unsigned long long* gr = p.sk_ws + ((size_t)bid * TILES) * 256 + tid;
#pragma unroll
                    for (int tt = 0; tt < TILES; ++tt) {
                        unsigned long long g = __hip_atomic_load(gr + tt * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        unsigned int spins = 0;
                        while ((g & 0xffffffff00000000ull) != TAG) {
                            __builtin_amdgcn_s_sleep(2);
                            if (++spins > (1u << 22)) __builtin_trap();   // a broken hand-off must never become a silent wrong answer
                            g = __hip_atomic_load(gr + tt * 256, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        }
                        v[tt] += __uint_as_float((unsigned int)g);
                        __hip_atomic_store(gr + tt * 256, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
            }
            if (FP8) {
#pragma unroll
                for (int tt = 0; tt < TILES; ++tt) v[tt] *= p.wscale[tile_of(t, tt) * 16 + row_in];
            }
            if (finalise && c < B) {
                if (MODE == MODE_PLAIN) {
                    ((bf16_t*)p.y)[(size_t)c * p.ldy + t * 16 + row_in] = f2bf(v[0]);
                } else if (MODE == MODE_RESID) {
                    // fp32 residual stream (GemvParams::h32): the master copy + its bf16 mirror (what the NORM modes read)
                    if (p.h32) p.h32[(size_t)c * p.ldh + t * 16 + row_in] = pf_a + v[0];
                    bf16_t* hp = (bf16_t*)p.y + (size_t)c * p.ldy + t * 16 + row_in;
                    *hp = f2bf(pf_a + v[0]);
                } else if (MODE == MODE_GATEUP) {
                    ((bf16_t*)p.y)[(size_t)c * p.ldy + t * 16 + row_in] = f2bf(silu(v[0]) * v[TILES - 1]);
                } else if (MODE == MODE_QKV) {
                    const int hd = p.head_dim, half = hd >> 1;
                    const int hb = t >> p.qk_shift, d = (t & ((1 << p.qk_shift) - 1)) * 16 + row_in;
                    const int pos = pre_pos;
                    const float x0 = bf2f(f2bf(v[0])), x1 = bf2f(f2bf(v[TILES - 1]));
                    if (hb < p.Hq + p.Hkv) {
                        const float cs = pf_a, sn = pf_b;
                        const bf16_t y0 = f2bf(x0 * cs - x1 * sn), y1 = f2bf(x1 * cs + x0 * sn);
                        if (hb < p.Hq) {
                            bf16_t* q = (bf16_t*)p.y + (size_t)c * p.ldy + hb * hd;
                            q[d] = y0;
                            q[d + half] = y1;
                        } else {
                            const int pg = pre_pg;
                            bf16_t* kc = gemv_kv_row(p, false, c, pg, pos, hb - p.Hq);
                            kc[d] = y0;
                            kc[d + half] = y1;
                        }
                    } else {
                        const int pg = pre_pg;
                        bf16_t* vc = gemv_kv_row(p, true, c, pg, pos, hb - p.Hq - p.Hkv);
                        vc[d] = f2bf(x0);
                        vc[d + half] = f2bf(x1);
                    }
                } else if (MODE == MODE_LMHEAD) {
                    const int row = t * 16 + row_in;
                    if (row < p.n_rows) {
                        if (v[0] > best || (v[0] == best && row < besti)) {
                            best = v[0];
                            besti = row;
                        }
                        if (p.logits_out) p.logits_out[(size_t)c * p.n_rows + row] = v[0];
                    }
                }
            }
        }
    };

    const int xrow = c16 < B ? c16 : B;   // padding columns of the 16-wide batch side read the zero row
    f32x4_t acc[TILES];
#pragma unroll
    for (int tt = 0; tt < TILES; ++tt) acc[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
    int c_n = 0;
    const unsigned char* xb = smem;
    bool pend = false;
    Seg pend_sg = {0, 0, 0};
    int par = 0, pend_par = 0;
    while (C.i < n_seg) {
        if (n_phase > 1 && C.kb == 0 && (C.ph != 0 || C.i != 0)) {   // new phase: restage x (the ring keeps flying)
            __syncthreads();
            stage_x(C.ph, std::false_type{});
            __syncthreads();
        }
        if (C.kb == 0) {
            int k_lo;
            slice(C.sg, C.ph, k_lo, c_n);
            xb = smem + (size_t)xrow * pitch + ((size_t)(k_lo - C.ph * (KC / KS)) * KS + g4 * 8) * 2;
        }
        if (P.kb == 0 && P.i < n_seg) producer_setup(P);
        // one U-block: consume step u, then refill its registers with step u of the NEXT block (unconditionally -- the last
        // block of the wave's work runs the copy of the loop without loads)
        auto run_block = [&](auto refill_tag) {
            constexpr bool REFILL = decltype(refill_tag)::value;
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int k = C.kb * U + u;
                if (k < c_n) {   // wave-uniform
                    const bf16x8_t xf = *(const bf16x8_t*)(xb + (size_t)k * (KS * 2));
                    if (FP8) {
                        const bf16x8_t xf2 = *(const bf16x8_t*)(xb + (size_t)k * (KS * 2) + 64);
#pragma unroll
                        for (int tt = 0; tt < TILES; ++tt) {
                            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16x8(wr[tt][u][0], wr[tt][u][1]), xf, acc[tt], 0, 0, 0);
                            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fp8x8_to_bf16x8(wr[tt][u][2], wr[tt][u][3]), xf2, acc[tt], 0, 0, 0);
                        }
                    } else {
#pragma unroll
                        for (int tt = 0; tt < TILES; ++tt)
                            acc[tt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wr[tt][u]), xf, acc[tt], 0, 0, 0);
                    }
                }
                if (REFILL) issue_step(P, u);
            }
        };
        if (P.i < n_seg) run_block(std::true_type{}); else run_block(std::false_type{});
#ifdef DECODE_LAB_TRACE
        if (C.i == 0 && C.ph == 0 && C.kb == 0) DEC_STAMP(3);
#endif
        const bool task_done = (C.ph == n_phase - 1) && (C.kb == nkb_of(C.sg, C.ph) - 1);
        const Seg cs = C.sg;
        const int t = cs.t;
        advance(C);
        advance(P);
        // ---- partial tile sums -> red[par][wave][tile][r][lane]; the cross-wave reduction and the epilogue run one U-block
        // later (after the refills of the next segment are in flight): a barrier + epilogue right here kept every wave of the
        // block from issuing loads for 0.7-2 us per task ----
        if (pend) {
            __syncthreads();
            reduce_epilogue(pend_sg, red + pend_par * RSZ);
            pend = false;
        }
        if (!task_done) continue;
#pragma unroll
        for (int tt = 0; tt < TILES; ++tt)
#pragma unroll
            for (int r = 0; r < 4; ++r) red[par * RSZ + ((wave * TILES + tt) * 4 + r) * 64 + lane] = acc[tt][r];
#pragma unroll
        for (int tt = 0; tt < TILES; ++tt) acc[tt] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        prefetch_epilogue(cs);
        pend = true;
        pend_sg = cs;
        pend_par = par;
        par ^= 1;
    }
    DEC_STAMP(4);
    if (pend) {
        __syncthreads();
        reduce_epilogue(pend_sg, red + pend_par * RSZ);
    }
    DEC_STAMP(5);

    if (MODE == MODE_LMHEAD) {
        __syncthreads();                      // the last reduction has read red[]
        float* bv = red;                      // [256]
        int* bi = (int*)(red + 256);          // [256]
        if (tid < 256) {
            bv[tid] = best;
            bi[tid] = besti;
        }
        __syncthreads();
        if (tid < B) {   // the 16 (row slot) entries of batch column tid: l = g*16 + tid, r = 0..3
            float v0 = -INFINITY;
            int i0 = 0x7fffffff;
            for (int g = 0; g < 4; ++g)
                for (int r = 0; r < 4; ++r) {
                    const int e = r * 64 + g * 16 + tid;
                    const float v = bv[e];
                    const int ii = bi[e];
                    if (v > v0 || (v == v0 && ii < i0)) {
                        v0 = v;
                        i0 = ii;
                    }
                }
            p.part_val[(size_t)blockIdx.x * B + tid] = v0;
            p.part_idx[(size_t)blockIdx.x * B + tid] = i0;
        }
    }
}

}  // namespace

int launch_repack_fm(const void* src, int ld, void* dst, int N, int K, hipStream_t stream) {
    if (N % 16 || K % 32 || ld % 8) return -1;
    hipLaunchKernelGGL(emmax_repack_fm_kernel, dim3(2048), dim3(256), 0, stream, (const bf16_t*)src, ld, (u32x4_t*)dst, N, K);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

static size_t mfma_smem(int B, int kc, int tiles) { return (size_t)(B + 1) * (kc * 2 + 16) + (size_t)2 * GW * tiles * 256 * 4 + 64; }

// K phase length: multiple of 32, activations (B+1 rows) + reduction buffer within ~150 KiB
static int mfma_kc(int B, int K, int tiles, int kstep) {
    const size_t budget = 150 * 1024 - (size_t)2 * GW * tiles * 256 * 4 - 64;
    int cap = (int)(budget / (B + 1) - 16) / 2;
    cap = cap / kstep * kstep;
    if (K <= cap) return K;
    const int nph = cdiv(K, cap);
    return cdiv(cdiv(K, nph), kstep) * kstep;
}

template <int MODE, bool NORM, bool XATTN, bool FP8>
static int launch_mfma_t(GemvParams p, int B, hipStream_t stream, int* grid_out) {
    constexpr int TILES = (MODE == MODE_QKV || MODE == MODE_GATEUP) ? 2 : 1;
    if (FP8 && !p.wscale) return -1;
    if (p.K % (FP8 ? 64 : 32) || p.n_rows % (16 * TILES)) {
        if (!(MODE == MODE_LMHEAD && p.n_rows % 16 == 0)) return -1;
    }
    p.batch = B;
    p.x_bar = XATTN ? 0 : emmax_tune().mfma_xbar;   // default on for the prologues that load activations ahead of the head
    p.kc = mfma_kc(B, p.K, TILES, FP8 ? 64 : 32);
    if (NORM && p.kc != p.K) return -1;
    p.n_groups = p.n_rows / (16 * TILES);
    int grid = min(256, p.n_groups);
    if (MODE == MODE_LMHEAD) grid = min(grid, p.max_parts);
    if (grid_out) *grid_out = grid;
    if (MODE == MODE_QKV) {
        p.qk_shift = 0;
        while ((32 << p.qk_shift) < p.head_dim) ++p.qk_shift;
        if ((32 << p.qk_shift) != p.head_dim) return -1;   // head_dim / 32 must be a power of two
    }
    {   // work split (see the kernel): stream-K when it applies, whole tasks otherwise
        const int KT = p.K / (FP8 ? 64 : 32);
        const bool sk = p.sk_ws != nullptr && p.kc == p.K && KT % GW == 0 && KT / GW >= 2 && p.n_groups >= grid &&
                        (long long)p.n_groups * KT < (1ll << 31) / (KT / GW);
        p.sk_kt8 = sk ? KT / GW : 0;
        const int total = sk ? p.n_groups * p.sk_kt8 : p.n_groups;
        p.sk_q = total / grid;
        p.sk_r = total % grid;
        p.sk_magic = sk ? (unsigned)(((1ull << 32) + (unsigned)p.sk_kt8 - 1) / (unsigned)p.sk_kt8) : 0u;
    }
    const size_t smem = mfma_smem(B, p.kc, TILES);
    hipLaunchKernelGGL((emmax_decode_mfma_kernel<MODE, NORM, XATTN, FP8>), dim3(grid), dim3(GW * 64), smem, stream, p);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

template <bool FP8>
static int launch_mfma_mode(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    switch (mode) {
        case MODE_QKV: return launch_mfma_t<MODE_QKV, true, false, FP8>(p, B, stream, grid_out);
        case MODE_RESID:
            return p.attn_part ? launch_mfma_t<MODE_RESID, false, true, FP8>(p, B, stream, grid_out) : launch_mfma_t<MODE_RESID, false, false, FP8>(p, B, stream, grid_out);
        case MODE_GATEUP: return launch_mfma_t<MODE_GATEUP, true, false, FP8>(p, B, stream, grid_out);
        case MODE_LMHEAD: return launch_mfma_t<MODE_LMHEAD, true, false, FP8>(p, B, stream, grid_out);
        case MODE_PLAIN: return launch_mfma_t<MODE_PLAIN, false, false, FP8>(p, B, stream, grid_out);
        default: return -1;
    }
}

int launch_decode_mfma(int mode, const GemvParams& p, int B, hipStream_t stream, int* grid_out) {
    if (B < 1 || B > MFMA_MAX_B) return -1;
    return p.wscale ? launch_mfma_mode<true>(mode, p, B, stream, grid_out) : launch_mfma_mode<false>(mode, p, B, stream, grid_out);
}

int launch_quant_fm8(const void* src, int ld, void* dst, float* scales, int N, int K, hipStream_t stream, int perm, int head_dim) {
    if (N % 16 || K % 64 || ld % 8) return -1;
    if ((perm == 1 && (head_dim % 16 || head_dim < 16 || N % head_dim)) || (perm == 2 && N % 32) || perm < 0 || perm > 2) return -1;
    hipLaunchKernelGGL(emmax_quant_fm8_kernel, dim3(N), dim3(256), 0, stream, (const bf16_t*)src, ld, (uint8_t*)dst, scales, N, K, perm, head_dim);
    return hipGetLastError() == hipSuccess ? 0 : -4;
}

int decode_mfma_init() {
    static int done = -1;
    if (done == 0) return 0;
    const int lim = 160 * 1024 - 4096;
    hipError_t e = hipSuccess;
#define SET(M, N_, X)                                                                                                          \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_mfma_kernel<M, N_, X, false>, hipFuncAttributeMaxDynamicSharedMemorySize, lim); \
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)emmax_decode_mfma_kernel<M, N_, X, true>, hipFuncAttributeMaxDynamicSharedMemorySize, lim)
    SET(MODE_QKV, true, false); SET(MODE_RESID, false, true); SET(MODE_RESID, false, false); SET(MODE_GATEUP, true, false);
    SET(MODE_LMHEAD, true, false); SET(MODE_PLAIN, false, false);
#undef SET
    done = (e == hipSuccess) ? 0 : -4;
    return done;
}

#ifdef DECODE_LAB_TRACE
// lab builds only (tools/decode_stage_trace.py): the phase stamps of the most recent small-batch MFMA launch, [256 blocks][8]
extern "C" int emmax_debug_mfma_trace(unsigned long long* host_out, int n_words) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_dec_trace), (size_t)n_words * 8) == hipSuccess ? 0 : -1;
}
#endif
