// decode_attn_tail.h -- the split-KV decode attention of ONE (row, kv head, split) as the TAIL of the qkv launch
// (decode_ks.hip, batch 1-2, MHA, head_dim 128): the arithmetic of emmax_decode_attn_kernel<128, 1> (decode.hip), operation for
// operation -- same key -> lane-group assignment (4 waves x 4 groups of 16 lanes, 4 keys per group and chunk), same online
// softmax, same merge order -- so that the partials it writes are BIT-identical to the stand-alone launch's
// (tests/test_operating_point_gpu.py compares the logits of both routings bit for bit).
//
// What differs is where the time goes.  The 16 blocks of the qkv launch that own head h's q / k / v rows form a cluster
// (decode_ks.hip); after its RoPE + K/V-append epilogue (write-through stores) a block arrives at the cluster's counter, and
// the 8 x B blocks with an attention role then (1) request the page table and the first two K/V chunks of their split --
// everything that does not depend on THIS step's qkv -- (2) wait until all 16 blocks of the cluster have arrived, (3) read q
// and the newest key (agent-scope loads: written by sibling blocks, possibly on another XCD) and run the pipeline.  The
// attention launch, its ramp and its kernel boundary disappear; the K/V latency hides under the cluster hand-off.
// Replaces HF cached attention at q_len = 1 (prismatic/extern/hf/modeling_prismatic.py:325-341), like decode.hip's kernel.
#pragma once

#include "common.h"
#include "kernels.h"

namespace attn_tail {

constexpr int HD = 128, NW = 4, NT = NW * 64, KU = 4, SP = 512, PSTRIDE = EMMAX_PSTRIDE;

struct Shared {
    int pages[SP];
    float red_o[NW][HD];
    float red_ml[NW][2];
    unsigned target;
};

// `wait_cluster`: called once, by every thread of the block, after the loads that do not depend on this step's qkv are out
template <class WaitFn>
__device__ __forceinline__ void run(const GemvParams& p, int b, int hk, int split, int nsplit, Shared& sh, WaitFn&& wait_cluster) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool worker = tid < NT;                     // waves 0-3 carry the arithmetic (the stand-alone kernel's block shape)
    const int kg = lane >> 4, ch = lane & 15;
    const int page_shift = __builtin_ctz(p.page);
    const int ctx_now = p.ctx_len[b];
    const int row_done = p.attn_done ? p.attn_done[b] : 0;
    const int32_t* ptab = p.page_table + (size_t)b * p.max_pages;
    if (worker) {
        sh.pages[tid] = ptab[min(tid, p.max_pages - 1)];
        sh.pages[tid + NT] = ptab[min(tid + NT, p.max_pages - 1)];
    }
    const int L = ctx_now + 1;                        // keys including the one this launch appended
    int kps = (L + nsplit - 1) >> __builtin_ctz(nsplit);
    kps = (kps + 15) & ~15;
    const int k0 = split * kps;
    const int k1 = min(L, k0 + kps);
    const int Hq = p.Hq;
    float* part = p.attn_part_out + ((size_t)(b * Hq + hk) * nsplit + split) * PSTRIDE;
    const bf16_t* kc = (const bf16_t*)p.kcache;
    const bf16_t* vc = (const bf16_t*)p.vcache;
    __syncthreads();
    if (k0 >= L || row_done) {                        // empty split, or a row that no longer decodes: no K/V traffic, nothing to wait for
        for (int i = tid; i < PSTRIDE; i += blockDim.x) part[i] = (i == HD) ? -INFINITY : 0.f;
        return;
    }

    float m = -INFINITY, l = 0.f, o[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = 0.f;
    u32x4_t q = {0u, 0u, 0u, 0u};

    // `fresh`: the loads may touch the row this launch appended (position L - 1): only after the cluster has arrived, agent scope
    auto load_chunk = [&](int kb, u32x4_t (&kv)[KU], u32x4_t (&vv)[KU], bool (&ok)[KU], bool fresh) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int key = kb + u * (4 * NW) + wave * 4 + kg;
            ok[u] = key < k1;
            const int kk = ok[u] ? key : k0;
            const int pg = sh.pages[kk >> page_shift];
            const size_t off = ((((size_t)pg * p.Hkv + hk) << page_shift) + (kk & (p.page - 1))) * HD + ch * 8;
            if (kk == L - 1) {
                if (fresh) {
                    kv[u] = ld_act16((const u32x4_t*)(kc + off), true);
                    vv[u] = ld_act16((const u32x4_t*)(vc + off), true);
                }
            } else {
                kv[u] = *(const u32x4_t*)(kc + off);
                vv[u] = *(const u32x4_t*)(vc + off);
            }
        }
    };
    // the newest row of a chunk requested before the hand-off
    auto fix_fresh = [&](int kb, u32x4_t (&kv)[KU], u32x4_t (&vv)[KU]) {
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const int key = kb + u * (4 * NW) + wave * 4 + kg;
            if (key == L - 1 && key < k1) {
                const int pg = sh.pages[key >> page_shift];
                const size_t off = ((((size_t)pg * p.Hkv + hk) << page_shift) + (key & (p.page - 1))) * HD + ch * 8;
                kv[u] = ld_act16((const u32x4_t*)(kc + off), true);
                vv[u] = ld_act16((const u32x4_t*)(vc + off), true);
            }
        }
    };
    auto consume_chunk = [&](const u32x4_t (&kv)[KU], const u32x4_t (&vv)[KU], const bool (&ok)[KU]) {
        float sc[KU];
        float mc = -INFINITY;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            float s = 0.f;
            s = dot2_bf16(kv[u][0], q[0], s);
            s = dot2_bf16(kv[u][1], q[1], s);
            s = dot2_bf16(kv[u][2], q[2], s);
            s = dot2_bf16(kv[u][3], q[3], s);
            s = row16_sum(s);
            s = ok[u] ? s * p.attn_scale : -INFINITY;
            sc[u] = s;
            mc = fmaxf(mc, s);
        }
        const float mn = fmaxf(m, mc);
        const float msafe = (mn == -INFINITY) ? 0.f : mn;
        const float alpha = __expf(m - msafe);
        float ls = l * alpha;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] *= alpha;
#pragma unroll
        for (int u = 0; u < KU; ++u) {
            const float pw = __expf(sc[u] - msafe);
            ls += pw;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                o[2 * j] += pw * bf_lo(vv[u][j]);
                o[2 * j + 1] += pw * bf_hi(vv[u][j]);
            }
        }
        l = ls;
        m = mn;
    };

    constexpr int CH = (4 * NW) * KU;                 // keys per chunk of the block
    u32x4_t kvA[KU], vvA[KU], kvB[KU], vvB[KU];
    bool okA[KU], okB[KU];
#pragma unroll
    for (int u = 0; u < KU; ++u) {
        kvA[u] = vvA[u] = kvB[u] = vvB[u] = (u32x4_t){0u, 0u, 0u, 0u};
        okA[u] = okB[u] = false;
    }
    const bool hasB0 = k0 + CH < k1;                  // block-uniform
    if (worker) {
        load_chunk(k0, kvA, vvA, okA, false);
        if (hasB0) load_chunk(k0 + CH, kvB, vvB, okB, false);
    }
    wait_cluster();
    if (worker) {
        q = ld_act16((const u32x4_t*)((const bf16_t*)p.y + (size_t)b * p.ldy + hk * HD + ch * 8), true);
        fix_fresh(k0, kvA, vvA);
        if (hasB0) fix_fresh(k0 + CH, kvB, vvB);
        for (int kb = k0; kb < k1; kb += 2 * CH) {
            const bool hasB = kb + CH < k1;
            if (hasB && kb != k0) load_chunk(kb + CH, kvB, vvB, okB, true);
            consume_chunk(kvA, vvA, okA);
            if (kb + 2 * CH < k1) load_chunk(kb + 2 * CH, kvA, vvA, okA, true);
            if (hasB) consume_chunk(kvB, vvB, okB);
        }
        // merge the 4 key groups of the wave (lanes with equal ch), then the 4 waves through LDS
        const float mw = rows_max(m);
        const float msafe = (mw == -INFINITY) ? 0.f : mw;
        const float f = __expf(m - msafe);
        const float lv = rows_sum(l * f);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = rows_sum(o[j] * f);
            if (kg == 0) sh.red_o[wave][ch * 8 + j] = v;
        }
        if (lane == 0) {
            sh.red_ml[wave][0] = mw;
            sh.red_ml[wave][1] = lv;
        }
    }
    __syncthreads();
    if (worker) {
        for (int i = tid; i < PSTRIDE; i += NT) {
            float M = sh.red_ml[0][0];
#pragma unroll
            for (int w = 1; w < NW; ++w) M = fmaxf(M, sh.red_ml[w][0]);
            const float msafe = (M == -INFINITY) ? 0.f : M;
            float v = 0.f;
            if (i < HD) {
#pragma unroll
                for (int w = 0; w < NW; ++w) v += sh.red_o[w][i] * __expf(sh.red_ml[w][0] - msafe);
            } else if (i == HD) {
                v = M;
            } else if (i == HD + 1) {
#pragma unroll
                for (int w = 0; w < NW; ++w) v += sh.red_ml[w][1] * __expf(sh.red_ml[w][0] - msafe);
            }
            part[i] = v;
        }
    }
}

}  // namespace attn_tail
