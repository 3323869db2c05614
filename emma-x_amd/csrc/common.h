// common.h -- shared device helpers for the gfx950 kernels (wave64, bf16 bit patterns, MFMA fragment types).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bf16 bit pattern in memory

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;   // MFMA 16x16x32 A/B fragment (4 VGPRs)
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;     // MFMA 16x16 C/D fragment
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;  // 16-byte load unit (8 bf16)
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;

#define EMMAX_WAVE 64

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN-preserving (same rounding as torch's float->bfloat16): gfx950 converts in hardware
// (v_cvt_pk_bf16_f32, two values per instruction)
typedef __attribute__((ext_vector_type(2))) float f32x2_t;
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
    return __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){lo, hi}, bf16x2_t));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf16x2(f, 0.f) & 0xffffu); }
__device__ __forceinline__ float bf_lo(uint32_t p) { return __uint_as_float(p << 16); }
__device__ __forceinline__ float bf_hi(uint32_t p) { return __uint_as_float(p & 0xffff0000u); }

// acc += a.lo*b.lo + a.hi*b.hi on packed bf16 pairs (v_dot2c_f32_bf16)
__device__ __forceinline__ float dot2_bf16(uint32_t a, uint32_t b, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2_t, a), __builtin_bit_cast(bf16x2_t, b), acc, false);
}

// Wave-wide reductions on the DPP path.  `__shfl_xor` compiles to ds_bpermute_b32 + index arithmetic + s_waitcnt
// lgkmcnt(0): six serial trips through the LDS crossbar per reduction (~0.35 us; the eight RMSNorm row statistics of a
// batch-8 decode prologue took 2.7 us).  Here: a butterfly inside each row of 16 lanes (quad_perm xor 1, xor 2,
// row_half_mirror, row_mirror -- plain VALU instructions with a DPP operand), then the four row totals are read into
// SGPRs and combined; every lane returns the same value.
template <int CTRL>
__device__ __forceinline__ float dpp_move(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_value(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
// sum over the 16 lanes of each DPP row (lane >> 4), in every lane of the row
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror
    v += dpp_move<0x140>(v);   // row_mirror
    return v;
}
// max over the 16 lanes of each DPP row, in every lane of the row
__device__ __forceinline__ float row16_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return v;
}
// ---- e4m3 (OCP) rows with one fp32 scale per row: the opt-in fp8 KV cache (round 5) ----
// The row scale is a POWER OF TWO, the smallest one with amax / scale <= 448: v_cvt_scalef32_pk_bf16_fp8 applies only the exponent of its
// scale operand (measured: an arbitrary fp32 scale came back truncated to 2^floor(log2 s) -- up to 2 x off), and e4m3 being a floating
// format a power-of-two scale costs no relative precision, only (at most one binade of) range.  Both directions are then exact.
__device__ __forceinline__ float e4m3_row_scale(float amax) {
    const float r = amax * (1.0f / 448.0f);
    if (!(r > 0.f)) return 1.0f;
    uint32_t u = __float_as_uint(r), e = u & 0x7f800000u;
    if (u & 0x007fffffu) e += 0x00800000u;
    return __uint_as_float(e ? e : 0x00800000u);
}
// 8 bf16 (u32x4) -> 8 e4m3 (u32x2), values divided by the row's scale first; 8 e4m3 -> 8 bf16 multiplied by the scale
__device__ __forceinline__ u32x2_t quant8_e4m3(const u32x4_t& v, float inv_scale) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[0]) * inv_scale, bf_hi(v[0]) * inv_scale, lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[1]) * inv_scale, bf_hi(v[1]) * inv_scale, lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[2]) * inv_scale, bf_hi(v[2]) * inv_scale, hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(bf_lo(v[3]) * inv_scale, bf_hi(v[3]) * inv_scale, hi, true);
    return (u32x2_t){(uint32_t)lo, (uint32_t)hi};
}
__device__ __forceinline__ u32x4_t dequant8_e4m3(const u32x2_t& q, float scale) {
    return (u32x4_t){__builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[0], scale, false)),
                     __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[0], scale, true)),
                     __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[1], scale, false)),
                     __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_scalef32_pk_bf16_fp8(q[1], scale, true))};
}
// exchange across the four rows (lanes with equal lane & 15) with the gfx950 swap instructions:
//   v_permlane32_swap a, b: a.hi <-> b.lo   => with a = b = v:  a = {v.lo, v.lo}, b = {v.hi, v.hi}
//   v_permlane16_swap a, b: odd rows of a <-> even rows of b
// written as asm: the builtin form folded the two results into one register here (hipcc 7.2).
__device__ __forceinline__ void swap_halves(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ void swap_rows(float v, float& a, float& b) {
    a = v; b = v;
    asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
__device__ __forceinline__ float rows_sum(float v) {
    float a, b;
    swap_halves(v, a, b);
    swap_rows(a + b, a, b);
    return a + b;
}
__device__ __forceinline__ float rows_max(float v) {
    float a, b;
    swap_halves(v, a, b);
    swap_rows(fmaxf(a, b), a, b);
    return fmaxf(a, b);
}
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_move<0xB1>(v);    // quad_perm [1,0,3,2]
    v += dpp_move<0x4E>(v);    // quad_perm [2,3,0,1]
    v += dpp_move<0x141>(v);   // row_half_mirror
    v += dpp_move<0x140>(v);   // row_mirror: every lane holds its row's sum
    return (lane_value(v, 0) + lane_value(v, 16)) + (lane_value(v, 32) + lane_value(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_move<0xB1>(v));
    v = fmaxf(v, dpp_move<0x4E>(v));
    v = fmaxf(v, dpp_move<0x141>(v));
    v = fmaxf(v, dpp_move<0x140>(v));
    return fmaxf(fmaxf(lane_value(v, 0), lane_value(v, 16)), fmaxf(lane_value(v, 32), lane_value(v, 48)));
}

// 8 consecutive elements of an fp32 row (the decode step's fp32 residual stream)
struct f32x8_t { f32x4_t lo, hi; };
__device__ __forceinline__ f32x8_t ld_f32x8(const float* p) { return {*(const f32x4_t*)p, *(const f32x4_t*)(p + 4)}; }
__device__ __forceinline__ u32x4_t f32x8_to_bf16(const f32x8_t& v) {
    return (u32x4_t){pack_bf16x2(v.lo[0], v.lo[1]), pack_bf16x2(v.lo[2], v.lo[3]), pack_bf16x2(v.hi[0], v.hi[1]), pack_bf16x2(v.hi[2], v.hi[3])};
}
__device__ __forceinline__ f32x8_t bf16x8_to_f32(const u32x4_t& v) {
    return {(f32x4_t){bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])}, (f32x4_t){bf_lo(v[2]), bf_hi(v[2]), bf_lo(v[3]), bf_hi(v[3])}};
}
__device__ __forceinline__ float f32x8_at(const f32x8_t& v, int i) { return i < 4 ? v.lo[i] : v.hi[i - 4]; }

// exact-erf GELU, x Phi(x) = max(x, 0) - |x| h(|x|) with h = erfc(|x| / sqrt 2) / 2 (no cancellation on the negative side), and
// h = exp2(Q(|x|)): log2 of erfc is smooth (-1 at 0, ~ -x^2 log2(e) / 2 far out), a degree-6 polynomial weighted for the error of
// |x| h reproduces x Phi(x) to 2.8e-7 absolute in fp32 arithmetic (1.7 % of a bf16 ulp of the result at worst; tools/fit_gelu.py).
// Written for the instruction count -- VALU work of a GEMM epilogue does not overlap the MFMAs of its SIMD, and the quarter-rate
// transcendentals weigh four full-rate instructions each: min, 6 fma (packed two per instruction by hipcc), v_exp_f32, max, fma =
// 9 full-rate + ONE transcendental (rounds 2-4: Abramowitz-Stegun 7.1.26 with rcp + exp2, 12 + two; the ocml erff is ~40).  Beyond
// |x| = 6 (h < 1e-9) the argument of Q is clamped: its leading coefficient is positive.
__device__ __forceinline__ float gelu_erf(float x) {
    const float a = fminf(fabsf(x), 6.0f);
    float q = fmaf(3.3092673036e-05f, a, -7.6921858316e-04f);
    q = fmaf(q, a, 8.0807131948e-03f);
    q = fmaf(q, a, -5.3412099418e-02f);
    q = fmaf(q, a, -4.5877097672e-01f);
    q = fmaf(q, a, -1.1512017009e+00f);
    q = fmaf(q, a, -9.9999306113e-01f);
    return fmaf(-fabsf(x), __builtin_amdgcn_exp2f(q), fmaxf(x, 0.f));
}
// Two values at once on the packed fp32 pipe (v_pk_fma_f32: two fmas per issue slot).  From the scalar form hipcc builds the Horner
// chain out of v_fmaak_f32 with literal coefficients -- one slot per fma; as 2-vectors the six fmas of a PAIR take six slots.  Same
// operations in the same order: bit-identical to gelu_erf per element.
__device__ __forceinline__ f32x2_t gelu_erf2(f32x2_t x) {
    const f32x2_t ax = {fabsf(x[0]), fabsf(x[1])};
    const f32x2_t a = {fminf(ax[0], 6.0f), fminf(ax[1], 6.0f)};
    f32x2_t q = __builtin_elementwise_fma((f32x2_t){3.3092673036e-05f, 3.3092673036e-05f}, a, (f32x2_t){-7.6921858316e-04f, -7.6921858316e-04f});
    q = __builtin_elementwise_fma(q, a, (f32x2_t){8.0807131948e-03f, 8.0807131948e-03f});
    q = __builtin_elementwise_fma(q, a, (f32x2_t){-5.3412099418e-02f, -5.3412099418e-02f});
    q = __builtin_elementwise_fma(q, a, (f32x2_t){-4.5877097672e-01f, -4.5877097672e-01f});
    q = __builtin_elementwise_fma(q, a, (f32x2_t){-1.1512017009e+00f, -1.1512017009e+00f});
    q = __builtin_elementwise_fma(q, a, (f32x2_t){-9.9999306113e-01f, -9.9999306113e-01f});
    const f32x2_t e = {__builtin_amdgcn_exp2f(q[0]), __builtin_amdgcn_exp2f(q[1])};
    const f32x2_t m = {fmaxf(x[0], 0.f), fmaxf(x[1], 0.f)};
    return __builtin_elementwise_fma(-ax, e, m);
}
// x sigmoid(x) as x * rcp(1 + exp2(-x log2 e)): mul, v_exp_f32, add, v_rcp_f32, mul.  Written as a quotient it compiled to the IEEE division
// sequence (two v_div_scale, rcp, five fma, v_div_fmas, v_div_fixup): 13 instructions per element of a SwiGLU epilogue for a last-bit
// difference that the bf16 rounding of the product erases (v_rcp_f32: 1 ulp).
__device__ __forceinline__ float silu(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
// the exact-numerics path (round 6): the IEEE quotient (the product is NOT rounded to bf16 afterwards there)
__device__ __forceinline__ float silu_precise(float x) { return x / (1.0f + __expf(-x)); }

// ---- exact numerics (round 6): an activation as two bf16 terms, x = hi + lo with hi = bf16(x), lo = bf16(x - hi): 16 mantissa bits ----
struct hl2_t { uint32_t hi, lo; };
__device__ __forceinline__ hl2_t split_hl2(float a, float b) {
    const uint32_t hi = pack_bf16x2(a, b);
    return {hi, pack_bf16x2(a - bf_lo(hi), b - bf_hi(hi))};
}
// ---- exact numerics, the 24-bit K / V cache ("x24"): the top 24 bits of the fp32 value (sign, exponent, 15 mantissa bits, rounded to nearest) as a bf16
// plane (the top 16 bits) + an 8-bit extension plane: 2^-16 relative per element at 1.5 x the bytes of the bf16 cache (fp32: 2 x) ----
__device__ __forceinline__ uint32_t x24_bits(float x) { return __float_as_uint(x) + 0x80u; }            // (hi = bits >> 16, ext = (bits >> 8) & 0xff)
// 8 elements: hi = 8 x 16 bits (u32x4), ext = 8 x 8 bits (u32x2) -> 8 floats; v_perm_b32 assembles {hi.b1, hi.b0, ext.b, 0x00} per element
__device__ __forceinline__ void x24_unpack8(const u32x4_t& hi, const u32x2_t& ext, float (&f)[8]) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint32_t sel = ((e & 1) ? 0x07060000u : 0x05040000u) | ((uint32_t)(e & 3) << 8) | 0x0cu;
        f[e] = __uint_as_float(__builtin_amdgcn_perm(hi[e >> 1], ext[e >> 2], sel));
    }
}
// position of element c of a row inside its HL image (per 64 elements: 64 hi, then 64 lo)
__host__ __device__ __forceinline__ int hl_col(int c) { return ((c >> 6) << 7) + (c & 63); }

// ---------------------------------------------------------------------------------------------------------------------
// Activation accesses with a block-uniform `coh` switch: plain under ordinary stream ordering; agent-scope (sc1: past the
// XCD-private L2) where a kernel reads what ANOTHER block of the same launch wrote (the in-attention split merge, decode.hip).
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ u32x4_t ld_act16(const u32x4_t* p, bool coh) {
    if (!coh) return *p;
    const unsigned long long* q = (const unsigned long long*)p;
    const unsigned long long a = __hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long b = __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return (u32x4_t){(uint32_t)a, (uint32_t)(a >> 32), (uint32_t)b, (uint32_t)(b >> 32)};
}
__device__ __forceinline__ void st_act16(u32x4_t* p, u32x4_t v, bool coh) {
    if (!coh) { *p = v; return; }
    unsigned long long* q = (unsigned long long*)p;
    __hip_atomic_store(q, (unsigned long long)v[0] | ((unsigned long long)v[1] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __hip_atomic_store(q + 1, (unsigned long long)v[2] | ((unsigned long long)v[3] << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ bf16_t ld_act_bf16(const bf16_t* p, bool coh) {
    return coh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ void st_act_bf16(bf16_t* p, bf16_t v, bool coh) {
    if (coh) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ float ld_act_f32(const float* p, bool coh) {
    return coh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ void st_act_f32(float* p, float v, bool coh) {
    if (coh) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ int ld_act_i32(const int* p, bool coh) {
    return coh ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *p;
}
__device__ __forceinline__ void st_act_i32(int* p, int v, bool coh) {
    if (coh) __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else *p = v;
}
__device__ __forceinline__ f32x4_t ld_act_f32x4(const float* p, bool coh) {
    if (!coh) return *(const f32x4_t*)p;
    const u32x4_t v = ld_act16((const u32x4_t*)p, true);
    return __builtin_bit_cast(f32x4_t, v);
}

// 16-byte write-through (agent-scope, sc1) store: acknowledged once it is past the XCD-private L2.  Compiler-visible (the data
// registers are read before the statement ends: the trailing s_nop, cdna_hip_programming.md section 5.7 item 1).
__device__ __forceinline__ void st_sc1_f32x4(float* p, f32x4_t v) {
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(p), "v"(v) : "memory");
}

// ---------------------------------------------------------------------------------------------------------------------
// Split-KV attention partials: per (row, head, split) EMMAX_PSTRIDE floats = { o[0..128) un-normalised, m, l, 2 pad }.
// attn_merge_chunk merges the NS partials of one head for the 8 output elements d0..d0+8 and returns them as 8 bf16.
// Branch-free with every load issued before the first use (one L2 round trip for the scalars, one for the vectors):
// the merge sits on the critical path of every o-proj launch.
// ---------------------------------------------------------------------------------------------------------------------
#define EMMAX_PSTRIDE 132
template <int NS, int GMAX = 4>   // GMAX: vector loads (pairs) in flight per group -- the register budget of the caller
__device__ __forceinline__ u32x4_t attn_merge_chunk(const float* __restrict__ pp, int d0, float* out32 = nullptr) {
    float ms[NS], dn[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        ms[s] = pp[s * EMMAX_PSTRIDE + 128];
        dn[s] = pp[s * EMMAX_PSTRIDE + 129];
    }
    constexpr int GS = NS < GMAX ? NS : GMAX;
    float M = -INFINITY;
#pragma unroll
    for (int s = 0; s < NS; ++s) M = fmaxf(M, ms[s]);
    float den = 0.f, a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s0 = 0; s0 < NS; s0 += GS) {
        f32x4_t o0[GS], o1[GS];
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            o0[s] = *(const f32x4_t*)(pp + (s0 + s) * EMMAX_PSTRIDE + d0);
            o1[s] = *(const f32x4_t*)(pp + (s0 + s) * EMMAX_PSTRIDE + d0 + 4);
        }
#pragma unroll
        for (int s = 0; s < GS; ++s) {
            // explicit fma: the two merge variants (this one and the loop below) must round identically -- the chained launch
            // uses the loop, the plain launch this one, and their logits are compared bit for bit
            const float wgt = (ms[s0 + s] == -INFINITY) ? 0.f : __expf(ms[s0 + s] - M);
            den = __builtin_fmaf(dn[s0 + s], wgt, den);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                a8[j] = __builtin_fmaf(o0[s][j], wgt, a8[j]);
                a8[4 + j] = __builtin_fmaf(o1[s][j], wgt, a8[4 + j]);
            }
        }
    }
    const float inv = den > 0.f ? 1.0f / den : 0.f;
    if (out32) {   // exact numerics: the merged head chunk in fp32
#pragma unroll
        for (int j = 0; j < 8; ++j) out32[j] = a8[j] * inv;
    }
    u32x4_t v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(a8[2 * j] * inv, a8[2 * j + 1] * inv);
    return v;
}
// generic split count, small register footprint (the dot2 GEMV keeps its 16-load weight ring live across the prologue
// and must stay under 128 VGPRs)
__device__ __forceinline__ u32x4_t attn_merge_chunk_loop(const float* __restrict__ pp, int d0, int nsplit, bool coh = false, float* out32 = nullptr) {
    float M = -INFINITY;
    for (int s = 0; s < nsplit; ++s) M = fmaxf(M, ld_act_f32(pp + s * EMMAX_PSTRIDE + 128, coh));
    float den = 0.f, a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < nsplit; ++s) {
        const float m = ld_act_f32(pp + s * EMMAX_PSTRIDE + 128, coh);
        const float wgt = (m == -INFINITY) ? 0.f : __expf(m - M);
        den = __builtin_fmaf(ld_act_f32(pp + s * EMMAX_PSTRIDE + 129, coh), wgt, den);
        const f32x4_t o0 = ld_act_f32x4(pp + s * EMMAX_PSTRIDE + d0, coh);
        const f32x4_t o1 = ld_act_f32x4(pp + s * EMMAX_PSTRIDE + d0 + 4, coh);
#pragma unroll
        for (int j = 0; j < 4; ++j) {   // explicit fma: see attn_merge_chunk
            a8[j] = __builtin_fmaf(o0[j], wgt, a8[j]);
            a8[4 + j] = __builtin_fmaf(o1[j], wgt, a8[4 + j]);
        }
    }
    const float inv = den > 0.f ? 1.0f / den : 0.f;
    if (out32) {
#pragma unroll
        for (int j = 0; j < 8; ++j) out32[j] = a8[j] * inv;
    }
    u32x4_t v;
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = pack_bf16x2(a8[2 * j] * inv, a8[2 * j + 1] * inv);
    return v;
}

// row of the SOURCE matrix (row-major, decode.hip's orders) that row r of tile t of the permuted copy holds
//   perm 0: natural;  1: qkv -- head hb = t / (hd/16), block j = t % (hd/16): rows 8j..8j+7 and hd/2 + 8j .. of the head;
//   2: gate/up -- gate_g, up_g of g = 8t + r (source: 16-row interleaved groups, gate_g at (g>>4)*32 + (g&15), up_g 16 below)
__host__ __device__ __forceinline__ int km_src_row(int perm, int head_dim, int t, int r) {
    if (perm == 1) {
        const int tph = head_dim / 16, hb = t / tph, j = t - hb * tph;
        return hb * head_dim + (r < 8 ? 8 * j + r : head_dim / 2 + 8 * j + r - 8);
    }
    if (perm == 2) {
        const int g = 8 * t + (r & 7);
        return (g >> 4) * 32 + (g & 15) + (r < 8 ? 0 : 16);
    }
    return t * 16 + r;
}

static inline int cdiv(int a, int b) { return (a + b - 1) / b; }
static inline int64_t align_up(int64_t a, int64_t b) { return (a + b - 1) / b * b; }
