// Shared device helpers for the QSpec gfx950 kernels.
// Wave = 64 lanes everywhere; no CUDA-compat shims.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>

namespace qspec {

// FRAGMENT-MAJOR layout of a 16-row fp16 activation tile [16, K] (K % 128 == 0), the layout the W4A16 streaming kernel's
// MFMA fragments have (gemm_stream.hip, XP): [K / 128 steps][4 dwords][4 k-groups][16 rows][8 halves in the dequantiser's
// order 0,4,1,5,2,6,3,7].  A wave then takes a fragment with ONE fully coalesced 1-KiB load instead of the workgroup copying
// the tile through LDS.  Offset (in halves) of element (row r < 16, column k):
__host__ __device__ __forceinline__ size_t w4a16_xperm_offset(int r, int k) {
    const int kstep = k >> 7, rem = k & 127, g = rem >> 5, dd = (rem >> 3) & 3, e = k & 7;
    const int pos = ((e & 3) << 1) | (e >> 2);           // 0,1,2,3,4,5,6,7 -> 0,2,4,6,1,3,5,7
    return ((((size_t)kstep * 4 + dd) * 64 + g * 16 + r) << 3) + pos;
}
// 17..32 rows: two such tiles back to back (rows 0..15, rows 16..31), each 16 x K halves (gemm_w4a16_stream2_kernel reads both)
__host__ __device__ __forceinline__ size_t w4a16_xperm_offset(int r, int k, int K) {
    return (size_t)(r >> 4) * 16 * K + w4a16_xperm_offset(r & 15, k);
}


typedef _Float16 f16;
typedef f16 f16x2 __attribute__((ext_vector_type(2)));
typedef f16 f16x4 __attribute__((ext_vector_type(4)));
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32;
typedef u32 u32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));

// fp16 <-> fp32: v_cvt_f32_f16 is exact, v_cvt_f16_f32 rounds to nearest even
// in the default float mode hipcc sets for kernels.
__device__ __forceinline__ float h2f(f16 h) { return (float)h; }
// The empty asm keeps the fp32 value opaque: without it hipcc (even at -ffp-contract=off) may fold
// `(f16)(a * b)` into v_fma_mixlo_f16, which rounds the exact product ONCE to fp16, whereas the
// reference (and the oracle) round to fp32 first and then to fp16.
__device__ __forceinline__ f16 f2h(float f) {
    asm("" : "+v"(f));
    return (f16)f;
}
__device__ __forceinline__ f16 u2h(uint16_t u) { return __builtin_bit_cast(f16, u); }
__device__ __forceinline__ uint16_t h2u(f16 h) { return __builtin_bit_cast(uint16_t, h); }

// Deterministic expf -- bit-identical to oracle/qspec_oracle.c:qexpf (only
// v_fma_f32 / v_rndne_f32 / v_ldexp_f32, all exactly specified).
__device__ __forceinline__ float qexpf(float x) {
    // branch-free: the polynomial runs on a clamped argument and the special cases are selected afterwards
    // (same results as the early-return form of the oracle for every input, NaN included)
    const float xc = fminf(fmaxf(x, -86.0f), 88.0f);
    float n = __builtin_rintf(xc * 1.44269504088896341f);
    float r = __builtin_fmaf(n, -0.693145751953125f, xc);
    r = __builtin_fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = __builtin_fmaf(p, r, 1.3981999507e-3f);
    p = __builtin_fmaf(p, r, 8.3334519073e-3f);
    p = __builtin_fmaf(p, r, 4.1665795894e-2f);
    p = __builtin_fmaf(p, r, 1.6666665459e-1f);
    p = __builtin_fmaf(p, r, 5.0000001201e-1f);
    float z = r * r;
    float y = __builtin_fmaf(p, z, r) + 1.0f;
    float e = __builtin_ldexpf(y, (int)n);
    e = x > 88.0f ? __builtin_inff() : e;
    e = x < -86.0f ? 0.0f : e;
    return x != x ? x : e;
}

// round-to-nearest-even to integer with saturation, NaN -> 0
// e^x of the attention probabilities: the hardware 2^x (v_exp_f32, ~1 ulp) on x * log2(e).  Attention is compared with
// the oracle at 1e-3, not bit for bit, and the ~20-instruction deterministic qexpf (kept where bits are compared: SiLU,
// the sampler's softmax) was most of the softmax phase: one wave per SIMD has nothing to hide it behind.
// 2^0 = 1 and 2^-inf = 0 exactly, which the running-maximum logic relies on.
__device__ __forceinline__ float aexp(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

// x / s rounded to fp16, for x and s that are fp16 VALUES (the quantisers: h(x / scale), quant.cu:147), in three
// instructions: q0 = x * r, q1 = q0 + (x - q0 * s) * r with r = 1 / s correctly rounded (once per row).  The fp16
// rounding of q1 equals the fp16 rounding of the correctly rounded fp32 quotient for EVERY pair of finite fp16 x and
// positive finite fp16 s -- checked exhaustively (2^30 pairs) by tests/test_oracle_golden.py::test_three_op_fp16_division;
// s = 0 (all-zero row) gives NaN like 0 / 0 does.  (An IEEE fp32 division is ~11 instructions.)
__device__ __forceinline__ float div3_h(float x, float r, float s) {
    const float q0 = x * r;
    const float rem = __builtin_fmaf(-q0, s, x);
    return __builtin_fmaf(rem, r, q0);
}

__device__ __forceinline__ int rni_sat(float v, int lo, int hi) {
    if (v != v) return 0;
    float r = __builtin_rintf(v);
    r = r < (float)lo ? (float)lo : r;
    r = r > (float)hi ? (float)hi : r;
    return (int)r;
}

// cross-lane xor shuffle inside a wave64 (ds_bpermute; no LDS storage used)
__device__ __forceinline__ float shfl_xor_f(float v, int mask) { return __shfl_xor(v, mask, 64); }
__device__ __forceinline__ int shfl_xor_i(int v, int mask) { return __shfl_xor(v, mask, 64); }

// lane ^ M for M in {1, 2, 4, 8} as DPP moves (no LDS round trip, ~8 cycles instead of a ~100-cycle ds_bpermute):
// quad_perm for 1 and 2, row_ror:8 for 8, and for 4 a row_shl:4 into the even 4-lane banks + row_shr:4 into the odd.
template <int M>
__device__ __forceinline__ float dpp_xor(float x) {
    const int xi = __builtin_bit_cast(int, x);
    int r;
    if (M == 1) r = __builtin_amdgcn_update_dpp(0, xi, 0xB1, 0xF, 0xF, false);        // quad_perm:[1,0,3,2]
    else if (M == 2) r = __builtin_amdgcn_update_dpp(0, xi, 0x4E, 0xF, 0xF, false);   // quad_perm:[2,3,0,1]
    else if (M == 8) r = __builtin_amdgcn_update_dpp(0, xi, 0x128, 0xF, 0xF, false);  // row_ror:8
    else {
        r = __builtin_amdgcn_update_dpp(0, xi, 0x104, 0xF, 0x5, false);               // row_shl:4 -> banks 0,2
        r = __builtin_amdgcn_update_dpp(r, xi, 0x114, 0xF, 0xA, false);               // row_shr:4 -> banks 1,3
    }
    return __builtin_bit_cast(float, r);
}

// ---- one-instruction butterfly steps.  hipcc does not fold a v_mov_b32_dpp into a float add / max (its DPP combiner only
// knows integer identities), so x + dpp_xor<M>(x) costs two VALU issues; written as v_add_f32_dpp it is one (two for M = 4:
// the two bank-masked halves).  IEEE addition commutes, so the bits are those of x + dpp_xor<M>(x).  The leading s_nop
// gives the 2 wait states a DPP read needs behind the VALU write of its source: the hazard recogniser does not look into
// inline asm.
template <int M>
__device__ __forceinline__ float dpp_add_xor(float x) {
    float r;
    if (M == 1) asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else if (M == 2) asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else if (M == 8) asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else
        asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
            : "=&v"(r)
            : "v"(x));
    return r;
}
template <int M>
__device__ __forceinline__ float dpp_max_xor(float x) {
    float r;
    if (M == 1) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else if (M == 2) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else if (M == 8) asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(x));
    else
        asm("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
            "v_max_f32_dpp %0, %1, %1 row_shr:4 row_mask:0xf bank_mask:0xa"
            : "=&v"(r)
            : "v"(x));
    return r;
}
// p[c] += p[c] of lanes ^4, then ^2, then ^1, for four independent values at once: the four chains are interleaved, so
// every DPP read sits four instructions behind the write of its source and no wait states are needed in between.
__device__ __forceinline__ void dpp_add_tree421_x4(float (&p)[4]) {
    float t0, t1, t2, t3;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %2, %6, %6 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %3, %7, %7 row_shl:4 row_mask:0xf bank_mask:0x5\n\t"
        "v_add_f32_dpp %0, %4, %4 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %1, %5, %5 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %2, %6, %6 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %3, %7, %7 row_shr:4 row_mask:0xf bank_mask:0xa\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)
        : "v"(p[0]), "v"(p[1]), "v"(p[2]), "v"(p[3]));
    p[0] = t0; p[1] = t1; p[2] = t2; p[3] = t3;
}
// a, b += a, b of lane ^ 1 (four independent values, interleaved: no wait states in between)
__device__ __forceinline__ void dpp_add_xor1_x4(f32x2& a, f32x2& b) {
    float t0, t1, t2, t3;
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %1, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %2, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_f32_dpp %3, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"
        : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3)   // early clobber: t0 is written before the later inputs are read
        : "v"(a[0]), "v"(a[1]), "v"(b[0]), "v"(b[1]));
    a = f32x2{t0, t1};
    b = f32x2{t2, t3};
}
// x + x[lane ^ 32] and x + x[lane ^ 16] on the VALU (gfx950 v_permlane32_swap / v_permlane16_swap: the odd half / the odd
// rows of the first register change places with the even half / rows of the second; with both = x the two registers then
// hold the two operands of the butterfly level in every lane).  Inline asm: clang 19's builtin mis-pairs the two results.
__device__ __forceinline__ float add_xor32(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float add_xor16(float x) {
    float a = x, b = x;
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
// lane ^ 16 through the LDS crossbar without an address register (ds_swizzle, bit mode: and 0x1f, xor 0x10)
__device__ __forceinline__ float swizzle_xor16_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(__builtin_bit_cast(int, v), 0x401F));
}
// maximum over the wave as a wave-uniform value (SGPR): xor 1, 2, 4, 8 inside the rows of 16, then the GFX9 row
// broadcasts (lane 15 of the row before into rows 1 and 3, lane 31 into rows 2 and 3) leave the total in lane 63.
__device__ __forceinline__ float wave_max_uniform(float v) {
    v = dpp_max_xor<1>(v);
    v = dpp_max_xor<2>(v);
    v = dpp_max_xor<4>(v);
    v = dpp_max_xor<8>(v);
    asm("s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
        "s_nop 1\n\tv_max_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf"
        : "+v"(v));
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 63));
}
// max over the 64 lanes of a wave in every lane (order independent; all 64 lanes must be active)
__device__ __forceinline__ float wave_max_f(float v) { return wave_max_uniform(v); }
__device__ __forceinline__ float readlane_f(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}

// pack two int4 values (two's complement) into one byte: lo nibble = even index
__device__ __forceinline__ uint32_t pack_nib(int q0, int q1) { return (uint32_t)(q0 & 0xF) | ((uint32_t)(q1 & 0xF) << 4); }


// ---- W4A16 dequantiser: 8 signed int4 of one packed dword -> 8 fp16, k order 0,4,1,5,2,6,3,7 (pairs (0,4) (1,5) (2,6) (3,7)).
// Offset-binary nibble u = n ^ 8 spliced under the exponent of 1024.0 (0x6400 | u = 1024 + u) or of 64.0 after the x 1/16
// (0x6400 | u << 4); the sign flip, the nibble mask and the exponent splice are ONE v_bitop3_b32 each: per bit the result is
// p, ~p, 0 or 1, chosen by two constant words (A: the bit comes from p; B: invert it / the constant bit).  9 VALU per
// dword instead of the 14 the xor / and / or form compiled to -- the verify pass's streaming GEMMs issue VALU 35-49 % of
// their wave cycles (scripts/pmc_cycle_sq.sh), most of it this function.
__device__ __forceinline__ f16x8 dequant_s4x8_bitop(u32 p) {
    constexpr u32 A0 = 0x000F000Fu, B0 = 0x64086408u, A1 = 0x00F000F0u, B1 = 0x64806480u;
    constexpr int TT = 0x6A;   // (A & ~B & p) | (A & B & ~p) | (~A & B) with src0 = p, src1 = A, src2 = B
    const u32 q = p >> 8;
    const u32 r0 = __builtin_amdgcn_bitop3_b32(p, A0, B0, TT), r1 = __builtin_amdgcn_bitop3_b32(p, A1, B1, TT);
    const u32 r2 = __builtin_amdgcn_bitop3_b32(q, A0, B0, TT), r3 = __builtin_amdgcn_bitop3_b32(q, A1, B1, TT);
    const f16x2 c1032 = {(f16)1032.0f, (f16)1032.0f}, c16 = {(f16)0.0625f, (f16)0.0625f}, c72 = {(f16)72.0f, (f16)72.0f};
    const f16x2 h0 = __builtin_bit_cast(f16x2, r0) - c1032;
    const f16x2 h1 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, r1), c16, -c72);
    const f16x2 h2 = __builtin_bit_cast(f16x2, r2) - c1032;
    const f16x2 h3 = __builtin_elementwise_fma(__builtin_bit_cast(f16x2, r3), c16, -c72);
    return f16x8{h0[0], h0[1], h1[0], h1[1], h2[0], h2[1], h3[0], h3[1]};
}

}  // namespace qspec
