// Weight-streaming W4A4 GEMM for decode-sized M (<= 16 tokens), second generation: the kernel of the draft pass.
//
// Same arithmetic and the same shared packed-int4 weight buffer as gemm.hip:gemm_w4a4_kernel (reference:
// third-party/ao/torchao/csrc/cuda/rowwise_scaled_linear_cutlass/rowwise_scaled_linear_cutlass_unified.cuh:240-488,
// Python quarot_nn/linear.py:67-84), restructured around what bounds it at M = 4: bytes in flight and launches.
//
//   * A workgroup of NW waves (4 / 8 / 16 by K) owns 16-row weight tiles and LOOPS over them (stride gridDim.x).
//     The waves interleave over K in 64-byte steps, 8 steps (8 KiB per wave) per batch; the loads of the NEXT
//     batch -- usually the next tile -- are issued before the current one is consumed, so every wave keeps 8 KiB
//     in flight across the reduce / epilogue of a tile.  With K = 4096 and NW = 4, or K = 14336 and NW = 16, one
//     batch is one tile: all of a tile's bytes are requested at once.
//   * Activations live in LDS (int4-packed, [M][K/2 + 32]: the 32-byte pad spreads the token rows over the banks),
//     fragments are ds_read_b128 -- LDS reads count in lgkmcnt and never queue behind the weight loads in vmcnt.
//   * PRO_LN: the residual add + LayerNorm-no-gamma + per-token int4 quantisation that feeds qkv_proj / gate_up
//     (layernorm_kernels.cu:569-716, quarot_llama.py:373-388) runs as the PROLOGUE of the GEMM, redundantly in every
//     workgroup, underneath the latency of the first weight batch: the LN kernel and its launch boundary disappear.
//     It is the same arithmetic as norm_quant.hip:ln_kernel (reference reduction tree, bit for bit), batched over
//     rows.  Workgroup 0 also writes the updated residual stream (hidden_out != hidden_in: ping-pong, no race).
//     Forms: PRO_LN / PRO_LN1 (every wave of the workgroup takes part in the norm; LN1 = no delta, no write-back),
//     PRO_LNS / PRO_LN1S (M <= 4, what the draft pass launches: four extra norm waves, one row each with no barrier
//     inside, while the streaming waves already request their first two tiles), PRO_LNH (a few producer workgroups
//     hand the rows over through L2; M >= 8, not the default any more).
//   * PRO_RQ (M <= 4, K = 4096): fp16 rows + eight partial row maxima each (the spread head-Hadamard's output) -> the
//     row-absmax int4 quantiser (quant.cu:102-167: scale = h(h(amax / 7) h(clip)), q = clamp(rne(h(x / scale)), -8, 7)) in
//     the prologue of o_proj, under the latency of the workgroup's one weight tile.
//   * Epilogues as in gemm.hip (plain / RoPE + KV-cache write / silu(gate)*up), with their operands (channel
//     scale, cos/sin, position, slot) prefetched together with the tile's weights.
//   * Cross-wave K reduction through LDS in wave order: int32, exact, deterministic.
//
// Also in this file (round 3), all bit-identical to the register forms:
//   * gemm_f16_sdma_kernel: the lm_head's 1 GB stream through self-service LDS-DMA -- what the cycle launches;
//   * three LDS-DMA forms of the draft GEMMs, measured and left OFF ("Stage A"): the loader / consumer
//     engine (gemm_w4a4_engine_kernel, QSPEC_ENGINE=1), the self-service form (gemm_w4a4_sdma_kernel, QSPEC_SDMA=1..3) and
//     loader waves beside the register stream (template parameter DMA of the kernel below, QSPEC_DMA_TILES=1..3).
#include <stdlib.h>

#include "common.cuh"
#include "kernels.h"

namespace qspec {

// Weight loads.  Non-temporal loads (`global_load_dwordx4 ... nt`, -DQS_NT_WEIGHTS) were measured on the register
// refill ring of these kernels and lost: cycle 8.66 -> 8.90 ms, gate_up 15.6 -> 16.1 us, down 8.4 -> 8.8 us (round 2;
// the guide's gain is for LDS-DMA loader rings).  Default cache policy it is.
template <typename T>
__device__ __forceinline__ T wload(const void* p) {
#ifdef QS_NT_WEIGHTS
    return __builtin_nontemporal_load(reinterpret_cast<const T*>(p));
#else
    return *reinterpret_cast<const T*>(p);
#endif
}

enum { SEPI_PLAIN = 0, SEPI_QKV = 1, SEPI_GATEUP = 2, SEPI_RESID = 3, SEPI_PARTIAL = 4, SEPI_IPART = 5 };   // RESID: plain + fp16 residual add
// SEPI_IPART (W4A4, (xq, xs) input): blockIdx.y picks one of gridDim.y K slices of length K; the raw int32 sums of the slice
// go to ipart[blockIdx.y][M][N] and whoever consumes them (norm_quant.hip: ln_kernel with ipart) adds the slices -- exact in
// any order -- and applies the epilogue expression.  What it is for: at 17..32 tokens a workgroup with ONE 16-row tile reads
// 2 x as many activation bytes as weight bytes (config 3's down_proj: 229 KB for 115 KB); with two K slices it takes TWO
// tiles of half the K and reads its activation fragments once for both.
enum { PRO_Q = 0, PRO_LN = 1, PRO_LNH = 2, PRO_LN1 = 3, PRO_LNS = 4, PRO_LN1S = 5, PRO_RQ = 6 };   // (see the header)

struct StreamArgs {
    const int8_t* xq;       // PRO_Q : [M, K/2] packed int4 activations
    const f16* xs;          // PRO_Q : [M] activation scales
    const f16* hidden_in;   // PRO_LN: [M, K] residual stream
    const f16* delta;       // PRO_LN: [M, K] output of the previous projection (added first) or nullptr
    f16* hidden_out;        // PRO_LN: [M, K] updated residual stream (written by workgroup 0) or nullptr
    float eps;
    int* sync;              // PRO_LNH: hand-off workspace (gemm_w4a4_stream_sync_bytes(), zero-filled once)
    const f16* x16;         // PRO_RQ: [M, K] fp16 rows
    const float* part_amax; // PRO_RQ: [M, 8] partial maxima of |x16| per row
    float clip;             // PRO_RQ: the quantiser's clip ratio
    const f16* resid_in;    // SEPI_RESID: [M, N] residual stream; resid_out = h(f(resid_in) + f(h(gemm)))
    f16* resid_out;         // SEPI_RESID: [M, N] (may alias resid_in: each element is read and written by one thread)
    const f16* x;           // W4A16: [M, K] fp16 activations, row stride ldx halves
    int64_t ldx, ldw;       // W4A16: activation row stride (halves) / packed weight row stride (bytes); 0 = dense
    int tile0;              // W4A16: first tile of the launch (column-parallel shards)
    float* part;            // W4A16 SEPI_PARTIAL: [gridDim.y][M][N] raw fp32 sums of K slice blockIdx.y (K = slice length)
    int xperm;              // W4A16: x is the 16-row activation tile in fragment-major layout (w4a16_xperm_offset)
    int* ipart;             // W4A4 SEPI_IPART: [gridDim.y][M][N] raw int32 sums of K slice blockIdx.y (ldx / ldw = row strides in BYTES)
    const uint8_t* wq;      // [N, K/2]
    const f16* ws;          // [N]
    f16* out;
    int M, N, K, ntiles;
    const int64_t* positions;
    const f16* cos_sin_cache;
    f16* key_cache;
    f16* value_cache;
    const int64_t* slot_mapping;
    int nq, nkv, I;
    int hdl = 7;   // SEPI_QKV: log2(head size), 7 or 6
};

__device__ __forceinline__ i32x4 widen16(u32 p0, u32 p1) {
    return i32x4{(int)((p0 << 4) & 0xF0F0F0F0u), (int)(p0 & 0xF0F0F0F0u), (int)((p1 << 4) & 0xF0F0F0F0u),
                 (int)(p1 & 0xF0F0F0F0u)};
}

// Byte offset (inside one batch of NW*UB*64 bytes of a weight row) of step u of `wave`.  Steps are paired so that a
// wave's two consecutive loads cover one 128-byte line of every row (a 16x16 MFMA tile puts only 4 lanes = 64 bytes
// on a row per load instruction); an odd UB leaves one unpaired step at the end of the batch.
template <int NW, int UB>
__device__ __forceinline__ int step_off(int wave, int u) {
    constexpr int UE = UB & ~1;
    if (u < UE) return (u >> 1) * (2 * NW * 64) + wave * 128 + (u & 1) * 64;
    return UE * NW * 64 + wave * 64;
}

// SEPI_QKV tiles.  A tile is eight RoPE pairs (i, i + 64) of one head: rows 0..7 = the pairs' first halves, 8..15 = their
// second halves.  Classic: tile t = pairs 8 (t & 7) .. + 7 of head t >> 3.  LEVELLED (I = L > 0, used where N / 16 tiles
// are 1.5 per workgroup, e.g. Llama-3-8B's 384 on 256 CUs: half the workgroups streamed two tiles, half one): the
// (heads * 64) pairs are dealt twelve to a workgroup (measured slower, off by default: see qkv_level_workgroups) -- "tile" v < L = pairs 12 v .. 12 v + 7 (a full tile), "tile" L + v =
// pairs 12 v + 8 .. 12 v + 11 (a HALF tile: slots j >= 4 repeat slots j - 4, their results are never stored; the wave
// loads coalesce the repeated rows) -- so every workgroup streams 48 KB instead of 64 or 32.
struct QkvPair {
    int head, i;
    bool valid;
};
// hdl = log2(head size): 7 (128: 64 pairs = 8 tiles per head) or 6 (64, e.g. TinyLlama: 32 pairs = 4 tiles per head)
__device__ __forceinline__ QkvPair qkv_pair(int tb, int j, int L, int hdl) {   // j = 0..7: pair slot of the tile
    if (L == 0) return QkvPair{tb >> (hdl - 4), (tb & ((1 << (hdl - 4)) - 1)) * 8 + j, true};
    const bool half = tb >= L;
    const int p = 12 * (half ? tb - L : tb) + (half ? 8 + (j & 3) : j);
    return QkvPair{p >> 6, p & 63, !half || j < 4};
}
template <int EPI>
__device__ __forceinline__ int stile_row(int tb, int r, int I, int hdl) {
    if (EPI == SEPI_QKV) {
        const QkvPair q = qkv_pair(tb, r & 7, I, hdl);
        return (q.head << hdl) + ((r >> 3) << (hdl - 1)) + q.i;
    }
    if (EPI == SEPI_GATEUP) return (r >> 3) * I + tb * 8 + (r & 7);
    return tb * 16 + r;
}

// Reference summation tree over 1024 virtual-thread partials held 4 per lane (norm_quant.hip:ref_tree_sum_1024),
// for RB rows at once.  j = index inside the 256-thread group; red = [RB][32] floats of this group, written once.
// Same pairings, same order as the reference's two xor butterflies (so the same bits):
//   * first butterfly (virtual lane = 4 j + c): lanes ^4, ^2, ^1 as one-instruction DPP adds, the four chains of a row
//     interleaved (dpp_add_tree421_x4), then the c pairs (0,2), (1,3) and the last add in registers;
//   * second butterfly over the 32 warp sums: lane l of a wave takes warp sum l & 31 of row (l >> 5) (two rows per
//     register), and the five levels 16, 8, 4, 2, 1 run ACROSS LANES (ds_swizzle for 16, DPP adds for the rest) --
//     10 instructions for two rows where every thread used to add up all 32 sums of every row (8 broadcast
//     ds_read_b128 + 31 adds per row).  Every lane of a butterfly ends with the same value; lane 0 / 32 is read back
//     as a wave-uniform value.
template <int RB>
__device__ __forceinline__ void tree_sum_rows(float (&p)[RB][4], float* red, int j, float (&out)[RB]) {
    const int lane = j & 63;
#pragma unroll
    for (int i = 0; i < RB; i++) {
        dpp_add_tree421_x4(p[i]);
        const float r0 = p[i][0] + p[i][2], r1 = p[i][1] + p[i][3];
        const float s = r0 + r1;
        if ((j & 7) == 0) red[i * 32 + (j >> 3)] = s;
    }
    __syncthreads();
#pragma unroll
    for (int i0 = 0; i0 < RB; i0 += 2) {
        float x = red[i0 * 32 + (RB == 1 ? (lane & 31) : lane)];
        x = x + swizzle_xor16_f(x);
        x = dpp_add_xor<8>(x);
        x = dpp_add_xor<4>(x);
        x = dpp_add_xor<2>(x);
        x = dpp_add_xor<1>(x);
        out[i0] = readlane_f(x, 0);
        if (RB > 1) out[i0 + 1] = readlane_f(x, 32);
    }
}

// Residual add + LN-no-gamma + int4 quant of all M rows into LDS (xq_lds [MP][RS] bytes, xs_lds [16] floats).
// NG groups of 256 threads, RB rows per group per pass.  Split in a load half and a compute half so that the caller
// can put the first weight batch between them: vector-memory results return in issue order, so the residual
// stream must be requested BEFORE the weights for the norm to run underneath their latency.
// Arithmetic = norm_quant.hip:ln_kernel (layernorm_kernels.cu:569-716), with cheaper but bit-equivalent forms:
// |fp16| maxima compared as fp16, round-to-nearest-even to integer by adding 1.5 * 2^23.
template <int NI, int RB>
struct LnRegs {
    f16x4 x[RB][NI], d[RB][NI];
};

template <int NI, int NG, int RB, bool HASD = true>
__device__ __forceinline__ void ln_load(const StreamArgs& a, int base, LnRegs<NI, RB>& rg) {
    const int tid = threadIdx.x, j = tid & 255, grp = tid >> 8;
    const int H = a.K;
    const f16* dptr = a.delta ? a.delta : a.hidden_in;
#pragma unroll
    for (int i = 0; i < RB; i++) {
        const int row = base + grp * RB + i;
        const int rr = row < a.M ? row : 0;
#pragma unroll
        for (int it = 0; it < NI; it++) {
            // no branch around a load: hipcc answers control flow with s_waitcnt vmcnt(0), which would serialise
            // every load of the prologue (delta == nullptr re-reads hidden_in and the value is discarded)
            rg.x[i][it] = *reinterpret_cast<const f16x4*>(a.hidden_in + (size_t)rr * H + it * 1024 + 4 * j);
            if (HASD) rg.d[i][it] = *reinterpret_cast<const f16x4*>(dptr + (size_t)rr * H + it * 1024 + 4 * j);
        }
    }
}

#ifdef QS_STREAM_STAMPS
#ifdef QS_STAMPS_DRAIN   // drains vmcnt first: "the time until the residual stream has arrived" (perturbs: also waits for the weights)
#define QS_LNSTAMP(i) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lnst[i])::"memory")
#else
#define QS_LNSTAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lnst[i])::"memory")
#endif
__device__ long long g_lnst[8];
#else
#define QS_LNSTAMP(i)
#endif
template <int NI, int NG, int RB, bool HASD = true>
__device__ __forceinline__ void ln_compute(const StreamArgs& a, int base, LnRegs<NI, RB>& rg,
                                           unsigned char* xq_lds, int RS, float* xs_lds,
                                           float* lnred /* [3][NG][RB][32] */, bool write_hidden) {
    static_assert(RB == 1 || RB == 2 || RB == 4, "rows per 256-thread group");
    const int tid = threadIdx.x, j = tid & 255, grp = tid >> 8, lane = tid & 63;
    const int H = a.K;
    float* red_mean = lnred + (0 * NG + grp) * RB * 32;
    float* red_var = lnred + (1 * NG + grp) * RB * 32;
    float* red_max = lnred + (2 * NG + grp) * RB * 32;
    // v[i][it][h] = elements (2h, 2h + 1) of the thread's 4-element piece: the element-wise passes run on pairs
    // (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32: one VALU issue per two elements, each half rounded like the scalar op)
    f32x2 v[RB][NI][2];
    const int row0 = base + grp * RB;   // rows row0 .. row0 + RB - 1 (< MP <= 16: the caller's LDS rows exist even beyond M)
#ifdef QS_STREAM_STAMPS
    long long lnst[8];
#endif
    QS_LNSTAMP(0);   // (stamps build: drains vmcnt first -> the time until the residual stream has arrived)
    if (HASD && a.delta) {  // uniform; no load inside
#pragma unroll
        for (int i = 0; i < RB; i++)
#pragma unroll
            for (int it = 0; it < NI; it++)
#pragma unroll
                for (int c = 0; c < 4; c++) rg.x[i][it][c] = f2h(h2f(rg.x[i][it][c]) + h2f(rg.d[i][it][c]));
    }
#pragma unroll
    for (int i = 0; i < RB; i++) {
        const bool act = row0 + i < a.M;
        const int rr = act ? row0 + i : 0;
#pragma unroll
        for (int it = 0; it < NI; it++) {
            if (HASD && write_hidden && act)
                *reinterpret_cast<f16x4*>(a.hidden_out + (size_t)rr * H + it * 1024 + 4 * j) = rg.x[i][it];
            v[i][it][0] = f32x2{h2f(rg.x[i][it][0]), h2f(rg.x[i][it][1])};
            v[i][it][1] = f32x2{h2f(rg.x[i][it][2]), h2f(rg.x[i][it][3])};
        }
    }
    float p[RB][4], mean[RB], var[RB];
#pragma unroll
    for (int i = 0; i < RB; i++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            // (the reference starts its chain at +0: 0 + v == v except that an all -0 chain ends as +0 there and -0
            // here; a mean of -0 instead of +0 changes no deviation's magnitude and no rounded quotient)
            f32x2 s = v[i][0][h];
#pragma unroll
            for (int it = 1; it < NI; it++) s = s + v[i][it][h];
            p[i][2 * h] = s[0];
            p[i][2 * h + 1] = s[1];
        }
    tree_sum_rows<RB>(p, red_mean, j, mean);
    QS_LNSTAMP(1);
    // x / H for a power of two H is x * (1 / H) bit for bit (both are the one correct rounding of the same real)
    // (compile-time: H = 1024 NI; left to a run-time test hipcc computes the division anyway and selects)
    constexpr bool pow2 = (NI & (NI - 1)) == 0;
    constexpr float invH = 1.0f / (float)(NI > 0 ? NI * 1024 : 1);   // (NI = 0: instantiated but never run)
    // Second pass: sum of squared deviations AND max |deviation|.  The reference takes the maximum over
    // |h((x - mean) * rstd)| after the variance is known (a third block reduction); rstd > 0 and both roundings are
    // monotonic and sign-symmetric, so that maximum is |h(max|x - mean| * rstd)| bit for bit and the maximum can ride
    // on the variance pass's barrier.  The deviations replace the values in registers: the last pass needs only them.
#pragma unroll
    for (int i = 0; i < RB; i++) {
        mean[i] = pow2 ? mean[i] * invH : mean[i] / (float)H;
        const f32x2 m2 = {mean[i], mean[i]};
        float dm = 0.0f;
#pragma unroll
        for (int h = 0; h < 2; h++) {
            f32x2 s;
#pragma unroll
            for (int it = 0; it < NI; it++) {
                const f32x2 d = v[i][it][h] - m2;
                v[i][it][h] = d;
                s = it == 0 ? d * d : __builtin_elementwise_fma(d, d, s);   // fma(d, d, +0) == d * d
                dm = fmaxf(fmaxf(dm, __builtin_fabsf(d[0])), __builtin_fabsf(d[1]));
            }
            p[i][2 * h] = s[0];
            p[i][2 * h + 1] = s[1];
        }
        dm = wave_max_uniform(dm);
        if ((j & 63) == 0) red_max[i * 32 + (j >> 6)] = dm;
    }
    tree_sum_rows<RB>(p, red_var, j, var);
    QS_LNSTAMP(2);
    // Row scalars, ONE IEEE sqrt and TWO IEEE divisions per call instead of one sqrt and five divisions per row (each
    // ~11 dependent VALU instructions that every wave of every workgroup repeats): lane l works on row l % RB; lanes
    // with bit RB clear produce 7 / amax (the quantiser's multiplier), lanes with it set amax / 7 (the stored scale).
    const int li = lane & (RB - 1);
    float tv = pow2 ? var[0] * invH : var[0] / (float)H;
    tv = tv + a.eps;
    f32x4 m4 = *reinterpret_cast<const f32x4*>(red_max + li * 32);
#pragma unroll
    for (int i = 1; i < RB; i++) {
        float t = pow2 ? var[i] * invH : var[i] / (float)H;
        t = t + a.eps;
        tv = li == i ? t : tv;
    }
    const float rl = 1.0f / __builtin_sqrtf(tv);
    const float dmax = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
    const f16 a16 = f2h(dmax * rl), floor16 = f2h(1e-6f);
    const float amax = h2f(a16 > floor16 ? a16 : floor16);
    const bool second = (lane & RB) != 0;
    const float ql = (second ? amax : 7.0f) / (second ? 7.0f : amax);
    if (j >= RB && j < 2 * RB) xs_lds[row0 + j - RB] = h2f(f2h(ql));
    // Last pass.  |(x - mean) * rstd| <= dmax * rstd =: y and amax >= h(y) >= y (1 - 2^-11) (fp16 normal range; below it
    // the absolute rounding error 2^-25 against the floor of 17 * 2^-24 bounds the ratio by 1.03; h(y) = inf gives s = 0),
    // so |t| <= 7.21 for every finite row and rounds into [-7, 7]: the reference's clamp to [-8, 7] never acts and is
    // not executed.  Round to nearest even by adding 1.5 * 2^23 (+ 8): the low mantissa nibble is then q + 8 in 0..15
    // with zeros above it up to bit 22, so the four nibbles of a piece are spliced with three shift-or's, no masks, and
    // one xor with 0x8888 turns the offset nibbles into two's complement.
#pragma unroll
    for (int i = 0; i < RB; i++) {
        const float rstd = readlane_f(rl, i), s = readlane_f(ql, i);
        const f32x2 r2 = {rstd, rstd}, s2 = {s, s}, mg2 = {12582920.0f, 12582920.0f};
#pragma unroll
        for (int it = 0; it < NI; it++) {
            u32 b[4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                f32x2 t = (v[i][it][h] * r2) * s2;
                t = t + mg2;
                float t0 = t[0], t1 = t[1];
                asm("" : "+v"(t0), "+v"(t1));
                b[2 * h] = __builtin_bit_cast(u32, t0);
                b[2 * h + 1] = __builtin_bit_cast(u32, t1);
            }
            u32 w = (b[1] << 4) | b[0];
            w = (b[2] << 8) | w;
            w = (b[3] << 12) | w;
            *reinterpret_cast<uint16_t*>(xq_lds + (size_t)(row0 + i) * RS + it * 512 + 2 * j) = (uint16_t)(w ^ 0x8888u);
        }
    }
    QS_LNSTAMP(3);
    __syncthreads();  // the reduction scratch is reused by the next pass; also publishes xq_lds / xs_lds
    QS_LNSTAMP(4);
#ifdef QS_STREAM_STAMPS
    if (blockIdx.x == 100 && threadIdx.x == 0 && base == 0)
        for (int i = 0; i < 5; i++) g_lnst[i] = lnst[i];
#endif
}

// ---- The same norm + quantiser with ONE WAVE PER ROW: no LDS, no barrier inside.  Lane l of the row's wave plays the
// reference's virtual threads t = 16 l + i, i = 0..15, i.e. holds the 16 CONSECUTIVE elements 1024 it + 16 l + i of every
// 1024-block `it` (two 16-byte loads per block, one 8-byte LDS store):
//   first butterfly (t ^ 16, 8, 4, 2, 1):  lane ^ 1 as DPP adds on all 16 partials, then i ^ 8, 4, 2, 1 in registers;
//   second butterfly over the 32 warp sums (warp = t >> 5 = lane >> 1; levels 16, 8, 4, 2, 1): lane ^ 32 and lane ^ 16 with
//   v_permlane32_swap / v_permlane16_swap, lane ^ 8, 4, 2 as DPP adds.
// Same pairings in the same order as ref_tree_sum_1024, every addition commutative: the same bits.  On the critical path
// of all eight waves this form is SLOWER than ln_compute (four waves x 550 instructions against eight x 430 with three
// barriers); it is what the split forms run on their four norm waves, where having no barrier is the point.
template <int NI>
struct LnwRegs {
    u32x4 x[NI > 0 ? NI : 1][2], d[NI > 0 ? NI : 1][2];
};

template <int NI, bool HASD>
__device__ __forceinline__ void lnw_load(const StreamArgs& a, int row, LnwRegs<NI>& rg) {
    const int lane = threadIdx.x & 63;
    const int rr = row < a.M ? row : 0;   // no branch around a load: clamped, the value is discarded
    const f16* xp = a.hidden_in + (size_t)rr * a.K + 16 * lane;
    const f16* dp = (a.delta ? a.delta : a.hidden_in) + (size_t)rr * a.K + 16 * lane;
#pragma unroll
    for (int it = 0; it < NI; it++)
#pragma unroll
        for (int h = 0; h < 2; h++) {
            rg.x[it][h] = *reinterpret_cast<const u32x4*>(xp + it * 1024 + 8 * h);
            if (HASD) rg.d[it][h] = *reinterpret_cast<const u32x4*>(dp + it * 1024 + 8 * h);
        }
}

__device__ __forceinline__ float lnw_tree(f32x2 (&p)[8]) {
    dpp_add_xor1_x4(p[0], p[1]);
    dpp_add_xor1_x4(p[2], p[3]);
    dpp_add_xor1_x4(p[4], p[5]);
    dpp_add_xor1_x4(p[6], p[7]);
#pragma unroll
    for (int k = 0; k < 4; k++) p[k] = p[k] + p[k + 4];
    p[0] = p[0] + p[2];
    p[1] = p[1] + p[3];
    p[0] = p[0] + p[1];
    float s = p[0][0] + p[0][1];
    s = add_xor32(s);
    s = add_xor16(s);
    s = dpp_add_xor<8>(s);
    s = dpp_add_xor<4>(s);
    s = dpp_add_xor<2>(s);
    return readlane_f(s, 0);
}

// row < a.M (the caller skips the others); the arithmetic, step for step, is ln_compute's (see there for the proofs of
// the maximum riding on the variance pass, the dead clamp and the nibble splice)
template <int NI, bool HASD>
__device__ __forceinline__ void lnw_compute(const StreamArgs& a, int row, LnwRegs<NI>& rg, unsigned char* xq_lds, int RS,
                                            float* xs_lds, bool write_hidden) {
    const int lane = threadIdx.x & 63;
    f32x2 v[NI > 0 ? NI : 1][8];
#ifdef QS_STREAM_STAMPS
    long long lnst[8];
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lnst[0])::"memory");
    asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(lnst[1])::"memory");
#endif
#pragma unroll
    for (int it = 0; it < NI; it++) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            f16x8 x8 = __builtin_bit_cast(f16x8, rg.x[it][h]);
            if (HASD && a.delta) {   // uniform; no load inside
                const f16x8 d8 = __builtin_bit_cast(f16x8, rg.d[it][h]);
#pragma unroll
                for (int e = 0; e < 8; e++) x8[e] = f2h(h2f(x8[e]) + h2f(d8[e]));
            }
            if (HASD && write_hidden)
                *reinterpret_cast<u32x4*>(a.hidden_out + (size_t)row * a.K + it * 1024 + 16 * lane + 8 * h) =
                    __builtin_bit_cast(u32x4, x8);
#pragma unroll
            for (int k = 0; k < 4; k++) v[it][4 * h + k] = f32x2{h2f(x8[2 * k]), h2f(x8[2 * k + 1])};
        }
    }
    f32x2 p[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f32x2 s = v[0][k];
#pragma unroll
        for (int it = 1; it < NI; it++) s = s + v[it][k];
        p[k] = s;
    }
    constexpr bool pow2 = (NI & (NI - 1)) == 0;
    constexpr float invH = 1.0f / (float)(NI > 0 ? NI * 1024 : 1);
    float mean = lnw_tree(p);
    QS_LNSTAMP(2);
    mean = pow2 ? mean * invH : mean / (float)a.K;
    const f32x2 m2 = {mean, mean};
    float dm = 0.0f;
#pragma unroll
    for (int k = 0; k < 8; k++) {
        f32x2 s;
#pragma unroll
        for (int it = 0; it < NI; it++) {
            const f32x2 d = v[it][k] - m2;
            v[it][k] = d;
            s = it == 0 ? d * d : __builtin_elementwise_fma(d, d, s);
            dm = fmaxf(fmaxf(dm, __builtin_fabsf(d[0])), __builtin_fabsf(d[1]));
        }
        p[k] = s;
    }
    float var = lnw_tree(p);
    QS_LNSTAMP(3);
    const float dmax = wave_max_uniform(dm);
    var = pow2 ? var * invH : var / (float)a.K;
    const float rstd = 1.0f / __builtin_sqrtf(var + a.eps);
    const f16 a16 = f2h(dmax * rstd), floor16 = f2h(1e-6f);
    const float amax = h2f(a16 > floor16 ? a16 : floor16);
    // 7 / amax (the multiplier) in the even lanes, amax / 7 (the stored scale) in the odd ones: one division
    const bool second = (lane & 1) != 0;
    const float ql = (second ? amax : 7.0f) / (second ? 7.0f : amax);
    if (lane == 1) xs_lds[row] = h2f(f2h(ql));
    const float sq = readlane_f(ql, 0);
    const f32x2 r2 = {rstd, rstd}, s2 = {sq, sq}, mg2 = {12582920.0f, 12582920.0f};
#pragma unroll
    for (int it = 0; it < NI; it++) {
        u32 half[4];
#pragma unroll
        for (int q4 = 0; q4 < 4; q4++) {   // 4 elements -> 16 bits (upper bits: junk)
            u32 b[4];
#pragma unroll
            for (int h = 0; h < 2; h++) {
                f32x2 t = (v[it][2 * q4 + h] * r2) * s2;
                t = t + mg2;
                float t0 = t[0], t1 = t[1];
                asm("" : "+v"(t0), "+v"(t1));
                b[2 * h] = __builtin_bit_cast(u32, t0);
                b[2 * h + 1] = __builtin_bit_cast(u32, t1);
            }
            u32 w = (b[1] << 4) | b[0];
            w = (b[2] << 8) | w;
            half[q4] = (b[3] << 12) | w;
        }
        // bytes 0, 1 of the even piece, bytes 0, 1 of the odd piece
        u32x2 o;
        o[0] = __builtin_amdgcn_perm(half[1], half[0], 0x05040100u) ^ 0x88888888u;
        o[1] = __builtin_amdgcn_perm(half[3], half[2], 0x05040100u) ^ 0x88888888u;
        *reinterpret_cast<u32x2*>(xq_lds + (size_t)row * RS + it * 512 + 8 * lane) = o;
    }
#ifdef QS_STREAM_STAMPS
    QS_LNSTAMP(4);
    if (blockIdx.x == 100 && row == 0 && lane == 0)
        for (int i = 0; i < 5; i++) g_lnst[i] = lnst[i];
#endif
}

// global_load_lds_dwordx4: 16 bytes per lane from memory straight into LDS (lane-linear image at lds_dst); nt = non-temporal.
template <int NT>
__device__ __forceinline__ void glds16(const void* gsrc, u32 lds_dst) {
    u32 keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
#ifdef QS_EXPERIMENTAL
#include "experimental/w4a4_engine.inc"   // loader / consumer ("engine") form of the draft GEMMs: QSPEC_ENGINE=1
#endif

// NW waves; UB steps of 64 packed bytes per wave and batch (K/2 = 64 * NW * UB * NB bytes, NB batches per tile);
// NI = K / 1024 for the LN prologue (0 otherwise).
#ifdef QS_STREAM_STAMPS
__device__ long long g_sstamps[8];
__device__ long long g_wgspan[1024][2];   // per workgroup: first / last instruction of thread 0, s_memrealtime (100 MHz, chip-wide)
#define QS_SSTAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i])::"memory")
#else
#define QS_SSTAMP(i)
#endif
// MT = 16-token tiles per workgroup (1: M <= 16; 2: M <= 32, PRO_Q only, activation fragments kept PACKED in registers
// and widened at each use -- 8 VGPRs per step and tile would not fit beside the weight ring at 1024 threads).
// PRO_LNS / PRO_LN1S ("split", M <= 4): the workgroup has NW + 4 waves.  Waves 0 .. NW-1 are the STREAM waves: they request
// the workgroup's first 1 + PF weight tiles into registers right away and then wait; waves NW .. NW+3 are NORM waves: one
// row each (lnw_compute: no barrier inside), then they retire.  One barrier joins them.  Why: a CU holds ~32 KB of loads in
// flight and a wave that issues a load beyond that stalls AT THE ISSUE, so weight requests inside the norm's instruction
// stream stop the norm ("Measured and rejected") and HBM idled underneath it; waves that have nothing else to
// do can stall, provided no barrier of the norm waits for them.
template <int PRO>
constexpr bool pro_split() { return PRO == PRO_LNS || PRO == PRO_LN1S; }
// Tiles beyond the first that the stream waves request up front.  Measured in the engine (same box, ms per cycle): unsplit
// 7.96, split with 1 / 2 extra tiles 7.92 / 7.95; gate_up with 3 / 4 / 5 extra tiles 14.3 / 14.5 / 15.2 us per launch
// against 13.4 -- also with a vmcnt(0) between the tiles' requests (13.3 / 13.7 / 14.0 / 14.6 us for 1 / 2 / 3 / 4): the
// stream waves reach the joining barrier only when their last request has been ISSUED, i.e. when all but one of the
// requested tiles have returned, and the serial consumption of the hoard then costs more than the overlap won.
#ifndef QS_LN_PF
#define QS_LN_PF 1
#endif
// DMA > 0 (split forms, M <= 4): the workgroup's tiles 1 .. DMA are brought in by FOUR LOADER WAVES with LDS-DMA
// (global_load_lds_dwordx4 nt) from the first microsecond on, beside tile 0 in the stream waves' registers.  What bounds a
// register-streaming CU is ~32 KB of loads in flight (a wave stalls at the issue beyond that); LDS-DMA loads are not subject
// to it (scripts/micro/ldsdma.hip: four loader waves keep > 100 KB in flight and reach 7.3 TB/s over the chip), so the
// bytes of the next tiles travel underneath the norm prologue and tile 0's latency instead of behind them.  The LDS image
// of a tile is the register image (load j = stream wave j / UB, step j % UB: lane-linear, no swizzle), consumed with
// ds_read_b128.  No flags: the loaders take part in the stream waves' barriers and wait for their tile's loads
// (`s_waitcnt vmcnt`) in front of the barrier after which the tile is read -- the barrier IS the publication -- and retire.
template <int EPI>
constexpr int finish_barriers() { return (EPI == SEPI_QKV || EPI == SEPI_GATEUP) ? 2 : 1; }
template <int N>
__device__ __forceinline__ void vmcnt_le() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int EPI, int PRO, int NW, int UB, int NI, int MT = 1, int DMA = 0>
__global__ __launch_bounds__(NW * 64 + (pro_split<PRO>() ? 256 : 0) + (DMA > 0 ? 256 : 0)) void gemm_w4a4_stream_kernel(StreamArgs a) {
    static_assert(MT == 1 || (PRO == PRO_Q && NW >= 8), "two token tiles: (xq, xs) input, >= 512 threads for the epilogue");
    static_assert(EPI != SEPI_IPART || PRO == PRO_Q, "K-slice partials: (xq, xs) input");
    static_assert(DMA == 0 || (pro_split<PRO>() && MT == 1 && NW * UB == 32 && DMA <= 3), "LDS-DMA tiles: split forms, 32 KiB tiles");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
#ifdef QS_STREAM_STAMPS
    long long stamp[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    QS_SSTAMP(0);
#ifdef QS_STREAM_STAMPS
    long long wg_t0;
    asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(wg_t0)::"memory");
#endif
    constexpr int NG = NW / 4;
    constexpr int RB = NW == 4 ? 4 : (NW == 8 ? 2 : 1);
    constexpr bool SPLIT = pro_split<PRO>();
    const int tid = threadIdx.x, lane = tid & 63;   // (split: the norm waves are tid >= NW * 64: no epilogue thread among them)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int Kb = a.K >> 1, RS = Kb + 32;
    // SEPI_IPART: this workgroup's K slice starts kofs bytes into every row; rows are ldw / ldx bytes apart
    const size_t kofs = EPI == SEPI_IPART ? (size_t)blockIdx.y * Kb : 0;
    const size_t ldwb = EPI == SEPI_IPART ? (size_t)a.ldw : (size_t)Kb, ldxb = EPI == SEPI_IPART ? (size_t)a.ldx : (size_t)Kb;
    const int MP = a.M <= 4 ? 4 : (a.M <= 8 ? 8 : 16);
    unsigned char* xq_lds = smem;
    float* xs_lds = reinterpret_cast<float*>(smem + (PRO == PRO_Q ? (size_t)0 : (size_t)MP * RS));   // (xq, xs) input: no staged rows
    int* red = reinterpret_cast<int*>(xs_lds + 16);                 // [2][NW][MT][256]
    f16* ex = reinterpret_cast<f16*>(red + 2 * NW * MT * 256);       // [2][MT * 256]
    float* lnred = reinterpret_cast<float*>(ex + 2 * MT * 256);      // [3][NG][RB][32]
    const int NB = Kb / (64 * NW * UB);                              // batches per tile (exact: checked on the host)
    unsigned char* dma_lds = reinterpret_cast<unsigned char*>(lnred + 3 * NG * RB * 32);   // [DMA][32 KiB] (1 KiB aligned by the host)
    dma_lds = smem + (((size_t)(dma_lds - smem) + 1023) & ~(size_t)1023);
    if constexpr (DMA > 0) {
        if (tid >= (NW + 4) * 64) {   // ---- loader waves (wave-uniform): issue, mirror the barriers, retire
            const int lw = wave - (NW + 4);
            const int mt = (a.ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
            const int nd = min(DMA, mt - 1);   // tiles this workgroup takes through LDS
            const u32 base3 = (u32)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)dma_lds;
            __builtin_amdgcn_s_barrier();   // #0 (the norm waves' row requests are out)
            for (int d = 1; d <= nd; d++) {
                const int td = blockIdx.x + d * (int)gridDim.x;
                const uint8_t* wp = a.wq + (size_t)stile_row<EPI>(td, lane & 15, a.I, a.hdl) * Kb + (lane >> 4) * 16;
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    const int j = lw + 4 * i;   // load j of the tile = (stream wave j / UB, step j % UB)
                    glds16<1>(wp + step_off<NW, UB>(j / UB, j % UB), base3 + (u32)((d - 1) * 32 + j) * 1024u);
                }
            }
            __builtin_amdgcn_s_barrier();   // #1
            for (int d = 1; d <= nd; d++) {   // the barriers of finish(tile d - 1); tile d is read behind the last of them
#pragma unroll
                for (int b2 = 0; b2 < finish_barriers<EPI>() - 1; b2++) __builtin_amdgcn_s_barrier();
                const int left = nd - d;      // tiles still allowed in flight (8 loads each)
                if (left >= 2) vmcnt_le<16>();
                else if (left == 1) vmcnt_le<8>();
                else vmcnt_le<0>();
                __builtin_amdgcn_s_barrier();
            }
            return;
        }
    }

    // epilogue thread (tid < 256) owns accumulator element (token m, tile column c)
    // epilogue thread t owns output (token m = t / 16, tile column c = t % 16): for M <= 4 that is wave 0 alone, the
    // other waves go straight back to streaming.  ridx = where the MFMA left that element in a wave's accumulators.
    const int c = tid & 15, m = tid >> 4;
    const bool ethread = m < a.M;
    const int ridx = (m >> 4) * 256 + (m & 3) * 64 + ((m >> 2) & 3) * 16 + c;   // (token tile, element) in a wave's block
    const int mc = m < a.M ? m : 0;

    // No load in this kernel sits behind a branch: hipcc resolves control flow around vector-memory operations
    // with s_waitcnt vmcnt(0), which would drain the weight stream.  Addresses are clamped instead.
    int64_t pos_m = 0, slot_m = -1;
    if (EPI == SEPI_QKV) {  // the kernel's first loads
        pos_m = a.positions[mc];
        slot_m = a.slot_mapping[mc];
    }
    struct Pre {
        f16 swn, cf, sf;
    };
    auto load_pre = [&](Pre& pre, int tile) {  // epilogue operands of (token m, column c) of `tile`
        if (EPI != SEPI_IPART) pre.swn = a.ws[stile_row<EPI>(tile, c, a.I, a.hdl)];
        if (EPI == SEPI_RESID) pre.cf = a.resid_in[(size_t)mc * a.N + tile * 16 + c];   // the residual element
        if (EPI == SEPI_QKV) {
            const int o = qkv_pair(tile, c & 7, a.I, a.hdl).i;
            const f16* cs = a.cos_sin_cache + (pos_m << a.hdl);   // rot_dim = head size
            pre.cf = cs[o];
            pre.sf = cs[(1 << (a.hdl - 1)) + o];
        }
    };
    auto wptr = [&](int tile, int b) -> const uint8_t* {
        return a.wq + (size_t)stile_row<EPI>(tile, r, a.I, a.hdl) * ldwb + kofs + (size_t)(b * UB * NW) * 64 + g * 16;
    };
    // A wave owns the same K steps of every tile (one batch per tile: NB == 1, checked on the host), so its
    // activation fragments are widened ONCE and stay in registers: no per-step LDS read, and with (xq, xs) given
    // (PRO_Q) no LDS staging and no barrier in front of the first MFMA at all.
    const unsigned char* arow = xq_lds + (size_t)(r & (MP - 1)) * RS + g * 16;
    i32x4 af0[UB], af1[UB];
    u32x4 apk[MT][UB];   // MT == 2: packed fragments
    float xs_m = 0.0f;   // activation scale of this epilogue thread's token
    i32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = i32x4{0, 0, 0, 0};
    auto use = [&](const u32x4& w, int b, int u) {
        const i32x4 b0 = widen16(w[0], w[1]), b1 = widen16(w[2], w[3]);
        if constexpr (MT == 1) {
            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af0[u], b0, acc[0], 0, 0, 0);
            acc[0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(af1[u], b1, acc[0], 0, 0, 0);
        } else {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                // the widening is loop invariant (a wave meets the same K steps in every tile): left visible, hipcc
                // hoists it and keeps 8 VGPRs per step and tile live -- spills at 1024 threads.  Opaque copies keep it here.
                // (Up to 512 threads the budget is 256 VGPRs: there the hoist is what we want -- the fragments are widened
                // once and config 3's gate_up no longer spends a third of its issue cycles re-widening them.)
                u32 p0 = apk[mt][u][0], p1 = apk[mt][u][1], p2 = apk[mt][u][2], p3 = apk[mt][u][3];
                if constexpr (NW > 8 || UB > 7) asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));   // (8 x 8 steps: 1-4 spills)
                acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(widen16(p0, p1), b0, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(widen16(p2, p3), b1, acc[mt], 0, 0, 0);
            }
        }
    };
    auto finish = [&](int tile, int par, const Pre& pre) {
        int* rb = red + par * NW * MT * 256;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int i = 0; i < 4; i++) rb[(wave * MT + mt) * 256 + i * 64 + lane] = acc[mt][i];
            acc[mt] = i32x4{0, 0, 0, 0};
        }
        __syncthreads();
        f16 hv = (f16)0.0f;
        if (ethread) {
            int sum = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; w2++) sum += rb[w2 * MT * 256 + ridx];
            if (EPI == SEPI_IPART) {   // the slice's raw sum (both operands carried a factor 16: an exact multiple of 256)
                a.ipart[((size_t)blockIdx.y * a.M + m) * a.N + tile * 16 + c] = sum >> 8;
                return;
            }
            const float v = ((float)(sum >> 8) * xs_m) * h2f(pre.swn);  // both operands carried a factor 16
            hv = f2h(v);
        }
        if (EPI == SEPI_IPART) return;
        if (EPI == SEPI_PLAIN) {
            if (ethread) a.out[(size_t)m * a.N + tile * 16 + c] = hv;
            return;
        }
        if (EPI == SEPI_RESID) {   // hidden = residual + proj_out, an fp16 add of two fp16 tensors (quarot_llama.py:380,390)
            if (ethread) a.resid_out[(size_t)m * a.N + tile * 16 + c] = f2h(h2f(pre.cf) + h2f(hv));
            return;
        }
        f16* e = ex + par * MT * 256;
        if (tid < MT * 256) e[tid] = hv;
        __syncthreads();
        if (!ethread) return;
        const f16 partner = e[tid ^ 8];
        if (EPI == SEPI_GATEUP) {
            if (c < 8) {  // hv = up, partner = gate
                const float gt = h2f(partner);
                const float act = h2f(f2h(gt / (1.0f + qexpf(-gt))));
                a.out[(size_t)m * a.I + tile * 8 + c] = f2h(act * h2f(hv));
            }
            return;
        }
        // SEPI_QKV
        const QkvPair qp = qkv_pair(tile, c & 7, a.I, a.hdl);
        if (!qp.valid) return;   // a half tile's repeated slots
        const int head = qp.head, o = qp.i;
        const int n = (head << a.hdl) + ((c >> 3) << (a.hdl - 1)) + o;
        f16 res = hv;
        if (head < a.nq + a.nkv) {
            const float cff = h2f(pre.cf), sff = h2f(pre.sf);
            const float xf = h2f(c < 8 ? hv : partner), yf = h2f(c < 8 ? partner : hv);
            res = c < 8 ? f2h(h2f(f2h(xf * cff)) - h2f(f2h(yf * sff))) : f2h(h2f(f2h(yf * cff)) + h2f(f2h(xf * sff)));
        }
        a.out[(size_t)m * a.N + n] = res;
        if (head >= a.nq && slot_m >= 0) {
            const bool is_k = head < a.nq + a.nkv;
            const int kvh = is_k ? head - a.nq : head - a.nq - a.nkv;
            f16* cache = is_k ? a.key_cache : a.value_cache;
            cache[((slot_m * a.nkv + kvh) << a.hdl) + ((c >> 3) << (a.hdl - 1)) + o] = res;
        }
    };

    int tile = blockIdx.x, b = 0, par = 0;
    const int my_tiles = (a.ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x;  // >= 1: grid <= ntiles
    const int n_units = my_tiles * NB;
    u32x4 w[UB];
    Pre pre = {};
    const uint8_t* wp0 = wptr(tile, 0);
    // split forms: tiles beyond the first that the stream waves request underneath the norm (one batch per tile there)
    constexpr int PF = (SPLIT && DMA == 0) ? QS_LN_PF : 0;
    constexpr int SKIP = DMA > 0 ? DMA : PF;   // tiles behind tile 0 that do not come through `w`
    Pre dpre[DMA > 0 ? DMA : 1];
    u32x4 pfw[PF > 0 ? PF : 1][UB];
    Pre pfpre[PF > 0 ? PF : 1];

    // ---- prologue: activations -> LDS.  Their loads are issued BEFORE the first weight batch (results return in
    // issue order), the arithmetic runs underneath the weights' latency.
    if (PRO == PRO_LNH) {
        // The norm is NOT recomputed by every workgroup: the first P workgroups ("producers") each normalise and
        // quantise NG rows (one per 256-thread group) and publish the packed rows through memory; every workgroup
        // meanwhile has its first weight batch in flight, waits for the P flags and copies the M x K/2 bytes into
        // LDS.  Hand-off without fences (MI355X_MICROARCH.md, "Valid forms"): write-through (sc0 sc1) stores, every
        // storing wave drains vmcnt, workgroup barrier, one lane's flag store; the consumer's polling lane sees the
        // flags, the workgroup barrier releases the other waves, every load of the published bytes is an sc1 load.
        // The call cleans up after itself: each workgroup takes a ticket once it has read the rows, the last one
        // zeroes flags and ticket counter for the next call (launch boundary orders that against the next call).
        const int P = (a.M + NG - 1) / NG;
        int* flags = a.sync;
        int* done = a.sync + 16;
        unsigned char* xq_glob = reinterpret_cast<unsigned char*>(a.sync + 32);
        u32* xs_glob = reinterpret_cast<u32*>(xq_glob + (size_t)16 * Kb);
        if ((int)blockIdx.x < P) {
            LnRegs<NI, 1> rg;
            const int base = blockIdx.x * NG;
            ln_load<NI, NG, 1>(a, base, rg);
            ln_compute<NI, NG, 1>(a, base, rg, xq_lds, RS, xs_lds, lnred, a.hidden_out != nullptr);
            const int nrows = min(NG, a.M - base), per_row = Kb >> 3;
            for (int i = tid; i < nrows * per_row; i += NW * 64) {
                const int row = base + i / per_row, q8 = i % per_row;
                const uint64_t v = *reinterpret_cast<const uint64_t*>(xq_lds + (size_t)row * RS + q8 * 8);
                __hip_atomic_store(reinterpret_cast<uint64_t*>(xq_glob + (size_t)row * Kb + q8 * 8), v, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < nrows)
                __hip_atomic_store(&xs_glob[base + tid], __builtin_bit_cast(u32, xs_lds[base + tid]), __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) __hip_atomic_store(&flags[blockIdx.x], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
        load_pre(pre, tile);
        QS_SSTAMP(1);
        if (tid == 0) {
            int ok = 0;
            for (int guard = 0; guard < (1 << 22) && !ok; guard++) {   // bounded: a lost producer must not hang the GPU
                ok = 1;
                for (int i = 0; i < P; i++)
                    ok &= __hip_atomic_load(&flags[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
                if (!ok) __builtin_amdgcn_s_sleep(4);
            }
            // ... and must not pass silently either: sticky error word of the workspace (word 31), read by the host once
            // per cycle (QSpecEngine.error_flag)
            if (!ok) __hip_atomic_store(&a.sync[31], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        {
            const int per_row = Kb >> 3, total = a.M * per_row;
            constexpr int FI = 2;   // M = 4, K = 4096, 512 threads: both items of a thread in flight together
            uint64_t v[FI];
            int off[FI];
#pragma unroll
            for (int f = 0; f < FI; f++) {
                const int i = min(tid + f * NW * 64, total - 1);
                const int row = i / per_row, q8 = i - row * per_row;
                off[f] = row * RS + q8 * 8;
                v[f] = __hip_atomic_load(reinterpret_cast<const uint64_t*>(xq_glob + (size_t)row * Kb + q8 * 8),
                                         __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            const u32 xsb = __hip_atomic_load(&xs_glob[tid < a.M ? tid : 0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int f = 0; f < FI; f++) *reinterpret_cast<uint64_t*>(xq_lds + off[f]) = v[f];
            for (int i = tid + FI * NW * 64; i < total; i += NW * 64) {
                const int row = i / per_row, q8 = i - row * per_row;
                *reinterpret_cast<uint64_t*>(xq_lds + (size_t)row * RS + q8 * 8) =
                    __hip_atomic_load(reinterpret_cast<const uint64_t*>(xq_glob + (size_t)row * Kb + q8 * 8),
                                      __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (tid < a.M) xs_lds[tid] = __builtin_bit_cast(float, xsb);
        }
        __syncthreads();
        if (tid == 0) {
            const int ticket = __hip_atomic_fetch_add(done, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (ticket == (int)gridDim.x - 1) {
                for (int i = 0; i < P; i++) __hip_atomic_store(&flags[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    } else if (PRO == PRO_RQ) {
        // thread tid owns the 16-byte chunks tid, tid + 512, ... of every row (K = 4096: 512 chunks per row = the 512
        // threads; K = 5120: 640 chunks, the first 128 threads take a second one)
        constexpr int NCH = NI * 128;                          // 16-byte chunks of a row (K = 1024 NI)
        constexpr int CPT = (NCH + NW * 64 - 1) / (NW * 64);   // chunks per thread
        u32x4 xr[4][CPT];
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
            for (int cc = 0; cc < CPT; cc++)   // (clamped: no branch around a load; a clamped chunk is never stored)
                xr[i][cc] = *reinterpret_cast<const u32x4*>(a.x16 + (size_t)min(i, a.M - 1) * a.K + 8 * min(tid + cc * NW * 64, NCH - 1));
        float pv = a.part_amax[min(lane & 31, a.M * 8 - 1)];   // lanes 8 r .. 8 r + 7: the eight partial maxima of row r
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();   // every wave's row requests before any weight request (in-order L1, see PRO_LN)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
        __builtin_amdgcn_sched_barrier(0);
        load_pre(pre, tile);
        __builtin_amdgcn_sched_barrier(0);
        pv = dpp_max_xor<1>(pv);
        pv = dpp_max_xor<2>(pv);
        pv = dpp_max_xor<4>(pv);
        // scale = h(h(amax / 7) * h(clip)) and its correctly rounded reciprocal, lane-parallel (lane 8 r works on row r)
        const f16 scl = f2h(h2f(f2h(pv / 7.0f)) * h2f(f2h(a.clip)));
        const float scfl = h2f(scl), rcfl = 1.0f / scfl;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const float scf = readlane_f(scfl, 8 * i), rcf = readlane_f(rcfl, 8 * i);
#pragma unroll
            for (int cc = 0; cc < CPT; cc++) {
                const f16x8 x8 = __builtin_bit_cast(f16x8, xr[i][cc]);
                // rni_sat(h(x / scale), -8, 7) as: clamp, then round to nearest even by adding 1.5 * 2^23 -- the two's
                // complement integer is the low mantissa bits.  (The quotient of finite values by a non-zero scale is never
                // NaN; an all-zero row has scale 0, every quotient NaN and the reference's 0 for it: selected per row below.)
                u32 pk = 0;
#pragma unroll
                for (int c = 0; c < 8; c++) {
                    float dq = h2f(f2h(div3_h(h2f(x8[c]), rcf, scf)));
                    dq = __builtin_amdgcn_fmed3f(dq, -8.0f, 7.0f);
                    float mg = dq + 12582912.0f;
                    asm("" : "+v"(mg));
                    pk |= (__builtin_bit_cast(u32, mg) & 0xFu) << (4 * c);
                }
                if (scf == 0.0f) pk = 0;   // uniform
                const int ch = tid + cc * NW * 64;
                if (i < a.M && (CPT * NW * 64 == NCH || ch < NCH)) *reinterpret_cast<u32*>(xq_lds + (size_t)i * RS + 4 * ch) = pk;
            }
            if (i < a.M && tid == 0) xs_lds[i] = scf;
        }
        __syncthreads();   // publishes xq_lds / xs_lds
    } else if (SPLIT) {
        constexpr bool HASD = PRO == PRO_LNS;
        if (tid >= NW * 64) {   // ---- norm waves (wave-uniform branch): row = wave - NW, then retire
            const int row = wave - NW;
            LnwRegs<NI> rg;
            lnw_load<NI, HASD>(a, row, rg);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();   // #0: every row request is out before any weight request (in-order L1)
            __builtin_amdgcn_sched_barrier(0);
            if (row < a.M) lnw_compute<NI, HASD>(a, row, rg, xq_lds, RS, xs_lds, blockIdx.x == 0 && a.hidden_out != nullptr);
            __syncthreads();                // #1: publishes xq_lds / xs_lds
            return;                         // retired waves leave the workgroup's later barriers
        }
        __builtin_amdgcn_s_barrier();       // #0
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
        __builtin_amdgcn_sched_barrier(0);
        load_pre(pre, tile);
        __builtin_amdgcn_sched_barrier(0);
        // The next PF tiles, where the workgroup has them (uniform branches around loads cost nothing HERE: a CU holds one
        // tile's worth of loads in flight, so tile d + 1 could not be requested before tile d has returned anyway, and
        // this wave has nothing to do but wait).
#pragma unroll
        for (int d = 1; d <= PF; d++) {
            if (d < my_tiles) {
                const uint8_t* wpd = wptr(tile + d * (int)gridDim.x, 0);
#pragma unroll
                for (int u = 0; u < UB; u++) pfw[d - 1][u] = wload<u32x4>(wpd + step_off<NW, UB>(wave, u));
                load_pre(pfpre[d - 1], tile + d * (int)gridDim.x);
            }
        }
        if constexpr (DMA > 0) {
#pragma unroll
            for (int d = 1; d <= DMA; d++)   // (clamped: no branch around a load; unused beyond my_tiles)
                load_pre(dpre[d - 1], min(tile + d * (int)gridDim.x, a.ntiles - 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        QS_SSTAMP(1);
        // #1 (a bare barrier: __syncthreads() is fine too, the loads above are what this wave waits for next anyway; the
        // norm waves drained their LDS writes in front of theirs, the clobber keeps this wave's LDS reads behind it)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
    } else if (PRO == PRO_LN || PRO == PRO_LN1) {
        constexpr bool HASD = PRO == PRO_LN;
        LnRegs<NI, RB> rg;
        ln_load<NI, NG, RB, HASD>(a, 0, rg);
        __builtin_amdgcn_sched_barrier(0);
        // The CU's vector L1 returns data in request order ACROSS waves, and its address path takes the eight waves' row
        // requests one by one (~1 k cycles from the first to the last; the waves themselves start within 30 cycles of
        // each other, scripts/micro/wave_start.hip): a wave whose (L2-resident) rows were requested behind an earlier
        // wave's weight loads gets them only when those HBM misses have returned, and the norm's first barrier waited
        // ~2.5 k cycles for that wave (in-kernel stamps).  So: every wave's row requests first (a bare barrier, no
        // memory wait), then the weights.
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
        __builtin_amdgcn_sched_barrier(0);   // (else hipcc hoists load_pre's wait for the position load above the weight loads)
        load_pre(pre, tile);
        __builtin_amdgcn_sched_barrier(0);
        const bool wh = blockIdx.x == 0 && a.hidden_out != nullptr;
        QS_SSTAMP(1);
        ln_compute<NI, NG, RB, HASD>(a, 0, rg, xq_lds, RS, xs_lds, lnred, wh);
        for (int base = NG * RB; base < a.M; base += NG * RB) {
            ln_load<NI, NG, RB, HASD>(a, base, rg);
            ln_compute<NI, NG, RB, HASD>(a, base, rg, xq_lds, RS, xs_lds, lnred, wh);
        }
    } else {
        u32x4 araw[UB];
        const unsigned char* xrow = reinterpret_cast<const unsigned char*>(a.xq) + (size_t)(r < a.M ? r : 0) * ldxb + kofs + g * 16;
#pragma unroll
        for (int u = 0; u < UB; u++) araw[u] = *reinterpret_cast<const u32x4*>(xrow + step_off<NW, UB>(wave, u));
        if constexpr (MT > 1) {
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int row = mt * 16 + r;
                const unsigned char* xr = reinterpret_cast<const unsigned char*>(a.xq) + (size_t)(row < a.M ? row : 0) * ldxb + kofs + g * 16;
#pragma unroll
                for (int u = 0; u < UB; u++) apk[mt][u] = *reinterpret_cast<const u32x4*>(xr + step_off<NW, UB>(wave, u));
            }
        }
        f16 xsh = (f16)0.0f;
        if (EPI != SEPI_IPART) xsh = a.xs[mc];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
        __builtin_amdgcn_sched_barrier(0);   // (else hipcc hoists load_pre's wait for the position load above the weight loads)
        load_pre(pre, tile);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < UB; u++) {
            af0[u] = widen16(araw[u][0], araw[u][1]);
            af1[u] = widen16(araw[u][2], araw[u][3]);
        }
        xs_m = h2f(xsh);
    }
    if (PRO != PRO_Q) {   // the norm left the packed rows in LDS: take this wave's fragments out once
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const u32x4 av = *reinterpret_cast<const u32x4*>(arow + step_off<NW, UB>(wave, u));
            af0[u] = widen16(av[0], av[1]);
            af1[u] = widen16(av[2], av[3]);
        }
        xs_m = xs_lds[mc];
    }

    // ---- main loop: one unit = UB steps of every wave.  Each step's register is refilled with the same step of
    // the NEXT unit as soon as it has been consumed: UB loads per wave stay in flight across tile boundaries,
    // reductions and epilogues.  The last unit is peeled (nothing to refill), so the loop body has no branch
    // around a load and the waits hipcc inserts are the exact counted ones.
    QS_SSTAMP(2);
    int n_left = n_units;
    if constexpr (SKIP > 0) {   // split forms (one batch per tile): tile 0 from w, tiles 1 .. SKIP from registers / from LDS
        {
            const bool more = SKIP + 1 < my_tiles;   // uniform
            const int tn = tile + (SKIP + 1) * (int)gridDim.x;
            Pre npre = pre;
            if (more) load_pre(npre, tn);          // (a branch around loads: everything in flight has arrived by now)
#pragma unroll
            for (int u = 0; u < UB; u++) use(w[u], 0, u);
            if (more) {
                const uint8_t* wp = wptr(tn, 0);
#pragma unroll
                for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp + step_off<NW, UB>(wave, u));
            }
            finish(tile, par, pre);
            par ^= 1;
            pre = npre;
        }
        if constexpr (DMA > 0) {
#pragma unroll
            for (int d = 1; d <= DMA; d++) {
                if (d < my_tiles) {   // uniform.  Tile d has landed: the loaders waited for it in front of the last barrier
                    const unsigned char* tl = dma_lds + (size_t)((d - 1) * 32 + wave * UB) * 1024 + lane * 16;
                    u32x4 lw4[UB];
#pragma unroll
                    for (int u = 0; u < UB; u++) lw4[u] = *reinterpret_cast<const u32x4*>(tl + u * 1024);
#pragma unroll
                    for (int u = 0; u < UB; u++) use(lw4[u], 0, u);
                    finish(tile + d * (int)gridDim.x, par, dpre[d - 1]);
                    par ^= 1;
                }
            }
        } else {
#pragma unroll
            for (int d = 1; d <= PF; d++) {
                if (d < my_tiles) {   // uniform; no load inside
#pragma unroll
                    for (int u = 0; u < UB; u++) use(pfw[d - 1][u], 0, u);
                    finish(tile + d * (int)gridDim.x, par, pfpre[d - 1]);
                    par ^= 1;
                }
            }
        }
        tile += (SKIP + 1) * (int)gridDim.x;
        n_left = my_tiles - (SKIP + 1);
    }
    if (SKIP == 0 || n_left > 0) {
    for (int q = 0; q < n_left - 1; q++) {
        int nb = b + 1, nt = tile;
        if (nb == NB) {
            nb = 0;
            nt = tile + gridDim.x;
        }
        const uint8_t* wp = wptr(nt, nb);
        Pre npre;
        load_pre(npre, nt);
#pragma unroll
        for (int u = 0; u < UB; u++) {
            use(w[u], b, u);
            // pin the refill right behind its consumer: left alone, the scheduler sinks all UB loads below the last
            // use and the stream drains every unit
            __builtin_amdgcn_sched_barrier(0);
            w[u] = wload<u32x4>(wp + step_off<NW, UB>(wave, u));
            __builtin_amdgcn_sched_barrier(0);
        }
        if (b == NB - 1) {
            finish(tile, par, pre);
            par ^= 1;
        }
        pre = npre;
        tile = nt;
        b = nb;
    }
    QS_SSTAMP(3);
#pragma unroll
    for (int u = 0; u < UB; u++) use(w[u], b, u);
    QS_SSTAMP(4);
    finish(tile, par, pre);
    }
    QS_SSTAMP(5);
#ifdef QS_STREAM_STAMPS
    if (PRO != PRO_Q && a.hidden_out && blockIdx.x == 100 && tid == 0) {  // debug build only: stamps into the tail of hidden_out
        long long* sb = reinterpret_cast<long long*>(a.hidden_out + (size_t)a.M * a.K) - 8;
        for (int i = 0; i < 6; i++) sb[i] = stamp[i];
        long long* sb2 = sb - 8;
        for (int i = 0; i < 5; i++) sb2[i] = g_lnst[i];
    }
    // forms without a hidden_out (LN1 / LN1S gate_up): into a device array read back by qspec_debug_stamps()
    if (EPI == SEPI_GATEUP && !a.hidden_out && blockIdx.x == 100 && tid == 0)
        for (int i = 0; i < 6; i++) g_sstamps[i] = stamp[i];
    if (EPI == SEPI_GATEUP && !a.hidden_out && tid == 0 && blockIdx.x < 1024) {
        long long t1;
        asm volatile("s_waitcnt vmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        g_wgspan[blockIdx.x][0] = wg_t0;
        g_wgspan[blockIdx.x][1] = t1;
    }
#endif
}

// ------------------------------------------------------------------ long rows: several batches per tile
// K = 28672 (Llama-3-70B's down_proj: 224 steps of 64 bytes per weight row) does not fit the one-batch-per-tile kernel above:
// 16 waves x 14 steps would hold 112 VGPRs of widened activation fragments per lane against a budget of 128 at 1024 threads.
// Here a tile is NB batches of NW x UB steps; a wave keeps its activation fragments PACKED (4 VGPRs per step and batch,
// widened at the use: 8 VALU beside the 2 MFMAs of a step -- this kernel's waves wait for memory 90 % of their cycles) and
// refills each weight register with the same step of the NEXT batch -- the next tile's first batch behind the last one --
// right behind its consumer, so UB KiB per wave stay in flight across tile boundaries exactly as above.  (xq, xs) input,
// M <= 16, plain / residual epilogue; same int32 sums, same epilogue expression, same bits as gemm.hip's kernel.
template <int EPI, int NW, int UB, int NB>
__global__ __launch_bounds__(NW * 64) void gemm_w4a4_longk_kernel(StreamArgs a) {
    static_assert(EPI == SEPI_PLAIN || EPI == SEPI_RESID, "epilogues built so far");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int* red = reinterpret_cast<int*>(smem);   // [2][NW][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int Kb = a.K >> 1;
    const int c = tid & 15, m = tid >> 4;
    const bool ethread = m < a.M;
    const int ridx = (m & 3) * 64 + ((m >> 2) & 3) * 16 + c;
    const int mc = m < a.M ? m : 0;
    constexpr int BB = UB * NW * 64;           // bytes of a weight row per batch
    struct Pre {
        f16 swn, cf;
    };
    auto load_pre = [&](Pre& pre, int tile) {
        pre.swn = a.ws[tile * 16 + c];
        if (EPI == SEPI_RESID) pre.cf = a.resid_in[(size_t)mc * a.N + tile * 16 + c];
    };
    auto wptr = [&](int tile, int b) -> const uint8_t* { return a.wq + (size_t)(tile * 16 + r) * Kb + (size_t)b * BB + g * 16; };
    u32x4 apk[NB][UB];
    {
        const unsigned char* xrow = reinterpret_cast<const unsigned char*>(a.xq) + (size_t)(r < a.M ? r : 0) * Kb + g * 16;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int u = 0; u < UB; u++) apk[b][u] = *reinterpret_cast<const u32x4*>(xrow + (size_t)b * BB + step_off<NW, UB>(wave, u));
    }
    const f16 xsh = a.xs[mc];
    __builtin_amdgcn_sched_barrier(0);
    int tile = blockIdx.x, par = 0;
    const int my_tiles = (a.ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x;
    u32x4 w[UB];
    Pre pre = {};
    {
        const uint8_t* wp0 = wptr(tile, 0);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
    }
    __builtin_amdgcn_sched_barrier(0);
    load_pre(pre, tile);
    __builtin_amdgcn_sched_barrier(0);
    const float xs_m = h2f(xsh);
    i32x4 acc = {0, 0, 0, 0};
    auto use = [&](const u32x4& wv, const u32x4& av) {
        u32 p0 = av[0], p1 = av[1], p2 = av[2], p3 = av[3];
        asm volatile("" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));   // (keeps hipcc from hoisting the loop-invariant widening)
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(widen16(p0, p1), widen16(wv[0], wv[1]), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(widen16(p2, p3), widen16(wv[2], wv[3]), acc, 0, 0, 0);
    };
    auto finish = [&](int tile, int par, const Pre& pre) {
        int* rb = red + par * NW * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) rb[wave * 256 + i * 64 + lane] = acc[i];
        acc = i32x4{0, 0, 0, 0};
        __syncthreads();
        if (!ethread) return;
        int sum = 0;
#pragma unroll
        for (int w2 = 0; w2 < NW; w2++) sum += rb[w2 * 256 + ridx];
        const f16 hv = f2h(((float)(sum >> 8) * xs_m) * h2f(pre.swn));   // both operands carried a factor 16
        if (EPI == SEPI_PLAIN) a.out[(size_t)m * a.N + tile * 16 + c] = hv;
        else a.resid_out[(size_t)m * a.N + tile * 16 + c] = f2h(h2f(pre.cf) + h2f(hv));
    };
    for (int q = 0; q < my_tiles - 1; q++) {
        const int nt = tile + (int)gridDim.x;
        Pre npre;
        load_pre(npre, nt);
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const uint8_t* wp = b + 1 < NB ? wptr(tile, b + 1) : wptr(nt, 0);
#pragma unroll
            for (int u = 0; u < UB; u++) {
                use(w[u], apk[b][u]);
                __builtin_amdgcn_sched_barrier(0);   // the refill right behind its consumer (see the kernel above)
                w[u] = wload<u32x4>(wp + step_off<NW, UB>(wave, u));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        finish(tile, par, pre);
        par ^= 1;
        pre = npre;
        tile = nt;
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {   // the last tile: nothing to refill behind its last batch
#pragma unroll
        for (int u = 0; u < UB; u++) {
            use(w[u], apk[b][u]);
            if (b + 1 < NB) {
                __builtin_amdgcn_sched_barrier(0);
                w[u] = wload<u32x4>(wptr(tile, b + 1) + step_off<NW, UB>(wave, u));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    finish(tile, par, pre);
}

static int stream_cap();
static bool longk_shape(int K) { return K == 28672; }   // 224 steps = 16 waves x 7 steps x 2 batches
template <int EPI>
static int launch_longk(const StreamArgs& a, hipStream_t st) {
    if (a.M < 1 || a.M > 16 || !longk_shape(a.K)) return -1;
    const int cap = stream_cap();
    int grid = a.ntiles;
    if (grid > cap) {
        const int per = (a.ntiles + cap - 1) / cap;
        grid = (a.ntiles + per - 1) / per;
    }
    hipLaunchKernelGGL((gemm_w4a4_longk_kernel<EPI, 16, 7, 2>), dim3(grid), dim3(16 * 64), (size_t)2 * 16 * 1024, st, a);
    return 0;
}

#ifdef QS_EXPERIMENTAL
#include "experimental/w4a4_sdma.inc"     // self-service LDS-DMA form of the draft GEMMs: QSPEC_SDMA=1..3
#endif

// ------------------------------------------------------------------ W4A16 (verify pass), same streaming skeleton
// out[m,n] = h( (sum_k f(x[m,k]) * w[n,k]) * f(sw[n]) ), fp32 accumulate  (bitblas.Matmul, quarot_nn/linear.py:102-124)
// over the SAME packed buffer.  A wave always owns the same K steps of every tile, so its activation fragments
// (M <= 16 rows x its K slice, 16 * UB registers per lane) are loaded ONCE, pre-shuffled into the nibble order of
// the dequantiser, and stay in registers for the whole kernel: no LDS tile, no per-step activation traffic (a
// 16-row weight tile alone would need 4x its own bytes in activations per step).
__device__ __forceinline__ f16x8 sdequant_s4x8(u32 p) { return dequant_s4x8_bitop(p); }   // common.cuh (k order 0,4,1,5,2,6,3,7)
__device__ __forceinline__ f16x8 sshuffle_act8(u32x4 a) {  // 8 consecutive fp16 -> order 0,4,1,5,2,6,3,7
    u32x4 o;
    o[0] = __builtin_amdgcn_perm(a[2], a[0], 0x05040100u);
    o[1] = __builtin_amdgcn_perm(a[2], a[0], 0x07060302u);
    o[2] = __builtin_amdgcn_perm(a[3], a[1], 0x05040100u);
    o[3] = __builtin_amdgcn_perm(a[3], a[1], 0x07060302u);
    return __builtin_bit_cast(f16x8, o);
}

// XP: the activations arrive in FRAGMENT-MAJOR layout (a.x = the 16-row tile as [K / 128 steps][4 dwords][4 k-groups][16 rows][8
// halves, dequantiser order]: w4a16_xperm_offset) -- what this kernel's MFMA fragments are -- so a wave takes its fragments
// straight from memory with fully coalesced 1-KiB loads: no LDS staging pass, no barrier in front of the first MFMA.
template <int EPI, int NW, int UB, bool XP = false>
__global__ __launch_bounds__(NW * 64) void gemm_w4a16_stream_kernel(StreamArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int Kb = a.K >> 1;
    const size_t ldw = a.ldw ? (size_t)a.ldw : (size_t)Kb, ldx = a.ldx ? (size_t)a.ldx : (size_t)a.K;
    // SEPI_PARTIAL: blockIdx.y picks one of gridDim.y K slices of length K (x row stride ldx, weight row stride ldw);
    // the raw fp32 sums go to part[blockIdx.y] and are combined in slice order by whoever consumes them
    const size_t kofs = EPI == SEPI_PARTIAL ? (size_t)blockIdx.y * a.K : 0;
    float* red = reinterpret_cast<float*>(smem);                     // [2][NW][256]
    f16* ex = reinterpret_cast<f16*>(red + 2 * NW * 256);            // [2][256]
    // epilogue thread t owns output (token m = t / 16, tile column c = t % 16): for M <= 4 that is wave 0 alone, the
    // other waves go straight back to streaming.  ridx = where the MFMA left that element in a wave's accumulators.
    const int c = tid & 15, m = tid >> 4;
    const bool ethread = m < a.M;
    const int ridx = (m & 3) * 64 + ((m >> 2) & 3) * 16 + c;
    const int mc = m < a.M ? m : 0;
    int64_t pos_m = 0, slot_m = -1;
    if (EPI == SEPI_QKV) {
        pos_m = a.positions[mc];
        slot_m = a.slot_mapping[mc];
    }
    struct Pre {
        f16 swn, cf, sf;
    };
    auto load_pre = [&](Pre& pre, int tile) {
        pre.swn = a.ws[stile_row<EPI>(tile, c, a.I, a.hdl)];
        if (EPI == SEPI_QKV) {
            const int o = qkv_pair(tile, c & 7, a.I, a.hdl).i;
            const f16* cs = a.cos_sin_cache + (pos_m << a.hdl);   // rot_dim = head size
            pre.cf = cs[o];
            pre.sf = cs[(1 << (a.hdl - 1)) + o];
        }
    };
    auto wptr = [&](int tile) -> const uint8_t* {
        return a.wq + (size_t)stile_row<EPI>(tile, r, a.I, a.hdl) * ldw + (kofs >> 1) + g * 16;
    };
    // Activation fragments of this wave's K slice (rows >= M repeat row 0: their outputs are never stored).
    // As MFMA fragments they are 16 rows x 64-byte pieces per load instruction (measured: the 128 KB per workgroup cost
    // ~3 us that way), so the workgroup copies the [16 x K] tile through LDS instead: fully coalesced global loads
    // (1 KiB of one row per wave instruction), 16-byte chunks XOR-swizzled by the row so the ds_read_b128 fragment
    // reads are conflict-free, one stage = the K range of a pair of steps (NW * 256 k), double buffered.
    constexpr int UE = UB & ~1, NST = (UB + 1) / 2;            // paired steps / stages
    constexpr int CPT = 8;                                      // 16-byte chunks per thread and (full) stage
    unsigned char* abuf = smem + (size_t)2 * NW * 1024 + 1024;  // [2][16 rows][NW * 512 B]
    constexpr int SBMAX = NW * 512;
    u32x4 areg[2][CPT];
    auto stage_load = [&](u32x4(&dst)[CPT], int st_) {
        const bool paired = 2 * st_ < UE;
        const int cpr = paired ? NW * 32 : NW * 16;             // chunks per row in this stage
        const size_t kbase = paired ? (size_t)st_ * NW * 256 : (size_t)UE * NW * 128;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int cidx = tid + i * NW * 64;
            const int row = paired ? cidx / (NW * 32) : (cidx / (NW * 16)) & 15;   // unpaired: CPT covers the rows twice
            const int q = cidx % cpr;
            dst[i] = *reinterpret_cast<const u32x4*>(a.x + (size_t)(row < a.M ? row : 0) * ldx + kofs + kbase + (size_t)q * 8);
        }
    };
    auto stage_store = [&](const u32x4(&src)[CPT], int st_) {
        const bool paired = 2 * st_ < UE;
        const int cpr = paired ? NW * 32 : NW * 16;
        unsigned char* buf = abuf + (size_t)(st_ & 1) * 16 * SBMAX;
#pragma unroll
        for (int i = 0; i < CPT; i++) {
            const int cidx = tid + i * NW * 64;
            const int row = paired ? cidx / (NW * 32) : (cidx / (NW * 16)) & 15;
            const int q = cidx % cpr;
            *reinterpret_cast<u32x4*>(buf + (size_t)row * SBMAX + ((q ^ row) << 4)) = src[i];
        }
    };
    f16x8 af[UB][4];
    if constexpr (XP) {
        const f16* xp = a.x + (kofs >> 7) * 2048 + (size_t)lane * 8;     // (step, dword) blocks of 64 lanes x 8 halves
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int kstep = step_off<NW, UB>(wave, u) >> 6;            // this wave's step u = K step kstep of the row
#pragma unroll
            for (int dd = 0; dd < 4; dd++) af[u][dd] = *reinterpret_cast<const f16x8*>(xp + ((size_t)kstep * 4 + dd) * 512);
        }
    } else {
        stage_load(areg[0], 0);
        if (NST > 1) stage_load(areg[1], 1);
    }
    int tile = a.tile0 + blockIdx.x, par = 0;
    const int tile_end = a.tile0 + a.ntiles;
    const int my_tiles = (tile_end - tile + (int)gridDim.x - 1) / (int)gridDim.x;
    u32x4 w[UB];
    Pre pre = {};
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint8_t* wp0 = wptr(tile);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
        load_pre(pre, tile);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!XP)
#pragma unroll
    for (int st_ = 0; st_ < NST; st_++) {
        stage_store(areg[st_ & 1], st_);
        __syncthreads();
        if (st_ + 2 < NST) stage_load(areg[st_ & 1], st_ + 2);   // compile-time condition (unrolled)
        const bool paired = 2 * st_ < UE;
        const unsigned char* buf = abuf + (size_t)(st_ & 1) * 16 * SBMAX + (size_t)r * SBMAX;
#pragma unroll
        for (int j = 0; j < 2; j++) {
            const int u = 2 * st_ + j;
            if (u < UB) {
#pragma unroll
                for (int dd = 0; dd < 4; dd++) {
                    const int qq = paired ? wave * 32 + j * 16 + g * 4 + dd : wave * 16 + g * 4 + dd;
                    af[u][dd] = sshuffle_act8(*reinterpret_cast<const u32x4*>(buf + ((qq ^ r) << 4)));
                }
            }
        }
    }

    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    auto use = [&](const u32x4& wv, int u) {
#pragma unroll
        for (int dd = 0; dd < 4; dd++)
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u][dd], sdequant_s4x8(wv[dd]), acc, 0, 0, 0);
    };
    auto finish = [&](int tile, int par, const Pre& pre) {
        float* rb = red + par * NW * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) rb[wave * 256 + i * 64 + lane] = acc[i];
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        f16 hv = (f16)0.0f;
        if (ethread) {
            float sum = rb[ridx];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) sum = sum + rb[w2 * 256 + ridx];   // wave order: deterministic
            hv = f2h(sum * h2f(pre.swn));
            if (EPI == SEPI_PARTIAL) a.part[((size_t)blockIdx.y * a.M + m) * a.N + tile * 16 + c] = sum;
        }
        if (EPI == SEPI_PARTIAL) return;
        if (EPI == SEPI_PLAIN) {
            if (ethread) a.out[(size_t)m * a.N + tile * 16 + c] = hv;
            return;
        }
        f16* e = ex + par * 256;
        if (tid < 256) e[tid] = hv;
        __syncthreads();
        if (!ethread) return;
        const f16 partner = e[tid ^ 8];
        if (EPI == SEPI_GATEUP) {
            if (c < 8) {
                const float gt = h2f(partner);
                const float act = h2f(f2h(gt / (1.0f + qexpf(-gt))));
                a.out[(size_t)m * a.I + tile * 8 + c] = f2h(act * h2f(hv));
            }
            return;
        }
        const QkvPair qp = qkv_pair(tile, c & 7, a.I, a.hdl);
        if (!qp.valid) return;   // a half tile's repeated slots
        const int head = qp.head, o = qp.i;
        const int n = (head << a.hdl) + ((c >> 3) << (a.hdl - 1)) + o;
        f16 res = hv;
        if (head < a.nq + a.nkv) {
            const float cff = h2f(pre.cf), sff = h2f(pre.sf);
            const float xf = h2f(c < 8 ? hv : partner), yf = h2f(c < 8 ? partner : hv);
            res = c < 8 ? f2h(h2f(f2h(xf * cff)) - h2f(f2h(yf * sff))) : f2h(h2f(f2h(yf * cff)) + h2f(f2h(xf * sff)));
        }
        a.out[(size_t)m * a.N + n] = res;
        if (head >= a.nq && slot_m >= 0) {
            const bool is_k = head < a.nq + a.nkv;
            const int kvh = is_k ? head - a.nq : head - a.nq - a.nkv;
            f16* cache = is_k ? a.key_cache : a.value_cache;
            cache[((slot_m * a.nkv + kvh) << a.hdl) + ((c >> 3) << (a.hdl - 1)) + o] = res;
        }
    };
    for (int q = 0; q < my_tiles - 1; q++) {
        const int nt = tile + gridDim.x;
        const uint8_t* wp = wptr(nt);
        Pre npre;
        load_pre(npre, nt);
#pragma unroll
        for (int u = 0; u < UB; u++) {
            use(w[u], u);
            __builtin_amdgcn_sched_barrier(0);
            w[u] = wload<u32x4>(wp + step_off<NW, UB>(wave, u));
            __builtin_amdgcn_sched_barrier(0);
        }
        finish(tile, par, pre);
        par ^= 1;
        pre = npre;
        tile = nt;
    }
#pragma unroll
    for (int u = 0; u < UB; u++) use(w[u], u);
    finish(tile, par, pre);
}

#ifdef QS_EXPERIMENTAL
#include "experimental/prefetch.inc"      // weight prefetch launches: qspec_prefetch / qspec_prefetch_tiles
#endif

// ------------------------------------------------------------------ fp16 x fp16^T (lm_head), same skeleton
// out[m,n] = h( sum_k f(x[m,k]) f(w[n,k]) ), fp32 accumulate  (nn.Linear lm_head, logits_processor.py:92-97).
// K = 32 * NW * UB: a wave keeps UB loads (UB KiB) of the 1 GB vocabulary matrix in flight and its activation
// fragments in registers; a workgroup walks over its tiles with the refill-behind-use pipeline of the kernels above.
// part_max (may be NULL; the sampler front end, sampler.hip): every workgroup also leaves, per token row, the largest
// fp16 logit among ITS tiles and the first column holding it -- part_max[m * gridDim.x + blockIdx.x] = {value, column}
// -- so that the softmax that follows does not have to read the logits once more just to find the row maximum.
struct HeadMax {
    float v;
    int i;
};
template <int NW, int UB>
__global__ __launch_bounds__(NW * 64) void gemm_f16_stream_kernel(const f16* __restrict__ x, const f16* __restrict__ wt,
                                                                  f16* __restrict__ out, int M, int N, int K,
                                                                  int ntiles, HeadMax* __restrict__ part_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [2][NW][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int c = tid & 15, m = tid >> 4;   // epilogue thread -> (token, tile column), as in the kernels above
    const bool ethread = m < M;
    const int ridx = (m & 3) * 64 + ((m >> 2) & 3) * 16 + c;
    // step u of this wave = bytes [step_off(wave, u) + 16 g, +16) of a weight row = k (halves) offset / 2
    f16x8 af[UB];
    {
        const f16* xrow = x + (size_t)(r < M ? r : 0) * K + g * 8;
#pragma unroll
        for (int u = 0; u < UB; u++) af[u] = *reinterpret_cast<const f16x8*>(xrow + step_off<NW, UB>(wave, u) / 2);
    }
    int tile = blockIdx.x, par = 0;
    const int my_tiles = (ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x;
    auto wptr = [&](int t) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(wt + (size_t)(t * 16 + r) * K) + g * 16;
    };
    f16x8 w[UB];
    __builtin_amdgcn_sched_barrier(0);
    {
        const unsigned char* wp0 = wptr(tile);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<f16x8>(wp0 + step_off<NW, UB>(wave, u));
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;
    auto finish = [&](int t, int par) {
        float* rb = red + par * NW * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) rb[wave * 256 + i * 64 + lane] = acc[i];
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (ethread) {
            float sum = rb[ridx];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) sum = sum + rb[w2 * 256 + ridx];
            const f16 hv = f2h(sum);
            out[(size_t)m * N + t * 16 + c] = hv;
            const float fv = h2f(hv);
            if (fv > best_v) {          // tiles come in increasing column order: a tie keeps the first column
                best_v = fv;
                best_i = t * 16 + c;
            }
        }
    };
    for (int q = 0; q < my_tiles - 1; q++) {
        const int nt = tile + gridDim.x;
        const unsigned char* wp = wptr(nt);
#pragma unroll
        for (int u = 0; u < UB; u++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], w[u], acc, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            w[u] = wload<f16x8>(wp + step_off<NW, UB>(wave, u));
            __builtin_amdgcn_sched_barrier(0);
        }
        finish(tile, par);
        par ^= 1;
        tile = nt;
    }
#pragma unroll
    for (int u = 0; u < UB; u++) acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[u], w[u], acc, 0, 0, 0);
    finish(tile, par);
    if (part_max) {   // the 16 columns of a row sit in 16 consecutive lanes: larger value, then smaller column, wins
#pragma unroll
        for (int mk = 1; mk < 16; mk <<= 1) {
            const float ov = __shfl_xor(best_v, mk, 64);
            const int oi = __shfl_xor(best_i, mk, 64);
            if (ov > best_v || (ov == best_v && oi < best_i)) {
                best_v = ov;
                best_i = oi;
            }
        }
        if (ethread && c == 0) part_max[(size_t)m * gridDim.x + blockIdx.x] = HeadMax{best_v, best_i};
    }
}

// ---- long rows (K = 8192, Llama-3-70B's lm_head: 2.1 GB): NB batches of NW x UB steps per tile, as gemm_w4a4_longk_kernel
// does for the int4 GEMM -- 16 waves x 16 steps of fp16 fragments do not fit the 128-VGPR budget of 1024 threads, two batches
// of 8 do (64 VGPRs of activation fragments + 32 of weights).  Each weight register is refilled with the same step of the next
// batch (the next tile's first batch behind the last one) right behind its consumer.  Same sums per output in the same wave
// order as gemm_f16_kernel's, fp32; same HeadMax partials as gemm_f16_stream_kernel.
template <int NW, int UB, int NB>
__global__ __launch_bounds__(NW * 64) void gemm_f16_longk_kernel(const f16* __restrict__ x, const f16* __restrict__ wt,
                                                                 f16* __restrict__ out, int M, int N, int K, int ntiles,
                                                                 HeadMax* __restrict__ part_max) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);   // [2][NW][256]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    const int c = tid & 15, m = tid >> 4;
    const bool ethread = m < M;
    const int ridx = (m & 3) * 64 + ((m >> 2) & 3) * 16 + c;
    constexpr int BB = UB * NW * 64;               // bytes of a weight row per batch
    f16x8 af[NB][UB];
    {
        const unsigned char* xrow = reinterpret_cast<const unsigned char*>(x + (size_t)(r < M ? r : 0) * K) + g * 16;
#pragma unroll
        for (int b = 0; b < NB; b++)
#pragma unroll
            for (int u = 0; u < UB; u++) af[b][u] = *reinterpret_cast<const f16x8*>(xrow + (size_t)b * BB + step_off<NW, UB>(wave, u));
    }
    int tile = blockIdx.x, par = 0;
    const int my_tiles = (ntiles - tile + (int)gridDim.x - 1) / (int)gridDim.x;
    auto wptr = [&](int t, int b) -> const unsigned char* {
        return reinterpret_cast<const unsigned char*>(wt + (size_t)(t * 16 + r) * K) + (size_t)b * BB + g * 16;
    };
    f16x8 w[UB];
    __builtin_amdgcn_sched_barrier(0);
    {
        const unsigned char* wp0 = wptr(tile, 0);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<f16x8>(wp0 + step_off<NW, UB>(wave, u));
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;
    auto finish = [&](int t, int par) {
        float* rb = red + par * NW * 256;
#pragma unroll
        for (int i = 0; i < 4; i++) rb[wave * 256 + i * 64 + lane] = acc[i];
        acc = f32x4{0.f, 0.f, 0.f, 0.f};
        __syncthreads();
        if (ethread) {
            float sum = rb[ridx];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) sum = sum + rb[w2 * 256 + ridx];
            const f16 hv = f2h(sum);
            out[(size_t)m * N + t * 16 + c] = hv;
            const float fv = h2f(hv);
            if (fv > best_v) {          // tiles come in increasing column order: a tie keeps the first column
                best_v = fv;
                best_i = t * 16 + c;
            }
        }
    };
    for (int q = 0; q < my_tiles - 1; q++) {
        const int nt = tile + gridDim.x;
#pragma unroll
        for (int b = 0; b < NB; b++) {
            const unsigned char* wp = b + 1 < NB ? wptr(tile, b + 1) : wptr(nt, 0);
#pragma unroll
            for (int u = 0; u < UB; u++) {
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[b][u], w[u], acc, 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                w[u] = wload<f16x8>(wp + step_off<NW, UB>(wave, u));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        finish(tile, par);
        par ^= 1;
        tile = nt;
    }
#pragma unroll
    for (int b = 0; b < NB; b++) {
#pragma unroll
        for (int u = 0; u < UB; u++) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[b][u], w[u], acc, 0, 0, 0);
            if (b + 1 < NB) {
                __builtin_amdgcn_sched_barrier(0);
                w[u] = wload<f16x8>(wptr(tile, b + 1) + step_off<NW, UB>(wave, u));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    finish(tile, par);
    if (part_max) {   // the 16 columns of a row sit in 16 consecutive lanes: larger value, then smaller column, wins
#pragma unroll
        for (int mk = 1; mk < 16; mk <<= 1) {
            const float ov = __shfl_xor(best_v, mk, 64);
            const int oi = __shfl_xor(best_i, mk, 64);
            if (ov > best_v || (ov == best_v && oi < best_i)) {
                best_v = ov;
                best_i = oi;
            }
        }
        if (ethread && c == 0) part_max[(size_t)m * gridDim.x + blockIdx.x] = HeadMax{best_v, best_i};
    }
}

// ---- the same lm_head with its 1 GB weight stream through self-service LDS-DMA (K = 4096, M <= 16).
// The draft GEMMs did not gain from LDS-DMA (what is above their floor is the launch's fixed
// sequence, not stream time); THIS launch is 99 % stream: 31 tiles of 128 KB per workgroup, where the register path's
// pure read reaches 6.2 TB/s and four or more issuing waves per CU with `global_load_lds_dwordx4 ... nt` 6.9 TB/s
// (scripts/micro/ldsdma.hip: 168 -> 151 us for these bytes).  Stream wave w owns the k range [512 w, 512 w + 512) of
// every weight row (1 KiB per row) and walks it in GROUPS of four 64-byte steps: one group = 16 rows x 256 B = four loads
// (4 rows x 256 B each, lane l -> row 4 q + l / 16, 16-byte piece (l % 16) ^ row: the LDS image stays lane-linear, the
// fragment reads are conflict-free).  A wave keeps a rolling window of four groups (16 KiB of its own LDS, 16 loads) in
// flight -- 128 KB per CU -- and learns that a group has landed from its own `s_waitcnt vmcnt(12)`: no loader waves, no
// flags.  The epilogue (sum of the eight waves' fp32 partials in wave order, one rounding, store, running row maximum)
// belongs to four waves that stream nothing; they meet the stream waves at one barrier per tile.
template <int MT /* 16-token tiles: M <= 16 MT */>
__global__ __launch_bounds__(12 * 64) void gemm_f16_sdma_kernel(const f16* __restrict__ x, const f16* __restrict__ wt,
                                                                f16* __restrict__ out, int M, int N, int ntiles,
                                                                HeadMax* __restrict__ part_max) {
    constexpr int NW = 8, K = 4096, WIN = 4;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                 // [2][NW][MT][256]
    unsigned char* ring = smem + (size_t)2 * NW * MT * 1024;     // [NW][WIN][4 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;

    if (wave >= NW) {   // ---------------------------------------------------------------- epilogue waves
        const int et = tid - NW * 64;                            // 0..255 -> (token m (+ 16 mt), tile column c)
        const int c = et & 15, m0 = et >> 4;
        const int ridx = (m0 & 3) * 64 + ((m0 >> 2) & 3) * 16 + c;
        float best_v[MT];
        int best_i[MT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            best_v[mt] = -__builtin_inff();
            best_i[mt] = 0x7fffffff;
        }
        int tile = blockIdx.x, par = 0;
        for (int ti = 0; ti < my_tiles; ti++) {
            __syncthreads();                                     // A(ti): the stream waves have posted the tile's partials
            const float* rb = red + par * NW * MT * 256;
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const int m = m0 + 16 * mt;
                if (m < M) {
                    float sum = rb[mt * 256 + ridx];
#pragma unroll
                    for (int w2 = 1; w2 < NW; w2++) sum = sum + rb[(w2 * MT + mt) * 256 + ridx];
                    const f16 hv = f2h(sum);
                    out[(size_t)m * N + tile * 16 + c] = hv;
                    const float fv = h2f(hv);
                    if (fv > best_v[mt]) {          // tiles come in increasing column order: a tie keeps the first column
                        best_v[mt] = fv;
                        best_i[mt] = tile * 16 + c;
                    }
                }
            }
            par ^= 1;
            tile += gridDim.x;
        }
        if (part_max) {   // the 16 columns of a row sit in 16 consecutive lanes: larger value, then smaller column, wins
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                float bv = best_v[mt];
                int bi = best_i[mt];
#pragma unroll
                for (int mk = 1; mk < 16; mk <<= 1) {
                    const float ov = __shfl_xor(bv, mk, 64);
                    const int oi = __shfl_xor(bi, mk, 64);
                    if (ov > bv || (ov == bv && oi < bi)) {
                        bv = ov;
                        bi = oi;
                    }
                }
                const int m = m0 + 16 * mt;
                if (m < M && c == 0) part_max[(size_t)m * gridDim.x + blockIdx.x] = HeadMax{bv, bi};
            }
        }
        return;
    }

    // ---------------------------------------------------------------- stream waves
    const int r = lane & 15, g = lane >> 4;
    // activation fragments of this wave's k range: step s = halves [512 w + 32 s + 8 g, + 8) of row r + 16 mt (rows >= M: row 0)
    f16x8 af[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int row = r + 16 * mt;
        const f16* xrow = x + (size_t)(row < M ? row : 0) * K + wave * 512 + g * 8;
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) af[mt][s2] = *reinterpret_cast<const f16x8*>(xrow + s2 * 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // before the first LDS-DMA load: hipcc's own waits do not know them
    const u32 my0 = (u32)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)ring + (u32)wave * (WIN * 4096);
    // source of this lane inside a group: row 4 q + (lane >> 4) of the tile, 16-byte piece ((lane & 15) ^ row) of the group's
    // 256-byte segment of the wave's KiB
    const int lrow = lane >> 4;
    auto issue_group = [&](int gi) {   // group gi = (tile gi / 4, segment gi % 4) of this workgroup's sequence
        const int t = blockIdx.x + (gi >> 2) * (int)gridDim.x;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(wt + (size_t)t * 16 * K) + wave * 1024 + (gi & 3) * 256;
        const u32 dst = my0 + (u32)(gi & (WIN - 1)) * 4096u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = 4 * q + lrow;
            glds16<1>(base + (size_t)row * (K * 2) + (((lane & 15) ^ row) << 4), dst + (u32)q * 1024u);
        }
    };
    const int n_groups = my_tiles * 4;
#pragma unroll
    for (int gi = 0; gi < WIN; gi++)
        if (gi < n_groups) issue_group(gi);
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
    const __attribute__((address_space(3))) unsigned char* my3 =
        (const __attribute__((address_space(3))) unsigned char*)ring + (size_t)wave * (WIN * 4096);
    // fragment of step q2 (0..3) of a group, lane (r, g): load r / 4, lane (r % 4) * 16 + ((4 q2 + g) ^ r)
    u32 foff[4];
#pragma unroll
    for (int q2 = 0; q2 < 4; q2++) foff[q2] = (u32)(r >> 2) * 1024u + (u32)((((r & 3) << 4) | ((q2 * 4 + g) ^ r)) << 4);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int par = 0, gi = 0;
    for (int ti = 0; ti < my_tiles; ti++) {
        const bool last = ti == my_tiles - 1;   // (wave-uniform) the window runs empty behind the last tile
#pragma unroll
        for (int seg = 0; seg < 4; seg++, gi++) {   // unrolled: the fragment registers are indexed statically
            if (!last) vmcnt_le<12>();              // three groups issued behind this one
            else if (seg == 0) vmcnt_le<12>();
            else if (seg == 1) vmcnt_le<8>();
            else if (seg == 2) vmcnt_le<4>();
            else vmcnt_le<0>();
            u32x4 wv[4];
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++)
                wv[q2] = *reinterpret_cast<const volatile lds_u32x4_t*>(my3 + (size_t)seg * 4096 + foff[q2]);   // slot = gi % WIN = seg
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (gi + WIN < n_groups) issue_group(gi + WIN);   // into the slot just read
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][seg * 4 + q2], __builtin_bit_cast(f16x8, wv[q2]), acc[mt], 0, 0, 0);
        }
        // tile complete: post the partial sums, meet the epilogue waves
        float* rb = red + par * NW * MT * 256;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int i = 0; i < 4; i++) rb[(wave * MT + mt) * 256 + i * 64 + lane] = acc[mt][i];
            acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // A (raw: no vmcnt drain)
        par ^= 1;
    }
}
// ---- 17..32 tokens (config 3's draft steps): two token tiles over ONE pass of the weight stream.
// The 12-wave form above has no room for a second tile's 64 fragment registers (three waves per SIMD: 170 VGPRs; it was built
// and spilled: 240 us against the M-tiled kernel's 230).  Here ALL eight waves stream -- two per SIMD, 256 VGPRs, 128 of them
// activation fragments -- with the same ring (four 4-KiB groups per wave in flight, 128 KB per CU), and the epilogue is theirs
// too: 8 x 64 lanes = the tile's 32 x 16 outputs, one per lane, summed over the waves' partials in wave order.  The partials
// are single-buffered (16 KB: ring + partials = 144 KB): barrier B in front of the post, barrier A behind it; the LDS-DMA
// window keeps filling across both (raw s_barrier, no vmcnt drain).
__global__ __launch_bounds__(8 * 64) void gemm_f16_sdma2_kernel(const f16* __restrict__ x, const f16* __restrict__ wt,
                                                                f16* __restrict__ out, int M, int N, int ntiles,
                                                                HeadMax* __restrict__ part_max) {
    constexpr int NW = 8, K = 4096, WIN = 4, MT = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* red = reinterpret_cast<float*>(smem);                 // [NW][MT][256]
    unsigned char* ring = smem + (size_t)NW * MT * 1024;         // [NW][WIN][4 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int my_tiles = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    // epilogue role of this lane: output (token em, tile column ec); its partial sits at index eidx of tile emt of every wave
    const int ec = tid & 15, em = tid >> 4, emt = em >> 4;
    const int eidx = (em & 3) * 64 + ((em >> 2) & 3) * 16 + ec;
    float best_v = -__builtin_inff();
    int best_i = 0x7fffffff;

    const int r = lane & 15, g = lane >> 4;
    f16x8 af[MT][16];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int row = r + 16 * mt;
        const f16* xrow = x + (size_t)(row < M ? row : 0) * K + wave * 512 + g * 8;
#pragma unroll
        for (int s2 = 0; s2 < 16; s2++) af[mt][s2] = *reinterpret_cast<const f16x8*>(xrow + s2 * 32);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // before the first LDS-DMA load: hipcc's own waits do not know them
    const u32 my0 = (u32)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)ring + (u32)wave * (WIN * 4096);
    const int lrow = lane >> 4;
    auto issue_group = [&](int gi) {   // as in gemm_f16_sdma_kernel
        const int t = blockIdx.x + (gi >> 2) * (int)gridDim.x;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(wt + (size_t)t * 16 * K) + wave * 1024 + (gi & 3) * 256;
        const u32 dst = my0 + (u32)(gi & (WIN - 1)) * 4096u;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = 4 * q + lrow;
            glds16<1>(base + (size_t)row * (K * 2) + (((lane & 15) ^ row) << 4), dst + (u32)q * 1024u);
        }
    };
    const int n_groups = my_tiles * 4;
#pragma unroll
    for (int gi = 0; gi < WIN; gi++)
        if (gi < n_groups) issue_group(gi);
    typedef __attribute__((address_space(3))) u32x4 lds_u32x4_t;
    const __attribute__((address_space(3))) unsigned char* my3 =
        (const __attribute__((address_space(3))) unsigned char*)ring + (size_t)wave * (WIN * 4096);
    u32 foff[4];
#pragma unroll
    for (int q2 = 0; q2 < 4; q2++) foff[q2] = (u32)(r >> 2) * 1024u + (u32)((((r & 3) << 4) | ((q2 * 4 + g) ^ r)) << 4);
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    int gi = 0, tile = blockIdx.x;
    for (int ti = 0; ti < my_tiles; ti++) {
        const bool last = ti == my_tiles - 1;   // (wave-uniform) the window runs empty behind the last tile
#pragma unroll
        for (int seg = 0; seg < 4; seg++, gi++) {
            if (!last) vmcnt_le<12>();
            else if (seg == 0) vmcnt_le<12>();
            else if (seg == 1) vmcnt_le<8>();
            else if (seg == 2) vmcnt_le<4>();
            else vmcnt_le<0>();
            u32x4 wv[4];
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++)
                wv[q2] = *reinterpret_cast<const volatile lds_u32x4_t*>(my3 + (size_t)seg * 4096 + foff[q2]);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (gi + WIN < n_groups) issue_group(gi + WIN);   // into the slot just read
#pragma unroll
            for (int q2 = 0; q2 < 4; q2++)
#pragma unroll
                for (int mt = 0; mt < MT; mt++)
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[mt][seg * 4 + q2], __builtin_bit_cast(f16x8, wv[q2]), acc[mt], 0, 0, 0);
        }
        __builtin_amdgcn_s_barrier();   // B: every lane has read the previous tile's partials
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
#pragma unroll
            for (int i = 0; i < 4; i++) red[(wave * MT + mt) * 256 + i * 64 + lane] = acc[mt][i];
            acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();   // A: the tile's partials are posted
        if (em < M) {
            float sum = red[emt * 256 + eidx];
#pragma unroll
            for (int w2 = 1; w2 < NW; w2++) sum = sum + red[(w2 * MT + emt) * 256 + eidx];
            const f16 hv = f2h(sum);
            out[(size_t)em * N + tile * 16 + ec] = hv;
            const float fv = h2f(hv);
            if (fv > best_v) {          // tiles come in increasing column order: a tie keeps the first column
                best_v = fv;
                best_i = tile * 16 + ec;
            }
        }
        tile += gridDim.x;
    }
    if (part_max) {   // the 16 columns of a row sit in 16 consecutive lanes: larger value, then smaller column, wins
#pragma unroll
        for (int mk = 1; mk < 16; mk <<= 1) {
            const float ov = __shfl_xor(best_v, mk, 64);
            const int oi = __shfl_xor(best_i, mk, 64);
            if (ov > best_v || (ov == best_v && oi < best_i)) {
                best_v = ov;
                best_i = oi;
            }
        }
        if (em < M && ec == 0) part_max[(size_t)em * gridDim.x + blockIdx.x] = HeadMax{best_v, best_i};
    }
}
// (dev knob QSPEC_HEAD_SDMA=0: the register-streaming lm_head)
static bool head_sdma_on() {
    static const int v = QS_DEV_KNOB("QSPEC_HEAD_SDMA", 1);
    return v != 0;
}

template <int MT>
static int gemm_f16_sdma_launch_inst(const f16* x, const f16* w, f16* out, int M, int N, int ntiles, int grid, HeadMax* pm,
                                     hipStream_t st) {
    const size_t lds = (size_t)2 * 8 * MT * 1024 + (size_t)8 * 4 * 4096;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_sdma_kernel<MT>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)lds) != hipSuccess)
            return -8;
        attr_set = true;
    }
    hipLaunchKernelGGL(gemm_f16_sdma_kernel<MT>, dim3(grid), dim3(12 * 64), lds, st, x, w, out, M, N, ntiles, pm);
    return 0;
}
// (MT = 2 of THIS kernel -- 17..32 tokens as two token tiles over one weight pass -- was built and measured in round 3: 240 us
// against 245 for the M-tiled kernel at M = 32: the second tile's 64 fragment registers push the kernel to 168 VGPRs + scratch
// at 12 waves, and scratch traffic shares the vmcnt queue with the LDS-DMA window.  gemm_f16_sdma2_kernel -- eight waves, all
// streaming, 184 VGPRs -- is what 17..32 tokens launch: 183 us.)
static int gemm_f16_sdma_launch(const f16* x, const f16* w, f16* out, int M, int N, int ntiles, int grid, HeadMax* pm,
                                hipStream_t st) {
    if (M > 16) {   // two token tiles: gemm_f16_sdma2_kernel
        const size_t lds = (size_t)8 * 2 * 1024 + (size_t)8 * 4 * 4096;
        static bool attr_set = false;
        if (!attr_set) {
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_f16_sdma2_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)lds) != hipSuccess)
                return -8;
            attr_set = true;
        }
        hipLaunchKernelGGL(gemm_f16_sdma2_kernel, dim3(grid), dim3(8 * 64), lds, st, x, w, out, M, N, ntiles, pm);
        return 0;
    }
    return gemm_f16_sdma_launch_inst<1>(x, w, out, M, N, ntiles, grid, pm, st);
}
int gemm_f16_stream_grid(int N);
// 17..32 tokens: the two-token-tile LDS-DMA form, for a long stream only (the lm_head: >= 4 tiles per workgroup)
static bool gemm_f16_stream2_supported(int M, int N, int K) {
    return M > 16 && M <= 32 && K == 4096 && N % 16 == 0 && N / 16 >= 4 * gemm_f16_stream_grid(N) && head_sdma_on();
}
bool gemm_f16_stream_supported(int M, int N, int K) {
    if (gemm_f16_stream2_supported(M, N, K)) return true;
    return M >= 1 && M <= 16 && N % 16 == 0 && (K == 1024 || K == 2048 || K == 4096 || K == 5120 || K == 8192);
}
int gemm_f16_stream_grid(int N) {
    const int ntiles = N / 16;
    int grid = ntiles;
    if (grid > 256) {
        const int per = (ntiles + 255) / 256;
        grid = (ntiles + per - 1) / per;
    }
    return grid;
}
// part_max != NULL: the grid is gemm_f16_stream_grid(N) (<= 256) and part_max holds [M][grid] {float, int} pairs.
int gemm_f16_stream(const f16* x, const f16* w, f16* out, int M, int N, int K, void* part_max, hipStream_t st) {
    if (!gemm_f16_stream_supported(M, N, K)) return -1;
    const int ntiles = N / 16;
    static const int cap_knob = QS_DEV_KNOB("QSPEC_HEAD_CAP", 256);   // one workgroup per CU: 6.3 TB/s measured (512: 6.1, 1024: 5.9)
    const int cap = cap_knob < 1 ? 256 : cap_knob;
    int grid = ntiles;
    if (part_max || M > 16) {
        grid = gemm_f16_stream_grid(N);
    } else if (grid > cap) {
        const int per = (ntiles + cap - 1) / cap;
        grid = (ntiles + per - 1) / per;
    }
    HeadMax* pm = reinterpret_cast<HeadMax*>(part_max);
    if (K == 4096 && ntiles >= 4 * grid && head_sdma_on()) {   // a long stream: LDS-DMA (see gemm_f16_sdma_kernel)
        return gemm_f16_sdma_launch(x, w, out, M, N, ntiles, grid, pm, st);
    }
    if (K == 8192) {   // 256 steps of 64 bytes per row = 16 waves x 4 steps x 4 batches (8 x 2: 20 VGPRs spilled at 1024 threads)
        hipLaunchKernelGGL((gemm_f16_longk_kernel<16, 4, 4>), dim3(grid), dim3(16 * 64), (size_t)2 * 16 * 1024, st, x, w, out, M, N,
                           K, ntiles, pm);
        return 0;
    }
#define QS_F16S(NWV, UBV) hipLaunchKernelGGL((gemm_f16_stream_kernel<NWV, UBV>), dim3(grid), dim3(NWV * 64), (size_t)2 * NWV * 1024, st, x, w, out, M, N, K, ntiles, pm)
    if (K == 4096) QS_F16S(8, 16);
    else if (K == 2048) QS_F16S(8, 8);
    else if (K == 1024) QS_F16S(4, 8);
    else QS_F16S(8, 20);
#undef QS_F16S
    return 0;
}

// Shape classes: K/2 bytes of a weight row = 64 * NW * UB * NB exactly.
struct StreamShape {
    int NW, UB, NI;
};
static bool stream_shape(int K, StreamShape* sh) {
    const int nsteps = K / 128;
    if (K % 128) return false;
    // K / 128 = 32 (4096), 112 (14336), 64 (8192), 40 (5120), 16 (2048), 8 (1024), 28 (3584), 108 (13824: Llama-2-13B's
    // down_proj), 44 (5632: TinyLlama's)
    static const StreamShape cand[] = {{8, 4, 0}, {16, 7, 0}, {8, 8, 0}, {8, 5, 0}, {4, 4, 0}, {4, 2, 0}, {4, 7, 0}, {12, 9, 0},
                                       {4, 11, 0}};
    // one batch per tile (the activation fragments of a wave's K slice live in registers)
    for (const StreamShape& c : cand) {
        if (nsteps != c.NW * c.UB) continue;
        *sh = c;
        sh->NI = (K % 1024 == 0 && K / 1024 <= 8) ? K / 1024 : 0;
        return true;
    }
    return false;
}

static size_t stream_lds_bytes(int M, int K, int NW, bool staged_rows, int MT) {
    const int MP = M <= 4 ? 4 : (M <= 8 ? 8 : 16);
    const int NG = NW / 4, RB = NW == 4 ? 4 : (NW == 8 ? 2 : 1);
    return (staged_rows ? (size_t)MP * (K / 2 + 32) : 0) + 64 + (size_t)2 * NW * MT * 1024 + (size_t)MT * 1024 +
           (size_t)3 * NG * RB * 32 * 4;
}

// workgroups per launch above which a workgroup loops over several tiles
static int stream_cap() {
    static const int v = QS_DEV_KNOB("QSPEC_STREAM_CAP", 256);   // one workgroup per CU measured best (bench_stream.py)
    return v < 1 ? 256 : v;
}

template <int EPI, int PRO, int NW, int UB, int NI, int MT = 1, int DMA = 0>
static int launch_stream_inst(const StreamArgs& a, hipStream_t st) {
    const size_t lds = stream_lds_bytes(a.M, a.K, NW, PRO != PRO_Q, MT) + (DMA > 0 ? 1024 + (size_t)DMA * 32768 : 0);
    if (lds > 160 * 1024) return -7;
    static size_t attr_set = 0;  // per instantiation
    auto kern = gemm_w4a4_stream_kernel<EPI, PRO, NW, UB, NI, MT, DMA>;
    if (lds > 64 * 1024 && lds > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -8;
        attr_set = lds;
    }
    int cap = stream_cap();
    if (PRO == PRO_LNH && cap > 256) cap = 256;   // the fence-free hand-off is calibrated for one workgroup per CU
    const int slices = EPI == SEPI_IPART ? a.nq : 1;   // (nq carries the slice count for SEPI_IPART)
    if (slices > 1) cap = cap / slices > 0 ? cap / slices : 1;   // one workgroup per CU in total
    int grid = a.ntiles;
    if (grid > cap) {
        const int per = (a.ntiles + cap - 1) / cap;
        grid = (a.ntiles + per - 1) / per;
    }
    hipLaunchKernelGGL(kern, dim3(grid, slices), dim3(NW * 64 + (pro_split<PRO>() ? 256 : 0) + (DMA > 0 ? 256 : 0)), lds, st, a);
    return 0;
}

// (dev knob QSPEC_LN_SPLIT=0: the norm on the streaming waves themselves)
static bool ln_split() {
    static const int v = QS_DEV_KNOB("QSPEC_LN_SPLIT", 1);
    return v != 0;
}

#ifdef QS_EXPERIMENTAL
// QSPEC_DMA_TILES=0..3: tiles behind the first that a multi-tile workgroup takes through LDS-DMA.  Default 0: 1 tile gains
// 0.6 us on back-to-back gate_up launches (12.9 against 13.5), 2 / 3 tiles lose (13.45 / 13.75): round 3
static int dma_tiles() {
    static const int v = QS_DEV_KNOB("QSPEC_DMA_TILES", 0);
    return (v < 0 || v > 3) ? 0 : v;
}
#endif
template <int EPI>
static int launch_stream(const StreamArgs& a, bool ln, hipStream_t st) {
    StreamShape sh;
    if (a.M < 1 || a.M > 32 || !stream_shape(a.K, &sh)) return -1;
    if (a.M > 16) {   // two token tiles: (xq, xs) input, shapes with >= 512 threads
        if (ln) return -1;
        if (sh.NW == 8 && sh.UB == 4) return launch_stream_inst<EPI, PRO_Q, 8, 4, 0, 2>(a, st);
        if (sh.NW == 16 && sh.UB == 7) return launch_stream_inst<EPI, PRO_Q, 16, 7, 0, 2>(a, st);
        if (sh.NW == 8 && sh.UB == 8) return launch_stream_inst<EPI, PRO_Q, 8, 8, 0, 2>(a, st);
        if (sh.NW == 8 && sh.UB == 5) return launch_stream_inst<EPI, PRO_Q, 8, 5, 0, 2>(a, st);
        if (sh.NW == 12 && sh.UB == 9) return launch_stream_inst<EPI, PRO_Q, 12, 9, 0, 2>(a, st);
        return -1;
    }
    if (!ln) {
#ifdef QS_EXPERIMENTAL
        if constexpr (EPI == SEPI_RESID)
            if (a.K == 14336 && a.M <= 4 && a.ntiles <= 256 && engine_on()) return launch_engine_inst<EPI, PRO_Q, 0, 7, 7>(a, st);
#endif
        if (sh.NW == 8 && sh.UB == 4) return launch_stream_inst<EPI, PRO_Q, 8, 4, 0>(a, st);
        if (sh.NW == 16 && sh.UB == 7) return launch_stream_inst<EPI, PRO_Q, 16, 7, 0>(a, st);
        if (sh.NW == 8 && sh.UB == 8) return launch_stream_inst<EPI, PRO_Q, 8, 8, 0>(a, st);
        if (sh.NW == 8 && sh.UB == 5) return launch_stream_inst<EPI, PRO_Q, 8, 5, 0>(a, st);
        if (sh.NW == 4 && sh.UB == 4) return launch_stream_inst<EPI, PRO_Q, 4, 4, 0>(a, st);
        if (sh.NW == 4 && sh.UB == 2) return launch_stream_inst<EPI, PRO_Q, 4, 2, 0>(a, st);
        if (sh.NW == 4 && sh.UB == 7) return launch_stream_inst<EPI, PRO_Q, 4, 7, 0>(a, st);
        if (sh.NW == 12 && sh.UB == 9) return launch_stream_inst<EPI, PRO_Q, 12, 9, 0>(a, st);
        if (sh.NW == 4 && sh.UB == 11) return launch_stream_inst<EPI, PRO_Q, 4, 11, 0>(a, st);
        return -1;
    }
    // LN prologue: the reference's 1024 virtual threads -> K a multiple of 1024; one batch per tile
    if (a.sync) {   // producers + hand-off
        if (a.K == 4096) return launch_stream_inst<EPI, PRO_LNH, 8, 4, 4>(a, st);
        if (a.K == 8192) return launch_stream_inst<EPI, PRO_LNH, 8, 8, 8>(a, st);
        if (a.K == 5120) return launch_stream_inst<EPI, PRO_LNH, 8, 5, 5>(a, st);
        if (a.K == 2048) return launch_stream_inst<EPI, PRO_LNH, 4, 4, 2>(a, st);
        if (a.K == 1024) return launch_stream_inst<EPI, PRO_LNH, 4, 2, 1>(a, st);
        return -1;
    }
    if (!a.delta && !a.hidden_out) {   // pure norm of hidden_in: half the prologue loads
        // (16 waves instead of 8 at M <= 4 -- one row per 256-thread group, 4 waves per SIMD -- measured equal: qkv 9.4 vs
        // 9.1 us, gate_up 15.4 vs 15.6; the prologue is bound by the CU's VALU issue, not by one wave's latency chain)
        if (a.M <= 4 && ln_split()) {   // (K = 8192: the row-wave's 128 values per lane do not fit 168 VGPRs)
#ifdef QS_EXPERIMENTAL   // the three LDS-DMA forms (Stage A: measured, none is the default)
            if constexpr (EPI == SEPI_GATEUP)
                if (a.K == 4096 && engine_on()) return launch_engine_inst<EPI, PRO_LN1S, 4, 2, 8>(a, st);
            if constexpr (EPI == SEPI_GATEUP || EPI == SEPI_QKV)
                if (a.K == 4096 && sdma_on<EPI>()) return launch_sdma_inst<EPI, PRO_LN1S, 4, QS_SDMA_R>(a, st);
            if (a.K == 4096 && a.ntiles > 256) {   // several tiles per workgroup: the next ones through LDS-DMA
                const int dt = dma_tiles();
                if (dt == 3) return launch_stream_inst<EPI, PRO_LN1S, 8, 4, 4, 1, 3>(a, st);
                if (dt == 2) return launch_stream_inst<EPI, PRO_LN1S, 8, 4, 4, 1, 2>(a, st);
                if (dt == 1) return launch_stream_inst<EPI, PRO_LN1S, 8, 4, 4, 1, 1>(a, st);
            }
#endif
            if (a.K == 4096) return launch_stream_inst<EPI, PRO_LN1S, 8, 4, 4>(a, st);
            if (a.K == 5120) return launch_stream_inst<EPI, PRO_LN1S, 8, 5, 5>(a, st);
            if (a.K == 2048) return launch_stream_inst<EPI, PRO_LN1S, 4, 4, 2>(a, st);
            if (a.K == 1024) return launch_stream_inst<EPI, PRO_LN1S, 4, 2, 1>(a, st);
        }
        if (a.K == 4096) return launch_stream_inst<EPI, PRO_LN1, 8, 4, 4>(a, st);
        if (a.K == 8192) return launch_stream_inst<EPI, PRO_LN1, 8, 8, 8>(a, st);
        if (a.K == 5120) return launch_stream_inst<EPI, PRO_LN1, 8, 5, 5>(a, st);
        if (a.K == 2048) return launch_stream_inst<EPI, PRO_LN1, 4, 4, 2>(a, st);
        if (a.K == 1024) return launch_stream_inst<EPI, PRO_LN1, 4, 2, 1>(a, st);
        return -1;
    }
    if (a.M <= 4 && ln_split()) {
#ifdef QS_EXPERIMENTAL
        if constexpr (EPI == SEPI_GATEUP || EPI == SEPI_QKV)
            if (a.K == 4096 && sdma_on<EPI>()) return launch_sdma_inst<EPI, PRO_LNS, 4, QS_SDMA_R>(a, st);
#endif
        if (a.K == 4096) return launch_stream_inst<EPI, PRO_LNS, 8, 4, 4>(a, st);
        if (a.K == 5120) return launch_stream_inst<EPI, PRO_LNS, 8, 5, 5>(a, st);
        if (a.K == 2048) return launch_stream_inst<EPI, PRO_LNS, 4, 4, 2>(a, st);
        if (a.K == 1024) return launch_stream_inst<EPI, PRO_LNS, 4, 2, 1>(a, st);
    }
    if (a.K == 4096) return launch_stream_inst<EPI, PRO_LN, 8, 4, 4>(a, st);
    if (a.K == 8192) return launch_stream_inst<EPI, PRO_LN, 8, 8, 8>(a, st);
    if (a.K == 5120) return launch_stream_inst<EPI, PRO_LN, 8, 5, 5>(a, st);
    if (a.K == 2048) return launch_stream_inst<EPI, PRO_LN, 4, 4, 2>(a, st);
    if (a.K == 1024) return launch_stream_inst<EPI, PRO_LN, 4, 2, 1>(a, st);
    return -1;
}

size_t gemm_w4a4_stream_sync_bytes() { return 32 * sizeof(int) + (size_t)16 * (8192 / 2) + 16 * sizeof(float) + 64; }

// out[m,n] = h( (p_0 + p_1 + ...)[m,n] * f(sw[n]) ): K slices combined in slice order (deterministic).  The verify
// pass does this inside the next norm kernel instead (norm_quant.hip, ln_kernel with `part`); same expression.
__global__ __launch_bounds__(256) void w4a16_partial_finish_kernel(const float* __restrict__ part, const f16* __restrict__ ws,
                                                                     f16* __restrict__ out, int MN, int N, int S) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= MN) return;
    float sum = part[i];
    for (int s2 = 1; s2 < S; s2++) sum = sum + part[(size_t)s2 * MN + i];
    out[i] = f2h(sum * h2f(ws[i % N]));
}

// ---- 17..32 tokens: TWO 16-token tiles over one pass of the weight stream, K in NB passes.
// The M-tiled kernel (gemm_tiled.hip, built for prompt-sized M) dequantises every packed dword for one 32-token block only and
// leaves a SIMD with one wave: Llama-3-70B's verify pass at 32 tokens spent 151 us per layer in its four GEMMs where the draft
// pass streams the same bytes in 82.  Here the streaming kernel takes a second token tile: x is TWO fragment-major 16-row tiles
// (common.cuh: w4a16_xperm_offset32; the producers write them), loaded straight into the MFMA operand registers -- 2 x UB x 16
// VGPRs, which is why K is walked in NB passes of 128 * NW * UB k: pass p holds the fragments of its K range, streams that range
// of EVERY tile of the workgroup and leaves the tile's fp32 sums in LDS (tsum; wave order, then pass order: deterministic); the
// last pass adds its own and runs the epilogue.  8 waves: the tile's 32 x 16 outputs are one per thread.
template <int EPI, int NW, int UB, int NB>
__global__ __launch_bounds__(NW * 64) void gemm_w4a16_stream2_kernel(StreamArgs a) {
    static_assert(NW == 8, "one epilogue thread per output of the two-tile block");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, g = lane >> 4;
    // SEPI_PARTIAL (NB = 1): blockIdx.y picks one of gridDim.y K slices of length K out of rows of ldx halves (x tiles) / ldw bytes
    // (weights); the raw fp32 sums go to part[blockIdx.y] and are combined in slice order by the norm that follows
    const size_t kofs = EPI == SEPI_PARTIAL ? (size_t)blockIdx.y * a.K : 0;
    const size_t ldw = a.ldw ? (size_t)a.ldw : (size_t)(a.K >> 1), Kx = a.ldx ? (size_t)a.ldx : (size_t)a.K;
    float* red = reinterpret_cast<float*>(smem);                     // [2][NW][2][256]
    f16* ex = reinterpret_cast<f16*>(red + 2 * NW * 512);            // [2][512]
    float* tsum = reinterpret_cast<float*>(ex + 2 * 512);            // [tiles of this workgroup][512] (NB > 1)
    const int c = tid & 15, m = tid >> 4;                            // epilogue thread: (token m of 32, tile column c)
    const bool ethread = m < a.M;
    const int ridx = (m >> 4) * 256 + (m & 3) * 64 + ((m >> 2) & 3) * 16 + c;
    const int mc = m < a.M ? m : 0;
    int64_t pos_m = 0, slot_m = -1;
    if (EPI == SEPI_QKV) {
        pos_m = a.positions[mc];
        slot_m = a.slot_mapping[mc];
    }
    struct Pre {
        f16 swn, cf, sf;
    };
    auto load_pre = [&](Pre& pre, int tile) {
        if (EPI != SEPI_PARTIAL) pre.swn = a.ws[stile_row<EPI>(tile, c, a.I, a.hdl)];
        if (EPI == SEPI_QKV) {
            const int o = qkv_pair(tile, c & 7, a.I, a.hdl).i;
            const f16* cs = a.cos_sin_cache + (pos_m << a.hdl);
            pre.cf = cs[o];
            pre.sf = cs[(1 << (a.hdl - 1)) + o];
        }
    };
    auto wptr = [&](int tile, int p) -> const uint8_t* {
        return a.wq + (size_t)stile_row<EPI>(tile, r, a.I, a.hdl) * ldw + (kofs >> 1) + (size_t)p * (NW * UB * 64) + g * 16;
    };
    const int tile_first = blockIdx.x;
    const int my_tiles = (a.ntiles - tile_first + (int)gridDim.x - 1) / (int)gridDim.x;
    f16x8 af[2][UB][4];
    auto load_af = [&](int p) {
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
            const f16* xp = a.x + (size_t)mt * 16 * Kx + (kofs >> 7) * 2048 + (size_t)lane * 8;
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int kstep = p * (NW * UB) + (step_off<NW, UB>(wave, u) >> 6);
#pragma unroll
                for (int dd = 0; dd < 4; dd++) af[mt][u][dd] = *reinterpret_cast<const f16x8*>(xp + ((size_t)kstep * 4 + dd) * 512);
            }
        }
    };
    f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
    auto use = [&](const u32x4& wv, int u) {
#pragma unroll
        for (int dd = 0; dd < 4; dd++) {
            const f16x8 b = sdequant_s4x8(wv[dd]);
            acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[0][u][dd], b, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af[1][u][dd], b, acc[1], 0, 0, 0);
        }
    };
    int par = 0;
    auto finish = [&](int tile, int ti, int p, const Pre& pre) {
        float* rb = red + par * NW * 512;
#pragma unroll
        for (int mt = 0; mt < 2; mt++) {
#pragma unroll
            for (int i = 0; i < 4; i++) rb[(wave * 2 + mt) * 256 + i * 64 + lane] = acc[mt][i];
            acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        __syncthreads();
        float sum = rb[ridx];
#pragma unroll
        for (int w2 = 1; w2 < NW; w2++) sum = sum + rb[w2 * 512 + ridx];   // wave order: deterministic
        par ^= 1;
        if (NB > 1) {
            if (p > 0) sum = tsum[ti * 512 + tid] + sum;                   // pass order
            if (p < NB - 1) {
                tsum[ti * 512 + tid] = sum;
                return;
            }
        }
        if (EPI == SEPI_PARTIAL) {
            if (ethread) a.part[((size_t)blockIdx.y * a.M + m) * a.N + tile * 16 + c] = sum;
            return;
        }
        const f16 hv = f2h(sum * h2f(pre.swn));
        if (EPI == SEPI_PLAIN) {
            if (ethread) a.out[(size_t)m * a.N + tile * 16 + c] = hv;
            return;
        }
        f16* e = ex + (par ^ 1) * 512;
        e[tid] = hv;
        __syncthreads();
        if (!ethread) return;
        const f16 partner = e[tid ^ 8];
        if (EPI == SEPI_GATEUP) {
            if (c < 8) {
                const float gt = h2f(partner);
                const float act = h2f(f2h(gt / (1.0f + qexpf(-gt))));
                a.out[(size_t)m * a.I + tile * 8 + c] = f2h(act * h2f(hv));
            }
            return;
        }
        const QkvPair qp = qkv_pair(tile, c & 7, a.I, a.hdl);
        const int head = qp.head, o = qp.i;
        const int n = (head << a.hdl) + ((c >> 3) << (a.hdl - 1)) + o;
        f16 res = hv;
        if (head < a.nq + a.nkv) {
            const float cff = h2f(pre.cf), sff = h2f(pre.sf);
            const float xf = h2f(c < 8 ? hv : partner), yf = h2f(c < 8 ? partner : hv);
            res = c < 8 ? f2h(h2f(f2h(xf * cff)) - h2f(f2h(yf * sff))) : f2h(h2f(f2h(yf * cff)) + h2f(f2h(xf * sff)));
        }
        a.out[(size_t)m * a.N + n] = res;
        if (head >= a.nq && slot_m >= 0) {
            const bool is_k = head < a.nq + a.nkv;
            const int kvh = is_k ? head - a.nq : head - a.nq - a.nkv;
            f16* cache = is_k ? a.key_cache : a.value_cache;
            cache[((slot_m * a.nkv + kvh) << a.hdl) + ((c >> 3) << (a.hdl - 1)) + o] = res;
        }
    };
    // the (pass, tile) units of this workgroup in order; the weights of unit q + 1 are requested behind the use of unit q's
    const int n_units = NB * my_tiles;
    u32x4 w[UB];
    Pre pre = {};
    load_af(0);
    __builtin_amdgcn_sched_barrier(0);
    {
        const uint8_t* wp0 = wptr(tile_first, 0);
#pragma unroll
        for (int u = 0; u < UB; u++) w[u] = wload<u32x4>(wp0 + step_off<NW, UB>(wave, u));
        if (NB == 1) load_pre(pre, tile_first);
    }
    __builtin_amdgcn_sched_barrier(0);
    int p = 0, ti = 0, tile = tile_first;
    for (int q = 0; q < n_units - 1; q++) {
        int np = p, nti = ti + 1, ntile = tile + (int)gridDim.x;
        if (nti == my_tiles) {
            np = p + 1;
            nti = 0;
            ntile = tile_first;
        }
        const uint8_t* wp = wptr(ntile, np);
        Pre npre = {};
        if (np == NB - 1) load_pre(npre, ntile);
#pragma unroll
        for (int u = 0; u < UB; u++) {
            use(w[u], u);
            __builtin_amdgcn_sched_barrier(0);
            w[u] = wload<u32x4>(wp + step_off<NW, UB>(wave, u));
            __builtin_amdgcn_sched_barrier(0);
        }
        finish(tile, ti, p, pre);
        if (np != p) load_af(np);   // (behind the last use of pass p's fragments)
        pre = npre;
        p = np;
        ti = nti;
        tile = ntile;
    }
#pragma unroll
    for (int u = 0; u < UB; u++) use(w[u], u);
    finish(tile, ti, p, pre);
}

// shapes of the two-tile kernel: K = 128 * 8 * UB * NB
static bool stream32_shape(int K, int* UB, int* NB) {
    static const int cand[][3] = {{4096, 4, 1}, {8192, 4, 2}, {28672, 4, 7}, {14336, 2, 7}, {5120, 5, 1}};
    for (const auto& cnd : cand)
        if (K == cnd[0]) {
            *UB = cnd[1];
            *NB = cnd[2];
            return true;
        }
    return false;
}
bool gemm_w4a16_stream32_supported(int M, int N, int K) {
    int UB, NB;
    if (M < 17 || M > 32 || N % 16 || !stream32_shape(K, &UB, &NB)) return false;
    const int cap = stream_cap(), nt = N / 16;
    const int per = (nt + cap - 1) / cap;
    return NB == 1 || (size_t)per * 2048 <= 96 * 1024;   // tsum: 2 KB per tile of a workgroup
}
template <int EPI, int UB, int NB>
static int launch_stream32_inst(const StreamArgs& a, hipStream_t st) {
    const int slices = EPI == SEPI_PARTIAL ? a.nq : 1;   // (nq carries the slice count for SEPI_PARTIAL)
    const int cap = slices > 1 ? (stream_cap() / slices > 0 ? stream_cap() / slices : 1) : stream_cap();   // one workgroup per CU in all
    int grid = a.ntiles, per = 1;
    if (grid > cap) {
        per = (a.ntiles + cap - 1) / cap;
        grid = (a.ntiles + per - 1) / per;
    }
    const size_t lds = (size_t)2 * 8 * 512 * 4 + 2 * 512 * 2 + (NB > 1 ? (size_t)per * 2048 : 0);
    static size_t attr_set = 0;
    if (lds > 64 * 1024 && lds > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4a16_stream2_kernel<EPI, 8, UB, NB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -8;
        attr_set = lds;
    }
    hipLaunchKernelGGL((gemm_w4a16_stream2_kernel<EPI, 8, UB, NB>), dim3(grid, slices), dim3(8 * 64), lds, st, a);
    return 0;
}
// K slices of the two-tile kernel for a long-K layer whose sums the following norm finishes (down_proj at 17..32 tokens): S one-pass
// slices of 4096 (UB = 4) or 2048 (UB = 2) k across the workgroups instead of S passes inside one -- 0 where no such cut exists
int gemm_w4a16_stream32_partial_slices(int M, int N, int K) {
    if (M < 17 || M > 32 || N % 16) return 0;
    for (int len : {4096, 2048})
        if (K % len == 0 && K / len >= 2 && K / len <= 8) return K / len;
    return 0;
}
int gemm_w4a16_stream32_partial(const f16* x, const int8_t* wq, float* part, int M, int N, int K, int S, hipStream_t st) {
    if (S < 2 || S != gemm_w4a16_stream32_partial_slices(M, N, K) || !part) return -1;
    StreamArgs a{};
    a.x = x; a.ldx = K; a.ldw = K / 2; a.wq = reinterpret_cast<const uint8_t*>(wq); a.M = M; a.N = N; a.K = K / S; a.ntiles = N / 16;
    a.part = part; a.nq = S;
    if (a.K == 4096) return launch_stream32_inst<SEPI_PARTIAL, 4, 1>(a, st);
    return launch_stream32_inst<SEPI_PARTIAL, 2, 1>(a, st);
}
template <int EPI>
static int launch_stream32(const StreamArgs& a, hipStream_t st) {
    int UB, NB;
    if (!gemm_w4a16_stream32_supported(a.M, a.N, a.K) || !stream32_shape(a.K, &UB, &NB)) return -1;
#define QS_S32(UBV, NBV) if (UB == UBV && NB == NBV) return launch_stream32_inst<EPI, UBV, NBV>(a, st);
    QS_S32(4, 1) QS_S32(4, 2) QS_S32(4, 7) QS_S32(2, 7) QS_S32(5, 1)
#undef QS_S32
    return -1;
}
// x: two fragment-major 16-row tiles (rows 0..15, 16..31), each 16 x K halves
int gemm_w4a16_stream32(const f16* x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K, hipStream_t st) {
    StreamArgs a{};
    a.x = x; a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = out; a.M = M; a.N = N; a.K = K; a.ntiles = N / 16;
    return launch_stream32<SEPI_PLAIN>(a, st);
}
int gemm_w4a16_stream32_qkv_rope(const f16* x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                                 const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                                 const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st) {
    if ((d != 128 && d != 64) || rot_dim != d || N != (nq + 2 * nkv) * d) return -1;
    StreamArgs a{};
    a.hdl = d == 64 ? 6 : 7;
    a.x = x; a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = qkv; a.M = M; a.N = N; a.K = K;
    a.ntiles = N / 16; a.positions = positions; a.cos_sin_cache = cos_sin_cache; a.key_cache = key_cache;
    a.value_cache = value_cache; a.slot_mapping = slot_mapping; a.nq = nq; a.nkv = nkv;
    return launch_stream32<SEPI_QKV>(a, st);
}
int gemm_w4a16_stream32_gate_up_silu(const f16* x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K, hipStream_t st) {
    if (I % 8) return -1;
    StreamArgs a{};
    a.x = x; a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = act; a.M = M; a.N = 2 * I; a.K = K; a.I = I;
    a.ntiles = 2 * I / 16;
    return launch_stream32<SEPI_GATEUP>(a, st);
}

template <int EPI, int NW, int UB>
static int launch_stream16_inst(const StreamArgs& a, hipStream_t st) {
    const size_t lds = (size_t)2 * NW * 1024 + 1024 + (size_t)2 * 16 * NW * 512;   // reduction + activation staging
    static size_t attr_set = 0;
    if (lds > 64 * 1024 && lds > attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_w4a16_stream_kernel<EPI, NW, UB>),
                                hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return -8;
        attr_set = lds;
    }
    const int cap = stream_cap();
    int grid = a.ntiles;
    if (grid > cap) {
        const int per = (a.ntiles + cap - 1) / cap;
        grid = (a.ntiles + per - 1) / per;
    }
    const int slices = EPI == SEPI_PARTIAL ? a.nq : 1;   // (nq carries the slice count for SEPI_PARTIAL)
    if (slices > 1) {   // one workgroup per CU in total: the tile groups share the grid with the slices
        const int per_slice = cap / slices > 0 ? cap / slices : 1;
        if (grid > per_slice) {
            const int per = (a.ntiles + per_slice - 1) / per_slice;
            grid = (a.ntiles + per - 1) / per;
        }
    }
    if constexpr ((NW == 8 && (UB == 4 || UB == 5)) || (NW == 4 && (UB == 7 || UB == 9))) {   // the Llama-3-8B / Llama-2-13B shapes
        if (a.xperm) {
            hipLaunchKernelGGL((gemm_w4a16_stream_kernel<EPI, NW, UB, true>), dim3(grid, slices), dim3(NW * 64),
                               (size_t)2 * NW * 1024 + 1024, st, a);
            return 0;
        }
    }
    if (a.xperm) return -1;
    hipLaunchKernelGGL((gemm_w4a16_stream_kernel<EPI, NW, UB>), dim3(grid, slices), dim3(NW * 64), lds, st, a);
    return 0;
}

// one batch per tile only (the activation fragments of a wave's K slice live in registers): K = 128 * NW * UB
static bool stream16_shape(int K, int* NW, int* UB) {
    // (16, 7) = K 14336 is NOT here: 16 rows x 14336 k of fp16 activations (458 KB) do not fit one CU's registers
    // (16 waves x 112 VGPRs would leave nothing else); that layer stays on gemm.hip's 2-D kernel
    // (4, 9) = K 4608: a third of Llama-2-13B's down_proj (13824 = 3 x 4608, gemm_w4a16_stream_partial_slices); the
    // activation staging buffer is 2 x 16 rows x NW x 512 B, so NW stays <= 8 (12 waves x 3 steps would need 222 KB of LDS)
    static const int cand[][2] = {{8, 4}, {8, 8}, {8, 5}, {4, 4}, {4, 2}, {8, 7}, {4, 7}, {4, 1}, {4, 9}, {4, 11}};   // (4, 11) = K 5632: TinyLlama's down_proj
    if (K % 128) return false;
    for (const auto& c : cand)
        if (K / 128 == c[0] * c[1]) {
            *NW = c[0];
            *UB = c[1];
            return true;
        }
    return false;
}

bool gemm_w4a16_stream_supported(int M, int N, int K) {
    int NW, UB;
    return M >= 1 && M <= 16 && N % 16 == 0 && stream16_shape(K, &NW, &UB);
}

template <int EPI>
static int launch_stream16(const StreamArgs& a, hipStream_t st) {
    int NW, UB;
    if (a.M < 1 || a.M > 16 || !stream16_shape(a.K, &NW, &UB)) return -1;
#define QS_S16(NWV, UBV) if (NW == NWV && UB == UBV) return launch_stream16_inst<EPI, NWV, UBV>(a, st);
    QS_S16(8, 4) QS_S16(8, 8) QS_S16(8, 5) QS_S16(4, 4) QS_S16(4, 2) QS_S16(8, 7) QS_S16(4, 7) QS_S16(4, 1) QS_S16(4, 9) QS_S16(4, 11)
#undef QS_S16
    return -1;
}

// xperm (every entry below): x is the 16-row activation tile in fragment-major layout (common.cuh: w4a16_xperm_offset; the
// producers write it: norm_quant.hip ln_kernel, hadamard.hip spread forms) -- built for the shapes gemm_w4a16_xperm_supported names
bool gemm_w4a16_xperm_supported(int M, int K) {
    int NW, UB;
    if (M < 1 || M > 16 || !stream16_shape(K, &NW, &UB)) return false;
    return (NW == 8 && (UB == 4 || UB == 5)) || (NW == 4 && (UB == 7 || UB == 9));   // K = 4096 / 5120 / 3584 / 4608
}
int gemm_w4a16_stream(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, const f16* ws, f16* out, int M, int N,
                      int K, hipStream_t st, int xperm) {
    if (!gemm_w4a16_stream_supported(M, N, K) || (ldx && ldx % 8) || (ldw && ldw % 16)) return -1;
    if (xperm && (ldx || !gemm_w4a16_xperm_supported(M, K))) return -1;
    StreamArgs a{};
    a.x = x; a.ldx = ldx; a.ldw = ldw; a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = out; a.xperm = xperm;
    a.M = M; a.N = N; a.K = K; a.ntiles = N / 16;
    return launch_stream16<SEPI_PLAIN>(a, st);
}

// Long-K layers (down_proj, K = 14336): 16 rows x K of fp16 activations do not fit one CU's registers, so K is cut
// into S slices of a built length (14336 = 2 x 7168), slice s on workgroups (tile group, s); raw fp32 sums per slice
// go to part [S][M][N].  Returns the slice count to use for (M, N, K), 0 if the shape is not covered.
int gemm_w4a16_stream_partial_slices(int M, int N, int K) {
    if (gemm_w4a16_stream_supported(M, N, K)) return 0;   // fits unsliced: use the plain entry
    static const int forced = QS_DEV_KNOB("QSPEC_W4A16_SLICES", 0);   // (dev knob for sweeps)
    if (forced >= 2 && forced <= 4 && K % forced == 0 && gemm_w4a16_stream_supported(M, N, K / forced)) return forced;
    // Four slices first: a workgroup stages the whole [16, K / S] fp16 activation slice (16 * K / S * 2 bytes) for
    // tiles-per-workgroup * 16 * K / S / 2 bytes of weights, and the grid is one workgroup per CU in all, so S slices mean
    // S tiles per workgroup at N = 4096: S = 2 moves 229 KB of activations for 115 KB of weights per workgroup, S = 4
    // 115 KB for 115 KB (Llama-3-8B down_proj, verify pass: cycle 7.66 -> 7.61 ms).
    static const int order[3] = {4, 2, 3};
    for (int S : order)
        if (K % S == 0 && gemm_w4a16_stream_supported(M, N, K / S)) return S;
    return 0;
}
int gemm_w4a16_stream_partial(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, float* part, int M, int N, int K,
                              int S, hipStream_t st, int xperm) {
    if (S < 1 || K % S || !gemm_w4a16_stream_supported(M, N, K / S) || !part) return -1;
    if (xperm && (ldx || (K / S) % 128 || !gemm_w4a16_xperm_supported(M, K / S))) return -1;
    StreamArgs a{};
    a.xperm = xperm;
    a.x = x; a.ldx = ldx ? ldx : K; a.ldw = ldw ? ldw : K / 2; a.wq = reinterpret_cast<const uint8_t*>(wq);
    a.M = M; a.N = N; a.K = K / S; a.ntiles = N / 16; a.part = part; a.nq = S;
    return launch_stream16<SEPI_PARTIAL>(a, st);
}
int gemm_w4a16_partial_finish(const float* part, const f16* ws, f16* out, int M, int N, int S, hipStream_t st) {
    const int MN = M * N;
    hipLaunchKernelGGL(w4a16_partial_finish_kernel, dim3((MN + 255) / 256), dim3(256), 0, st, part, ws, out, MN, N, S);
    return 0;
}

// Levelled QKV tiling (qkv_pair): L = workgroups = pairs / 12, or 0 for the classic 16-row tiles.  Only where the classic
// tiling is uneven (more tiles than workgroups, not a multiple of them) and the pairs divide by twelve: Llama-3-8B's
// (32 + 8 + 8) heads x 64 pairs = 3072 = 12 x 256.  OFF by default (QSPEC_QKV_LEVEL=1 switches it on): measured in round 3,
// bit-identical, but the draft qkv launch takes 7.25-7.38 us against 6.89 and the cycle 7.58-7.65 ms against 7.51-7.54 -- two
// tile iterations (reduction, barriers, epilogue) in EVERY workgroup cost more than the 16 KB less that the slowest ones stream.
#ifdef QS_EXPERIMENTAL
static int qkv_level_workgroups(int nq, int nkv, int N) {
    static const int on = QS_DEV_KNOB("QSPEC_QKV_LEVEL", 0);
    const int pairs = (nq + 2 * nkv) * 64, cap = stream_cap(), classic = N / 16;
    if (!on || pairs % 12 || classic <= cap || classic % cap == 0 || pairs / 12 > cap) return 0;
    return pairs / 12;
}
#endif

int gemm_w4a16_stream_qkv_rope(const f16* x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                               const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                               const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st, int xperm) {
    if ((d != 128 && d != 64) || rot_dim != d || N != (nq + 2 * nkv) * d || !gemm_w4a16_stream_supported(M, N, K)) return -1;
    if (xperm && !gemm_w4a16_xperm_supported(M, K)) return -1;
    StreamArgs a{};
    a.xperm = xperm;
    a.hdl = d == 64 ? 6 : 7;
    a.x = x; a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = qkv; a.M = M; a.N = N; a.K = K;
    a.ntiles = N / 16; a.positions = positions; a.cos_sin_cache = cos_sin_cache; a.key_cache = key_cache;
    a.value_cache = value_cache; a.slot_mapping = slot_mapping; a.nq = nq; a.nkv = nkv;
#ifdef QS_EXPERIMENTAL
    if (const int L = d == 128 ? qkv_level_workgroups(nq, nkv, N) : 0) {   // twelve RoPE pairs per workgroup: a full + a half tile
        a.I = L;
        a.ntiles = 2 * L;
    }
#endif
    return launch_stream16<SEPI_QKV>(a, st);
}

int gemm_w4a16_stream_gate_up_silu(const f16* x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K, int ch0,
                                   int nch, hipStream_t st, int xperm) {
    if (I % 8 || ch0 % 8 || nch % 8 || ch0 + nch > I || !gemm_w4a16_stream_supported(M, 2 * I, K)) return -1;
    if (xperm && !gemm_w4a16_xperm_supported(M, K)) return -1;
    StreamArgs a{};
    a.xperm = xperm;
    a.x = x; a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = act; a.M = M; a.N = 2 * I; a.K = K;
    a.I = I; a.tile0 = ch0 / 8; a.ntiles = nch / 8;
    return launch_stream16<SEPI_GATEUP>(a, st);
}

bool gemm_w4a4_stream_supported(int M, int N, int K, bool ln) {
    StreamShape sh;
    if (!ln && M >= 1 && M <= 16 && N % 16 == 0 && longk_shape(K)) return true;   // several batches per tile (longk kernel)
    if (M < 1 || M > 32 || N % 16 || K > (1 << 19) || !stream_shape(K, &sh)) return false;
    if (M > 16 && (ln || sh.NW < 8)) return false;   // two token tiles: (xq, xs) input only
    if (ln && !(K == 1024 || K == 2048 || K == 4096 || K == 5120 || K == 8192)) return false;
    return stream_lds_bytes(M, K, sh.NW, ln, M > 16 ? 2 : 1) <= 160 * 1024;
}

// act: either (xq, xs) or (hidden_in, delta, hidden_out, eps) -- see StreamArgs.
int gemm_w4a4_stream(const StreamActs& x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K,
                     hipStream_t st) {
    if (!gemm_w4a4_stream_supported(M, N, K, x.hidden_in != nullptr)) return -1;
    StreamArgs a{};
    a.xq = x.xq; a.xs = x.xs; a.hidden_in = x.hidden_in; a.delta = x.delta; a.hidden_out = x.hidden_out; a.eps = x.eps; a.sync = x.sync;
    a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = out; a.M = M; a.N = N; a.K = K; a.ntiles = N / 16;
    if (longk_shape(K)) return launch_longk<SEPI_PLAIN>(a, st);
    return launch_stream<SEPI_PLAIN>(a, x.hidden_in != nullptr, st);
}

// resid_out = h(f(resid_in) + f(h(x W^T scales)))  -- the projection and the residual add that follows it
int gemm_w4a4_stream_residual(const StreamActs& x, const int8_t* wq, const f16* ws, const f16* resid_in, f16* resid_out,
                              int M, int N, int K, hipStream_t st) {
    if (!gemm_w4a4_stream_supported(M, N, K, x.hidden_in != nullptr)) return -1;
    StreamArgs a{};
    a.xq = x.xq; a.xs = x.xs; a.hidden_in = x.hidden_in; a.delta = x.delta; a.hidden_out = x.hidden_out; a.eps = x.eps; a.sync = x.sync;
    a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.M = M; a.N = N; a.K = K; a.ntiles = N / 16;
    a.resid_in = resid_in; a.resid_out = resid_out;
    if (longk_shape(K)) return launch_longk<SEPI_RESID>(a, st);
    return launch_stream<SEPI_RESID>(a, x.hidden_in != nullptr, st);
}

// K-sliced W4A4 at 17..32 tokens: slice count for (M, N, K), 0 = not built / does not pay.  Two slices where a workgroup
// would otherwise hold ONE 16-row tile (N / 16 <= the chip's 256 workgroups): K = 14336 -> 2 x 7168 (Llama-3-8B down_proj).
int gemm_w4a4_stream_partial_slices(int M, int N, int K) {
    if (M < 17 || M > 32 || N % 32 || N / 16 > stream_cap()) return 0;
    if (K == 14336) return 2;
    return 0;
}
int gemm_w4a4_stream_partial(const int8_t* xq, const int8_t* wq, int* ipart, int M, int N, int K, int S, hipStream_t st) {
    if (S != gemm_w4a4_stream_partial_slices(M, N, K) || S < 2 || !ipart) return -1;
    StreamArgs a{};
    a.xq = xq; a.wq = reinterpret_cast<const uint8_t*>(wq); a.ipart = ipart; a.M = M; a.N = N; a.K = K / S; a.ntiles = N / 16;
    a.ldx = K / 2; a.ldw = K / 2; a.nq = S;
    // K slices of 7168 (14336 / 2), two token tiles: 8 waves x 7 steps
    if (a.K == 7168) return launch_stream_inst<SEPI_IPART, PRO_Q, 8, 7, 0, 2>(a, st);
    return -1;
}

bool gemm_w4a4_stream_residual_hq_supported(int M, int N, int K, int nparts) {
    return M >= 1 && M <= 4 && (K == 4096 || K == 5120) && nparts == 8 && N % 16 == 0 && N > 0;
}
int gemm_w4a4_stream_residual_hq(const f16* x16, const float* part_amax, int nparts, float clip, const int8_t* wq, const f16* ws,
                                 const f16* resid_in, f16* resid_out, int M, int N, int K, hipStream_t st) {
    if (!gemm_w4a4_stream_residual_hq_supported(M, N, K, nparts)) return -1;
    StreamArgs a{};
    a.x16 = x16; a.part_amax = part_amax; a.clip = clip;
    a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.M = M; a.N = N; a.K = K; a.ntiles = N / 16;
    a.resid_in = resid_in; a.resid_out = resid_out;
    if (K == 5120) return launch_stream_inst<SEPI_RESID, PRO_RQ, 8, 5, 5>(a, st);
    return launch_stream_inst<SEPI_RESID, PRO_RQ, 8, 4, 4>(a, st);
}
int gemm_w4a4_stream_qkv_rope(const StreamActs& x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                              const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                              const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st) {
    if ((d != 128 && d != 64) || rot_dim != d || N != (nq + 2 * nkv) * d) return -1;
    if (!gemm_w4a4_stream_supported(M, N, K, x.hidden_in != nullptr)) return -1;
    StreamArgs a{};
    a.hdl = d == 64 ? 6 : 7;
    a.xq = x.xq; a.xs = x.xs; a.hidden_in = x.hidden_in; a.delta = x.delta; a.hidden_out = x.hidden_out; a.eps = x.eps; a.sync = x.sync;
    a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = qkv; a.M = M; a.N = N; a.K = K; a.ntiles = N / 16;
    a.positions = positions; a.cos_sin_cache = cos_sin_cache; a.key_cache = key_cache; a.value_cache = value_cache;
    a.slot_mapping = slot_mapping; a.nq = nq; a.nkv = nkv;
#ifdef QS_EXPERIMENTAL
    if (!sdma_on<SEPI_QKV>()) {   // (the LDS-DMA forms keep the classic tiles)
        if (const int L = d == 128 ? qkv_level_workgroups(nq, nkv, N) : 0) {
            a.I = L;
            a.ntiles = 2 * L;
        }
    }
#endif
    return launch_stream<SEPI_QKV>(a, x.hidden_in != nullptr, st);
}

int gemm_w4a4_stream_gate_up_silu(const StreamActs& x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K,
                                  hipStream_t st) {
    if (I % 8) return -1;
    if (!gemm_w4a4_stream_supported(M, 2 * I, K, x.hidden_in != nullptr)) return -1;
    StreamArgs a{};
    a.xq = x.xq; a.xs = x.xs; a.hidden_in = x.hidden_in; a.delta = x.delta; a.hidden_out = x.hidden_out; a.eps = x.eps; a.sync = x.sync;
    a.wq = reinterpret_cast<const uint8_t*>(wq); a.ws = ws; a.out = act; a.M = M; a.N = 2 * I; a.K = K; a.ntiles = I / 8;
    a.I = I;
    return launch_stream<SEPI_GATEUP>(a, x.hidden_in != nullptr, st);
}

}  // namespace qspec

#ifdef QS_STREAM_STAMPS
extern "C" int qspec_debug_stamps(long long* host8) {   // dev builds only (scripts/stream_stamps.py)
    return hipMemcpyFromSymbol(host8, HIP_SYMBOL(qspec::g_sstamps), 8 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
extern "C" int qspec_debug_lnst(long long* host8) {
    return hipMemcpyFromSymbol(host8, HIP_SYMBOL(qspec::g_lnst), 8 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
#ifdef QS_ENG_STAMPS
extern "C" int qspec_debug_engst(long long* host128) {
    return hipMemcpyFromSymbol(host128, HIP_SYMBOL(qspec::g_engst), 384 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
#ifdef QS_STREAM_STAMPS
extern "C" int qspec_debug_wgspan(long long* host2048) {
    return hipMemcpyFromSymbol(host2048, HIP_SYMBOL(qspec::g_wgspan), 2048 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
#endif
