// W4A16 GEMM for prefill-sized M (prompt pass, verify pass of large batches): matrix-core bound, M-tiled.
//
//   out[m,n] = h( (sum_k f(x[m,k]) * w[n,k]) * f(sw[n]) ),  fp32 accumulate  (bitblas.Matmul, quarot_nn/linear.py:102-124)
//
// over the same packed int4 buffer as every other GEMM of the engine (no dequantised copy, no library GEMM).
//   * Workgroup = 4 waves, tile = (32 * MT) tokens x 128 weight rows; wave w owns weight rows 32w..32w+31 against
//     ALL the tile's tokens, so every packed dword is dequantised exactly once per token block
//     (v_mfma_f32_32x32x16_f16: 4 MFMAs of MT token tiles per 16 packed bytes -> the 40 VALU of the dequantiser
//     ride under 16 * MT matrix instructions).
//   * Weights HBM -> VGPR, 16 B per lane (lane = weight row + 32 * k-group), 8 K steps (8 KiB per wave) in flight with
//     the refill-behind-use pipeline of gemm_stream.hip; no load sits behind a branch.
//   * Activations: [32 MT x 256 k] stages through LDS (coalesced 16-byte loads, chunks XOR-swizzled by the row), double
//     buffered: the loads of stage p+1 are issued before stage p is consumed.
//   * grid = (N / 128, ceil(M / (32 MT))); MT in {1, 2, 3, 4} chosen so that the token blocks cover M with little
//     padding (192 = 2 x 96, 512 = 4 x 128).
#include "common.cuh"
#include "kernels.h"
#include <stdlib.h>

namespace qspec {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f16x8 tdequant_s4x8(u32 p) { return dequant_s4x8_bitop(p); }   // common.cuh (k order 0,4,1,5,2,6,3,7)
__device__ __forceinline__ f16x8 tshuffle_act8(u32x4 a) {  // 8 consecutive fp16 -> order 0,4,1,5,2,6,3,7
    u32x4 o;
    o[0] = __builtin_amdgcn_perm(a[2], a[0], 0x05040100u);
    o[1] = __builtin_amdgcn_perm(a[2], a[0], 0x07060302u);
    o[2] = __builtin_amdgcn_perm(a[3], a[1], 0x05040100u);
    o[3] = __builtin_amdgcn_perm(a[3], a[1], 0x07060302u);
    return __builtin_bit_cast(f16x8, o);
}

#define QS_T_STAGE_K 128                    // k per activation stage = 2 weight steps of 64 k

// RING = weight steps (64 k each, 16 B per lane and weight tile) in flight per wave: 8 (K % 512 == 0) or 4 (K % 256 == 0).
// NT = 32-row weight tiles per wave (workgroup = 128 * NT weight rows): with 2, every activation fragment read from LDS
// feeds two MFMAs -- the LDS read per MFMA is what bounds the NT = 1 form at large M.
template <int MT, int RING, int NT>
__global__ __launch_bounds__(256) void gemm_w4a16_tiled_kernel(const f16* __restrict__ x, const uint8_t* __restrict__ wq,
                                                                const f16* __restrict__ ws, f16* __restrict__ out,
                                                                float* __restrict__ part, int M, int N, int K, int Ks) {
    constexpr int BM = 32 * MT;
    constexpr int ROWB = QS_T_STAGE_K * 2;                 // bytes of one token row in a stage (256)
    constexpr int CPR = ROWB / 16;                          // 16-byte chunks per row (16)
    constexpr int LPT = BM * CPR / 256;                     // activation chunks per thread and stage (2 * MT)
    constexpr int NS = RING / 2;                            // stages per ring round
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [2][BM][256 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kg = lane >> 5;
    const int n0 = blockIdx.x * (128 * NT) + wave * (32 * NT), m0 = blockIdx.y * BM;
    const int Kb = K >> 1;
    const int nrounds = Ks / (64 * RING);                  // this workgroup's K slice: [blockIdx.z * Ks, + Ks)
    const int k0 = blockIdx.z * Ks;
    const uint8_t* wrow = wq + (size_t)(n0 + nl) * Kb + (k0 >> 1) + kg * 16;     // step s: + 32 s bytes; tile nt: + 32 nt rows
    const size_t tstride = (size_t)32 * Kb;
    x += k0;

    // activation stage: coalesced 16-byte loads, stored k-shuffled (order of the dequantiser) and chunk-swizzled
    u32x4 areg[LPT];
    auto a_load = [&](int stage) {
#pragma unroll
        for (int i = 0; i < LPT; i++) {
            const int cidx = tid + i * 256, row = cidx / CPR, q = cidx % CPR;
            const int m = min(m0 + row, M - 1);           // rows beyond M repeat the last row (never stored)
            areg[i] = *reinterpret_cast<const u32x4*>(x + (size_t)m * K + (size_t)stage * QS_T_STAGE_K + q * 8);
        }
    };
    auto a_store = [&](int buf) {
        unsigned char* b = smem + (size_t)buf * BM * ROWB;
#pragma unroll
        for (int i = 0; i < LPT; i++) {
            const int cidx = tid + i * 256, row = cidx / CPR, q = cidx % CPR;
            *reinterpret_cast<f16x8*>(b + (size_t)row * ROWB + ((q ^ (row & 15)) << 4)) = tshuffle_act8(areg[i]);
        }
    };
    f32x16 acc[MT][NT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++)
#pragma unroll
            for (int i = 0; i < 16; i++) acc[mt][nt][i] = 0.0f;

    a_load(0);
    u32x4 w[RING][NT];
#pragma unroll
    for (int u = 0; u < RING; u++)
#pragma unroll
        for (int nt = 0; nt < NT; nt++) w[u][nt] = *reinterpret_cast<const u32x4*>(wrow + nt * tstride + (size_t)u * 32);
    a_store(0);
    __syncthreads();

    // one stage = 2 weight steps; lane (weight row nl, k-group kg), step j, dword dd <-> k = 64 j + 32 kg + 8 dd.
    // The activation fragment of token row mt * 32 + nl for (j, dd) is the 16-byte chunk (8 j + 4 kg + dd) ^ (nl & 15) of
    // that row: eight per-lane byte offsets computed ONCE (the stage buffer and the token tile are immediates of the
    // ds_read), and the fragments of sub-step i + 1 are requested BEFORE the matrix instructions of sub-step i: left to the
    // compiler, every read sat one MFMA in front of its use behind an `s_waitcnt lgkmcnt(1)` -- an LDS round trip (~100
    // cycles) exposed per 32-cycle MFMA, which is what held the kernel at 0.38 of the dense peak.
    int aoff[8];
#pragma unroll
    for (int i = 0; i < 8; i++) aoff[i] = nl * ROWB + ((((i >> 2) * 8 + kg * 4 + (i & 3)) ^ (nl & 15)) << 4);
    auto a_frag = [&](int buf, int mt, int i) -> f16x8 {
        return *reinterpret_cast<const f16x8*>(smem + aoff[i] + (buf * BM + mt * 32) * ROWB);
    };
    // A wave issues in order: an MFMA behind a busy matrix pipe blocks the VALU work queued after it, so a block of
    // dequantiser VALU followed by a block of MFMAs executes one AFTER the other (counters of that form, M = 2048:
    // 43 % of the wave cycles issuing, 42 % stalled at the issue, matrix pipe busy 37 %).  The dequantiser of sub-step
    // i + 1 is therefore INTERLEAVED with the MFMAs of sub-step i, a few VALU instructions behind each MFMA
    // (sched_group_barrier), and the fragments of i + 1 are requested in front of them.
    constexpr int NMF = MT * NT;                             // MFMAs per sub-step
    constexpr int VPM = (9 * NT + NMF - 1) / NMF;           // dequantiser VALU instructions to place behind each
    auto stage_compute = [&](int buf, const u32x4 (&w0)[NT], const u32x4 (&w1)[NT]) {
        f16x8 av[2][MT], bf[2][NT];
#pragma unroll
        for (int mt = 0; mt < MT; mt++) av[0][mt] = a_frag(buf, mt, 0);
#pragma unroll
        for (int nt = 0; nt < NT; nt++) bf[0][nt] = tdequant_s4x8(w0[nt][0]);
#pragma unroll
        for (int i = 0; i < 8; i++) {
            if (i + 1 < 8) {
#pragma unroll
                for (int mt = 0; mt < MT; mt++) av[(i + 1) & 1][mt] = a_frag(buf, mt, i + 1);
            }
            __builtin_amdgcn_sched_barrier(0);   // the next fragments are requested HERE; below: MFMA / VALU / MFMA / VALU ...
            if (i + 1 < 8) {
#pragma unroll
                for (int nt = 0; nt < NT; nt++) bf[(i + 1) & 1][nt] = tdequant_s4x8(i + 1 < 4 ? w0[nt][(i + 1) & 3] : w1[nt][(i + 1) & 3]);
            }
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int nt = 0; nt < NT; nt++)
                    acc[mt][nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i & 1][mt], bf[i & 1][nt], acc[mt][nt], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < NMF; m++) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                if (i + 1 < 8) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // the last round is peeled so that no weight refill sits behind a branch
    for (int r = 0; r < nrounds - 1; r++) {
        const uint8_t* wnext = wrow + (size_t)(r + 1) * RING * 32;
#pragma unroll
        for (int s = 0; s < NS; s++) {
            a_load(r * NS + s + 1);                       // next stage's activations fly under this stage's MFMAs
            stage_compute(s & 1, w[2 * s], w[2 * s + 1]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < NT; nt++) {
                w[2 * s][nt] = *reinterpret_cast<const u32x4*>(wnext + nt * tstride + (size_t)(2 * s) * 32);
                w[2 * s + 1][nt] = *reinterpret_cast<const u32x4*>(wnext + nt * tstride + (size_t)(2 * s + 1) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
            a_store((s & 1) ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int s = 0; s < NS; s++) {
        if (s + 1 < NS) a_load((nrounds - 1) * NS + s + 1);
        stage_compute(s & 1, w[2 * s], w[2 * s + 1]);
        if (s + 1 < NS) {
            a_store((s & 1) ^ 1);
            __syncthreads();
        }
    }

    // epilogue: 32x32 tile: lane holds column n = nl, rows (i / 4) * 8 + kg * 4 + (i % 4)
    if (part) {   // K slices: raw fp32 sums, [slice][M][N]; scaled and rounded by the finish launch (or the next norm)
        float* pz = part + (size_t)blockIdx.z * M * N;
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int nt = 0; nt < NT; nt++)
#pragma unroll
                for (int i = 0; i < 16; i++) {
                    const int m = m0 + mt * 32 + (i >> 2) * 8 + kg * 4 + (i & 3);
                    if (m < M) pz[(size_t)m * N + n0 + nt * 32 + nl] = acc[mt][nt][i];
                }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; nt++) {
        const float swn = h2f(ws[n0 + nt * 32 + nl]);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = m0 + mt * 32 + (i >> 2) * 8 + kg * 4 + (i & 3);
                if (m < M) out[(size_t)m * N + n0 + nt * 32 + nl] = f2h(acc[mt][nt][i] * swn);
            }
    }
}

bool gemm_w4a16_tiled_supported(int M, int N, int K) { return M >= 1 && N % 128 == 0 && K % 256 == 0 && K >= 512; }

static int env_int(const char* name, int dflt) {   // sweep knobs (scripts/sweep_tiled.sh): experimental build only
#ifdef QS_EXPERIMENTAL
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
#else
    (void)name;
    return dflt;
#endif
}

// Launch plan: token block 32 * MT and K slices S, by a small cost model fitted to a sweep on MI355X
// (scripts/sweep_tiled.sh; within ~10 % of the best plan on every swept shape).  Wide layers fill the chip with
// (N / 128) x token blocks; narrow ones (o_proj, down_proj at a few hundred tokens) would leave most CUs idle, so they
// take smaller token blocks and/or K slices whose raw sums go through `part` ([S][M][N] fp32) and one finish launch.
void gemm_w4a16_tiled_plan(int M, int N, int K, size_t part_bytes, int* MT_out, int* S_out) {
    static const int force_mt = env_int("QSPEC_TILED_MT", 0), force_s = env_int("QSPEC_TILED_S", 0);
    const int unit = K % 512 == 0 ? 512 : 256;            // one ring round
    double best = 1e30;
    int bmt = 1, bs = 1;
    for (int mt = 1; mt <= 4; mt++) {
        if (force_mt >= 1 && force_mt <= 4 && mt != force_mt) continue;
        const int mb = (M + 32 * mt - 1) / (32 * mt);
        for (int sl = 1; sl <= 8; sl++) {
            if (K % (sl * unit)) continue;
            if (sl > 1 && (size_t)sl * M * N * sizeof(float) > part_bytes) continue;
            if (force_s >= 1 && sl != force_s && K % (force_s * unit) == 0 &&
                (size_t)force_s * M * N * sizeof(float) <= part_bytes)
                continue;
            const double wgs = (double)(N / 128) * mb * sl;
            const double steps = (double)(K / sl / 64);
            const double rounds = (double)(((long long)wgs + 511) / 512);            // two workgroups per CU
            const double share = wgs > 256 ? 1.0 : 0.62;                              // a lone workgroup runs faster
            const double comp = rounds * steps * (0.06 + 0.265 * mt) * share;         // us
            const double traffic = ((double)N * K / 2 * mb) / 5.5e6;                  // weight re-reads through L2
            double t = (comp > traffic ? comp : traffic) + 4.0;
            if (sl > 1) t += 1.5 + (double)sl * M * N * 8 / 5e6;                      // partial sums out and back in
            if (t < best) { best = t; bmt = mt; bs = sl; }
        }
    }
    *MT_out = bmt;
    *S_out = bs;
}

static int tiled_launch(const f16* x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K, float* part, int MT, int S,
                        bool finish, hipStream_t st);
int gemm_w4a16_tiled(const f16* x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K, float* part,
                     size_t part_bytes, hipStream_t st) {
    if (!gemm_w4a16_tiled_supported(M, N, K)) return -1;
    int MT, S;
    gemm_w4a16_tiled_plan(M, N, K, part ? part_bytes : 0, &MT, &S);
    return tiled_launch(x, wq, ws, out, M, N, K, part, MT, S, true, st);
}
// The plan's K slices for a caller that finishes the raw sums itself (the verify pass: inside the norm that follows o_proj /
// down_proj, norm_quant.hip ln_fp16_partial -- the expression of gemm_w4a16_partial_finish, so the same bits, one launch less):
// slice count for (M, N, K), 0 = the plan does not slice this shape; then the launch that leaves part [S][M][N].
int gemm_w4a16_tiled_partial_slices(int M, int N, int K) {
    if (!gemm_w4a16_tiled_supported(M, N, K)) return 0;
    int MT, S;
    gemm_w4a16_tiled_plan(M, N, K, gemm_w4a16_ws_bytes() - 8192, &MT, &S);   // the cap of qspec_w4a16_linear's own workspace: the same plan
    return S > 1 ? S : 0;
}
int gemm_w4a16_tiled_partial(const f16* x, const int8_t* wq, float* part, int M, int N, int K, int S, hipStream_t st) {
    if (!gemm_w4a16_tiled_supported(M, N, K) || !part) return -1;
    int MT, PS;
    gemm_w4a16_tiled_plan(M, N, K, gemm_w4a16_ws_bytes() - 8192, &MT, &PS);
    if (PS != S || S < 2) return -1;
    return tiled_launch(x, wq, nullptr, nullptr, M, N, K, part, MT, S, false, st);
}
static int tiled_launch(const f16* x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K, float* part, int MT, int S,
                        bool finish, hipStream_t st) {
    // two weight tiles per wave when that still leaves two workgroups per CU (large M): halves the LDS reads per MFMA
    static const int force_nt = env_int("QSPEC_TILED_NT", 0);
    const int mblocks = (M + 32 * MT - 1) / (32 * MT);
    // (round 3, with the interleaved inner loop: two tiles per wave win from ~200 workgroups on -- gate_up at M = 192:
    // 50.9 against 54.2 us, at 512: 126.6 against 132.7 -- not only from two workgroups per CU)
    int NT = (N % 256 == 0 && (long long)(N / 256) * mblocks * S >= 192) ? 2 : 1;
    if (force_nt == 1 || (force_nt == 2 && N % 256 == 0)) NT = force_nt;
    const dim3 grid(N / (128 * NT), mblocks, S);
    const size_t lds = (size_t)2 * 32 * MT * (QS_T_STAGE_K * 2);   // <= 64 KiB
    const uint8_t* w8 = reinterpret_cast<const uint8_t*>(wq);
    const bool r8 = K % 512 == 0 && NT == 1;   // two tiles per wave: 4 steps in flight (the registers go to the accumulators)
    float* p = S > 1 ? part : nullptr;
    const int Ks = K / S;
#define QS_TILED(MTV)                                                                                                   \
    case MTV:                                                                                                           \
        if (NT == 2)                                                                                                    \
            hipLaunchKernelGGL((gemm_w4a16_tiled_kernel<MTV, 4, 2>), grid, dim3(256), lds, st, x, w8, ws, out, p, M, N, \
                               K, Ks);                                                                                  \
        else if (r8)                                                                                                    \
            hipLaunchKernelGGL((gemm_w4a16_tiled_kernel<MTV, 8, 1>), grid, dim3(256), lds, st, x, w8, ws, out, p, M, N, \
                               K, Ks);                                                                                  \
        else                                                                                                            \
            hipLaunchKernelGGL((gemm_w4a16_tiled_kernel<MTV, 4, 1>), grid, dim3(256), lds, st, x, w8, ws, out, p, M, N, \
                               K, Ks);                                                                                  \
        break;
    switch (MT) {
        QS_TILED(1) QS_TILED(2) QS_TILED(3) QS_TILED(4)
        default: return -1;
    }
#undef QS_TILED
    if (S > 1 && finish) return gemm_w4a16_partial_finish(part, ws, out, M, N, S, st);
    return 0;
}

// ------------------------------------------------------------------ fp16 x fp16 (lm_head) at prefill-sized M
// out[m,n] = h( sum_k f(x[m,k]) * f(w[n,k]) ), fp32 accumulate (LogitsProcessor's plain nn.Linear,
// vllm/model_executor/layers/logits_processor.py:92-97).  Same tiling as above; the weight fragment of
// v_mfma_f32_32x32x16_f16 is 8 consecutive halves of a row, i.e. the 16-byte load itself (no dequantiser).
// One weight step = 16 k (32 bytes of a row, 16 per k-group); a stage of 128 k = 8 steps; 16 steps (32 KiB per
// wave... 16 B x 64 lanes x 16) in flight.  blockIdx.x walks the token blocks so that workgroups sharing a weight tile
// are dispatched together (the second reader hits the infinity cache).  N % 32 == 0; rows past N are clamped, not stored.
template <int MT>
__global__ __launch_bounds__(256) void gemm_f16_tiled_kernel(const f16* __restrict__ x, const f16* __restrict__ wt,
                                                              f16* __restrict__ out, int M, int N, int K) {
    constexpr int BM = 32 * MT, ROWB = QS_T_STAGE_K * 2, CPR = ROWB / 16, LPT = BM * CPR / 256;
    constexpr int RING = 16, SPS = 8, NS = RING / SPS;      // steps in flight, steps per stage, stages per round
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nl = lane & 31, kg = lane >> 5;
    const int n0 = blockIdx.y * 128 + wave * 32, m0 = blockIdx.x * BM;
    const int nrow = min(n0 + nl, N - 1);
    const int nrounds = K / (16 * RING);
    const unsigned char* wrow = reinterpret_cast<const unsigned char*>(wt + (size_t)nrow * K) + kg * 16;  // step s: + 32 s

    u32x4 areg[LPT];
    auto a_load = [&](int stage) {
#pragma unroll
        for (int i = 0; i < LPT; i++) {
            const int cidx = tid + i * 256, row = cidx / CPR, q = cidx % CPR;
            const int m = min(m0 + row, M - 1);
            areg[i] = *reinterpret_cast<const u32x4*>(x + (size_t)m * K + (size_t)stage * QS_T_STAGE_K + q * 8);
        }
    };
    auto a_store = [&](int buf) {
        unsigned char* b = smem + (size_t)buf * BM * ROWB;
#pragma unroll
        for (int i = 0; i < LPT; i++) {
            const int cidx = tid + i * 256, row = cidx / CPR, q = cidx % CPR;
            *reinterpret_cast<u32x4*>(b + (size_t)row * ROWB + ((q ^ (row & 15)) << 4)) = areg[i];
        }
    };
    f32x16 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 16; i++) acc[mt][i] = 0.0f;
    a_load(0);
    u32x4 w[RING];
#pragma unroll
    for (int u = 0; u < RING; u++) w[u] = *reinterpret_cast<const u32x4*>(wrow + (size_t)u * 32);
    a_store(0);
    __syncthreads();
    auto step_compute = [&](int buf, int j, const u32x4& wv) {   // step j of the stage: k = 16 j + 8 kg .. + 7 = chunk 2 j + kg
        const unsigned char* b = smem + (size_t)buf * BM * ROWB;
        const f16x8 bf = __builtin_bit_cast(f16x8, wv);
        const int q = 2 * j + kg;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            const int row = mt * 32 + nl;
            const f16x8 av = *reinterpret_cast<const f16x8*>(b + (size_t)row * ROWB + ((q ^ (row & 15)) << 4));
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bf, acc[mt], 0, 0, 0);
        }
    };
    for (int r = 0; r < nrounds - 1; r++) {
        const unsigned char* wnext = wrow + (size_t)(r + 1) * RING * 32;
#pragma unroll
        for (int sg = 0; sg < NS; sg++) {
            a_load(r * NS + sg + 1);
#pragma unroll
            for (int j = 0; j < SPS; j++) {
                step_compute(sg & 1, j, w[sg * SPS + j]);
                __builtin_amdgcn_sched_barrier(0);
                w[sg * SPS + j] = *reinterpret_cast<const u32x4*>(wnext + (size_t)(sg * SPS + j) * 32);
                __builtin_amdgcn_sched_barrier(0);
            }
            a_store((sg & 1) ^ 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int sg = 0; sg < NS; sg++) {
        if (sg + 1 < NS) a_load((nrounds - 1) * NS + sg + 1);
#pragma unroll
        for (int j = 0; j < SPS; j++) step_compute(sg & 1, j, w[sg * SPS + j]);
        if (sg + 1 < NS) {
            a_store((sg & 1) ^ 1);
            __syncthreads();
        }
    }
    if (n0 + nl < N) {
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int i = 0; i < 16; i++) {
                const int m = m0 + mt * 32 + (i >> 2) * 8 + kg * 4 + (i & 3);
                if (m < M) out[(size_t)m * N + n0 + nl] = f2h(acc[mt][i]);
            }
    }
}

bool gemm_f16_tiled_supported(int M, int N, int K) { return M >= 1 && N % 32 == 0 && K % 256 == 0 && K >= 512; }

int gemm_f16_tiled(const f16* x, const f16* w, f16* out, int M, int N, int K, hipStream_t st) {
    if (!gemm_f16_tiled_supported(M, N, K)) return -1;
    const int blocks = (M + 127) / 128;
    const int per = (M + blocks - 1) / blocks;
    const int MT = (per + 31) / 32;   // 1..4: fewest token blocks, least padding (192 -> 2 x 96): the weights stream once per block
    const dim3 grid((M + 32 * MT - 1) / (32 * MT), (N + 127) / 128);
    const size_t lds = (size_t)2 * 32 * MT * (QS_T_STAGE_K * 2);
    switch (MT) {
        case 1: hipLaunchKernelGGL((gemm_f16_tiled_kernel<1>), grid, dim3(256), lds, st, x, w, out, M, N, K); break;
        case 2: hipLaunchKernelGGL((gemm_f16_tiled_kernel<2>), grid, dim3(256), lds, st, x, w, out, M, N, K); break;
        case 3: hipLaunchKernelGGL((gemm_f16_tiled_kernel<3>), grid, dim3(256), lds, st, x, w, out, M, N, K); break;
        case 4: hipLaunchKernelGGL((gemm_f16_tiled_kernel<4>), grid, dim3(256), lds, st, x, w, out, M, N, K); break;
        default: return -1;
    }
    return 0;
}

}  // namespace qspec
