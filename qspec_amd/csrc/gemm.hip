// Skinny GEMMs of the draft/verify path over ONE shared packed-int4 weight buffer.
//
// Replaces (reference, relative to /root/reference):
//   rowwise_scaled_linear_kernel_cutlass_sm8x_unified<s4,s4>
//       third-party/ao/torchao/csrc/cuda/rowwise_scaled_linear_cutlass/rowwise_scaled_linear_cutlass_unified.cuh:240-488
//       (Python: vllm/model_executor/layers/quarot_nn/linear.py:67-84)            -> gemm_w4a4
//   bitblas.Matmul (W4A16, per-channel scale, `weight ^ 0x88` copy each call)
//       quarot_nn/linear.py:102-124,156-211                                       -> gemm_w4a16
//   lm_head nn.Linear (vllm/model_executor/layers/logits_processor.py:92-97)      -> gemm_f16
//
// Weight layout is the checkpoint's own: W[N, K/2] bytes, byte j of a row =
// (w[2j] & 0xF) | (w[2j+1] << 4), two's complement.  Both kernels read those
// bytes directly ("two views of one buffer"): the draft view widens nibbles to
// int8*16 with two bit-ops per dword and feeds v_mfma_i32_16x16x64_i8; the
// verify view flips the sign bit (offset binary, what the reference does with
// a full-weight XOR pass per call), splices nibbles into fp16 mantissas and
// feeds v_mfma_f32_16x16x32_f16.  No repacking, no second copy.
//
// Decode shapes (M <= 64 tokens) are pure weight streaming, so:
//   * one workgroup = one 16-row weight tile, its 4 waves interleave over K in
//     64-byte steps -> the 4 waves together read 256 contiguous bytes of each
//     of the 16 rows per trip, every byte of W is fetched exactly once;
//   * weights go HBM -> VGPR directly (16 B per lane, deep unroll); there is
//     no reuse inside a workgroup so an LDS round trip would be pure overhead;
//   * the MFMA contraction index may be permuted freely as long as A and B
//     fragments use the same permutation, so nibbles are never re-ordered:
//     activation fragments are loaded with the same per-lane byte offsets and
//     pushed through the same bit-ops;
//   * cross-wave (split-K) reduction goes through 4 KB of LDS in a fixed
//     order -> deterministic, no atomics, epilogue fused.
#include "common.cuh"
#include "kernels.h"

namespace qspec {

#define QS_UNROLL_K 4

// Two packed dwords (16 nibbles) -> one MFMA i8 operand (16 int8, each = nibble*16):
// even nibbles land in the high half of each byte via (p << 4), odd nibbles are already there.
__device__ __forceinline__ i32x4 widen_s4x16(u32 p0, u32 p1) {
    return i32x4{(int)((p0 << 4) & 0xF0F0F0F0u), (int)(p0 & 0xF0F0F0F0u), (int)((p1 << 4) & 0xF0F0F0F0u),
                 (int)(p1 & 0xF0F0F0F0u)};
}

// ------------------------------------------------------------------ fused epilogues
// The 16 weight rows of a workgroup need not be contiguous.  Choosing them as 8 rows + the 8 rows that pair with
// them lets the GEMM finish what the reference does in separate kernels right after it:
//   EPI_QKV    rows = channels {8j..8j+7} and {64+8j..} of one head: NeoX RoPE pairs (i, i+64) meet in one
//              workgroup -> rotate q,k (csrc/pos_encoding_kernels.cu:10-35) and scatter k,v into the paged
//              cache (csrc/cache_kernels.cu:207-247) from the epilogue; the rope + cache-write kernel disappears.
//   EPI_GATEUP rows = 8 `up` channels and their 8 `gate` channels of the fused gate_up weight:
//              out[m, c] = h(h(silu(gate)) * up)   (quarot_llama.py:279-284), half the output bytes.
// Rounding points are unchanged: the GEMM result is rounded to fp16 exactly where the reference's op returns
// fp16, and only then rotated / activated.
enum { EPI_PLAIN = 0, EPI_QKV = 1, EPI_GATEUP = 2, EPI_RAW = 3 };   // RAW (W4A16 only): unscaled fp32 sums, out viewed as float [M, N]

struct EpiArgs {
    const int64_t* positions;   // [M]
    const f16* cos_sin_cache;   // [max_pos, 128]
    f16* key_cache;             // [slots, nkv, 128]
    f16* value_cache;
    const int64_t* slot_mapping;  // [M]
    int nq, nkv;
    int I;                      // intermediate size (EPI_GATEUP)
    int tb0;                    // first tile of this launch (tensor-parallel column shards launch a sub-range)
    int64_t ldw;                // bytes between packed weight rows (0 = K/2): K-sliced views of the shared buffer
    int64_t ldx;                // halves between fp16 activation rows (0 = K)
};

template <int EPI>
__device__ __forceinline__ int tile_row(int tb, int r, const EpiArgs& ea) {
    if (EPI == EPI_QKV) return (tb >> 3) * 128 + (r >> 3) * 64 + (tb & 7) * 8 + (r & 7);
    if (EPI == EPI_GATEUP) return (r >> 3) * ea.I + tb * 8 + (r & 7);
    return tb * 16 + r;
}

// hv = the fp16 GEMM result of element (m, tile column c); ex = 256 halves of LDS per m-tile for the pair exchange.
template <int EPI>
__device__ __forceinline__ void epilogue_store(f16 hv, bool valid, int m, int c, int tb, int N, f16* __restrict__ out,
                                               f16* ex, int ex_idx, const EpiArgs& ea) {
    if (EPI == EPI_PLAIN) {
        if (valid) out[(size_t)m * N + tb * 16 + c] = hv;
        return;
    }
    ex[ex_idx] = hv;
    __syncthreads();
    const f16 partner = ex[ex_idx ^ 8];
    __syncthreads();
    if (!valid) return;
    if (EPI == EPI_GATEUP) {
        if (c < 8) {  // hv = up, partner = gate
            const float g = h2f(partner);
            const float a = h2f(f2h(g / (1.0f + qexpf(-g))));
            out[(size_t)m * ea.I + tb * 8 + c] = f2h(a * h2f(hv));
        }
        return;
    }
    // EPI_QKV
    const int head = tb >> 3, o = (tb & 7) * 8 + (c & 7);  // o = index inside the first half of the head
    const int n = head * 128 + (c >> 3) * 64 + o;
    f16 res = hv;
    if (head < ea.nq + ea.nkv) {
        const f16* cs = ea.cos_sin_cache + ea.positions[m] * 128;
        const float cf = h2f(cs[o]), sf = h2f(cs[64 + o]);
        const float xf = h2f(c < 8 ? hv : partner), yf = h2f(c < 8 ? partner : hv);
        res = c < 8 ? f2h(h2f(f2h(xf * cf)) - h2f(f2h(yf * sf))) : f2h(h2f(f2h(yf * cf)) + h2f(f2h(xf * sf)));
    }
    out[(size_t)m * N + n] = res;
    if (head >= ea.nq) {
        const int64_t slot = ea.slot_mapping[m];
        if (slot >= 0) {
            const bool is_k = head < ea.nq + ea.nkv;
            const int kvh = is_k ? head - ea.nq : head - ea.nq - ea.nkv;
            f16* cache = is_k ? ea.key_cache : ea.value_cache;
            cache[(slot * ea.nkv + kvh) * 128 + (c >> 3) * 64 + o] = res;
        }
    }
}

// ------------------------------------------------------------------ W4A4
// out[m,n] = h( (f(acc[m,n]) * f(sa[m])) * f(sw[n]) (+ f(bias[n])) ),  acc = sum_k a[m,k] w[n,k]  (int32, exact)
template <int MT, int EPI>
__global__ __launch_bounds__(256) void gemm_w4a4_kernel(const int8_t* __restrict__ xq, const f16* __restrict__ xs,
                                                         const int8_t* __restrict__ wq, const f16* __restrict__ ws,
                                                         const f16* __restrict__ bias, f16* __restrict__ out, int M,
                                                         int N, int K, int m_base, EpiArgs ea) {
    __shared__ int red[4][MT][256];
    __shared__ f16 ex[MT][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int tb = blockIdx.x + ea.tb0;
    const int Kb = K >> 1;
    const int nsteps = Kb >> 6;  // 64 bytes of every row per step
    const size_t ldw = ea.ldw ? (size_t)ea.ldw : (size_t)Kb;
    const uint8_t* wrow = reinterpret_cast<const uint8_t*>(wq) + (size_t)tile_row<EPI>(tb, r, ea) * ldw + g * 16;
    const uint8_t* arow[MT];
    bool aval[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int m = m_base + mt * 16 + r;
        aval[mt] = m < M;
        arow[mt] = reinterpret_cast<const uint8_t*>(xq) + (size_t)(aval[mt] ? m : 0) * Kb + g * 16;
    }
    i32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = i32x4{0, 0, 0, 0};

    int s = wave;
    for (; s + 4 * (QS_UNROLL_K - 1) < nsteps; s += 4 * QS_UNROLL_K) {
        u32x4 w[QS_UNROLL_K];
        u32x4 a[QS_UNROLL_K][MT];
#pragma unroll
        for (int u = 0; u < QS_UNROLL_K; u++) {
            w[u] = *reinterpret_cast<const u32x4*>(wrow + (size_t)(s + 4 * u) * 64);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                a[u][mt] = u32x4{0, 0, 0, 0};
                if (aval[mt]) a[u][mt] = *reinterpret_cast<const u32x4*>(arow[mt] + (size_t)(s + 4 * u) * 64);
            }
        }
#pragma unroll
        for (int u = 0; u < QS_UNROLL_K; u++) {
            const i32x4 b0 = widen_s4x16(w[u][0], w[u][1]), b1 = widen_s4x16(w[u][2], w[u][3]);
#pragma unroll
            for (int mt = 0; mt < MT; mt++) {
                const i32x4 a0 = widen_s4x16(a[u][mt][0], a[u][mt][1]), a1 = widen_s4x16(a[u][mt][2], a[u][mt][3]);
                acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc[mt], 0, 0, 0);
                acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc[mt], 0, 0, 0);
            }
        }
    }
    for (; s < nsteps; s += 4) {
        u32x4 w = *reinterpret_cast<const u32x4*>(wrow + (size_t)s * 64);
        const i32x4 b0 = widen_s4x16(w[0], w[1]), b1 = widen_s4x16(w[2], w[3]);
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            u32x4 a = u32x4{0, 0, 0, 0};
            if (aval[mt]) a = *reinterpret_cast<const u32x4*>(arow[mt] + (size_t)s * 64);
            const i32x4 a0 = widen_s4x16(a[0], a[1]), a1 = widen_s4x16(a[2], a[3]);
            acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a0, b0, acc[mt], 0, 0, 0);
            acc[mt] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a1, b1, acc[mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) red[wave][mt][i * 64 + lane] = acc[mt][i];
    __syncthreads();
    // thread t owns accumulator element (reg = t>>6, lane = t&63): row = 4*(lane>>4)+reg, col = lane&15
    const int t = threadIdx.x, el = t & 63, reg = t >> 6;
    const int c = el & 15;
    const int n = tile_row<EPI>(tb, c, ea);
    const float swn = EPI == EPI_RAW ? 1.0f : h2f(ws[n]);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int row16 = 4 * (el >> 4) + reg;
        const int m = m_base + mt * 16 + row16;
        const bool valid = m < M;
        f16 hv = (f16)0.0f;
        if (valid) {
            int sum = red[0][mt][t] + red[1][mt][t] + red[2][mt][t] + red[3][mt][t];
            float v = ((float)(sum >> 8) * h2f(xs[m])) * swn;  // both operands carried a factor 16
            if (bias) v = v + h2f(bias[n]);
            hv = f2h(v);
        }
        epilogue_store<EPI>(hv, valid, m, c, tb, N, out, &ex[mt][0], row16 * 16 + c, ea);
    }
}

template <int EPI>
static int launch_w4a4(const int8_t* xq, const f16* xs, const int8_t* wq, const f16* ws, const f16* bias, f16* out,
                       int M, int N, int K, const EpiArgs& ea, hipStream_t st) {
    // up to 64 rows per pass over the weights; larger M re-streams W (prefill-sized M is not this kernel's job)
    for (int mb = 0; mb < M; mb += 64) {
        int rows = M - mb;
        if (rows <= 16)
            hipLaunchKernelGGL((gemm_w4a4_kernel<1, EPI>), dim3(N / 16), dim3(256), 0, st, xq, xs, wq, ws, bias, out, M, N, K, mb, ea);
        else if (rows <= 32)
            hipLaunchKernelGGL((gemm_w4a4_kernel<2, EPI>), dim3(N / 16), dim3(256), 0, st, xq, xs, wq, ws, bias, out, M, N, K, mb, ea);
        else
            hipLaunchKernelGGL((gemm_w4a4_kernel<4, EPI>), dim3(N / 16), dim3(256), 0, st, xq, xs, wq, ws, bias, out, M, N, K, mb, ea);
    }
    return 0;
}

int gemm_w4a4(const int8_t* xq, const f16* xs, const int8_t* wq, const f16* ws, const f16* bias, f16* out, int M,
              int N, int K, hipStream_t st) {
    if (M == 0 || N == 0) return 0;
    if (N % 16 || K % 128 || K > (1 << 19)) return -1;
    return launch_w4a4<EPI_PLAIN>(xq, xs, wq, ws, bias, out, M, N, K, EpiArgs{}, st);
}

static int check_qkv(int N, int K, int nq, int nkv, int d, int rot_dim) {
    if (d != 128 || rot_dim != 128 || N != (nq + 2 * nkv) * 128 || K % 128) return -1;
    return 0;
}

int gemm_w4a4_qkv_rope(const int8_t* xq, const f16* xs, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                       const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                       const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st) {
    if (M == 0) return 0;
    if (check_qkv(N, K, nq, nkv, d, rot_dim)) return -1;
    EpiArgs ea{positions, cos_sin_cache, key_cache, value_cache, slot_mapping, nq, nkv, 0};
    return launch_w4a4<EPI_QKV>(xq, xs, wq, ws, nullptr, qkv, M, N, K, ea, st);
}

int gemm_w4a4_gate_up_silu(const int8_t* xq, const f16* xs, const int8_t* wq, const f16* ws, f16* act, int M, int I,
                           int K, hipStream_t st) {
    if (M == 0) return 0;
    if (I % 8 || K % 128) return -1;
    EpiArgs ea{};
    ea.I = I;
    return launch_w4a4<EPI_GATEUP>(xq, xs, wq, ws, nullptr, act, M, 2 * I, K, ea, st);
}

// ------------------------------------------------------------------ W4A16
// One dword = 8 nibbles n0..n7.  After p ^= 0x88888888 each nibble is u = w + 8.
//   (p & 0x000F000F) | 0x64006400  -> halves (1024+u0, 1024+u4);  - 1032        -> (w0, w4)
//   (p & 0x00F000F0) | 0x64006400  -> halves (1024+16u1, 1024+16u5); *1/16 - 72 -> (w1, w5)
//   same on p >> 8                 -> (w2, w6), (w3, w7)
// so one B fragment (8 fp16) holds k-order 0,4,1,5,2,6,3,7 of its dword; the
// activation fragment is shuffled into the same order with 4 v_perm_b32.
__device__ __forceinline__ f16x8 dequant_s4x8(u32 p) { return dequant_s4x8_bitop(p); }   // common.cuh (k order 0,4,1,5,2,6,3,7)
// 8 consecutive fp16 (4 dwords) -> order 0,4,1,5,2,6,3,7
__device__ __forceinline__ f16x8 shuffle_act8(u32x4 a) {
    u32x4 o;
    o[0] = __builtin_amdgcn_perm(a[2], a[0], 0x05040100u);
    o[1] = __builtin_amdgcn_perm(a[2], a[0], 0x07060302u);
    o[2] = __builtin_amdgcn_perm(a[3], a[1], 0x05040100u);
    o[3] = __builtin_amdgcn_perm(a[3], a[1], 0x07060302u);
    return __builtin_bit_cast(f16x8, o);
}

// out[m,n] = h( (sum_k f(x[m,k]) * w[n,k]) * f(sw[n]) (+ f(bias[n])) ), fp32 accumulate
// Activations are 4x the bytes of the weight rows they multiply (fp16 vs int4), and as MFMA A fragments they
// would be 64 scattered 16-byte pieces per load instruction (the address unit, not bandwidth, then bounds the
// kernel: 4x slower measured).  So each round the workgroup copies the [16*MT rows x 512 k] activation tile it needs
// into LDS with fully coalesced loads (one 1 KiB row segment per wave instruction, double buffered, one barrier
// per round) and the waves take their fragments from there; 16-byte chunks are XOR-swizzled by the row so a
// ds_read_b128 wave access sweeps all 64 banks exactly 4 times (the minimum).
template <int MT, int EPI>
__global__ __launch_bounds__(256) void gemm_w4a16_kernel(const f16* __restrict__ x, const int8_t* __restrict__ wq,
                                                          const f16* __restrict__ ws, const f16* __restrict__ bias,
                                                          f16* __restrict__ out, int M, int N, int K, int m_base,
                                                          EpiArgs ea) {
    constexpr int ROWS = 16 * MT;
    __shared__ __attribute__((aligned(16))) unsigned char atile[2][ROWS * 1024];  // [buf][row][1024 B = 4 steps]
    __shared__ float red[4][MT][256];
    __shared__ f16 ex[MT][256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int tb = blockIdx.x + ea.tb0;
    const int Kb = K >> 1;
    const int nsteps = Kb >> 6;              // 64 packed bytes (128 k) of every weight row per step
    const int nrounds = (nsteps + 3) >> 2;   // 4 steps (one per wave) per round
    const size_t ldw = ea.ldw ? (size_t)ea.ldw : (size_t)Kb, ldx = ea.ldx ? (size_t)ea.ldx : (size_t)K;
    const uint8_t* wrow = reinterpret_cast<const uint8_t*>(wq) + (size_t)tile_row<EPI>(tb, r, ea) * ldw + g * 16;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // cooperative A loader: chunk id c = tid + 256*j covers ROWS rows x 64 chunks of 16 B
    constexpr int CPT = ROWS * 64 / 256;  // chunks per thread (4 * MT)
    u32x4 areg[CPT];
    auto load_a = [&](int round) {
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int c = tid + 256 * j, row = c >> 6, q = c & 63;
            const int m = m_base + row;
            const size_t koff = (size_t)round * 512 + q * 8;  // halves
            areg[j] = u32x4{0, 0, 0, 0};
            if (m < M && koff < (size_t)K) areg[j] = *reinterpret_cast<const u32x4*>(x + (size_t)m * ldx + koff);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int c = tid + 256 * j, row = c >> 6, q = c & 63;
            *reinterpret_cast<u32x4*>(&atile[buf][row * 1024 + ((q ^ (row & 15)) << 4)]) = areg[j];
        }
    };
    load_a(0);
    u32x4 wcur = u32x4{0, 0, 0, 0};
    if (wave < nsteps) wcur = *reinterpret_cast<const u32x4*>(wrow + (size_t)wave * 64);
    for (int rd = 0; rd < nrounds; rd++) {
        const int buf = rd & 1;
        store_a(buf);
        __syncthreads();
        const int s = rd * 4 + wave;
        u32x4 wnext = u32x4{0, 0, 0, 0};
        if (rd + 1 < nrounds) {
            load_a(rd + 1);
            if (s + 4 < nsteps) wnext = *reinterpret_cast<const u32x4*>(wrow + (size_t)(s + 4) * 64);
        }
        if (s < nsteps) {
#pragma unroll
            for (int dd = 0; dd < 4; dd++) {
                const f16x8 b = dequant_s4x8(wcur[dd]);
                const int q = wave * 16 + g * 4 + dd;  // 16-byte chunk of this lane's 8 k values inside the row
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int row = mt * 16 + r;
                    const u32x4 a = *reinterpret_cast<const u32x4*>(&atile[buf][row * 1024 + ((q ^ (row & 15)) << 4)]);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(shuffle_act8(a), b, acc[mt], 0, 0, 0);
                }
            }
        }
        wcur = wnext;
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) red[wave][mt][i * 64 + lane] = acc[mt][i];
    __syncthreads();
    const int t = threadIdx.x, el = t & 63, reg = t >> 6;
    const int c = el & 15;
    const int n = tile_row<EPI>(tb, c, ea);
    const float swn = EPI == EPI_RAW ? 1.0f : h2f(ws[n]);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        const int row16 = 4 * (el >> 4) + reg;
        const int m = m_base + mt * 16 + row16;
        const bool valid = m < M;
        f16 hv = (f16)0.0f;
        if (valid) {
            float sum = ((red[0][mt][t] + red[1][mt][t]) + red[2][mt][t]) + red[3][mt][t];
            if (EPI == EPI_RAW) {
                reinterpret_cast<float*>(out)[(size_t)m * N + tb * 16 + c] = sum;
                continue;
            }
            float v = sum * swn;
            if (bias) v = v + h2f(bias[n]);
            hv = f2h(v);
        }
        if (EPI == EPI_RAW) continue;
        epilogue_store<EPI>(hv, valid, m, c, tb, N, out, &ex[mt][0], row16 * 16 + c, ea);
    }
}

// ---- W4A16, decode-sized M, 2-D decomposition (the kernel the verify pass uses).
// A workgroup owns 64 weight rows (4 tiles, one per wave) x one K block (grid.y).  The 4 waves share every
// activation tile through LDS, so activation bytes per workgroup equal weight bytes (a 16-row tile alone needs 4x its
// weight bytes of activations at M = 16), and K blocks keep >= ~256 workgroups in flight for the narrow layers.
// K blocks are combined by the last workgroup to arrive: fp32 partial tiles in the workspace, summed in block
// order (deterministic), agent-scope release/acquire around a ticket counter that every call leaves at zero.
#define QS_W16_CNT_SLOTS 2048
#define QS_W16_PART_BYTES (32u * 1024u * 1024u)
template <int MT, int EPI>
__global__ __launch_bounds__(256) void gemm_w4a16_2d_kernel(const f16* __restrict__ x, const int8_t* __restrict__ wq,
                                                             const f16* __restrict__ ws, const f16* __restrict__ bias,
                                                             f16* __restrict__ out, int M, int N, int K,
                                                             int rounds_per, int* cnt, float* part, EpiArgs ea) {
    constexpr int ROWS = 16 * MT;
    __shared__ __attribute__((aligned(16))) unsigned char atile[2][ROWS * 1024];  // [buf][row][512 k fp16]
    __shared__ f16 ex[4][MT][256];
    __shared__ int flag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int nb = blockIdx.x, S = gridDim.y, split = blockIdx.y;
    const int tb = nb * 4 + wave + ea.tb0;
    const int Kb = K >> 1;
    const int nsteps = Kb >> 6;
    const int nrounds = (nsteps + 3) >> 2;
    const int rd_lo = split * rounds_per, rd_hi = min(nrounds, rd_lo + rounds_per);
    const size_t ldw = ea.ldw ? (size_t)ea.ldw : (size_t)Kb, ldx = ea.ldx ? (size_t)ea.ldx : (size_t)K;
    const uint8_t* wrow = reinterpret_cast<const uint8_t*>(wq) + (size_t)tile_row<EPI>(tb, r, ea) * ldw + g * 16;
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    constexpr int CPT = ROWS * 64 / 256;
    u32x4 areg[CPT];
    auto load_a = [&](int round) {
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int c = tid + 256 * j, row = c >> 6, q = c & 63;
            const size_t koff = (size_t)round * 512 + q * 8;
            areg[j] = u32x4{0, 0, 0, 0};
            if (row < M && koff < (size_t)K) areg[j] = *reinterpret_cast<const u32x4*>(x + (size_t)row * ldx + koff);
        }
    };
    auto store_a = [&](int buf) {
#pragma unroll
        for (int j = 0; j < CPT; j++) {
            const int c = tid + 256 * j, row = c >> 6, q = c & 63;
            *reinterpret_cast<u32x4*>(&atile[buf][row * 1024 + ((q ^ (row & 15)) << 4)]) = areg[j];
        }
    };
    u32x4 wcur[4], wnext[4];
    auto load_w = [&](u32x4* w, int round) {
#pragma unroll
        for (int u = 0; u < 4; u++) {
            const int s = round * 4 + u;
            w[u] = u32x4{0, 0, 0, 0};
            if (s < nsteps) w[u] = *reinterpret_cast<const u32x4*>(wrow + (size_t)s * 64);
        }
    };
    if (rd_lo < rd_hi) {
        load_a(rd_lo);
        load_w(wcur, rd_lo);
    }
    for (int rd = rd_lo; rd < rd_hi; rd++) {
        const int buf = (rd - rd_lo) & 1;
        store_a(buf);
        __syncthreads();
        if (rd + 1 < rd_hi) {
            load_a(rd + 1);
            load_w(wnext, rd + 1);
        }
#pragma unroll
        for (int u = 0; u < 4; u++)
#pragma unroll
            for (int dd = 0; dd < 4; dd++) {
                const f16x8 b = dequant_s4x8(wcur[u][dd]);
                const int q = u * 16 + g * 4 + dd;
#pragma unroll
                for (int mt = 0; mt < MT; mt++) {
                    const int row = mt * 16 + r;
                    const u32x4 a = *reinterpret_cast<const u32x4*>(&atile[buf][row * 1024 + ((q ^ (row & 15)) << 4)]);
                    acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(shuffle_act8(a), b, acc[mt], 0, 0, 0);
                }
            }
#pragma unroll
        for (int u = 0; u < 4; u++) wcur[u] = wnext[u];
    }
    if (S > 1) {
        float* my = part + (((size_t)split * gridDim.x + nb) * 4 + wave) * (MT * 256);
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) my[mt * 256 + reg * 64 + lane] = acc[mt][reg];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const int ticket = __hip_atomic_fetch_add(cnt + nb, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            flag = ticket == S - 1;
        }
        __syncthreads();
        if (!flag) return;
        if (tid == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(cnt + nb, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        float pv[8][MT][4];  // up to 8 K blocks: all partial loads in flight before the first add
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++)
#pragma unroll
            for (int mt = 0; mt < MT; mt++)
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {
                    pv[s2][mt][reg] = 0.0f;
                    if (s2 < S)
                        pv[s2][mt][reg] = part[(((size_t)s2 * gridDim.x + nb) * 4 + wave) * (MT * 256) + mt * 256 + reg * 64 + lane];
                }
#pragma unroll
        for (int mt = 0; mt < MT; mt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                float sum = pv[0][mt][reg];
#pragma unroll
                for (int s2 = 1; s2 < 8; s2++)
                    if (s2 < S) sum += pv[s2][mt][reg];
                acc[mt][reg] = sum;
            }
    }
    const int n = tile_row<EPI>(tb, r, ea);
    const float swn = EPI == EPI_RAW ? 1.0f : h2f(ws[n]);
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int row16 = 4 * g + reg;
            const int m = mt * 16 + row16;
            const bool valid = m < M;
            if (EPI == EPI_RAW) {
                if (valid) reinterpret_cast<float*>(out)[(size_t)m * N + tb * 16 + r] = acc[mt][reg];
                continue;
            }
            float v = acc[mt][reg] * swn;
            if (bias) v = v + h2f(bias[n]);
            epilogue_store<EPI>(f2h(v), valid, m, r, tb, N, out, &ex[wave][mt][0], row16 * 16 + r, ea);
        }
}

size_t gemm_w4a16_ws_bytes() { return QS_W16_CNT_SLOTS * sizeof(int) + QS_W16_PART_BYTES; }

template <int EPI>
static int launch_w4a16(const f16* x, const int8_t* wq, const f16* ws, const f16* bias, f16* out, int M, int N, int K,
                        const EpiArgs& ea, void* wsp, hipStream_t st) {
    // 2-D kernel where its activation sharing pays: wide layers (enough 64-row blocks without K blocks) and long K;
    // the narrow K = 4096 layers are faster on the 16-row kernel (no split merge: ~4 us of fences and tickets)
    if (N % 64 == 0 && M <= 32 && (N / 64 >= 192 || K >= 8192)) {
        const int nblocks = N / 64, nrounds = (K / 128 + 3) / 4, MT = M <= 16 ? 1 : 2;
        int S = 1;
        if (wsp && nblocks <= QS_W16_CNT_SLOTS) {
            S = (256 + nblocks - 1) / nblocks;
            if (S > 8) S = 8;
            while (S > 1 && nrounds / S < 2) S--;
            while (S > 1 && (size_t)S * nblocks * 4 * MT * 256 * sizeof(float) > QS_W16_PART_BYTES) S--;
        }
        const int rounds_per = (nrounds + S - 1) / S;
        S = (nrounds + rounds_per - 1) / rounds_per;  // no empty K blocks
        int* cnt = reinterpret_cast<int*>(wsp);
        float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(wsp) + QS_W16_CNT_SLOTS * sizeof(int));
        if (MT == 1)
            hipLaunchKernelGGL((gemm_w4a16_2d_kernel<1, EPI>), dim3(nblocks, S), dim3(256), 0, st, x, wq, ws, bias, out, M, N, K, rounds_per, cnt, part, ea);
        else
            hipLaunchKernelGGL((gemm_w4a16_2d_kernel<2, EPI>), dim3(nblocks, S), dim3(256), 0, st, x, wq, ws, bias, out, M, N, K, rounds_per, cnt, part, ea);
        return 0;
    }
    for (int mb = 0; mb < M; mb += 32) {
        int rows = M - mb;
        if (rows <= 16)
            hipLaunchKernelGGL((gemm_w4a16_kernel<1, EPI>), dim3(N / 16), dim3(256), 0, st, x, wq, ws, bias, out, M, N, K, mb, ea);
        else
            hipLaunchKernelGGL((gemm_w4a16_kernel<2, EPI>), dim3(N / 16), dim3(256), 0, st, x, wq, ws, bias, out, M, N, K, mb, ea);
    }
    return 0;
}

int gemm_w4a16(const f16* x, const int8_t* wq, const f16* ws, const f16* bias, f16* out, int M, int N, int K,
               void* wsp, hipStream_t st) {
    if (M == 0 || N == 0) return 0;
    if (N % 16 || K % 128) return -1;
    return launch_w4a16<EPI_PLAIN>(x, wq, ws, bias, out, M, N, K, EpiArgs{}, wsp, st);
}

// K-sliced (row-parallel) view: x [M, K] with row stride ldx halves, wq [N, K/2] with row stride ldw bytes.
int gemm_w4a16_strided(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, const f16* ws, f16* out, int M, int N,
                       int K, void* wsp, hipStream_t st) {
    if (M == 0 || N == 0) return 0;
    if (N % 16 || K % 128 || ldw % 16 || ldx % 8) return -1;
    EpiArgs ea{};
    ea.ldw = ldw;
    ea.ldx = ldx;
    return launch_w4a16<EPI_PLAIN>(x, wq, ws, nullptr, out, M, N, K, ea, wsp, st);
}

// Raw fp32 sums (no channel scale, no rounding) of a K range: part [M, N] float.  The row-parallel shard of the
// tensor-parallel verify pass: partials are reduced across ranks in fp32 and rounded once, as on one GPU.
int gemm_w4a16_strided_raw(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, float* part, int M, int N, int K,
                           void* wsp, hipStream_t st) {
    if (M == 0 || N == 0) return 0;
    if (N % 16 || K % 128 || ldw % 16 || ldx % 8 || M > 32) return -1;
    EpiArgs ea{};
    ea.ldw = ldw;
    ea.ldx = ldx;
    return launch_w4a16<EPI_RAW>(x, wq, nullptr, nullptr, reinterpret_cast<f16*>(part), M, N, K, ea, wsp, st);
}

int gemm_w4a16_qkv_rope(const f16* x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                        const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                        const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, void* wsp, hipStream_t st) {
    if (M == 0) return 0;
    if (check_qkv(N, K, nq, nkv, d, rot_dim)) return -1;
    EpiArgs ea{positions, cos_sin_cache, key_cache, value_cache, slot_mapping, nq, nkv, 0};
    return launch_w4a16<EPI_QKV>(x, wq, ws, nullptr, qkv, M, N, K, ea, wsp, st);
}

int gemm_w4a16_gate_up_silu(const f16* x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K, int ch0,
                            int nch, void* wsp, hipStream_t st) {
    // channels [ch0, ch0 + nch) of the intermediate dimension (column-parallel shard; nch = I for all of it)
    if (M == 0 || nch == 0) return 0;
    if (I % 8 || K % 128 || ch0 % 8 || nch % 8 || ch0 + nch > I) return -1;
    EpiArgs ea{};
    ea.I = I;
    ea.tb0 = ch0 / 8;
    return launch_w4a16<EPI_GATEUP>(x, wq, ws, nullptr, act, M, 2 * nch, K, ea, wsp, st);
}

// ------------------------------------------------------------------ fp16 x fp16^T (lm_head)
template <int MT>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const f16* __restrict__ x, const f16* __restrict__ w,
                                                        f16* __restrict__ out, int M, int N, int K, int m_base) {
    __shared__ float red[4][MT][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    const int n0 = blockIdx.x * 16;
    const int nsteps = K >> 5;  // 32 k (64 bytes) of every row per step
    const bool nval = n0 + r < N;
    const f16* wrow = w + (size_t)(nval ? n0 + r : 0) * K + g * 8;
    const f16* arow[MT];
    bool aval[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int m = m_base + mt * 16 + r;
        aval[mt] = m < M;
        arow[mt] = x + (size_t)(aval[mt] ? m : 0) * K + g * 8;
    }
    f32x4 acc[MT];
#pragma unroll
    for (int mt = 0; mt < MT; mt++) acc[mt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const f16x8 zero8 = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll 4
    for (int s = wave; s < nsteps; s += 4) {
        f16x8 b = nval ? *reinterpret_cast<const f16x8*>(wrow + (size_t)s * 32) : zero8;
#pragma unroll
        for (int mt = 0; mt < MT; mt++) {
            f16x8 a = aval[mt] ? *reinterpret_cast<const f16x8*>(arow[mt] + (size_t)s * 32) : zero8;
            acc[mt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[mt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int mt = 0; mt < MT; mt++)
#pragma unroll
        for (int i = 0; i < 4; i++) red[wave][mt][i * 64 + lane] = acc[mt][i];
    __syncthreads();
    const int t = threadIdx.x, el = t & 63, reg = t >> 6;
    const int n = n0 + (el & 15);
#pragma unroll
    for (int mt = 0; mt < MT; mt++) {
        int m = m_base + mt * 16 + 4 * (el >> 4) + reg;
        if (m < M && n < N) {
            float sum = ((red[0][mt][t] + red[1][mt][t]) + red[2][mt][t]) + red[3][mt][t];
            out[(size_t)m * N + n] = f2h(sum);
        }
    }
}

int gemm_f16(const f16* x, const f16* w, f16* out, int M, int N, int K, hipStream_t st) {
    if (M == 0 || N == 0) return 0;
    if (K % 32) return -1;
    const int grid = (N + 15) / 16;
    for (int mb = 0; mb < M; mb += 32) {
        int rows = M - mb;
        if (rows <= 16)
            hipLaunchKernelGGL((gemm_f16_kernel<1>), dim3(grid), dim3(256), 0, st, x, w, out, M, N, K, mb);
        else
            hipLaunchKernelGGL((gemm_f16_kernel<2>), dim3(grid), dim3(256), 0, st, x, w, out, M, N, K, mb);
    }
    return 0;
}

// fp16 view of the packed weights with the channel scale folded in: out[n,k] = h(w[n,k] * f(sw[n])).
// Used only for prefill-sized M (library GEMM on the dequantised tile), never on the decode path.
__global__ __launch_bounds__(256) void dequant_w4_kernel(const int8_t* __restrict__ wq, const f16* __restrict__ ws,
                                                          f16* __restrict__ out, int Kb) {
    const int n = blockIdx.x;
    const float s = h2f(ws[n]);
    const uint8_t* wr = reinterpret_cast<const uint8_t*>(wq) + (size_t)n * Kb;
    for (int j = threadIdx.x; j < Kb / 4; j += 256) {
        u32 p = *reinterpret_cast<const u32*>(wr + 4 * j);
        f16x8 o;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            int v = (int)((p >> (4 * c)) & 0xF);
            v = v >= 8 ? v - 16 : v;
            o[c] = f2h((float)v * s);
        }
        *reinterpret_cast<f16x8*>(out + (size_t)n * Kb * 2 + 8 * j) = o;
    }
}
int dequant_w4(const int8_t* wq, const f16* ws, f16* out, int N, int K, hipStream_t st) {
    if (N == 0) return 0;
    if (K % 8) return -1;
    hipLaunchKernelGGL(dequant_w4_kernel, dim3(N), dim3(256), 0, st, wq, ws, out, K / 2);
    return 0;
}

}  // namespace qspec
