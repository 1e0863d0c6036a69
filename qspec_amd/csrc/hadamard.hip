// Online Hadamard rotations (and what is fused around them).
//
// Replaces (reference, relative to /root/reference):
//   fast_hadamard_transform_kernel        third-party/fast-hadamard-transform/csrc/fast_hadamard_transform_cuda.cu:124-198
//   opt_matmul_hadU_cuda / matmul_hadU_cuda  third-party/QuaRot/quarot/functional/hadamard.py:94-124
//   OnlineHadamard.forward                vllm/model_executor/layers/quarot_nn/hadamard.py:23-41
//   heads transpose/FWHT/transpose        vllm/model_executor/models/quarot_llama.py:231-234
//   silu(gate)*up                         quarot_llama.py:279-284
//   rowAbsMaxQuantizeKernel (fused tail)  third-party/QuaRot/quarot/kernels/quant.cu:102-167
//
// All butterflies run in fp32 in increasing-stride order (element-index bit 0
// first), which is the order of the reference kernel (in-thread bits, lane
// bits, chunk bits) and of the oracle, so results are bit-identical whatever
// the lane mapping.  The reference needs 2 transposes + .contiguous() around
// the head transform and a cuBLAS batched GEMM for the had28 mix; here each is
// one kernel that reads the producer's layout directly.
#include <stdlib.h>

#include "common.cuh"
#include "kernels.h"

namespace qspec {

// ------------------------------------------------------------ generic FWHT
// One workgroup per row, row staged in LDS as fp32.  API-parity kernel for
// fast_hadamard_transform(x, scale); the engine uses the fused kernels below.
__global__ __launch_bounds__(256) void fwht_kernel(const f16* __restrict__ x, float scale, f16* __restrict__ out,
                                                    int N) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* v = reinterpret_cast<float*>(smem_raw);
    const size_t base = (size_t)blockIdx.x * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) v[i] = h2f(x[base + i]);
    __syncthreads();
    for (int stride = 1; stride < N; stride <<= 1) {
        for (int b = threadIdx.x; b < N / 2; b += blockDim.x) {
            int lo = b & (stride - 1);
            int i = ((b - lo) << 1) + lo;
            float a = v[i], c = v[i + stride];
            v[i] = a + c;
            v[i + stride] = a - c;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < N; i += blockDim.x) out[base + i] = f2h(v[i] * scale);
}

// rows of one element (get_hadK(n) with n / K == 1: 12, 20, 28, 36, 40 ... heads): only the scale and the rounding are left
__global__ __launch_bounds__(256) void scale_round_kernel(const f16* __restrict__ x, float scale, f16* __restrict__ out,
                                                           int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = f2h(h2f(x[i]) * scale);
}

int fwht(const f16* x, float scale, f16* out, int64_t rows, int N, hipStream_t st) {
    if (rows == 0) return 0;
    if (N == 1) {
        hipLaunchKernelGGL(scale_round_kernel, dim3((unsigned)((rows + 255) / 256)), dim3(256), 0, st, x, scale, out, rows);
        return 0;
    }
    if (N < 2 || N > 32768 || (N & (N - 1))) return -1;
    int threads = N / 2 < 64 ? 64 : (N / 2 > 256 ? 256 : N / 2);
    hipLaunchKernelGGL(fwht_kernel, dim3((unsigned)rows), dim3(threads), (size_t)N * sizeof(float), st, x, scale, out, N);
    return 0;
}

// z[t,i,j] = h(sum_k f(hadK[i,k]) * f(y[t,k,j])), fp32 fma chain in k order.
__global__ __launch_bounds__(256) void hadk_mix_kernel(const f16* __restrict__ y, const f16* __restrict__ hadK,
                                                        f16* __restrict__ out, int K, int M) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* had = reinterpret_cast<float*>(smem_raw);
    for (int i = threadIdx.x; i < K * K; i += blockDim.x) had[i] = h2f(hadK[i]);
    __syncthreads();
    const size_t base = (size_t)blockIdx.x * K * M;
    for (int j = threadIdx.x; j < M; j += blockDim.x)
        for (int i = 0; i < K; i++) {
            float acc = 0.0f;
            for (int k = 0; k < K; k++) acc = __builtin_fmaf(had[i * K + k], h2f(y[base + (size_t)k * M + j]), acc);
            out[base + (size_t)i * M + j] = f2h(acc);
        }
}

int hadk_mix(const f16* y, const f16* hadK, f16* out, int T, int K, int M, hipStream_t st) {
    if (T == 0) return 0;
    if (K < 1 || K > 172) return -1;
    hipLaunchKernelGGL(hadk_mix_kernel, dim3(T), dim3(256), (size_t)K * K * sizeof(float), st, y, hadK, out, K, M);
    return 0;
}

// ------------------------------------------------- heads Hadamard, head count K * 2^p with a table factor K > 1
// (Llama-2-13B: 40 heads = had40; matmul_hadU_cuda on rows of `heads`, quarot/functional/hadamard.py:94-124, between the
// two transposes of quarot_llama.py:231-234).  attn [T, heads, d], head index h = k * P + p:
//   y[h, j] = h( WHT_P over p (fp32, increasing stride) * scale );  out[i * P + p, j] = h( sum_k hadK[i, k] * y[k * P + p, j] )
// One workgroup per token, the token's [heads, d] tile in LDS.  Not a tuned kernel (the configs on the hot path have
// 32 / 64 heads); it keeps the other head counts on the packed-weight engine.
__global__ __launch_bounds__(256) void heads_hadamard_mix_kernel(const f16* __restrict__ attn,
                                                                  const f16* __restrict__ hadK, f16* __restrict__ out,
                                                                  float scale, int heads, int d, int K) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* v = reinterpret_cast<float*>(smem_raw);          // [heads * d]
    float* had = v + (size_t)heads * d;                      // [K * K]
    const int P = heads / K, n = heads * d;
    const size_t base = (size_t)blockIdx.x * n;
    for (int i = threadIdx.x; i < n; i += 256) v[i] = h2f(attn[base + i]);
    for (int i = threadIdx.x; i < K * K; i += 256) had[i] = h2f(hadK[i]);
    __syncthreads();
    for (int stride = 1; stride < P; stride <<= 1) {
        for (int idx = threadIdx.x; idx < (heads / 2) * d; idx += 256) {
            const int b = idx / d, j = idx - b * d;
            const int lo = b & (stride - 1), h = ((b - lo) << 1) + lo;
            const float a = v[h * d + j], c = v[(h + stride) * d + j];
            v[h * d + j] = a + c;
            v[(h + stride) * d + j] = a - c;
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < n; i += 256) v[i] = h2f(f2h(v[i] * scale));
    __syncthreads();
    const int M = P * d;
    for (int idx = threadIdx.x; idx < n; idx += 256) {
        const int i = idx / M, m = idx - i * M;
        float acc = 0.0f;
        for (int k = 0; k < K; k++) acc = __builtin_fmaf(had[i * K + k], v[k * M + m], acc);
        out[base + idx] = f2h(acc);
    }
}

int heads_hadamard_mix(const f16* attn, const f16* hadK, f16* out, float had_scale, int T, int heads, int d, int K,
                       hipStream_t st) {
    if (T == 0) return 0;
    if (K < 2 || K > 172 || heads % K) return -1;
    const int P = heads / K;
    if (P & (P - 1)) return -1;
    const size_t lds = ((size_t)heads * d + (size_t)K * K) * sizeof(float);
    if (lds > 64 * 1024) return -2;
    hipLaunchKernelGGL(heads_hadamard_mix_kernel, dim3(T), dim3(256), lds, st, attn, hadK, out, had_scale, heads, d, K);
    return 0;
}

// ------------------------------------------------- heads Hadamard (o_proj)
// attn [T, NH, d] -> y[t, h', d] = h( (sum_h H[h',h] attn[t,h,d]) * had_scale ), then either fp16 out
// (verify) or row-absmax int4 quant of the whole [NH*d] row (draft).
// One workgroup (d/2 lanes) per token; each lane owns a column pair and all NH heads in registers.
template <int NH, bool QUANT>
__global__ __launch_bounds__(128) void heads_hadamard_kernel(const f16* __restrict__ attn, f16* __restrict__ out16,
                                                             int8_t* __restrict__ q, f16* __restrict__ scale,
                                                             float had_scale, float clip, int d) {
    __shared__ float red[2];
    const int t = blockIdx.x, j = threadIdx.x;
    const f16* ar = attn + (size_t)t * NH * d + 2 * j;
    const bool act = 2 * j < d;  // lanes past d/2 only take part in the wave reduction
    float v0[NH], v1[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) {
        f16x2 a = {(f16)0.0f, (f16)0.0f};
        if (act) a = *reinterpret_cast<const f16x2*>(ar + (size_t)h * d);
        v0[h] = h2f(a[0]);
        v1[h] = h2f(a[1]);
    }
#pragma unroll
    for (int stride = 1; stride < NH; stride <<= 1) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            if (!(h & stride)) {
                float a = v0[h], b = v0[h + stride];
                v0[h] = a + b;
                v0[h + stride] = a - b;
                a = v1[h];
                b = v1[h + stride];
                v1[h] = a + b;
                v1[h + stride] = a - b;
            }
        }
    }
    float amax = 0.0f;
#pragma unroll
    for (int h = 0; h < NH; h++) {
        v0[h] = h2f(f2h(v0[h] * had_scale));
        v1[h] = h2f(f2h(v1[h] * had_scale));
        if (QUANT) {
            float a0 = __builtin_fabsf(v0[h]), a1 = __builtin_fabsf(v1[h]);
            amax = a0 > amax ? a0 : amax;
            amax = a1 > amax ? a1 : amax;
        }
    }
    if (!QUANT) {
        if (!act) return;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            f16x2 o = {f2h(v0[h]), f2h(v1[h])};
            *reinterpret_cast<f16x2*>(out16 + (size_t)t * NH * d + (size_t)h * d + 2 * j) = o;
        }
        return;
    }
    amax = wave_max_f(amax);
    if (blockDim.x > 64) {
        if ((j & 63) == 0) red[j >> 6] = amax;
        __syncthreads();
        amax = fmaxf(red[0], red[1]);
    }
    const f16 sc = f2h(h2f(f2h(amax / 7.0f)) * h2f(f2h(clip)));
    const float scf = h2f(sc);
    const float rcf = 1.0f / scf;   // correctly rounded reciprocal for div3_h
    if (j == 0) scale[t] = sc;
    if (!act) return;
#pragma unroll
    for (int h = 0; h < NH; h++) {
        int q0 = rni_sat(h2f(f2h(div3_h(v0[h], rcf, scf))), -8, 7);
        int q1 = rni_sat(h2f(f2h(div3_h(v1[h], rcf, scf))), -8, 7);
        q[(size_t)t * (NH * d / 2) + (size_t)h * (d / 2) + j] = (int8_t)pack_nib(q0, q1);
    }
}

// Wide form for 32 / 64 heads and d = 128: NH/8 waves per token instead of one.  Thread (hg = wave, dc = lane) owns
// heads 8*hg..8*hg+7 of column pair dc: index bits 0-2 are butterflied in registers, the remaining log2(NH/8)
// stages run over LDS, every thread evaluating the staged butterflies of its own output (same expression tree as
// the sequential transform, so the bits are unchanged) -- 4x fewer instructions per wave than the 64-thread form.
template <int NH, bool QUANT>
__global__ __launch_bounds__(NH * 8) void heads_hadamard_wide_kernel(const f16* __restrict__ attn,
                                                                      f16* __restrict__ out16, int8_t* __restrict__ q,
                                                                      f16* __restrict__ scale, float had_scale,
                                                                      float clip) {
    constexpr int HG = NH / 8, D = 128;
    __shared__ float xl[NH][D];
    __shared__ float red[HG];
    const int t = blockIdx.x, tid = threadIdx.x, dc = tid & 63, hg = tid >> 6;
    const f16* ar = attn + (size_t)t * NH * D + (size_t)hg * 8 * D + 2 * dc;
    float v0[8], v1[8];
#pragma unroll
    for (int h = 0; h < 8; h++) {
        f16x2 a = *reinterpret_cast<const f16x2*>(ar + (size_t)h * D);
        v0[h] = h2f(a[0]);
        v1[h] = h2f(a[1]);
    }
#pragma unroll
    for (int stride = 1; stride < 8; stride <<= 1)
#pragma unroll
        for (int h = 0; h < 8; h++)
            if (!(h & stride)) {
                float a = v0[h], b = v0[h + stride];
                v0[h] = a + b;
                v0[h + stride] = a - b;
                a = v1[h];
                b = v1[h + stride];
                v1[h] = a + b;
                v1[h + stride] = a - b;
            }
#pragma unroll
    for (int h = 0; h < 8; h++) *reinterpret_cast<float2*>(&xl[hg * 8 + h][2 * dc]) = float2{v0[h], v1[h]};
    __syncthreads();
    float amax = 0.0f;
#pragma unroll
    for (int h = 0; h < 8; h++) {
        float x0[HG], x1[HG];
#pragma unroll
        for (int j = 0; j < HG; j++) {
            const float2 p = *reinterpret_cast<const float2*>(&xl[j * 8 + h][2 * dc]);
            x0[j] = p.x;
            x1[j] = p.y;
        }
#pragma unroll
        for (int stride = 1; stride < HG; stride <<= 1)
#pragma unroll
            for (int j = 0; j < HG; j++)
                if (!(j & stride)) {
                    float a = x0[j], b = x0[j + stride];
                    x0[j] = a + b;
                    x0[j + stride] = a - b;
                    a = x1[j];
                    b = x1[j + stride];
                    x1[j] = a + b;
                    x1[j + stride] = a - b;
                }
        float r0 = x0[0], r1 = x1[0];
#pragma unroll
        for (int j = 1; j < HG; j++) {  // select own output without dynamic register indexing
            r0 = hg == j ? x0[j] : r0;
            r1 = hg == j ? x1[j] : r1;
        }
        v0[h] = h2f(f2h(r0 * had_scale));
        v1[h] = h2f(f2h(r1 * had_scale));
        if (QUANT) {
            float a0 = __builtin_fabsf(v0[h]), a1 = __builtin_fabsf(v1[h]);
            amax = a0 > amax ? a0 : amax;
            amax = a1 > amax ? a1 : amax;
        }
    }
    const size_t obase = (size_t)t * NH * D + (size_t)hg * 8 * D + 2 * dc;
    if (!QUANT) {
#pragma unroll
        for (int h = 0; h < 8; h++) {
            f16x2 o = {f2h(v0[h]), f2h(v1[h])};
            *reinterpret_cast<f16x2*>(out16 + obase + (size_t)h * D) = o;
        }
        return;
    }
    amax = wave_max_f(amax);
    if (dc == 0) red[hg] = amax;
    __syncthreads();
    amax = red[0];
#pragma unroll
    for (int j = 1; j < HG; j++) amax = fmaxf(amax, red[j]);
    const f16 sc = f2h(h2f(f2h(amax / 7.0f)) * h2f(f2h(clip)));
    const float scf = h2f(sc);
    const float rcf = 1.0f / scf;   // correctly rounded reciprocal for div3_h
    if (tid == 0) scale[t] = sc;
#pragma unroll
    for (int h = 0; h < 8; h++) {
        int q0 = rni_sat(h2f(f2h(div3_h(v0[h], rcf, scf))), -8, 7);
        int q1 = rni_sat(h2f(f2h(div3_h(v1[h], rcf, scf))), -8, 7);
        q[(obase + (size_t)h * D) / 2] = (int8_t)pack_nib(q0, q1);
    }
}

// heads_hadamard_wide_kernel with the split merge of the attention kernel in front of it: the attention launch
// (attention.hip, merge = 0) leaves per-split partials o [T, NH, S, 128] and (m, l) [T, NH, S, 2] in its workspace and
// this kernel combines them -- out = sum_s w_s o_s / sum_s w_s l_s, w_s = e^(m_s - M), splits in order, the same
// expression as attention.hip's in-kernel merge, rounded to fp16 exactly where flash-attn returns fp16 -- before the
// head transform.  The launch boundary replaces the release / ticket / acquire hand-off (6 of the kernel's 12 us).
// Thread (head = tid / 8, 16 columns) merges with 16-byte loads, all issued up front; the merged row goes through
// LDS to the Hadamard thread mapping.
template <int NH, bool QUANT>
__global__ __launch_bounds__(1024) void heads_hadamard_merge_kernel(const float* __restrict__ ws_o,
                                                                        const float* __restrict__ ws_ml, int S,
                                                                        f16* __restrict__ out16, int8_t* __restrict__ q,
                                                                        f16* __restrict__ scale, float had_scale,
                                                                        float clip, int xp = 0) {
    // 1024 threads merge (CP = 4 or 8 columns each: 4x / 2x fewer loads and instructions per wave than with NH * 8
    // threads, and 4 waves per SIMD to hide them); the first NH * 8 threads then run the head transform, the others
    // retire.
    constexpr int HG = NH / 8, D = 128, SMAX = 8;
    constexpr int CP = NH * D / 1024, TPH = D / CP;   // columns per merge thread, merge threads per head
    __shared__ float xl[NH][D];
    __shared__ __attribute__((aligned(16))) f16 al[NH][D];
    __shared__ float red[HG];
    const int t = blockIdx.x, tid = threadIdx.x, dc = tid & 63, hg = (tid >> 6) & (HG - 1);
    {   // ---- merge
        const int head = tid / TPH, c0 = (tid % TPH) * CP;
        const size_t th = (size_t)t * NH + head;
        const float* ob = ws_o + th * S * D + c0;
        const float* mlb = ws_ml + th * S * 2;
        f32x4 num[CP / 4];
#pragma unroll
        for (int v4 = 0; v4 < CP / 4; v4++) num[v4] = f32x4{0.f, 0.f, 0.f, 0.f};
        float den = 0.0f;
        float M = -__builtin_inff();
        for (int s0 = 0; s0 < S; s0 += SMAX) {   // S <= 8 in one trip: every load in flight before the first use
            float2 ml[SMAX];
            f32x4 o[SMAX][CP / 4];
#pragma unroll
            for (int s2 = 0; s2 < SMAX; s2++) {
                const int sc = min(s0 + s2, S - 1);
                ml[s2] = *reinterpret_cast<const float2*>(mlb + sc * 2);
            }
#pragma unroll
            for (int s2 = 0; s2 < SMAX; s2++) {
                const int sc = min(s0 + s2, S - 1);
#pragma unroll
                for (int v4 = 0; v4 < CP / 4; v4++) o[s2][v4] = *reinterpret_cast<const f32x4*>(ob + (size_t)sc * D + v4 * 4);
            }
            if (s0 == 0) {   // M over ALL splits first (as attention.hip does): from the registers just loaded;
                             // only splits beyond the first trip (S > 8, long contexts) are re-read
#pragma unroll
                for (int s2 = 0; s2 < SMAX; s2++) M = fmaxf(M, s2 < S ? ml[s2].x : -__builtin_inff());
                for (int s2 = SMAX; s2 < S; s2++) M = fmaxf(M, mlb[s2 * 2]);
            }
#pragma unroll
            for (int s2 = 0; s2 < SMAX; s2++) {
                if (s0 + s2 < S) {
                    const float m = ml[s2].x;
                    const float w = m == -__builtin_inff() ? 0.0f : aexp(m - M);   // = attention.hip's own merge
                    den = __builtin_fmaf(w, ml[s2].y, den);
#pragma unroll
                    for (int v4 = 0; v4 < CP / 4; v4++)
#pragma unroll
                        for (int e = 0; e < 4; e++) num[v4][e] = __builtin_fmaf(w, o[s2][v4][e], num[v4][e]);
                }
            }
        }
#pragma unroll
        for (int v4 = 0; v4 < CP / 4; v4++) {
            f16x4 h0;
#pragma unroll
            for (int e = 0; e < 4; e++) h0[e] = f2h(num[v4][e] / den);
            *reinterpret_cast<f16x4*>(&al[head][c0 + v4 * 4]) = h0;
        }
    }
    __syncthreads();
    if (tid >= NH * 8) return;   // retired waves no longer take part in the barriers below
    float v0[8], v1[8];
#pragma unroll
    for (int h = 0; h < 8; h++) {
        const f16x2 a = *reinterpret_cast<const f16x2*>(&al[hg * 8 + h][2 * dc]);
        v0[h] = h2f(a[0]);
        v1[h] = h2f(a[1]);
    }
#pragma unroll
    for (int stride = 1; stride < 8; stride <<= 1)
#pragma unroll
        for (int h = 0; h < 8; h++)
            if (!(h & stride)) {
                float a = v0[h], b = v0[h + stride];
                v0[h] = a + b;
                v0[h + stride] = a - b;
                a = v1[h];
                b = v1[h + stride];
                v1[h] = a + b;
                v1[h + stride] = a - b;
            }
#pragma unroll
    for (int h = 0; h < 8; h++) *reinterpret_cast<float2*>(&xl[hg * 8 + h][2 * dc]) = float2{v0[h], v1[h]};
    __syncthreads();
    float amax = 0.0f;
#pragma unroll
    for (int h = 0; h < 8; h++) {
        float x0[HG], x1[HG];
#pragma unroll
        for (int j = 0; j < HG; j++) {
            const float2 p = *reinterpret_cast<const float2*>(&xl[j * 8 + h][2 * dc]);
            x0[j] = p.x;
            x1[j] = p.y;
        }
#pragma unroll
        for (int stride = 1; stride < HG; stride <<= 1)
#pragma unroll
            for (int j = 0; j < HG; j++)
                if (!(j & stride)) {
                    float a = x0[j], b = x0[j + stride];
                    x0[j] = a + b;
                    x0[j + stride] = a - b;
                    a = x1[j];
                    b = x1[j + stride];
                    x1[j] = a + b;
                    x1[j + stride] = a - b;
                }
        float r0 = x0[0], r1 = x1[0];
#pragma unroll
        for (int j = 1; j < HG; j++) {
            r0 = hg == j ? x0[j] : r0;
            r1 = hg == j ? x1[j] : r1;
        }
        v0[h] = h2f(f2h(r0 * had_scale));
        v1[h] = h2f(f2h(r1 * had_scale));
        if (QUANT) {
            float a0 = __builtin_fabsf(v0[h]), a1 = __builtin_fabsf(v1[h]);
            amax = a0 > amax ? a0 : amax;
            amax = a1 > amax ? a1 : amax;
        }
    }
    const size_t obase = (size_t)t * NH * D + (size_t)hg * 8 * D + 2 * dc;
    if (!QUANT) {
#pragma unroll
        for (int h = 0; h < 8; h++) {
            f16x2 o = {f2h(v0[h]), f2h(v1[h])};
            if (xp) {   // verify pass at <= 32 tokens: fragment-major tiles.  Lane dc holds columns (2 dc, 2 dc + 1); the layout
                // wants (k, k + 4) side by side: lanes with dc & 2 == 0 take their partner's pair (lane ^ 2) and store the four
                // halves k, k + 4, k + 1, k + 5 as one 8-byte group
                const f16x2 po = __builtin_bit_cast(f16x2, dpp_xor<2>(__builtin_bit_cast(float, o)));
                if (!(dc & 2))
                    *reinterpret_cast<f16x4*>(out16 + w4a16_xperm_offset(t, (hg * 8 + h) * D + 2 * dc, NH * D)) =
                        f16x4{o[0], po[0], o[1], po[1]};
            } else {
                *reinterpret_cast<f16x2*>(out16 + obase + (size_t)h * D) = o;
            }
        }
        return;
    }
    amax = wave_max_f(amax);
    if (dc == 0) red[hg] = amax;
    __syncthreads();
    amax = red[0];
#pragma unroll
    for (int j = 1; j < HG; j++) amax = fmaxf(amax, red[j]);
    const f16 sc = f2h(h2f(f2h(amax / 7.0f)) * h2f(f2h(clip)));
    const float scf = h2f(sc);
    const float rcf = 1.0f / scf;   // correctly rounded reciprocal for div3_h
    if (tid == 0) scale[t] = sc;
#pragma unroll
    for (int h = 0; h < 8; h++) {
        int q0 = rni_sat(h2f(f2h(div3_h(v0[h], rcf, scf))), -8, 7);
        int q1 = rni_sat(h2f(f2h(div3_h(v1[h], rcf, scf))), -8, 7);
        q[(obase + (size_t)h * D) / 2] = (int8_t)pack_nib(q0, q1);
    }
}

// The same merge + head transform SPREAD over 8 workgroups per token (32 heads): workgroup (t, y) owns the 16 columns
// [16 y, 16 y + 16) of every head -- the head transform mixes heads, never columns, so the workgroups of a token share
// nothing (fp16 output; the draft pass's quantiser needs the row maximum and stays with the one-workgroup form).
// Why: a CU holds ~32 KB of loads in flight, and a token's 131 KB of split partials (written by other XCDs: they come
// from memory, not from this XCD's L2) took one workgroup four round trips; 16 KB per workgroup is one.
// Thread (wave w, lane): head = 4 w + (lane >> 4), column = lane & 15.  Same merge expression per element and the same
// butterfly order (head strides 1, 2 across lanes; 4, 8, 16 across the waves through 2 KB of LDS) as the kernel above.
// AMAX: also leaves max |output| of the workgroup's 512 values in part_amax[t][y] -- the o_proj launch of the draft pass
// combines the eight maxima of a row and quantises the fp16 row in its own prologue (gemm_stream.hip, PRO_RQ), so the
// eight workgroups of a token need no exchange here either.
template <bool AMAX>
__global__ __launch_bounds__(512) void heads_hadamard_merge_spread32_kernel(const float* __restrict__ ws_o,
                                                                             const float* __restrict__ ws_ml, int S,
                                                                             f16* __restrict__ out16, float had_scale,
                                                                             float* __restrict__ part_amax, int xp = 0) {
    constexpr int NH = 32, D = 128, SMAX = 8;
    __shared__ float xl[NH][16];
    __shared__ float red8[8];
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int hl = lane >> 4, col = lane & 15, head = 4 * w + hl, d = 16 * blockIdx.y + col;
    const size_t th = (size_t)t * NH + head;
    const float* ob = ws_o + th * S * D + d;
    const float* mlb = ws_ml + th * S * 2;
    float num = 0.0f, den = 0.0f, M = -__builtin_inff();
    for (int s0 = 0; s0 < S; s0 += SMAX) {   // S <= 8 in one trip: every load in flight before the first use
        float2 ml[SMAX];
        float o[SMAX];
#pragma unroll
        for (int s2 = 0; s2 < SMAX; s2++) {
            const int sc = min(s0 + s2, S - 1);
            ml[s2] = *reinterpret_cast<const float2*>(mlb + sc * 2);
            o[s2] = ob[(size_t)sc * D];
        }
        if (s0 == 0) {
#pragma unroll
            for (int s2 = 0; s2 < SMAX; s2++) M = fmaxf(M, s2 < S ? ml[s2].x : -__builtin_inff());
            for (int s2 = SMAX; s2 < S; s2++) M = fmaxf(M, mlb[s2 * 2]);
        }
#pragma unroll
        for (int s2 = 0; s2 < SMAX; s2++) {
            if (s0 + s2 < S) {
                const float m = ml[s2].x;
                const float wgt = m == -__builtin_inff() ? 0.0f : aexp(m - M);
                den = __builtin_fmaf(wgt, ml[s2].y, den);
                num = __builtin_fmaf(wgt, o[s2], num);
            }
        }
    }
    float v = h2f(f2h(num / den));
    {   // head strides 1 and 2: lanes ^ 16, ^ 32
        float o = swizzle_xor16_f(v);
        v = (lane & 16) ? (o - v) : (v + o);
        o = shfl_xor_f(v, 32);
        v = (lane & 32) ? (o - v) : (v + o);
    }
    xl[head][col] = v;
    __syncthreads();
    float x[8];
#pragma unroll
    for (int j = 0; j < 8; j++) x[j] = xl[4 * j + hl][col];
#pragma unroll
    for (int stride = 1; stride < 8; stride <<= 1)
#pragma unroll
        for (int j = 0; j < 8; j++)
            if (!(j & stride)) {
                const float a = x[j], b = x[j + stride];
                x[j] = a + b;
                x[j + stride] = a - b;
            }
    float r = x[0];
#pragma unroll
    for (int j = 1; j < 8; j++) r = w == j ? x[j] : r;
    const f16 y = f2h(r * had_scale);
    if (xp) {   // verify pass: the o_proj launch's fragment-major layout (T <= 16): k and k + 4 are neighbours there
        const f16 y4 = f2h(dpp_xor<4>(h2f(y)));
        if (!(col & 4)) *reinterpret_cast<f16x2*>(out16 + w4a16_xperm_offset(t, head * D + d, NH * D)) = f16x2{y, y4};
    } else {
        out16[(size_t)t * NH * D + (size_t)head * D + d] = y;
    }
    if (AMAX) {
        const float am = wave_max_uniform(__builtin_fabsf(h2f(y)));
        if (lane == 0) red8[w] = am;
        __syncthreads();
        if (tid == 0) {
            float m8 = red8[0];
#pragma unroll
            for (int j = 1; j < 8; j++) m8 = fmaxf(m8, red8[j]);
            part_amax[(size_t)t * 8 + blockIdx.y] = m8;
        }
    }
}

// Split merge + head transform for head counts with a TABLE FACTOR (heads = K * P, get_hadK(heads) = hadK; Llama-2-13B: 40
// heads = had40, P = 1), spread over 8 workgroups per token like the 32-head form above: workgroup (t, y) owns the 16
// columns [16 y, 16 y + 16) of every head.  Per element the arithmetic of paged_attention(out != NULL) followed by
// heads_hadamard_mix_kernel (same merge expression, FWHT over p with increasing stride, h(v * scale), then the k-ordered
// fp32 fma chain over the table), so the same bits -- without the in-kernel split merge of the attention launch (ticket +
// fences: 6 of its 12 us), without the one-workgroup-per-token transform (17 us for 4 tokens on 4 CUs) and, for the
// draft pass, without the separate quantiser launch: AMAX leaves max |output| of the workgroup's share in
// part_amax[t][y] and the row-absmax quantiser runs in the prologue of the o_proj launch (gemm_stream.hip, PRO_RQ).
// Thread (head = tid / 16, column = tid % 16); LDS: y [heads][16] fp32 + the table as fp32.
template <bool AMAX>
__global__ __launch_bounds__(1024) void heads_hadamard_mix_merge_spread_kernel(const float* __restrict__ ws_o,
                                                                                const float* __restrict__ ws_ml, int S,
                                                                                const f16* __restrict__ hadK,
                                                                                f16* __restrict__ out16, float had_scale,
                                                                                float* __restrict__ part_amax, int heads,
                                                                                int K, int xp = 0) {
    constexpr int D = 128, SMAX = 8;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* yl = reinterpret_cast<float*>(smem_raw);          // [heads][16]
    float* had = yl + (size_t)heads * 16;                     // [K][K]
    float* red = had + (size_t)K * K;                         // [16]
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int head = tid >> 4, col = tid & 15, d = 16 * blockIdx.y + col;
    const bool act = head < heads;
    const int hc = act ? head : heads - 1;                    // clamped: every load below is unconditional
    const size_t th = (size_t)t * heads + hc;
    const float* ob = ws_o + th * S * D + d;
    const float* mlb = ws_ml + th * S * 2;
    float num = 0.0f, den = 0.0f, M = -__builtin_inff();
    for (int s0 = 0; s0 < S; s0 += SMAX) {   // S <= 8 in one trip: every load in flight before the first use
        float2 ml[SMAX];
        float o[SMAX];
#pragma unroll
        for (int s2 = 0; s2 < SMAX; s2++) {
            const int sc = min(s0 + s2, S - 1);
            ml[s2] = *reinterpret_cast<const float2*>(mlb + sc * 2);
            o[s2] = ob[(size_t)sc * D];
        }
        if (s0 == 0) {
#pragma unroll
            for (int s2 = 0; s2 < SMAX; s2++) M = fmaxf(M, s2 < S ? ml[s2].x : -__builtin_inff());
            for (int s2 = SMAX; s2 < S; s2++) M = fmaxf(M, mlb[s2 * 2]);
        }
#pragma unroll
        for (int s2 = 0; s2 < SMAX; s2++) {
            if (s0 + s2 < S) {
                const float m = ml[s2].x;
                const float wgt = m == -__builtin_inff() ? 0.0f : aexp(m - M);
                den = __builtin_fmaf(wgt, ml[s2].y, den);
                num = __builtin_fmaf(wgt, o[s2], num);
            }
        }
    }
    for (int i = tid; i < K * K; i += blockDim.x) had[i] = h2f(hadK[i]);
    float v = h2f(f2h(num / den));                            // the attention output, rounded where flash-attn returns fp16
    const int P = heads / K;
    for (int stride = 1; stride < P; stride <<= 1) {          // FWHT over p (head index h = k P + p), increasing stride
        if (act) yl[head * 16 + col] = v;
        __syncthreads();
        const float o = yl[(hc ^ stride) * 16 + col];
        v = (head & stride) ? (o - v) : (v + o);
        __syncthreads();
    }
    if (act) yl[head * 16 + col] = h2f(f2h(v * had_scale));
    __syncthreads();
    const int i = hc / P, pp = hc - i * P;
    const float* hrow = had + (size_t)i * K;
    float acc = 0.0f;
    for (int k = 0; k < K; k++) acc = __builtin_fmaf(hrow[k], yl[(k * P + pp) * 16 + col], acc);   // = heads_hadamard_mix_kernel
    const f16 y = f2h(acc);
    if (xp) {   // (k, k + 4) pairs, as in heads_hadamard_merge_spread32_kernel; heads in whole 16-lane groups
        const f16 y4 = f2h(dpp_xor<4>(h2f(y)));
        if (act && !(col & 4)) *reinterpret_cast<f16x2*>(out16 + w4a16_xperm_offset(t, head * D + d, heads * D)) = f16x2{y, y4};
    } else if (act) {
        out16[(size_t)t * heads * D + (size_t)head * D + d] = y;
    }
    if (AMAX) {
        const float am = wave_max_f(act ? __builtin_fabsf(h2f(y)) : 0.0f);
        if (lane == 0) red[w] = am;
        __syncthreads();
        if (tid == 0) {
            float m8 = red[0];
            const int nw = (blockDim.x + 63) >> 6;
            for (int j = 1; j < nw; j++) m8 = fmaxf(m8, red[j]);
            part_amax[(size_t)t * 8 + blockIdx.y] = m8;
        }
    }
}

bool heads_hadamard_mix_merge_spread_supported(int T, int heads, int d, int K) {
    if (d != 128 || K < 2 || K > 172 || heads % K || heads * 16 > 1024 || T < 0 || T * 8 > 1024) return false;
    const int P = heads / K;
    return (P & (P - 1)) == 0;
}
// fp16 rows + (part_amax != nullptr) 8 partial row maxima per token
int heads_hadamard_mix_merge_spread(const float* ws, int max_tokens, int n_splits, const f16* hadK, f16* out_f16,
                                    float* part_amax, float had_scale, int T, int heads, int d, int K, hipStream_t st, int xp) {
    if (T == 0) return 0;
    if (xp && (part_amax != nullptr || T > 32)) return -1;
    if (!heads_hadamard_mix_merge_spread_supported(T, heads, d, K) || n_splits < 1 || T > max_tokens) return -1;
    const float* ws_o = ws + paged_attention_ws_o_offset();
    const float* ws_ml = ws + paged_attention_ws_ml_offset(max_tokens, heads, d, n_splits);
    const int threads = ((heads * 16 + 63) / 64) * 64;
    const size_t lds = ((size_t)heads * 16 + (size_t)K * K + 16) * sizeof(float);
    if (part_amax)
        hipLaunchKernelGGL(heads_hadamard_mix_merge_spread_kernel<true>, dim3(T, 8), dim3(threads), lds, st, ws_o, ws_ml,
                           n_splits, hadK, out_f16, had_scale, part_amax, heads, K, 0);
    else
        hipLaunchKernelGGL(heads_hadamard_mix_merge_spread_kernel<false>, dim3(T, 8), dim3(threads), lds, st, ws_o, ws_ml,
                           n_splits, hadK, out_f16, had_scale, part_amax, heads, K, xp);
    return 0;
}

// heads_hadamard_kernel with the split merge of the generic-head-size attention kernel (attention.hip:
// paged_attention_generic_vec_kernel, head sizes other than 128 -- TinyLlama's 64) in front of it: 512 threads merge
// (head, four columns) items -- out = h(sum_s w_s o_s / sum_s w_s l_s), w_s = e^(m_s - M), splits in order: the expression of
// paged_attention_generic_merge_kernel, so the same fp16 rows -- into LDS, then the first d / 2 lanes run
// heads_hadamard_kernel's transform on them (same arithmetic: same bits as merge -> heads_hadamard).
template <int NH, bool QUANT>
__global__ __launch_bounds__(512) void heads_hadamard_merge_cols_kernel(const float* __restrict__ ws_o,
                                                                        const float* __restrict__ ws_ml, int S,
                                                                        f16* __restrict__ out16, int8_t* __restrict__ q,
                                                                        f16* __restrict__ scale, float had_scale, float clip,
                                                                        int d) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    f16* al = reinterpret_cast<f16*>(smem_raw);   // [NH][d]
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    const int per_head = d >> 2;
    for (int it = tid; it < NH * per_head; it += 512) {
        const int head = it / per_head, c0 = (it - head * per_head) * 4;
        const size_t th = (size_t)t * NH + head;
        const float* mlb = ws_ml + th * S * 2;
        const float* ob = ws_o + th * S * d + c0;
        // S <= 4 (paged_attention_generic_splits; 8 slots): every load of the item in flight before the first use
        constexpr int SMAX = 8;
        float2 ml[SMAX];
        f32x4 o[SMAX];
#pragma unroll
        for (int s2 = 0; s2 < SMAX; s2++) {
            const int sc = min(s2, S - 1);
            ml[s2] = *reinterpret_cast<const float2*>(mlb + sc * 2);
            o[s2] = *reinterpret_cast<const f32x4*>(ob + (size_t)sc * d);
        }
        float M = -__builtin_inff();
#pragma unroll
        for (int s2 = 0; s2 < SMAX; s2++) M = fmaxf(M, s2 < S ? ml[s2].x : -__builtin_inff());
        f32x4 num = f32x4{0.f, 0.f, 0.f, 0.f};
        float den = 0.0f;
#pragma unroll
        for (int s2 = 0; s2 < SMAX; s2++) {
            if (s2 < S) {
                const float w = ml[s2].x == -__builtin_inff() ? 0.0f : aexp(ml[s2].x - M);
                den = __builtin_fmaf(w, ml[s2].y, den);
#pragma unroll
                for (int e = 0; e < 4; e++) num[e] = __builtin_fmaf(w, o[s2][e], num[e]);
            }
        }
        f16x4 h4;
#pragma unroll
        for (int e = 0; e < 4; e++) h4[e] = f2h(num[e] / den);
        *reinterpret_cast<f16x4*>(al + (size_t)head * d + c0) = h4;
    }
    __syncthreads();
    const int TT = ((d / 2 + 63) / 64) * 64;       // transform threads (whole waves), as heads_hadamard's launch
    if (tid >= TT) return;
    const int j = tid;
    const bool act = 2 * j < d;
    float v0[NH], v1[NH];
#pragma unroll
    for (int h = 0; h < NH; h++) {
        f16x2 a = {(f16)0.0f, (f16)0.0f};
        if (act) a = *reinterpret_cast<const f16x2*>(al + (size_t)h * d + 2 * j);
        v0[h] = h2f(a[0]);
        v1[h] = h2f(a[1]);
    }
#pragma unroll
    for (int stride = 1; stride < NH; stride <<= 1) {
#pragma unroll
        for (int h = 0; h < NH; h++) {
            if (!(h & stride)) {
                float a = v0[h], b = v0[h + stride];
                v0[h] = a + b;
                v0[h + stride] = a - b;
                a = v1[h];
                b = v1[h + stride];
                v1[h] = a + b;
                v1[h + stride] = a - b;
            }
        }
    }
    float amax = 0.0f;
#pragma unroll
    for (int h = 0; h < NH; h++) {
        v0[h] = h2f(f2h(v0[h] * had_scale));
        v1[h] = h2f(f2h(v1[h] * had_scale));
        if (QUANT) {
            float a0 = __builtin_fabsf(v0[h]), a1 = __builtin_fabsf(v1[h]);
            amax = a0 > amax ? a0 : amax;
            amax = a1 > amax ? a1 : amax;
        }
    }
    if (!QUANT) {
        if (!act) return;
#pragma unroll
        for (int h = 0; h < NH; h++) {
            f16x2 o = {f2h(v0[h]), f2h(v1[h])};
            *reinterpret_cast<f16x2*>(out16 + (size_t)t * NH * d + (size_t)h * d + 2 * j) = o;
        }
        return;
    }
    amax = wave_max_f(amax);
    if (TT > 64) {   // (the retired waves take no part in this barrier)
        if ((j & 63) == 0) red[j >> 6] = amax;
        __syncthreads();
        amax = fmaxf(red[0], red[1]);
    }
    const f16 sc = f2h(h2f(f2h(amax / 7.0f)) * h2f(f2h(clip)));
    const float scf = h2f(sc);
    const float rcf = 1.0f / scf;
    if (j == 0) scale[t] = sc;
    if (!act) return;
#pragma unroll
    for (int h = 0; h < NH; h++) {
        int q0 = rni_sat(h2f(f2h(div3_h(v0[h], rcf, scf))), -8, 7);
        int q1 = rni_sat(h2f(f2h(div3_h(v1[h], rcf, scf))), -8, 7);
        q[((size_t)t * NH * d + (size_t)h * d + 2 * j) / 2] = (int8_t)pack_nib(q0, q1);
    }
}

// partials: the workspace of paged_attention(..., out = nullptr) called for `max_tokens` = n_seqs * max_q_len tokens
int heads_hadamard_merge(const float* ws, int max_tokens, int n_splits, f16* out_f16, int8_t* q, f16* scale,
                         float had_scale, float clip, int T, int heads, int d, hipStream_t st, int xp) {
    if (T == 0) return 0;
    if (xp && (q != nullptr || !(heads == 32 || heads == 64) || T > 32 || d != 128)) return -1;   // fragment-major fp16 rows (no quantiser)
    if (d != 128) {   // generic head sizes (TinyLlama: 32 heads of 64): partials of paged_attention_generic_vec_kernel
        if (heads != 32 || d % 8 || d < 8 || d > 256 || n_splits < 1 || T > max_tokens) return -1;
        const int S = paged_attention_generic_splits(n_splits);
        const float* go = ws + paged_attention_ws_o_offset();
        const float* gml = ws + paged_attention_ws_ml_offset(max_tokens, heads, d, S);
        const size_t lds = (size_t)heads * d * sizeof(f16);
        if (q)
            hipLaunchKernelGGL((heads_hadamard_merge_cols_kernel<32, true>), dim3(T), dim3(512), lds, st, go, gml, S, out_f16, q, scale,
                               had_scale, clip, d);
        else
            hipLaunchKernelGGL((heads_hadamard_merge_cols_kernel<32, false>), dim3(T), dim3(512), lds, st, go, gml, S, out_f16, q, scale,
                               had_scale, clip, d);
        return 0;
    }
    if (!(heads == 32 || heads == 64) || n_splits < 1 || T > max_tokens) return -1;
    const float* ws_o = ws + paged_attention_ws_o_offset();
    const float* ws_ml = ws + paged_attention_ws_ml_offset(max_tokens, heads, d, n_splits);
    const bool quant = q != nullptr;
    static const int spread = QS_DEV_KNOB("QSPEC_HHM_SPREAD", 1);   // (0: one workgroup per token also for the fp16 form)
    if (!quant && heads == 32 && spread && T * 8 <= 1024) {
        hipLaunchKernelGGL(heads_hadamard_merge_spread32_kernel<false>, dim3(T, 8), dim3(512), 0, st, ws_o, ws_ml, n_splits,
                           out_f16, had_scale, (float*)nullptr, xp);
        return 0;
    }
    if (xp && heads != 64) return -1;
#define QS_HHM(NHV)                                                                                                \
    if (heads == NHV) {                                                                                             \
        if (quant)                                                                                                  \
            hipLaunchKernelGGL((heads_hadamard_merge_kernel<NHV, true>), dim3(T), dim3(1024), 0, st, ws_o, ws_ml, \
                               n_splits, out_f16, q, scale, had_scale, clip);                                       \
        else                                                                                                        \
            hipLaunchKernelGGL((heads_hadamard_merge_kernel<NHV, false>), dim3(T), dim3(1024), 0, st, ws_o, ws_ml, \
                               n_splits, out_f16, q, scale, had_scale, clip, xp);                                   \
        return 0;                                                                                                   \
    }
    QS_HHM(32) QS_HHM(64)
#undef QS_HHM
    return -1;
}

// fp16 output + 8 partial row maxima per token (see the kernel); 32 heads of 128
int heads_hadamard_merge_spread(const float* ws, int max_tokens, int n_splits, f16* out_f16, float* part_amax, float had_scale,
                                int T, int heads, int d, hipStream_t st) {
    if (T == 0) return 0;
    if (d != 128 || heads != 32 || n_splits < 1 || T > max_tokens || T * 8 > 1024) return -1;
    const float* ws_o = ws + paged_attention_ws_o_offset();
    const float* ws_ml = ws + paged_attention_ws_ml_offset(max_tokens, heads, d, n_splits);
    hipLaunchKernelGGL(heads_hadamard_merge_spread32_kernel<true>, dim3(T, 8), dim3(512), 0, st, ws_o, ws_ml, n_splits, out_f16,
                       had_scale, part_amax);
    return 0;
}

int heads_hadamard(const f16* attn, f16* out_f16, int8_t* q, f16* scale, float had_scale, float clip, int T, int heads,
                   int d, hipStream_t st) {
    if (T == 0) return 0;
    if (d % 2 || d / 2 > 128) return -1;
    const int threads = ((d / 2 + 63) / 64) * 64;
    const bool quant = q != nullptr;
    if (d == 128 && (heads == 32 || heads == 64)) {
#define QS_HHW(NHV)                                                                                              \
    if (heads == NHV) {                                                                                           \
        if (quant)                                                                                                \
            hipLaunchKernelGGL((heads_hadamard_wide_kernel<NHV, true>), dim3(T), dim3(NHV * 8), 0, st, attn,      \
                               out_f16, q, scale, had_scale, clip);                                               \
        else                                                                                                      \
            hipLaunchKernelGGL((heads_hadamard_wide_kernel<NHV, false>), dim3(T), dim3(NHV * 8), 0, st, attn,     \
                               out_f16, q, scale, had_scale, clip);                                               \
        return 0;                                                                                                 \
    }
        QS_HHW(32) QS_HHW(64)
#undef QS_HHW
    }
#define QS_HH(NHV)                                                                                               \
    if (heads == NHV) {                                                                                           \
        if (quant)                                                                                                \
            hipLaunchKernelGGL((heads_hadamard_kernel<NHV, true>), dim3(T), dim3(threads), 0, st, attn, out_f16, q, \
                               scale, had_scale, clip, d);                                                        \
        else                                                                                                      \
            hipLaunchKernelGGL((heads_hadamard_kernel<NHV, false>), dim3(T), dim3(threads), 0, st, attn, out_f16, q, \
                               scale, had_scale, clip, d);                                                        \
        return 0;                                                                                                 \
    }
    QS_HH(8) QS_HH(16) QS_HH(32) QS_HH(64)
#undef QS_HH
    return -1;
}

// standalone silu(gate)*up (API-parity op for the module-wise path; the engine uses the fused kernel below)
__global__ __launch_bounds__(256) void silu_mul_kernel(const f16* __restrict__ gate_up, f16* __restrict__ out, int I) {
    const int t = blockIdx.x;
    const f16* up = gate_up + (size_t)t * 2 * I;
    const f16* gate = up + I;
    for (int i = threadIdx.x; i < I; i += 256) {
        float g = h2f(gate[i]);
        float a = h2f(f2h(g / (1.0f + qexpf(-g))));
        out[(size_t)t * I + i] = f2h(a * h2f(up[i]));
    }
}
int silu_mul(const f16* gate_up, f16* out, int T, int I, hipStream_t st) {
    if (T == 0) return 0;
    hipLaunchKernelGGL(silu_mul_kernel, dim3(T), dim3(256), 0, st, gate_up, out, I);
    return 0;
}

// ------------------------------------- silu*up -> (hadK (x) H_P) -> quant (down_proj)
// gate_up [T, 2I] (up first, gate second).  I = K * P, P a power of two.
//   g[e]    = h( h(silu(gate[e])) * up[e] )
//   y[k,:]  = h( WHT_P(g[k,:]) * had_scale )
//   z[i,j]  = h( sum_k hadK[i,k] y[k,j] )            (K == 1: z = y)
//   draft: row-absmax int4 of z (index i*P + j); verify: fp16 z.
// One 256-thread workgroup per token.  LDS: y fp16 [K][P], z fp16 [K][P], hadK fp32 [K][K].
#define QS_SMH_THREADS 1024

// Row abs-max across the NB workgroups that share a token (the spread forms below), without atomics or fences:
// data-tagged 8-byte granules {amax, tag} (MI355X_MICROARCH.md price list, handoff-1to1: one naturally aligned 8-byte
// sc1 store per producer, sc1 polling loads; observed untorn).  Workspace per token: a generation word gen[t] and NB
// granules.  Every workgroup of token t reads g = gen[t] when it starts (latency hidden under the transform), publishes
// {its maximum, g + 1}, and polls the token's NB granules until every tag reads g + 1.  Workgroup 0 of the token then
// stores gen[t] = g + 1: by then every workgroup of the token has published, hence has read gen[t] -- and the next
// launch starts behind a kernel boundary.  Zero-filled ONCE, never reset; tags only ever compare equal, so the
// generation may wrap.  All NB workgroups of a token must be resident together: the host launches at most one
// workgroup per CU.  A poll that exceeds its guard raises the workspace's sticky error word (read by the host once per
// cycle) instead of hanging; the result is then this workgroup's own maximum.
#define QS_XWG_ERR_WORD 0          // uint32 [0]: sticky error flag; gen[] from word 16, granules behind them
#define QS_XWG_MAX_TOKENS 256
#define QS_XWG_MAX_NB 16
__device__ __forceinline__ uint32_t xwg_read_gen(const uint32_t* ws, int t) {
    return __hip_atomic_load(ws + 16 + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// Call from every thread of the workgroup BEHIND the barrier that made wave_max[0 .. nwaves) (the waves' maxima, LDS)
// visible; gen uniform over the workgroup; lds_slot: one float of LDS that nothing else touches until the next barrier.
// One barrier inside; only wave 0 reduces, publishes and polls.
__device__ __forceinline__ float xwg_row_amax(const float* wave_max, int nwaves, uint32_t gen, int t, int b, int nb,
                                              uint32_t* ws, float* lds_slot) {
    unsigned long long* gran = reinterpret_cast<unsigned long long*>(ws + 16 + QS_XWG_MAX_TOKENS) + (size_t)t * QS_XWG_MAX_NB;
    if (threadIdx.x < 64) {   // wave 0
        const int lane = threadIdx.x;
        const uint32_t tag = gen + 1u;
        const float wg_amax = wave_max_f(lane < nwaves ? wave_max[lane] : 0.0f);
        if (lane == 0)
            __hip_atomic_store(gran + b, ((unsigned long long)tag << 32) | __builtin_bit_cast(uint32_t, wg_amax),
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        float m = wg_amax;
        int guard = 0;
        while (true) {
            unsigned long long v = ((unsigned long long)tag << 32);
            if (lane < nb) v = __hip_atomic_load(gran + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const bool ok = (uint32_t)(v >> 32) == tag;
            if (__builtin_amdgcn_ballot_w64(ok) == ~0ull) {
                m = lane < nb ? __builtin_bit_cast(float, (uint32_t)v) : 0.0f;
                break;
            }
            if (++guard > (1 << 20)) {
                if (lane == 0) __hip_atomic_store(ws + QS_XWG_ERR_WORD, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                break;
            }
            __builtin_amdgcn_s_sleep(1);
        }
        m = wave_max_f(m);
        if (lane == 0) {
            *lds_slot = m;
            if (b == 0) __hip_atomic_store(ws + 16 + t, tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __syncthreads();
    return *lds_slot;
}

// NB > 1 (KH > 0, PREACT): NB workgroups per token.  Each repeats the cheap FWHT phase for the whole token and mixes /
// quantises only its block of P / NB columns of every hadK row (the mix is what bounds the one-workgroup form: 6.3 k
// cycles of v_pk_fma on ONE CU); the row maximum of the quantiser is exchanged through xwg_row_amax.
// PW < EPL * 64 (P = 128 with 16-byte lanes: Llama-2-13B's 13824 = had108 x H128): a wave trip of the FWHT phase covers
// EPL * 64 / PW rows of P at once -- the lane stages stop below P -- so that the token's row still travels in 1 KiB wave loads,
// all of them in flight before the first transform (with EPL = P / 64 = 2 a wave took 4-byte lanes and seven dependent trips).
// (Round 4, measured and rejected: 14 waves = 896 threads for the spread had28 x H512 form -- two rows of the transform per wave
// instead of 12 waves with two and 4 with one, and exactly 28 x 32 mix items: cycle 7.48 -> 7.565 ms in two alternating runs.)
template <int EPL, int KH, bool PREACT, int NB = 1, int PW = EPL * 64>  // EPL = elements per lane in the FWHT phase; P = PW; KH = K if specialised, else 0
__global__ __launch_bounds__(QS_SMH_THREADS) void silu_mul_hadamard_kernel(const f16* __restrict__ gate_up,
                                                                            const f16* __restrict__ hadK,
                                                                            f16* __restrict__ out16,
                                                                            int8_t* __restrict__ q,
                                                                            f16* __restrict__ scale, float had_scale,
                                                                            float clip, int I, int K, uint32_t* xws, int xp = 0) {
    constexpr bool pre_activated = PREACT;   // the input is already g = silu(gate)*up, [T, I] (gate_up GEMM epilogue)
    constexpr int P = PW, CHUNK = EPL * 64;   // row length of the transform; elements per wave trip of phase A
    static_assert(CHUNK % P == 0 && P >= EPL, "a wave trip covers whole rows");
    constexpr int NT = QS_SMH_THREADS, NW = NT / 64;
    const int nchk = I / CHUNK;               // (K rows of P = nchk wave trips; the host checks I % CHUNK == 0)
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    // all LDS in the one dynamic region so every carve stays 16-byte aligned
    float* red = reinterpret_cast<float*>(smem_raw);  // [16]
    f16* ylds = reinterpret_cast<f16*>(smem_raw + 64);
    f16* zlds = ylds + (size_t)K * P;
    float* had = reinterpret_cast<float*>(zlds + (K > 1 ? (size_t)K * P : 0));
    const int t = NB > 1 ? blockIdx.x / NB : blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // pre_activated: the input is already g = silu(gate)*up, [T, I] (fused into the gate_up GEMM epilogue)
    const f16* up = gate_up + (size_t)t * (pre_activated ? I : 2 * I);
    const f16* gate = up + I;
    uint32_t xgen = 0;
    if (NB > 1 && q != nullptr) xgen = xwg_read_gen(xws, t);   // consumed at the exchange, far behind this load
    // hadK -> LDS AFTER phase A (it is only read in phase B): staged here it would put an L2 round trip in front of
    // the first activation load (results return in issue order).

    // phase A: one chunk of P elements per wave trip (a token's 14336 elements keep 16 waves = 4 per SIMD busy:
    // this kernel is VALU bound -- correctly rounded fp32 divisions in SiLU and in the quantiser)
    // pre-activated input with 16-byte lanes (the engine's path): the loads of a wave's first two trips (all of them
    // for K <= 32) are issued before the first transform -- one round trip to the cold activations instead of two
    constexpr bool PF = PREACT && EPL >= 8;
    constexpr int NB8 = EPL >= 8 ? EPL / 8 : 1;
    f16x8 pre8[2][NB8];
    if (PF) {
#pragma unroll
        for (int tr = 0; tr < 2; tr++)
#pragma unroll
            for (int b = 0; b < NB8; b++)
                pre8[tr][b] = *reinterpret_cast<const f16x8*>(up + min(wave + tr * NW, nchk - 1) * CHUNK + lane * EPL + 8 * b);
    }
    int trip = 0;
    for (int c = wave; c < nchk; c += NW, trip++) {
        float v[EPL];
        const int e0 = c * CHUNK + lane * EPL;
        if (PF && trip < 2) {
#pragma unroll
            for (int b = 0; b < NB8; b++) {
                const f16x8 u8 = trip == 0 ? pre8[0][b] : pre8[1][b];
#pragma unroll
                for (int i = 0; i < 8; i++) v[8 * b + i] = h2f(u8[i]);
            }
        } else if (pre_activated) {
            if (EPL >= 8) {
#pragma unroll
                for (int b = 0; b < EPL / 8; b++) {
                    f16x8 u8 = *reinterpret_cast<const f16x8*>(up + e0 + 8 * b);
#pragma unroll
                    for (int i = 0; i < 8; i++) v[8 * b + i] = h2f(u8[i]);
                }
            } else {
#pragma unroll
                for (int i = 0; i < EPL; i++) v[i] = h2f(up[e0 + i]);
            }
        } else if (EPL >= 8) {
#pragma unroll
            for (int b = 0; b < EPL / 8; b++) {
                f16x8 g8 = *reinterpret_cast<const f16x8*>(gate + e0 + 8 * b);
                f16x8 u8 = *reinterpret_cast<const f16x8*>(up + e0 + 8 * b);
#pragma unroll
                for (int i = 0; i < 8; i++) {
                    float g = h2f(g8[i]);
                    float a = h2f(f2h(g / (1.0f + qexpf(-g))));
                    v[8 * b + i] = h2f(f2h(a * h2f(u8[i])));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPL; i++) {
                float g = h2f(gate[e0 + i]);
                float a = h2f(f2h(g / (1.0f + qexpf(-g))));
                v[i] = h2f(f2h(a * h2f(up[e0 + i])));
            }
        }
#pragma unroll
        for (int stride = 1; stride < EPL; stride <<= 1) {
#pragma unroll
            for (int i = 0; i < EPL; i++)
                if (!(i & stride)) {
                    float a = v[i], b = v[i + stride];
                    v[i] = a + b;
                    v[i + stride] = a - b;
                }
        }
        // lane-exchange stages: xor 1, 2, 4, 8 as DPP moves (no LDS round trip), 16 and 32 through ds_bpermute
#define QS_FWHT_STAGE(M, EXCH)                                  \
        if constexpr (M * EPL < P) {                            \
            const bool hi = lane & M;                           \
            _Pragma("unroll") for (int i = 0; i < EPL; i++) {   \
                const float o = EXCH;                           \
                v[i] = hi ? (o - v[i]) : (v[i] + o);            \
            }                                                   \
        }
        QS_FWHT_STAGE(1, dpp_xor<1>(v[i]))
        QS_FWHT_STAGE(2, dpp_xor<2>(v[i]))
        QS_FWHT_STAGE(4, dpp_xor<4>(v[i]))
        QS_FWHT_STAGE(8, dpp_xor<8>(v[i]))
        QS_FWHT_STAGE(16, shfl_xor_f(v[i], 16))
        QS_FWHT_STAGE(32, shfl_xor_f(v[i], 32))
#undef QS_FWHT_STAGE
        if (EPL >= 8) {  // one 16-byte LDS store per 8 values (2-byte stores at a 16-byte lane stride are 8-way conflicted)
#pragma unroll
            for (int b = 0; b < EPL / 8; b++) {
                f16x8 o8;
#pragma unroll
                for (int i = 0; i < 8; i++) o8[i] = f2h(v[8 * b + i] * had_scale);
                *reinterpret_cast<f16x8*>(ylds + (size_t)c * CHUNK + lane * EPL + 8 * b) = o8;
            }
        } else {
#pragma unroll
            for (int i = 0; i < EPL; i++) ylds[(size_t)c * CHUNK + lane * EPL + i] = f2h(v[i] * had_scale);
        }
    }
#ifdef QS_SMH_STAMPS
    long long stp[6];
#define QS_ST2(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stp[i])::"memory")
#else
#define QS_ST2(i)
#endif
    if (K > 1)
        for (int i = tid; i < K * K; i += NT) had[i] = h2f(hadK[i]);
    QS_ST2(1);
    __syncthreads();
    QS_ST2(2);
    if constexpr (NB > 1) {
        static_assert(KH > 0 && KH % 4 == 0 && (P / NB) % 8 == 0, "spread form");
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        constexpr int CB = P / NB, PAIRS = CB / 2;            // this workgroup's columns of every row
        constexpr int ITEMS = KH * PAIRS;                     // (row, column pair) items, one per thread
        static_assert(ITEMS <= NT && PAIRS % 4 == 0, "one item per thread; four lanes = one 8-element group (xp stores)");
        const int b = blockIdx.x % NB;
        const int row = tid / PAIRS, jp = tid % PAIRS;
        const bool active = tid < ITEMS;
        f32x2 a = {0.0f, 0.0f};
        if (active) {
            const f16* ycol = ylds + b * CB + 2 * jp;
            const float* hrow = had + row * KH;
#pragma unroll
            for (int kq = 0; kq < KH / 4; kq++) {
                const float4 h = *reinterpret_cast<const float4*>(hrow + 4 * kq);
                f32x2 y[4];
#pragma unroll
                for (int kk = 0; kk < 4; kk++) {
                    const f16x2 yy = *reinterpret_cast<const f16x2*>(ycol + (size_t)(4 * kq + kk) * P);
                    y[kk] = f32x2{h2f(yy[0]), h2f(yy[1])};
                }
                a = __builtin_elementwise_fma(f32x2{h.x, h.x}, y[0], a);   // the k-ordered fp32 chain of the oracle
                a = __builtin_elementwise_fma(f32x2{h.y, h.y}, y[1], a);
                a = __builtin_elementwise_fma(f32x2{h.z, h.z}, y[2], a);
                a = __builtin_elementwise_fma(f32x2{h.w, h.w}, y[3], a);
            }
        }
        const f16x2 zz = {f2h(a[0]), f2h(a[1])};
        const size_t e0 = (size_t)row * P + b * CB + 2 * jp;  // element index inside the token's row of I
        if (q == nullptr) {
            if (xp) {
                // verify pass: down_proj's fragment-major layout (T <= 16).  Four lanes hold the 8 elements of one 16-byte
                // group as (0,1) (2,3) (4,5) (6,7); the layout wants (0,4) (1,5) (2,6) (3,7): lane j takes element j from
                // lane j / 2 and element j + 4 from lane 2 + j / 2 (quad permutes), then stores one dword
                const int mine = __builtin_bit_cast(int, zz), j = lane & 3;
                const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, mine, 0x50, 0xF, 0xF, false);   // quad_perm:[0,0,1,1]
                const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, mine, 0xFA, 0xF, 0xF, false);   // quad_perm:[2,2,3,3]
                const uint32_t pair = (j & 1) ? ((lo >> 16) | (hi & 0xFFFF0000u)) : ((lo & 0xFFFFu) | (hi << 16));
                if (active) *reinterpret_cast<uint32_t*>(out16 + w4a16_xperm_offset(t, (int)e0 - 2 * j, I) + 2 * j) = pair;
            } else if (active) {
                *reinterpret_cast<f16x2*>(out16 + (size_t)t * I + e0) = zz;
            }
            return;
        }
        float amax = active ? fmaxf(__builtin_fabsf(h2f(zz[0])), __builtin_fabsf(h2f(zz[1]))) : 0.0f;
        amax = wave_max_f(amax);
        if (lane == 0) red[wave] = amax;
        __syncthreads();
        amax = xwg_row_amax(red, NW, xgen, t, b, NB, xws, reinterpret_cast<float*>(zlds) + 1024);
        const f16 sc = f2h(h2f(f2h(amax / 7.0f)) * h2f(f2h(clip)));
        const float scf = h2f(sc);
        const float rcf = 1.0f / scf;
        if (tid == 0 && b == 0) scale[t] = sc;
        // one byte per thread -> LDS -> 16-byte stores (a row's block is CB / 2 contiguous bytes of the packed row)
        unsigned char* qb = reinterpret_cast<unsigned char*>(zlds);
        if (active) {
            const int v0 = rni_sat(h2f(f2h(div3_h(h2f(zz[0]), rcf, scf))), -8, 7);
            const int v1 = rni_sat(h2f(f2h(div3_h(h2f(zz[1]), rcf, scf))), -8, 7);
            qb[row * PAIRS + jp] = (unsigned char)pack_nib(v0, v1);
        }
        __syncthreads();
        if constexpr (PAIRS >= 16) {
            constexpr int V16 = PAIRS / 16;                    // 16-byte pieces per row block
            if (tid < KH * V16) {
                const int r2 = tid / V16, pc = tid % V16;
                *reinterpret_cast<u32x4*>(q + (size_t)t * (I / 2) + ((size_t)r2 * P + b * CB) / 2 + 16 * pc) =
                    *reinterpret_cast<const u32x4*>(qb + r2 * PAIRS + 16 * pc);
            }
        } else if constexpr (PAIRS == 8) {   // a row block of 16 columns is one 8-byte piece
            if (tid < KH)
                *reinterpret_cast<uint64_t*>(q + (size_t)t * (I / 2) + ((size_t)tid * P + b * CB) / 2) =
                    *reinterpret_cast<const uint64_t*>(qb + tid * PAIRS);
        } else {
            static_assert(PAIRS == 4, "a row block of 8 columns is one 4-byte piece");
            if (tid < KH)
                *reinterpret_cast<uint32_t*>(q + (size_t)t * (I / 2) + ((size_t)tid * P + b * CB) / 2) =
                    *reinterpret_cast<const uint32_t*>(qb + tid * PAIRS);
        }
        return;
    }

    // phase B: hadK mix.  KH > 0: a thread owns a column pair and KH/IQ output rows; the column pair lives in
    // registers and the k loop is fully unrolled with 16-byte broadcast reads of hadK (a naive loop is LDS
    // latency bound: 2*K*K dependent ds_reads per lane).  KH == 0: generic K.
    const f16* zsrc = ylds;
    if (K > 1) {
        zsrc = zlds;
        if constexpr (KH > 0) {
            constexpr int IQ = (NT / (P / 2)) >= 4 ? 4 : ((NT / (P / 2)) >= 2 ? 2 : 1);
            static_assert(KH % IQ == 0 && KH % 4 == 0, "row split");
            constexpr int ROWS = KH / IQ;
            const int j2 = tid % (P / 2), iq = tid / (P / 2);
            if constexpr (P >= 512 && KH % 4 == 0) {
                // Four columns per thread (half the threads): the table reads -- 16-byte broadcasts whose 1 KB return
                // per wave is what bounds this phase -- are shared by twice as many fmas.  Per output the fp32 chain
                // in k order is unchanged.  k in quads outside, the thread's KH/4 rows inside: 2 x KH/4 accumulators.
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                constexpr int RQ = KH / 4;                     // rows per thread
                const int j4 = tid % (P / 4), rq = tid / (P / 4);
                if (rq < 4) {
                    f32x2 a0[RQ], a1[RQ];
#pragma unroll
                    for (int r = 0; r < RQ; r++) a0[r] = a1[r] = f32x2{0.0f, 0.0f};
#pragma unroll
                    for (int kq = 0; kq < KH / 4; kq++) {
                        f32x2 y0[4], y1[4];
#pragma unroll
                        for (int kk = 0; kk < 4; kk++) {
                            const f16x4 yy = *reinterpret_cast<const f16x4*>(ylds + (size_t)(4 * kq + kk) * P + 4 * j4);
                            y0[kk] = f32x2{h2f(yy[0]), h2f(yy[1])};
                            y1[kk] = f32x2{h2f(yy[2]), h2f(yy[3])};
                        }
#pragma unroll
                        for (int r = 0; r < RQ; r++) {
                            const float4 h = *reinterpret_cast<const float4*>(had + (rq * RQ + r) * KH + 4 * kq);
                            a0[r] = __builtin_elementwise_fma(f32x2{h.x, h.x}, y0[0], a0[r]);
                            a1[r] = __builtin_elementwise_fma(f32x2{h.x, h.x}, y1[0], a1[r]);
                            a0[r] = __builtin_elementwise_fma(f32x2{h.y, h.y}, y0[1], a0[r]);
                            a1[r] = __builtin_elementwise_fma(f32x2{h.y, h.y}, y1[1], a1[r]);
                            a0[r] = __builtin_elementwise_fma(f32x2{h.z, h.z}, y0[2], a0[r]);
                            a1[r] = __builtin_elementwise_fma(f32x2{h.z, h.z}, y1[2], a1[r]);
                            a0[r] = __builtin_elementwise_fma(f32x2{h.w, h.w}, y0[3], a0[r]);
                            a1[r] = __builtin_elementwise_fma(f32x2{h.w, h.w}, y1[3], a1[r]);
                        }
                    }
#pragma unroll
                    for (int r = 0; r < RQ; r++) {
                        const f16x4 zz = {f2h(a0[r][0]), f2h(a0[r][1]), f2h(a1[r][0]), f2h(a1[r][1])};
                        *reinterpret_cast<f16x4*>(zlds + (size_t)(rq * RQ + r) * P + 4 * j4) = zz;
                    }
                }
            } else
            if (iq < IQ) {
                // the column pair rides in one 64-bit register pair: v_pk_fma_f32 does both columns per instruction
                // (each component is an ordinary fp32 fma, so the k-ordered chain of the oracle is unchanged)
                typedef float f32x2 __attribute__((ext_vector_type(2)));
                f32x2 y[KH];
#pragma unroll
                for (int k = 0; k < KH; k++) {
                    f16x2 yy = *reinterpret_cast<const f16x2*>(ylds + (size_t)k * P + 2 * j2);
                    y[k] = f32x2{h2f(yy[0]), h2f(yy[1])};
                }
                for (int i = iq * ROWS; i < (iq + 1) * ROWS; i++) {
                    f32x2 a = {0.0f, 0.0f};
#pragma unroll
                    for (int kq = 0; kq < KH / 4; kq++) {
                        const float4 h = *reinterpret_cast<const float4*>(had + i * KH + 4 * kq);
                        a = __builtin_elementwise_fma(f32x2{h.x, h.x}, y[4 * kq + 0], a);
                        a = __builtin_elementwise_fma(f32x2{h.y, h.y}, y[4 * kq + 1], a);
                        a = __builtin_elementwise_fma(f32x2{h.z, h.z}, y[4 * kq + 2], a);
                        a = __builtin_elementwise_fma(f32x2{h.w, h.w}, y[4 * kq + 3], a);
                    }
                    f16x2 zz = {f2h(a[0]), f2h(a[1])};
                    *reinterpret_cast<f16x2*>(zlds + (size_t)i * P + 2 * j2) = zz;
                }
            }
        } else {
            // generic table (had108, had40, ...): (output row, column pair) items over all threads -- with the column
            // pairs alone only P/2 threads would work (64 of 1024 for 13824 = 108 x 128)
            for (int item = tid; item < K * (P / 2); item += NT) {
                const int i = item / (P / 2), j2 = item - i * (P / 2);
                float a0 = 0.0f, a1 = 0.0f;
                for (int k = 0; k < K; k++) {
                    float h = had[i * K + k];
                    f16x2 yy = *reinterpret_cast<const f16x2*>(ylds + (size_t)k * P + 2 * j2);
                    a0 = __builtin_fmaf(h, h2f(yy[0]), a0);
                    a1 = __builtin_fmaf(h, h2f(yy[1]), a1);
                }
                f16x2 zz = {f2h(a0), f2h(a1)};
                *reinterpret_cast<f16x2*>(zlds + (size_t)i * P + 2 * j2) = zz;
            }
        }
        __syncthreads();
    }

    QS_ST2(3);
    // phase C: fp16 out, or abs-max + int4
    const int nvec = I / 8;
    if (q == nullptr) {
        for (int i = tid; i < nvec; i += NT) {
            const f16x8 z8 = *reinterpret_cast<const f16x8*>(zsrc + 8 * i);
            if (xp)   // fragment-major tiles (<= 32 tokens): the eight halves k .. k + 7 are one 16-byte group in the order 0,4,1,5,2,6,3,7
                *reinterpret_cast<f16x8*>(out16 + w4a16_xperm_offset(t, 8 * i, I)) = f16x8{z8[0], z8[4], z8[1], z8[5], z8[2], z8[6], z8[3], z8[7]};
            else
                *reinterpret_cast<f16x8*>(out16 + (size_t)t * I + 8 * i) = z8;
        }
        return;
    }
    float amax = 0.0f;
    for (int i = tid; i < nvec; i += NT) {
        f16x8 a = *reinterpret_cast<const f16x8*>(zsrc + 8 * i);
#pragma unroll
        for (int c = 0; c < 8; c++) {
            float f = __builtin_fabsf(h2f(a[c]));
            amax = f > amax ? f : amax;
        }
    }
    amax = wave_max_f(amax);
    if (lane == 0) red[wave] = amax;
    __syncthreads();
    amax = red[0];
#pragma unroll
    for (int w = 1; w < NW; w++) amax = fmaxf(amax, red[w]);
    const f16 sc = f2h(h2f(f2h(amax / 7.0f)) * h2f(f2h(clip)));
    const float scf = h2f(sc);
    const float rcf = 1.0f / scf;   // correctly rounded reciprocal for div3_h
    if (tid == 0) scale[t] = sc;
    for (int i = tid; i < nvec; i += NT) {
        f16x8 a = *reinterpret_cast<const f16x8*>(zsrc + 8 * i);
        uint32_t w = 0;
#pragma unroll
        for (int c = 0; c < 8; c++) {
            int v = rni_sat(h2f(f2h(div3_h(h2f(a[c]), rcf, scf))), -8, 7);
            w |= (uint32_t)(v & 0xF) << (4 * c);
        }
        *reinterpret_cast<uint32_t*>(q + (size_t)t * (I / 2) + 4 * i) = w;
    }
#ifdef QS_SMH_STAMPS
    QS_ST2(4);
    if (tid == 0 && t == 0) {  // debug: stamps land in the first bytes of the NEXT token's int4 row region is unsafe; use scale[64..]
        long long* sb = reinterpret_cast<long long*>(scale + 64);
        sb[0] = stp[2] - stp[1]; sb[1] = stp[3] - stp[2]; sb[2] = stp[4] - stp[3]; sb[3] = stp[1];  sb[4] = stp[4];
    }
#endif
}

size_t xwg_workspace_bytes() { return (16 + QS_XWG_MAX_TOKENS) * 4 + (size_t)QS_XWG_MAX_TOKENS * QS_XWG_MAX_NB * 8; }

// Workgroups per token of the spread form for this shape (1 = the one-workgroup form).
static int smh_spread(int T, int P, int K, int pre_activated, const void* xws) {
    if (!xws || !pre_activated || T > QS_XWG_MAX_TOKENS) return 1;
    int nb = 1;
    if (K == 28) nb = P == 512 ? 8 : (P == 1024 ? 16 : 1);
    if (K == 108 && P == 128) {            // Llama-2-13B: 13824 = had108 x H128.  The 108-term mix is what bounds a workgroup
        const int nb108 = 16;   // (measured: 8 workgroups per token 13.16 ms per cycle, 16: 13.06)
        nb = (nb108 == 8 || T * 16 > 256) ? 8 : 16;
    }
    return T * nb <= 256 ? nb : 1;     // every workgroup of a token resident: at most one workgroup per CU
}

// the forms that can write the fragment-major tile: the spread ones (a workspace is given), at most 16 tokens
bool mlp_hadamard_xperm_supported(int T, int I, int K) {
    if (T < 1 || T > 32 || K < 1 || K > 172 || I % K) return false;
    const int P = I / K;
    if (P & (P - 1)) return false;
    static const int dummy = 0;
    if (smh_spread(T, P, K, 1, &dummy) > 1) return true;      // the spread forms
    return T > 16 && (P == 128 || P == 256 || P == 512 || P == 1024 || P == 2048) && I % 128 == 0;   // 17..32 tokens: also the one-workgroup form
}

int silu_mul_hadamard(const f16* gate_up, const f16* hadK, f16* out_f16, int8_t* q, f16* scale, float had_scale,
                      float clip, int T, int I, int K, int pre_activated, void* xws, hipStream_t st, int xp) {
    if (T == 0) return 0;
    if (K < 1 || K > 172 || I % K) return -1;
    const int P = I / K;
    if (P & (P - 1)) return -1;
    size_t lds = 64 + (size_t)I * 2 * (K > 1 ? 2 : 1) + (K > 1 ? (size_t)K * K * 4 : 0);
    if (lds > 160 * 1024 - 64) return -2;
    const int nb = smh_spread(T, P, K, pre_activated, xws);
#define QS_SMH_SPREAD(EPLV, NBV)                                                                                 \
    if (nb == NBV && P == EPLV * 64) {                                                                           \
        if (lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&silu_mul_hadamard_kernel<EPLV, 28, true, NBV>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
        hipLaunchKernelGGL((silu_mul_hadamard_kernel<EPLV, 28, true, NBV>), dim3(T * NBV), dim3(QS_SMH_THREADS), lds, st, \
                           gate_up, hadK, out_f16, q, scale, had_scale, clip, I, K, reinterpret_cast<uint32_t*>(xws), xp); \
        return 0;                                                                                                \
    }
    QS_SMH_SPREAD(8, 8) QS_SMH_SPREAD(16, 16)
#undef QS_SMH_SPREAD
#define QS_SMH_108(NBV)                                                                                          \
    if (nb == NBV && K == 108 && P == 128) {   /* 16-byte lanes, four rows of 128 per wave trip (PW = 128) */       \
        if (lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&silu_mul_hadamard_kernel<8, 108, true, NBV, 128>), \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
        hipLaunchKernelGGL((silu_mul_hadamard_kernel<8, 108, true, NBV, 128>), dim3(T * NBV), dim3(QS_SMH_THREADS), lds, st, \
                           gate_up, hadK, out_f16, q, scale, had_scale, clip, I, K, reinterpret_cast<uint32_t*>(xws), xp); \
        return 0;                                                                                                \
    }
    QS_SMH_108(8) QS_SMH_108(16)
#undef QS_SMH_108
    if (xp && (q != nullptr || T > 32)) return -1;   // fragment-major fp16 rows: one or two 16-row tiles
#define QS_SMH2(EPLV, KHV)                                                                                      \
    {                                                                                                            \
        if (pre_activated) QS_SMH3(EPLV, KHV, true)                                                              \
        QS_SMH3(EPLV, KHV, false)                                                                                \
    }
#define QS_SMH3(EPLV, KHV, PAV)                                                                                 \
    {                                                                                                            \
        if (lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&silu_mul_hadamard_kernel<EPLV, KHV, PAV>),  \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                     \
        hipLaunchKernelGGL((silu_mul_hadamard_kernel<EPLV, KHV, PAV>), dim3(T), dim3(QS_SMH_THREADS), lds, st, gate_up, \
                           hadK, out_f16, q, scale, had_scale, clip, I, K, (uint32_t*)nullptr, xp);              \
        return 0;                                                                                                \
    }
#define QS_SMH(EPLV)                                                                                            \
    if (P == EPLV * 64) {                                                                                        \
        if (K == 28) QS_SMH2(EPLV, 28)                                                                           \
        if (K == 44) QS_SMH2(EPLV, 44)                                                                           \
        if (K == 12) QS_SMH2(EPLV, 12)                                                                           \
        QS_SMH2(EPLV, 0)                                                                                         \
    }
    QS_SMH(2) QS_SMH(4) QS_SMH(8) QS_SMH(16) QS_SMH(32)
#undef QS_SMH
#undef QS_SMH2
#undef QS_SMH3
    return -1;
}

}  // namespace qspec
