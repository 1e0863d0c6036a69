/*
 * qspec_oracle.c -- CPU restatement of the QSpec draft/verify hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under qspec_amd/ may import, link or
 * execute this file; it is the checker for tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg, never the product path.
 *
 * Every function cites the reference file:line (relative to /root/reference)
 * whose arithmetic it restates.  The reference kernels are CUDA and cannot be
 * built here (no nvcc, CUTLASS / bitblas / flash-attn un-vendored), so this is
 * a restatement in plain C with IEEE-754 binary32/binary16 semantics:
 *   - fp16 values travel as uint16_t bit patterns; h2f/f2h are exact
 *     round-to-nearest-even conversions;
 *   - every fp32 operation is a single correctly rounded IEEE operation
 *     (compile with -ffp-contract=off); fused multiply-adds are written
 *     explicitly as fmaf() where the reference's nvcc build contracts them;
 *   - where the reference is built with --use_fast_math (approximate div /
 *     rsqrt, third-party/kernels/setup.py) the oracle uses the correctly
 *     rounded operation: those steps are "parity unpinned" (no reference test
 *     or golden vector exists for them).
 *
 * Pinned against the reference (tests/test_oracle_golden.py):
 *   int4 pack/unpack  <- third-party/QuaRot/quarot/functional/quantization.py:42-75
 *   WHT + hadK mix    <- quarot/functional/hadamard.py:59-80 (matmul_hadU), scipy hadamard
 *   W4A4 GEMM         <- third-party/ao/test/test_rowwise_scaled_linear_cutlass.py:64-84
 *   rejection sampler <- vllm/model_executor/layers/rejection_sampler.py (imported, run on CPU)
 * Parity unpinned (restated from CUDA source only): LN+int4 quant, row-absmax
 * quant, W4A16 GEMM (BitBLAS, un-vendored), attention (vllm_flash_attn).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------ fp16 */

static inline float h2f(uint16_t h) {
    uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t exp = (h >> 10) & 0x1Fu;
    uint32_t man = h & 0x3FFu;
    uint32_t bits;
    if (exp == 0) {
        if (man == 0) {
            bits = sign;
        } else { /* subnormal: normalise */
            int e = -1;
            do { man <<= 1; e++; } while (!(man & 0x400u));
            man &= 0x3FFu;
            bits = sign | ((uint32_t)(127 - 15 - e) << 23) | (man << 13);
        }
    } else if (exp == 31) {
        bits = sign | 0x7F800000u | (man << 13);
    } else {
        bits = sign | ((exp + 127 - 15) << 23) | (man << 13);
    }
    float f;
    memcpy(&f, &bits, 4);
    return f;
}

static inline uint16_t f2h(float f) {
    uint32_t x;
    memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7FFFFFFFu;
    if (ax >= 0x7F800000u) { /* inf / nan */
        if (ax > 0x7F800000u) return (uint16_t)(sign | 0x7E00u | ((ax >> 13) & 0x3FFu));
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax >= 0x477FF000u) { /* >= 65520 rounds to inf */
        return (uint16_t)(sign | 0x7C00u);
    }
    if (ax < 0x33000001u) { /* <= 2^-25 rounds to zero (2^-25 ties to even = 0) */
        return (uint16_t)sign;
    }
    int32_t e = (int32_t)(ax >> 23) - 127;
    uint32_t man = (ax & 0x7FFFFFu) | 0x800000u;
    if (e < -14) { /* subnormal result */
        int shift = 13 + (-14 - e); /* total right shift of 24-bit mantissa to 10-bit frac */
        uint32_t q = man >> shift;
        uint32_t rem = man & ((1u << shift) - 1u);
        uint32_t half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1u))) q++;
        return (uint16_t)(sign | q);
    }
    uint32_t q = ((uint32_t)(e + 15) << 10) | ((man >> 13) & 0x3FFu);
    uint32_t rem = man & 0x1FFFu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++; /* carry into exponent is correct */
    return (uint16_t)(sign | q);
}

void qo_h2f(const uint16_t* h, float* f, int64_t n) { for (int64_t i = 0; i < n; i++) f[i] = h2f(h[i]); }
void qo_f2h(const float* f, uint16_t* h, int64_t n) { for (int64_t i = 0; i < n; i++) h[i] = f2h(f[i]); }

/* round-to-nearest-even float -> int with saturation to [lo,hi]; NaN -> 0
 * (PTX cvt.rni.sat.s8.f32 / cvt.rni.s32.f16: third-party/kernels/csrc/utils.cuh:225-237) */
static inline int rni_sat(float v, int lo, int hi) {
    if (v != v) return 0;
    float r = nearbyintf(v); /* default rounding mode: to nearest even */
    if (r < (float)lo) return lo;
    if (r > (float)hi) return hi;
    return (int)r;
}

/* Deterministic expf shared bit-for-bit with the HIP kernels
 * (qspec_amd/csrc/common.cuh: qexpf).  Cody-Waite reduction + degree-6
 * polynomial (Cephes coefficients), only fmaf / rint / ldexp: max error ~1 ulp.
 * Replaces CUDA's fast-math expf inside SiLU / softmax, which is not
 * reproducible off NVIDIA hardware. */
static inline float qexpf(float x) {
    if (x != x) return x;
    if (x > 88.0f) return INFINITY;
    if (x < -86.0f) return 0.0f;
    float n = nearbyintf(x * 1.44269504088896341f);
    float r = fmaf(n, -0.693145751953125f, x);
    r = fmaf(n, -1.42860682030941723212e-6f, r);
    float p = 1.9875691500e-4f;
    p = fmaf(p, r, 1.3981999507e-3f);
    p = fmaf(p, r, 8.3334519073e-3f);
    p = fmaf(p, r, 4.1665795894e-2f);
    p = fmaf(p, r, 1.6666665459e-1f);
    p = fmaf(p, r, 5.0000001201e-1f);
    float z = r * r;
    float y = fmaf(p, z, r) + 1.0f;
    return ldexpf(y, (int)n);
}
void qo_expf(const float* x, float* y, int64_t n) { for (int64_t i = 0; i < n; i++) y[i] = qexpf(x[i]); }

/* ------------------------------------------------------------ int4 pack  */
/* byte j = (q[2j] & 0xF) | (q[2j+1] << 4), two's complement nibbles
 * (quarot/functional/quantization.py:42-49, quant.cu:154-165,
 *  layernorm_kernels.cu:700-710, ao test_rowwise_scaled_linear_cutlass.py:50,60) */
void qo_pack_i4(const int8_t* q, int8_t* packed, int64_t rows, int64_t cols) {
    for (int64_t r = 0; r < rows; r++)
        for (int64_t j = 0; j < cols / 2; j++) {
            int lo = q[r * cols + 2 * j] & 0xF, hi = q[r * cols + 2 * j + 1] & 0xF;
            packed[r * (cols / 2) + j] = (int8_t)(lo | (hi << 4));
        }
}
static inline int nib_lo(int8_t b) { int v = b & 0xF; return v >= 8 ? v - 16 : v; }
static inline int nib_hi(int8_t b) { int v = (b >> 4) & 0xF; return v >= 8 ? v - 16 : v; }
void qo_unpack_i4(const int8_t* packed, int8_t* q, int64_t rows, int64_t cols) {
    for (int64_t r = 0; r < rows; r++)
        for (int64_t j = 0; j < cols / 2; j++) {
            int8_t b = packed[r * (cols / 2) + j];
            q[r * cols + 2 * j] = (int8_t)nib_lo(b);
            q[r * cols + 2 * j + 1] = (int8_t)nib_hi(b);
        }
}

/* ------------------------------------------- block reductions (LN kernel) */
/* blockReduceSum / blockAllReduceSum: third-party/kernels/csrc/reduction_utils.cuh:24-82.
 * warp butterfly xor 16,8,4,2,1 (every lane ends with the same bits because
 * a+b == b+a), lane 0 of each warp -> shared[wid], then lanes < nwarps re-load
 * and butterfly again.  partial[] has B entries (B = blockDim.x, multiple of 32). */
static float warp_butterfly_sum(float v[32]) {
    float t[32];
    for (int mask = 16; mask > 0; mask >>= 1) {
        for (int i = 0; i < 32; i++) t[i] = v[i] + v[i ^ mask];
        memcpy(v, t, sizeof(t));
    }
    return v[0];
}
static float block_reduce_sum(const float* partial, int B) {
    int nw = B / 32;
    float shared[32];
    for (int w = 0; w < 32; w++) shared[w] = 0.0f;
    for (int w = 0; w < nw; w++) {
        float v[32];
        memcpy(v, partial + 32 * w, sizeof(v));
        shared[w] = warp_butterfly_sum(v);
    }
    float v[32];
    for (int l = 0; l < 32; l++) v[l] = (l < nw) ? shared[l] : 0.0f;
    return warp_butterfly_sum(v);
}

/* LayerNorm-without-gamma ("RMSNorm" in the reference) + per-token int4 quant.
 * generalLayerNorm_fuse_sum_i4<half, at::Half>, third-party/kernels/csrc/layernorm_kernels.cu:569-716,
 * host launcher :890-923 (block = min(H,1024) rounded up to 32, T = half so one
 * element per loop trip, use_shmem = false).  Python caller:
 * vllm/model_executor/layers/quarot_nn/normalization.py:34-65.
 * mode 0: q/scale/input_sum outputs; mode 1: fp16 output (:927-957, normalization.py:67-82). */
static void ln_row(const uint16_t* x, int H, float eps, int mode, int8_t* q, uint16_t* scale,
                   uint16_t* input_sum, uint16_t* out) {
    int B = H < 1024 ? H : 1024;
    B = 32 * ((B + 31) / 32);
    float* part = (float*)calloc((size_t)B, sizeof(float));
    /* pass 1: mean (layernorm_kernels.cu:596-631) */
    for (int t = 0; t < B; t++) {
        float s = 0.0f;
        for (int i = t; i < H; i += B) s += h2f(x[i]);
        part[t] = s;
    }
    float mean = block_reduce_sum(part, B) / (float)H;
    /* pass 2: centred variance (:634-649); nvcc contracts diff*diff + acc into an FMA */
    for (int t = 0; t < B; t++) {
        float s = 0.0f;
        for (int i = t; i < H; i += B) {
            float d = h2f(x[i]) - mean;
            s = fmaf(d, d, s);
        }
        part[t] = s;
    }
    float var = block_reduce_sum(part, B);
    float rstd = 1.0f / sqrtf(var / (float)H + eps); /* rsqrtf, fast-math in the reference */
    if (mode == 1) {
        for (int i = 0; i < H; i++) out[i] = f2h((h2f(x[i]) - mean) * rstd);
        free(part);
        return;
    }
    /* pass 3: fp16 amax of the ROUNDED values and fp16 per-thread sums (:651-684) */
    uint16_t amax_h = f2h(1e-6f);
    float amax = h2f(amax_h);
    for (int t = 0; t < B; t++) {
        uint16_t s16 = 0;
        for (int i = t; i < H; i += B) {
            uint16_t v = f2h((h2f(x[i]) - mean) * rstd);
            float a = fabsf(h2f(v));
            if (a > amax) amax = a; /* NaN never wins, as (val1 > val2) ? val1 : val2 */
            s16 = f2h(h2f(s16) + h2f(v));
        }
        part[t] = h2f(s16);
    }
    float sum_f = block_reduce_sum(part, B);
    /* pass 4: quantise the UN-rounded fp32 value (:686-710) */
    float s = 7.0f / amax;
    for (int j = 0; j < H / 2; j++) {
        int qq[2];
        for (int e = 0; e < 2; e++) {
            float v = ((h2f(x[2 * j + e]) - mean) * rstd) * s;
            v = fmaxf(fminf(v, 7.0f), -8.0f); /* clamp_int4 :33-36 */
            qq[e] = rni_sat(v, -128, 127);
        }
        q[j] = (int8_t)((qq[0] & 0x0F) | ((qq[1] & 0x0F) << 4));
    }
    *scale = f2h(amax / 7.0f);
    *input_sum = f2h(sum_f);
    free(part);
}

void qo_ln_quant_i4(const uint16_t* x, int T, int H, float eps, int8_t* q, uint16_t* scale,
                    uint16_t* input_sum) {
#pragma omp parallel for
    for (int t = 0; t < T; t++)
        ln_row(x + (int64_t)t * H, H, eps, 0, q + (int64_t)t * (H / 2), scale + t, input_sum + t, NULL);
}
void qo_ln_fp16(const uint16_t* x, int T, int H, float eps, uint16_t* out) {
#pragma omp parallel for
    for (int t = 0; t < T; t++) ln_row(x + (int64_t)t * H, H, eps, 1, NULL, NULL, NULL, out + (int64_t)t * H);
}

/* Row abs-max int4 quantiser: rowAbsMaxQuantizeKernel,
 * third-party/QuaRot/quarot/kernels/quant.cu:102-167 (block 256, bindings.cpp:128-147);
 * Python: quarot/__init__.py:119-144, quarot_nn/quantization.py:13-19.  All fp16:
 * scale = h(h(amax/7) * h(clip)); q = clamp(half2int_rn(h(x/scale)), -8, 7). */
void qo_rowabsmax_quant_i4(const uint16_t* x, int T, int K, float clip, int8_t* q, uint16_t* scale) {
#pragma omp parallel for
    for (int t = 0; t < T; t++) {
        const uint16_t* xr = x + (int64_t)t * K;
        float amax = 0.0f;
        for (int i = 0; i < K; i++) {
            float a = fabsf(h2f(xr[i]));
            if (a > amax) amax = a; /* __hmax drops NaN */
        }
        uint16_t sc = f2h(h2f(f2h(amax / 7.0f)) * h2f(f2h(clip)));
        scale[t] = sc;
        float scf = h2f(sc);
        for (int j = 0; j < K / 2; j++) {
            int qq[2];
            for (int e = 0; e < 2; e++) {
                float d = h2f(f2h(h2f(xr[2 * j + e]) / scf)); /* __hdiv */
                int v = rni_sat(d, -2147483647 - 1, 2147483647);
                qq[e] = v < -8 ? -8 : (v > 7 ? 7 : v);
            }
            q[(int64_t)t * (K / 2) + j] = (int8_t)((qq[0] & 0x0F) | ((qq[1] & 0x0F) << 4));
        }
    }
}

/* Fast Walsh-Hadamard transform, Sylvester order, fp32 butterflies in
 * increasing stride (in-thread bits, then lane bits, then chunk bits =
 * element-index bits 0..logN-1 in order), out = h(acc * scale):
 * third-party/fast-hadamard-transform/csrc/fast_hadamard_transform_cuda.cu:124-198,
 * fast_hadamard_transform_common.h:86-161. */
void qo_fwht(const uint16_t* x, int64_t rows, int N, float scale, uint16_t* out) {
#pragma omp parallel for
    for (int64_t r = 0; r < rows; r++) {
        float* v = (float*)malloc(sizeof(float) * (size_t)N);
        for (int i = 0; i < N; i++) v[i] = h2f(x[r * N + i]);
        for (int stride = 1; stride < N; stride <<= 1)
            for (int i = 0; i < N; i++)
                if (!(i & stride)) {
                    float a = v[i], b = v[i + stride];
                    v[i] = a + b;
                    v[i + stride] = a - b;
                }
        for (int i = 0; i < N; i++) out[r * N + i] = f2h(v[i] * scale);
        free(v);
    }
}

/* hadK mix: z[t,i,j] = h( sum_k f(hadK[i,k]) * f(y[t,k,j]) ), fp32 accumulate in k order.
 * quarot/functional/hadamard.py:104-108 (`hadK @ out.view(-1, K, n//K)`, fp16 GEMM). */
void qo_hadk_mix(const uint16_t* y, const uint16_t* hadK, int T, int K, int M, uint16_t* z) {
#pragma omp parallel for
    for (int t = 0; t < T; t++)
        for (int i = 0; i < K; i++)
            for (int j = 0; j < M; j++) {
                float acc = 0.0f;
                for (int k = 0; k < K; k++)
                    acc = fmaf(h2f(hadK[i * K + k]), h2f(y[((int64_t)t * K + k) * M + j]), acc);
                z[((int64_t)t * K + i) * M + j] = f2h(acc);
            }
}

/* SiLU(gate) * up on the fused gate_up output: up = [:, :I], gate = [:, I:]
 * (vllm/model_executor/models/quarot_llama.py:279-284).  fp16 tensor ops:
 * a = h(silu_f32(gate)); g = h(a * up). */
void qo_silu_mul(const uint16_t* gate_up, int T, int I, uint16_t* out) {
#pragma omp parallel for
    for (int t = 0; t < T; t++)
        for (int i = 0; i < I; i++) {
            float up = h2f(gate_up[(int64_t)t * 2 * I + i]);
            float g = h2f(gate_up[(int64_t)t * 2 * I + I + i]);
            float a = h2f(f2h(g / (1.0f + qexpf(-g))));
            out[(int64_t)t * I + i] = f2h(a * up);
        }
}

/* fp16 residual add: hidden = residual + delta (quarot_llama.py:380,390) */
void qo_add_f16(const uint16_t* a, const uint16_t* b, uint16_t* out, int64_t n) {
    for (int64_t i = 0; i < n; i++) out[i] = f2h(h2f(a[i]) + h2f(b[i]));
}

/* W4A4 GEMM: acc int32 exact; out = h((f(acc)*f(sa[m]))*f(sw[n]) (+ f(bias[n]))).
 * third-party/ao/torchao/csrc/cuda/rowwise_scaled_linear_cutlass/
 * rowwise_scaled_linear_cutlass_unified.cuh:342-377 (EVT epilogue), reference formula
 * third-party/ao/test/test_rowwise_scaled_linear_cutlass.py:64-84. */
void qo_gemm_w4a4(const int8_t* xq, const uint16_t* xs, const int8_t* wq, const uint16_t* ws,
                  const uint16_t* bias, uint16_t* out, int M, int N, int K) {
    int Kb = K / 2;
    int8_t* xu = (int8_t*)malloc((size_t)M * K);
    qo_unpack_i4(xq, xu, M, K);
#pragma omp parallel
    {
        int8_t* wu = (int8_t*)malloc((size_t)K);
#pragma omp for schedule(static)
        for (int n = 0; n < N; n++) {
            for (int j = 0; j < Kb; j++) {
                int8_t b = wq[(int64_t)n * Kb + j];
                wu[2 * j] = (int8_t)nib_lo(b);
                wu[2 * j + 1] = (int8_t)nib_hi(b);
            }
            float swn = h2f(ws[n]);
            for (int m = 0; m < M; m++) {
                const int8_t* xr = xu + (int64_t)m * K;
                int32_t acc = 0;
                for (int k = 0; k < K; k++) acc += (int32_t)xr[k] * (int32_t)wu[k];
                float v = ((float)acc * h2f(xs[m])) * swn;
                if (bias) v = v + h2f(bias[n]);
                out[(int64_t)m * N + n] = f2h(v);
            }
        }
        free(wu);
    }
    free(xu);
}

/* W4A16 GEMM (oracle definition, tolerance 1e-3): out = h((sum_k f(x[m,k])*w[n,k]) * f(sw[n]) (+bias)).
 * Reference: bitblas.Matmul (un-vendored pip dependency, version unpinned), config
 * vllm/model_executor/layers/quarot_nn/linear.py:169-185, call :122 (weight ^ 0x88 =
 * offset-binary view of the same nibbles).  Second statements of the same semantics:
 * quarot_nn/qspec_gemm.py:20-88 (fp32 accumulate) and linear.py:111-119.
 * The sum is accumulated in fp64 so that any fp32 summation order lands within
 * the stated tolerance of it. */
void qo_gemm_w4a16(const uint16_t* x, const int8_t* wq, const uint16_t* ws, const uint16_t* bias,
                   uint16_t* out, int M, int N, int K) {
    int Kb = K / 2;
    float* xf = (float*)malloc(sizeof(float) * (size_t)M * K);
    for (int64_t i = 0; i < (int64_t)M * K; i++) xf[i] = h2f(x[i]);
#pragma omp parallel
    {
        float* wu = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
        for (int n = 0; n < N; n++) {
            for (int j = 0; j < Kb; j++) {
                int8_t b = wq[(int64_t)n * Kb + j];
                wu[2 * j] = (float)nib_lo(b);
                wu[2 * j + 1] = (float)nib_hi(b);
            }
            float swn = h2f(ws[n]);
            for (int m = 0; m < M; m++) {
                const float* xr = xf + (int64_t)m * K;
                double acc = 0.0;
                for (int k = 0; k < K; k++) acc += (double)xr[k] * (double)wu[k];
                float v = (float)acc * swn;
                if (bias) v = v + h2f(bias[n]);
                out[(int64_t)m * N + n] = f2h(v);
            }
        }
        free(wu);
    }
    free(xf);
}

/* The same W4A16 GEMM as a SECOND admissible implementation: fp32 accumulation in 32 interleaved partial sums
 * (k mod 32) combined pairwise -- the kind of order a tiled kernel produces.  Used only to measure the noise floor
 * between two implementations that both meet the 1e-3 bar stage by stage (tests/test_model_gpu.py full-depth test):
 * how far apart 32 layers of fp16 roundings carry two such implementations. */
void qo_gemm_w4a16_f32acc(const uint16_t* x, const int8_t* wq, const uint16_t* ws, uint16_t* out, int M, int N, int K) {
    int Kb = K / 2;
    float* xf = (float*)malloc(sizeof(float) * (size_t)M * K);
    for (int64_t i = 0; i < (int64_t)M * K; i++) xf[i] = h2f(x[i]);
#pragma omp parallel
    {
        float* wu = (float*)malloc(sizeof(float) * (size_t)K);
#pragma omp for schedule(static)
        for (int n = 0; n < N; n++) {
            for (int j = 0; j < Kb; j++) {
                int8_t b = wq[(int64_t)n * Kb + j];
                wu[2 * j] = (float)nib_lo(b);
                wu[2 * j + 1] = (float)nib_hi(b);
            }
            float swn = h2f(ws[n]);
            for (int m = 0; m < M; m++) {
                const float* xr = xf + (int64_t)m * K;
                float part[32];
                for (int j = 0; j < 32; j++) part[j] = 0.0f;
                int k = 0;
                for (; k + 32 <= K; k += 32)
                    for (int j = 0; j < 32; j++) part[j] = fmaf(xr[k + j], wu[k + j], part[j]);
                for (; k < K; k++) part[k & 31] = fmaf(xr[k], wu[k], part[k & 31]);
                for (int w = 16; w >= 1; w >>= 1)
                    for (int j = 0; j < w; j++) part[j] = part[j] + part[j + w];
                out[(int64_t)m * N + n] = f2h(part[0] * swn);
            }
        }
        free(wu);
    }
    free(xf);
}

/* fp16 x fp16^T GEMM with fp32 result rounded to fp16 (lm_head nn.Linear,
 * vllm/model_executor/layers/logits_processor.py:92-97).  fp64 accumulate. */
void qo_gemm_f16(const uint16_t* x, const uint16_t* w, uint16_t* out, int M, int N, int K) {
    float* xf = (float*)malloc(sizeof(float) * (size_t)M * K);
    for (int64_t i = 0; i < (int64_t)M * K; i++) xf[i] = h2f(x[i]);
#pragma omp parallel for schedule(static)
    for (int n = 0; n < N; n++) {
        for (int m = 0; m < M; m++) {
            double acc = 0.0;
            const uint16_t* wr = w + (int64_t)n * K;
            const float* xr = xf + (int64_t)m * K;
            for (int k = 0; k < K; k++) acc += (double)xr[k] * (double)h2f(wr[k]);
            out[(int64_t)m * N + n] = f2h((float)acc);
        }
    }
    free(xf);
}

/* NeoX rotary embedding in place on q [T, nq, d] and k [T, nk, d], every
 * operator rounding to fp16 (c10::Half arithmetic): csrc/pos_encoding_kernels.cu:10-35,71-92.
 * cos_sin_cache [max_pos, rot_dim] fp16 = [cos(rot/2) | sin(rot/2)]. */
static void rope_head(uint16_t* arr, const uint16_t* cs, int embed) {
    for (int i = 0; i < embed; i++) {
        float c = h2f(cs[i]), s = h2f(cs[embed + i]);
        float x = h2f(arr[i]), y = h2f(arr[embed + i]);
        arr[i] = f2h(h2f(f2h(x * c)) - h2f(f2h(y * s)));
        arr[embed + i] = f2h(h2f(f2h(y * c)) + h2f(f2h(x * s)));
    }
}
void qo_rope_neox(const int64_t* positions, uint16_t* q, uint16_t* k, const uint16_t* cos_sin_cache, int T,
                  int nq, int nk, int head_size, int rot_dim, int64_t q_stride, int64_t k_stride) {
    for (int t = 0; t < T; t++) {
        const uint16_t* cs = cos_sin_cache + positions[t] * rot_dim;
        for (int h = 0; h < nq; h++) rope_head(q + t * q_stride + (int64_t)h * head_size, cs, rot_dim / 2);
        for (int h = 0; h < nk; h++) rope_head(k + t * k_stride + (int64_t)h * head_size, cs, rot_dim / 2);
    }
}

/* Paged causal attention over an fp16 KV cache laid out
 * [num_blocks, block_size, n_kv, d] (csrc/cache_kernels.cu:207-247 layout;
 * call sites vllm/attention/backends/flash_attn.py:741-830).  Query token t
 * of sequence s sits at absolute position ctx_len[s] - q_len[s] + i and
 * attends to positions <= its own.  fp64 accumulate, softmax via qexpf in
 * fp32, P kept in fp32 (flash-attn rounds P to fp16: unpinned, tolerance). */
void qo_paged_attention(const uint16_t* q, const uint16_t* kc, const uint16_t* vc, const int32_t* block_tables,
                        int max_blocks, const int32_t* ctx_lens, const int32_t* q_start, int n_seqs, int nq,
                        int nkv, int d, int block_size, float sm_scale, uint16_t* out) {
    int group = nq / nkv;
#pragma omp parallel for collapse(2)
    for (int s = 0; s < n_seqs; s++)
        for (int h = 0; h < nq; h++) {
            int qlen = q_start[s + 1] - q_start[s];
            int ctx = ctx_lens[s];
            float* sc = (float*)malloc(sizeof(float) * (size_t)(ctx > 0 ? ctx : 1));
            int kvh = h / group;
            for (int i = 0; i < qlen; i++) {
                int tok = q_start[s] + i;
                int pos = ctx - qlen + i;
                const uint16_t* qv = q + ((int64_t)tok * nq + h) * d;
                float mx = -INFINITY;
                for (int p = 0; p <= pos; p++) {
                    int64_t slot = (int64_t)block_tables[s * max_blocks + p / block_size] * block_size + p % block_size;
                    const uint16_t* kv = kc + (slot * nkv + kvh) * d;
                    double acc = 0.0;
                    for (int e = 0; e < d; e++) acc += (double)h2f(qv[e]) * (double)h2f(kv[e]);
                    sc[p] = (float)acc * sm_scale;
                    if (sc[p] > mx) mx = sc[p];
                }
                double den = 0.0;
                for (int p = 0; p <= pos; p++) {
                    sc[p] = qexpf(sc[p] - mx);
                    den += sc[p];
                }
                for (int e = 0; e < d; e++) {
                    double acc = 0.0;
                    for (int p = 0; p <= pos; p++) {
                        int64_t slot = (int64_t)block_tables[s * max_blocks + p / block_size] * block_size + p % block_size;
                        acc += (double)sc[p] * (double)h2f(vc[(slot * nkv + kvh) * d + e]);
                    }
                    out[((int64_t)tok * nq + h) * d + e] = f2h((float)(acc / den));
                }
            }
            free(sc);
        }
}

/* Greedy sampler front end: probs = softmax_fp32(float(logits)), token = argmax
 * (first index on ties).  vllm/model_executor/layers/sampler.py:270-287 with
 * modify_greedy_probs hard-wired to False (:293), temperature 0 -> 1.0
 * (vllm/model_executor/sampling_metadata.py:413-417).  exp via qexpf, sum in fp64. */
void qo_softmax_argmax(const uint16_t* logits, int T, int V, float* probs, int64_t* token) {
#pragma omp parallel for
    for (int t = 0; t < T; t++) {
        const uint16_t* l = logits + (int64_t)t * V;
        float mx = -INFINITY;
        int64_t am = 0;
        for (int v = 0; v < V; v++) {
            float f = h2f(l[v]);
            if (f > mx) { mx = f; am = v; }
        }
        double den = 0.0;
        float* p = probs + (int64_t)t * V;
        for (int v = 0; v < V; v++) { p[v] = qexpf(h2f(l[v]) - mx); den += p[v]; }
        float inv = (float)den;
        for (int v = 0; v < V; v++) p[v] = p[v] / inv;
        token[t] = am;
    }
}

/* Test support for qspec_amd/csrc/common.cuh:div3_h -- the three-instruction form of h(x / s) used by the HIP
 * quantisers (x, s fp16 values): counts the pairs (s in [s_lo, s_hi) as fp16 bit patterns, every non-negative finite
 * fp16 x; the expression is odd in x) whose fp16 result differs from h(fl32(x / s)), the reference's __hdiv
 * (third-party/QuaRot/quarot/kernels/quant.cu:147).  Expected: 0 for every positive finite s. */
long long qo_count_div3_mismatches(int s_lo, int s_hi) {
    long long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 64)
    for (int si = s_lo; si < s_hi; si++) {
        const float s = h2f((uint16_t)si);
        const float r = 1.0f / s;
        for (int xi = 0; xi < 0x7C00; xi++) {
            const float x = h2f((uint16_t)xi);
            const float q0 = x * r;
            const float rem = fmaf(-q0, s, x);
            const float q1 = fmaf(rem, r, q0);
            bad += f2h(x / s) != f2h(q1);
        }
    }
    return bad;
}
