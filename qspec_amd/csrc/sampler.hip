// Token-side kernels: embedding gather, greedy sampler front end, rejection
// sampler (accept/reject + recovered token + output assembly + counters) and
// the on-GPU advance step of the draft loop.
//
// Replaces (reference, relative to /root/reference):
//   Sampler.forward greedy path            vllm/model_executor/layers/sampler.py:216-316 (modify_greedy_probs=False :293)
//   RejectionSampler.forward & helpers     vllm/model_executor/layers/rejection_sampler.py:60-399
//   SpecDecodeBaseSampler._create_output   vllm/model_executor/layers/spec_decode_base_sampler.py:69-131
//   advance_step_flashattn_kernel          csrc/prepare_inputs/advance_step.cu:14-64
// The reference runs ~15 small torch kernels over [B,k,V] fp32 per step; here it is two.
#include "common.cuh"
#include "kernels.h"

namespace qspec {

__global__ __launch_bounds__(256) void embedding_kernel(const int64_t* __restrict__ ids, const f16* __restrict__ table,
                                                        f16* __restrict__ out, int H, int V) {
    const int t = blockIdx.x;
    int64_t id = ids[t];
    if (id < 0 || id >= V) id = 0;  // padded slot
    const f16* src = table + id * H;
    for (int i = threadIdx.x; i < H / 8; i += blockDim.x)
        *reinterpret_cast<f16x8*>(out + (size_t)t * H + 8 * i) = *reinterpret_cast<const f16x8*>(src + 8 * i);
}
int embedding(const int64_t* ids, const f16* table, f16* out, int T, int H, int V, hipStream_t st) {
    if (T == 0) return 0;
    if (H % 8) return -1;
    hipLaunchKernelGGL(embedding_kernel, dim3(T), dim3(256), 0, st, ids, table, out, H, V);
    return 0;
}

// block-wide helpers (1024 threads = 16 waves)
struct ArgMax {
    float v;
    int i;
};
__device__ __forceinline__ ArgMax argmax_combine(ArgMax a, ArgMax b) {
    // larger value wins; on ties the smaller index (first occurrence) wins; NaN never wins
    if (b.v > a.v || (b.v == a.v && b.i < a.i)) return b;
    return a;
}
__device__ __forceinline__ ArgMax block_argmax(ArgMax a, ArgMax* red) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) {
        ArgMax o;
        o.v = __shfl_xor(a.v, m, 64);
        o.i = __shfl_xor(a.i, m, 64);
        a = argmax_combine(a, o);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = a;
    __syncthreads();
    ArgMax r = red[0];
    for (int w = 1; w < nw; w++) r = argmax_combine(r, red[w]);
    return r;
}
__device__ __forceinline__ double block_sum_f64(double v, double* red) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if (lane == 0) red[wave] = v;
    __syncthreads();
    double r = red[0];
    for (int w = 1; w < nw; w++) r += red[w];
    return r;
}

// probs = softmax_fp32(float(logits)); token = argmax (first index on ties).
// exp via qexpf, denominator accumulated in fp64 and rounded once (the oracle does the same), so the fp32
// probabilities are the oracle's bit for bit up to a ~2^-29 chance per row.
// A row of 128256 logits is ~1.8 MB of traffic; one workgroup per row would sit on one CU's ~25 GB/s share
// (150 us).  The row is cut into QS_SM_CHUNKS chunks, three small launches:
//   1. per chunk max / argmax            -> part_a[row][chunk]
//   2. row max from the partials, e = qexpf(l - M) -> probs, per chunk sum (fp64) -> part_s[row][chunk]
//   3. total in chunk order (deterministic), p = e / float(total); chunk 0 writes the token
#define QS_SM_CHUNKS 64
struct SmPartA {
    float v;
    int i;
};
__device__ __forceinline__ void sm_chunk_range(int V, int c, int& lo, int& hi) {
    const int len = (((V + QS_SM_CHUNKS - 1) / QS_SM_CHUNKS) + 7) & ~7;
    lo = min(V, c * len);
    hi = min(V, lo + len);
}
__global__ __launch_bounds__(256) void softmax_max_kernel(const f16* __restrict__ logits, SmPartA* __restrict__ part_a,
                                                          int V) {
    __shared__ ArgMax red_a[4];
    const int c = blockIdx.x, t = blockIdx.y;
    int lo, hi;
    sm_chunk_range(V, c, lo, hi);
    const f16* l = logits + (size_t)t * V;
    ArgMax am{-__builtin_inff(), 0x7fffffff};
    for (int v = lo + threadIdx.x; v < hi; v += 256) {
        float f = h2f(l[v]);
        if (f > am.v) {
            am.v = f;
            am.i = v;
        }
    }
    am = block_argmax(am, red_a);
    if (threadIdx.x == 0) part_a[t * QS_SM_CHUNKS + c] = SmPartA{am.v, am.i};
}
__device__ __forceinline__ ArgMax sm_row_max(const SmPartA* part_a, int t) {
    ArgMax am{-__builtin_inff(), 0x7fffffff};
    for (int c = 0; c < QS_SM_CHUNKS; c++) {
        SmPartA pa = part_a[t * QS_SM_CHUNKS + c];
        am = argmax_combine(am, ArgMax{pa.v, pa.i});
    }
    return am;
}
// np <= 256 partial maxima per row (QS_SM_CHUNKS from softmax_max_kernel, or one per workgroup of the lm_head
// launch): one load per thread, then a block reduction (call from all 256 threads of the workgroup)
__device__ __forceinline__ ArgMax sm_row_max_n(const SmPartA* part_a, int t, int np, ArgMax* red_a) {
    ArgMax am{-__builtin_inff(), 0x7fffffff};
    if ((int)threadIdx.x < np) {
        const SmPartA pa = part_a[(size_t)t * np + threadIdx.x];
        am = ArgMax{pa.v, pa.i};
    }
    return block_argmax(am, red_a);
}
// pass 2: the chunk's share of the denominator, sum of qexpf(l - M) in fp64.  Nothing is written to probs here: the
// fp16 logits (2 B per element, cache resident) are cheaper to read twice than an fp32 e-array is to write and re-read
// (the reference's softmax + log_softmax make two passes over T x V x 4 B, sampler.py:270-287).
__global__ __launch_bounds__(256) void softmax_sum_kernel(const f16* __restrict__ logits, const SmPartA* part_a, int np,
                                                          double* __restrict__ part_s, int V) {
    __shared__ double red_d[4];
    __shared__ ArgMax red_a[4];
    const int c = blockIdx.x, t = blockIdx.y;
    int lo, hi;
    sm_chunk_range(V, c, lo, hi);
    const float mx = sm_row_max_n(part_a, t, np, red_a).v;
    const f16* l = logits + (size_t)t * V;
    double den = 0.0;
    for (int v = lo + threadIdx.x; v < hi; v += 256) den += (double)qexpf(h2f(l[v]) - mx);
    den = block_sum_f64(den, red_d);
    if (threadIdx.x == 0) part_s[t * QS_SM_CHUNKS + c] = den;
}
// pass 3: p = qexpf(l - M) / float(total) written ONCE; chunk 0 writes the token
__global__ __launch_bounds__(256) void softmax_write_kernel(const f16* __restrict__ logits, const SmPartA* part_a,
                                                            int np, const double* part_s, float* __restrict__ probs,
                                                            int64_t* __restrict__ token, int64_t token_stride, int V) {
    __shared__ ArgMax red_a[4];
    const int c = blockIdx.x, t = blockIdx.y;
    int lo, hi;
    sm_chunk_range(V, c, lo, hi);
    double den = 0.0;
    for (int cc = 0; cc < QS_SM_CHUNKS; cc++) den += part_s[t * QS_SM_CHUNKS + cc];
    const float inv = (float)den;
    const ArgMax am = sm_row_max_n(part_a, t, np, red_a);
    const f16* l = logits + (size_t)t * V;
    float* p = probs + (size_t)t * V;
    for (int v = lo + threadIdx.x; v < hi; v += 256) p[v] = qexpf(h2f(l[v]) - am.v) / inv;
    if (c == 0 && threadIdx.x == 0) token[t * token_stride] = am.i == 0x7fffffff ? 0 : am.i;
}
size_t sampler_ws_bytes(int rows) { return (size_t)rows * QS_SM_CHUNKS * 16; }
int softmax_argmax(const f16* logits, float* probs, int64_t* token, int T, int V, void* ws, hipStream_t st) {
    if (T == 0) return 0;
    SmPartA* part_a = reinterpret_cast<SmPartA*>(ws);
    double* part_s = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + (size_t)T * QS_SM_CHUNKS * 8);
    hipLaunchKernelGGL(softmax_max_kernel, dim3(QS_SM_CHUNKS, T), dim3(256), 0, st, logits, part_a, V);
    hipLaunchKernelGGL(softmax_sum_kernel, dim3(QS_SM_CHUNKS, T), dim3(256), 0, st, logits, part_a, QS_SM_CHUNKS, part_s, V);
    hipLaunchKernelGGL(softmax_write_kernel, dim3(QS_SM_CHUNKS, T), dim3(256), 0, st, logits, part_a, QS_SM_CHUNKS,
                       part_s, probs, token, (int64_t)1, V);
    return 0;
}
// Behind an lm_head launch that left nparts <= 256 partial maxima per row: two launches, the logits are read twice
// (fp16) and probs written once.  ws: [rows][256] SmPartA, then [rows][QS_SM_CHUNKS] double.
size_t head_softmax_ws_bytes(int rows) { return (size_t)rows * (256 * 8 + QS_SM_CHUNKS * 8); }
void* head_softmax_part_max(void* ws, int rows) { return ws; }
int head_softmax_argmax(const f16* logits, float* probs, int64_t* token, int T, int V, int nparts, void* ws,
                        hipStream_t st) {
    if (T == 0) return 0;
    if (nparts < 1 || nparts > 256) return -1;
    SmPartA* part_a = reinterpret_cast<SmPartA*>(ws);
    double* part_s = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + (size_t)T * 256 * 8);
    hipLaunchKernelGGL(softmax_sum_kernel, dim3(QS_SM_CHUNKS, T), dim3(256), 0, st, logits, part_a, nparts, part_s, V);
    hipLaunchKernelGGL(softmax_write_kernel, dim3(QS_SM_CHUNKS, T), dim3(256), 0, st, logits, part_a, nparts, part_s,
                       probs, token, (int64_t)1, V);
    return 0;
}

// Philox4x32-10 (counter-based; the product's own stream -- CUDA's torch.rand stream is not reproducible here)
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
__device__ __forceinline__ float u01_open(uint32_t x) { return ((float)(x >> 8) + 0.5f) * (1.0f / 16777216.0f); }

// accepted[b,i] = U < min(1, q[x]/p[x]);
// recovered[b,i] = argmax_v ( max(q_v - p_v, FLT_MIN) / S ) / E_v ,  S = sum_v max(q_v - p_v, FLT_MIN)
// target_probs [B, k+1, V], draft_probs (b,i,v) at b*dp_sb + i*dp_sk + v.  uniform [B,k] / exponential [B,k,V] may
// be injected (tests); when null they come from Philox keyed by (seed, offset).
// Same shape as the softmax: each (b,i) row is cut into QS_SM_CHUNKS chunks over the chip.
//   1. per chunk sum of max(q-p, tiny) in fp64                       -> part_s
//   2. S = float(total in chunk order); per chunk argmax of (f/S)/E  -> part_a
//   3. one workgroup: final argmax per (b,i), accept test, output assembly + counters
__global__ __launch_bounds__(256) void rejection_sum_kernel(const float* __restrict__ target_probs,
                                                            const float* __restrict__ draft_probs, int k, int V,
                                                            int64_t dp_sb, int64_t dp_sk,
                                                            double* __restrict__ part_s) {
    __shared__ double red_d[4];
    const int c = blockIdx.x, bk = blockIdx.y, b = bk / k, i = bk % k;
    int lo, hi;
    sm_chunk_range(V, c, lo, hi);
    const float* q = target_probs + ((size_t)b * (k + 1) + i) * V;
    const float* p = draft_probs + b * dp_sb + i * dp_sk;
    const float tiny = 1.17549435e-38f;
    double s = 0.0;
    for (int v = lo + threadIdx.x; v < hi; v += 256) s += (double)fmaxf(q[v] - p[v], tiny);
    s = block_sum_f64(s, red_d);
    if (threadIdx.x == 0) part_s[bk * QS_SM_CHUNKS + c] = s;
}
__global__ __launch_bounds__(256) void rejection_argmax_kernel(const float* __restrict__ target_probs,
                                                               const float* __restrict__ draft_probs,
                                                               const float* __restrict__ exponential, uint64_t seed,
                                                               uint64_t offset, const uint64_t* __restrict__ rng_state,
                                                               int k, int V, int64_t dp_sb, int64_t dp_sk,
                                                               const double* part_s, SmPartA* __restrict__ part_a) {
    __shared__ ArgMax red_a[4];
    const int c = blockIdx.x, bk = blockIdx.y, b = bk / k, i = bk % k;
    int lo, hi;
    sm_chunk_range(V, c, lo, hi);
    const float* q = target_probs + ((size_t)b * (k + 1) + i) * V;
    const float* p = draft_probs + b * dp_sb + i * dp_sk;
    if (rng_state) {  // device-resident (seed, offset): lets a captured graph draw fresh numbers every replay
        seed = rng_state[0];
        offset = rng_state[1];
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const float tiny = 1.17549435e-38f;
    double tot = 0.0;
    for (int cc = 0; cc < QS_SM_CHUNKS; cc++) tot += part_s[bk * QS_SM_CHUNKS + cc];
    const float S = (float)tot;
    ArgMax am{-__builtin_inff(), 0x7fffffff};
    for (int v = lo + threadIdx.x; v < hi; v += 256) {
        float f = fmaxf(q[v] - p[v], tiny) / S;
        float e;
        if (exponential) {
            e = exponential[(size_t)bk * V + v];
        } else {
            uint32_t r[4];
            philox4x32((uint32_t)v, (uint32_t)bk, (uint32_t)offset, (uint32_t)(offset >> 32), k0, k1, r);
            e = -__logf(u01_open(r[0]));
        }
        float val = f / e;
        if (val > am.v) {
            am.v = val;
            am.i = v;
        }
    }
    am = block_argmax(am, red_a);
    if (threadIdx.x == 0) part_a[bk * QS_SM_CHUNKS + c] = SmPartA{am.v, am.i};
}
// SpecDecodeBaseSampler._create_output + the counters (spec_decode_base_sampler.py:69-131), shared by the rejection and the
// typical-acceptance sampler: out[b, i] = draft token (i < limit) | recovered (i == limit) | -1; out[b, k] = bonus iff every
// draft token was accepted; counters[0] += accepted.sum() (non-causal), counters[1] += #(out != -1), counters[2] += B*k.
// Call from every thread of ONE workgroup, behind the barrier that made accepted / recovered visible.
__device__ __forceinline__ void create_output(int B, int k, const int64_t* __restrict__ draft_ids, int64_t di_sb, int64_t di_sk,
                                              const int64_t* __restrict__ bonus_ids, int64_t bonus_stride,
                                              const uint8_t* accepted, const int64_t* recovered, int64_t* __restrict__ out,
                                              int64_t* __restrict__ counters, const int32_t* __restrict__ active_lens) {
    int acc_cnt = 0, emit_cnt = 0, rows_on = 0;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
        if (active_lens && active_lens[b] <= 0) {
            for (int i = 0; i <= k; i++) out[b * (k + 1) + i] = -1;
            continue;
        }
        rows_on++;
        int limit = k;
        for (int i = 0; i < k; i++) {
            if (accepted[b * k + i]) acc_cnt++;
            else if (limit == k) limit = i;
        }
        for (int i = 0; i < k; i++) {
            int64_t v = i < limit ? draft_ids[b * di_sb + i * di_sk] : (i == limit ? recovered[b * k + i] : -1);
            out[b * (k + 1) + i] = v;
            emit_cnt += v != -1;
        }
        int64_t last = limit == k ? bonus_ids[b * bonus_stride] : -1;
        out[b * (k + 1) + k] = last;
        emit_cnt += last != -1;
    }
    if (counters) {
        if (acc_cnt) atomicAdd(reinterpret_cast<unsigned long long*>(counters), (unsigned long long)acc_cnt);
        if (emit_cnt) atomicAdd(reinterpret_cast<unsigned long long*>(counters + 1), (unsigned long long)emit_cnt);
        if (rows_on) atomicAdd(reinterpret_cast<unsigned long long*>(counters + 2), (unsigned long long)rows_on * k);
    }
}

// Final step + output assembly + counters; one workgroup.
__global__ __launch_bounds__(1024) void rejection_output_kernel(
    const float* __restrict__ target_probs, const float* __restrict__ draft_probs,
    const int64_t* __restrict__ draft_ids, const int64_t* __restrict__ bonus_ids, const float* __restrict__ uniform,
    uint64_t seed, uint64_t offset, uint64_t* __restrict__ rng_state, const SmPartA* part_a, int B, int k, int V,
    int64_t dp_sb, int64_t dp_sk, int64_t di_sb, int64_t di_sk, int64_t bonus_stride, uint8_t* __restrict__ accepted,
    int64_t* __restrict__ recovered, int64_t* __restrict__ out, int64_t* __restrict__ counters,
    const int32_t* __restrict__ active_lens) {
    // active_lens (may be NULL): row b takes part iff active_lens[b] > 0 (an empty batch slot of the engine emits
    // nothing and is not counted: the reference only ever sees the sequences that are scheduled)
    if (rng_state) {
        seed = rng_state[0];
        offset = rng_state[1];
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int bk = threadIdx.x; bk < B * k; bk += blockDim.x) {
        const int b = bk / k, i = bk % k;
        ArgMax am{-__builtin_inff(), 0x7fffffff};
        for (int c = 0; c < QS_SM_CHUNKS; c++) {
            SmPartA pa = part_a[bk * QS_SM_CHUNKS + c];
            am = argmax_combine(am, ArgMax{pa.v, pa.i});
        }
        const int64_t x = draft_ids[b * di_sb + i * di_sk];
        float u;
        if (uniform) {
            u = uniform[bk];
        } else {
            uint32_t r[4];
            philox4x32(0xFFFFFFFFu, (uint32_t)bk, (uint32_t)offset, (uint32_t)(offset >> 32), k0, k1, r);
            u = (float)(r[0] >> 8) * (1.0f / 16777216.0f);  // [0,1) like torch.rand
        }
        const float qx = target_probs[((size_t)b * (k + 1) + i) * V + x];
        const float px = draft_probs[b * dp_sb + i * dp_sk + x];
        const float rq = qx / px;
        // torch.minimum propagates NaN (0/0) and `u < NaN` is false: such a token is rejected
        const bool row_on = !active_lens || active_lens[b] > 0;
        accepted[bk] = (row_on && (rq == rq) && (u < fminf(rq, 1.0f))) ? 1 : 0;
        recovered[bk] = am.i == 0x7fffffff ? 0 : am.i;
    }
    __syncthreads();
    create_output(B, k, draft_ids, di_sb, di_sk, bonus_ids, bonus_stride, accepted, recovered, out, counters, active_lens);
    if (rng_state && threadIdx.x == 0) rng_state[1] = offset + 1;  // after every draw of this call
}

int rejection_sample(const float* target_probs, const float* draft_probs, const int64_t* draft_ids,
                     const int64_t* bonus_ids, const float* uniform, const float* exponential, uint64_t seed,
                     uint64_t offset, uint64_t* rng_state, int B, int k, int V, int64_t dp_sb, int64_t dp_sk,
                     int64_t di_sb, int64_t di_sk, int64_t bonus_stride, int64_t* out_tokens, uint8_t* accepted,
                     int64_t* recovered, int64_t* counters, const int32_t* active_lens, void* ws, hipStream_t st) {
    if (B == 0) return 0;
    if (k < 1) return -1;
    SmPartA* part_a = reinterpret_cast<SmPartA*>(ws);
    double* part_s = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + (size_t)B * k * QS_SM_CHUNKS * 8);
    hipLaunchKernelGGL(rejection_sum_kernel, dim3(QS_SM_CHUNKS, B * k), dim3(256), 0, st, target_probs, draft_probs, k,
                       V, dp_sb, dp_sk, part_s);
    hipLaunchKernelGGL(rejection_argmax_kernel, dim3(QS_SM_CHUNKS, B * k), dim3(256), 0, st, target_probs, draft_probs,
                       exponential, seed, offset, rng_state, k, V, dp_sb, dp_sk, part_s, part_a);
    hipLaunchKernelGGL(rejection_output_kernel, dim3(1), dim3(1024), 0, st, target_probs, draft_probs, draft_ids,
                       bonus_ids, uniform, seed, offset, rng_state, part_a, B, k, V, dp_sb, dp_sk, di_sb, di_sk,
                       bonus_stride, accepted, recovered, out_tokens, counters, active_lens);
    return 0;
}

// ---------------------------------------------------------------- typical acceptance (MEDUSA 3.3.1)
// TypicalAcceptanceSampler.forward (vllm/model_executor/layers/typical_acceptance_sampler.py:37-172), deterministic:
//   accepted[b,i]  = q[b,i,x] > min(posterior_threshold, posterior_alpha * exp(-H)),  H = -sum_v q_v log(q_v + 1e-5)
//   recovered[b,i] = argmax_v q[b,i,v]           (first index on ties, as torch.argmax)
// then the shared _create_output.  q = target_with_bonus_probs[:, :-1].  Same launch shape as the rejection sampler: each
// (b, i) row in QS_SM_CHUNKS chunks over the chip -- the entropy terms in fp32 as the reference forms them (add, log,
// multiply), summed in fp64 in chunk order and rounded once -- then one workgroup for the decision and the output.
__global__ __launch_bounds__(256) void typical_entropy_kernel(const float* __restrict__ target_probs, int k, int V,
                                                              double* __restrict__ part_s, SmPartA* __restrict__ part_a) {
    __shared__ double red_d[4];
    __shared__ ArgMax red_a[4];
    const int c = blockIdx.x, bk = blockIdx.y, b = bk / k, i = bk % k;
    int lo, hi;
    sm_chunk_range(V, c, lo, hi);
    const float* q = target_probs + ((size_t)b * (k + 1) + i) * V;
    double s = 0.0;
    ArgMax am{-__builtin_inff(), 0x7fffffff};
    for (int v = lo + threadIdx.x; v < hi; v += 256) {
        const float qv = q[v];
        s += (double)(qv * logf(qv + 1e-5f));
        if (qv > am.v) {
            am.v = qv;
            am.i = v;
        }
    }
    s = block_sum_f64(s, red_d);
    am = block_argmax(am, red_a);
    if (threadIdx.x == 0) {
        part_s[bk * QS_SM_CHUNKS + c] = s;
        part_a[bk * QS_SM_CHUNKS + c] = SmPartA{am.v, am.i};
    }
}
__global__ __launch_bounds__(1024) void typical_output_kernel(
    const float* __restrict__ target_probs, const int64_t* __restrict__ draft_ids, const int64_t* __restrict__ bonus_ids,
    float posterior_threshold, float posterior_alpha, const double* part_s, const SmPartA* part_a, int B, int k, int V,
    int64_t di_sb, int64_t di_sk, int64_t bonus_stride, uint8_t* __restrict__ accepted, int64_t* __restrict__ recovered,
    int64_t* __restrict__ out, int64_t* __restrict__ counters, const int32_t* __restrict__ active_lens) {
    for (int bk = threadIdx.x; bk < B * k; bk += blockDim.x) {
        const int b = bk / k, i = bk % k;
        double tot = 0.0;
        ArgMax am{-__builtin_inff(), 0x7fffffff};
        for (int c = 0; c < QS_SM_CHUNKS; c++) {
            tot += part_s[bk * QS_SM_CHUNKS + c];
            SmPartA pa = part_a[bk * QS_SM_CHUNKS + c];
            am = argmax_combine(am, ArgMax{pa.v, pa.i});
        }
        const float H = -(float)tot;
        const float thr = fminf(posterior_threshold, qexpf(-H) * posterior_alpha);
        const int64_t x = draft_ids[b * di_sb + i * di_sk];
        const float cand = target_probs[((size_t)b * (k + 1) + i) * V + x];
        const bool row_on = !active_lens || active_lens[b] > 0;
        accepted[bk] = (row_on && cand > thr) ? 1 : 0;
        recovered[bk] = am.i == 0x7fffffff ? 0 : am.i;
    }
    __syncthreads();
    create_output(B, k, draft_ids, di_sb, di_sk, bonus_ids, bonus_stride, accepted, recovered, out, counters, active_lens);
}
int typical_acceptance_sample(const float* target_probs, const int64_t* draft_ids, const int64_t* bonus_ids,
                              float posterior_threshold, float posterior_alpha, int B, int k, int V, int64_t di_sb,
                              int64_t di_sk, int64_t bonus_stride, int64_t* out_tokens, uint8_t* accepted, int64_t* recovered,
                              int64_t* counters, const int32_t* active_lens, void* ws, hipStream_t st) {
    if (B == 0) return 0;
    if (k < 1) return -1;
    SmPartA* part_a = reinterpret_cast<SmPartA*>(ws);
    double* part_s = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + (size_t)B * k * QS_SM_CHUNKS * 8);
    hipLaunchKernelGGL(typical_entropy_kernel, dim3(QS_SM_CHUNKS, B * k), dim3(256), 0, st, target_probs, k, V, part_s, part_a);
    hipLaunchKernelGGL(typical_output_kernel, dim3(1), dim3(1024), 0, st, target_probs, draft_ids, bonus_ids, posterior_threshold,
                       posterior_alpha, part_s, part_a, B, k, V, di_sb, di_sk, bonus_stride, accepted, recovered, out_tokens,
                       counters, active_lens);
    return 0;
}

// ---------------------------------------------------------------- Sampler.forward, the non-greedy rows
// vllm/model_executor/layers/sampler.py:216-316: l = float(logits) / temperature; _apply_top_k_top_p (:387-413: ascending
// sort; keep the top_k largest -- ties at the k-th value kept --; softmax of what is left; mask where the ASCENDING cumulative
// probability is <= 1 - top_p, never the last); probs = softmax(masked l); token = argmax(probs / Exp(1)) (_multinomial
// :585-604) or, for rows with temperature < 1e-5 (greedy requests of a mixed batch: scaled by 1.0, sampling_metadata.py:413-417),
// argmax(probs).
//
// No sort.  The logits are fp16 and l is monotonic in the logit for a positive temperature, so the ORDER of a row is the
// order of 65536 possible keys: (1) a histogram of the row's keys (integer atomics: deterministic); (2) one workgroup per row
// walks the keys in descending order -- 64 per thread, exclusive block scans of the counts and of the probability mass
// count x exp(l - max) in fp64 -- and finds the top-k key (the k-th largest value), the mass Z_k of what top-k keeps, the
// top-p key (the lowest key whose strictly-larger keys hold less than top_p x Z_k of the mass: the reference's ascending-
// cumsum rule evaluated per group of EQUAL logits -- inside such a group the reference masks whichever members its unstable
// sort happened to put first; here a boundary group stays whole, deterministic) and the final denominator; (3) an
// elementwise pass writes probs ONCE and leaves per-chunk argmax partials of probs / noise; (4) a last small launch
// combines them.  The histogram is left zeroed for the next call.
#define QS_TK_KEYS 65536
__device__ __forceinline__ uint32_t tk_key(f16 h) {   // order-preserving: larger logit -> larger key
    const uint16_t b = __builtin_bit_cast(uint16_t, h);
    return (b & 0x8000u) ? (uint32_t)(uint16_t)~b : (uint32_t)(b | 0x8000u);
}
__device__ __forceinline__ float tk_val(uint32_t key) {   // the logit of a key
    const uint16_t b = (key & 0x8000u) ? (uint16_t)(key & 0x7FFFu) : (uint16_t)~key;
    return h2f(__builtin_bit_cast(f16, b));
}
struct TkRow {      // what pass 3 needs of a row
    uint32_t tau;   // keys >= tau are kept
    float temp;     // the divisor (1.0 for greedy rows)
    float mx;       // max l
    float z;        // float(sum of exp(l - mx) over the kept tokens)
    int greedy;
    int pad[3];
};
// One workgroup per (row, eighth of the vocabulary): a PRIVATE histogram of all 65536 keys in LDS -- 16-bit counters packed two
// to a dword (an eighth of a 128 K vocabulary cannot overflow one; larger vocabularies take more chunks) -- then only the non-zero
// counters go to the row's histogram in memory: a few thousand integer atomics per workgroup instead of one per logit (the first
// form: 38.6 us per launch at 4 rows, 79 at 16).  Integer adds: the histogram does not depend on the order.
#define QS_TK_HCHUNK 16384   // logits per workgroup (< 65536)
__global__ __launch_bounds__(1024) void topk_hist_kernel(const f16* __restrict__ logits, uint32_t* __restrict__ hist, int V) {
    extern __shared__ __attribute__((aligned(16))) uint32_t lh[];   // [32768] = 65536 16-bit counters
    const int c = blockIdx.x, t = blockIdx.y, tid = threadIdx.x;
    uint4* lz = reinterpret_cast<uint4*>(lh);
#pragma unroll
    for (int i = 0; i < 8; i++) lz[tid + 1024 * i] = uint4{0u, 0u, 0u, 0u};
    __syncthreads();
    const int lo = c * QS_TK_HCHUNK, hi = min(V, lo + QS_TK_HCHUNK);
    const f16* l = logits + (size_t)t * V;
    for (int v = lo + tid; v < hi; v += 1024) {
        const f16 x = l[v];
        if (x == x) {   // (a NaN logit takes no part)
            const uint32_t key = tk_key(x);
            atomicAdd(lh + (key >> 1), (key & 1u) ? 65536u : 1u);
        }
    }
    __syncthreads();
    uint32_t* h = hist + (size_t)t * QS_TK_KEYS;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int q = tid + 1024 * i;
        const uint4 v = lz[q];
        const uint32_t w4[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (w4[e] & 0xFFFFu) atomicAdd(h + 8 * q + 2 * e, w4[e] & 0xFFFFu);
            if (w4[e] >> 16) atomicAdd(h + 8 * q + 2 * e + 1, w4[e] >> 16);
        }
    }
}
__global__ __launch_bounds__(1024) void topk_select_kernel(uint32_t* __restrict__ hist, const float* __restrict__ temperature,
                                                            const int32_t* __restrict__ top_k, const float* __restrict__ top_p,
                                                            TkRow* __restrict__ rows, int V) {
    __shared__ uint32_t s_cnt[16];
    __shared__ double s_mass[16];
    __shared__ uint32_t s_u[4];
    __shared__ double s_zk, s_z;
    const int t = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    uint32_t* h = hist + (size_t)t * QS_TK_KEYS;
    const float traw = temperature ? temperature[t] : 1.0f;
    const bool greedy = traw < 1e-5f;
    const float temp = greedy ? 1.0f : traw;
    int kk = top_k ? top_k[t] : -1;
    if (kk <= 0 || kk > V) kk = V;
    const double pp = top_p ? (double)top_p[t] : 1.0;
    // thread tid owns keys 65535 - 64 tid - j, j = 0 .. 63 (descending).  Its 64 counts are read ONCE, as sixteen 16-byte loads
    // of its 256 contiguous bytes, and stay in registers: every loop over them is fully unrolled (static indices; a dynamic index
    // spills them), and the three boundary searches below are walked only by the thread that holds the boundary -- the other
    // 1023 answer from their block-scan prefixes.  (Round 4: the kernel re-read the histogram five times with 4-byte loads at a
    // 256-byte lane stride: 149 us per launch, 0.6 ms of a sampled cycle.)
    const uint32_t k_hi = 65535u - 64u * tid;
    uint32_t cnt[64];
    {
        const uint4* hv = reinterpret_cast<const uint4*>(h + (k_hi - 63u));   // key k_hi - 63 + 4 q + e  <->  j = 63 - 4 q - e
#pragma unroll
        for (int q = 0; q < 16; q++) {
            const uint4 v = hv[q];
            cnt[63 - 4 * q] = v.x;
            cnt[62 - 4 * q] = v.y;
            cnt[61 - 4 * q] = v.z;
            cnt[60 - 4 * q] = v.w;
        }
    }
    uint32_t c_loc = 0, first = 0xFFFFFFFFu;
#pragma unroll
    for (int j = 0; j < 64; j++) {
        const uint32_t cj = cnt[j];
        c_loc += cj;
        if (cj && first == 0xFFFFFFFFu) first = 64u * tid + j;   // descending position of this thread's largest key
    }
    // the row maximum = the first non-empty key in descending order
    uint32_t fm = first;
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) fm = min(fm, (uint32_t)__shfl_xor((int)fm, m, 64));
    if (lane == 0) s_cnt[wave] = fm;
    __syncthreads();
    uint32_t pos_max = s_cnt[0];
    for (int w2 = 1; w2 < 16; w2++) pos_max = min(pos_max, s_cnt[w2]);
    __syncthreads();
    const float mx = pos_max == 0xFFFFFFFFu ? 0.0f : tk_val(65535u - pos_max) / temp;
    auto wgt = [&](int j) -> double { return (double)cnt[j] * (double)qexpf(tk_val(k_hi - j) / temp - mx); };
    double m_loc = 0.0;
    if (c_loc) {
#pragma unroll
        for (int j = 0; j < 64; j++)
            if (cnt[j]) m_loc += wgt(j);
    }
    // exclusive block scans (thread order = descending keys): counts and mass
    uint32_t c_inc = c_loc;
    double m_inc = m_loc;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const uint32_t oc = (uint32_t)__shfl_up((int)c_inc, d, 64);
        const double om = __shfl_up(m_inc, d, 64);
        if (lane >= d) {
            c_inc += oc;
            m_inc += om;
        }
    }
    if (lane == 63) {
        s_cnt[wave] = c_inc;
        s_mass[wave] = m_inc;
    }
    __syncthreads();
    uint32_t c_ex = c_inc - c_loc;
    double m_ex = m_inc - m_loc;
    for (int w2 = 0; w2 < wave; w2++) {
        c_ex += s_cnt[w2];
        m_ex += s_mass[w2];
    }
    if (tid == 0) {
        s_u[0] = 0xFFFFFFFFu;   // descending position of the top-k key
        s_u[1] = 0xFFFFFFFFu;   // descending position of the first key top-p masks
    }
    __syncthreads();
    // (a) top-k: the key at which the descending cumulative count reaches k; Z_k = the mass down to and including it.
    // Exactly one thread's count range (c_ex, c_ex + c_loc] contains k: it walks its keys.
    if (c_ex < (uint32_t)kk && c_ex + c_loc >= (uint32_t)kk) {
        uint32_t run = c_ex;
        double mrun = m_ex;
#pragma unroll
        for (int j = 0; j < 64; j++) {
            const uint32_t cj = cnt[j];
            if (cj) {
                const double w = wgt(j);
                if (run < (uint32_t)kk && run + cj >= (uint32_t)kk) {
                    s_u[0] = 64u * tid + j;
                    s_zk = mrun + w;
                }
                run += cj;
                mrun += w;
            }
        }
    }
    if (tid == 1023 && c_ex + c_loc < (uint32_t)kk) {   // fewer than k finite logits: everything is kept
        s_u[0] = 65535u;
        s_zk = m_ex + m_loc;
    }
    __syncthreads();
    const uint32_t pos_k = s_u[0];
    const double zk = s_zk;
    // (b) top-p: the first key (descending) whose strictly-larger keys already hold >= p Z_k of the mass is masked, and
    // everything below it; a group of equal logits is kept or masked as a whole.  The mass above a key never decreases along
    // the positions, so the answer lies in the thread where it crosses p Z_k -- or, if it crosses behind that thread's last
    // key, it is the first key of a later thread: those offer theirs without a walk, the smallest position wins.
    {
        const double thr = pp * zk;
        uint32_t best = 0xFFFFFFFFu;
        if (c_loc && m_ex + m_loc >= thr && 64u * tid <= pos_k) {
            if (m_ex >= thr && first != pos_max) {
                if (first <= pos_k) best = first;
            } else {
                double mrun = m_ex;
#pragma unroll
                for (int j = 0; j < 64; j++) {
                    const uint32_t cj = cnt[j];
                    if (cj) {
                        const uint32_t pos = 64u * tid + j;
                        if (best == 0xFFFFFFFFu && pos <= pos_k && pos != pos_max && mrun >= thr) best = pos;
                        mrun += wgt(j);
                    }
                }
            }
        }
        if (best != 0xFFFFFFFFu) atomicMin(&s_u[1], best);
    }
    __syncthreads();
    const uint32_t pos_p = s_u[1];                                       // first masked position, or none
    const uint32_t pos_last = pos_p == 0xFFFFFFFFu ? pos_k : min(pos_k, pos_p - 1);   // last kept position (descending)
    // (c) the final denominator: the mass of positions <= pos_last (pos_last itself may be an empty key: the position just
    // above the first masked one); the thread whose 64 positions contain pos_last walks and publishes it
    if ((pos_last >> 6) == (uint32_t)tid) {
        double mrun = m_ex;
#pragma unroll
        for (int j = 0; j < 64; j++)
            if (cnt[j] && 64u * tid + j <= pos_last) mrun += wgt(j);
        s_z = mrun;
    }
    if (c_loc) {   // leave the histogram zeroed for the next call
        uint4* hz = reinterpret_cast<uint4*>(h + (k_hi - 63u));
#pragma unroll
        for (int q = 0; q < 16; q++) hz[q] = uint4{0u, 0u, 0u, 0u};
    }
    __syncthreads();
    if (tid == 0) {
        TkRow r;
        r.tau = 65535u - pos_last;
        // (between pos_last and the next non-empty key nothing lives, so "key >= tau" keeps exactly positions <= pos_last)
        r.temp = temp;
        r.mx = mx;
        r.z = (float)s_z;
        r.greedy = greedy ? 1 : 0;
        r.pad[0] = r.pad[1] = r.pad[2] = 0;
        rows[t] = r;
    }
}
__global__ __launch_bounds__(256) void topk_write_kernel(const f16* __restrict__ logits, const TkRow* __restrict__ rows,
                                                         const float* __restrict__ exponential, uint64_t seed, uint64_t offset,
                                                         const uint64_t* __restrict__ rng_state, float* __restrict__ probs,
                                                         SmPartA* __restrict__ part_a, int V) {
    __shared__ ArgMax red_a[4];
    const int c = blockIdx.x, t = blockIdx.y;
    int lo, hi;
    sm_chunk_range(V, c, lo, hi);
    const TkRow r = rows[t];
    if (rng_state) {
        seed = rng_state[0];
        offset = rng_state[1];
    }
    const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    const f16* l = logits + (size_t)t * V;
    float* p = probs + (size_t)t * V;
    ArgMax am{-__builtin_inff(), 0x7fffffff};
    for (int v = lo + threadIdx.x; v < hi; v += 256) {
        const f16 x = l[v];
        float pv = 0.0f;
        if (x == x && tk_key(x) >= r.tau) pv = qexpf(h2f(x) / r.temp - r.mx) / r.z;
        p[v] = pv;
        float val = pv;
        if (!r.greedy) {
            float e;
            if (exponential) {
                e = exponential[(size_t)t * V + v];
            } else {
                uint32_t rr[4];
                philox4x32((uint32_t)v, 0x40000000u | (uint32_t)t, (uint32_t)offset, (uint32_t)(offset >> 32), k0, k1, rr);
                e = -__logf(u01_open(rr[0]));
            }
            val = pv / e;
        }
        if (val > am.v) {
            am.v = val;
            am.i = v;
        }
    }
    am = block_argmax(am, red_a);
    if (threadIdx.x == 0) part_a[t * QS_SM_CHUNKS + c] = SmPartA{am.v, am.i};
}
__global__ __launch_bounds__(256) void topk_token_kernel(const SmPartA* __restrict__ part_a, int64_t* __restrict__ token,
                                                         int64_t token_stride, uint64_t* __restrict__ rng_state, int T) {
    for (int t = threadIdx.x; t < T; t += 256) {
        const ArgMax am = sm_row_max(part_a, t);
        token[t * token_stride] = am.i == 0x7fffffff ? 0 : am.i;
    }
    if (rng_state && threadIdx.x == 0) rng_state[1] = rng_state[1] + 1;   // after every draw of this call
}
// ws: [rows][65536] u32 histogram (zero-filled ONCE by the caller, left zeroed by every call) | [rows] TkRow | [rows][CHUNKS] SmPartA
size_t sample_ws_bytes(int rows) {
    return (size_t)rows * QS_TK_KEYS * 4 + (size_t)rows * sizeof(TkRow) + (size_t)rows * QS_SM_CHUNKS * sizeof(SmPartA);
}
int sample_top_k_top_p(const f16* logits, const float* temperature, const int32_t* top_k, const float* top_p,
                       const float* exponential, uint64_t seed, uint64_t offset, uint64_t* rng_state, float* probs,
                       int64_t* token, int64_t token_stride, int T, int V, void* ws, hipStream_t st) {
    if (T == 0) return 0;
    uint32_t* hist = reinterpret_cast<uint32_t*>(ws);
    TkRow* rows = reinterpret_cast<TkRow*>(reinterpret_cast<char*>(ws) + (size_t)T * QS_TK_KEYS * 4);
    SmPartA* part_a = reinterpret_cast<SmPartA*>(reinterpret_cast<char*>(rows) + (size_t)T * sizeof(TkRow));
    static bool hist_attr = false;
    if (!hist_attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(topk_hist_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                QS_TK_KEYS * 2) != hipSuccess)
            return -8;
        hist_attr = true;
    }
    hipLaunchKernelGGL(topk_hist_kernel, dim3((V + QS_TK_HCHUNK - 1) / QS_TK_HCHUNK, T), dim3(1024), QS_TK_KEYS * 2, st, logits, hist, V);
    hipLaunchKernelGGL(topk_select_kernel, dim3(T), dim3(1024), 0, st, hist, temperature, top_k, top_p, rows, V);
    hipLaunchKernelGGL(topk_write_kernel, dim3(QS_SM_CHUNKS, T), dim3(256), 0, st, logits, rows, exponential, seed, offset,
                       rng_state, probs, part_a, V);
    hipLaunchKernelGGL(topk_token_kernel, dim3(1), dim3(256), 0, st, part_a, token, token_stride, rng_state, T);
    return 0;
}

// advance_step_flashattn_kernel without the CUDA-graph padding branch (num_seqs == num_queries).
__global__ void advance_step_kernel(int n, int block_size, int64_t* __restrict__ input_tokens,
                                    const int64_t* __restrict__ sampled, int64_t* __restrict__ positions,
                                    int32_t* __restrict__ seq_lens, int64_t* __restrict__ slot_mapping,
                                    const int32_t* __restrict__ block_tables, int64_t bt_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    input_tokens[i] = sampled[i];
    const int next_len = seq_lens[i] + 1;
    const int pos = next_len - 1;
    seq_lens[i] = next_len;
    positions[i] = pos;
    slot_mapping[i] = (int64_t)block_tables[bt_stride * i + pos / block_size] * block_size + pos % block_size;
}
int advance_step(int n, int block_size, int64_t* input_tokens, const int64_t* sampled, int64_t* positions,
                 int32_t* seq_lens, int64_t* slot_mapping, const int32_t* block_tables, int64_t bt_stride,
                 hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(advance_step_kernel, dim3((n + 255) / 256), dim3(256), 0, st, n, block_size, input_tokens,
                       sampled, positions, seq_lens, slot_mapping, block_tables, bt_stride);
    return 0;
}


// ---------------------------------------------------------------- spec-decode cycle glue
// The per-cycle input assembly that the reference does on the host between GPU passes
// (TP1DraftModelRunner first-step inputs, vllm/spec_decode/draft_model_runner.py:169-262;
//  MQAScorer.score_proposals, vllm/spec_decode/mqa_scorer.py:12-76;
//  SpecDecodeWorker._create_output_sampler_list bookkeeping, spec_decode_worker.py:972-1063)
// kept on the GPU so that a whole draft+verify+accept cycle is one graph with no host sync.
// Sequence state: seq_lens[b] = L tokens known, KV valid for positions < L-1, last_token[b] = token L-1.
// Empty batch slots (seq_lens[b] <= 0: a finished / not yet admitted request) run through the captured cycle as a
// harmless dummy: token 0 at position 0, context of one key, slot -1 (nothing is written to the cache), no output.
// A position beyond the sequence's block table (max_blocks entries) also gets slot -1 instead of reading past the row:
// the host refuses such a step beforehand (QSpecEngine.step), this is the in-kernel safety net.
__device__ __forceinline__ int64_t spec_slot(const int32_t* block_tables, int64_t bt_stride, int b, int pos,
                                             int block_size, int max_blocks) {
    const int blk = pos / block_size;
    if (pos < 0 || (max_blocks > 0 && blk >= max_blocks)) return -1;
    return (int64_t)block_tables[bt_stride * b + blk] * block_size + pos % block_size;
}
// (each bookkeeping step below is a per-row device function returning the row's input token: the plain kernels run one row per
// thread; the `_embed` kernels run one row per WORKGROUP -- thread 0 does the bookkeeping, then all 256 copy the token's embedding
// row into the forward's hidden buffer (embedding_kernel's copy): one launch of the cycle's chain less per forward)
__device__ __forceinline__ int64_t spec_prepare_draft_row(int b, int block_size, int max_blocks, const int64_t* last_token,
                                                          const int32_t* seq_lens, const int32_t* step_mask, int32_t* eff_lens,
                                                          const int32_t* block_tables, int64_t bt_stride, int64_t* input_tokens,
                                                          int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens) {
    int L = seq_lens[b];
    if (step_mask) {   // eff_lens = seq_lens * step_mask (slots that sit this step out count as empty)
        L *= step_mask[b];
        eff_lens[b] = L;
    }
    const int pos = L - 1;
    if (L <= 0) {
        input_tokens[b] = 0;
        positions[b] = 0;
        ctx_lens[b] = 1;
        slot_mapping[b] = -1;
        return 0;
    }
    const int64_t tok = last_token[b];
    input_tokens[b] = tok;
    positions[b] = pos;
    ctx_lens[b] = L;
    slot_mapping[b] = spec_slot(block_tables, bt_stride, b, pos, block_size, max_blocks);
    return tok;
}
__global__ void spec_prepare_draft_kernel(int B, int block_size, int max_blocks, const int64_t* __restrict__ last_token,
                                          const int32_t* __restrict__ seq_lens,
                                          const int32_t* __restrict__ block_tables, int64_t bt_stride,
                                          int64_t* __restrict__ input_tokens, int64_t* __restrict__ positions,
                                          int64_t* __restrict__ slot_mapping, int32_t* __restrict__ ctx_lens) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    spec_prepare_draft_row(b, block_size, max_blocks, last_token, seq_lens, nullptr, nullptr, block_tables, bt_stride, input_tokens,
                           positions, slot_mapping, ctx_lens);
}
struct SpecEmbed {
    const f16* table;   // [V, H]
    f16* out;           // [rows, H]
    int H, V;
};
__device__ __forceinline__ void spec_embed_row(const SpecEmbed& e, int row, int64_t tok_thread0) {
    __shared__ int64_t s_tok;
    if (threadIdx.x == 0) s_tok = tok_thread0;
    __syncthreads();
    int64_t id = s_tok;
    if (id < 0 || id >= e.V) id = 0;  // padded slot (embedding_kernel's rule)
    const f16* src = e.table + id * e.H;
    for (int i = threadIdx.x; i < e.H / 8; i += blockDim.x)
        *reinterpret_cast<f16x8*>(e.out + (size_t)row * e.H + 8 * i) = *reinterpret_cast<const f16x8*>(src + 8 * i);
}
__global__ __launch_bounds__(256) void spec_prepare_draft_embed_kernel(int block_size, int max_blocks, const int64_t* __restrict__ last_token,
                                                                       const int32_t* __restrict__ seq_lens, const int32_t* __restrict__ step_mask,
                                                                       int32_t* __restrict__ eff_lens, const int32_t* __restrict__ block_tables,
                                                                       int64_t bt_stride, int64_t* __restrict__ input_tokens,
                                                                       int64_t* __restrict__ positions, int64_t* __restrict__ slot_mapping,
                                                                       int32_t* __restrict__ ctx_lens, SpecEmbed e) {
    const int b = blockIdx.x;
    int64_t tok = 0;
    if (threadIdx.x == 0)
        tok = spec_prepare_draft_row(b, block_size, max_blocks, last_token, seq_lens, step_mask, eff_lens, block_tables, bt_stride,
                                     input_tokens, positions, slot_mapping, ctx_lens);
    spec_embed_row(e, b, tok);
}
// _gpu_advance_step between two draft steps (draft_model_runner.py:78-135) with the two engine rules above: a row
// whose slot is -1 (empty, or out of blocks) stays put; a new position beyond the block table gets slot -1.
__device__ __forceinline__ int64_t spec_advance_draft_row(int i, int block_size, int max_blocks, int64_t* input_tokens,
                                                          const int64_t* sampled, int64_t* positions, int32_t* ctx_lens,
                                                          int64_t* slot_mapping, const int32_t* block_tables, int64_t bt_stride) {
    if (slot_mapping[i] < 0) return input_tokens[i];
    const int64_t tok = sampled[i];
    input_tokens[i] = tok;
    const int next_len = ctx_lens[i] + 1;
    const int pos = next_len - 1;
    const int64_t slot = spec_slot(block_tables, bt_stride, i, pos, block_size, max_blocks);
    if (slot < 0) {   // out of blocks: freeze the row (its remaining draft steps recompute the same token)
        slot_mapping[i] = -1;
        return tok;
    }
    ctx_lens[i] = next_len;
    positions[i] = pos;
    slot_mapping[i] = slot;
    return tok;
}
__global__ void spec_advance_draft_kernel(int n, int block_size, int max_blocks, int64_t* __restrict__ input_tokens,
                                          const int64_t* __restrict__ sampled, int64_t* __restrict__ positions,
                                          int32_t* __restrict__ ctx_lens, int64_t* __restrict__ slot_mapping,
                                          const int32_t* __restrict__ block_tables, int64_t bt_stride) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    spec_advance_draft_row(i, block_size, max_blocks, input_tokens, sampled, positions, ctx_lens, slot_mapping, block_tables, bt_stride);
}
__global__ __launch_bounds__(256) void spec_advance_draft_embed_kernel(int block_size, int max_blocks, int64_t* __restrict__ input_tokens,
                                                                       const int64_t* __restrict__ sampled, int64_t* __restrict__ positions,
                                                                       int32_t* __restrict__ ctx_lens, int64_t* __restrict__ slot_mapping,
                                                                       const int32_t* __restrict__ block_tables, int64_t bt_stride, SpecEmbed e) {
    const int i = blockIdx.x;
    int64_t tok = 0;
    if (threadIdx.x == 0)
        tok = spec_advance_draft_row(i, block_size, max_blocks, input_tokens, sampled, positions, ctx_lens, slot_mapping, block_tables,
                                     bt_stride);
    spec_embed_row(e, i, tok);
}
// verify query of sequence b = [last_token, d_1 .. d_k] at positions L-1 .. L-1+k (mqa_scorer.py:42-60)
__device__ __forceinline__ int64_t spec_prepare_verify_row(int i, int k, int block_size, int max_blocks, const int64_t* last_token,
                                                           const int64_t* draft_ids, int64_t di_sb, int64_t di_sk, const int32_t* seq_lens,
                                                           const int32_t* block_tables, int64_t bt_stride, int64_t* v_tokens,
                                                           int64_t* v_positions, int64_t* v_slots, int32_t* v_ctx_lens) {
    const int b = i / (k + 1), j = i % (k + 1);
    const int L = seq_lens[b], pos = L - 1 + j;
    if (L <= 0) {   // empty slot: k + 1 dummy queries over the first k + 1 keys of the row, nothing written
        v_tokens[i] = 0;
        v_positions[i] = j;
        v_slots[i] = -1;
        if (j == 0) v_ctx_lens[b] = k + 1;
        return 0;
    }
    const int64_t tok = j == 0 ? last_token[b] : draft_ids[b * di_sb + (j - 1) * di_sk];
    v_tokens[i] = tok;
    v_positions[i] = pos;
    v_slots[i] = spec_slot(block_tables, bt_stride, b, pos, block_size, max_blocks);
    if (j == 0) v_ctx_lens[b] = L + k;
    return tok;
}
__global__ void spec_prepare_verify_kernel(int B, int k, int block_size, int max_blocks, const int64_t* __restrict__ last_token,
                                           const int64_t* __restrict__ draft_ids, int64_t di_sb, int64_t di_sk,
                                           const int32_t* __restrict__ seq_lens,
                                           const int32_t* __restrict__ block_tables, int64_t bt_stride,
                                           int64_t* __restrict__ v_tokens, int64_t* __restrict__ v_positions,
                                           int64_t* __restrict__ v_slots, int32_t* __restrict__ v_ctx_lens) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B * (k + 1)) return;
    spec_prepare_verify_row(i, k, block_size, max_blocks, last_token, draft_ids, di_sb, di_sk, seq_lens, block_tables, bt_stride, v_tokens,
                            v_positions, v_slots, v_ctx_lens);
}
__global__ __launch_bounds__(256) void spec_prepare_verify_embed_kernel(int k, int block_size, int max_blocks, const int64_t* __restrict__ last_token,
                                                                        const int64_t* __restrict__ draft_ids, int64_t di_sb, int64_t di_sk,
                                                                        const int32_t* __restrict__ seq_lens,
                                                                        const int32_t* __restrict__ block_tables, int64_t bt_stride,
                                                                        int64_t* __restrict__ v_tokens, int64_t* __restrict__ v_positions,
                                                                        int64_t* __restrict__ v_slots, int32_t* __restrict__ v_ctx_lens, SpecEmbed e) {
    const int i = blockIdx.x;
    int64_t tok = 0;
    if (threadIdx.x == 0)
        tok = spec_prepare_verify_row(i, k, block_size, max_blocks, last_token, draft_ids, di_sb, di_sk, seq_lens, block_tables, bt_stride,
                                      v_tokens, v_positions, v_slots, v_ctx_lens);
    spec_embed_row(e, i, tok);
}
// append the emitted tokens (out != -1, a prefix of the row) and advance the sequence state
__global__ void spec_commit_kernel(int B, int k, const int64_t* __restrict__ out_tokens,
                                   int32_t* __restrict__ seq_lens, int64_t* __restrict__ last_token,
                                   int64_t* __restrict__ gen_tokens, int32_t* __restrict__ gen_lens, int gen_cap) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int j = 0; j <= k; j++) {
        int64_t t = out_tokens[b * (k + 1) + j];
        if (t == -1) break;
        if (gen_tokens && gen_lens[b] + n < gen_cap) gen_tokens[(int64_t)b * gen_cap + gen_lens[b] + n] = t;
        n++;
    }
    if (n > 0) {
        last_token[b] = out_tokens[b * (k + 1) + n - 1];
        seq_lens[b] += n;
        if (gen_lens) gen_lens[b] += n;
    }
}
// The small per-cycle sequence state, copied aside at the start of a cycle (restore = the same kernel with the
// arguments swapped): seq_lens / gen_lens [B] i32, last_token [B] i64, counters [3] i64, rng_state [2] i64.
// snap_i32 [2B], snap_i64 [B + 5].
__global__ void spec_snapshot_kernel(int B, const int32_t* __restrict__ seq_lens, const int32_t* __restrict__ gen_lens,
                                     const int64_t* __restrict__ last_token, const int64_t* __restrict__ counters,
                                     const int64_t* __restrict__ rng_state, int32_t* __restrict__ snap_i32,
                                     int64_t* __restrict__ snap_i64) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        snap_i32[i] = seq_lens[i];
        snap_i32[B + i] = gen_lens[i];
        snap_i64[i] = last_token[i];
    }
    if (i < 3) snap_i64[B + i] = counters[i];
    if (i < 2) snap_i64[B + 3 + i] = rng_state[i];
}
__global__ void spec_restore_kernel(int B, int32_t* __restrict__ seq_lens, int32_t* __restrict__ gen_lens,
                                    int64_t* __restrict__ last_token, int64_t* __restrict__ counters,
                                    int64_t* __restrict__ rng_state, const int32_t* __restrict__ snap_i32,
                                    const int64_t* __restrict__ snap_i64) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < B) {
        seq_lens[i] = snap_i32[i];
        gen_lens[i] = snap_i32[B + i];
        last_token[i] = snap_i64[i];
    }
    if (i < 3) counters[i] = snap_i64[B + i];
    if (i < 2) rng_state[i] = snap_i64[B + 3 + i];
}
// out[0] = OR of |*w| over up to four sticky error words (hand-off workspaces, one-shot all-reduce; NULL = none): read
// by the host together with the cycle's output tokens; all-reduced (sum) over the ranks under tensor parallelism.
// The addresses are kernel arguments (constants of a captured graph), not a device table.
__global__ void collect_error_words_kernel(int32_t* w0, int32_t* w1, int32_t* w2, int32_t* w3, int clear,
                                           int64_t* __restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int32_t* ws[4] = {w0, w1, w2, w3};
    int64_t acc = 0;
    for (int i = 0; i < 4; i++) {
        if (!ws[i]) continue;
        const int32_t v = __hip_atomic_load(ws[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        acc |= (int64_t)(v < 0 ? -(int64_t)v : (int64_t)v);
        if (clear) __hip_atomic_store(ws[i], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (out) out[0] = acc;
}
int spec_snapshot(int B, int restore, int32_t* seq_lens, int32_t* gen_lens, int64_t* last_token, int64_t* counters,
                  int64_t* rng_state, int32_t* snap_i32, int64_t* snap_i64, hipStream_t st) {
    const int n = B > 3 ? B : 3;
    if (restore)
        hipLaunchKernelGGL(spec_restore_kernel, dim3((n + 63) / 64), dim3(64), 0, st, B, seq_lens, gen_lens, last_token,
                           counters, rng_state, snap_i32, snap_i64);
    else
        hipLaunchKernelGGL(spec_snapshot_kernel, dim3((n + 63) / 64), dim3(64), 0, st, B, seq_lens, gen_lens, last_token,
                           counters, rng_state, snap_i32, snap_i64);
    return 0;
}
int collect_error_words(int32_t* w0, int32_t* w1, int32_t* w2, int32_t* w3, int clear, int64_t* out, hipStream_t st) {
    hipLaunchKernelGGL(collect_error_words_kernel, dim3(1), dim3(64), 0, st, w0, w1, w2, w3, clear, out);
    return 0;
}
int spec_prepare_draft_embed(int B, int block_size, int max_blocks, const int64_t* last_token, const int32_t* seq_lens,
                             const int32_t* step_mask, int32_t* eff_lens, const int32_t* block_tables, int64_t bt_stride,
                             int64_t* input_tokens, int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens, const f16* table,
                             f16* hidden_out, int H, int V, hipStream_t st) {
    if (B == 0) return 0;
    if (H % 8 || (step_mask && !eff_lens)) return -1;
    hipLaunchKernelGGL(spec_prepare_draft_embed_kernel, dim3(B), dim3(256), 0, st, block_size, max_blocks, last_token, seq_lens, step_mask,
                       eff_lens, block_tables, bt_stride, input_tokens, positions, slot_mapping, ctx_lens, SpecEmbed{table, hidden_out, H, V});
    return 0;
}
int spec_advance_draft_embed(int n, int block_size, int max_blocks, int64_t* input_tokens, const int64_t* sampled, int64_t* positions,
                             int32_t* ctx_lens, int64_t* slot_mapping, const int32_t* block_tables, int64_t bt_stride, const f16* table,
                             f16* hidden_out, int H, int V, hipStream_t st) {
    if (n == 0) return 0;
    if (H % 8) return -1;
    hipLaunchKernelGGL(spec_advance_draft_embed_kernel, dim3(n), dim3(256), 0, st, block_size, max_blocks, input_tokens, sampled, positions,
                       ctx_lens, slot_mapping, block_tables, bt_stride, SpecEmbed{table, hidden_out, H, V});
    return 0;
}
int spec_prepare_verify_embed(int B, int k, int block_size, int max_blocks, const int64_t* last_token, const int64_t* draft_ids,
                              int64_t di_sb, int64_t di_sk, const int32_t* seq_lens, const int32_t* block_tables, int64_t bt_stride,
                              int64_t* v_tokens, int64_t* v_positions, int64_t* v_slots, int32_t* v_ctx_lens, const f16* table,
                              f16* hidden_out, int H, int V, hipStream_t st) {
    if (B == 0) return 0;
    if (H % 8) return -1;
    hipLaunchKernelGGL(spec_prepare_verify_embed_kernel, dim3(B * (k + 1)), dim3(256), 0, st, k, block_size, max_blocks, last_token, draft_ids,
                       di_sb, di_sk, seq_lens, block_tables, bt_stride, v_tokens, v_positions, v_slots, v_ctx_lens,
                       SpecEmbed{table, hidden_out, H, V});
    return 0;
}
int spec_prepare_draft(int B, int block_size, int max_blocks, const int64_t* last_token, const int32_t* seq_lens,
                       const int32_t* block_tables, int64_t bt_stride, int64_t* input_tokens, int64_t* positions,
                       int64_t* slot_mapping, int32_t* ctx_lens, hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(spec_prepare_draft_kernel, dim3((B + 63) / 64), dim3(64), 0, st, B, block_size, max_blocks,
                       last_token, seq_lens, block_tables, bt_stride, input_tokens, positions, slot_mapping, ctx_lens);
    return 0;
}
int spec_advance_draft(int n, int block_size, int max_blocks, int64_t* input_tokens, const int64_t* sampled,
                       int64_t* positions, int32_t* ctx_lens, int64_t* slot_mapping, const int32_t* block_tables,
                       int64_t bt_stride, hipStream_t st) {
    if (n == 0) return 0;
    hipLaunchKernelGGL(spec_advance_draft_kernel, dim3((n + 63) / 64), dim3(64), 0, st, n, block_size, max_blocks,
                       input_tokens, sampled, positions, ctx_lens, slot_mapping, block_tables, bt_stride);
    return 0;
}
int spec_prepare_verify(int B, int k, int block_size, int max_blocks, const int64_t* last_token, const int64_t* draft_ids,
                        int64_t di_sb, int64_t di_sk, const int32_t* seq_lens, const int32_t* block_tables, int64_t bt_stride, int64_t* v_tokens,
                        int64_t* v_positions, int64_t* v_slots, int32_t* v_ctx_lens, hipStream_t st) {
    if (B == 0) return 0;
    const int n = B * (k + 1);
    hipLaunchKernelGGL(spec_prepare_verify_kernel, dim3((n + 63) / 64), dim3(64), 0, st, B, k, block_size, max_blocks, last_token,
                       draft_ids, di_sb, di_sk, seq_lens, block_tables, bt_stride, v_tokens, v_positions, v_slots, v_ctx_lens);
    return 0;
}
int spec_commit(int B, int k, const int64_t* out_tokens, int32_t* seq_lens, int64_t* last_token, int64_t* gen_tokens,
                int32_t* gen_lens, int gen_cap, hipStream_t st) {
    if (B == 0) return 0;
    hipLaunchKernelGGL(spec_commit_kernel, dim3((B + 63) / 64), dim3(64), 0, st, B, k, out_tokens, seq_lens,
                       last_token, gen_tokens, gen_lens, gen_cap);
    return 0;
}

}  // namespace qspec
