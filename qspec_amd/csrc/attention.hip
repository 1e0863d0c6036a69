// RoPE, paged KV-cache write and paged causal attention for the two QSpec
// query shapes (draft q_len = 1, verify q_len = k+1).
//
// Replaces (reference, relative to /root/reference):
//   rotary_embedding_kernel              csrc/pos_encoding_kernels.cu:10-35,71-92  (call: quarot_llama.py:207-211)
//   reshape_and_cache_flash_kernel       csrc/cache_kernels.cu:207-247             (call: flash_attn.py:711-720)
//   flash_attn_with_kvcache / flash_attn_varlen_func (vllm_flash_attn wheel, un-vendored; call sites
//                                        vllm/attention/backends/flash_attn.py:741-830)
//
// KV cache layout is vLLM's flash layout: [num_blocks, block_size, n_kv, d] fp16,
// one tensor for K and one for V, shared by the draft and the verify pass.
#include "common.cuh"
#include "kernels.h"

namespace qspec {

// NeoX rotation with every operator rounded to fp16, as c10::Half arithmetic does.
__device__ __forceinline__ void rope_pair(f16& x, f16& y, f16 c, f16 s) {
    float xf = h2f(x), yf = h2f(y), cf = h2f(c), sf = h2f(s);
    f16 nx = f2h(h2f(f2h(xf * cf)) - h2f(f2h(yf * sf)));
    f16 ny = f2h(h2f(f2h(yf * cf)) + h2f(f2h(xf * sf)));
    x = nx;
    y = ny;
}

__global__ __launch_bounds__(256) void rotary_embedding_kernel(const int64_t* __restrict__ positions,
                                                               f16* __restrict__ q, f16* __restrict__ k,
                                                               const f16* __restrict__ cos_sin_cache, int nq, int nkv,
                                                               int d, int rot_dim, int64_t q_stride,
                                                               int64_t k_stride) {
    const int t = blockIdx.x;
    const int embed = rot_dim / 2;
    const f16* cs = cos_sin_cache + positions[t] * rot_dim;
    for (int i = threadIdx.x; i < (nq + nkv) * embed; i += blockDim.x) {
        int h = i / embed, o = i % embed;
        f16* base = h < nq ? q + t * q_stride + (size_t)h * d : k + t * k_stride + (size_t)(h - nq) * d;
        rope_pair(base[o], base[embed + o], cs[o], cs[embed + o]);
    }
}

int rotary_embedding(const int64_t* positions, f16* q, f16* k, const f16* cos_sin_cache, int T, int nq, int nkv, int d,
                     int rot_dim, int64_t q_stride, int64_t k_stride, hipStream_t st) {
    if (T == 0) return 0;
    if (rot_dim > d || rot_dim % 2) return -1;
    hipLaunchKernelGGL(rotary_embedding_kernel, dim3(T), dim3(256), 0, st, positions, q, k, cos_sin_cache, nq, nkv, d,
                       rot_dim, q_stride, k_stride);
    return 0;
}

__global__ __launch_bounds__(256) void reshape_and_cache_flash_kernel(const f16* __restrict__ key,
                                                                      const f16* __restrict__ value,
                                                                      f16* __restrict__ key_cache,
                                                                      f16* __restrict__ value_cache,
                                                                      const int64_t* __restrict__ slot_mapping, int n,
                                                                      int64_t k_stride, int64_t v_stride) {
    const int t = blockIdx.x;
    const int64_t slot = slot_mapping[t];
    if (slot < 0) return;  // padded token
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        key_cache[slot * n + i] = key[t * k_stride + i];
        value_cache[slot * n + i] = value[t * v_stride + i];
    }
}

int reshape_and_cache_flash(const f16* key, const f16* value, f16* key_cache, f16* value_cache,
                            const int64_t* slot_mapping, int T, int nkv, int d, int64_t k_stride, int64_t v_stride,
                            hipStream_t st) {
    if (T == 0) return 0;
    hipLaunchKernelGGL(reshape_and_cache_flash_kernel, dim3(T), dim3(256), 0, st, key, value, key_cache, value_cache,
                       slot_mapping, nkv * d, k_stride, v_stride);
    return 0;
}

// Fused: in-place RoPE on the q and k slices of the fused qkv row, then scatter k,v into the paged cache.
// qkv [T, (nq + 2 nkv) * d]; one workgroup per token, 8 fp16 (16 B) per lane trip.
__global__ __launch_bounds__(256) void rope_kv_write_kernel(const int64_t* __restrict__ positions,
                                                            f16* __restrict__ qkv,
                                                            const f16* __restrict__ cos_sin_cache,
                                                            f16* __restrict__ key_cache, f16* __restrict__ value_cache,
                                                            const int64_t* __restrict__ slot_mapping, int nq, int nkv,
                                                            int d, int rot_dim) {
    const int t = blockIdx.x;
    const int embed = rot_dim / 2;
    const int row = (nq + 2 * nkv) * d;
    f16* base = qkv + (size_t)t * row;
    const f16* cs = cos_sin_cache + positions[t] * rot_dim;
    const int64_t slot = slot_mapping[t];
    const int vec_per_head = embed / 8;  // 16-byte chunks of the x half
    for (int i = threadIdx.x; i < (nq + nkv) * vec_per_head; i += blockDim.x) {
        int h = i / vec_per_head, o = (i % vec_per_head) * 8;
        f16* hp = base + (size_t)h * d;
        f16x8 x = *reinterpret_cast<f16x8*>(hp + o), y = *reinterpret_cast<f16x8*>(hp + embed + o);
        f16x8 c = *reinterpret_cast<const f16x8*>(cs + o), s = *reinterpret_cast<const f16x8*>(cs + embed + o);
#pragma unroll
        for (int e = 0; e < 8; e++) {
            f16 xe = x[e], ye = y[e];
            rope_pair(xe, ye, c[e], s[e]);
            x[e] = xe;
            y[e] = ye;
        }
        *reinterpret_cast<f16x8*>(hp + o) = x;
        *reinterpret_cast<f16x8*>(hp + embed + o) = y;
        if (h >= nq && slot >= 0) {
            f16* kc = key_cache + (slot * nkv + (h - nq)) * d;
            *reinterpret_cast<f16x8*>(kc + o) = x;
            *reinterpret_cast<f16x8*>(kc + embed + o) = y;
        }
    }
    if (slot >= 0) {
        const f16* vsrc = base + (size_t)(nq + nkv) * d;
        f16* vc = value_cache + slot * nkv * d;
        for (int i = threadIdx.x; i < nkv * d / 8; i += blockDim.x)
            *reinterpret_cast<f16x8*>(vc + 8 * i) = *reinterpret_cast<const f16x8*>(vsrc + 8 * i);
        if (rot_dim < d) {  // un-rotated tail of k
            for (int i = threadIdx.x; i < nkv * (d - rot_dim); i += blockDim.x) {
                int h = i / (d - rot_dim), o = rot_dim + i % (d - rot_dim);
                key_cache[(slot * nkv + h) * d + o] = base[(size_t)(nq + h) * d + o];
            }
        }
    }
}

int rope_kv_write(const int64_t* positions, f16* qkv, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                  const int64_t* slot_mapping, int T, int nq, int nkv, int d, int rot_dim, hipStream_t st) {
    if (T == 0) return 0;
    if (rot_dim > d || rot_dim % 16 || d % 8) return -1;
    hipLaunchKernelGGL(rope_kv_write_kernel, dim3(T), dim3(256), 0, st, positions, qkv, cos_sin_cache, key_cache,
                       value_cache, slot_mapping, nq, nkv, d, rot_dim);
    return 0;
}

// ---------------------------------------------------------------- attention
// Split-context ("flash-decoding") paged attention.
//   grid (n_seqs, n_kv_heads, n_splits); each workgroup takes the keys
//   [split*chunk, (split+1)*chunk) of one sequence and one kv head, and all
//   R = q_len * group query rows that share that kv head (R <= 16 for
//   k+1 = 4 tokens x GQA group 4).  16 lanes cover one key row (16 B each),
//   so a wave reads 4 whole 256-byte key rows per trip.
//   Pass 1: scores -> LDS; pass 2: max/exp/sum; pass 3: P.V; partial
//   (o, m, l) per split in fp32, merged by attn_combine_kernel.
// d = 128 only (16 lanes x 8 dims).
#define QS_ATT_MAXR 16
#define QS_ATT_CHUNK 128

__global__ __launch_bounds__(256) void paged_attention_kernel(
    const f16* __restrict__ q, int64_t q_stride, const f16* __restrict__ key_cache, const f16* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ ctx_lens,
    const int32_t* __restrict__ q_start, int nq, int nkv, int block_size, float sm_scale, int n_splits,
    int n_rb, float* __restrict__ ws_o, float* __restrict__ ws_ml) {
    constexpr int D = 128;
    __shared__ __attribute__((aligned(16))) float q_lds[QS_ATT_MAXR][D];
    __shared__ float sc[QS_ATT_MAXR][QS_ATT_CHUNK];
    __shared__ float row_m[QS_ATT_MAXR], row_l[QS_ATT_MAXR];
    const int seq = blockIdx.x, kvh = blockIdx.y / n_rb, rb = blockIdx.y % n_rb, split = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int group = nq / nkv;
    const int qs = q_start[seq], qlen = q_start[seq + 1] - qs;
    const int r0 = rb * QS_ATT_MAXR;                       // first query row of this row-block
    const int R = min(QS_ATT_MAXR, qlen * group - r0);     // rows handled here
    if (R <= 0) return;                                     // uniform for the whole workgroup
    const int ctx = ctx_lens[seq];
    const int k_begin = split * QS_ATT_CHUNK;
    const int k_end = min(ctx, k_begin + QS_ATT_CHUNK);
    const int nkeys = max(0, k_end - k_begin);
    // row r <-> (token i = (r0+r) / group, head = kvh*group + (r0+r) % group)
    for (int i = tid; i < R * D; i += 256) {
        int r = i / D, e = i % D;
        int tok = qs + (r0 + r) / group, head = kvh * group + (r0 + r) % group;
        q_lds[r][e] = h2f(q[(size_t)tok * q_stride + (size_t)head * D + e]);
    }
    __syncthreads();
    const int sub = lane >> 4, dl = lane & 15;  // 4 keys per wave trip, 8 dims per lane
    const int32_t* bt = block_tables + (size_t)seq * max_blocks;
    // pass 1: scores
    for (int kk = wave * 4 + sub; kk < nkeys; kk += 16) {
        int p = k_begin + kk;
        int64_t slot = (int64_t)bt[p / block_size] * block_size + p % block_size;
        f16x8 kv = *reinterpret_cast<const f16x8*>(key_cache + (slot * nkv + kvh) * D + dl * 8);
        float kf[8];
#pragma unroll
        for (int e = 0; e < 8; e++) kf[e] = h2f(kv[e]);
        for (int r = 0; r < R; r++) {
            float acc = 0.0f;
#pragma unroll
            for (int e = 0; e < 8; e++) acc = __builtin_fmaf(q_lds[r][dl * 8 + e], kf[e], acc);
#pragma unroll
            for (int m = 8; m > 0; m >>= 1) acc += shfl_xor_f(acc, m);
            if (dl == 0) {
                int pos = ctx - qlen + (r0 + r) / group;  // absolute position of this query token
                sc[r][kk] = p <= pos ? acc * sm_scale : -__builtin_inff();
            }
        }
    }
    __syncthreads();
    // pass 2: per-row max / exp / sum (one wave per rows r = wave, wave+4, ...)
    for (int r = wave; r < R; r += 4) {
        float mx = -__builtin_inff();
        for (int kk = lane; kk < nkeys; kk += 64) mx = fmaxf(mx, sc[r][kk]);
        mx = wave_max_f(mx);
        float sum = 0.0f;
        for (int kk = lane; kk < nkeys; kk += 64) {
            float pv = mx == -__builtin_inff() ? 0.0f : qexpf(sc[r][kk] - mx);
            sc[r][kk] = pv;
            sum += pv;
        }
#pragma unroll
        for (int m = 32; m > 0; m >>= 1) sum += shfl_xor_f(sum, m);
        if (lane == 0) {
            row_m[r] = mx;
            row_l[r] = sum;
        }
    }
    __syncthreads();
    // pass 3: o[r][:] = sum_kk p[r][kk] * v[kk][:]; each 16-lane group owns keys kk = g16, g16+16, ...
    float oacc[QS_ATT_MAXR][8];
#pragma unroll
    for (int r = 0; r < QS_ATT_MAXR; r++)
#pragma unroll
        for (int e = 0; e < 8; e++) oacc[r][e] = 0.0f;
    for (int kk = wave * 4 + sub; kk < nkeys; kk += 16) {
        int p = k_begin + kk;
        int64_t slot = (int64_t)bt[p / block_size] * block_size + p % block_size;
        f16x8 vv = *reinterpret_cast<const f16x8*>(value_cache + (slot * nkv + kvh) * D + dl * 8);
        float vf[8];
#pragma unroll
        for (int e = 0; e < 8; e++) vf[e] = h2f(vv[e]);
#pragma unroll
        for (int r = 0; r < QS_ATT_MAXR; r++) {
            if (r < R) {
                float pv = sc[r][kk];
#pragma unroll
                for (int e = 0; e < 8; e++) oacc[r][e] = __builtin_fmaf(pv, vf[e], oacc[r][e]);
            }
        }
    }
    // reduce the 16 key-groups (4 per wave x 4 waves): lanes with equal dl across sub via shuffles, waves via LDS
    __syncthreads();  // sc no longer needed; reuse q_lds as the cross-wave buffer [wave][r][D]... needs 4*16*128 floats
    float* xw = &q_lds[0][0];  // 16*128 floats = one wave's worth; accumulate wave by wave
#pragma unroll
    for (int r = 0; r < QS_ATT_MAXR; r++)
#pragma unroll
        for (int e = 0; e < 8; e++) {
            float v = oacc[r][e];
            v += shfl_xor_f(v, 16);
            v += shfl_xor_f(v, 32);
            oacc[r][e] = v;
        }
    for (int w = 0; w < 4; w++) {
        if (wave == w && sub == 0) {
#pragma unroll
            for (int r = 0; r < QS_ATT_MAXR; r++)
                if (r < R)
#pragma unroll
                    for (int e = 0; e < 8; e++) {
                        float prev = w == 0 ? 0.0f : xw[r * D + dl * 8 + e];
                        xw[r * D + dl * 8 + e] = prev + oacc[r][e];
                    }
        }
        __syncthreads();
    }
    // write partials: ws_o [T, nq, n_splits, D], ws_ml [T, nq, n_splits, 2]
    for (int i = tid; i < R * D; i += 256) {
        int r = i / D, e = i % D;
        int tok = qs + (r0 + r) / group, head = kvh * group + (r0 + r) % group;
        ws_o[(((size_t)tok * nq + head) * n_splits + split) * D + e] = xw[i];
    }
    for (int r = tid; r < R; r += 256) {
        int tok = qs + (r0 + r) / group, head = kvh * group + (r0 + r) % group;
        size_t o = (((size_t)tok * nq + head) * n_splits + split) * 2;
        ws_ml[o] = row_m[r];
        ws_ml[o + 1] = row_l[r];
    }
}

// out[t, h, :] = sum_s e^(m_s - M) o_s / sum_s e^(m_s - M) l_s
__global__ __launch_bounds__(128) void attn_combine_kernel(const float* __restrict__ ws_o,
                                                           const float* __restrict__ ws_ml, f16* __restrict__ out,
                                                           int n_splits) {
    constexpr int D = 128;
    const size_t th = blockIdx.x;  // token * nq + head
    const int e = threadIdx.x;
    float M = -__builtin_inff();
    for (int s = 0; s < n_splits; s++) M = fmaxf(M, ws_ml[(th * n_splits + s) * 2]);
    float num = 0.0f, den = 0.0f;
    for (int s = 0; s < n_splits; s++) {
        float m = ws_ml[(th * n_splits + s) * 2], l = ws_ml[(th * n_splits + s) * 2 + 1];
        if (m == -__builtin_inff()) continue;
        float w = qexpf(m - M);
        num = __builtin_fmaf(w, ws_o[(th * n_splits + s) * D + e], num);
        den = __builtin_fmaf(w, l, den);
    }
    out[th * D + e] = f2h(num / den);
}

size_t paged_attention_ws_bytes(int T, int nq, int d, int n_splits) {
    return (size_t)T * nq * n_splits * (d + 2) * sizeof(float);
}

int paged_attention(const f16* q, int64_t q_stride, const f16* key_cache, const f16* value_cache,
                    const int32_t* block_tables, int max_blocks, const int32_t* ctx_lens, const int32_t* q_start,
                    int n_seqs, int max_q_len, int nq, int nkv, int d, int block_size, float sm_scale, int n_splits,
                    float* ws, f16* out, hipStream_t st) {
    if (n_seqs == 0) return 0;
    if (d != 128 || nq % nkv) return -1;
    if (n_splits < 1) return -3;
    const int n_rb = (max_q_len * (nq / nkv) + QS_ATT_MAXR - 1) / QS_ATT_MAXR;
    // T is not known here; the host sizes ws with paged_attention_ws_bytes(T,...) and lays it out as [o | ml]
    // with o first: the split between them is passed through max tokens = n_seqs * max_q_len.
    const size_t Tmax = (size_t)n_seqs * max_q_len;
    float* ws_o = ws;
    float* ws_ml = ws + Tmax * nq * n_splits * d;
    hipLaunchKernelGGL(paged_attention_kernel, dim3(n_seqs, nkv * n_rb, n_splits), dim3(256), 0, st, q, q_stride,
                       key_cache, value_cache, block_tables, max_blocks, ctx_lens, q_start, nq, nkv, block_size,
                       sm_scale, n_splits, n_rb, ws_o, ws_ml);
    return 0;
}

int paged_attention_combine(const float* ws, int T, int Tmax, int nq, int d, int n_splits, f16* out, hipStream_t st) {
    if (T == 0) return 0;
    const float* ws_o = ws;
    const float* ws_ml = ws + (size_t)Tmax * nq * n_splits * d;
    hipLaunchKernelGGL(attn_combine_kernel, dim3(T * nq), dim3(d), 0, st, ws_o, ws_ml, out, n_splits);
    return 0;
}

}  // namespace qspec
