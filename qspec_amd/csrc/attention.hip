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
#include <stdlib.h>

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
// Split-context ("flash-decoding") paged attention, one launch, both products on the matrix cores.
//   grid (n_seqs, n_kv_heads * n_rb, n_splits), 256 threads = 4 waves.  A workgroup takes the keys
//   [split*128, split*128+128) of one sequence and one kv head, and up to 16 query rows (row = query token x
//   GQA head of that kv head: 4 rows for the draft pass, 16 for verify k=3; longer queries use n_rb row blocks).
//
//   At these sizes a wave's INSTRUCTION COUNT is the cost (one wave per SIMD issues ~1 instruction / 4 cycles),
//   so the kernel is shaped to need few of them:
//     * S = Q K^T: Q and K are both d-contiguous, so MFMA fragments are plain 16-byte global loads
//       (wave w owns key tiles 2w, 2w+1); 8 v_mfma_f32_16x16x32_f16 per wave.
//     * softmax: S (fp32) sits in LDS as [row][key]; 16 lanes share a row, reduce with 4 shuffles, P goes back
//       as fp16 [row][key] = exactly the A fragment layout of the second product.
//     * O = P V: V rows are staged once in LDS ([key][d], 288-byte row stride) and read back TRANSPOSED with
//       ds_read_b64_tr_b16, so B fragments cost 2 LDS reads instead of 8 gathers; wave w owns output columns
//       32w..32w+31 (no cross-wave reduction); 8 MFMAs per wave.
//     * block_size and the GQA group are powers of two: no integer division anywhere.
//   Every global load of the workgroup is issued before the first use (one dependent chain:
//   block table -> K/V).  The partial (o, m, l) of a split goes to the workspace; the LAST workgroup to
//   arrive for a (sequence, kv head, row block) merges the splits in split order (deterministic) and writes
//   fp16 -- agent-scope release / acquire around a relaxed ticket counter (reset for the next call).
// d = 128 only.
#define QS_ATT_MAXR 16
#define QS_ATT_CHUNK 128
#define QS_ATT_CNT_SLOTS 4096  // ints reserved at the head of the workspace for the ticket counters
#define QS_ATT_VSTRIDE 144     // halves per V row in LDS (288 B): 4 rows x 4 column quads hit 16 distinct bank pairs
#define QS_ATT_MAXSPLIT 64

// ds_read_b64_tr_b16 as a compiler builtin: the register allocator places the two halves of a B fragment next to each
// other (no v_mov pairs) and the waits are the compiler's (per use, not lgkmcnt(0) behind every read, as the inline-asm
// form of rounds 1-2 had it).
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ s16x4 lds_tr16(uint32_t lds_byte_addr) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<__attribute__((address_space(3))) s16x4*>(lds_byte_addr));
}

// Eight d tiles of one 32-key step at once (the prompt kernel): 16 transposing reads from one base address with
// immediate offsets (d tile: +32 B, second key quad: +4 V rows), ONE wait behind all of them.
__device__ __forceinline__ void lds_read_tr_b16_x16(uint32_t a, u32x2 (&lo)[8], u32x2 (&hi)[8]) {
    static_assert(QS_ATT_VSTRIDE * 8 == 1152, "immediate offsets below assume 288-byte V rows");
    asm volatile(
        "ds_read_b64_tr_b16 %0, %16 offset:0\n\tds_read_b64_tr_b16 %8, %16 offset:1152\n\t"
        "ds_read_b64_tr_b16 %1, %16 offset:32\n\tds_read_b64_tr_b16 %9, %16 offset:1184\n\t"
        "ds_read_b64_tr_b16 %2, %16 offset:64\n\tds_read_b64_tr_b16 %10, %16 offset:1216\n\t"
        "ds_read_b64_tr_b16 %3, %16 offset:96\n\tds_read_b64_tr_b16 %11, %16 offset:1248\n\t"
        "ds_read_b64_tr_b16 %4, %16 offset:128\n\tds_read_b64_tr_b16 %12, %16 offset:1280\n\t"
        "ds_read_b64_tr_b16 %5, %16 offset:160\n\tds_read_b64_tr_b16 %13, %16 offset:1312\n\t"
        "ds_read_b64_tr_b16 %6, %16 offset:192\n\tds_read_b64_tr_b16 %14, %16 offset:1344\n\t"
        "ds_read_b64_tr_b16 %7, %16 offset:224\n\tds_read_b64_tr_b16 %15, %16 offset:1376\n\t"
        "s_waitcnt lgkmcnt(0)"
        : "=&v"(lo[0]), "=&v"(lo[1]), "=&v"(lo[2]), "=&v"(lo[3]), "=&v"(lo[4]), "=&v"(lo[5]), "=&v"(lo[6]), "=&v"(lo[7]),
          "=&v"(hi[0]), "=&v"(hi[1]), "=&v"(hi[2]), "=&v"(hi[3]), "=&v"(hi[4]), "=&v"(hi[5]), "=&v"(hi[6]), "=&v"(hi[7])
        : "v"(a)
        : "memory");
}

// PF: the loads of chunk i+1 fly underneath chunk i (64 more VGPRs: one workgroup per CU -- the decode shape at small
// batch, where the launch is 256 workgroups of dependent round trips).  PF = false halves the register footprint so that
// two workgroups share a CU: large batches, where occupancy hides the same latency.
// BT: the sequence's block-table row (<= 128 entries) is fetched into two registers per lane together with the
// per-sequence metadata, and the chunk lookups become lane permutes: the dependent chain metadata -> table entry -> K / V
// loses one memory round trip.
// (A wave-uniform form of the table lookup -- a K tile / the 16-key group of a V row lies in ONE block when block_size >= 16,
// so v_readlane + SALU instead of 20 ds_bpermute -- is what paged_attention_waves_kernel<.., FAST> uses; HERE it measured
// slower, 6.16 against 6.05 us per launch and 7.57 against 7.53 ms per cycle: the serial SALU chain is longer than the
// pipelined permutes.)
template <bool PF, bool BT>
__global__ __launch_bounds__(256) void paged_attention_kernel(
    const f16* __restrict__ q, int64_t q_stride, const f16* __restrict__ key_cache, const f16* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ ctx_lens,
    const int32_t* __restrict__ q_start, int nq, int nkv, int bs_log2, int group_log2, float sm_scale, int n_splits,
    int n_rb, int* cnt, float* ws_o, float* ws_ml, f16* __restrict__ out, int merge) {
    constexpr int D = 128;
#ifdef QS_ATT_STAMPS
    long long stamp[10];
#define QS_STAMP(i) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(stamp[i])::"memory")
#else
#define QS_STAMP(i)
#endif
    QS_STAMP(9);   // entry
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sc = reinterpret_cast<float*>(smem_raw);                          // [16 rows][128 keys] fp32 scores
    f16* pl = reinterpret_cast<f16*>(sc + QS_ATT_MAXR * QS_ATT_CHUNK);       // [16 rows][128 keys] fp16 probabilities
    f16* pl2 = pl + QS_ATT_MAXR * QS_ATT_CHUNK;                              // low half of the fp32 probability (p - fp16(p))
    f16* vl = pl2 + QS_ATT_MAXR * QS_ATT_CHUNK;                              // [128 keys][QS_ATT_VSTRIDE]
    float* row_m = reinterpret_cast<float*>(vl + QS_ATT_CHUNK * QS_ATT_VSTRIDE);  // [16]
    float* row_l = row_m + QS_ATT_MAXR;                                      // [16]
    float* wgt = row_l + QS_ATT_MAXR;                                        // [16][QS_ATT_MAXSPLIT] merge weights
    int* flag = reinterpret_cast<int*>(wgt + QS_ATT_MAXR * QS_ATT_MAXSPLIT); // [1]
    const int seq = blockIdx.x, kvh = blockIdx.y / n_rb, rb = blockIdx.y % n_rb, split = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int gmask = (1 << group_log2) - 1, bmask = (1 << bs_log2) - 1;
    // per-sequence metadata through the VECTOR memory path: a uniform address would become an s_load, and a scalar
    // cache miss on data the previous kernel has just written costs microseconds, not hundreds of cycles
    int btv0 = 0, btv1 = 0;
    if constexpr (BT) {   // in the same round trip as the metadata (the early return below depends on it)
        const int32_t* btr = block_tables + (size_t)blockIdx.x * max_blocks;
        btv0 = btr[min((int)(threadIdx.x & 63), max_blocks - 1)];
        btv1 = btr[min((int)(threadIdx.x & 63) + 64, max_blocks - 1)];
    }
    int meta = 0;
    if (lane < 3) meta = lane < 2 ? q_start[seq + lane] : ctx_lens[seq];
    const int qs = __shfl(meta, 0, 64), qlen = __shfl(meta, 1, 64) - qs;
    const int ctx = __shfl(meta, 2, 64);
    const int r0 = rb * QS_ATT_MAXR;
    const int R = min(QS_ATT_MAXR, (qlen << group_log2) - r0);
    if (R <= 0) return;  // uniform for the whole workgroup
    // Keys per split adapt to the sequence: ceil(ctx / n_splits) rounded up to a 16-key tile, so that a short context
    // still spreads over all n_splits workgroups (each CU only draws ~25 GB/s).  A split longer than 128 keys is walked
    // in 128-key chunks with a running (max, sum, output) -- flash-decoding with an inner loop -- so the number of
    // splits (and of partials to merge) stays put while the context grows; the loads of chunk i+1 are issued before
    // chunk i is consumed.  One chunk per split is the same arithmetic as before the loop existed.
    const int kps = (((ctx + n_splits - 1) / n_splits) + 15) & ~15;
    const int k_begin = split * kps;
    const int k_end = min(ctx, k_begin + kps);
    const int n_it = max(1, (min(kps, max(k_end - k_begin, 0)) + QS_ATT_CHUNK - 1) / QS_ATT_CHUNK);
    const int c16 = lane & 15, g4 = lane >> 4;
    const int32_t* bt = block_tables + (size_t)seq * max_blocks;
    auto bt_get = [&](int bi) -> int {   // bi already clamped to the table
        if constexpr (BT) {
            const int a = __shfl(btv0, bi & 63, 64), b = __shfl(btv1, bi & 63, 64);
            return bi < 64 ? a : b;
        } else {
            return bt[bi];
        }
    };
    QS_STAMP(0);

    // No load below sits behind a branch (hipcc answers control flow around a load with s_waitcnt vmcnt(0), which
    // would serialise the prefetch): block-table indices and key rows are clamped, out-of-range keys are masked
    // after the load (scores -> -inf, V rows -> 0).
    struct Slots {
        int64_t k[2], v[8];
    };
    struct KV {
        u32x4 kfrag[2][4], vraw[8];
    };
    // Keys past the split's end are masked after the load; their ADDRESSES are those of the split's last key, so that a
    // 64-key split does not pull the neighbouring split's 64 keys through this CU as well (a chunk is 128 keys: twice the
    // bytes in flight, i.e. a second round trip of the CU's ~32 KB window -- 6.7 us per launch at 8 splits of a 512-key
    // context).  Lanes with equal addresses coalesce inside the wave instruction.
    const int k_last = max(k_end - 1, 0);
    // Slots hold ELEMENT offsets of the rows' first element of this kv head: (slot * nkv + kvh) * D
    auto lookup = [&](Slots& sl, int kb) {   // block-table entries of the chunk starting at key kb
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
            const int p = min(kb + (wave * 2 + t2) * 16 + c16, k_last);
            sl.k[t2] = ((((int64_t)bt_get(min(p >> bs_log2, max_blocks - 1)) << bs_log2) + (p & bmask)) * nkv + kvh) * D;
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int p = min(kb + wave * 4 + g4 + 16 * i, k_last);
            sl.v[i] = ((((int64_t)bt_get(min(p >> bs_log2, max_blocks - 1)) << bs_log2) + (p & bmask)) * nkv + kvh) * D;
        }
    };
    auto fetch = [&](KV& kv, const Slots& sl) {
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {  // lane (key c16 of tile 2w+t2, d slice 8*g4 + 32j)
            const f16* kp = key_cache + sl.k[t2] + g4 * 8;
#pragma unroll
            for (int j = 0; j < 4; j++) kv.kfrag[t2][j] = *reinterpret_cast<const u32x4*>(kp + 32 * j);
        }
#pragma unroll
        for (int i = 0; i < 8; i++)  // 16 lanes cover one 256-byte V row; wave w, group g4 -> keys 4w+g4 + 16i
            kv.vraw[i] = *reinterpret_cast<const u32x4*>(value_cache + sl.v[i] + c16 * 8);
    };
    Slots sl_cur, sl_nxt;
    lookup(sl_cur, k_begin);
    if constexpr (PF) lookup(sl_nxt, k_begin + (n_it > 1 ? QS_ATT_CHUNK : 0));
    u32x4 qfrag[4];
    {  // lane (row c16, d slice 8*g4 + 32j); rows >= R repeat row 0 (their outputs are never stored)
        const int r = r0 + (c16 < R ? c16 : 0);
        const int tok = qs + (r >> group_log2), head = (kvh << group_log2) + (r & gmask);
        const f16* qp = q + (size_t)tok * q_stride + (size_t)head * D + g4 * 8;
#pragma unroll
        for (int j = 0; j < 4; j++) qfrag[j] = *reinterpret_cast<const u32x4*>(qp + 32 * j);
    }
    KV cur, nxt;
    fetch(cur, sl_cur);
    if (tid < QS_ATT_MAXR) {
        row_m[tid] = -__builtin_inff();
        row_l[tid] = 0.0f;
    }
    float* row_a = wgt;   // [16] rescale factors of the running output (wgt is free until the merge)
    f32x4 o0 = {0.f, 0.f, 0.f, 0.f}, o1 = {0.f, 0.f, 0.f, 0.f};
    QS_STAMP(1);

    int qpos[4];   // absolute position of the query token of this lane's score rows 4 g4 + reg
#pragma unroll
    for (int reg = 0; reg < 4; reg++) qpos[reg] = ctx - qlen + ((r0 + 4 * g4 + reg) >> group_log2);
    auto process = [&](const KV& kv, int kb, bool last) {
        const int nkeys = __builtin_amdgcn_readfirstlane(max(0, min(k_end - kb, QS_ATT_CHUNK)));   // uniform: scalar branches below
        // ---- S = Q K^T (fp32 accumulate) -> sc[row][key], masked and scaled
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, qfrag[j]),
                                                             __builtin_bit_cast(f16x8, kv.kfrag[t2][j]), acc, 0, 0, 0);
            const int kk = (wave * 2 + t2) * 16 + c16;  // lane holds rows 4*g4 + reg of key column kk
            const int p = kb + kk;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {   // one compare-and-select per score: no short-circuit branches
                const bool ok = (kk < nkeys) & (p <= qpos[reg]);
                sc[(4 * g4 + reg) * QS_ATT_CHUNK + kk] = ok ? acc[reg] * sm_scale : -__builtin_inff();
            }
        }
        // V rows -> LDS: only the 32-key steps P.V walks (st * 32 < nkeys), zeros beyond nkeys inside the last of them
        // (uniform branches around LDS stores; no global load sits behind them)
        if ((nkeys & 31) == 0) {
#pragma unroll
            for (int i = 0; i < 8; i++)
                if (16 * i < nkeys)
                    *reinterpret_cast<u32x4*>(vl + (wave * 4 + g4 + 16 * i) * QS_ATT_VSTRIDE + c16 * 8) = kv.vraw[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int kk = wave * 4 + g4 + 16 * i;
                const u32x4 vz = kk < nkeys ? kv.vraw[i] : u32x4{0, 0, 0, 0};
                if (16 * i < ((nkeys + 31) & ~31)) *reinterpret_cast<u32x4*>(vl + kk * QS_ATT_VSTRIDE + c16 * 8) = vz;
            }
        }
        if (last) QS_STAMP(3);
        __syncthreads();
        // ---- softmax over the 128 keys of a row, folded into the running (max, sum): thread (row = tid>>4, 8 keys)
        {
            const int r = tid >> 4, ks = tid & 15;
            const float4 a = *reinterpret_cast<const float4*>(sc + r * QS_ATT_CHUNK + ks * 8);
            const float4 b = *reinterpret_cast<const float4*>(sc + r * QS_ATT_CHUNK + ks * 8 + 4);
            float s8[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
            float mx = s8[0];
#pragma unroll
            for (int i = 1; i < 8; i++) mx = fmaxf(mx, s8[i]);
            // the 16 lanes of a row exchange through DPP (max: order independent)
            mx = fmaxf(mx, dpp_xor<8>(mx));
            mx = fmaxf(mx, dpp_xor<4>(mx));
            mx = fmaxf(mx, dpp_xor<2>(mx));
            mx = fmaxf(mx, dpp_xor<1>(mx));
            const float m_old = row_m[r];              // the row's 16 lanes sit in one wave: read before lane 0 writes
            const float m_new = fmaxf(m_old, mx);
            float sum = 0.0f;
            f16x8 p8, p8lo;
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const float pv = m_new == -__builtin_inff() ? 0.0f : aexp(s8[i] - m_new);
                // the matrix core takes fp16 operands: carry the fp32 probability as hi + lo so that P.V keeps
                // fp32-class accuracy (22 bits) for two MFMAs instead of one
                const f16 ph = f2h(pv);
                p8[i] = ph;
                p8lo[i] = f2h(pv - h2f(ph));
                sum += pv;
            }
            sum += dpp_xor<8>(sum);   // same pairing order as the xor butterfly it replaces (8, 4, 2, 1)
            sum += dpp_xor<4>(sum);
            sum += dpp_xor<2>(sum);
            sum += dpp_xor<1>(sum);
            *reinterpret_cast<f16x8*>(pl + r * QS_ATT_CHUNK + ks * 8) = p8;
            *reinterpret_cast<f16x8*>(pl2 + r * QS_ATT_CHUNK + ks * 8) = p8lo;
            if (ks == 0) {
                const float alpha = m_old == -__builtin_inff() ? 0.0f : aexp(m_old - m_new);   // e^0 = 1 exactly
                row_a[r] = alpha;
                row_m[r] = m_new;
                row_l[r] = __builtin_fmaf(row_l[r], alpha, sum);   // first chunk: 0 * 0 + sum
            }
        }
        if (last) QS_STAMP(7);
        __syncthreads();
        if (last) QS_STAMP(8);
        // ---- O = O * alpha + P V: wave w owns d columns 32w..32w+31 (two 16-wide tiles), k over the 128 keys in 4 steps
        {
#pragma unroll
            for (int reg = 0; reg < 4; reg++) {
                const float al = row_a[4 * g4 + reg];
                o0[reg] *= al;
                o1[reg] *= al;
            }
            const uint32_t vl_base = (uint32_t)(uintptr_t)vl;  // LDS byte address (low 32 bits of the shared pointer)
            const int qd = c16 >> 2, pq = c16 & 3;              // lane 4q+p of its 16-lane group
#pragma unroll
            for (int st = 0; st < 4; st++) {
                if (st * 32 >= nkeys) continue;   // uniform: P is all zero there (short splits: 64 keys = two of the four steps)
                const f16x8 pa = *reinterpret_cast<const f16x8*>(pl + c16 * QS_ATT_CHUNK + st * 32 + g4 * 8);
                const f16x8 pb = *reinterpret_cast<const f16x8*>(pl2 + c16 * QS_ATT_CHUNK + st * 32 + g4 * 8);
                // B fragment of lane (col c16, group g4) = V[key st*32 + 8*g4 + j][d0 + c16], j = 0..7
                const int krow = st * 32 + g4 * 8 + qd;
#pragma unroll
                for (int dt = 0; dt < 2; dt++) {
                    const int d0 = wave * 32 + dt * 16;
                    const uint32_t a0 = vl_base + (uint32_t)((krow * QS_ATT_VSTRIDE + d0 + 4 * pq) * 2);
                    const s16x8 bw = __builtin_shufflevector(lds_tr16(a0), lds_tr16(a0 + 4 * QS_ATT_VSTRIDE * 2), 0, 1, 2, 3, 4, 5, 6, 7);
                    if (dt == 0) {
                        o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pb, __builtin_bit_cast(f16x8, bw), o0, 0, 0, 0);
                        o0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, __builtin_bit_cast(f16x8, bw), o0, 0, 0, 0);
                    } else {
                        o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pb, __builtin_bit_cast(f16x8, bw), o1, 0, 0, 0);
                        o1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, __builtin_bit_cast(f16x8, bw), o1, 0, 0, 0);
                    }
                }
            }
        }
        if (!last) __syncthreads();   // sc / pl / vl / row_a are rewritten by the next chunk
    };

    int kb = k_begin;
    for (int it = 0; it < n_it - 1; it++) {
        if constexpr (PF) {
            fetch(nxt, sl_nxt);                                     // chunk it+1: in flight underneath chunk it
            lookup(sl_nxt, kb + 2 * QS_ATT_CHUNK);                  // table entries of chunk it+2 (clamped)
            process(cur, kb, false);
            cur = nxt;
        } else {
            lookup(sl_nxt, kb + QS_ATT_CHUNK);                      // the next chunk's table entries under this chunk
            process(cur, kb, false);
            fetch(cur, sl_nxt);
        }
        kb += QS_ATT_CHUNK;
    }
#ifdef QS_ATT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    QS_STAMP(2);   // the (last) chunk's K / V are here
#endif
    process(cur, kb, true);
    QS_STAMP(4);
    // ---- partial of this split -> workspace: ws_o [T, nq, n_splits, D], ws_ml [T, nq, n_splits, 2]
    // lane holds rows 4*g4 + reg, columns wave*32 + {0,16} + c16
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int r = 4 * g4 + reg;
        if (r < R) {
            const int rr = r0 + r;
            const int tok = qs + (rr >> group_log2), head = (kvh << group_log2) + (rr & gmask);
            float* dst = ws_o + (((size_t)tok * nq + head) * n_splits + split) * D + wave * 32 + c16;
            dst[0] = o0[reg];
            dst[16] = o1[reg];
        }
    }
    if (tid < R) {
        const int rr = r0 + tid;
        const int tok = qs + (rr >> group_log2), head = (kvh << group_log2) + (rr & gmask);
        const size_t o = (((size_t)tok * nq + head) * n_splits + split) * 2;
        ws_ml[o] = row_m[tid];
        ws_ml[o + 1] = row_l[tid];
    }
    QS_STAMP(5);
    // merge == 0: the consumer kernel (heads_hadamard_merge) combines the splits; the launch boundary is the hand-off
#ifdef QS_ATT_STAMPS
    if (!merge && tid == 0 && seq == 0 && kvh == 0 && rb == 0 && split == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        QS_STAMP(6);
        long long* sb = reinterpret_cast<long long*>(cnt + 2048);
        for (int i = 0; i < 10; i++) sb[i] = stamp[i];
    }
#endif
    if (!merge) return;
    // ---- hand-off: every storing wave drains, one lane releases and takes a ticket
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    int* my_cnt = cnt + (seq * nkv + kvh) * n_rb + rb;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const int ticket = __hip_atomic_fetch_add(my_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        *flag = ticket == n_splits - 1;
    }
    __syncthreads();
    QS_STAMP(6);
    if (!*flag) return;
    if (tid == 0) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(my_cnt, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // ready for the next call
    }
    __syncthreads();
    QS_STAMP(7);
    // ---- merge: out = sum_s w_s o_s / sum_s w_s l_s, w_s = e^(m_s - M), splits in order.
    // Thread i owns 4 output columns of one row; its o loads (up to 8 splits at a time) go out first, the
    // per-row weights are computed meanwhile by 16 threads and passed through LDS.
    const int NV = R * (D / 4);
    for (int i0 = 0; i0 < NV; i0 += 256) {
        const int i = i0 + tid;
        const bool act = i < NV;
        const int r = act ? i / (D / 4) : 0, e4 = act ? (i % (D / 4)) * 4 : 0;
        const int rr = r0 + r;
        const size_t th = (size_t)(qs + (rr >> group_log2)) * nq + (kvh << group_log2) + (rr & gmask);
        float4 o8[8];
#pragma unroll
        for (int s2 = 0; s2 < 8; s2++) {
            o8[s2] = float4{0.f, 0.f, 0.f, 0.f};
            if (act && s2 < n_splits) o8[s2] = *reinterpret_cast<const float4*>(ws_o + (th * n_splits + s2) * D + e4);
        }
        if (i0 == 0) {
            if (tid < R) {
                const int r2 = r0 + tid;
                const size_t t2 = (size_t)(qs + (r2 >> group_log2)) * nq + (kvh << group_log2) + (r2 & gmask);
                float M = -__builtin_inff();
                for (int s2 = 0; s2 < n_splits; s2++) M = fmaxf(M, ws_ml[(t2 * n_splits + s2) * 2]);
                float den = 0.0f;
                for (int s2 = 0; s2 < n_splits; s2++) {
                    const float m = ws_ml[(t2 * n_splits + s2) * 2];
                    const float w = m == -__builtin_inff() ? 0.0f : aexp(m - M);
                    wgt[tid * QS_ATT_MAXSPLIT + s2] = w;
                    den = __builtin_fmaf(w, ws_ml[(t2 * n_splits + s2) * 2 + 1], den);
                }
                row_l[tid] = den;
            }
            __syncthreads();
        }
        if (act) {
            float4 num = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s2 = 0; s2 < 8; s2++) {
                if (s2 < n_splits) {
                    const float w = wgt[r * QS_ATT_MAXSPLIT + s2];
                    num.x = __builtin_fmaf(w, o8[s2].x, num.x);
                    num.y = __builtin_fmaf(w, o8[s2].y, num.y);
                    num.z = __builtin_fmaf(w, o8[s2].z, num.z);
                    num.w = __builtin_fmaf(w, o8[s2].w, num.w);
                }
            }
            for (int s2 = 8; s2 < n_splits; s2++) {
                const float w = wgt[r * QS_ATT_MAXSPLIT + s2];
                const float4 o4 = *reinterpret_cast<const float4*>(ws_o + (th * n_splits + s2) * D + e4);
                num.x = __builtin_fmaf(w, o4.x, num.x);
                num.y = __builtin_fmaf(w, o4.y, num.y);
                num.z = __builtin_fmaf(w, o4.z, num.z);
                num.w = __builtin_fmaf(w, o4.w, num.w);
            }
            const float den = row_l[r];
            f16x4 o = {f2h(num.x / den), f2h(num.y / den), f2h(num.z / den), f2h(num.w / den)};
            *reinterpret_cast<f16x4*>(out + th * D + e4) = o;
        }
    }
#ifdef QS_ATT_STAMPS
    QS_STAMP(8);
    if (tid == 0 && seq == 0 && kvh == 0 && rb == 0) {
        long long* sb = reinterpret_cast<long long*>(cnt + 2048);
        for (int i = 0; i < 9; i++) sb[i] = stamp[i];
    }
#endif
}

// ---------------------------------------------------------------- decode, the keys of a workgroup split over its WAVES
// Same grid, same partials as paged_attention_kernel (16 rows per workgroup, context splits on blockIdx.z), but no
// workgroup-wide phase inside the key loop: wave w walks the 32-key slices w, w+4, ... of the split on its own -- K
// fragments straight from the cache into registers, V through a wave-private LDS tile for the transposing reads,
// S / softmax / P.V and a running (max, sum, output) per wave -- and the four waves' results are combined once at
// the end (wave order: deterministic).  The next slice's loads fly underneath the current one.  Where the 128-key
// chunk loop of paged_attention_kernel spends 2.7 us in barrier-separated phases per chunk, the waves here overlap
// each other: large batches and long contexts (more than one chunk per workgroup).  Partials only (out == NULL form).
#define QS_AW_KEYS 32
// NW waves per workgroup (4; 8 behind QSPEC_ATTN_NW): a wave has ONE slice (16 KB of K + V) in flight beside the one it
// works on.  A pure read of config 3's 67 MB with 32 KB per CU in flight takes 21.8 us, 13-14 us with twice that
// (scripts/micro/kvread.hip) -- but eight waves = two per SIMD are SLOWER here (21.6 against 19.1 us): in-kernel stamps
// (-DQS_ATT_STAMPS, scripts/bench_attn_cold.py) show a wave stuck 0.5-1.8 us per slice at the ISSUE of its refill loads
// (the CU's request window is full) with its softmax + P.V (1.35 us per slice) behind that.
// FAST (needs BT, block_size >= 16; a split starts on a 16-key boundary): a 32-key slice is two 16-key halves of ONE
// block each, so a slice needs two table entries, not ten -- wave-uniform, taken with v_readlane from the table row in
// registers -- and every K / V address is a uniform 64-bit base (SALU) plus a lane offset that never changes.  The
// general form spends ~250 of its ~700 loop instructions on lane permutes and 64-bit address arithmetic per slice and
// needs AGPR copies (291 VGPRs); at one wave per SIMD the instruction count IS the slice time.
template <bool BT, int NW, bool FAST>   // BT: block-table row in registers, lookups as lane permutes (see paged_attention_kernel)
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void paged_attention_waves_kernel(
    const f16* __restrict__ q, int64_t q_stride, const f16* __restrict__ key_cache, const f16* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ ctx_lens,
    const int32_t* __restrict__ q_start, int nq, int nkv, int bs_log2, int group_log2, float sm_scale, int n_splits,
    int n_rb, float* __restrict__ ws_o, float* __restrict__ ws_ml) {
    constexpr int D = 128, PSTR = 40;   // P row stride in halves (80 B)
    constexpr int WAVE_LDS = QS_AW_KEYS * QS_ATT_VSTRIDE * 2 + 2 * 16 * PSTR * 2;   // V tile + P hi/lo tile, bytes
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    const int seq = blockIdx.x, kvh = blockIdx.y / n_rb, rb = blockIdx.y % n_rb, split = blockIdx.z;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int gmask = (1 << group_log2) - 1, bmask = (1 << bs_log2) - 1;
    f16* vl = reinterpret_cast<f16*>(smem_raw + wave * WAVE_LDS);
    f16* pl = vl + QS_AW_KEYS * QS_ATT_VSTRIDE;
    int btv0 = 0, btv1 = 0;
    if constexpr (BT) {
        const int32_t* btr = block_tables + (size_t)blockIdx.x * max_blocks;
        btv0 = btr[min(lane, max_blocks - 1)];
        btv1 = btr[min(lane + 64, max_blocks - 1)];
    }
    int meta = 0;
    if (lane < 3) meta = lane < 2 ? q_start[seq + lane] : ctx_lens[seq];
    const int qs = __shfl(meta, 0, 64), qlen = __shfl(meta, 1, 64) - qs;
    const int ctx = __shfl(meta, 2, 64);
    const int r0 = rb * QS_ATT_MAXR;
    const int R = min(QS_ATT_MAXR, (qlen << group_log2) - r0);
    if (R <= 0) return;  // uniform for the whole workgroup
    const int kps = (((ctx + n_splits - 1) / n_splits) + 15) & ~15;
    const int k_begin = split * kps;
    const int k_end = min(ctx, k_begin + kps);
    const int n_sl = (max(k_end - k_begin, 0) + QS_AW_KEYS - 1) / QS_AW_KEYS;
    const int32_t* bt = block_tables + (size_t)seq * max_blocks;

    auto bt_get = [&](int bi) -> int {   // bi already clamped to the table
        if constexpr (BT) {
            const int a = __shfl(btv0, bi & 63, 64), b = __shfl(btv1, bi & 63, 64);
            return bi < 64 ? a : b;
        } else {
            return bt[bi];
        }
    };
    struct Slots {
        int64_t k[2], v[8];
    };
    struct KV {
        u32x4 kf[2][4], vr[8];
    };
    auto lookup = [&](Slots& sl, int sb) {   // clamped: no load behind a branch
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
            const int p = sb + t2 * 16 + c16;
            sl.k[t2] = ((int64_t)bt_get(min(p >> bs_log2, max_blocks - 1)) << bs_log2) + (p & bmask);
        }
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const int p = sb + i * 4 + g4;
            sl.v[i] = ((int64_t)bt_get(min(p >> bs_log2, max_blocks - 1)) << bs_log2) + (p & bmask);
        }
    };
    // FAST: uniform element offsets of the slice's two 16-key halves, ((slot * nkv) + kvh) * D
    static_assert(!FAST || BT, "the fast lookup reads the table row from registers");
    auto ublock = [&](int bi) -> int {   // bi wave-uniform
        const int i = __builtin_amdgcn_readfirstlane(min(bi, max_blocks - 1));
        const int a = __builtin_amdgcn_readlane(btv0, i & 63), b2 = __builtin_amdgcn_readlane(btv1, i & 63);
        return i < 64 ? a : b2;
    };
    auto lookup_u = [&](int64_t (&base)[2], int sbu) {
#pragma unroll
        for (int h = 0; h < 2; h++) {
            const int p = sbu + 16 * h;
            const int64_t slot = ((int64_t)ublock(p >> bs_log2) << bs_log2) + (p & bmask);
            base[h] = (slot * nkv + kvh) * D;
        }
    };
    const int k_lane = (c16 * nkv) * D + g4 * 8, v_lane = (g4 * nkv) * D + c16 * 8;   // elements; < 2^31 (nkv * 16 * 128)
    const int v_step = 4 * nkv * D;                                                   // four rows further
    auto fetch_k_u = [&](KV& kv, const int64_t (&base)[2]) {
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
            const f16* kp = key_cache + base[t2] + k_lane;
#pragma unroll
            for (int j = 0; j < 4; j++) kv.kf[t2][j] = *reinterpret_cast<const u32x4*>(kp + 32 * j);
        }
    };
    auto fetch_v_u = [&](KV& kv, const int64_t (&base)[2]) {
#pragma unroll
        for (int i = 0; i < 8; i++)
            kv.vr[i] = *reinterpret_cast<const u32x4*>(value_cache + base[i >> 2] + (int64_t)((i & 3) * v_step) + v_lane);
    };
    auto fetch = [&](KV& kv, const Slots& sl) {
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {   // lane (key c16 of tile t2, d slice 8 g4 + 32 j)
            const f16* kp = key_cache + (sl.k[t2] * nkv + kvh) * D + g4 * 8;
#pragma unroll
            for (int j = 0; j < 4; j++) kv.kf[t2][j] = *reinterpret_cast<const u32x4*>(kp + 32 * j);
        }
#pragma unroll
        for (int i = 0; i < 8; i++)   // 16 lanes cover one 256-byte V row: key 4 i + g4 of the slice
            kv.vr[i] = *reinterpret_cast<const u32x4*>(value_cache + (sl.v[i] * nkv + kvh) * D + c16 * 8);
    };
#ifdef QS_ATT_STAMPS
    long long awst[24];
    int awn = 0;
#define QS_AWST() do { if (awn < 24) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(awst[awn])::"memory"); awn++; } } while (0)
#define QS_AWST_NW() do { if (awn < 24) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(awst[awn])::"memory"); awn++; } } while (0)
#else
#define QS_AWST()
#define QS_AWST_NW()
#endif
    QS_AWST_NW();   // 0: metadata here
    int sb = (FAST ? __builtin_amdgcn_readfirstlane(k_begin) : k_begin) + wave * QS_AW_KEYS;
    Slots sl_cur, sl_nxt;
    int64_t ub[2];
    if constexpr (FAST) {
        lookup_u(ub, sb);
    } else {
        lookup(sl_cur, sb);
        lookup(sl_nxt, sb + NW * QS_AW_KEYS);
    }
    u32x4 qfrag[4];
    {
        const int r = r0 + (c16 < R ? c16 : 0);
        const int tok = qs + (r >> group_log2), head = (kvh << group_log2) + (r & gmask);
        const f16* qp = q + (size_t)tok * q_stride + (size_t)head * D + g4 * 8;
#pragma unroll
        for (int j = 0; j < 4; j++) qfrag[j] = *reinterpret_cast<const u32x4*>(qp + 32 * j);
    }
    KV cur;
    if constexpr (FAST) {
        fetch_k_u(cur, ub);
        fetch_v_u(cur, ub);
    } else {
        fetch(cur, sl_cur);
    }
    int pos[4];
    float row_m[4], row_l[4];
    f32x4 o[8];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        pos[reg] = ctx - qlen + ((r0 + 4 * g4 + reg) >> group_log2);
        row_m[reg] = -__builtin_inff();
        row_l[reg] = 0.0f;
    }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint32_t vl_base = (uint32_t)(uintptr_t)vl;
    const int qd = c16 >> 2, pq = c16 & 3;
    QS_AWST_NW();   // 1: first slice requested

    for (int s2 = wave; s2 < n_sl; s2 += NW) {
        QS_AWST();  // loop top: the slice's K and V have arrived (stamp build only: waits for everything)
        const int nkeys = max(0, min(k_end - sb, QS_AW_KEYS));
        // ---- S = Q K^T: lane holds rows 4 g4 + reg, key column t2 * 16 + c16.  Each K register is refilled with the
        // wave's NEXT slice right behind its last use (as the weight ring of gemm_stream.hip): the loads never stop.
        f32x4 sacc[2];
#pragma unroll
        for (int t2 = 0; t2 < 2; t2++) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 4; j++)
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, qfrag[j]),
                                                             __builtin_bit_cast(f16x8, cur.kf[t2][j]), acc, 0, 0, 0);
            const int kk = t2 * 16 + c16, p = sb + kk;
#pragma unroll
            for (int reg = 0; reg < 4; reg++)
                sacc[t2][reg] = (kk < nkeys && p <= pos[reg]) ? acc[reg] * sm_scale : -__builtin_inff();
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FAST) {
            lookup_u(ub, sb + NW * QS_AW_KEYS);   // the wave's next slice (clamped to the table: no load behind a branch)
            fetch_k_u(cur, ub);
        } else {
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                const f16* kp = key_cache + (sl_nxt.k[t2] * nkv + kvh) * D + g4 * 8;
#pragma unroll
                for (int j = 0; j < 4; j++) cur.kf[t2][j] = *reinterpret_cast<const u32x4*>(kp + 32 * j);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // V rows -> the wave's LDS tile (zeros past the end keep the MFMA clean), registers refilled the same way
        if (nkeys == QS_AW_KEYS) {   // uniform; LDS stores only (no global load sits behind this branch)
#pragma unroll
            for (int i = 0; i < 8; i++)
                *reinterpret_cast<u32x4*>(vl + (i * 4 + g4) * QS_ATT_VSTRIDE + c16 * 8) = cur.vr[i];
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int kk = i * 4 + g4;
                const u32x4 vz = kk < nkeys ? cur.vr[i] : u32x4{0, 0, 0, 0};
                *reinterpret_cast<u32x4*>(vl + kk * QS_ATT_VSTRIDE + c16 * 8) = vz;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FAST) {
            fetch_v_u(cur, ub);
        } else {
#pragma unroll
            for (int i = 0; i < 8; i++)
                cur.vr[i] = *reinterpret_cast<const u32x4*>(value_cache + (sl_nxt.v[i] * nkv + kvh) * D + c16 * 8);
        }
        __builtin_amdgcn_sched_barrier(0);
        QS_AWST_NW();   // QK + refills issued + V in LDS
        if constexpr (!FAST) lookup(sl_nxt, sb + 2 * NW * QS_AW_KEYS);           // table entries of the slice after the next (clamped)
        // ---- running softmax: the 16 lanes of a g4 group hold the 32 keys of rows 4 g4 + reg
        float alpha[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            float mx = fmaxf(sacc[0][reg], sacc[1][reg]);
            mx = fmaxf(mx, dpp_xor<8>(mx));
            mx = fmaxf(mx, dpp_xor<4>(mx));
            mx = fmaxf(mx, dpp_xor<2>(mx));
            mx = fmaxf(mx, dpp_xor<1>(mx));
            const float m_old = row_m[reg], m_new = fmaxf(m_old, mx);
            float sum = 0.0f;
#pragma unroll
            for (int t2 = 0; t2 < 2; t2++) {
                const float pv = m_new == -__builtin_inff() ? 0.0f : aexp(sacc[t2][reg] - m_new);
                const f16 ph = f2h(pv);
                pl[(4 * g4 + reg) * PSTR + t2 * 16 + c16] = ph;
                pl[(16 + 4 * g4 + reg) * PSTR + t2 * 16 + c16] = f2h(pv - h2f(ph));
                sum += pv;
            }
            sum += dpp_xor<8>(sum);
            sum += dpp_xor<4>(sum);
            sum += dpp_xor<2>(sum);
            sum += dpp_xor<1>(sum);
            alpha[reg] = m_old == -__builtin_inff() ? 0.0f : aexp(m_old - m_new);
            row_m[reg] = m_new;
            row_l[reg] = __builtin_fmaf(row_l[reg], alpha[reg], sum);
        }
#pragma unroll
        for (int dt = 0; dt < 8; dt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) o[dt][reg] *= alpha[reg];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own tiles: LDS is in order per wave
        // ---- O += P V (one 32-key step): A = P (row c16, keys 8 g4 ..), B = V through the transposing reads
        {
            const f16x8 pa = *reinterpret_cast<const f16x8*>(pl + c16 * PSTR + g4 * 8);
            const f16x8 pb = *reinterpret_cast<const f16x8*>(pl + (16 + c16) * PSTR + g4 * 8);
            const uint32_t va = vl_base + (uint32_t)(((g4 * 8 + qd) * QS_ATT_VSTRIDE + 4 * pq) * 2);
            s16x8 bw[8];   // d tile dt: +32 B; the fragment's second key quad: +4 V rows
#pragma unroll
            for (int dt = 0; dt < 8; dt++)
                bw[dt] = __builtin_shufflevector(lds_tr16(va + 32 * dt), lds_tr16(va + 32 * dt + 4 * QS_ATT_VSTRIDE * 2), 0, 1, 2,
                                                 3, 4, 5, 6, 7);
#pragma unroll
            for (int dt = 0; dt < 8; dt++) {
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pb, __builtin_bit_cast(f16x8, bw[dt]), o[dt], 0, 0, 0);
                o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, __builtin_bit_cast(f16x8, bw[dt]), o[dt], 0, 0, 0);
            }
        }
        sb += NW * QS_AW_KEYS;
        QS_AWST_NW();   // slice done (softmax + P.V)
    }
    QS_AWST_NW();
    // ---- combine the four waves (wave order), store the split's partial: ws_o [T, nq, n_splits, D], ws_ml [.., 2]
    __syncthreads();   // every wave is done with its private tiles: the same LDS now carries the exchange
    float* co = reinterpret_cast<float*>(smem_raw);          // [NW][16][128]
    float* cml = co + NW * 16 * D;                            // [NW][16][2]
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int r = 4 * g4 + reg;
#pragma unroll
        for (int dt = 0; dt < 8; dt++) co[(wave * 16 + r) * D + dt * 16 + c16] = o[dt][reg];
        if (c16 == 0) {
            cml[(wave * 16 + r) * 2] = row_m[reg];
            cml[(wave * 16 + r) * 2 + 1] = row_l[reg];
        }
    }
    __syncthreads();
    {
        const int r = tid >> 4, c8 = (tid & 15) * 8;
        if (r < R) {   // R <= 16: the first 256 threads
            float M = -__builtin_inff();
#pragma unroll
            for (int w = 0; w < NW; w++) M = fmaxf(M, cml[(w * 16 + r) * 2]);
            float l = 0.0f, acc8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < NW; w++) {
                const float m = cml[(w * 16 + r) * 2];
                const float wg = m == -__builtin_inff() ? 0.0f : aexp(m - M);
                l = __builtin_fmaf(wg, cml[(w * 16 + r) * 2 + 1], l);
                const float4 a = *reinterpret_cast<const float4*>(co + (w * 16 + r) * D + c8);
                const float4 b = *reinterpret_cast<const float4*>(co + (w * 16 + r) * D + c8 + 4);
                acc8[0] = __builtin_fmaf(wg, a.x, acc8[0]); acc8[1] = __builtin_fmaf(wg, a.y, acc8[1]);
                acc8[2] = __builtin_fmaf(wg, a.z, acc8[2]); acc8[3] = __builtin_fmaf(wg, a.w, acc8[3]);
                acc8[4] = __builtin_fmaf(wg, b.x, acc8[4]); acc8[5] = __builtin_fmaf(wg, b.y, acc8[5]);
                acc8[6] = __builtin_fmaf(wg, b.z, acc8[6]); acc8[7] = __builtin_fmaf(wg, b.w, acc8[7]);
            }
            const int rr = r0 + r;
            const size_t th = (size_t)(qs + (rr >> group_log2)) * nq + (kvh << group_log2) + (rr & gmask);
            float* dst = ws_o + (th * n_splits + split) * D + c8;
            *reinterpret_cast<float4*>(dst) = float4{acc8[0], acc8[1], acc8[2], acc8[3]};
            *reinterpret_cast<float4*>(dst + 4) = float4{acc8[4], acc8[5], acc8[6], acc8[7]};
            if ((tid & 15) == 0) {
                ws_ml[(th * n_splits + split) * 2] = M;
                ws_ml[(th * n_splits + split) * 2 + 1] = l;
            }
        }
    }
#ifdef QS_ATT_STAMPS
    QS_AWST_NW();
    if (tid == 0 && blockIdx.x == 1 && blockIdx.y == 1 && blockIdx.z == 0) {
        long long* sbp = reinterpret_cast<long long*>(ws_o) - (QS_ATT_CNT_SLOTS / 2) + 1024;   // int slot 2048 of the counters
        sbp[0] = awn;
        for (int i = 0; i < awn; i++) sbp[1 + i] = awst[i];
    }
#endif
}

// ---------------------------------------------------------------- prompt-sized queries (flash-attn varlen, head 128)
// The kernel above keeps 16 rows per workgroup: right for decode (1 .. k+1 query tokens per sequence), but a prompt pass
// would stream K/V once per 4 query tokens.  Here a workgroup holds 64 rows (16 query tokens x the GQA group of one kv
// head), 16 per wave, and walks the visible keys in 64-key chunks (FlashAttention-2 form):
//   * the chunk's K and V go through LDS once per workgroup (K rows XOR-swizzled in 16-byte pieces for the fragment
//     reads, V rows padded for the transposing reads); the next chunk's rows are in flight in registers meanwhile;
//   * a wave's S = Q K^T (16 x 64), its running (max, sum) and its 16 x 128 output live in registers; only P goes
//     through a wave-private LDS tile to change from the accumulator layout to the A-operand layout (as hi + lo fp16,
//     two MFMAs, so that P.V keeps fp32-class accuracy);
//   * no context split: one (o, m, l) per row, stored as the single-split partial heads_hadamard_merge expects and/or
//     normalised into `out`.
#define QS_FA_ROWS 64
#define QS_FA_KEYS 64
__global__ __launch_bounds__(256) void paged_attention_prefill_kernel(
    const f16* __restrict__ q, int64_t q_stride, const f16* __restrict__ key_cache, const f16* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ ctx_lens,
    const int32_t* __restrict__ q_start, int nq, int nkv, int bs_log2, int group_log2, float sm_scale, int n_rb,
    float* __restrict__ ws_o, float* __restrict__ ws_ml, f16* __restrict__ out) {
    constexpr int D = 128;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    unsigned char* kl = reinterpret_cast<unsigned char*>(smem_raw);                       // [64 keys][256 B], swizzled
    f16* vl = reinterpret_cast<f16*>(kl + QS_FA_KEYS * 256);                               // [64 keys][QS_ATT_VSTRIDE]
    f16* pl_all = vl + QS_FA_KEYS * QS_ATT_VSTRIDE;                                        // [4 waves][hi, lo][16][72]
    constexpr int PSTR = 72;   // halves per P row (144 B): the 16 rows of an A-fragment read fall on distinct banks
    // the last row blocks see the most keys (causal): dispatch them first so that the tail of the launch is short blocks
    const int seq = blockIdx.x, rb = n_rb - 1 - blockIdx.y / nkv, kvh = blockIdx.y % nkv;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int gmask = (1 << group_log2) - 1, bmask = (1 << bs_log2) - 1;
    int meta = 0;
    if (lane < 3) meta = lane < 2 ? q_start[seq + lane] : ctx_lens[seq];
    const int qs = __shfl(meta, 0, 64), qlen = __shfl(meta, 1, 64) - qs;
    const int ctx = __shfl(meta, 2, 64);
    const int total_rows = qlen << group_log2;
    if (rb * QS_FA_ROWS >= total_rows) return;   // uniform
    const int rbase = rb * QS_FA_ROWS + wave * 16;
    f16* pl = pl_all + wave * 2 * 16 * PSTR;
    const int32_t* bt = block_tables + (size_t)seq * max_blocks;

    // causal horizon of the workgroup: the last valid row's token sees keys 0 .. pos
    const int last_row = min(rb * QS_FA_ROWS + QS_FA_ROWS - 1, total_rows - 1);
    const int n_keys = min(ctx, ctx - qlen + (last_row >> group_log2) + 1);
    const int n_chunks = (n_keys + QS_FA_KEYS - 1) / QS_FA_KEYS;

    // Q fragments: lane (row c16, d slice 8 g4 + 32 j); rows past the end repeat the last valid row (never stored)
    u32x4 qfrag[4];
    {
        const int r = min(rbase + c16, total_rows - 1);
        const f16* qp = q + (size_t)(qs + (r >> group_log2)) * q_stride + (size_t)((kvh << group_log2) + (r & gmask)) * D + g4 * 8;
#pragma unroll
        for (int j = 0; j < 4; j++) qfrag[j] = *reinterpret_cast<const u32x4*>(qp + 32 * j);
    }
    // positions of this lane's accumulator rows (4 g4 + reg)
    int pos[4];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int rr = rbase + 4 * g4 + reg;
        pos[reg] = rr < total_rows ? ctx - qlen + (rr >> group_log2) : -1;
    }
    float row_m[4], row_l[4];
    f32x4 o[8];
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        row_m[reg] = -__builtin_inff();
        row_l[reg] = 0.0f;
    }
#pragma unroll
    for (int dt = 0; dt < 8; dt++) o[dt] = f32x4{0.f, 0.f, 0.f, 0.f};

    // staging: thread (key = tid / 4, quarter = tid % 4) moves 64 B of the key's K row and of its V row
    const int skey = tid >> 2, spart = tid & 3;
    u32x4 kreg[4], vreg[4];
    auto stage_load = [&](int kb) {   // unconditional, clamped (no load behind a branch)
        const int p = min(kb + skey, ctx - 1);
        const int64_t slot = ((int64_t)bt[min(p >> bs_log2, max_blocks - 1)] << bs_log2) + (p & bmask);
        const f16* kp = key_cache + (slot * nkv + kvh) * D + spart * 32;
        const f16* vp = value_cache + (slot * nkv + kvh) * D + spart * 32;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            kreg[i] = *reinterpret_cast<const u32x4*>(kp + 8 * i);
            vreg[i] = *reinterpret_cast<const u32x4*>(vp + 8 * i);
        }
    };
    stage_load(0);
    for (int c = 0; c < n_chunks; c++) {
        const int kb = c * QS_FA_KEYS;
        const int nk = min(n_keys - kb, QS_FA_KEYS);
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const int ci = spart * 4 + i;
            *reinterpret_cast<u32x4*>(kl + skey * 256 + ((ci ^ (skey & 15)) << 4)) = kreg[i];
            const u32x4 vz = skey < nk ? vreg[i] : u32x4{0, 0, 0, 0};   // zero rows keep the MFMA clean
            *reinterpret_cast<u32x4*>(vl + skey * QS_ATT_VSTRIDE + ci * 8) = vz;
        }
        __syncthreads();
        stage_load(kb + QS_FA_KEYS);   // next chunk underneath this one (clamped re-read behind the last chunk)

        // ---- S = Q K^T for the wave's 16 rows x 64 keys: lane holds rows 4 g4 + reg, key column t * 16 + c16
        f32x4 sacc[4];
#pragma unroll
        for (int t = 0; t < 4; t++) {
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const int row = t * 16 + c16;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const u32x4 kf = *reinterpret_cast<const u32x4*>(kl + row * 256 + (((g4 + 4 * j) ^ (row & 15)) << 4));
                acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, qfrag[j]),
                                                             __builtin_bit_cast(f16x8, kf), acc, 0, 0, 0);
            }
            const int p = kb + t * 16 + c16;
#pragma unroll
            for (int reg = 0; reg < 4; reg++) sacc[t][reg] = p <= pos[reg] ? acc[reg] * sm_scale : -__builtin_inff();
        }
        // ---- online softmax per row: the 16 lanes of a g4 group hold the 64 keys of rows 4 g4 + reg
        float alpha[4];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            float mx = fmaxf(fmaxf(sacc[0][reg], sacc[1][reg]), fmaxf(sacc[2][reg], sacc[3][reg]));
            mx = fmaxf(mx, dpp_xor<8>(mx));
            mx = fmaxf(mx, dpp_xor<4>(mx));
            mx = fmaxf(mx, dpp_xor<2>(mx));
            mx = fmaxf(mx, dpp_xor<1>(mx));
            const float m_old = row_m[reg], m_new = fmaxf(m_old, mx);
            float sum = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; t++) {
                const float pv = m_new == -__builtin_inff() ? 0.0f : aexp(sacc[t][reg] - m_new);
                const f16 ph = f2h(pv);
                pl[(4 * g4 + reg) * PSTR + t * 16 + c16] = ph;
                pl[(16 + 4 * g4 + reg) * PSTR + t * 16 + c16] = f2h(pv - h2f(ph));
                sum += pv;
            }
            sum += dpp_xor<8>(sum);
            sum += dpp_xor<4>(sum);
            sum += dpp_xor<2>(sum);
            sum += dpp_xor<1>(sum);
            alpha[reg] = m_old == -__builtin_inff() ? 0.0f : aexp(m_old - m_new);
            row_m[reg] = m_new;
            row_l[reg] = __builtin_fmaf(row_l[reg], alpha[reg], sum);
        }
#pragma unroll
        for (int dt = 0; dt < 8; dt++)
#pragma unroll
            for (int reg = 0; reg < 4; reg++) o[dt][reg] *= alpha[reg];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the wave's own P tile: LDS is in order per wave
        // ---- O += P V: A = P (row c16, keys st * 32 + 8 g4 ..), B = V[keys][d0 + c16] through the transposing read
        {
            const uint32_t vl_base = (uint32_t)(uintptr_t)vl;
            const int qd = c16 >> 2, pq = c16 & 3;
#pragma unroll
            for (int st = 0; st < 2; st++) {
                const f16x8 pa = *reinterpret_cast<const f16x8*>(pl + c16 * PSTR + st * 32 + g4 * 8);
                const f16x8 pb = *reinterpret_cast<const f16x8*>(pl + (16 + c16) * PSTR + st * 32 + g4 * 8);
                const int krow = st * 32 + g4 * 8 + qd;
                u32x2 lo[8], hi[8];
                lds_read_tr_b16_x16(vl_base + (uint32_t)((krow * QS_ATT_VSTRIDE + 4 * pq) * 2), lo, hi);
#pragma unroll
                for (int dt = 0; dt < 8; dt++) {
                    const u32x4 bw = {lo[dt][0], lo[dt][1], hi[dt][0], hi[dt][1]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pb, __builtin_bit_cast(f16x8, bw), o[dt], 0, 0, 0);
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(pa, __builtin_bit_cast(f16x8, bw), o[dt], 0, 0, 0);
                }
            }
        }
        __syncthreads();   // kl / vl are rewritten by the next chunk
    }
    // ---- results: lane holds rows 4 g4 + reg, columns dt * 16 + c16
#pragma unroll
    for (int reg = 0; reg < 4; reg++) {
        const int rr = rbase + 4 * g4 + reg;
        if (rr < total_rows) {
            const size_t th = (size_t)(qs + (rr >> group_log2)) * nq + (kvh << group_log2) + (rr & gmask);
            if (ws_o) {
#pragma unroll
                for (int dt = 0; dt < 8; dt++) ws_o[th * D + dt * 16 + c16] = o[dt][reg];
                if (c16 == 0) {
                    ws_ml[th * 2] = row_m[reg];
                    ws_ml[th * 2 + 1] = row_l[reg];
                }
            }
            if (out) {
#pragma unroll
                for (int dt = 0; dt < 8; dt++) out[th * D + dt * 16 + c16] = f2h(o[dt][reg] / row_l[reg]);
            }
        }
    }
}

// ---------------------------------------------------------------- generic head size (plumbing path)
// Any head_size <= 256 (TinyLlama: 64).  One 128-thread workgroup per (query token, head): scores of all visible keys
// into LDS (fp32 dot products), one softmax over them, then thread j accumulates output dimension j over the keys in
// order (fp32, deterministic).  No context split, no matrix cores: correctness path for configurations the MFMA
// kernel (head_size 128) does not cover, not a tuned one.
__global__ __launch_bounds__(128) void paged_attention_generic_kernel(
    const f16* __restrict__ q, int64_t q_stride, const f16* __restrict__ key_cache, const f16* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ ctx_lens,
    const int32_t* __restrict__ q_start, int n_seqs, int nq, int nkv, int d, int block_size, float sm_scale,
    f16* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    float* sc = reinterpret_cast<float*>(smem_raw);   // [ctx]
    __shared__ float red[4];
    const int t = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    int seq = 0;
    while (seq + 1 < n_seqs && q_start[seq + 1] <= t) seq++;
    const int qs = q_start[seq], qlen = q_start[seq + 1] - qs, ctx = ctx_lens[seq];
    if (t - qs >= qlen) return;
    const int pos = ctx - qlen + (t - qs);            // absolute position of this query token
    const int nvis = pos + 1;                          // causal: keys 0..pos
    const int kvh = h / (nq / nkv);
    const f16* qp = q + (size_t)t * q_stride + (size_t)h * d;
    const int32_t* bt = block_tables + (size_t)seq * max_blocks;
    float mx = -__builtin_inff();
    for (int k = tid; k < nvis; k += 128) {
        const int64_t slot = (int64_t)bt[k / block_size] * block_size + k % block_size;
        const f16* kp = key_cache + (slot * nkv + kvh) * d;
        float acc = 0.0f;
        for (int e = 0; e < d; e++) acc = __builtin_fmaf(h2f(qp[e]), h2f(kp[e]), acc);
        acc *= sm_scale;
        sc[k] = acc;
        mx = fmaxf(mx, acc);
    }
    mx = wave_max_f(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(red[0], red[1]);
    __syncthreads();
    float sum = 0.0f;
    for (int k = tid; k < nvis; k += 128) {
        const float p = qexpf(sc[k] - mx);
        sc[k] = p;
        sum += p;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += shfl_xor_f(sum, m);
    if ((tid & 63) == 0) red[2 + (tid >> 6)] = sum;
    __syncthreads();
    const float den = red[2] + red[3];
    for (int e = tid; e < d; e += 128) {
        float acc = 0.0f;
        for (int k = 0; k < nvis; k++) {
            const int64_t slot = (int64_t)bt[k / block_size] * block_size + k % block_size;
            acc = __builtin_fmaf(sc[k], h2f(value_cache[(slot * nkv + kvh) * d + e]), acc);
        }
        out[((size_t)t * nq + h) * d + e] = f2h(acc / den);
    }
}

// The same arithmetic with vector loads (head_size a multiple of 8; round 3: TinyLlama's bs = 1 cycle spent 47 % of its time
// in the kernel above, 51 us per launch: its P.V phase ran head_size threads through one DEPENDENT 2-byte load per key).
// 256 threads: a thread's score is the same e-ascending fma chain over 16-byte pieces of the key row (all pieces of a row
// in flight together); P.V stages chunks of 256 / (head_size / 8) V rows through LDS with one 16-byte piece per thread
// (next chunk in flight in registers while the current one is accumulated) and thread e then adds the chunk's keys in
// ascending order: the output sums are those of the kernel above bit for bit; the softmax denominator is summed in a
// different (fixed) grouping.
__global__ __launch_bounds__(256) void paged_attention_generic_vec_kernel(
    const f16* __restrict__ q, int64_t q_stride, const f16* __restrict__ key_cache, const f16* __restrict__ value_cache,
    const int32_t* __restrict__ block_tables, int max_blocks, const int32_t* __restrict__ ctx_lens,
    const int32_t* __restrict__ q_start, int n_seqs, int nq, int nkv, int d, int block_size, float sm_scale,
    f16* __restrict__ out, float* __restrict__ ws_o, float* __restrict__ ws_ml) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    __shared__ float red[8];
    __shared__ __attribute__((aligned(16))) f16 qrow[256];
    const int t = blockIdx.x, h = blockIdx.y, tid = threadIdx.x;
    int seq = 0;
    while (seq + 1 < n_seqs && q_start[seq + 1] <= t) seq++;
    const int qs = q_start[seq], qlen = q_start[seq + 1] - qs, ctx = ctx_lens[seq];
    if (t - qs >= qlen) return;   // (uniform)
    const int pos = ctx - qlen + (t - qs);            // absolute position of this query token
    const int nall = pos + 1;                          // causal: keys 0..pos
    // context splits (gridDim.z = S > 1, or partials wanted): split sp takes keys [kbase, kbase + nvis) -- whole 16-key groups,
    // the same length for every split of a token; an empty split leaves (m, l, o) = (-inf, 0, 0)
    const int S = gridDim.z, sp = blockIdx.z;
    const int per = S > 1 ? ((((nall + S - 1) / S) + 15) & ~15) : nall;
    const int kbase = min(nall, sp * per);
    const int nvis = min(nall - kbase, per);           // keys of this workgroup (0: an empty split)
    const int kvh = h / (nq / nkv);
    const int32_t* bt = block_tables + (size_t)seq * max_blocks;
    const int pieces = d >> 3;                         // 16-byte pieces per row
    const int RG = 256 / pieces;                       // row groups of the P.V pass (threads beyond RG * pieces idle there)
    float* sc = reinterpret_cast<float*>(smem_raw);   // [nvis rounded up to 4]
    float* pvr = reinterpret_cast<float*>(smem_raw + (((size_t)nvis * 4 + 15) & ~(size_t)15));   // [RG][d] partial output rows
    if (tid < pieces)
        *reinterpret_cast<u32x4*>(qrow + tid * 8) = *reinterpret_cast<const u32x4*>(q + (size_t)t * q_stride + (size_t)h * d + tid * 8);
    // the V rows do not depend on the scores: this thread's first PRE of them are requested here, in front of the K rows
    constexpr int PRE = 8;
    const int rg = tid / pieces, vp = tid - rg * pieces;
    const bool pv_on = rg < RG;
    auto vaddr = [&](int k) -> const f16* {
        const int kc = min(kbase + min(k, nvis - 1), nall - 1);   // clamped: no branch around a load (rows past the end are not used)
        const int64_t slot = (int64_t)bt[kc / block_size] * block_size + kc % block_size;
        return value_cache + (slot * nkv + kvh) * d + vp * 8;
    };
    u32x4 vpre[PRE];
#pragma unroll
    for (int j = 0; j < PRE; j++) vpre[j] = *reinterpret_cast<const u32x4*>(vaddr(rg + RG * j));
    __syncthreads();
    // scores: NK keys per thread and trip -- their table entries first, then every K piece in flight, then the dot products
    // (one key per trip was two dependent round trips per key: 512 keys = four of them per thread)
    constexpr int NK = 4;
    float mx = -__builtin_inff();
    for (int kb = tid; kb < nvis; kb += 256 * NK) {
        const f16* kp[NK];
#pragma unroll
        for (int i = 0; i < NK; i++) {
            const int kc = min(kbase + min(kb + 256 * i, nvis - 1), nall - 1);   // clamped: rows past the end are loaded, not used
            const int64_t slot = (int64_t)bt[kc / block_size] * block_size + kc % block_size;
            kp[i] = key_cache + (slot * nkv + kvh) * d;
        }
        float acc[NK];
#pragma unroll
        for (int i = 0; i < NK; i++) acc[i] = 0.0f;
        for (int p0 = 0; p0 < pieces; p0 += 8) {      // up to 8 pieces of each row in flight
            u32x4 kr[NK][8];
#pragma unroll
            for (int i = 0; i < NK; i++)
#pragma unroll
                for (int j = 0; j < 8; j++) kr[i][j] = *reinterpret_cast<const u32x4*>(kp[i] + min(p0 + j, pieces - 1) * 8);
#pragma unroll
            for (int j = 0; j < 8; j++) {
                if (p0 + j < pieces) {
                    const f16x8 q8 = *reinterpret_cast<const f16x8*>(qrow + (p0 + j) * 8);
#pragma unroll
                    for (int i = 0; i < NK; i++) {
                        const f16x8 k8 = __builtin_bit_cast(f16x8, kr[i][j]);
#pragma unroll
                        for (int e = 0; e < 8; e++) acc[i] = __builtin_fmaf(h2f(q8[e]), h2f(k8[e]), acc[i]);
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NK; i++) {
            const int k = kb + 256 * i;
            if (k < nvis) {
                const float a = acc[i] * sm_scale;
                sc[k] = a;
                mx = fmaxf(mx, a);
            }
        }
    }
    mx = wave_max_f(mx);
    if ((tid & 63) == 0) red[tid >> 6] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    float sum = 0.0f;
    for (int k = tid; k < nvis; k += 256) {
        const float pv = qexpf(sc[k] - mx);
        sc[k] = pv;
        sum += pv;
    }
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) sum += shfl_xor_f(sum, m);
    if ((tid & 63) == 0) red[4 + (tid >> 6)] = sum;
    __syncthreads();
    const float den = (red[4] + red[5]) + (red[6] + red[7]);
    // ---- P.V: thread (row group rg, 16-byte piece vp) sums p_k * V[k][8 vp .. 8 vp + 8) over the keys rg, rg + RG, ...
    // (the first eight of its rows were requested in front of the scores), then the RG partial rows are added in order.
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; e++) acc[e] = 0.0f;
    auto fma8 = [&](float pk, u32x4 raw) {
        const f16x8 v8 = __builtin_bit_cast(f16x8, raw);
#pragma unroll
        for (int e = 0; e < 8; e++) acc[e] = __builtin_fmaf(pk, h2f(v8[e]), acc[e]);
    };
    const int pv_rows = min(RG, nvis);   // row groups that saw a key (the others wrote nothing)
    if (pv_on && rg < nvis) {
#pragma unroll
        for (int j = 0; j < PRE; j++) {
            const int k = rg + RG * j;
            if (k < nvis) fma8(sc[k], vpre[j]);
        }
        for (int k0 = rg + RG * PRE; k0 < nvis; k0 += RG * PRE) {   // long contexts: eight more rows in flight per trip
            u32x4 vb[PRE];
#pragma unroll
            for (int j = 0; j < PRE; j++) vb[j] = *reinterpret_cast<const u32x4*>(vaddr(k0 + RG * j));
#pragma unroll
            for (int j = 0; j < PRE; j++) {
                const int k = k0 + RG * j;
                if (k < nvis) fma8(sc[k], vb[j]);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; e++) pvr[(size_t)rg * d + vp * 8 + e] = acc[e];
    }
    __syncthreads();
    if (tid < d) {
        float o = pv_rows > 0 ? pvr[tid] : 0.0f;
        for (int r2 = 1; r2 < pv_rows; r2++) o += pvr[(size_t)r2 * d + tid];
        if (ws_o) {   // partials of split sp: unnormalised output, (row maximum, sum of exponentials)
            const size_t ths = ((size_t)t * nq + h) * S + sp;
            ws_o[ths * d + tid] = o;
            if (tid == 0) *reinterpret_cast<float2*>(ws_ml + ths * 2) = float2{mx, den};
        } else {
            out[((size_t)t * nq + h) * d + tid] = f2h(o / den);
        }
    }
}

// float offsets of the partials inside the workspace (shared with hadamard.hip:heads_hadamard_merge)
// Split merge for the generic head sizes when the caller wants fp16 rows (the reference-order path; the engine merges in
// the head-transform launch, hadamard.hip: heads_hadamard_merge_cols_kernel -- same expression, same bits):
// out = h( sum_s w_s o_s / sum_s w_s l_s ), w_s = e^(m_s - M), splits in order.
__global__ __launch_bounds__(256) void paged_attention_generic_merge_kernel(const float* __restrict__ ws_o,
                                                                            const float* __restrict__ ws_ml, int S, int d,
                                                                            f16* __restrict__ out, const int32_t* __restrict__ q_start,
                                                                            int n_seqs, int nq) {
    const size_t th = blockIdx.x;   // token * nq + head
    if ((int)(th / nq) >= q_start[n_seqs]) return;   // the grid covers n_seqs * max_q_len tokens, `out` the real ones
    const float* mlb = ws_ml + th * S * 2;
    float M = -__builtin_inff();
    for (int s2 = 0; s2 < S; s2++) M = fmaxf(M, mlb[s2 * 2]);
    for (int c = threadIdx.x; c < d; c += 256) {
        float num = 0.0f, den = 0.0f;
        for (int s2 = 0; s2 < S; s2++) {
            const float m = mlb[s2 * 2];
            const float w = m == -__builtin_inff() ? 0.0f : aexp(m - M);
            den = __builtin_fmaf(w, mlb[s2 * 2 + 1], den);
            num = __builtin_fmaf(w, ws_o[(th * S + s2) * d + c], num);
        }
        out[th * d + c] = f2h(num / den);
    }
}
// context splits the generic-head-size kernel uses for a caller's n_splits (shared with the merge in hadamard.hip)
int paged_attention_generic_splits(int n_splits) { return n_splits < 1 ? 1 : (n_splits > 4 ? 4 : n_splits); }   // (TinyLlama bs = 1, 512 / 1900 keys: 1: 4.61 / 6.27 ms per cycle, 2: 4.30 / 5.08, 4: 4.25 / 4.77, 8: 4.46 / 4.78)

size_t paged_attention_ws_o_offset() { return QS_ATT_CNT_SLOTS; }
size_t paged_attention_ws_ml_offset(int Tmax, int nq, int d, int n_splits) {
    return QS_ATT_CNT_SLOTS + (size_t)Tmax * nq * n_splits * d;
}
size_t paged_attention_ws_bytes(int T, int nq, int d, int n_splits) {
    return QS_ATT_CNT_SLOTS * sizeof(int) + (size_t)T * nq * n_splits * (d + 2) * sizeof(float);
}

static int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) l++;
    return l;
}

int paged_attention(const f16* q, int64_t q_stride, const f16* key_cache, const f16* value_cache,
                    const int32_t* block_tables, int max_blocks, const int32_t* ctx_lens, const int32_t* q_start,
                    int n_seqs, int max_q_len, int nq, int nkv, int d, int block_size, float sm_scale, int n_splits,
                    float* ws, f16* out, hipStream_t st) {
    // out == nullptr: leave the per-split partials (o, m, l) in the workspace for heads_hadamard_merge
    if (n_seqs == 0) return 0;
    if (nq % nkv) return -1;
    if (d != 128) {   // other head sizes (TinyLlama's 64): no matrix cores; context splits + a merge (here or in the head transform)
        if (d > 256 || d % 2) return -1;
        const int max_ctx_bytes = 64 * 1024;   // scores of one row in LDS: contexts up to 16 K keys
        if (d % 8 == 0 && d >= 8) {             // vector loads
            if (n_splits < 1 || n_splits > QS_ATT_MAXSPLIT) return -3;
            const int S = paged_attention_generic_splits(n_splits);
            if (!out && !ws) return -1;
            static bool attr_set = false;
            if (!attr_set) {
                if (hipFuncSetAttribute(reinterpret_cast<const void*>(paged_attention_generic_vec_kernel),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, max_ctx_bytes + 32768 + 16) != hipSuccess)
                    return -8;
                attr_set = true;
            }
            const size_t Tm = (size_t)n_seqs * max_q_len;
            const bool partials = !out || S > 1;
            float* ws_o = partials ? ws + paged_attention_ws_o_offset() : nullptr;
            float* ws_ml = partials ? ws + paged_attention_ws_ml_offset((int)Tm, nq, d, S) : nullptr;
            // LDS: the scores of one split (the block table bounds a context: max_blocks * block_size keys) + the [RG][d] fp32
            // partial rows of the P.V pass (<= 8 KiB) -- sized to the launch, so that several workgroups share a CU (with the
            // 96 KB of a 16 K-key context every launch ran one workgroup per CU: the verify pass's 512 in two rounds)
            const size_t per = (((size_t)max_blocks * block_size + S - 1) / S + 15) & ~(size_t)15;
            if (per * 4 > (size_t)max_ctx_bytes) return -2;
            const size_t lds = ((per * 4 + 15) & ~(size_t)15) + 8192 + 16;
            hipLaunchKernelGGL(paged_attention_generic_vec_kernel, dim3(n_seqs * max_q_len, nq, S), dim3(256), lds,
                               st, q, q_stride, key_cache, value_cache, block_tables, max_blocks, ctx_lens, q_start, n_seqs, nq,
                               nkv, d, block_size, sm_scale, out, ws_o, ws_ml);
            if (out && partials)
                hipLaunchKernelGGL(paged_attention_generic_merge_kernel, dim3((unsigned)(Tm * nq)), dim3(256), 0, st, ws_o, ws_ml, S, d,
                                   out, q_start, n_seqs, nq);
            return 0;
        }
        if (!out) return -1;
        hipLaunchKernelGGL(paged_attention_generic_kernel, dim3(n_seqs * max_q_len, nq), dim3(128), max_ctx_bytes, st, q,
                           q_stride, key_cache, value_cache, block_tables, max_blocks, ctx_lens, q_start, n_seqs, nq, nkv,
                           d, block_size, sm_scale, out);
        return 0;
    }
    if (n_splits < 1 || n_splits > QS_ATT_MAXSPLIT) return -3;
    const int bs_log2 = ilog2_exact(block_size), group_log2 = ilog2_exact(nq / nkv);
    if (bs_log2 < 0 || group_log2 < 0) return -5;  // block size and GQA group must be powers of two
    static const int fa_env = QS_DEV_KNOB("QSPEC_ATTN_PREFILL", 1);
    if (fa_env && n_splits == 1 && (max_q_len << group_log2) >= 2 * QS_FA_ROWS) {   // prompt-sized queries
        const int n_rb64 = ((max_q_len << group_log2) + QS_FA_ROWS - 1) / QS_FA_ROWS;
        const size_t Tm = (size_t)n_seqs * max_q_len;
        float* fo = ws + QS_ATT_CNT_SLOTS;
        float* fml = fo + Tm * nq * d;
        const size_t flds = (size_t)QS_FA_KEYS * 256 + (size_t)QS_FA_KEYS * QS_ATT_VSTRIDE * 2 + (size_t)4 * 2 * 16 * 72 * 2;
        hipLaunchKernelGGL(paged_attention_prefill_kernel, dim3(n_seqs, nkv * n_rb64), dim3(256), flds, st, q, q_stride,
                           key_cache, value_cache, block_tables, max_blocks, ctx_lens, q_start, nq, nkv, bs_log2,
                           group_log2, sm_scale, n_rb64, fo, fml, out);
        return 0;
    }
    const int n_rb = ((max_q_len << group_log2) + QS_ATT_MAXR - 1) / QS_ATT_MAXR;
    if (out && (size_t)n_seqs * nkv * n_rb > QS_ATT_CNT_SLOTS) return -4;   // ticket counters: in-kernel merge only
    // workspace: [ticket counters | o partials | (m,l) partials]; sized by the host for n_seqs*max_q_len tokens.
    // It must be zero-filled once before its first use; every call leaves the counters zero again.
    const size_t Tmax = (size_t)n_seqs * max_q_len;
    int* cnt = reinterpret_cast<int*>(ws);
    float* ws_o = ws + QS_ATT_CNT_SLOTS;
    float* ws_ml = ws_o + Tmax * nq * n_splits * d;
    const size_t lds = QS_ATT_MAXR * QS_ATT_CHUNK * 4 + 2 * QS_ATT_MAXR * QS_ATT_CHUNK * 2 +
                       QS_ATT_CHUNK * QS_ATT_VSTRIDE * 2 + (2 * QS_ATT_MAXR + QS_ATT_MAXR * QS_ATT_MAXSPLIT + 4) * 4;
    // Partials only, and a split may be longer than one 128-key chunk (large batch: one split; long contexts): the keys
    // go over the waves of a workgroup instead of through barrier-separated chunk phases (bs=32: 21.1 -> 20.3 us, 4 K
    // context: 20.8 -> 19.1 us per launch; equal at 64 keys per split).  QSPEC_ATTN_WAVES=0/1 forces either kernel.
    static const int aw_env = QS_DEV_KNOB("QSPEC_ATTN_WAVES", -1);
    const bool long_splits = ((int64_t)max_blocks << bs_log2) > (int64_t)QS_ATT_CHUNK * n_splits;
    if (!out && (aw_env >= 0 ? aw_env != 0 : long_splits)) {
        // QSPEC_ATTN_NW=8 (dev knob): eight waves, two per SIMD -- measured SLOWER (config 3: 21.6 against 19.1 us back to
        // back): the kernel is not short of bytes in flight
        static const int nw_env = QS_DEV_KNOB("QSPEC_ATTN_NW", 0);
        const int nw = nw_env == 8 ? 8 : 4;
        const size_t wave_lds = QS_AW_KEYS * QS_ATT_VSTRIDE * 2 + 2 * 16 * 40 * 2;
        const size_t wlds = std::max((size_t)nw * wave_lds, (size_t)nw * 16 * (128 + 2) * 4);
#define QS_AW_LAUNCH(BTV, NWV, FV)                                                                                           \
    do {                                                                                                                   \
        static bool attr_set = false;                                                                                      \
        if (!attr_set) {                                                                                                   \
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(paged_attention_waves_kernel<BTV, NWV, FV>),                 \
                                    hipFuncAttributeMaxDynamicSharedMemorySize, (int)wlds) != hipSuccess)                  \
                return -8;                                                                                                 \
            attr_set = true;                                                                                               \
        }                                                                                                                  \
        hipLaunchKernelGGL((paged_attention_waves_kernel<BTV, NWV, FV>), dim3(n_seqs, nkv * n_rb, n_splits), dim3(64 * NWV),   \
                           wlds, st, q, q_stride, key_cache, value_cache, block_tables, max_blocks, ctx_lens, q_start, nq, \
                           nkv, bs_log2, group_log2, sm_scale, n_splits, n_rb, ws_o, ws_ml);                               \
    } while (0)
        static const int fast_env = QS_DEV_KNOB("QSPEC_ATTN_FAST", 1);
        if (max_blocks <= 128 && bs_log2 >= 4 && fast_env) {   // two uniform table entries per 32-key slice
#ifdef QS_EXPERIMENTAL
            if (nw == 8) QS_AW_LAUNCH(true, 8, true);
            else
#endif
            QS_AW_LAUNCH(true, 4, true);
        } else if (max_blocks <= 128) {
#ifdef QS_EXPERIMENTAL
            if (nw == 8) QS_AW_LAUNCH(true, 8, false);
            else
#endif
            QS_AW_LAUNCH(true, 4, false);
        } else {
#ifdef QS_EXPERIMENTAL
            if (nw == 8) QS_AW_LAUNCH(false, 8, false);
            else
#endif
            QS_AW_LAUNCH(false, 4, false);
        }
#undef QS_AW_LAUNCH
        return 0;
    }
    // more workgroups than CUs: occupancy (two per CU) instead of the in-workgroup prefetch
    static const int pf_env = QS_DEV_KNOB("QSPEC_ATTN_PF", -1);
    const bool pf = pf_env >= 0 ? pf_env != 0 : (size_t)n_seqs * nkv * n_rb * n_splits <= 256;
    const bool btreg = max_blocks <= 128;
#define QS_ATT_LAUNCH(PFV, BTV)                                                                                        \
    hipLaunchKernelGGL((paged_attention_kernel<PFV, BTV>), dim3(n_seqs, nkv * n_rb, n_splits), dim3(256), lds, st, q,     \
                       q_stride, key_cache, value_cache, block_tables, max_blocks, ctx_lens, q_start, nq, nkv, bs_log2, \
                       group_log2, sm_scale, n_splits, n_rb, cnt, ws_o, ws_ml, out, out != nullptr ? 1 : 0)
    if (pf && btreg) QS_ATT_LAUNCH(true, true);
    else if (pf) QS_ATT_LAUNCH(true, false);
    else if (btreg) QS_ATT_LAUNCH(false, true);
    else QS_ATT_LAUNCH(false, false);
#undef QS_ATT_LAUNCH
    return 0;
}

}  // namespace qspec
