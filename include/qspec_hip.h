/*
 * qspec_hip.h -- C ABI of libqspec_hip.so, the MI355X (gfx950) replacement for
 * the native operators on the QSpec draft/verify hot path.
 *
 * Conventions (same as the reference's operator boundary):
 *   - every pointer is a DEVICE pointer owned by the caller; outputs are
 *     pre-allocated and passed in; tensors are contiguous row-major;
 *   - fp16 tensors are `qspec_half*` (uint16_t bit patterns);
 *   - `stream` is a hipStream_t (NULL = default stream); every call only
 *     enqueues, never synchronises, never allocates: calls may be captured
 *     into a hipGraph;
 *   - return 0 on success; non-zero = error, text from qspec_last_error()
 *     (thread-local).  The reference raises TORCH_CHECK exceptions at the same
 *     places (shape / dtype / contiguity checks); the Python host re-raises.
 *
 * Each entry point names the reference interface it replaces (paths relative
 * to the reference checkout).
 */
#ifndef QSPEC_HIP_H
#define QSPEC_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef uint16_t qspec_half;

int qspec_abi_version(void);
const char* qspec_last_error(void);

/* ---- normalisation + activation quantisation ------------------------------------------------ */

/* qserve_backend.layernorm_ops.rms_norm_general_fuse_sum_i4(out_q, x, input_sum, scaling, eps, True)
 *   third-party/kernels/csrc/layernorm.cpp:80-83, layernorm_kernels.cu:569-716,890-923
 *   (called from vllm/model_executor/layers/quarot_nn/normalization.py:56-63)
 * LayerNorm without gamma + per-token int4 quant.  hidden % 1024 == 0, hidden <= 8192.
 * input_sum may be NULL (the reference's wrapper drops it). */
int qspec_rms_norm_general_fuse_sum_i4(int8_t* out_q, const qspec_half* x, qspec_half* input_sum, qspec_half* scaling,
                                       float eps, int tokens, int hidden, void* stream);

/* layernorm_ops.rms_norm_general_fuse_sum_fp16(out, x, eps)   layernorm.cpp:85-87, layernorm_kernels.cu:927-957
 *   (normalization.py:76-80) */
int qspec_rms_norm_general_fuse_sum_fp16(qspec_half* out, const qspec_half* x, float eps, int tokens, int hidden,
                                         void* stream);

/* Fused forms used by the decoder loop: hidden_out = h(x + delta) (the fp16 residual add of
 * vllm/model_executor/models/quarot_llama.py:380,390), then the same norm on hidden_out.
 * delta == NULL -> plain norm of x (hidden_out ignored). */
int qspec_add_rms_norm_i4(int8_t* out_q, qspec_half* scaling, qspec_half* hidden_out, const qspec_half* x,
                          const qspec_half* delta, float eps, int tokens, int hidden, void* stream);
int qspec_add_rms_norm_fp16(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const qspec_half* delta,
                            float eps, int tokens, int hidden, void* stream);

/* quarot._CUDA.fuse_sym_quant(x, scale, q, clip)   third-party/QuaRot/quarot/kernels/bindings.cpp:128-147,
 *   quant.cu:102-186 (quarot/__init__.py:119-144, quarot_nn/quantization.py:13-19).  k even. */
int qspec_fuse_sym_quant(const qspec_half* x, qspec_half* scale, int8_t* q, float clip_ratio, int tokens, int k,
                         void* stream);

/* ---- online Hadamard ------------------------------------------------------------------------- */

/* fast_hadamard_transform_cuda.faster_fast_hadamard_transform(x, scale, out) / fast_hadamard_transform(x, scale)
 *   third-party/fast-hadamard-transform/csrc/fast_hadamard_transform.cpp:69-154, ..._cuda.cu:124-198.
 *   rows of length n (power of two, 1..32768; n = 1 only scales and rounds); out = h(WHT(x) * scale). */
int qspec_fast_hadamard_transform(const qspec_half* x, float scale, qspec_half* out, int64_t rows, int n,
                                  void* stream);

/* `hadK @ y.view(-1, K, m)` of quarot/functional/hadamard.py:104-108,120 (cuBLAS batched GEMM in the reference).
 *   y, out [tokens, K, m]; hadK [K, K] fp16; fp32 accumulate in k order. */
int qspec_hadamard_mix(const qspec_half* y, const qspec_half* hadK, qspec_half* out, int tokens, int K, int m,
                       void* stream);

/* o_proj input path of QuarotLlamaAttention.forward (quarot_llama.py:231-238): transpose -> FWHT over heads
 * (* had_scale) -> transpose back [-> Quantizer], one kernel.  attn [tokens, heads, head_dim].
 *   q == NULL: fp16 result in out_f16 [tokens, heads*head_dim]      (verify, OnlineHadamard w4a4=False)
 *   q != NULL: int4 rows in q [tokens, heads*head_dim/2] + scale[tokens]  (draft; out_f16 unused) */
int qspec_heads_hadamard(const qspec_half* attn, qspec_half* out_f16, int8_t* q, qspec_half* scale, float had_scale,
                         float clip_ratio, int tokens, int heads, int head_dim, void* stream);

/* The same head transform for head counts K * 2^p whose get_hadK factor is a table (Llama-2-13B: 40 heads = had40;
 * matmul_hadU_cuda on rows of `heads`, quarot/functional/hadamard.py:94-124, inside the transposes of
 * quarot_llama.py:231-234).  attn, out [tokens, heads, head_dim]; hadK [K, K] fp16; heads = K * 2^p, K in 2..172.
 *   y = h(WHT over the 2^p axis * had_scale);  out[i*2^p + p] = h(sum_k hadK[i,k] * y[k*2^p + p]), fp32 in k order.
 * The Quantizer behind it is qspec_fuse_sym_quant. */
int qspec_heads_hadamard_mix(const qspec_half* attn, const qspec_half* hadK, qspec_half* out, float had_scale, int tokens,
                             int heads, int head_dim, int K, void* stream);

/* `self.act_fn(gate) * up_proj` on the fused gate_up output (quarot_llama.py:279-284): up = [:, :I], gate = [:, I:];
 * out [tokens, I] = h(h(silu(gate)) * up).  Stand-alone form of the first stage of qspec_silu_mul_hadamard. */
int qspec_silu_mul(const qspec_half* gate_up, qspec_half* out, int tokens, int intermediate, void* stream);

/* down_proj input path of QuarotLlamaMLP.forward (quarot_llama.py:279-295): up = gate_up[:, :I],
 * gate = gate_up[:, I:]; silu(gate)*up -> (hadK (x) H_{I/K}) * had_scale [-> Quantizer], one kernel.
 * hadK [K,K] fp16 (ignored when K == 1).  Same q == NULL / != NULL convention. */
int qspec_silu_mul_hadamard(const qspec_half* gate_up, const qspec_half* hadK, qspec_half* out_f16, int8_t* q,
                            qspec_half* scale, float had_scale, float clip_ratio, int tokens, int intermediate, int K,
                            void* stream);

/* The Hadamard + quantiser tail of qspec_silu_mul_hadamard on an input that is ALREADY silu(gate)*up
 * (act [tokens, I], produced by qspec_gate_up_silu_linear_*).
 *   workspace: NULL, or qspec_xwg_workspace_bytes() bytes ZERO-FILLED once before first use (never reset afterwards)
 *   and used by ONE stream at a time: lets a token's transform spread over several workgroups (each mixes a block of
 *   columns; the quantiser's row maximum is exchanged through the workspace).  Same bytes out either way.
 *   Word 0 of the workspace is a sticky error flag: non-zero = an exchange timed out (results of that launch invalid). */
size_t qspec_xwg_workspace_bytes(void);
int qspec_mlp_hadamard(const qspec_half* act, const qspec_half* hadK, qspec_half* out_f16, int8_t* q, qspec_half* scale,
                       float had_scale, float clip_ratio, int tokens, int intermediate, int K, void* workspace,
                       void* stream);

/* ---- linear layers over the shared packed-int4 weight buffer --------------------------------- */

/* torch.ops.torchao.rowwise_scaled_linear_cutlass_s4s4_unified(xq, x_scale, wq, w_scale, bias, out)
 *   third-party/ao/torchao/ops.py:29,600-636; ..._cutlass_s4s4.cu:25-39; ..._unified.cuh:240-488
 *   (quarot_nn/linear.py:82).  xq [M,K/2], wq [N,K/2] int8 (two s4 per byte), xs [M], ws [N], bias [N] or NULL,
 *   out [M,N] fp16.  N % 16 == 0, K % 128 == 0. */
int qspec_rowwise_scaled_linear_s4s4(const int8_t* xq, const qspec_half* xs, const int8_t* wq, const qspec_half* ws,
                                     const qspec_half* bias, qspec_half* out, int M, int N, int K, void* stream);

/* The s4s4 linear above followed by the residual add of QuarotDecoderLayer (quarot_llama.py:380 `hidden = residual +
 * o_proj(...)`, :390 `residual + down_proj(...)`): resid_out = h(f(resid_in) + f(h(linear))), an fp16 add of the
 * fp16 GEMM result, in the GEMM's epilogue.  resid_out may alias resid_in.  M <= 32.  The norm that follows then
 * reads one tensor (qspec_ln_*_linear_s4s4 with delta = NULL, hidden_out = NULL). */
int qspec_rowwise_scaled_linear_s4s4_residual(const int8_t* xq, const qspec_half* xs, const int8_t* wq,
                                              const qspec_half* ws, const qspec_half* resid_in, qspec_half* resid_out,
                                              int M, int N, int K, void* stream);
int qspec_rowwise_scaled_linear_s4s4_residual_supported(int M, int N, int K);

/* qspec_rowwise_scaled_linear_s4s4_residual fed with UNQUANTISED fp16 rows x16 [M, K] and n_parts partial row maxima each
 * (part_amax [M, n_parts] fp32): row-absmax int4 quantisation (quant.cu:102-167: scale = h(h(amax / 7) * h(clip_ratio)),
 * q = clamp(rne(h(x / scale)), -8, 7), all-zero row -> 0) in the prologue of the GEMM launch.  M <= 4, K = 4096,
 * n_parts = 8 (what qspec_heads_hadamard_merged_spread leaves). */
int qspec_rowwise_scaled_linear_s4s4_residual_hq(const qspec_half* x16, const float* part_amax, int n_parts,
                                                 float clip_ratio, const int8_t* wq, const qspec_half* ws,
                                                 const qspec_half* resid_in, qspec_half* resid_out, int M, int N, int K,
                                                 void* stream);
int qspec_rowwise_scaled_linear_s4s4_residual_hq_supported(int M, int N, int K, int n_parts);

/* bitblas.Matmul.__call__(x, w ^ 0x88, output=C, scale=ws, bias=bias)  (quarot_nn/linear.py:102-124,156-211).
 *   Takes the SAME wq buffer as the s4s4 op (no XOR copy).  x [M,K] fp16.
 *   workspace: NULL, or qspec_w4a16_workspace_bytes() bytes ZERO-FILLED once before first use: lets narrow layers
 *   split K across workgroups (partial tiles + ticket counters, left zeroed by every call); with NULL the
 *   layer runs unsplit.  Results are deterministic either way. */
size_t qspec_w4a16_workspace_bytes(void);
int qspec_w4a16_linear(const qspec_half* x, const int8_t* wq, const qspec_half* ws, const qspec_half* bias,
                       qspec_half* out, int M, int N, int K, void* workspace, void* stream);

/* Long-K W4A16 layers at M <= 16 (down_proj, K = 14336: 16 rows x K of fp16 activations do not fit a CU's registers):
 * K is cut into `slices` = qspec_w4a16_linear_partial_slices(M, N, K) slices (0: not needed / not built) and the raw
 * fp32 sums of each slice go to part [slices][M][N].  Whoever consumes them forms h((p_0 + p_1 + ...) * f(ws[n])):
 * qspec_w4a16_linear does it with a finishing launch, the verify pass inside the norm that follows
 * (qspec_add_rms_norm_fp16_partial: hidden_out = x + that, out = LN(hidden_out)); same expression, same bits.
 * A shape that also runs unsliced (qspec_w4a16_linear_partial_slices = 0) takes any `slices` in 2..8 that divides K
 * into lengths the streaming kernel supports (K / slices = 128 * waves * steps). */
int qspec_w4a16_linear_partial_slices(int M, int N, int K);
int qspec_w4a16_linear_partial(const qspec_half* x, const int8_t* wq, float* part, int M, int N, int K, int slices,
                               void* stream);
int qspec_add_rms_norm_fp16_partial(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const float* part,
                                    const qspec_half* ws, int slices, float eps, int tokens, int hidden, void* stream);

/* qkv_proj fused with what follows it in QuarotLlamaAttention.forward (quarot_llama.py:183-226): the GEMM
 * (s4s4: linear.py:82 / w4a16: linear.py:122) -> ops.rotary_embedding on q,k (csrc/pos_encoding_kernels.cu:71-122)
 * -> reshape_and_cache_flash of k,v (csrc/cache_kernels.cu:207-303).  wq rows are [q; k; v] (fuse_qkv,
 * quarot_llama.py:152-173).  qkv [M, (num_heads + 2 num_kv_heads) * 128] receives rotated q,k and v;
 * head_size = rot_dim = 128, or 64 (TinyLlama) where the streaming kernels take the shape: qspec_qkv_rope_linear_supported
 * (w4a4 != 0: the s4s4 form) answers that.  Bit-identical to running the three ops one after the other. */
int qspec_qkv_rope_linear_supported(int w4a4, int M, int N, int K, int head_size);
int qspec_qkv_rope_linear_s4s4(const int8_t* xq, const qspec_half* xs, const int8_t* wq, const qspec_half* ws,
                               qspec_half* qkv, int M, int N, int K, const int64_t* positions,
                               const qspec_half* cos_sin_cache, qspec_half* key_cache, qspec_half* value_cache,
                               const int64_t* slot_mapping, int num_heads, int num_kv_heads, int head_size,
                               int rot_dim, void* stream);
int qspec_qkv_rope_linear_w4a16(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M,
                                int N, int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                int num_heads, int num_kv_heads, int head_size, int rot_dim, void* workspace,
                                void* stream);

/* gate_up GEMM fused with `act_fn(gate) * up` (quarot_llama.py:276-284): wq rows are [up; gate] (fuse_gate_up,
 * :301-314), act [M, I] = h(h(silu(gate)) * up).  Bit-identical to GEMM followed by qspec_silu_mul. */
int qspec_gate_up_silu_linear_s4s4(const int8_t* xq, const qspec_half* xs, const int8_t* wq, const qspec_half* ws,
                                   qspec_half* act, int M, int intermediate, int K, void* stream);
int qspec_gate_up_silu_linear_w4a16(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                    int intermediate, int K, void* workspace, void* stream);

/* Draft pass, decode-sized M (<= 16): the residual add + LN-no-gamma + per-token int4 quantisation that feeds
 * qkv_proj / gate_up (quarot_llama.py:373-374,380-388: `hidden = residual + proj_out; x = norm(hidden)`;
 * kernel third-party/kernels/csrc/layernorm_kernels.cu:569-716) runs as the PROLOGUE of the GEMM launch:
 *   h = fp16(hidden_in + delta) (delta NULL: h = hidden_in);  hidden_out = h;  (xq, xs) = ln_i4(h);  then the fused
 *   GEMM of qspec_qkv_rope_linear_s4s4 / qspec_gate_up_silu_linear_s4s4.
 * hidden_out must not alias hidden_in (every workgroup re-reads hidden_in; workgroup 0 writes hidden_out);
 * hidden_out may be NULL (nothing written): with delta NULL too this is the plain norm of hidden_in.
 * Bit-identical to qspec_add_rms_norm_i4 followed by the GEMM entry.  K = hidden size in {1024, 2048, 4096, 5120, 8192}. 
 * sync_workspace: NULL -> every workgroup recomputes the norm; else qspec_ln_linear_workspace_bytes() bytes,
 * ZERO-FILLED once before first use (every call leaves it zeroed): a few producer workgroups compute the norm and
 * hand the packed rows to the others through L2 (write-through stores + flags, no fences) while all of them already
 * stream weights.  Faster than recomputing from 8 tokens on, but slower than a separate qspec_add_rms_norm_i4 launch +
 * the (xq, xs) GEMM entry at those sizes (measured): the engine fuses up to 4 tokens only.  One workspace
 * serves all calls of a stream; not for concurrent streams.  The consumers WAIT for the producers on the GPU: the launch
 * must have the device to itself (<= one workgroup per CU; two processes sharing a GPU void this).  The wait is
 * bounded: on expiry int32 word 31 of the workspace is set (sticky) and the launch continues with stale rows -- the host
 * must read that word (QSpecEngine.error_flag does, once per cycle) and discard the step. */
size_t qspec_ln_linear_workspace_bytes(void);
int qspec_ln_qkv_rope_linear_s4s4(const qspec_half* hidden_in, const qspec_half* delta, qspec_half* hidden_out,
                                  float eps, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M, int N,
                                  int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                  qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                  int num_heads, int num_kv_heads, int head_size, int rot_dim, void* sync_workspace,
                                  void* stream);
int qspec_ln_gate_up_silu_linear_s4s4(const qspec_half* hidden_in, const qspec_half* delta, qspec_half* hidden_out,
                                      float eps, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                      int intermediate, int K, void* sync_workspace, void* stream);
int qspec_ln_linear_s4s4_supported(int M, int N, int K);   /* 1 if the two entries above accept the shape */

/* Tensor-parallel views of the SAME buffers (no reference counterpart: the reference QSpec model has no TP).
 * _ksliced: row-parallel shard = a K range of x [M, *] (row stride ldx halves) and of wq [N, *] (row stride ldw_bytes);
 *           the caller all-reduces the partial outputs.
 * _ksliced_raw: the same K range as raw fp32 sums part [M, N] (no channel scale, no rounding): reduced across ranks in
 *           fp32 and finished (scale, one fp16 rounding, residual add, norm) by qspec_add_rms_norm_fp16_partial with
 *           slices = 1, so that the sharded sum is rounded once, as on one GPU.  M <= 32.
 * _shard:   column-parallel shard of gate_up = intermediate channels [first_channel, first_channel + num_channels);
 *           act keeps its full [M, intermediate] layout, only the shard's columns are written. */
int qspec_w4a16_linear_ksliced(const qspec_half* x, int64_t ldx, const int8_t* wq, int64_t ldw_bytes,
                               const qspec_half* ws, qspec_half* out, int M, int N, int K, void* workspace,
                               void* stream);
int qspec_w4a16_linear_ksliced_raw(const qspec_half* x, int64_t ldx, const int8_t* wq, int64_t ldw_bytes, float* part,
                                   int M, int N, int K, void* workspace, void* stream);
int qspec_gate_up_silu_linear_w4a16_shard(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act,
                                          int M, int intermediate, int K, int first_channel, int num_channels,
                                          void* workspace, void* stream);

/* One-shot all-reduce over peer-mapped buffers (xGMI) for the small fp32 messages of the tensor-parallel verify pass.
 * Contract mirrored: vllm's custom all-reduce, one-shot form (csrc/custom_all_reduce.cuh;
 * vllm/distributed/device_communicators/custom_all_reduce.py:50-56,242-255): every rank maps every peer's buffer through
 * IPC handles exchanged on the host, messages below a size limit take this path, the library collective (RCCL) stays
 * the fallback.  Each rank PUSHES its vector into a slot of every peer's buffer (xGMI is point to point: one hop per
 * peer, all links at once), flags, and sums the world slots in RANK ORDER in fp32: same bits on every rank and on every
 * run.  No host-side state per call (device-side generation tags): capturable in a hipGraph.
 *   create: this rank's buffer (uncached device memory); max_bytes = capacity of one message, a multiple of 16.
 *   local_handle / open_peers: hipIpcMemHandle_t exchange (handles rank-major, qspec_oneshot_handle_bytes() each).
 *   all_reduce_f32: data [n] fp32 in place, n * 4 <= max_bytes, data 16-byte aligned; every rank must call it in the
 *     same order with the same n; ranks' kernels wait for each other on the GPU (bounded; qspec_oneshot_error != 0 if a
 *     wait timed out). */
int qspec_oneshot_create(int rank, int world, size_t max_bytes, void** ctx_out);
int qspec_oneshot_handle_bytes(void);
int qspec_oneshot_local_handle(void* ctx, void* handle_out);
int qspec_oneshot_open_peers(void* ctx, const void* handles);
int qspec_oneshot_all_reduce_f32(void* ctx, float* data, int n, void* stream);
int qspec_oneshot_error(void* ctx);
/* device address of this rank's sticky error word (int32; for qspec_collect_error_words, so that the engine reads it
 * with the cycle's output instead of synchronising the device from the host) */
void* qspec_oneshot_error_word(void* ctx);
int qspec_oneshot_destroy(void* ctx);

/* lm_head: F.linear(hidden, lm_head.weight)  (vllm/model_executor/layers/logits_processor.py:92-97). w [N,K] fp16. */
int qspec_linear_f16(const qspec_half* x, const qspec_half* w, qspec_half* out, int M, int N, int K, void* stream);

/* fp16 view of a packed weight with the channel scale folded in (prefill-sized M only). */
int qspec_dequant_w4(const int8_t* wq, const qspec_half* ws, qspec_half* out, int N, int K, void* stream);

/* ---- attention side --------------------------------------------------------------------------- */

/* torch.ops._C.rotary_embedding(positions, q, k, head_size, cos_sin_cache, is_neox=True)
 *   csrc/pos_encoding_kernels.cu:71-122 (quarot_llama.py:208-211).  In place. */
int qspec_rotary_embedding(const int64_t* positions, qspec_half* q, qspec_half* k, const qspec_half* cos_sin_cache,
                           int tokens, int num_heads, int num_kv_heads, int head_size, int rot_dim, int64_t q_stride,
                           int64_t k_stride, void* stream);

/* torch.ops._C_cache_ops.reshape_and_cache_flash(key, value, key_cache, value_cache, slot_mapping, "auto", 1, 1)
 *   csrc/cache_kernels.cu:207-303 (vllm/attention/backends/flash_attn.py:711-720).
 *   caches [num_blocks, block_size, num_kv_heads, head_size] fp16. */
int qspec_reshape_and_cache_flash(const qspec_half* key, const qspec_half* value, qspec_half* key_cache,
                                  qspec_half* value_cache, const int64_t* slot_mapping, int tokens, int num_kv_heads,
                                  int head_size, int64_t key_stride, int64_t value_stride, void* stream);

/* The two above fused over the fused qkv row [tokens, (num_heads + 2*num_kv_heads) * head_size]. */
int qspec_rope_kv_write(const int64_t* positions, qspec_half* qkv, const qspec_half* cos_sin_cache,
                        qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping, int tokens,
                        int num_heads, int num_kv_heads, int head_size, int rot_dim, void* stream);

/* flash_attn_with_kvcache (draft, q_len 1) / flash_attn_varlen_func (verify, q_len k+1, causal) over the paged
 * cache  (vllm/attention/backends/flash_attn.py:741-830).  q rows of sequence s are tokens q_start[s]..q_start[s+1]-1
 * and sit at absolute positions ctx_lens[s]-q_len .. ctx_lens[s]-1.
 * workspace: qspec_paged_attention_workspace_bytes(n_seqs*max_q_len, ...) bytes, ZERO-FILLED once before its
 * first use (it starts with the split-merge ticket counters, which every call leaves at zero again).
 * out == NULL: the context splits are NOT merged; their partials (o, m, l) stay in the workspace for
 * qspec_heads_hadamard_merged, which merges them in front of the head transform (the launch boundary then is the
 * hand-off between the split workgroups and no ticket / fence is needed).
 * Three kernels sit behind this entry (attention.hip), chosen from the shape alone: 16-row workgroups walking 128-key
 * chunks (decode / verify, splits of at most one chunk; also every out != NULL call with n_splits > 1); the same grid
 * with the keys over the waves of a workgroup (out == NULL and block_tables wide enough for splits longer than 128
 * keys); 64-row flash-style workgroups for prompt-sized queries (n_splits == 1, max_q_len * num_heads / num_kv_heads
 * >= 128).  Every block_tables entry must be a valid block number (unused tail entries: 0, as vLLM pads them): rows
 * past a sequence's context are read (clamped prefetch) and masked, never used.  head_size != 128 (even, <= 256; TinyLlama:
 * 64): a generic kernel without matrix cores, one workgroup per (token, head, split) over min(n_splits, 4) context splits of
 * whole 16-key groups; out != NULL merges them with a second small launch, out == NULL (head_size % 8 == 0) leaves the
 * partials to qspec_heads_hadamard_merged -- the same merge expression, so the same bits either way. */
size_t qspec_paged_attention_workspace_bytes(int max_tokens, int num_heads, int head_size, int n_splits);
int qspec_paged_attention(const qspec_half* q, int64_t q_stride, const qspec_half* key_cache,
                          const qspec_half* value_cache, const int32_t* block_tables, int max_blocks_per_seq,
                          const int32_t* ctx_lens, const int32_t* q_start, int n_seqs, int tokens, int max_q_len,
                          int num_heads, int num_kv_heads, int head_size, int block_size, float sm_scale, int n_splits,
                          void* workspace, qspec_half* out, void* stream);

/* qspec_heads_hadamard on the un-merged output of qspec_paged_attention(..., out = NULL): split merge (same
 * expression as the attention kernel's own merge, rounded to fp16 where flash-attn returns fp16) + head Hadamard
 * (+ row-absmax int4 quant when q != NULL).  attn_workspace / max_tokens (= n_seqs * max_q_len) / n_splits are
 * those of the attention call.  head_dim 128 with 32 or 64 heads; 32 heads of another size % 8 == 0 (64).  The merge itself
 * is the attention kernel's expression;
 * with splits of at most 128 keys the result is bit-identical to qspec_paged_attention(out != NULL) followed by
 * qspec_heads_hadamard (longer splits take a kernel with another summation order inside a split: equal within 1e-3). */
int qspec_heads_hadamard_merged(const void* attn_workspace, int max_tokens, int n_splits, qspec_half* out_f16, int8_t* q,
                                qspec_half* scale, float had_scale, float clip_ratio, int tokens, int heads,
                                int head_dim, void* stream);

/* The same split merge + head Hadamard SPREAD over 8 workgroups per token (32 heads of 128; each workgroup owns 16 columns
 * of every head -- the transform mixes heads, never columns): fp16 rows out_f16 [tokens, heads * head_dim] plus
 * part_amax [tokens, 8] fp32, max |out| of each workgroup's share.  The quantiser of quarot_llama.py:235-238
 * (Quantizer -> quant.cu:102-167) then runs in the prologue of the o_proj launch:
 * qspec_rowwise_scaled_linear_s4s4_residual_hq below.  Same bits as qspec_heads_hadamard_merged(q != NULL) followed by
 * qspec_rowwise_scaled_linear_s4s4_residual. */
int qspec_heads_hadamard_merged_spread(const void* attn_workspace, int max_tokens, int n_splits, qspec_half* out_f16,
                                       float* part_amax, float had_scale, int tokens, int heads, int head_dim,
                                       void* stream);
int qspec_heads_hadamard_merged_spread_supported(int tokens, int heads, int head_dim);

/* The spread form for head counts with a TABLE FACTOR (heads = K * 2^p, hadK [K, K] = get_hadK(heads) of
 * third-party/QuaRot/quarot/functional/hadamard.py:94-124; Llama-2-13B: 40 heads = had40): split merge + FWHT over 2^p +
 * h(v * had_scale) + the k-ordered table mix, 8 workgroups per token.  out_f16 [tokens, heads * head_dim]; part_amax
 * [tokens, 8] or NULL (verify pass: fp16 rows only).  Same bits as qspec_paged_attention(out != NULL) followed by
 * qspec_heads_hadamard_mix (and, with part_amax, qspec_fuse_sym_quant inside
 * qspec_rowwise_scaled_linear_s4s4_residual_hq).  head_dim 128. */
int qspec_heads_hadamard_mix_merged_spread(const void* attn_workspace, int max_tokens, int n_splits, const qspec_half* hadK,
                                           int K, qspec_half* out_f16, float* part_amax, float had_scale, int tokens,
                                           int heads, int head_dim, void* stream);
int qspec_heads_hadamard_mix_merged_spread_supported(int tokens, int heads, int head_dim, int K);

/* The s4s4 linear (rowwise_scaled_linear_cutlass_s4s4_unified, quarot_nn/linear.py:67-84) at 17..32 tokens as K SLICES whose raw
 * int32 sums are finished by the consumer: ipart [slices, M, N] int32, slice s = sum over k in [s K / slices, (s + 1) K / slices)
 * of xq[m, k] wq[n, k].  Integer sums are exact in any order, so qspec_add_rms_norm_ipartial below -- which adds the slices,
 * applies the reference epilogue h((f(acc) f(xs[m])) f(ws[n])) (rowwise_scaled_linear_cutlass_unified.cuh:342-377), the fp16
 * residual add (quarot_llama.py:380,390) and the following norm (layernorm_kernels.cu:569-716) -- gives the bits of
 * qspec_rowwise_scaled_linear_s4s4 followed by qspec_add_rms_norm_i4 / _fp16.  Why: with one 16-row weight tile per workgroup
 * a workgroup reads twice as many activation bytes as weight bytes at 32 tokens; with two K slices it owns two tiles of half
 * the K and reads its activation fragments once for both.  _slices: the slice count built for (M, N, K), 0 = use the plain entry. */
int qspec_rowwise_scaled_linear_s4s4_partial_slices(int M, int N, int K);
int qspec_rowwise_scaled_linear_s4s4_partial(const int8_t* xq, const int8_t* wq, int32_t* ipart, int M, int N, int K, int slices,
                                             void* stream);
/* hidden_out = h(f(x) + f(delta)), delta as above from ipart / xs [tokens] / ws [hidden]; then the norm of hidden_out:
 * q != NULL: int4 rows + scale (generalLayerNorm_fuse_sum_i4), else fp16 rows out_f16.  hidden_out may alias x. */
int qspec_add_rms_norm_ipartial(int8_t* q, qspec_half* scale, qspec_half* out_f16, qspec_half* hidden_out, const qspec_half* x,
                                const int32_t* ipart, const qspec_half* xs, const qspec_half* ws, int slices, float eps,
                                int tokens, int hidden, void* stream);

/* ---- token side ------------------------------------------------------------------------------- */

/* nn.Embedding lookup (quarot_llama.py:497). */
int qspec_embedding(const int64_t* ids, const qspec_half* table, qspec_half* out, int tokens, int hidden, int vocab,
                    void* stream);

/* Sampler.forward, greedy, modify_greedy_probs=False  (vllm/model_executor/layers/sampler.py:216-316):
 *   probs = softmax(float(logits)) [tokens, vocab] fp32, token = argmax. */
size_t qspec_sampler_workspace_bytes(int rows);   /* rows = tokens (softmax) or batch*k (rejection sampler) */
int qspec_softmax_argmax(const qspec_half* logits, float* probs, int64_t* token, int tokens, int vocab,
                         void* workspace, void* stream);

/* lm_head + Sampler front end in one call (logits_processor.py:92-97 + sampler.py:270-287): the lm_head
 * launch leaves, next to the fp16 logits, every workgroup's row maxima, so the softmax needs no pass of its own to
 * find them and probs is written ONCE: two more launches read the (cache-resident) fp16 logits twice -- the
 * denominator in fp64, then p = qexp(l - max) / sum -- instead of writing and re-reading an fp32 array.  Same bits as
 * qspec_linear_f16 + qspec_softmax_argmax.  tokens <= 16 (_supported says); logits [tokens, vocab] is scratch the
 * caller provides; workspace: qspec_lm_head_sampler_workspace_bytes(tokens). */
size_t qspec_lm_head_sampler_workspace_bytes(int rows);
int qspec_lm_head_softmax_argmax_supported(int tokens, int vocab, int K);
int qspec_lm_head_softmax_argmax(const qspec_half* hidden, const qspec_half* lm_head, qspec_half* logits, float* probs,
                                 int64_t* token, int tokens, int vocab, int K, void* workspace, void* stream);

/* RejectionSampler.forward(target_with_bonus_probs, bonus_token_ids, draft_probs, draft_token_ids)
 *   vllm/model_executor/layers/rejection_sampler.py:60-154 + spec_decode_base_sampler.py:69-131.
 *   uniform [B,k] / exponential [B,k,V] fp32: injected random draws (tests); NULL -> Philox(seed, offset).
 *   rng_state: NULL, or a device uint64[2] = {seed, offset} that overrides the two scalars and whose offset
 *   is incremented by the call (so a captured graph draws fresh numbers on every replay).
 *   draft_probs element (b,i,v) is at b*dp_stride_b + i*dp_stride_k + v, draft_token_ids (b,i) at
 *   b*ids_stride_b + i*ids_stride_k, bonus b at b*bonus_stride (elements): contiguous [B,k,V]/[B,k]/[B] tensors
 *   pass (k*V, V, k, 1, 1); the draft loop writes step-major buffers and passes their strides instead of copying.
 *   out_tokens [B,k+1] (-1 = no token); accepted [B,k] u8; recovered [B,k];
 *   counters[3] += {accepted, emitted, draft} (may be NULL). */
int qspec_rejection_sample(const float* target_with_bonus_probs, const int64_t* bonus_token_ids,
                           const float* draft_probs, const int64_t* draft_token_ids, const float* uniform,
                           const float* exponential, uint64_t seed, uint64_t offset, uint64_t* rng_state, int batch,
                           int k, int vocab, int64_t dp_stride_b, int64_t dp_stride_k, int64_t ids_stride_b,
                           int64_t ids_stride_k, int64_t bonus_stride,
                           int64_t* out_tokens, uint8_t* accepted, int64_t* recovered, int64_t* counters,
                           const int32_t* active_lens, void* workspace, void* stream);

/* Sampler.forward for rows that are NOT plain greedy (vllm/model_executor/layers/sampler.py:216-316):
 *   l = float(logits) / temperature[row] (a temperature < 1e-5 marks a greedy row of a mixed batch: divisor 1.0, token =
 *   argmax; sampling_metadata.py:413-417), _apply_top_k_top_p (:387-413; top_k <= 0 or >= vocab: off; top_p >= 1: off),
 *   probs = softmax(masked l) fp32 [tokens, vocab], token = argmax(probs / Exp(1)) (_multinomial :585-604).
 *   temperature / top_k / top_p: per-row device arrays, each may be NULL (1.0 / off / off).
 *   exponential [tokens, vocab] fp32: injected draws (tests); NULL -> Philox(seed, offset) as qspec_rejection_sample, with
 *   rng_state (device uint64[2]) overriding the scalars and advanced by the call.
 *   Tokens with EQUAL logits at the top-p boundary are kept or masked together (the reference's unstable sort splits such a
 *   group arbitrarily); everything else follows the reference element for element.
 *   workspace: qspec_sample_workspace_bytes(tokens), zero-filled ONCE by the caller (every call leaves it zeroed). */
size_t qspec_sample_workspace_bytes(int rows);
int qspec_sample_top_k_top_p(const qspec_half* logits, const float* temperature, const int32_t* top_k, const float* top_p,
                             const float* exponential, uint64_t seed, uint64_t offset, uint64_t* rng_state, float* probs,
                             int64_t* token, int64_t token_stride, int tokens, int vocab, void* workspace, void* stream);

/* TypicalAcceptanceSampler.forward(target_with_bonus_probs, bonus_token_ids, draft_probs, draft_token_ids)
 *   vllm/model_executor/layers/typical_acceptance_sampler.py:37-172 (MEDUSA 3.3.1), selected by
 *   draft_token_acceptance_method = "typical_acceptance_sampler" (vllm/spec_decode/spec_decode_worker.py:95-110).
 *   Deterministic: accepted[b,i] = q[b,i,x] > min(posterior_threshold, posterior_alpha * exp(-H)) with
 *   H = -sum_v q_v log(q_v + 1e-5) over q = target_with_bonus_probs[:, :-1]; the replacement at the first rejected position is
 *   argmax_v q (first index on ties); output layout and counters as qspec_rejection_sample (draft_probs is unused by the
 *   reference and is not an argument).  Strides and workspace as there. */
int qspec_typical_acceptance_sample(const float* target_with_bonus_probs, const int64_t* bonus_token_ids,
                                    const int64_t* draft_token_ids, float posterior_threshold, float posterior_alpha, int batch,
                                    int k, int vocab, int64_t ids_stride_b, int64_t ids_stride_k, int64_t bonus_stride,
                                    int64_t* out_tokens, uint8_t* accepted, int64_t* recovered, int64_t* counters,
                                    const int32_t* active_lens, void* workspace, void* stream);

/* ops.advance_step_flashattn(num_seqs, num_queries, block_size, input_tokens, sampled_token_ids,
 *   input_positions, seq_lens, slot_mapping, block_tables)   csrc/prepare_inputs/advance_step.cu:14-64,192
 *   (vllm/attention/backends/flash_attn.py:365-373), num_seqs == num_queries. */
int qspec_advance_step_flashattn(int num_seqs, int block_size, int64_t* input_tokens,
                                 const int64_t* sampled_token_ids, int64_t* input_positions, int32_t* seq_lens,
                                 int64_t* slot_mapping, const int32_t* block_tables, int64_t block_tables_stride,
                                 void* stream);

/* ---- spec-decode cycle glue (input assembly + bookkeeping kept on the GPU) -------------------- */
/* Sequence state: seq_lens[b] = L known tokens (KV valid below L-1), last_token[b] = token at L-1.
 * seq_lens[b] <= 0 marks an EMPTY batch slot (request finished / not admitted yet): it runs through the cycle as a
 * dummy (token 0, position 0, slot -1: nothing written to the KV cache), emits nothing and is not counted
 * (qspec_rejection_sample's active_lens).  max_blocks_per_seq: entries of a block-table row; a position beyond it
 * gets slot -1 (no write) instead of an out-of-row lookup -- the host refuses such a step beforehand; 0 = unchecked.

 * First draft-step inputs (TP1DraftModelRunner.execute_model, vllm/spec_decode/draft_model_runner.py:169-262):
 *   token = last_token, position = L-1, ctx_len = L, slot from the block table. */
int qspec_spec_prepare_draft(int batch, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                             const int32_t* seq_lens, const int32_t* block_tables, int64_t block_tables_stride,
                             int64_t* input_tokens, int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens,
                             void* stream);

/* _gpu_advance_step between two draft steps (vllm/spec_decode/draft_model_runner.py:78-135) = qspec_advance_step_flashattn
 * with the two engine rules: a row whose slot is -1 stays put; a new position beyond the block table freezes the row. */
int qspec_spec_advance_draft(int batch, int block_size, int max_blocks_per_seq, int64_t* input_tokens,
                             const int64_t* sampled_token_ids, int64_t* positions, int32_t* ctx_lens,
                             int64_t* slot_mapping, const int32_t* block_tables, int64_t block_tables_stride,
                             void* stream);

/* MQAScorer.score_proposals (vllm/spec_decode/mqa_scorer.py:12-76): per sequence the k+1 query tokens
 * [last_token, d_1..d_k] at positions L-1..L-1+k over the SAME block table (verify KV overwrites draft KV). */
int qspec_spec_prepare_verify(int batch, int k, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                              const int64_t* draft_token_ids, int64_t ids_stride_b, int64_t ids_stride_k,
                              const int32_t* seq_lens, const int32_t* block_tables,
                              int64_t block_tables_stride, int64_t* tokens, int64_t* positions, int64_t* slot_mapping,
                              int32_t* ctx_lens, void* stream);

/* The three steps above with the embedding lookup of the forward that follows them (nn.Embedding, quarot_llama.py:484-500:
 * hidden_out[row] = embed_tokens[token of the row]; a padded row takes row 0) in the same launch: one workgroup per row, the
 * bookkeeping on its first thread.  Same outputs as the step followed by qspec_embedding.  prepare_draft additionally takes
 * step_mask (may be NULL): eff_lens[b] = seq_lens[b] * step_mask[b] is written and used in place of seq_lens (slots that sit a
 * step out count as empty). */
int qspec_spec_prepare_draft_embed(int batch, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                                   const int32_t* seq_lens, const int32_t* step_mask, int32_t* eff_lens,
                                   const int32_t* block_tables, int64_t block_tables_stride, int64_t* input_tokens,
                                   int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens, const qspec_half* embed_tokens,
                                   qspec_half* hidden_out, int hidden, int vocab, void* stream);
int qspec_spec_advance_draft_embed(int batch, int block_size, int max_blocks_per_seq, int64_t* input_tokens,
                                   const int64_t* sampled_token_ids, int64_t* positions, int32_t* ctx_lens,
                                   int64_t* slot_mapping, const int32_t* block_tables, int64_t block_tables_stride,
                                   const qspec_half* embed_tokens, qspec_half* hidden_out, int hidden, int vocab, void* stream);
int qspec_spec_prepare_verify_embed(int batch, int k, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                                    const int64_t* draft_token_ids, int64_t ids_stride_b, int64_t ids_stride_k,
                                    const int32_t* seq_lens, const int32_t* block_tables, int64_t block_tables_stride,
                                    int64_t* tokens, int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens,
                                    const qspec_half* embed_tokens, qspec_half* hidden_out, int hidden, int vocab, void* stream);

/* SpecDecodeWorker._create_output_sampler_list bookkeeping (vllm/spec_decode/spec_decode_worker.py:972-1063):
 * append the emitted prefix of out_tokens[b] (-1 = nothing) to gen_tokens[b] (may be NULL), advance seq_lens,
 * last_token = last emitted token. */
int qspec_spec_commit(int batch, int k, const int64_t* out_tokens, int32_t* seq_lens, int64_t* last_token,
                      int64_t* gen_tokens, int32_t* gen_lens, int gen_capacity, void* stream);

/* Recovery support of the cycle (no reference counterpart: the reference has no device-side hand-offs).
 * qspec_spec_snapshot: copies the small per-cycle sequence state aside (restore = 0) or back (restore = 1):
 *   seq_lens / gen_lens [batch] i32 -> snap_i32 [2 * batch]; last_token [batch], the sampler's counters [3]
 *   (spec_decode_base_sampler.py:29-31) and Philox state [2] i64 -> snap_i64 [batch + 5].
 * qspec_collect_error_words: out[0] = OR of |*w_i| over up to four sticky int32 error words (NULL = none: the
 *   hand-off workspaces, the one-shot all-reduce); clear != 0 also zeroes them; out may be NULL then. */
int qspec_spec_snapshot(int batch, int restore, int32_t* seq_lens, int32_t* gen_lens, int64_t* last_token,
                        int64_t* counters, int64_t* rng_state, int32_t* snap_i32, int64_t* snap_i64, void* stream);
int qspec_collect_error_words(int32_t* w0, int32_t* w1, int32_t* w2, int32_t* w3, int clear, int64_t* out,
                              void* stream);

/* Activation layout of the verify pass at <= 16 tokens (no reference counterpart: the reference's fp16 GEMM is
 * bitblas.Matmul on row-major x, quarot_nn/linear.py:122).  The W4A16 GEMMs above consume x as v_mfma_f32_16x16x32_f16
 * operand fragments; reading row-major x means one pass through LDS per launch to regroup it.  The `_xp` twins below
 * move that regrouping into the PRODUCER's store: the norm, the head transform and the MLP transform write a
 * FRAGMENT-MAJOR tile of always 16 rows x K halves (rows >= tokens are never read back into a result):
 *     offset(r, k) = ((((k / 128) * 4 + ((k % 128) / 8) % 4) * 64 + ((k % 128) / 32) * 16 + r) * 8 + p(k % 8),
 *     p(e) = 2 * (e % 4) + e / 4
 * (8 consecutive halves = the two 4-half operand registers of k-group g at k and k+4 of one 16x16x32 instruction pair),
 * and the GEMM twins load their operand registers straight from it.  Every `_xp` entry computes exactly the bits of
 * its row-major twin -- only where a value is stored or loaded changes.  Same prototypes, same error codes; an `_xp`
 * call at a shape without a fragment-major form fails (ask qspec_w4a16_act_layout_supported(M, K) first: 1 = the GEMMs
 * at (M tokens, K) read the tile).  Producers: out / out_f16 is the tile (tokens <= 16; heads_hadamard_merged: 32 heads,
 * q == NULL; mlp_hadamard: workspace != NULL, q == NULL; mix_merged_spread: part_amax == NULL).  Consumers: x is the tile. */
int qspec_w4a16_act_layout_supported(int M, int K);
/* 17..32 tokens: x is TWO such tiles back to back (rows 0..15, then rows 16..31: tile t at x + t * 16 * K), read by a streaming
 * kernel that takes two token tiles over one pass of the weights (K in several passes where the operand registers of one do not
 * hold it: 8192, 14336, 28672).  qspec_w4a16_act_layout32_supported(M, N, K) = 1 where built; the `_xp32` entries have the
 * argument meaning of their row-major twins (no workspace: nothing is split across workgroups), results within 1e-3 of them
 * (another fp32 summation order than the M-tiled kernel's that 17+ tokens otherwise take). */
int qspec_w4a16_act_layout32_supported(int M, int N, int K);
int qspec_w4a16_linear_xp32(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* out, int M, int N, int K,
                            void* stream);
/* long-K layers (down_proj): K slices across the workgroups, raw fp32 sums part [slices][M][N] for qspec_add_rms_norm_fp16_partial */
int qspec_w4a16_linear_partial_slices_xp32(int M, int N, int K);
int qspec_w4a16_linear_partial_xp32(const qspec_half* x, const int8_t* wq, float* part, int M, int N, int K, int slices, void* stream);
int qspec_qkv_rope_linear_w4a16_xp32(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M,
                                     int N, int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                     qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                     int num_heads, int num_kv_heads, int head_size, int rot_dim, void* stream);
int qspec_gate_up_silu_linear_w4a16_xp32(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                         int intermediate, int K, void* stream);
int qspec_mlp_hadamard_act_layout_supported(int tokens, int intermediate, int K);   /* 1: qspec_mlp_hadamard_xp exists here */
int qspec_add_rms_norm_fp16_xp(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const qspec_half* delta,
                               float eps, int tokens, int hidden, void* stream);
int qspec_add_rms_norm_fp16_partial_xp(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const float* part,
                                       const qspec_half* ws, int slices, float eps, int tokens, int hidden, void* stream);
int qspec_heads_hadamard_merged_xp(const void* attn_workspace, int max_tokens, int n_splits, qspec_half* out_f16, int8_t* q,
                                   qspec_half* scale, float had_scale, float clip_ratio, int tokens, int heads,
                                   int head_dim, void* stream);
int qspec_heads_hadamard_mix_merged_spread_xp(const void* attn_workspace, int max_tokens, int n_splits, const qspec_half* hadK,
                                              int K, qspec_half* out_f16, float* part_amax, float had_scale, int tokens,
                                              int heads, int head_dim, void* stream);
int qspec_mlp_hadamard_xp(const qspec_half* act, const qspec_half* hadK, qspec_half* out_f16, int8_t* q, qspec_half* scale,
                          float had_scale, float clip_ratio, int tokens, int intermediate, int K, void* workspace,
                          void* stream);
int qspec_w4a16_linear_xp(const qspec_half* x, const int8_t* wq, const qspec_half* ws, const qspec_half* bias,
                          qspec_half* out, int M, int N, int K, void* workspace, void* stream);
int qspec_w4a16_linear_partial_xp(const qspec_half* x, const int8_t* wq, float* part, int M, int N, int K, int slices,
                                  void* stream);
int qspec_qkv_rope_linear_w4a16_xp(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M,
                                   int N, int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                   qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                   int num_heads, int num_kv_heads, int head_size, int rot_dim, void* workspace,
                                   void* stream);
int qspec_gate_up_silu_linear_w4a16_xp(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                       int intermediate, int K, void* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* QSPEC_HIP_H */
