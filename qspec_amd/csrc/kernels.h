// Internal C++ launch functions behind the C ABI (include/qspec_hip.h).
// Every function enqueues on `st`, never synchronises, returns 0 or a
// negative code for an unsupported shape (the C ABI turns it into an error string).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

// Dev knobs (environment switches for A/B measurements and sweeps) exist only in the EXPERIMENTAL build
// (-DQS_EXPERIMENTAL: `make -C qspec_amd/csrc experimental`); in libqspec_hip.so every knob is its default, a constant.
#ifdef QS_EXPERIMENTAL
#define QS_DEV_KNOB(name, dflt) (getenv(name) ? atoi(getenv(name)) : (dflt))
#else
#define QS_DEV_KNOB(name, dflt) (dflt)
#endif

namespace qspec {
typedef _Float16 f16;

// norm_quant.hip
int ln_quant_i4(const f16* x, const f16* delta, f16* hidden_out, int8_t* q, f16* scale, f16* isum, float eps, int T,
                int H, hipStream_t st);
int ln_fp16(const f16* x, const f16* delta, f16* hidden_out, f16* out, float eps, int T, int H, hipStream_t st, int xp = 0);
// delta = h((part[0] + ... + part[S-1])[t, :] * f(ws[:])): the K-sliced W4A16 projection finished inside the norm
int ln_ipartial(const f16* x, const int* ipart, const f16* xs, const f16* ws, int S, f16* hidden_out, f16* out, int8_t* q,
                f16* scale, float eps, int T, int H, hipStream_t st);
int ln_fp16_partial(const f16* x, const float* part, const f16* ws, int S, f16* hidden_out, f16* out, float eps, int T,
                    int H, hipStream_t st, int xp = 0);
int rowabsmax_quant(const f16* x, f16* scale, int8_t* q, float clip, int T, int K, hipStream_t st);

// hadamard.hip
int fwht(const f16* x, float scale, f16* out, int64_t rows, int N, hipStream_t st);
int hadk_mix(const f16* y, const f16* hadK, f16* out, int T, int K, int M, hipStream_t st);
int heads_hadamard(const f16* attn, f16* out_f16, int8_t* q, f16* scale, float had_scale, float clip, int T,
                   int heads, int d, hipStream_t st);
int heads_hadamard_mix(const f16* attn, const f16* hadK, f16* out, float had_scale, int T, int heads, int d, int K,
                       hipStream_t st);
int heads_hadamard_merge(const float* ws, int max_tokens, int n_splits, f16* out_f16, int8_t* q, f16* scale,
                         float had_scale, float clip, int T, int heads, int d, hipStream_t st, int xp = 0);
int silu_mul(const f16* gate_up, f16* out, int T, int I, hipStream_t st);
bool mlp_hadamard_xperm_supported(int T, int I, int K);
// 17..32 tokens: two fragment-major 16-row tiles, the two-token-tile streaming kernel (gemm_stream.hip: gemm_w4a16_stream2_kernel)
bool gemm_w4a16_stream32_supported(int M, int N, int K);
int gemm_w4a16_stream32_partial_slices(int M, int N, int K);
int gemm_w4a16_stream32_partial(const f16* x, const int8_t* wq, float* part, int M, int N, int K, int S, hipStream_t st);
int gemm_w4a16_stream32(const f16* x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K, hipStream_t st);
int gemm_w4a16_stream32_qkv_rope(const f16* x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                                 const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                                 const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st);
int gemm_w4a16_stream32_gate_up_silu(const f16* x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K, hipStream_t st);
int silu_mul_hadamard(const f16* gate_up, const f16* hadK, f16* out_f16, int8_t* q, f16* scale, float had_scale,
                      float clip, int T, int I, int K, int pre_activated, void* xws, hipStream_t st, int xp = 0);
size_t xwg_workspace_bytes();

// gemm.hip
int gemm_w4a4(const int8_t* xq, const f16* xs, const int8_t* wq, const f16* ws, const f16* bias, f16* out, int M,
              int N, int K, hipStream_t st);
size_t gemm_w4a16_ws_bytes();
int gemm_w4a16(const f16* x, const int8_t* wq, const f16* ws, const f16* bias, f16* out, int M, int N, int K,
               void* wsp, hipStream_t st);
int gemm_w4a4_qkv_rope(const int8_t* xq, const f16* xs, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                       const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                       const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st);
int gemm_w4a16_qkv_rope(const f16* x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                        const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                        const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, void* wsp, hipStream_t st);
int gemm_w4a4_gate_up_silu(const int8_t* xq, const f16* xs, const int8_t* wq, const f16* ws, f16* act, int M, int I,
                           int K, hipStream_t st);
int gemm_w4a16_gate_up_silu(const f16* x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K, int ch0,
                            int nch, void* wsp, hipStream_t st);
int gemm_w4a16_strided_raw(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, float* part, int M, int N, int K,
                           void* wsp, hipStream_t st);
int gemm_w4a16_strided(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, const f16* ws, f16* out, int M, int N,
                       int K, void* wsp, hipStream_t st);
int gemm_f16(const f16* x, const f16* w, f16* out, int M, int N, int K, hipStream_t st);
int dequant_w4(const int8_t* wq, const f16* ws, f16* out, int N, int K, hipStream_t st);

// gemm_stream.hip: W4A4 for M <= 16 with the activations produced in the prologue.
// Either (xq, xs) -- packed int4 rows + scales -- or (hidden_in, delta, hidden_out, eps): the residual add +
// LN-no-gamma + int4 quant of quarot_llama.py:373-388 fused in front of the GEMM (hidden_out != hidden_in).
struct StreamActs {
    const int8_t* xq = nullptr;
    const f16* xs = nullptr;
    const f16* hidden_in = nullptr;
    const f16* delta = nullptr;
    f16* hidden_out = nullptr;
    float eps = 0.0f;
    int* sync = nullptr;   // gemm_w4a4_stream_sync_bytes() bytes, zero-filled once: LN by producer workgroups + hand-off
};
size_t gemm_w4a4_stream_sync_bytes();
bool gemm_w4a4_stream_supported(int M, int N, int K, bool ln);
int gemm_w4a4_stream(const StreamActs& x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K,
                     hipStream_t st);
int gemm_w4a4_stream_residual(const StreamActs& x, const int8_t* wq, const f16* ws, const f16* resid_in, f16* resid_out,
                              int M, int N, int K, hipStream_t st);
// the same launch fed with fp16 rows + `nparts` partial row maxima each: row-absmax int4 quantisation (quant.cu:102-167) in the
// prologue (M <= 4, K = 4096, nparts = 8: the spread head-Hadamard's output)
int gemm_w4a4_stream_residual_hq(const f16* x16, const float* part_amax, int nparts, float clip, const int8_t* wq, const f16* ws,
                                 const f16* resid_in, f16* resid_out, int M, int N, int K, hipStream_t st);
bool gemm_w4a4_stream_residual_hq_supported(int M, int N, int K, int nparts);
int gemm_w4a4_stream_partial_slices(int M, int N, int K);
int gemm_w4a4_stream_partial(const int8_t* xq, const int8_t* wq, int* ipart, int M, int N, int K, int S, hipStream_t st);
int heads_hadamard_mix_merge_spread(const float* ws, int max_tokens, int n_splits, const f16* hadK, f16* out_f16,
                                    float* part_amax, float had_scale, int T, int heads, int d, int K, hipStream_t st, int xp = 0);
bool heads_hadamard_mix_merge_spread_supported(int T, int heads, int d, int K);
int heads_hadamard_merge_spread(const float* ws, int max_tokens, int n_splits, f16* out_f16, float* part_amax, float had_scale,
                                int T, int heads, int d, hipStream_t st);
int gemm_w4a4_stream_qkv_rope(const StreamActs& x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                              const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                              const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st);
int gemm_w4a4_stream_gate_up_silu(const StreamActs& x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K,
                                  hipStream_t st);

// W4A16 on the same streaming skeleton (M <= 16, K = 128 * NW * UB for a built (NW, UB))
bool gemm_w4a16_stream_supported(int M, int N, int K);
bool gemm_w4a16_xperm_supported(int M, int K);
int gemm_w4a16_stream(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, const f16* ws, f16* out, int M, int N,
                      int K, hipStream_t st, int xperm = 0);
int gemm_w4a16_stream_partial_slices(int M, int N, int K);
int gemm_w4a16_stream_partial(const f16* x, int64_t ldx, const int8_t* wq, int64_t ldw, float* part, int M, int N, int K,
                              int S, hipStream_t st, int xperm = 0);
int gemm_w4a16_partial_finish(const float* part, const f16* ws, f16* out, int M, int N, int S, hipStream_t st);
int gemm_w4a16_stream_qkv_rope(const f16* x, const int8_t* wq, const f16* ws, f16* qkv, int M, int N, int K,
                               const int64_t* positions, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                               const int64_t* slot_mapping, int nq, int nkv, int d, int rot_dim, hipStream_t st, int xperm = 0);
int gemm_w4a16_stream_gate_up_silu(const f16* x, const int8_t* wq, const f16* ws, f16* act, int M, int I, int K, int ch0,
                                   int nch, hipStream_t st, int xperm = 0);

// gemm_tiled.hip: W4A16 for prefill-sized M (tiles of 32..128 tokens x 128 weight rows, 32x32x16 MFMA)
bool gemm_w4a16_tiled_supported(int M, int N, int K);
// part (may be NULL): part_bytes of fp32 scratch for the K-sliced plan of narrow layers
int gemm_w4a16_tiled_partial_slices(int M, int N, int K);
int gemm_w4a16_tiled_partial(const f16* x, const int8_t* wq, float* part, int M, int N, int K, int S, hipStream_t st);
int gemm_w4a16_tiled(const f16* x, const int8_t* wq, const f16* ws, f16* out, int M, int N, int K, float* part,
                     size_t part_bytes, hipStream_t st);

bool gemm_f16_tiled_supported(int M, int N, int K);
int gemm_f16_tiled(const f16* x, const f16* w, f16* out, int M, int N, int K, hipStream_t st);

bool gemm_f16_stream_supported(int M, int N, int K);
int gemm_f16_stream(const f16* x, const f16* w, f16* out, int M, int N, int K, void* part_max, hipStream_t st);
int gemm_f16_stream_grid(int N);
#ifdef QS_EXPERIMENTAL
int prefetch_l2(const void* p, size_t bytes, int workgroups, hipStream_t st);
int prefetch_tiles(const void* p, size_t tile_bytes, int first_tile, int ntiles, int workgroups, hipStream_t st);
#endif

// attention.hip
int rope_kv_write(const int64_t* positions, f16* qkv, const f16* cos_sin_cache, f16* key_cache, f16* value_cache,
                  const int64_t* slot_mapping, int T, int nq, int nkv, int d, int rot_dim, hipStream_t st);
int rotary_embedding(const int64_t* positions, f16* q, f16* k, const f16* cos_sin_cache, int T, int nq, int nkv,
                     int d, int rot_dim, int64_t q_stride, int64_t k_stride, hipStream_t st);
int reshape_and_cache_flash(const f16* key, const f16* value, f16* key_cache, f16* value_cache,
                            const int64_t* slot_mapping, int T, int nkv, int d, int64_t k_stride, int64_t v_stride,
                            hipStream_t st);
int paged_attention(const f16* q, int64_t q_stride, const f16* key_cache, const f16* value_cache,
                    const int32_t* block_tables, int max_blocks, const int32_t* ctx_lens, const int32_t* q_start,
                    int n_seqs, int max_q_len, int nq, int nkv, int d, int block_size, float sm_scale, int n_splits,
                    float* ws, f16* out, hipStream_t st);
size_t paged_attention_ws_bytes(int T, int nq, int d, int n_splits);
size_t paged_attention_ws_o_offset();
int paged_attention_generic_splits(int n_splits);   // head sizes other than 128: the splits actually used for a caller's n_splits
size_t paged_attention_ws_ml_offset(int Tmax, int nq, int d, int n_splits);

// sampler.hip
int embedding(const int64_t* ids, const f16* table, f16* out, int T, int H, int V, hipStream_t st);
size_t sampler_ws_bytes(int rows);
int softmax_argmax(const f16* logits, float* probs, int64_t* token, int T, int V, void* ws, hipStream_t st);
// sampler front end behind an lm_head launch that left per-workgroup row maxima (gemm_f16_stream, part_max)
size_t head_softmax_ws_bytes(int rows);
void* head_softmax_part_max(void* ws, int rows);
int head_softmax_argmax(const f16* logits, float* probs, int64_t* token, int T, int V, int nparts, void* ws, hipStream_t st);
int rejection_sample(const float* target_probs, const float* draft_probs, const int64_t* draft_ids,
                     const int64_t* bonus_ids, const float* uniform, const float* exponential, uint64_t seed,
                     uint64_t offset, uint64_t* rng_state, int B, int k, int V, int64_t dp_sb, int64_t dp_sk,
                     int64_t di_sb, int64_t di_sk, int64_t bonus_stride, int64_t* out_tokens, uint8_t* accepted,
                     int64_t* recovered, int64_t* counters, const int32_t* active_lens, void* ws, hipStream_t st);
size_t sample_ws_bytes(int rows);
int sample_top_k_top_p(const f16* logits, const float* temperature, const int32_t* top_k, const float* top_p,
                       const float* exponential, uint64_t seed, uint64_t offset, uint64_t* rng_state, float* probs,
                       int64_t* token, int64_t token_stride, int T, int V, void* ws, hipStream_t st);
int typical_acceptance_sample(const float* target_probs, const int64_t* draft_ids, const int64_t* bonus_ids,
                              float posterior_threshold, float posterior_alpha, int B, int k, int V, int64_t di_sb,
                              int64_t di_sk, int64_t bonus_stride, int64_t* out_tokens, uint8_t* accepted, int64_t* recovered,
                              int64_t* counters, const int32_t* active_lens, void* ws, hipStream_t st);
int advance_step(int n, int block_size, int64_t* input_tokens, const int64_t* sampled, int64_t* positions,
                 int32_t* seq_lens, int64_t* slot_mapping, const int32_t* block_tables, int64_t bt_stride,
                 hipStream_t st);
int spec_advance_draft(int n, int block_size, int max_blocks, int64_t* input_tokens, const int64_t* sampled,
                       int64_t* positions, int32_t* ctx_lens, int64_t* slot_mapping, const int32_t* block_tables,
                       int64_t bt_stride, hipStream_t st);
int spec_prepare_draft_embed(int B, int block_size, int max_blocks, const int64_t* last_token, const int32_t* seq_lens,
                             const int32_t* step_mask, int32_t* eff_lens, const int32_t* block_tables, int64_t bt_stride,
                             int64_t* input_tokens, int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens, const f16* table,
                             f16* hidden_out, int H, int V, hipStream_t st);
int spec_advance_draft_embed(int n, int block_size, int max_blocks, int64_t* input_tokens, const int64_t* sampled, int64_t* positions,
                             int32_t* ctx_lens, int64_t* slot_mapping, const int32_t* block_tables, int64_t bt_stride, const f16* table,
                             f16* hidden_out, int H, int V, hipStream_t st);
int spec_prepare_verify_embed(int B, int k, int block_size, int max_blocks, const int64_t* last_token, const int64_t* draft_ids,
                              int64_t di_sb, int64_t di_sk, const int32_t* seq_lens, const int32_t* block_tables, int64_t bt_stride,
                              int64_t* v_tokens, int64_t* v_positions, int64_t* v_slots, int32_t* v_ctx_lens, const f16* table,
                              f16* hidden_out, int H, int V, hipStream_t st);
int spec_prepare_draft(int B, int block_size, int max_blocks, const int64_t* last_token, const int32_t* seq_lens,
                       const int32_t* block_tables, int64_t bt_stride, int64_t* input_tokens, int64_t* positions,
                       int64_t* slot_mapping, int32_t* ctx_lens, hipStream_t st);
int spec_prepare_verify(int B, int k, int block_size, int max_blocks, const int64_t* last_token, const int64_t* draft_ids,
                        int64_t di_sb, int64_t di_sk, const int32_t* seq_lens, const int32_t* block_tables, int64_t bt_stride, int64_t* v_tokens,
                        int64_t* v_positions, int64_t* v_slots, int32_t* v_ctx_lens, hipStream_t st);
int spec_snapshot(int B, int restore, int32_t* seq_lens, int32_t* gen_lens, int64_t* last_token, int64_t* counters,
                  int64_t* rng_state, int32_t* snap_i32, int64_t* snap_i64, hipStream_t st);
int collect_error_words(int32_t* w0, int32_t* w1, int32_t* w2, int32_t* w3, int clear, int64_t* out, hipStream_t st);
int spec_commit(int B, int k, const int64_t* out_tokens, int32_t* seq_lens, int64_t* last_token, int64_t* gen_tokens,
                int32_t* gen_lens, int gen_cap, hipStream_t st);
}  // namespace qspec
