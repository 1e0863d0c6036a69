// extern "C" boundary: argument checks, error strings, launch-error capture.
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/qspec_hip.h"
#include "kernels.h"

using qspec::f16;

static thread_local char g_err[512] = "";

static int fail(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return 1;
}
static int finish(const char* op, int rc) {
    if (rc != 0) return fail("%s: unsupported shape/arguments (code %d)", op, rc);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail("%s: launch failed: %s", op, hipGetErrorString(e));
    return 0;
}
#define NONNULL(op, p) \
    if (!(p)) return fail("%s: argument `%s` is NULL", op, #p)
#define ST ((hipStream_t)stream)
#define H(p) reinterpret_cast<f16*>(p)
#define CH(p) reinterpret_cast<const f16*>(p)

// QSPEC_OLD_GEMM=1 keeps the first-generation one-tile-per-workgroup W4A4 kernel (A/B measurements)
static bool use_stream() {
    static const int old = QS_DEV_KNOB("QSPEC_OLD_GEMM", 0);
    return old != 1;
}
// smallest M routed to the M-tiled W4A16 kernel (gemm_tiled.hip); QSPEC_TILED_MIN_M overrides (0 disables)
static int tiled_min_m() {
    // (33 until round 4: 17..32 rows ran the 2-D kernel of gemm.hip -- 70B gate_up at 32 rows 83.8 us)
    static const int v = QS_DEV_KNOB("QSPEC_TILED_MIN_M", 17);
    return v <= 0 ? (1 << 30) : v;
}

// The `_xp` entry points (fragment-major activation tiles between the verify pass's producers and its W4A16 GEMMs) are the base
// entry points called with this flag raised: one body per op.  Thread-local, raised only for the duration of an `_xp` call.
static thread_local int g_xp = 0;
struct XpScope {
    XpScope() { g_xp = 1; }
    ~XpScope() { g_xp = 0; }
};

extern "C" {

int qspec_abi_version(void) { return 6; }
const char* qspec_last_error(void) { return g_err; }

int qspec_rms_norm_general_fuse_sum_i4(int8_t* out_q, const qspec_half* x, qspec_half* input_sum, qspec_half* scaling,
                                       float eps, int tokens, int hidden, void* stream) {
    const char* op = "qspec_rms_norm_general_fuse_sum_i4";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, out_q); NONNULL(op, x); NONNULL(op, scaling);
    if (hidden % 1024 || hidden > 8192 || hidden <= 0) return fail("%s: hidden=%d must be a multiple of 1024, <= 8192", op, hidden);
    return finish(op, qspec::ln_quant_i4(CH(x), nullptr, nullptr, out_q, H(scaling), H(input_sum), eps, tokens, hidden, ST));
}
int qspec_rms_norm_general_fuse_sum_fp16(qspec_half* out, const qspec_half* x, float eps, int tokens, int hidden,
                                         void* stream) {
    const char* op = "qspec_rms_norm_general_fuse_sum_fp16";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, out); NONNULL(op, x);
    if (hidden % 1024 || hidden > 8192 || hidden <= 0) return fail("%s: hidden=%d must be a multiple of 1024, <= 8192", op, hidden);
    return finish(op, qspec::ln_fp16(CH(x), nullptr, nullptr, H(out), eps, tokens, hidden, ST));
}
int qspec_add_rms_norm_i4(int8_t* out_q, qspec_half* scaling, qspec_half* hidden_out, const qspec_half* x,
                          const qspec_half* delta, float eps, int tokens, int hidden, void* stream) {
    const char* op = "qspec_add_rms_norm_i4";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, out_q); NONNULL(op, x); NONNULL(op, scaling);
    if (delta) NONNULL(op, hidden_out);
    if (hidden % 1024 || hidden > 8192 || hidden <= 0) return fail("%s: hidden=%d must be a multiple of 1024, <= 8192", op, hidden);
    return finish(op, qspec::ln_quant_i4(CH(x), CH(delta), H(hidden_out), out_q, H(scaling), nullptr, eps, tokens, hidden, ST));
}
int qspec_add_rms_norm_fp16(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const qspec_half* delta,
                            float eps, int tokens, int hidden, void* stream) {
    const char* op = "qspec_add_rms_norm_fp16";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, out); NONNULL(op, x);
    if (delta) NONNULL(op, hidden_out);
    if (hidden % 1024 || hidden > 8192 || hidden <= 0) return fail("%s: hidden=%d must be a multiple of 1024, <= 8192", op, hidden);
    if (g_xp && tokens > 32) return fail("%s: the fragment-major layout is one or two 16-row tiles (tokens=%d)", op, tokens);
    return finish(op, qspec::ln_fp16(CH(x), CH(delta), H(hidden_out), H(out), eps, tokens, hidden, ST, g_xp));
}
int qspec_add_rms_norm_fp16_xp(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const qspec_half* delta,
                               float eps, int tokens, int hidden, void* stream) {
    XpScope xp;
    return qspec_add_rms_norm_fp16(out, hidden_out, x, delta, eps, tokens, hidden, stream);
}
int qspec_fuse_sym_quant(const qspec_half* x, qspec_half* scale, int8_t* q, float clip_ratio, int tokens, int k,
                         void* stream) {
    const char* op = "qspec_fuse_sym_quant";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, x); NONNULL(op, scale); NONNULL(op, q);
    if (k <= 0 || k % 2) return fail("%s: k=%d must be positive and even", op, k);
    return finish(op, qspec::rowabsmax_quant(CH(x), H(scale), q, clip_ratio, tokens, k, ST));
}
int qspec_fast_hadamard_transform(const qspec_half* x, float scale, qspec_half* out, int64_t rows, int n,
                                  void* stream) {
    const char* op = "qspec_fast_hadamard_transform";
    if (rows < 0) return fail("%s: rows < 0", op);
    if (rows == 0) return 0;
    NONNULL(op, x); NONNULL(op, out);
    if (n < 1 || n > 32768 || (n & (n - 1))) return fail("%s: n=%d must be a power of two in [1, 32768]", op, n);
    if (rows > 2147483647LL) return fail("%s: too many rows", op);
    return finish(op, qspec::fwht(CH(x), scale, H(out), rows, n, ST));
}
int qspec_hadamard_mix(const qspec_half* y, const qspec_half* hadK, qspec_half* out, int tokens, int K, int m,
                       void* stream) {
    const char* op = "qspec_hadamard_mix";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, y); NONNULL(op, hadK); NONNULL(op, out);
    if (y == out) return fail("%s: in-place not supported", op);
    return finish(op, qspec::hadk_mix(CH(y), CH(hadK), H(out), tokens, K, m, ST));
}
int qspec_heads_hadamard(const qspec_half* attn, qspec_half* out_f16, int8_t* q, qspec_half* scale, float had_scale,
                         float clip_ratio, int tokens, int heads, int head_dim, void* stream) {
    const char* op = "qspec_heads_hadamard";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, attn);
    if (q) { NONNULL(op, scale); } else { NONNULL(op, out_f16); }
    return finish(op, qspec::heads_hadamard(CH(attn), H(out_f16), q, H(scale), had_scale, clip_ratio, tokens, heads, head_dim, ST));
}
int qspec_heads_hadamard_mix(const qspec_half* attn, const qspec_half* hadK, qspec_half* out, float had_scale, int tokens,
                             int heads, int head_dim, int K, void* stream) {
    const char* op = "qspec_heads_hadamard_mix";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, attn); NONNULL(op, hadK); NONNULL(op, out);
    if (attn == out) return fail("%s: in-place not supported", op);
    return finish(op, qspec::heads_hadamard_mix(CH(attn), CH(hadK), H(out), had_scale, tokens, heads, head_dim, K, ST));
}
int qspec_silu_mul(const qspec_half* gate_up, qspec_half* out, int tokens, int intermediate, void* stream) {
    const char* op = "qspec_silu_mul";
    if (tokens == 0) return 0;
    NONNULL(op, gate_up); NONNULL(op, out);
    return finish(op, qspec::silu_mul(CH(gate_up), H(out), tokens, intermediate, ST));
}
int qspec_silu_mul_hadamard(const qspec_half* gate_up, const qspec_half* hadK, qspec_half* out_f16, int8_t* q,
                            qspec_half* scale, float had_scale, float clip_ratio, int tokens, int intermediate, int K,
                            void* stream) {
    const char* op = "qspec_silu_mul_hadamard";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, gate_up);
    if (K > 1) NONNULL(op, hadK);
    if (q) { NONNULL(op, scale); } else { NONNULL(op, out_f16); }
    if (intermediate % 8) return fail("%s: intermediate %% 8 != 0", op);
    return finish(op, qspec::silu_mul_hadamard(CH(gate_up), CH(hadK), H(out_f16), q, H(scale), had_scale, clip_ratio, tokens, intermediate, K, 0, nullptr, ST));
}
int qspec_rowwise_scaled_linear_s4s4(const int8_t* xq, const qspec_half* xs, const int8_t* wq, const qspec_half* ws,
                                     const qspec_half* bias, qspec_half* out, int M, int N, int K, void* stream) {
    const char* op = "qspec_rowwise_scaled_linear_s4s4";
    if (M < 0 || N < 0) return fail("%s: negative size", op);
    if (M == 0 || N == 0) return 0;
    NONNULL(op, xq); NONNULL(op, xs); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, out);
    if (N % 16 || K % 128 || K <= 0) return fail("%s: need N %% 16 == 0 and K %% 128 == 0 (N=%d K=%d)", op, N, K);
    if (!bias && use_stream() && qspec::gemm_w4a4_stream_supported(M, N, K, false)) {
        qspec::StreamActs x;
        x.xq = xq; x.xs = CH(xs);
        return finish(op, qspec::gemm_w4a4_stream(x, wq, CH(ws), H(out), M, N, K, ST));
    }
    return finish(op, qspec::gemm_w4a4(xq, CH(xs), wq, CH(ws), CH(bias), H(out), M, N, K, ST));
}
int qspec_rowwise_scaled_linear_s4s4_residual(const int8_t* xq, const qspec_half* xs, const int8_t* wq,
                                              const qspec_half* ws, const qspec_half* resid_in, qspec_half* resid_out,
                                              int M, int N, int K, void* stream) {
    const char* op = "qspec_rowwise_scaled_linear_s4s4_residual";
    if (M < 0 || N < 0) return fail("%s: negative size", op);
    if (M == 0 || N == 0) return 0;
    NONNULL(op, xq); NONNULL(op, xs); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, resid_in); NONNULL(op, resid_out);
    if (!qspec::gemm_w4a4_stream_supported(M, N, K, false))
        return fail("%s: need M <= 32, N %% 16 == 0 and a built K (M=%d N=%d K=%d)", op, M, N, K);
    qspec::StreamActs x;
    x.xq = xq; x.xs = CH(xs);
    return finish(op, qspec::gemm_w4a4_stream_residual(x, wq, CH(ws), CH(resid_in), H(resid_out), M, N, K, ST));
}
int qspec_rowwise_scaled_linear_s4s4_residual_supported(int M, int N, int K) {
    return qspec::gemm_w4a4_stream_supported(M, N, K, false) ? 1 : 0;
}
int qspec_rowwise_scaled_linear_s4s4_residual_hq_supported(int M, int N, int K, int n_parts) {
    return qspec::gemm_w4a4_stream_residual_hq_supported(M, N, K, n_parts) ? 1 : 0;
}
int qspec_rowwise_scaled_linear_s4s4_residual_hq(const qspec_half* x16, const float* part_amax, int n_parts,
                                                 float clip_ratio, const int8_t* wq, const qspec_half* ws,
                                                 const qspec_half* resid_in, qspec_half* resid_out, int M, int N, int K,
                                                 void* stream) {
    const char* op = "qspec_rowwise_scaled_linear_s4s4_residual_hq";
    if (M < 0 || N < 0) return fail("%s: negative size", op);
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x16); NONNULL(op, part_amax); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, resid_in); NONNULL(op, resid_out);
    if (!qspec::gemm_w4a4_stream_residual_hq_supported(M, N, K, n_parts))
        return fail("%s: need M <= 4, K = 4096, n_parts = 8, N %% 16 == 0 (M=%d N=%d K=%d n_parts=%d)", op, M, N, K, n_parts);
    return finish(op, qspec::gemm_w4a4_stream_residual_hq(CH(x16), part_amax, n_parts, clip_ratio, wq, CH(ws), CH(resid_in),
                                                          H(resid_out), M, N, K, ST));
}
size_t qspec_w4a16_workspace_bytes(void) { return qspec::gemm_w4a16_ws_bytes(); }
int qspec_w4a16_linear(const qspec_half* x, const int8_t* wq, const qspec_half* ws, const qspec_half* bias,
                       qspec_half* out, int M, int N, int K, void* workspace, void* stream) {
    const char* op = "qspec_w4a16_linear";
    if (M < 0 || N < 0) return fail("%s: negative size", op);
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, out);
    if (N % 16 || K % 128 || K <= 0) return fail("%s: need N %% 16 == 0 and K %% 128 == 0 (N=%d K=%d)", op, N, K);
    if (g_xp && !(!bias && use_stream() && qspec::gemm_w4a16_stream_supported(M, N, K) && qspec::gemm_w4a16_xperm_supported(M, K)))
        return fail("%s: no fragment-major form for (M=%d N=%d K=%d): ask qspec_w4a16_act_layout_supported", op, M, N, K);
    if (!bias && use_stream() && qspec::gemm_w4a16_stream_supported(M, N, K))
        return finish(op, qspec::gemm_w4a16_stream(CH(x), 0, wq, 0, CH(ws), H(out), M, N, K, ST, g_xp));
    // long K (down_proj): K slices of a built length, raw sums in the workspace (behind its ticket counters), finish
    const int S = (!bias && use_stream() && workspace) ? qspec::gemm_w4a16_stream_partial_slices(M, N, K) : 0;
    if (S > 0 && (size_t)S * M * N * sizeof(float) + 8192 <= qspec::gemm_w4a16_ws_bytes()) {
        float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 8192);
        int rc = qspec::gemm_w4a16_stream_partial(CH(x), 0, wq, 0, part, M, N, K, S, ST);
        if (rc == 0) rc = qspec::gemm_w4a16_partial_finish(part, CH(ws), H(out), M, N, S, ST);
        return finish(op, rc);
    }
    if (!bias && M >= tiled_min_m() && qspec::gemm_w4a16_tiled_supported(M, N, K))   // prefill-sized M: M-tiled kernel
        return finish(op, qspec::gemm_w4a16_tiled(CH(x), wq, CH(ws), H(out), M, N, K,
                                                  workspace ? reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 8192) : nullptr,
                                                  workspace ? qspec::gemm_w4a16_ws_bytes() - 8192 : 0, ST));
    return finish(op, qspec::gemm_w4a16(CH(x), wq, CH(ws), CH(bias), H(out), M, N, K, workspace, ST));
}
int qspec_w4a16_act_layout32_supported(int M, int N, int K) { return use_stream() && qspec::gemm_w4a16_stream32_supported(M, N, K) ? 1 : 0; }
int qspec_w4a16_linear_xp32(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* out, int M, int N, int K,
                            void* stream) {
    const char* op = "qspec_w4a16_linear_xp32";
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, out);
    if (!qspec_w4a16_act_layout32_supported(M, N, K))
        return fail("%s: no two-tile streaming form for (M=%d N=%d K=%d): ask qspec_w4a16_act_layout32_supported", op, M, N, K);
    return finish(op, qspec::gemm_w4a16_stream32(CH(x), wq, CH(ws), H(out), M, N, K, ST));
}
int qspec_w4a16_linear_partial_slices_xp32(int M, int N, int K) { return use_stream() ? qspec::gemm_w4a16_stream32_partial_slices(M, N, K) : 0; }
int qspec_w4a16_linear_partial_xp32(const qspec_half* x, const int8_t* wq, float* part, int M, int N, int K, int slices, void* stream) {
    const char* op = "qspec_w4a16_linear_partial_xp32";
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, part);
    if (slices < 2 || slices != qspec_w4a16_linear_partial_slices_xp32(M, N, K))
        return fail("%s: (M=%d N=%d K=%d) takes %d slices (qspec_w4a16_linear_partial_slices_xp32), got %d", op, M, N, K,
                    qspec_w4a16_linear_partial_slices_xp32(M, N, K), slices);
    return finish(op, qspec::gemm_w4a16_stream32_partial(CH(x), wq, part, M, N, K, slices, ST));
}
int qspec_qkv_rope_linear_w4a16_xp32(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M,
                                     int N, int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                     qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                     int num_heads, int num_kv_heads, int head_size, int rot_dim, void* stream) {
    const char* op = "qspec_qkv_rope_linear_w4a16_xp32";
    if (M == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, qkv); NONNULL(op, positions);
    NONNULL(op, cos_sin_cache); NONNULL(op, key_cache); NONNULL(op, value_cache); NONNULL(op, slot_mapping);
    if ((head_size != 128 && head_size != 64) || rot_dim != head_size) return fail("%s: head_size = rot_dim = 128 or 64 only", op);
    if (!qspec_w4a16_act_layout32_supported(M, N, K))
        return fail("%s: no two-tile streaming form for (M=%d N=%d K=%d)", op, M, N, K);
    return finish(op, qspec::gemm_w4a16_stream32_qkv_rope(CH(x), wq, CH(ws), H(qkv), M, N, K, positions, CH(cos_sin_cache), H(key_cache), H(value_cache), slot_mapping, num_heads, num_kv_heads, head_size, rot_dim, ST));
}
int qspec_gate_up_silu_linear_w4a16_xp32(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                         int intermediate, int K, void* stream) {
    const char* op = "qspec_gate_up_silu_linear_w4a16_xp32";
    if (M == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, act);
    if (intermediate % 8 || !qspec_w4a16_act_layout32_supported(M, 2 * intermediate, K))
        return fail("%s: no two-tile streaming form for (M=%d I=%d K=%d)", op, M, intermediate, K);
    return finish(op, qspec::gemm_w4a16_stream32_gate_up_silu(CH(x), wq, CH(ws), H(act), M, intermediate, K, ST));
}
int qspec_mlp_hadamard_act_layout_supported(int tokens, int intermediate, int K) {
    return qspec::mlp_hadamard_xperm_supported(tokens, intermediate, K) ? 1 : 0;
}
int qspec_w4a16_act_layout_supported(int M, int K) { return use_stream() && qspec::gemm_w4a16_xperm_supported(M, K) ? 1 : 0; }
int qspec_w4a16_linear_xp(const qspec_half* x, const int8_t* wq, const qspec_half* ws, const qspec_half* bias,
                          qspec_half* out, int M, int N, int K, void* workspace, void* stream) {
    XpScope xp;
    return qspec_w4a16_linear(x, wq, ws, bias, out, M, N, K, workspace, stream);
}
int qspec_linear_f16(const qspec_half* x, const qspec_half* w, qspec_half* out, int M, int N, int K, void* stream) {
    const char* op = "qspec_linear_f16";
    if (M < 0 || N < 0) return fail("%s: negative size", op);
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x); NONNULL(op, w); NONNULL(op, out);
    if (K % 32 || K <= 0) return fail("%s: need K %% 32 == 0 (K=%d)", op, K);
    if (use_stream() && qspec::gemm_f16_stream_supported(M, N, K))
        return finish(op, qspec::gemm_f16_stream(CH(x), CH(w), H(out), M, N, K, nullptr, ST));
    if (M > 16 && tiled_min_m() < (1 << 30) && qspec::gemm_f16_tiled_supported(M, N, K))   // prefill / large-batch logits
        return finish(op, qspec::gemm_f16_tiled(CH(x), CH(w), H(out), M, N, K, ST));
    return finish(op, qspec::gemm_f16(CH(x), CH(w), H(out), M, N, K, ST));
}
int qspec_dequant_w4(const int8_t* wq, const qspec_half* ws, qspec_half* out, int N, int K, void* stream) {
    const char* op = "qspec_dequant_w4";
    if (N == 0) return 0;
    NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, out);
    return finish(op, qspec::dequant_w4(wq, CH(ws), H(out), N, K, ST));
}
int qspec_rotary_embedding(const int64_t* positions, qspec_half* q, qspec_half* k, const qspec_half* cos_sin_cache,
                           int tokens, int num_heads, int num_kv_heads, int head_size, int rot_dim, int64_t q_stride,
                           int64_t k_stride, void* stream) {
    const char* op = "qspec_rotary_embedding";
    if (tokens == 0) return 0;
    NONNULL(op, positions); NONNULL(op, q); NONNULL(op, k); NONNULL(op, cos_sin_cache);
    return finish(op, qspec::rotary_embedding(positions, H(q), H(k), CH(cos_sin_cache), tokens, num_heads, num_kv_heads, head_size, rot_dim, q_stride, k_stride, ST));
}
int qspec_reshape_and_cache_flash(const qspec_half* key, const qspec_half* value, qspec_half* key_cache,
                                  qspec_half* value_cache, const int64_t* slot_mapping, int tokens, int num_kv_heads,
                                  int head_size, int64_t key_stride, int64_t value_stride, void* stream) {
    const char* op = "qspec_reshape_and_cache_flash";
    if (tokens == 0) return 0;
    NONNULL(op, key); NONNULL(op, value); NONNULL(op, key_cache); NONNULL(op, value_cache); NONNULL(op, slot_mapping);
    return finish(op, qspec::reshape_and_cache_flash(CH(key), CH(value), H(key_cache), H(value_cache), slot_mapping, tokens, num_kv_heads, head_size, key_stride, value_stride, ST));
}
int qspec_rope_kv_write(const int64_t* positions, qspec_half* qkv, const qspec_half* cos_sin_cache,
                        qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping, int tokens,
                        int num_heads, int num_kv_heads, int head_size, int rot_dim, void* stream) {
    const char* op = "qspec_rope_kv_write";
    if (tokens == 0) return 0;
    NONNULL(op, positions); NONNULL(op, qkv); NONNULL(op, cos_sin_cache); NONNULL(op, key_cache); NONNULL(op, value_cache); NONNULL(op, slot_mapping);
    return finish(op, qspec::rope_kv_write(positions, H(qkv), CH(cos_sin_cache), H(key_cache), H(value_cache), slot_mapping, tokens, num_heads, num_kv_heads, head_size, rot_dim, ST));
}
size_t qspec_paged_attention_workspace_bytes(int max_tokens, int num_heads, int head_size, int n_splits) {
    return qspec::paged_attention_ws_bytes(max_tokens, num_heads, head_size, n_splits);
}
int qspec_paged_attention(const qspec_half* q, int64_t q_stride, const qspec_half* key_cache,
                          const qspec_half* value_cache, const int32_t* block_tables, int max_blocks_per_seq,
                          const int32_t* ctx_lens, const int32_t* q_start, int n_seqs, int tokens, int max_q_len,
                          int num_heads, int num_kv_heads, int head_size, int block_size, float sm_scale, int n_splits,
                          void* workspace, qspec_half* out, void* stream) {
    const char* op = "qspec_paged_attention";
    if (n_seqs == 0 || tokens == 0) return 0;
    NONNULL(op, q); NONNULL(op, key_cache); NONNULL(op, value_cache); NONNULL(op, block_tables); NONNULL(op, ctx_lens);
    NONNULL(op, q_start); NONNULL(op, workspace);   /* out == NULL: partials only, see qspec_heads_hadamard_merged */
    if (head_size != 128 && (head_size > 256 || head_size % 2 || (!out && head_size % 8)))
        return fail("%s: head_size=%d: the matrix-core kernel is built for 128; other even sizes <= 256 run the generic kernel (partials, out = NULL: multiples of 8)", op, head_size);
    if (tokens > n_seqs * max_q_len) return fail("%s: tokens=%d > n_seqs*max_q_len=%d", op, tokens, n_seqs * max_q_len);
    int rc = qspec::paged_attention(CH(q), q_stride, CH(key_cache), CH(value_cache), block_tables, max_blocks_per_seq, ctx_lens, q_start, n_seqs, max_q_len, num_heads, num_kv_heads, head_size, block_size, sm_scale, n_splits, (float*)workspace, H(out), ST);
    return finish(op, rc);
}
int qspec_heads_hadamard_merged(const void* attn_workspace, int max_tokens, int n_splits, qspec_half* out_f16, int8_t* q,
                                qspec_half* scale, float had_scale, float clip_ratio, int tokens, int heads,
                                int head_dim, void* stream) {
    const char* op = "qspec_heads_hadamard_merged";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, attn_workspace);
    if (q) { NONNULL(op, scale); } else { NONNULL(op, out_f16); }
    if (!((head_dim == 128 && (heads == 32 || heads == 64)) || (heads == 32 && head_dim % 8 == 0 && head_dim >= 8 && head_dim <= 256)))
        return fail("%s: built for 32 / 64 heads of 128 and 32 heads of another size %% 8 == 0 (got %d x %d)", op, heads, head_dim);
    if (g_xp && (q || !(heads == 32 || heads == 64) || tokens > 32 || head_dim != 128))
        return fail("%s: the fragment-major fp16 output exists for 32 / 64 heads of 128, <= 32 tokens, no quantiser", op);
    return finish(op, qspec::heads_hadamard_merge((const float*)attn_workspace, max_tokens, n_splits, H(out_f16), q, H(scale), had_scale, clip_ratio, tokens, heads, head_dim, ST, g_xp));
}
int qspec_heads_hadamard_merged_xp(const void* attn_workspace, int max_tokens, int n_splits, qspec_half* out_f16, int8_t* q,
                                   qspec_half* scale, float had_scale, float clip_ratio, int tokens, int heads,
                                   int head_dim, void* stream) {
    XpScope xp;
    return qspec_heads_hadamard_merged(attn_workspace, max_tokens, n_splits, out_f16, q, scale, had_scale, clip_ratio, tokens, heads,
                                       head_dim, stream);
}
int qspec_heads_hadamard_merged_spread_supported(int tokens, int heads, int head_dim) {
    return tokens >= 0 && tokens * 8 <= 1024 && heads == 32 && head_dim == 128;
}
int qspec_heads_hadamard_merged_spread(const void* attn_workspace, int max_tokens, int n_splits, qspec_half* out_f16,
                                       float* part_amax, float had_scale, int tokens, int heads, int head_dim,
                                       void* stream) {
    const char* op = "qspec_heads_hadamard_merged_spread";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, attn_workspace); NONNULL(op, out_f16); NONNULL(op, part_amax);
    if (!qspec_heads_hadamard_merged_spread_supported(tokens, heads, head_dim))
        return fail("%s: built for 32 heads of 128 and at most 128 tokens (got %d x %d, %d tokens)", op, heads, head_dim, tokens);
    return finish(op, qspec::heads_hadamard_merge_spread((const float*)attn_workspace, max_tokens, n_splits, H(out_f16),
                                                         part_amax, had_scale, tokens, heads, head_dim, ST));
}
int qspec_heads_hadamard_mix_merged_spread_supported(int tokens, int heads, int head_dim, int K) {
    return qspec::heads_hadamard_mix_merge_spread_supported(tokens, heads, head_dim, K) ? 1 : 0;
}
int qspec_heads_hadamard_mix_merged_spread(const void* attn_workspace, int max_tokens, int n_splits, const qspec_half* hadK,
                                           int K, qspec_half* out_f16, float* part_amax, float had_scale, int tokens,
                                           int heads, int head_dim, void* stream) {
    const char* op = "qspec_heads_hadamard_mix_merged_spread";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, attn_workspace); NONNULL(op, hadK); NONNULL(op, out_f16);
    if (!qspec::heads_hadamard_mix_merge_spread_supported(tokens, heads, head_dim, K))
        return fail("%s: needs head_dim 128, heads = K * 2^p <= 64 with 2 <= K <= 172, at most 128 tokens (got %d x %d, K = %d, %d tokens)",
                    op, heads, head_dim, K, tokens);
    if (g_xp && (part_amax || tokens > 32)) return fail("%s: the fragment-major fp16 output: <= 32 tokens, no part_amax", op);
    return finish(op, qspec::heads_hadamard_mix_merge_spread((const float*)attn_workspace, max_tokens, n_splits, CH(hadK), H(out_f16),
                                                             part_amax, had_scale, tokens, heads, head_dim, K, ST, g_xp));
}
int qspec_heads_hadamard_mix_merged_spread_xp(const void* attn_workspace, int max_tokens, int n_splits, const qspec_half* hadK,
                                              int K, qspec_half* out_f16, float* part_amax, float had_scale, int tokens,
                                              int heads, int head_dim, void* stream) {
    XpScope xp;
    return qspec_heads_hadamard_mix_merged_spread(attn_workspace, max_tokens, n_splits, hadK, K, out_f16, part_amax, had_scale, tokens,
                                                  heads, head_dim, stream);
}
int qspec_embedding(const int64_t* ids, const qspec_half* table, qspec_half* out, int tokens, int hidden, int vocab,
                    void* stream) {
    const char* op = "qspec_embedding";
    if (tokens == 0) return 0;
    NONNULL(op, ids); NONNULL(op, table); NONNULL(op, out);
    return finish(op, qspec::embedding(ids, CH(table), H(out), tokens, hidden, vocab, ST));
}
size_t qspec_sampler_workspace_bytes(int rows) { return qspec::sampler_ws_bytes(rows); }
int qspec_softmax_argmax(const qspec_half* logits, float* probs, int64_t* token, int tokens, int vocab,
                         void* workspace, void* stream) {
    const char* op = "qspec_softmax_argmax";
    if (tokens == 0) return 0;
    NONNULL(op, logits); NONNULL(op, probs); NONNULL(op, token); NONNULL(op, workspace);
    return finish(op, qspec::softmax_argmax(CH(logits), probs, token, tokens, vocab, workspace, ST));
}
size_t qspec_lm_head_sampler_workspace_bytes(int rows) { return qspec::head_softmax_ws_bytes(rows); }
int qspec_lm_head_softmax_argmax(const qspec_half* hidden, const qspec_half* lm_head, qspec_half* logits, float* probs,
                                 int64_t* token, int tokens, int vocab, int K, void* workspace, void* stream) {
    const char* op = "qspec_lm_head_softmax_argmax";
    if (tokens == 0) return 0;
    NONNULL(op, hidden); NONNULL(op, lm_head); NONNULL(op, logits); NONNULL(op, probs); NONNULL(op, token);
    NONNULL(op, workspace);
    if (!(use_stream() && qspec::gemm_f16_stream_supported(tokens, vocab, K)))
        return fail("%s: (tokens=%d vocab=%d K=%d) is not a streaming lm_head shape (tokens <= 16, vocab %% 16 == 0); "
                    "use qspec_linear_f16 + qspec_softmax_argmax", op, tokens, vocab, K);
    int rc = qspec::gemm_f16_stream(CH(hidden), CH(lm_head), H(logits), tokens, vocab, K,
                                    qspec::head_softmax_part_max(workspace, tokens), ST);
    if (rc == 0)
        rc = qspec::head_softmax_argmax(CH(logits), probs, token, tokens, vocab, qspec::gemm_f16_stream_grid(vocab),
                                        workspace, ST);
    return finish(op, rc);
}
int qspec_lm_head_softmax_argmax_supported(int tokens, int vocab, int K) {
    return use_stream() && qspec::gemm_f16_stream_supported(tokens, vocab, K) ? 1 : 0;
}
int qspec_rejection_sample(const float* target_with_bonus_probs, const int64_t* bonus_token_ids,
                           const float* draft_probs, const int64_t* draft_token_ids, const float* uniform,
                           const float* exponential, uint64_t seed, uint64_t offset, uint64_t* rng_state, int batch,
                           int k, int vocab, int64_t dp_stride_b, int64_t dp_stride_k, int64_t ids_stride_b,
                           int64_t ids_stride_k, int64_t bonus_stride,
                           int64_t* out_tokens, uint8_t* accepted, int64_t* recovered, int64_t* counters,
                           const int32_t* active_lens, void* workspace, void* stream) {
    const char* op = "qspec_rejection_sample";
    if (batch == 0) return 0;
    NONNULL(op, target_with_bonus_probs); NONNULL(op, bonus_token_ids); NONNULL(op, draft_probs);
    NONNULL(op, draft_token_ids); NONNULL(op, out_tokens); NONNULL(op, accepted); NONNULL(op, recovered);
    NONNULL(op, workspace);
    if (batch * k > 4096) return fail("%s: batch*k=%d too large", op, batch * k);
    if (k < 1) return fail("%s: k=%d must be >= 1", op, k);
    return finish(op, qspec::rejection_sample(target_with_bonus_probs, draft_probs, draft_token_ids, bonus_token_ids, uniform, exponential, seed, offset, rng_state, batch, k, vocab, dp_stride_b, dp_stride_k, ids_stride_b, ids_stride_k, bonus_stride, out_tokens, accepted, recovered, counters, active_lens, workspace, ST));
}
size_t qspec_sample_workspace_bytes(int rows) { return qspec::sample_ws_bytes(rows); }
int qspec_sample_top_k_top_p(const qspec_half* logits, const float* temperature, const int32_t* top_k, const float* top_p,
                             const float* exponential, uint64_t seed, uint64_t offset, uint64_t* rng_state, float* probs,
                             int64_t* token, int64_t token_stride, int tokens, int vocab, void* workspace, void* stream) {
    const char* op = "qspec_sample_top_k_top_p";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, logits); NONNULL(op, probs); NONNULL(op, token); NONNULL(op, workspace);
    if (vocab < 1) return fail("%s: vocab < 1", op);
    return finish(op, qspec::sample_top_k_top_p(CH(logits), temperature, top_k, top_p, exponential, seed, offset, rng_state, probs,
                                                token, token_stride, tokens, vocab, workspace, ST));
}
int qspec_typical_acceptance_sample(const float* target_with_bonus_probs, const int64_t* bonus_token_ids,
                                    const int64_t* draft_token_ids, float posterior_threshold, float posterior_alpha, int batch,
                                    int k, int vocab, int64_t ids_stride_b, int64_t ids_stride_k, int64_t bonus_stride,
                                    int64_t* out_tokens, uint8_t* accepted, int64_t* recovered, int64_t* counters,
                                    const int32_t* active_lens, void* workspace, void* stream) {
    const char* op = "qspec_typical_acceptance_sample";
    if (batch == 0) return 0;
    NONNULL(op, target_with_bonus_probs); NONNULL(op, bonus_token_ids); NONNULL(op, draft_token_ids); NONNULL(op, out_tokens);
    NONNULL(op, accepted); NONNULL(op, recovered); NONNULL(op, workspace);
    if (batch * k > 4096) return fail("%s: batch*k=%d too large", op, batch * k);
    if (k < 1) return fail("%s: k=%d must be >= 1", op, k);
    return finish(op, qspec::typical_acceptance_sample(target_with_bonus_probs, draft_token_ids, bonus_token_ids, posterior_threshold,
                                                       posterior_alpha, batch, k, vocab, ids_stride_b, ids_stride_k, bonus_stride,
                                                       out_tokens, accepted, recovered, counters, active_lens, workspace, ST));
}
int qspec_advance_step_flashattn(int num_seqs, int block_size, int64_t* input_tokens,
                                 const int64_t* sampled_token_ids, int64_t* input_positions, int32_t* seq_lens,
                                 int64_t* slot_mapping, const int32_t* block_tables, int64_t block_tables_stride,
                                 void* stream) {
    const char* op = "qspec_advance_step_flashattn";
    if (num_seqs == 0) return 0;
    NONNULL(op, input_tokens); NONNULL(op, sampled_token_ids); NONNULL(op, input_positions); NONNULL(op, seq_lens);
    NONNULL(op, slot_mapping); NONNULL(op, block_tables);
    return finish(op, qspec::advance_step(num_seqs, block_size, input_tokens, sampled_token_ids, input_positions, seq_lens, slot_mapping, block_tables, block_tables_stride, ST));
}

int qspec_spec_advance_draft(int batch, int block_size, int max_blocks_per_seq, int64_t* input_tokens,
                             const int64_t* sampled_token_ids, int64_t* positions, int32_t* ctx_lens,
                             int64_t* slot_mapping, const int32_t* block_tables, int64_t block_tables_stride,
                             void* stream) {
    const char* op = "qspec_spec_advance_draft";
    if (batch == 0) return 0;
    NONNULL(op, input_tokens); NONNULL(op, sampled_token_ids); NONNULL(op, positions); NONNULL(op, ctx_lens);
    NONNULL(op, slot_mapping); NONNULL(op, block_tables);
    return finish(op, qspec::spec_advance_draft(batch, block_size, max_blocks_per_seq, input_tokens, sampled_token_ids, positions, ctx_lens, slot_mapping, block_tables, block_tables_stride, ST));
}
int qspec_spec_prepare_draft(int batch, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                             const int32_t* seq_lens, const int32_t* block_tables, int64_t block_tables_stride,
                             int64_t* input_tokens, int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens,
                             void* stream) {
    const char* op = "qspec_spec_prepare_draft";
    if (batch == 0) return 0;
    NONNULL(op, last_token); NONNULL(op, seq_lens); NONNULL(op, block_tables); NONNULL(op, input_tokens);
    NONNULL(op, positions); NONNULL(op, slot_mapping); NONNULL(op, ctx_lens);
    return finish(op, qspec::spec_prepare_draft(batch, block_size, max_blocks_per_seq, last_token, seq_lens, block_tables, block_tables_stride, input_tokens, positions, slot_mapping, ctx_lens, ST));
}
int qspec_spec_prepare_verify(int batch, int k, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                              const int64_t* draft_token_ids, int64_t ids_stride_b, int64_t ids_stride_k,
                              const int32_t* seq_lens, const int32_t* block_tables,
                              int64_t block_tables_stride, int64_t* tokens, int64_t* positions, int64_t* slot_mapping,
                              int32_t* ctx_lens, void* stream) {
    const char* op = "qspec_spec_prepare_verify";
    if (batch == 0) return 0;
    NONNULL(op, last_token); NONNULL(op, draft_token_ids); NONNULL(op, seq_lens); NONNULL(op, block_tables);
    NONNULL(op, tokens); NONNULL(op, positions); NONNULL(op, slot_mapping); NONNULL(op, ctx_lens);
    return finish(op, qspec::spec_prepare_verify(batch, k, block_size, max_blocks_per_seq, last_token, draft_token_ids, ids_stride_b, ids_stride_k, seq_lens, block_tables, block_tables_stride, tokens, positions, slot_mapping, ctx_lens, ST));
}
int qspec_spec_prepare_draft_embed(int batch, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                                   const int32_t* seq_lens, const int32_t* step_mask, int32_t* eff_lens,
                                   const int32_t* block_tables, int64_t block_tables_stride, int64_t* input_tokens,
                                   int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens, const qspec_half* embed_tokens,
                                   qspec_half* hidden_out, int hidden, int vocab, void* stream) {
    const char* op = "qspec_spec_prepare_draft_embed";
    if (batch == 0) return 0;
    NONNULL(op, last_token); NONNULL(op, seq_lens); NONNULL(op, block_tables); NONNULL(op, input_tokens);
    NONNULL(op, positions); NONNULL(op, slot_mapping); NONNULL(op, ctx_lens); NONNULL(op, embed_tokens); NONNULL(op, hidden_out);
    if (step_mask && !eff_lens) return fail("%s: step_mask needs eff_lens", op);
    if (hidden % 8 || hidden <= 0) return fail("%s: hidden=%d must be a positive multiple of 8", op, hidden);
    return finish(op, qspec::spec_prepare_draft_embed(batch, block_size, max_blocks_per_seq, last_token, seq_lens, step_mask, eff_lens, block_tables, block_tables_stride, input_tokens, positions, slot_mapping, ctx_lens, CH(embed_tokens), H(hidden_out), hidden, vocab, ST));
}
int qspec_spec_advance_draft_embed(int batch, int block_size, int max_blocks_per_seq, int64_t* input_tokens,
                                   const int64_t* sampled_token_ids, int64_t* positions, int32_t* ctx_lens,
                                   int64_t* slot_mapping, const int32_t* block_tables, int64_t block_tables_stride,
                                   const qspec_half* embed_tokens, qspec_half* hidden_out, int hidden, int vocab, void* stream) {
    const char* op = "qspec_spec_advance_draft_embed";
    if (batch == 0) return 0;
    NONNULL(op, input_tokens); NONNULL(op, sampled_token_ids); NONNULL(op, positions); NONNULL(op, ctx_lens);
    NONNULL(op, slot_mapping); NONNULL(op, block_tables); NONNULL(op, embed_tokens); NONNULL(op, hidden_out);
    if (hidden % 8 || hidden <= 0) return fail("%s: hidden=%d must be a positive multiple of 8", op, hidden);
    return finish(op, qspec::spec_advance_draft_embed(batch, block_size, max_blocks_per_seq, input_tokens, sampled_token_ids, positions, ctx_lens, slot_mapping, block_tables, block_tables_stride, CH(embed_tokens), H(hidden_out), hidden, vocab, ST));
}
int qspec_spec_prepare_verify_embed(int batch, int k, int block_size, int max_blocks_per_seq, const int64_t* last_token,
                                    const int64_t* draft_token_ids, int64_t ids_stride_b, int64_t ids_stride_k,
                                    const int32_t* seq_lens, const int32_t* block_tables, int64_t block_tables_stride,
                                    int64_t* tokens, int64_t* positions, int64_t* slot_mapping, int32_t* ctx_lens,
                                    const qspec_half* embed_tokens, qspec_half* hidden_out, int hidden, int vocab, void* stream) {
    const char* op = "qspec_spec_prepare_verify_embed";
    if (batch == 0) return 0;
    NONNULL(op, last_token); NONNULL(op, draft_token_ids); NONNULL(op, seq_lens); NONNULL(op, block_tables);
    NONNULL(op, tokens); NONNULL(op, positions); NONNULL(op, slot_mapping); NONNULL(op, ctx_lens); NONNULL(op, embed_tokens);
    NONNULL(op, hidden_out);
    if (hidden % 8 || hidden <= 0) return fail("%s: hidden=%d must be a positive multiple of 8", op, hidden);
    return finish(op, qspec::spec_prepare_verify_embed(batch, k, block_size, max_blocks_per_seq, last_token, draft_token_ids, ids_stride_b, ids_stride_k, seq_lens, block_tables, block_tables_stride, tokens, positions, slot_mapping, ctx_lens, CH(embed_tokens), H(hidden_out), hidden, vocab, ST));
}
int qspec_spec_commit(int batch, int k, const int64_t* out_tokens, int32_t* seq_lens, int64_t* last_token,
                      int64_t* gen_tokens, int32_t* gen_lens, int gen_capacity, void* stream) {
    const char* op = "qspec_spec_commit";
    if (batch == 0) return 0;
    NONNULL(op, out_tokens); NONNULL(op, seq_lens); NONNULL(op, last_token);
    if (gen_tokens) NONNULL(op, gen_lens);
    return finish(op, qspec::spec_commit(batch, k, out_tokens, seq_lens, last_token, gen_tokens, gen_lens, gen_capacity, ST));
}

int qspec_spec_snapshot(int batch, int restore, int32_t* seq_lens, int32_t* gen_lens, int64_t* last_token,
                        int64_t* counters, int64_t* rng_state, int32_t* snap_i32, int64_t* snap_i64, void* stream) {
    const char* op = "qspec_spec_snapshot";
    if (batch < 0) return fail("%s: batch < 0", op);
    NONNULL(op, seq_lens); NONNULL(op, gen_lens); NONNULL(op, last_token); NONNULL(op, counters); NONNULL(op, rng_state);
    NONNULL(op, snap_i32); NONNULL(op, snap_i64);
    return finish(op, qspec::spec_snapshot(batch, restore, seq_lens, gen_lens, last_token, counters, rng_state, snap_i32, snap_i64, ST));
}
int qspec_collect_error_words(int32_t* w0, int32_t* w1, int32_t* w2, int32_t* w3, int clear, int64_t* out, void* stream) {
    const char* op = "qspec_collect_error_words";
    if (!out && !clear) return fail("%s: nothing to do (out == NULL and clear == 0)", op);
    return finish(op, qspec::collect_error_words(w0, w1, w2, w3, clear, out, ST));
}

size_t qspec_xwg_workspace_bytes(void) { return qspec::xwg_workspace_bytes(); }
int qspec_mlp_hadamard(const qspec_half* act, const qspec_half* hadK, qspec_half* out_f16, int8_t* q, qspec_half* scale,
                       float had_scale, float clip_ratio, int tokens, int intermediate, int K, void* workspace,
                       void* stream) {
    const char* op = "qspec_mlp_hadamard";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, act);
    if (K > 1) NONNULL(op, hadK);
    if (q) { NONNULL(op, scale); } else { NONNULL(op, out_f16); }
    if (intermediate % 8) return fail("%s: intermediate %% 8 != 0", op);
    if (g_xp && (q || tokens > 32))
        return fail("%s: the fragment-major fp16 output: <= 32 tokens, no quantiser", op);
    return finish(op, qspec::silu_mul_hadamard(CH(act), CH(hadK), H(out_f16), q, H(scale), had_scale, clip_ratio, tokens, intermediate, K, 1, workspace, ST, g_xp));
}
int qspec_mlp_hadamard_xp(const qspec_half* act, const qspec_half* hadK, qspec_half* out_f16, int8_t* q, qspec_half* scale,
                          float had_scale, float clip_ratio, int tokens, int intermediate, int K, void* workspace,
                          void* stream) {
    XpScope xp;
    return qspec_mlp_hadamard(act, hadK, out_f16, q, scale, had_scale, clip_ratio, tokens, intermediate, K, workspace, stream);
}
int qspec_qkv_rope_linear_s4s4(const int8_t* xq, const qspec_half* xs, const int8_t* wq, const qspec_half* ws,
                               qspec_half* qkv, int M, int N, int K, const int64_t* positions,
                               const qspec_half* cos_sin_cache, qspec_half* key_cache, qspec_half* value_cache,
                               const int64_t* slot_mapping, int num_heads, int num_kv_heads, int head_size,
                               int rot_dim, void* stream) {
    const char* op = "qspec_qkv_rope_linear_s4s4";
    if (M == 0) return 0;
    NONNULL(op, xq); NONNULL(op, xs); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, qkv); NONNULL(op, positions);
    NONNULL(op, cos_sin_cache); NONNULL(op, key_cache); NONNULL(op, value_cache); NONNULL(op, slot_mapping);
    if ((head_size != 128 && head_size != 64) || rot_dim != head_size) return fail("%s: head_size = rot_dim = 128 or 64 only", op);
    if (head_size == 64 && !(use_stream() && qspec::gemm_w4a4_stream_supported(M, N, K, false)))
        return fail("%s: head_size 64 exists on the streaming kernel only (M=%d N=%d K=%d): ask qspec_qkv_rope_linear_supported", op, M, N, K);
    if (use_stream() && qspec::gemm_w4a4_stream_supported(M, N, K, false)) {
        qspec::StreamActs x;
        x.xq = xq; x.xs = CH(xs);
        return finish(op, qspec::gemm_w4a4_stream_qkv_rope(x, wq, CH(ws), H(qkv), M, N, K, positions, CH(cos_sin_cache), H(key_cache), H(value_cache), slot_mapping, num_heads, num_kv_heads, head_size, rot_dim, ST));
    }
    return finish(op, qspec::gemm_w4a4_qkv_rope(xq, CH(xs), wq, CH(ws), H(qkv), M, N, K, positions, CH(cos_sin_cache), H(key_cache), H(value_cache), slot_mapping, num_heads, num_kv_heads, head_size, rot_dim, ST));
}
int qspec_qkv_rope_linear_supported(int w4a4, int M, int N, int K, int head_size) {
    if (head_size == 128) return 1;   // every shape: the streaming kernels, else the first-generation fused kernels
    if (head_size != 64 || !use_stream()) return 0;
    return (w4a4 ? qspec::gemm_w4a4_stream_supported(M, N, K, false) : qspec::gemm_w4a16_stream_supported(M, N, K)) ? 1 : 0;
}
int qspec_qkv_rope_linear_w4a16(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M,
                                int N, int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                int num_heads, int num_kv_heads, int head_size, int rot_dim, void* workspace,
                                void* stream) {
    const char* op = "qspec_qkv_rope_linear_w4a16";
    if (M == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, qkv); NONNULL(op, positions);
    NONNULL(op, cos_sin_cache); NONNULL(op, key_cache); NONNULL(op, value_cache); NONNULL(op, slot_mapping);
    if ((head_size != 128 && head_size != 64) || rot_dim != head_size) return fail("%s: head_size = rot_dim = 128 or 64 only", op);
    if (head_size == 64 && !(use_stream() && qspec::gemm_w4a16_stream_supported(M, N, K)))
        return fail("%s: head_size 64 exists on the streaming kernel only (M=%d N=%d K=%d): ask qspec_qkv_rope_linear_supported", op, M, N, K);
    if (g_xp && !(use_stream() && qspec::gemm_w4a16_stream_supported(M, N, K) && qspec::gemm_w4a16_xperm_supported(M, K)))
        return fail("%s: no fragment-major form for (M=%d N=%d K=%d)", op, M, N, K);
    if (use_stream() && qspec::gemm_w4a16_stream_supported(M, N, K))
        return finish(op, qspec::gemm_w4a16_stream_qkv_rope(CH(x), wq, CH(ws), H(qkv), M, N, K, positions, CH(cos_sin_cache), H(key_cache), H(value_cache), slot_mapping, num_heads, num_kv_heads, head_size, rot_dim, ST, g_xp));
    return finish(op, qspec::gemm_w4a16_qkv_rope(CH(x), wq, CH(ws), H(qkv), M, N, K, positions, CH(cos_sin_cache), H(key_cache), H(value_cache), slot_mapping, num_heads, num_kv_heads, head_size, rot_dim, workspace, ST));
}
int qspec_qkv_rope_linear_w4a16_xp(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M,
                                   int N, int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                   qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                   int num_heads, int num_kv_heads, int head_size, int rot_dim, void* workspace,
                                   void* stream) {
    XpScope xp;
    return qspec_qkv_rope_linear_w4a16(x, wq, ws, qkv, M, N, K, positions, cos_sin_cache, key_cache, value_cache, slot_mapping,
                                       num_heads, num_kv_heads, head_size, rot_dim, workspace, stream);
}
int qspec_gate_up_silu_linear_s4s4(const int8_t* xq, const qspec_half* xs, const int8_t* wq, const qspec_half* ws,
                                   qspec_half* act, int M, int intermediate, int K, void* stream) {
    const char* op = "qspec_gate_up_silu_linear_s4s4";
    if (M == 0) return 0;
    NONNULL(op, xq); NONNULL(op, xs); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, act);
    if (use_stream() && qspec::gemm_w4a4_stream_supported(M, 2 * intermediate, K, false)) {
        qspec::StreamActs x;
        x.xq = xq; x.xs = CH(xs);
        return finish(op, qspec::gemm_w4a4_stream_gate_up_silu(x, wq, CH(ws), H(act), M, intermediate, K, ST));
    }
    return finish(op, qspec::gemm_w4a4_gate_up_silu(xq, CH(xs), wq, CH(ws), H(act), M, intermediate, K, ST));
}
int qspec_gate_up_silu_linear_w4a16(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                    int intermediate, int K, void* workspace, void* stream) {
    const char* op = "qspec_gate_up_silu_linear_w4a16";
    if (M == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, act);
    if (g_xp && !(use_stream() && qspec::gemm_w4a16_stream_supported(M, 2 * intermediate, K) && qspec::gemm_w4a16_xperm_supported(M, K)))
        return fail("%s: no fragment-major form for (M=%d I=%d K=%d)", op, M, intermediate, K);
    if (use_stream() && qspec::gemm_w4a16_stream_supported(M, 2 * intermediate, K))
        return finish(op, qspec::gemm_w4a16_stream_gate_up_silu(CH(x), wq, CH(ws), H(act), M, intermediate, K, 0, intermediate, ST, g_xp));
    return finish(op, qspec::gemm_w4a16_gate_up_silu(CH(x), wq, CH(ws), H(act), M, intermediate, K, 0, intermediate, workspace, ST));
}
int qspec_gate_up_silu_linear_w4a16_xp(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                       int intermediate, int K, void* workspace, void* stream) {
    XpScope xp;
    return qspec_gate_up_silu_linear_w4a16(x, wq, ws, act, M, intermediate, K, workspace, stream);
}

int qspec_w4a16_linear_ksliced(const qspec_half* x, int64_t ldx, const int8_t* wq, int64_t ldw_bytes,
                               const qspec_half* ws, qspec_half* out, int M, int N, int K, void* workspace,
                               void* stream) {
    const char* op = "qspec_w4a16_linear_ksliced";
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, out);
    if (N % 16 || K % 128 || ldw_bytes % 16 || ldx % 8) return fail("%s: need N%%16, K%%128, ldw%%16, ldx%%8 == 0", op);
    if (use_stream() && qspec::gemm_w4a16_stream_supported(M, N, K))
        return finish(op, qspec::gemm_w4a16_stream(CH(x), ldx, wq, ldw_bytes, CH(ws), H(out), M, N, K, ST));
    {
        const int S = (use_stream() && workspace) ? qspec::gemm_w4a16_stream_partial_slices(M, N, K) : 0;
        if (S > 0 && (size_t)S * M * N * sizeof(float) + 8192 <= qspec::gemm_w4a16_ws_bytes()) {
            float* part = reinterpret_cast<float*>(reinterpret_cast<char*>(workspace) + 8192);
            int rc = qspec::gemm_w4a16_stream_partial(CH(x), ldx, wq, ldw_bytes, part, M, N, K, S, ST);
            if (rc == 0) rc = qspec::gemm_w4a16_partial_finish(part, CH(ws), H(out), M, N, S, ST);
            return finish(op, rc);
        }
    }
    return finish(op, qspec::gemm_w4a16_strided(CH(x), ldx, wq, ldw_bytes, CH(ws), H(out), M, N, K, workspace, ST));
}
int qspec_w4a16_linear_ksliced_raw(const qspec_half* x, int64_t ldx, const int8_t* wq, int64_t ldw_bytes, float* part,
                                   int M, int N, int K, void* workspace, void* stream) {
    const char* op = "qspec_w4a16_linear_ksliced_raw";
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, part);
    if (M > 32) return fail("%s: M=%d > 32 (decode-sized verify batches only)", op, M);
    if (N % 16 || K % 128 || ldw_bytes % 16 || ldx % 8) return fail("%s: need N%%16, K%%128, ldw%%16, ldx%%8 == 0", op);
    if (use_stream() && qspec::gemm_w4a16_stream_supported(M, N, K))
        return finish(op, qspec::gemm_w4a16_stream_partial(CH(x), ldx, wq, ldw_bytes, part, M, N, K, 1, ST));
    return finish(op, qspec::gemm_w4a16_strided_raw(CH(x), ldx, wq, ldw_bytes, part, M, N, K, workspace, ST));
}
int qspec_gate_up_silu_linear_w4a16_shard(const qspec_half* x, const int8_t* wq, const qspec_half* ws, qspec_half* act,
                                          int M, int intermediate, int K, int first_channel, int num_channels,
                                          void* workspace, void* stream) {
    const char* op = "qspec_gate_up_silu_linear_w4a16_shard";
    if (M == 0 || num_channels == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, act);
    if (use_stream() && qspec::gemm_w4a16_stream_supported(M, 2 * intermediate, K))
        return finish(op, qspec::gemm_w4a16_stream_gate_up_silu(CH(x), wq, CH(ws), H(act), M, intermediate, K, first_channel, num_channels, ST));
    return finish(op, qspec::gemm_w4a16_gate_up_silu(CH(x), wq, CH(ws), H(act), M, intermediate, K, first_channel, num_channels, workspace, ST));
}

int qspec_ln_qkv_rope_linear_s4s4(const qspec_half* hidden_in, const qspec_half* delta, qspec_half* hidden_out,
                                  float eps, const int8_t* wq, const qspec_half* ws, qspec_half* qkv, int M, int N,
                                  int K, const int64_t* positions, const qspec_half* cos_sin_cache,
                                  qspec_half* key_cache, qspec_half* value_cache, const int64_t* slot_mapping,
                                  int num_heads, int num_kv_heads, int head_size, int rot_dim, void* sync_workspace,
                                  void* stream) {
    const char* op = "qspec_ln_qkv_rope_linear_s4s4";
    if (M == 0) return 0;
    NONNULL(op, hidden_in); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, qkv); NONNULL(op, positions);
    NONNULL(op, cos_sin_cache); NONNULL(op, key_cache); NONNULL(op, value_cache); NONNULL(op, slot_mapping);
    if (hidden_out == hidden_in) return fail("%s: hidden_out must not alias hidden_in (every workgroup reads it)", op);
    if ((head_size != 128 && head_size != 64) || rot_dim != head_size) return fail("%s: head_size = rot_dim = 128 or 64 only", op);
    if (!qspec::gemm_w4a4_stream_supported(M, N, K, true))
        return fail("%s: need M <= 16, N %% 16 == 0, K in {1024, 2048, 4096, 5120, 8192} (M=%d N=%d K=%d)", op, M, N, K);
    qspec::StreamActs x;
    x.hidden_in = CH(hidden_in); x.delta = CH(delta); x.hidden_out = H(hidden_out); x.eps = eps;
    x.sync = reinterpret_cast<int*>(sync_workspace);
    return finish(op, qspec::gemm_w4a4_stream_qkv_rope(x, wq, CH(ws), H(qkv), M, N, K, positions, CH(cos_sin_cache), H(key_cache), H(value_cache), slot_mapping, num_heads, num_kv_heads, head_size, rot_dim, ST));
}
int qspec_ln_gate_up_silu_linear_s4s4(const qspec_half* hidden_in, const qspec_half* delta, qspec_half* hidden_out,
                                      float eps, const int8_t* wq, const qspec_half* ws, qspec_half* act, int M,
                                      int intermediate, int K, void* sync_workspace, void* stream) {
    const char* op = "qspec_ln_gate_up_silu_linear_s4s4";
    if (M == 0) return 0;
    NONNULL(op, hidden_in); NONNULL(op, wq); NONNULL(op, ws); NONNULL(op, act);
    if (hidden_out == hidden_in) return fail("%s: hidden_out must not alias hidden_in (every workgroup reads it)", op);
    if (!qspec::gemm_w4a4_stream_supported(M, 2 * intermediate, K, true) || intermediate % 8)
        return fail("%s: need M <= 16, intermediate %% 8 == 0, K in {1024, 2048, 4096, 5120, 8192} (M=%d I=%d K=%d)", op, M, intermediate, K);
    qspec::StreamActs x;
    x.hidden_in = CH(hidden_in); x.delta = CH(delta); x.hidden_out = H(hidden_out); x.eps = eps;
    x.sync = reinterpret_cast<int*>(sync_workspace);
    return finish(op, qspec::gemm_w4a4_stream_gate_up_silu(x, wq, CH(ws), H(act), M, intermediate, K, ST));
}
size_t qspec_ln_linear_workspace_bytes(void) { return qspec::gemm_w4a4_stream_sync_bytes(); }
int qspec_w4a16_linear_partial_slices(int M, int N, int K) {
    if (M > 16)   // the M-tiled kernel's launch plan (narrow layers at a few dozen to a few hundred tokens)
        return M >= tiled_min_m() ? qspec::gemm_w4a16_tiled_partial_slices(M, N, K) : 0;
    return use_stream() ? qspec::gemm_w4a16_stream_partial_slices(M, N, K) : 0;
}
int qspec_w4a16_linear_partial(const qspec_half* x, const int8_t* wq, float* part, int M, int N, int K, int slices,
                               void* stream) {
    const char* op = "qspec_w4a16_linear_partial";
    if (M == 0 || N == 0) return 0;
    NONNULL(op, x); NONNULL(op, wq); NONNULL(op, part);
    if (M > 16) {
        if (g_xp) return fail("%s: the fragment-major layout is a 16-row tile (M=%d)", op, M);
        if (slices < 2 || slices != qspec_w4a16_linear_partial_slices(M, N, K))
            return fail("%s: (M=%d N=%d K=%d) takes %d slices (qspec_w4a16_linear_partial_slices), got %d", op, M, N, K,
                        qspec_w4a16_linear_partial_slices(M, N, K), slices);
        return finish(op, qspec::gemm_w4a16_tiled_partial(CH(x), wq, part, M, N, K, slices, ST));
    }
    // the planned count (qspec_w4a16_linear_partial_slices), or -- for shapes that also run unsliced -- any count whose
    // slices the streaming kernel takes (the verify pass's o_proj experiment)
    const int planned = qspec::gemm_w4a16_stream_partial_slices(M, N, K);
    if (slices < 2 || slices > 8 || K % slices || (planned ? slices != planned : !qspec::gemm_w4a16_stream_supported(M, N, K / slices)))
        return fail("%s: (M=%d N=%d K=%d) takes %d slices (qspec_w4a16_linear_partial_slices), got %d", op, M, N, K, planned, slices);
    if (g_xp && !qspec::gemm_w4a16_xperm_supported(M, K / slices)) return fail("%s: no fragment-major form for slices of %d", op, K / slices);
    return finish(op, qspec::gemm_w4a16_stream_partial(CH(x), 0, wq, 0, part, M, N, K, slices, ST, g_xp));
}
int qspec_w4a16_linear_partial_xp(const qspec_half* x, const int8_t* wq, float* part, int M, int N, int K, int slices,
                                  void* stream) {
    XpScope xp;
    return qspec_w4a16_linear_partial(x, wq, part, M, N, K, slices, stream);
}
int qspec_add_rms_norm_fp16_partial(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const float* part,
                                    const qspec_half* ws, int slices, float eps, int tokens, int hidden, void* stream) {
    const char* op = "qspec_add_rms_norm_fp16_partial";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, out); NONNULL(op, x); NONNULL(op, part); NONNULL(op, ws); NONNULL(op, hidden_out);
    if (slices < 1) return fail("%s: slices < 1", op);
    if (hidden % 1024 || hidden > 8192 || hidden <= 0) return fail("%s: hidden=%d must be a multiple of 1024, <= 8192", op, hidden);
    if (g_xp && tokens > 32) return fail("%s: the fragment-major layout is one or two 16-row tiles (tokens=%d)", op, tokens);
    return finish(op, qspec::ln_fp16_partial(CH(x), part, CH(ws), slices, H(hidden_out), H(out), eps, tokens, hidden, ST, g_xp));
}
int qspec_add_rms_norm_fp16_partial_xp(qspec_half* out, qspec_half* hidden_out, const qspec_half* x, const float* part,
                                       const qspec_half* ws, int slices, float eps, int tokens, int hidden, void* stream) {
    XpScope xp;
    return qspec_add_rms_norm_fp16_partial(out, hidden_out, x, part, ws, slices, eps, tokens, hidden, stream);
}
int qspec_rowwise_scaled_linear_s4s4_partial_slices(int M, int N, int K) { return qspec::gemm_w4a4_stream_partial_slices(M, N, K); }
int qspec_rowwise_scaled_linear_s4s4_partial(const int8_t* xq, const int8_t* wq, int32_t* ipart, int M, int N, int K, int slices,
                                             void* stream) {
    const char* op = "qspec_rowwise_scaled_linear_s4s4_partial";
    if (M <= 0 || N <= 0) return fail("%s: empty problem", op);
    NONNULL(op, xq); NONNULL(op, wq); NONNULL(op, ipart);
    if (slices < 2 || slices != qspec::gemm_w4a4_stream_partial_slices(M, N, K))
        return fail("%s: (M=%d N=%d K=%d) is not built for %d K slices (ask qspec_rowwise_scaled_linear_s4s4_partial_slices)", op, M, N, K, slices);
    return finish(op, qspec::gemm_w4a4_stream_partial(xq, wq, ipart, M, N, K, slices, ST));
}
int qspec_add_rms_norm_ipartial(int8_t* q, qspec_half* scale, qspec_half* out_f16, qspec_half* hidden_out, const qspec_half* x,
                                const int32_t* ipart, const qspec_half* xs, const qspec_half* ws, int slices, float eps,
                                int tokens, int hidden, void* stream) {
    const char* op = "qspec_add_rms_norm_ipartial";
    if (tokens < 0) return fail("%s: tokens < 0", op);
    if (tokens == 0) return 0;
    NONNULL(op, x); NONNULL(op, ipart); NONNULL(op, xs); NONNULL(op, ws); NONNULL(op, hidden_out);
    if (q) { NONNULL(op, scale); } else { NONNULL(op, out_f16); }
    if (slices < 1) return fail("%s: slices < 1", op);
    if (hidden % 1024 || hidden > 8192 || hidden <= 0) return fail("%s: hidden=%d must be a multiple of 1024, <= 8192", op, hidden);
    return finish(op, qspec::ln_ipartial(CH(x), ipart, CH(xs), CH(ws), slices, H(hidden_out), H(out_f16), q, H(scale), eps, tokens,
                                         hidden, ST));
}
#ifdef QS_EXPERIMENTAL   // (csrc/experimental/qspec_hip_experimental.h)
int qspec_prefetch(const void* p, size_t bytes, int workgroups, void* stream) {
    const char* op = "qspec_prefetch";
    if (bytes == 0) return 0;
    NONNULL(op, p);
    if (((uintptr_t)p) % 16) return fail("%s: pointer must be 16-byte aligned", op);
    return finish(op, qspec::prefetch_l2(p, bytes, workgroups, ST));
}
int qspec_prefetch_tiles(const void* p, size_t tile_bytes, int first_tile, int ntiles, int workgroups, void* stream) {
    const char* op = "qspec_prefetch_tiles";
    if (ntiles <= 0) return 0;
    NONNULL(op, p);
    if (((uintptr_t)p) % 16 || tile_bytes % 16) return fail("%s: pointer and tile_bytes must be multiples of 16", op);
    return finish(op, qspec::prefetch_tiles(p, tile_bytes, first_tile, ntiles, workgroups, ST));
}
#endif
int qspec_ln_linear_s4s4_supported(int M, int N, int K) { return qspec::gemm_w4a4_stream_supported(M, N, K, true) ? 1 : 0; }

}  // extern "C"
