// bench.py ONLY -- not part of the product library, not declared in include/qspec_hip.h.
// Synthetic-workload knob: random int4 weights give a draft/target agreement near zero, a trained QSpec checkpoint
// ~0.96 (SURVEY.md 8d sanctions a controlled-agreement mode).  With probability rho the target logit of the
// proposed token is raised to the fp16 maximum, so the verify pass "agrees" with the draft at a controlled rate.  Every
// kernel of the cycle still runs on the same shapes; only token values change.  Built by __graft_entry__.build() into
// bench_kernels/libqspec_bench.so and loaded by bench.py's BenchEngine.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 f16;

__device__ __forceinline__ uint32_t mulhi32(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                           uint32_t (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 10; i++) {
        const uint32_t h0 = mulhi32(0xD2511F53u, c0), l0 = 0xD2511F53u * c0;
        const uint32_t h1 = mulhi32(0xCD9E8D57u, c2), l1 = 0xCD9E8D57u * c2;
        c0 = h1 ^ c1 ^ k0; c1 = l1; c2 = h0 ^ c3 ^ k1; c3 = l0;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}

__global__ void force_agreement_kernel(f16* __restrict__ logits, const int64_t* __restrict__ draft_ids, int64_t di_sb,
                                       int64_t di_sk, float rho, const uint64_t* __restrict__ rng_state, int B, int k,
                                       int V) {
    const int bk = blockIdx.x * blockDim.x + threadIdx.x;
    if (bk >= B * k) return;
    const int b = bk / k, i = bk % k;
    uint32_t r[4];
    philox4x32(0xA5A5A5A5u, (uint32_t)bk, (uint32_t)rng_state[1], (uint32_t)(rng_state[1] >> 32), (uint32_t)rng_state[0],
               (uint32_t)(rng_state[0] >> 32) ^ 0x51ED27u, r);
    if ((float)(r[0] >> 8) * (1.0f / 16777216.0f) < rho)
        logits[((size_t)b * (k + 1) + i) * V + draft_ids[b * di_sb + i * di_sk]] = (f16)60000.0f;
}

extern "C" int qspec_bench_force_agreement(void* target_logits, const int64_t* draft_token_ids, int64_t ids_stride_b,
                                           int64_t ids_stride_k, float rho, const uint64_t* rng_state, int batch, int k,
                                           int vocab, void* stream) {
    if (batch == 0) return 0;
    hipLaunchKernelGGL(force_agreement_kernel, dim3((batch * k + 63) / 64), dim3(64), 0, (hipStream_t)stream,
                       (f16*)target_logits, draft_token_ids, ids_stride_b, ids_stride_k, rho, rng_state, batch, k, vocab);
    return hipGetLastError() == hipSuccess ? 0 : 1;
}
