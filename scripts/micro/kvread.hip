// Calibration for decode attention: how fast can a launch that ONLY reads the K / V rows of one
// layer go, in the access pattern the paged cache imposes, against the same bytes laid out contiguously per kv head?
//   cache layout (reshape_and_cache_flash, vllm/attention/backends/flash_attn.py): [slot, kv head, 128] fp16 -> the rows
//   of ONE kv head are 256-byte pieces at a 2-KiB stride;  "HND": [kv head, slot, 128] -> 4-KiB contiguous per block.
// One workgroup per (sequence, kv head, split) as paged_attention*_kernel; NW waves; a wave-load takes 4 rows x 256 B;
// UB wave-loads of K and UB of V in flight per wave.  Graph replay over 8 rotating layers (> 256 MiB), boundary included.
//   hipcc --offload-arch=gfx950 -O3 -o kvread.bin kvread.hip && ./kvread.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int UB, bool HND>
__global__ void kvread(const unsigned char* __restrict__ kc, const unsigned char* __restrict__ vc, int ctx, int nkv,
                       int slots_per_seq, int n_splits, unsigned* out) {
    const int seq = blockIdx.x, kvh = blockIdx.y, split = blockIdx.z;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int c16 = lane & 15, g4 = lane >> 4;
    const int kps = (ctx + n_splits - 1) / n_splits, k0 = split * kps, k1 = min(ctx, k0 + kps);
    const size_t slot0 = (size_t)seq * slots_per_seq;
    const size_t total_slots = (size_t)gridDim.x * slots_per_seq;
    u32x4 acc = {0, 0, 0, 0};
    // wave w takes rows k0 + 4 (w + NW i) + g4
    for (int kb = k0 + 4 * wave; kb < k1; kb += 4 * NW * UB) {
        u32x4 a[UB], b[UB];
#pragma unroll
        for (int u = 0; u < UB; u++) {
            const int key = min(kb + 4 * NW * u + g4, k1 - 1);
            const size_t off = HND ? ((size_t)kvh * total_slots + slot0 + key) * 256 + c16 * 16
                                   : ((slot0 + key) * nkv + kvh) * 256 + c16 * 16;
            a[u] = *reinterpret_cast<const u32x4*>(kc + off);
            b[u] = *reinterpret_cast<const u32x4*>(vc + off);
        }
#pragma unroll
        for (int u = 0; u < UB; u++) acc ^= a[u] ^ b[u];
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}

template <typename F> float time_graph(F launch, int n, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; i++) launch(i);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int r = 0; r < 5; r++) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    hipGraphExecDestroy(ge); hipGraphDestroy(g);
    return ms * 1e3f / (5.f * n);
}

template <int UB, bool HND>
float run(const unsigned char* pool, size_t layer_bytes, int L, int B, int ctx, int nkv, int sps, int n_splits, int nw, unsigned* out,
          hipStream_t st) {
    return time_graph([&](int i) {
        const unsigned char* kc = pool + (size_t)(i % L) * layer_bytes;
        hipLaunchKernelGGL((kvread<UB, HND>), dim3(B, nkv, n_splits), dim3(64 * nw), 0, st, kc, kc + layer_bytes / 2, ctx, nkv, sps,
                           n_splits, out);
    }, 2 * L, st);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    struct Case { const char* name; int B, ctx, sps, n_splits, nkv; } cases[] = {
        {"config 3: bs 32, ctx 512, 1 split ", 32, 512, 640, 1, 8},
        {"config 3: bs 32, ctx 512, 2 splits", 32, 512, 640, 2, 8},
        {"headline: bs 4, ctx 512, 8 splits ", 4, 512, 640, 8, 8},
        {"bs 32, ctx 2048, 1 split          ", 32, 2048, 2176, 1, 8},
        {"Llama-2-13B: bs 4, ctx 512, 3 splits, 40 kv heads (rows of a head at a 10-KiB stride)", 4, 512, 640, 3, 40},
        {"Llama-2-13B: bs 4, ctx 512, 6 splits, 40 kv heads", 4, 512, 640, 6, 40},
    };
    unsigned* out; CK(hipMalloc(&out, 1 << 20));
    const char* only = getenv("KVREAD_ONLY");   // substring filter on the case name (counter passes)
    for (auto& c : cases) {
        if (only && !strstr(c.name, only)) continue;
        const int nkv = c.nkv;
        const size_t layer_bytes = 2ull * c.B * c.sps * nkv * 256;   // K then V
        const int L = (int)((600ull << 20) / layer_bytes) > 8 ? 8 : (int)((600ull << 20) / layer_bytes);
        unsigned char* pool; CK(hipMalloc(&pool, layer_bytes * L)); CK(hipMemset(pool, 1, layer_bytes * L));
        const double bytes = 2.0 * c.B * c.ctx * nkv * 256;
        printf("%s  %6.1f MB of K+V per launch, %d workgroups, %d rotating layers\n", c.name, bytes / 1e6, c.B * nkv * c.n_splits, L);
        for (int nw : {4, 8, 16}) {
            float a1 = run<1, false>(pool, layer_bytes, L, c.B, c.ctx, nkv, c.sps, c.n_splits, nw, out, st);
            float a2 = run<2, false>(pool, layer_bytes, L, c.B, c.ctx, nkv, c.sps, c.n_splits, nw, out, st);
            float a4 = run<4, false>(pool, layer_bytes, L, c.B, c.ctx, nkv, c.sps, c.n_splits, nw, out, st);
            float a8 = run<8, false>(pool, layer_bytes, L, c.B, c.ctx, nkv, c.sps, c.n_splits, nw, out, st);
            float h2 = run<2, true>(pool, layer_bytes, L, c.B, c.ctx, nkv, c.sps, c.n_splits, nw, out, st);
            float h4 = run<4, true>(pool, layer_bytes, L, c.B, c.ctx, nkv, c.sps, c.n_splits, nw, out, st);
            float h8 = run<8, true>(pool, layer_bytes, L, c.B, c.ctx, nkv, c.sps, c.n_splits, nw, out, st);
            printf("   %2d waves: cache layout, 1/2/4/8 K+V wave-load pairs in flight per wave: %6.2f %6.2f %6.2f %6.2f us (%5.0f GB/s)"
                   " | per-head contiguous 2/4/8: %6.2f %6.2f %6.2f us (%5.0f GB/s)\n",
                   nw, a1, a2, a4, a8, bytes / a8 / 1e3, h2, h4, h8, bytes / h8 / 1e3);
        }
        CK(hipFree(pool));
    }
    return 0;
}
