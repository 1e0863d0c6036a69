// Calibration: how fast can a kernel that ONLY reads a cold weight matrix go, per shape and access pattern?
// (graph replay over > 256 MiB of rotating copies; results reduced to one store per wave so nothing is elided)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// pattern 0: fully contiguous 1 KiB per wave-load; workgroup of NW waves owns a contiguous chunk of `bytes_per_wg`
// pattern 1: 16 rows x 64 B per wave-load (MFMA 16x16 tile, lane = row + 16*chunk), waves interleave 64-B steps
// pattern 2: as 1 but a wave's consecutive loads pair into 128-B lines
template <int PAT, int UB>
__global__ void rd(const unsigned char* __restrict__ w, int Kb, int nsteps_per_wave, int ntiles, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    const int r = lane & 15, g = lane >> 4;
    u32x4 acc = {0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned char* base;
        if (PAT == 0) base = w + (size_t)tile * 16 * Kb + (size_t)wave * 1024 + lane * 16;
        else base = w + ((size_t)tile * 16 + r) * Kb + g * 16;
        for (int s0 = 0; s0 < nsteps_per_wave; s0 += UB) {
            u32x4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) {
                const int i = s0 + u;
                size_t off;
                if (PAT == 0) off = (size_t)i * NW * 1024;
                else if (PAT == 1) off = (size_t)(i * NW + wave) * 64;
                else off = (size_t)((i >> 1) * 2 * NW + wave * 2 + (i & 1)) * 64;
                v[u] = *reinterpret_cast<const u32x4*>(base + off);
            }
#pragma unroll
            for (int u = 0; u < UB; u++) acc ^= v[u];
        }
    }
    if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) out[blockIdx.x] = 1;
}
__global__ void empty_k() {}

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
    return ms * 1e3f / (5.f * n);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t POOL = 640ull << 20;
    unsigned char* pool; unsigned* out;
    CK(hipMalloc(&pool, POOL)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(pool, 1, POOL));
    printf("empty 256 WG: %.2f us\n", time_graph([&](int) { hipLaunchKernelGGL(empty_k, dim3(256), dim3(256), 0, st); }, 100, st));
    struct Shape { const char* name; int N, K; } shapes[] = {{"o", 4096, 4096}, {"qkv", 6144, 4096}, {"down", 4096, 14336}, {"gate_up", 28672, 4096}};
    for (auto& sh : shapes) {
        const int Kb = sh.K / 2, ntiles = sh.N / 16;
        const size_t bytes = (size_t)sh.N * Kb;
        const int L = (int)(POOL / bytes);
        for (int nw : {4, 8, 16}) {
            const int nsteps = Kb / 64;
            if (nsteps % nw) continue;
            const int spw = nsteps / nw;
            for (int grid : {256, 512, 1024, ntiles}) {
                if (grid > ntiles) continue;
                float t[3];
                auto run = [&](auto kern) {
                    return time_graph([&](int i) { hipLaunchKernelGGL(kern, dim3(grid), dim3(nw * 64), 0, st, pool + (size_t)(i % L) * bytes, Kb, spw, ntiles, out); }, L * 2, st);
                };
                if (spw % 8 == 0 && getenv("UB4") == nullptr) { t[0] = run(rd<0, 8>); t[1] = run(rd<1, 8>); t[2] = run(rd<2, 8>); }
                else if (spw % 7 == 0) { t[0] = run(rd<0, 7>); t[1] = run(rd<1, 7>); t[2] = -1; }
                else if (spw % 4 == 0) { t[0] = run(rd<0, 4>); t[1] = run(rd<1, 4>); t[2] = run(rd<2, 4>); }
                else if (spw % 2 == 0) { t[0] = run(rd<0, 2>); t[1] = run(rd<1, 2>); t[2] = run(rd<2, 2>); }
                else continue;
                printf("%-8s NW=%2d grid=%5d  contiguous %6.2f us (%5.0f GB/s) | 16x64B %6.2f us (%5.0f) | paired128 %6.2f us (%5.0f)\n", sh.name, nw, grid,
                       t[0], bytes / t[0] / 1e3, t[1], bytes / t[1] / 1e3, t[2], t[2] > 0 ? bytes / t[2] / 1e3 : 0.0);
            }
        }
    }
    return 0;
}
