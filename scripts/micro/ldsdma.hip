// Calibration for the loader / consumer form of the draft GEMMs (stage A): how fast does ONE loader
// wave per CU (or two / four) pull a workgroup's weight tiles into an LDS ring by LDS-DMA (global_load_lds_dwordx4), with
// nobody consuming -- the floor such a kernel cannot beat -- against the register path's pure read of the same bytes
// (scripts/micro/stream.hip, contiguous pattern)?  Graph replay over > 256 MiB of rotating copies, launch boundary
// included, as in bench.py's roofline leg.
//   hipcc --offload-arch=gfx950 -O3 -o ldsdma.bin ldsdma.hip && ./ldsdma.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int NT>
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if (NT)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

// Workgroup = 4 idle waves + NLW loader waves.  Tile = 16 weight rows x Kb bytes; one glds moves 1 KiB of ONE row (lane l
// takes the 16-byte piece l ^ (row & 15): an XOR swizzle on the source address, the LDS image stays lane-linear).
template <int NT, int NLW>
__global__ __launch_bounds__(256 + NLW * 64) void dma_stream(const unsigned char* __restrict__ w, int Kb, int ntiles, int ring_kib,
                                                             unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (wave < 4) return;
    const int lw = wave - 4;
    const unsigned lds0 = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char*)lds;   // LDS byte address of the ring
    const int per_row = Kb >> 10, per_tile = 16 * per_row;
    int n = 0;
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        for (int j = lw; j < per_tile; j += NLW) {
            const int row = j / per_row, piece = j - row * per_row;
            const unsigned char* src = w + ((size_t)tile * 16 + row) * Kb + (size_t)piece * 1024 + ((lane ^ (row & 15)) << 4);
            const unsigned dst = lds0 + (unsigned)(((n * NLW + lw) & (ring_kib - 1)) << 10);
            glds16<NT>(src, dst);
            n++;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0 && lds[lw * 1024] == 0x5a && lds[1] == 0x77) out[blockIdx.x] = 1;   // keep the ring alive
}

// the register path's pure read, contiguous 1 KiB per wave-load (scripts/micro/stream.hip pattern 0, UB = 8)
template <int UB>
__global__ void rd(const unsigned char* __restrict__ w, int Kb, int nsteps_per_wave, int ntiles, unsigned* out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, NW = blockDim.x >> 6;
    u32x4 acc = {0, 0, 0, 0};
    for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const unsigned char* base = w + (size_t)tile * 16 * Kb + (size_t)wave * 1024 + lane * 16;
        for (int s0 = 0; s0 < nsteps_per_wave; s0 += UB) {
            u32x4 v[UB];
#pragma unroll
            for (int u = 0; u < UB; u++) v[u] = *reinterpret_cast<const u32x4*>(base + (size_t)(s0 + u) * NW * 1024);
#pragma unroll
            for (int u = 0; u < UB; u++) acc ^= v[u];
        }
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

template <int NT, int NLW>
float run_dma(const unsigned char* pool, size_t bytes, int L, int Kb, int ntiles, int grid, int ring_kib, unsigned* out, hipStream_t st) {
    auto k = dma_stream<NT, NLW>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, ring_kib * 1024);
    return time_graph([&](int i) {
        hipLaunchKernelGGL(k, dim3(grid), dim3(256 + NLW * 64), ring_kib * 1024, st, pool + (size_t)(i % L) * bytes, Kb, ntiles, ring_kib, out);
    }, L * 2, st);
}

int main() {
    hipStream_t st; CK(hipStreamCreate(&st));
    const size_t POOL = 1100ull << 20;
    unsigned char* pool; unsigned* out;
    CK(hipMalloc(&pool, POOL)); CK(hipMalloc(&out, 1 << 20));
    CK(hipMemset(pool, 1, POOL));
    struct Shape { const char* name; int N, K; } shapes[] = {{"o", 4096, 4096}, {"qkv", 6144, 4096}, {"down", 4096, 14336}, {"gate_up", 28672, 4096}, {"lm_head(fp16 [128256, 4096] as bytes)", 128256, 16384}};
    for (auto& sh : shapes) {
        const int Kb = sh.K / 2, ntiles = sh.N / 16;
        const size_t bytes = (size_t)sh.N * Kb;
        const int L = (int)(POOL / bytes) > 0 ? (int)(POOL / bytes) : 1;
        const int grid = ntiles < 256 ? ntiles : 256;
        const int nsteps = 16 * Kb / 1024;   // 1 KiB wave-loads per tile
        float tr = time_graph([&](int i) { hipLaunchKernelGGL(rd<4>, dim3(grid), dim3(512), 0, st, pool + (size_t)(i % L) * bytes, Kb, nsteps / 8, ntiles, out); }, L * 2, st);
        printf("%-8s %5.1f MB  register pure read (8 waves, contiguous) %6.2f us (%5.0f GB/s)\n", sh.name, bytes / 1e6, tr, bytes / tr / 1e3);
        for (int ring : {32, 64, 128}) {
            float a = run_dma<0, 1>(pool, bytes, L, Kb, ntiles, grid, ring, out, st);
            float b = run_dma<1, 1>(pool, bytes, L, Kb, ntiles, grid, ring, out, st);
            float c = run_dma<0, 2>(pool, bytes, L, Kb, ntiles, grid, ring, out, st);
            float d = run_dma<1, 2>(pool, bytes, L, Kb, ntiles, grid, ring, out, st);
            float e = run_dma<1, 4>(pool, bytes, L, Kb, ntiles, grid, ring, out, st);
            float f8 = run_dma<1, 8>(pool, bytes, L, Kb, ntiles, grid, ring, out, st);
            float f12 = run_dma<1, 12>(pool, bytes, L, Kb, ntiles, grid, ring, out, st);
            printf("         LDS-DMA ring %3d KiB: 1 loader %6.2f us (%5.0f GB/s) nt %6.2f (%5.0f) | 2 loaders %6.2f nt %6.2f | 4 loaders nt %6.2f | 8 nt %6.2f | 12 nt %6.2f (%5.0f GB/s)\n",
                   ring, a, bytes / a / 1e3, b, bytes / b / 1e3, c, d, e, f8, f12, bytes / f12 / 1e3);
        }
    }
    return 0;
}
