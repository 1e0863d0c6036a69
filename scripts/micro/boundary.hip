// Calibration: what does a dependent kernel boundary inside a hipGraph cost, as a function of the launch's shape?
// A chain of N launches of a kernel whose every thread stores one word (so the launch is not optimised away and has
// something to write back); grid, block, dynamic LDS and kernel-argument size vary.
// hipcc --offload-arch=gfx950 -O3 -o boundary.bin boundary.hip
#include <hip/hip_runtime.h>
#include <cstdio>
struct Big { int v[60]; };
template <int T>
__global__ __launch_bounds__(T) void k_small(int* out, int x) {
    extern __shared__ int lds[];
    if (x == 12345) lds[threadIdx.x] = x;
    if (threadIdx.x == 0) out[blockIdx.x] = x;
}
template <int T>
__global__ __launch_bounds__(T) void k_bigarg(int* out, Big b) {
    extern __shared__ int lds[];
    if (b.v[7] == 12345) lds[threadIdx.x] = b.v[3];
    if (threadIdx.x == 0) out[blockIdx.x] = b.v[1];
}
static int* d_out;
template <typename F>
static float chain(F launch, int n) {
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; i++) launch(st);
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, st);
    for (int r = 0; r < 5; r++) hipGraphLaunch(ge, st);
    hipEventRecord(b, st); hipStreamSynchronize(st);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3f / (5 * n);
}
template <int T>
static void row(int grid, int lds) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_small<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_bigarg<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    Big b{}; b.v[1] = 3;
    const float t0 = chain([&](hipStream_t st) { hipLaunchKernelGGL(k_small<T>, dim3(grid), dim3(T), lds, st, d_out, 3); }, 200);
    const float t1 = chain([&](hipStream_t st) { hipLaunchKernelGGL(k_bigarg<T>, dim3(grid), dim3(T), lds, st, d_out, b); }, 200);
    printf("grid %4d x %4d threads, %3d KB LDS: %.2f us per launch (16-byte arguments), %.2f us (256-byte arguments)\n", grid, T,
           lds / 1024, t0, t1);
}
int main() {
    hipMalloc(&d_out, 4096 * 4);
    for (int grid : {4, 32, 256, 512}) {
        row<256>(grid, 0); row<512>(grid, 0); row<768>(grid, 0); row<1024>(grid, 0);
        row<512>(grid, 25 * 1024); row<768>(grid, 25 * 1024); row<512>(grid, 100 * 1024); row<1024>(grid, 100 * 1024);
    }
    return 0;
}
