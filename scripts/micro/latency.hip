// Calibration: cost of a kernel launch, of dependent global-load hops, of fences/atomics on this box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void empty_k() {}
__global__ void chase(const int* __restrict__ next, int hops, int* out) {
    int p = threadIdx.x + blockIdx.x * blockDim.x;
    for (int i = 0; i < hops; i++) p = next[p];
    if (p == -12345) out[0] = p;
}
__global__ void store_k(int* out) { out[threadIdx.x + blockIdx.x * blockDim.x] = 1; }
__global__ void fence_k(int* cnt, int* out) {
    out[threadIdx.x + blockIdx.x * blockDim.x] = 1;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__global__ void clock_k(long long* out, int iters) {
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float a = threadIdx.x;
    for (int i = 0; i < iters; i++) a = a * 1.0001f + 0.5f;
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (a == 1234.5f) out[2] = 1;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

template <typename F> float time_graph(F launch, int n, hipStream_t st) {
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < n; i++) launch();
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
    const int N = 1 << 24;  // 64 MB of ints: beyond L2
    std::vector<int> h(N);
    for (int i = 0; i < N; i++) h[i] = (int)(((long long)i * 1000003LL + 12345) % N);
    int *next, *out, *cnt; long long* clk;
    CK(hipMalloc(&next, N * 4)); CK(hipMalloc(&out, 1 << 22)); CK(hipMalloc(&cnt, 4)); CK(hipMalloc(&clk, 64));
    CK(hipMemcpy(next, h.data(), N * 4, hipMemcpyHostToDevice)); CK(hipMemset(cnt, 0, 4));
    printf("empty kernel (4 WG x 256):       %.2f us\n", time_graph([&] { hipLaunchKernelGGL(empty_k, dim3(4), dim3(256), 0, st); }, 200, st));
    printf("empty kernel (1024 WG x 256):    %.2f us\n", time_graph([&] { hipLaunchKernelGGL(empty_k, dim3(1024), dim3(256), 0, st); }, 200, st));
    printf("store kernel (4 WG):             %.2f us\n", time_graph([&] { hipLaunchKernelGGL(store_k, dim3(4), dim3(256), 0, st, out); }, 200, st));
    for (int hops : {1, 2, 4, 8, 16})
        printf("chase %2d hops (4 WG x 64):       %.2f us\n", hops, time_graph([&] { hipLaunchKernelGGL(chase, dim3(4), dim3(64), 0, st, next, hops, out); }, 100, st));
    printf("store+release fence+atomic (160 WG): %.2f us\n", time_graph([&] { hipLaunchKernelGGL(fence_k, dim3(160), dim3(256), 0, st, cnt, out); }, 100, st));
    for (int it : {1000, 100000}) {
        hipLaunchKernelGGL(clock_k, dim3(256), dim3(256), 0, st, clk, it); hipStreamSynchronize(st);
        long long c[2]; hipMemcpy(c, clk, 16, hipMemcpyDeviceToHost);
        printf("clock: %lld shader cycles in %lld x10ns -> %.0f MHz (iters %d)\n", c[0], c[1], c[0] / (c[1] * 0.01), it);
    }
    return 0;
}
