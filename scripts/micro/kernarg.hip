// Calibration: how long does a wave wait for its kernel arguments (s_load through the scalar cache) at the start of a launch
// inside a hipGraph chain, and does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count=N) remove the wait?
// hipcc --offload-arch=gfx950 -O3 -o kernarg.bin kernarg.hip            (add -mllvm -amdgpu-kernarg-preload-count=8 for the second binary)
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
struct Big { long long* out; int pad[40]; int x; };
__global__ __launch_bounds__(256) void k_struct(Big b) {       // one by-value struct, as the library's kernels take it
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    long long* o = b.out;
    int x = b.x;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+s"(x), "+s"(o)::"memory");
    if ((threadIdx.x & 63) == 0) o[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0 + (x == 12345);
}
__global__ __launch_bounds__(256) void k_flat(long long* out, int x, int y, int z) {   // scalar arguments (preloadable)
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    long long* o = out;
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1), "+s"(x), "+s"(o)::"memory");
    if ((threadIdx.x & 63) == 0) o[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0 + (x == 12345) + (y + z == 12345);
}
__global__ void k_other(float* p) { p[threadIdx.x + blockIdx.x * blockDim.x] += 1.0f; }
int main() {
    long long* d; float* f;
    hipMalloc(&d, 256 * 4 * 8 * 2); hipMalloc(&f, 1 << 20);
    hipStream_t st; hipStreamCreate(&st);
    hipGraph_t g; hipGraphExec_t ge;
    Big b{}; b.out = d; b.x = 3;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 20; i++) {
        hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, st, f);
        hipLaunchKernelGGL(k_struct, dim3(256), dim3(256), 0, st, b);
        hipLaunchKernelGGL(k_other, dim3(256), dim3(256), 0, st, f);
        hipLaunchKernelGGL(k_flat, dim3(256), dim3(256), 0, st, d + 1024, 3, 4, 5);
    }
    hipStreamEndCapture(st, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int r = 0; r < 3; r++) { hipGraphLaunch(ge, st); hipStreamSynchronize(st); }
    std::vector<long long> h(2048);
    hipMemcpy(h.data(), d, 2048 * 8, hipMemcpyDeviceToHost);
    for (int k = 0; k < 2; k++) {
        std::vector<long long> v(h.begin() + k * 1024, h.begin() + (k + 1) * 1024);
        std::sort(v.begin(), v.end());
        printf("%s: cycles from a wave's first instruction to its arguments: min %lld median %lld 90%% %lld max %lld\n",
               k == 0 ? "by-value struct (176 B)" : "four scalar arguments ", v[0], v[512], v[921], v[1023]);
    }
    return 0;
}
