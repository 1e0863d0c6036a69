// Calibration: what does straight-line code cost the FIRST time a kernel runs it (instruction fetch) against its VALU
// issue time?  The fused-norm prologues are ~500-800 instructions executed once per workgroup.
//   straight<N>: N 8-byte VALU instructions (v_fma_f32, 8 rotating registers, no memory), stamped by s_memtime
//   thrash:      64 KB of other code, one wave per CU, to evict the instruction cache between launches
//   loop form:   the same N instructions as 32 iterations of N / 32
// hipcc --offload-arch=gfx950 -O3 -o icache.bin icache.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

#define FMA8 "v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t" \
             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9\n\t"

template <int REPT8>
__global__ __launch_bounds__(512) void straight(long long* out, float m, float c, float* sink) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(".rept %c10\n\t" FMA8 ".endr"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(m), "v"(c), "n"(REPT8));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int REPT8, int ITERS>
__global__ __launch_bounds__(512) void looped(long long* out, float m, float c, float* sink) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int i = 0; i < ITERS; i++)
        asm volatile(".rept %c10\n\t" FMA8 ".endr"
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                     : "v"(m), "v"(c), "n"(REPT8));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

__global__ __launch_bounds__(64) void thrash(float m, float c, float* sink) {
    float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;
    asm volatile(".rept 1024\n\t" FMA8 ".endr"   // 8192 x 8 bytes = 64 KB
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(m), "v"(c));
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f) sink[0] = 1;
}

static long long* d_out;
static float* d_sink;
static void report(const char* what, int waves) {
    std::vector<long long> h(256 * waves);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    printf("%-58s cycles per wave: min %6lld  median %6lld  max %6lld\n", what, h.front(), h[h.size() / 2], h.back());
}

int main() {
    CK(hipMalloc(&d_out, 256 * 16 * 8));
    CK(hipMalloc(&d_sink, 64));
    const float m = 1.0001f, c = 0.5f;
    for (int threads : {256, 512}) {
        const int waves = threads / 64;
        printf("---- %d waves per workgroup (one workgroup per CU); 1024 instructions, 8 KB of code\n", waves);
        hipLaunchKernelGGL(straight<128>, dim3(256), dim3(threads), 0, 0, d_out, m, c, d_sink); CK(hipDeviceSynchronize());
        report("straight 1024, first launch of the process", waves);
        hipLaunchKernelGGL(straight<128>, dim3(256), dim3(threads), 0, 0, d_out, m, c, d_sink); CK(hipDeviceSynchronize());
        report("straight 1024, launched again (same code just ran)", waves);
        hipLaunchKernelGGL(thrash, dim3(256), dim3(64), 0, 0, m, c, d_sink);
        hipLaunchKernelGGL(straight<128>, dim3(256), dim3(threads), 0, 0, d_out, m, c, d_sink); CK(hipDeviceSynchronize());
        report("straight 1024, after 64 KB of other code", waves);
        hipLaunchKernelGGL(thrash, dim3(256), dim3(64), 0, 0, m, c, d_sink);
        hipLaunchKernelGGL((looped<4, 32>), dim3(256), dim3(threads), 0, 0, d_out, m, c, d_sink); CK(hipDeviceSynchronize());
        report("loop 32 x 32 instructions, after 64 KB of other code", waves);
        hipLaunchKernelGGL((looped<4, 32>), dim3(256), dim3(threads), 0, 0, d_out, m, c, d_sink); CK(hipDeviceSynchronize());
        report("loop 32 x 32 instructions, launched again", waves);
        hipLaunchKernelGGL(thrash, dim3(256), dim3(64), 0, 0, m, c, d_sink);
        hipLaunchKernelGGL(straight<32>, dim3(256), dim3(threads), 0, 0, d_out, m, c, d_sink); CK(hipDeviceSynchronize());
        report("straight 256, after 64 KB of other code", waves);
        hipLaunchKernelGGL(straight<32>, dim3(256), dim3(threads), 0, 0, d_out, m, c, d_sink); CK(hipDeviceSynchronize());
        report("straight 256, launched again", waves);
    }
    return 0;
}
