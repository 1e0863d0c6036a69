// Calibration: how far apart do the waves of one workgroup start, as a function of the kernel's VGPR allocation?
// Every wave stamps s_memtime as its first instruction; the kernel does nothing else (one touched high register forces the
// allocation).   hipcc --offload-arch=gfx950 -O3 -o wave_start.bin wave_start.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
#define KERNEL(NAME, T, REG)                                                                       \
    __global__ __launch_bounds__(T) void NAME(long long* out) {                                    \
        long long t0;                                                                              \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                 \
        asm volatile("v_mov_b32 " REG ", 0" ::: REG);                                              \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (T / 64) + (threadIdx.x >> 6)] = t0;        \
    }
KERNEL(k256_16, 256, "v15")
KERNEL(k256_120, 256, "v119")
KERNEL(k256_250, 256, "v249")
KERNEL(k512_16, 512, "v15")
KERNEL(k512_64, 512, "v63")
KERNEL(k512_120, 512, "v119")
KERNEL(k512_250, 512, "v249")
KERNEL(k1024_16, 1024, "v15")
KERNEL(k1024_64, 1024, "v63")
KERNEL(k1024_120, 1024, "v119")
static long long* d_out;
template <typename K>
static void run(const char* name, K kern, int T, int grid) {
    for (int rep = 0; rep < 3; rep++) { hipLaunchKernelGGL(kern, dim3(grid), dim3(T), 0, 0, d_out); hipDeviceSynchronize(); }
    const int W = T / 64;
    std::vector<long long> h(grid * W);
    hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<long long> spread, first;
    const long long gmin = *std::min_element(h.begin(), h.end());
    for (int b = 0; b < grid; b++) {
        auto mm = std::minmax_element(h.begin() + b * W, h.begin() + (b + 1) * W);
        spread.push_back(*mm.second - *mm.first);
        first.push_back(*mm.first - gmin);
    }
    std::sort(spread.begin(), spread.end()); std::sort(first.begin(), first.end());
    printf("%-12s grid %4d: last - first wave of a workgroup: median %5lld max %5lld cycles; first wave of a workgroup after the grid's first: median %5lld max %5lld\n",
           name, grid, spread[grid / 2], spread.back(), first[grid / 2], first.back());
}
int main() {
    hipMalloc(&d_out, 1024 * 16 * 8);
    for (int grid : {256, 32}) {
        run("256t/16v", k256_16, 256, grid); run("256t/120v", k256_120, 256, grid); run("256t/250v", k256_250, 256, grid);
        run("512t/16v", k512_16, 512, grid); run("512t/64v", k512_64, 512, grid); run("512t/120v", k512_120, 512, grid); run("512t/250v", k512_250, 512, grid);
        run("1024t/16v", k1024_16, 1024, grid); run("1024t/64v", k1024_64, 1024, grid); run("1024t/120v", k1024_120, 1024, grid);
    }
    return 0;
}
