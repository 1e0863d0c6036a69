// Calibration: SIMD throughput per VALU instruction class on gfx950.  1024 independent-enough instructions per wave
// (8 rotating destinations), 1 / 2 / 4 waves per SIMD; cycles per instruction PER WAVE (s_memtime).  If a class costs the
// SIMD 4 cycles per wave64 instruction, two waves on a SIMD take 8 cycles per instruction each; if it costs 2 (fp32
// fma / add / mul at 256 flop/clk/CU), two waves still see 4.
// hipcc --offload-arch=gfx950 -O3 -o valu_rate.bin valu_rate.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

#define KERNEL(NAME, I0, I1, I2, I3, I4, I5, I6, I7)                                                              \
    __global__ __launch_bounds__(1024) void NAME(long long* out, float m, float c, float* sink) {                 \
        float a0 = threadIdx.x, a1 = 1, a2 = 2, a3 = 3, a4 = 4, a5 = 5, a6 = 6, a7 = 7;                            \
        float b0 = 1, b1 = 2, b2 = 3, b3 = 4, b4 = 5, b5 = 6, b6 = 7, b7 = 8;                                      \
        long long t0, t1;                                                                                         \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                \
        asm volatile(".rept 128\n\t" I0 "\n\t" I1 "\n\t" I2 "\n\t" I3 "\n\t" I4 "\n\t" I5 "\n\t" I6 "\n\t" I7 "\n\t.endr" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7),            \
                       "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)             \
                     : "v"(m), "v"(c));                                                                           \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 + b4 + b5 + b6 + b7 == 12345.f) sink[0] = 1; \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;         \
    }
// operands: %0..%7 = a0..a7, %8..%15 = b0..b7 (a_i, b_i consecutive pairs are NOT guaranteed: packed ops use explicit pairs via v[..] only through the compiler, so packed forms take 64-bit operands built from two asm operands is impossible -> use "+v" of double-width types below)
KERNEL(k_fma, "v_fma_f32 %0, %0, %16, %17", "v_fma_f32 %1, %1, %16, %17", "v_fma_f32 %2, %2, %16, %17", "v_fma_f32 %3, %3, %16, %17",
       "v_fma_f32 %4, %4, %16, %17", "v_fma_f32 %5, %5, %16, %17", "v_fma_f32 %6, %6, %16, %17", "v_fma_f32 %7, %7, %16, %17")
KERNEL(k_add, "v_add_f32 %0, %0, %16", "v_add_f32 %1, %1, %16", "v_add_f32 %2, %2, %16", "v_add_f32 %3, %3, %16",
       "v_add_f32 %4, %4, %16", "v_add_f32 %5, %5, %16", "v_add_f32 %6, %6, %16", "v_add_f32 %7, %7, %16")
KERNEL(k_mul, "v_mul_f32 %0, %0, %16", "v_mul_f32 %1, %1, %16", "v_mul_f32 %2, %2, %16", "v_mul_f32 %3, %3, %16",
       "v_mul_f32 %4, %4, %16", "v_mul_f32 %5, %5, %16", "v_mul_f32 %6, %6, %16", "v_mul_f32 %7, %7, %16")
KERNEL(k_max, "v_max_f32 %0, %0, %16", "v_max_f32 %1, %1, %16", "v_max_f32 %2, %2, %16", "v_max_f32 %3, %3, %16",
       "v_max_f32 %4, %4, %16", "v_max_f32 %5, %5, %16", "v_max_f32 %6, %6, %16", "v_max_f32 %7, %7, %16")
KERNEL(k_max3, "v_max3_f32 %0, %0, %16, %17", "v_max3_f32 %1, %1, %16, %17", "v_max3_f32 %2, %2, %16, %17", "v_max3_f32 %3, %3, %16, %17",
       "v_max3_f32 %4, %4, %16, %17", "v_max3_f32 %5, %5, %16, %17", "v_max3_f32 %6, %6, %16, %17", "v_max3_f32 %7, %7, %16, %17")
KERNEL(k_and, "v_and_b32 %0, %0, %16", "v_and_b32 %1, %1, %16", "v_and_b32 %2, %2, %16", "v_and_b32 %3, %3, %16",
       "v_and_b32 %4, %4, %16", "v_and_b32 %5, %5, %16", "v_and_b32 %6, %6, %16", "v_and_b32 %7, %7, %16")
KERNEL(k_lshl_or, "v_lshl_or_b32 %0, %0, 4, %16", "v_lshl_or_b32 %1, %1, 4, %16", "v_lshl_or_b32 %2, %2, 4, %16", "v_lshl_or_b32 %3, %3, 4, %16",
       "v_lshl_or_b32 %4, %4, 4, %16", "v_lshl_or_b32 %5, %5, 4, %16", "v_lshl_or_b32 %6, %6, 4, %16", "v_lshl_or_b32 %7, %7, 4, %16")
KERNEL(k_cvt_f32_f16, "v_cvt_f32_f16 %0, %8", "v_cvt_f32_f16 %1, %9", "v_cvt_f32_f16 %2, %10", "v_cvt_f32_f16 %3, %11",
       "v_cvt_f32_f16 %4, %12", "v_cvt_f32_f16 %5, %13", "v_cvt_f32_f16 %6, %14", "v_cvt_f32_f16 %7, %15")
KERNEL(k_cvt_f16_f32, "v_cvt_f16_f32 %0, %8", "v_cvt_f16_f32 %1, %9", "v_cvt_f16_f32 %2, %10", "v_cvt_f16_f32 %3, %11",
       "v_cvt_f16_f32 %4, %12", "v_cvt_f16_f32 %5, %13", "v_cvt_f16_f32 %6, %14", "v_cvt_f16_f32 %7, %15")
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %8, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %1, %9, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
       "v_add_f32_dpp %2, %10, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %3, %11, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
       "v_add_f32_dpp %4, %12, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %5, %13, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
       "v_add_f32_dpp %6, %14, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_add_f32_dpp %7, %15, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %1, %9 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
       "v_mov_b32_dpp %2, %10 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %3, %11 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
       "v_mov_b32_dpp %4, %12 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %5, %13 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf",
       "v_mov_b32_dpp %6, %14 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf", "v_mov_b32_dpp %7, %15 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf")
KERNEL(k_pk_add_f16, "v_pk_add_f16 %0, %0, %16", "v_pk_add_f16 %1, %1, %16", "v_pk_add_f16 %2, %2, %16", "v_pk_add_f16 %3, %3, %16",
       "v_pk_add_f16 %4, %4, %16", "v_pk_add_f16 %5, %5, %16", "v_pk_add_f16 %6, %6, %16", "v_pk_add_f16 %7, %7, %16")
KERNEL(k_rcp, "v_rcp_f32 %0, %8", "v_rcp_f32 %1, %9", "v_rcp_f32 %2, %10", "v_rcp_f32 %3, %11",
       "v_rcp_f32 %4, %12", "v_rcp_f32 %5, %13", "v_rcp_f32 %6, %14", "v_rcp_f32 %7, %15")
KERNEL(k_cndmask, "v_cndmask_b32 %0, %0, %16, vcc", "v_cndmask_b32 %1, %1, %16, vcc", "v_cndmask_b32 %2, %2, %16, vcc", "v_cndmask_b32 %3, %3, %16, vcc",
       "v_cndmask_b32 %4, %4, %16, vcc", "v_cndmask_b32 %5, %5, %16, vcc", "v_cndmask_b32 %6, %6, %16, vcc", "v_cndmask_b32 %7, %7, %16, vcc")

// packed fp32: 64-bit operands
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define KERNEL2(NAME, OP)                                                                                          \
    __global__ __launch_bounds__(1024) void NAME(long long* out, float m, float c, float* sink) {                  \
        f32x2 a0 = {(float)threadIdx.x, 1}, a1 = {1, 2}, a2 = {2, 3}, a3 = {3, 4}, a4 = {4, 5}, a5 = {5, 6}, a6 = {6, 7}, a7 = {7, 8}; \
        f32x2 mm = {m, m}, cc = {c, c};                                                                            \
        long long t0, t1;                                                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                                 \
        asm volatile(".rept 128\n\t" OP " %0, %0, %8" "\n\t" OP " %1, %1, %8" "\n\t" OP " %2, %2, %8" "\n\t" OP " %3, %3, %8" "\n\t" \
                     OP " %4, %4, %8" "\n\t" OP " %5, %5, %8" "\n\t" OP " %6, %6, %8" "\n\t" OP " %7, %7, %8" "\n\t.endr" \
                     : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)              \
                     : "v"(mm), "v"(cc));                                                                          \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                                 \
        f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;                                                           \
        if (s[0] + s[1] == 12345.f) sink[0] = 1;                                                                   \
        if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;          \
    }
KERNEL2(k_pk_add, "v_pk_add_f32")
KERNEL2(k_pk_mul, "v_pk_mul_f32")
__global__ __launch_bounds__(1024) void k_pk_fma(long long* out, float m, float c, float* sink) {
    f32x2 a0 = {(float)threadIdx.x, 1}, a1 = {1, 2}, a2 = {2, 3}, a3 = {3, 4}, a4 = {4, 5}, a5 = {5, 6}, a6 = {6, 7}, a7 = {7, 8};
    f32x2 mm = {m, m}, cc = {c, c};
    long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    asm volatile(".rept 128\n\tv_pk_fma_f32 %0, %0, %8, %9\n\tv_pk_fma_f32 %1, %1, %8, %9\n\tv_pk_fma_f32 %2, %2, %8, %9\n\tv_pk_fma_f32 %3, %3, %8, %9\n\t"
                 "v_pk_fma_f32 %4, %4, %8, %9\n\tv_pk_fma_f32 %5, %5, %8, %9\n\tv_pk_fma_f32 %6, %6, %8, %9\n\tv_pk_fma_f32 %7, %7, %8, %9\n\t.endr"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
                 : "v"(mm), "v"(cc));
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    f32x2 s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s[0] + s[1] == 12345.f) sink[0] = 1;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

static long long* d_out;
static float* d_sink;
template <typename K>
static void run(const char* name, K kern) {
    printf("%-16s", name);
    for (int threads : {256, 512, 1024}) {
        const int waves = threads / 64;
        for (int rep = 0; rep < 2; rep++) {
            hipLaunchKernelGGL(kern, dim3(256), dim3(threads), 0, 0, d_out, 1.0001f, 0.5f, d_sink);
            hipDeviceSynchronize();
        }
        std::vector<long long> h(256 * waves);
        hipMemcpy(h.data(), d_out, h.size() * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  %2d waves/CU: %5.2f (max %5.2f)", waves, h[h.size() / 2] / 1024.0, h.back() / 1024.0);
    }
    printf("   cycles per instruction per wave\n");
}

int main() {
    hipMalloc(&d_out, 256 * 16 * 8);
    hipMalloc(&d_sink, 64);
    run("v_fma_f32", k_fma);
    run("v_add_f32", k_add);
    run("v_mul_f32", k_mul);
    run("v_max_f32", k_max);
    run("v_max3_f32", k_max3);
    run("v_and_b32", k_and);
    run("v_lshl_or_b32", k_lshl_or);
    run("v_cndmask_b32", k_cndmask);
    run("v_cvt_f32_f16", k_cvt_f32_f16);
    run("v_cvt_f16_f32", k_cvt_f16_f32);
    run("v_add_f32_dpp", k_add_dpp);
    run("v_mov_b32_dpp", k_mov_dpp);
    run("v_pk_add_f16", k_pk_add_f16);
    run("v_rcp_f32", k_rcp);
    run("v_pk_add_f32", k_pk_add);
    run("v_pk_mul_f32", k_pk_mul);
    run("v_pk_fma_f32", k_pk_fma);
    return 0;
}
