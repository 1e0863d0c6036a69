// Probe: how does v_mfma_f32_16x16x32_f16 round?  D = A (16x32) * B (32x16) + 0 on inputs whose 32 products span many
// exponents, compared on the host with (a) the exactly rounded sum, (b) a sequential fp32 chain.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 f16;
typedef f16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const f16* A, const f16* B, float* D) {   // A [16][32] row-major, B [16 cols][32] (col-major k), D [16][16]
    const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
    f16x8 a = *reinterpret_cast<const f16x8*>(A + r * 32 + g * 8);
    f16x8 b = *reinterpret_cast<const f16x8*>(B + r * 32 + g * 8);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    for (int i = 0; i < 4; i++) D[(4 * g + i) * 16 + r] = acc[i];   // row 4g+i (A row), col r (B col)
}
int main() {
    srand(1);
    std::vector<f16> A(512), B(512);
    int bad_exact = 0, bad_seq = 0, total = 0;
    f16 *dA, *dB; float* dD; hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dD, 1024);
    for (int trial = 0; trial < 2000; trial++) {
        for (int i = 0; i < 512; i++) {
            A[i] = (f16)((rand() & 1) ? 1.0f : -1.0f);                       // Hadamard-like signs
            float mag = ldexpf(1.0f + (rand() % 1024) / 1024.0f, (rand() % 22) - 14);   // wide exponent spread
            B[i] = (f16)((rand() & 1) ? mag : -mag);
            if ((i & 31) >= 28) { A[i] = (f16)0.0f; }                        // K = 28 used
        }
        hipMemcpy(dA, A.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD);
        float D[256]; hipMemcpy(D, dD, 1024, hipMemcpyDeviceToHost);
        for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
            double ex = 0; float seq = 0.f;
            for (int kk = 0; kk < 32; kk++) { double p = (double)(float)A[m * 32 + kk] * (double)(float)B[n * 32 + kk]; ex += p; seq = fmaf((float)A[m * 32 + kk], (float)B[n * 32 + kk], seq); }
            float exr = (float)ex;
            total++;
            if (D[m * 16 + n] != exr) bad_exact++;
            if (D[m * 16 + n] != seq) bad_seq++;
        }
    }
    printf("outputs %d: differ from exactly-rounded sum %d, differ from sequential fp32 chain %d\n", total, bad_exact, bad_seq);
    return 0;
}
