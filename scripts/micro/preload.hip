// Calibration: does kernarg preloading (-mllvm -amdgpu-kernarg-preload-count) shorten a dependent-launch chain?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k2(const float* p, float* q, int n) { q[threadIdx.x + blockIdx.x * 64] = p[threadIdx.x + blockIdx.x * 64] + n; }
int main() {
    hipStream_t st; hipStreamCreate(&st);
    float *a, *b; hipMalloc(&a, 1 << 20); hipMalloc(&b, 1 << 20); hipMemset(a, 0, 1 << 20);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(st, hipStreamCaptureModeGlobal);
    for (int i = 0; i < 200; i++) { hipLaunchKernelGGL(k2, dim3(256), dim3(64), 0, st, a, b, i); hipLaunchKernelGGL(k2, dim3(256), dim3(64), 0, st, b, a, i); }
    hipStreamEndCapture(st, &g); hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    hipGraphLaunch(ge, st); hipStreamSynchronize(st);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, st);
    for (int r = 0; r < 5; r++) hipGraphLaunch(ge, st);
    hipEventRecord(e1, st); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%.3f us per launch\n", ms * 1e3f / (5 * 400));
    return 0;
}
