// anyorder.hip — does hipExtAnyOrderLaunch let two kernels of one stream overlap on gfx950?
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ __launch_bounds__(256) void spin(float* p, int iters) {
    float v = threadIdx.x;
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    if (v == 12345.f) p[blockIdx.x] = v;
}
int main() {
    float* d; CK(hipMalloc(&d, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int flags = 0; flags < 2; ++flags) {
        for (int wgs : {128, 256}) {
            // warm
            hipExtLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, st, nullptr, nullptr, 0, d, 1000);
            CK(hipStreamSynchronize(st));
            CK(hipEventRecord(e0, st));
            for (int i = 0; i < 8; ++i)
                hipExtLaunchKernelGGL(spin, dim3(wgs), dim3(256), 0, st, nullptr, nullptr, (i & 3) ? flags : 0, d, 20000);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            printf("flags=%d  8 kernels of %d WGs (256 CUs): %.1f us total\n", flags, wgs, ms * 1e3);
        }
    }
    return 0;
}
