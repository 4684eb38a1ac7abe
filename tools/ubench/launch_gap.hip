// launch_gap.hip — cost of back-to-back dependent kernel launches in one stream (eager and graph replay)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void tiny(float* p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1.f; }
__global__ __launch_bounds__(256) void wide(float* p, int spin) {
    __shared__ float s[256];
    float v = threadIdx.x;
    for (int i = 0; i < spin; ++i) v = v * 1.0001f + 0.5f;
    s[threadIdx.x] = v;
    __syncthreads();
    if (v == 12345.f) p[blockIdx.x] = s[0];
}
int main() {
    float* d; CK(hipMalloc(&d, 1 << 20));
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int N = 1000;
    for (int mode = 0; mode < 4; ++mode) {
        auto launch = [&]() {
            if (mode == 0) tiny<<<1, 64, 0, st>>>(d);
            else if (mode == 1) wide<<<512, 256, 0, st>>>(d, 0);
            else if (mode == 2) wide<<<512, 256, 40960, st>>>(d, 0);
            else wide<<<512, 256, 40960, st>>>(d, 2000);
        };
        for (int i = 0; i < 10; ++i) launch();
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < N; ++i) launch();
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        // graph of 100 launches, replayed 10 times
        hipGraph_t g; hipGraphExec_t ge;
        CK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
        for (int i = 0; i < 100; ++i) launch();
        CK(hipStreamEndCapture(st, &g));
        CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
        CK(hipGraphLaunch(ge, st));
        CK(hipStreamSynchronize(st));
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < 10; ++i) CK(hipGraphLaunch(ge, st));
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float msg; CK(hipEventElapsedTime(&msg, e0, e1));
        const char* names[] = {"tiny <<<1,64>>>", "wide <<<512,256>>>", "wide + 40 KB LDS", "wide + 40 KB LDS + ~8k cycles of work"};
        printf("%-40s eager %6.2f us/launch   graph %6.2f us/launch\n", names[mode], ms * 1e3 / N, msg * 1e3 / 1000);
    }
    return 0;
}
