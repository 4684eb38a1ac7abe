// Do independent branches of a HIP graph overlap on gfx950 / ROCm 7.2?  Two chains of small kernels (64 work-groups, ~20 us
// each) are run (a) serially in one stream, (b) captured with a fork/join over two streams, (c) composed from per-kernel
// child graphs with explicit edges.   hipcc -O2 --offload-arch=gfx950 graph_branches.hip -o graph_branches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s -> %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); return 1; } } while (0)
__global__ void spin(float* p, int iters) {
    float v = p[blockIdx.x * blockDim.x + threadIdx.x];
    for (int i = 0; i < iters; ++i) v = v * 1.0001f + 0.5f;
    p[blockIdx.x * blockDim.x + threadIdx.x] = v;
}
int main() {
    float *a, *b; OK(hipMalloc(&a, 1 << 20)); OK(hipMalloc(&b, 1 << 20));
    OK(hipMemset(a, 0, 1 << 20)); OK(hipMemset(b, 0, 1 << 20));
    hipStream_t s0, s1; OK(hipStreamCreateWithFlags(&s0, hipStreamNonBlocking)); OK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    hipEvent_t fork, join, t0, t1; OK(hipEventCreateWithFlags(&fork, hipEventDisableTiming)); OK(hipEventCreateWithFlags(&join, hipEventDisableTiming));
    OK(hipEventCreate(&t0)); OK(hipEventCreate(&t1));
    const int L = 10, IT = 20000;
    auto timeit = [&](hipGraphExec_t ex, const char* name) -> int {
        OK(hipGraphLaunch(ex, s0)); OK(hipStreamSynchronize(s0));
        OK(hipEventRecord(t0, s0));
        for (int i = 0; i < 20; ++i) OK(hipGraphLaunch(ex, s0));
        OK(hipEventRecord(t1, s0)); OK(hipStreamSynchronize(s0));
        float ms; OK(hipEventElapsedTime(&ms, t0, t1)); printf("%-28s %.1f us per graph\n", name, ms / 20 * 1e3); fflush(stdout);
        return 0;
    };
    hipGraph_t g; hipGraphExec_t ex;
    // (a) serial
    OK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < L; ++i) { spin<<<64, 256, 0, s0>>>(a, IT); spin<<<64, 256, 0, s0>>>(b, IT); }
    OK(hipStreamEndCapture(s0, &g)); OK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    if (timeit(ex, "serial capture")) return 1;
    // (b) fork/join capture
    OK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
    OK(hipEventRecord(fork, s0)); OK(hipStreamWaitEvent(s1, fork, 0));
    for (int i = 0; i < L; ++i) { spin<<<64, 256, 0, s0>>>(a, IT); spin<<<64, 256, 0, s1>>>(b, IT); }
    OK(hipEventRecord(join, s1)); OK(hipStreamWaitEvent(s0, join, 0));
    OK(hipStreamEndCapture(s0, &g)); OK(hipGraphInstantiate(&ex, g, nullptr, nullptr, 0));
    if (timeit(ex, "fork/join capture")) return 1;
    // (c) child graphs with explicit edges
    hipGraph_t main_g; OK(hipGraphCreate(&main_g, 0));
    hipGraphNode_t prev[2] = {nullptr, nullptr};
    for (int i = 0; i < L; ++i)
        for (int c = 0; c < 2; ++c) {
            hipGraph_t child;
            OK(hipStreamBeginCapture(s0, hipStreamCaptureModeThreadLocal));
            spin<<<64, 256, 0, s0>>>(c ? b : a, IT);
            OK(hipStreamEndCapture(s0, &child));
            hipGraphNode_t node;
            OK(hipGraphAddChildGraphNode(&node, main_g, prev[c] ? &prev[c] : nullptr, prev[c] ? 1 : 0, child));
            prev[c] = node;
        }
    OK(hipGraphInstantiate(&ex, main_g, nullptr, nullptr, 0));
    if (timeit(ex, "child graphs, two chains")) return 1;
    return 0;
}
