// mfma_chain.hip — does the ORDER in which a wave walks its accumulators matter for v_mfma_f32_16x16x32_bf16?
// The stream convolution kernel issues, per input row, 3 dependent MFMAs (hi/lo terms) on each of up to 3
// accumulators (the three output rows a row fragment feeds): chains of 3, rotating over 3..8 accumulators.
// Alternative orders keep one accumulator for 9 (one kx, three ky) or 27 (a whole output row) MFMAs.
// Measured in shader cycles (s_memtime) per MFMA, operands from registers (18 weight fragments, 2 x fragments
// re-read from LDS every 9 MFMAs as in the kernel), 1 or 2 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 mfma_chain.hip -o mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// CHAIN: consecutive MFMAs on one accumulator before moving to the next (3, 9, 27); 8 accumulators, 216 MFMAs per
// iteration like one (item, chunk) step of the kernel
template <int CHAIN, bool LDSR>
__global__ __launch_bounds__(256, 2) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u + (i & 3);
    __syncthreads();
    f32x4 acc[8];
    for (int a = 0; a < 8; ++a) acc[a] = f32x4{0, 0, 0, 0};
    bf16x8 wh[9], wl[9];
    for (int j = 0; j < 9; ++j) {
        wh[j] = *reinterpret_cast<const bf16x8*>(lds + ((j * 2 + 0) * 64 + lane) * 16 % 32768);
        wl[j] = *reinterpret_cast<const bf16x8*>(lds + ((j * 2 + 1) * 64 + lane) * 16 % 32768);
    }
    const char* xr = lds + lane * 16;
    bf16x8 xh = *reinterpret_cast<const bf16x8*>(xr), xo = *reinterpret_cast<const bf16x8*>(xr + 1024);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 216; m += 3) {          // 72 (tap, row) triples
            const int trip = m / 3;
            // which accumulator and which weights: CHAIN/3 consecutive triples share an accumulator
            const int a = (trip / (CHAIN / 3)) % 8;
            const int j = trip % 9;
            if (LDSR && trip % 3 == 0) {            // a new row fragment every 3 triples (9 MFMAs), as in the kernel
                xh = *reinterpret_cast<const bf16x8*>(xr + ((it + trip) & 15) * 2048);
                xo = *reinterpret_cast<const bf16x8*>(xr + ((it + trip) & 15) * 2048 + 1024);
            }
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wl[j], xh, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xo, acc[a], 0, 0, 0);
            acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wh[j], xh, acc[a], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int a = 0; a < 8; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int CHAIN, bool LDSR>
void run(const char* name, int wgs_per_cu, float* out, unsigned long long* cyc) {
    const int iters = 400;
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<CHAIN, LDSR><<<grid, 256>>>(out, cyc, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<CHAIN, LDSR><<<grid, 256>>>(out, cyc, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[2048 * 4];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < grid * 4; ++i) sum += (double)h[i];
    const double per_wave = sum / (grid * 4) / ((double)iters * 216);          // cycles per MFMA of ONE wave
    const double per_simd = per_wave / wgs_per_cu;                              // per MFMA issued on the SIMD
    const double tf = (double)grid * 4 * iters * 216 * 16384 / (ms * 1e-3) / 1e12;
    printf("chain %2d %-8s waves/SIMD=%d  %7.3f ms  %7.1f TF/s  %6.2f cyc/MFMA/wave  %6.2f cyc/MFMA/SIMD\n", CHAIN,
           LDSR ? "lds-x" : "reg-x", wgs_per_cu, ms, tf, per_wave, per_simd);
}

int main() {
    float* out; hipMalloc(&out, 2048 * 256 * 4);
    unsigned long long* cyc; hipMalloc(&cyc, 2048 * 4 * 8);
    for (int w = 1; w <= 2; ++w) {
        run<3, false>("", w, out, cyc);
        run<9, false>("", w, out, cyc);
        run<27, false>("", w, out, cyc);
        run<3, true>("", w, out, cyc);
        run<9, true>("", w, out, cyc);
        run<27, true>("", w, out, cyc);
    }
    return 0;
}
