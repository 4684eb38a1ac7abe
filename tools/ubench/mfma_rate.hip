// mfma_rate.hip — micro-benchmark: issue rate of v_mfma_f32_16x16x32_bf16 on gfx950 under the dependency and
// LDS-read patterns the convolution kernels use.  hipcc --offload-arch=gfx950 -O3 mfma_rate.hip -o mfma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// MODE 0: NACC independent accumulators, round-robin (dependency distance NACC)
// MODE 1: same + one ds_read_b128 pair per RPM MFMAs (operands consumed from LDS)
template <int NACC, int LDSR>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[32768];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0x3f803f80u;
    __syncthreads();
    f32x4 acc[NACC];
    for (int a = 0; a < NACC; ++a) acc[a] = f32x4{0, 0, 0, 0};
    bf16x8 w = *reinterpret_cast<const bf16x8*>(lds + lane * 16);
    bf16x8 x = *reinterpret_cast<const bf16x8*>(lds + 4096 + lane * 16);
    const char* xr = lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 24 / NACC; ++r) {
            if (LDSR) {
                // LDSR reads per 24 MFMAs
#pragma unroll
                for (int q = 0; q < LDSR * NACC / 24 + (LDSR * NACC < 24 ? (r % (24 / (LDSR * NACC)) == 0) : 0); ++q)
                    x = *reinterpret_cast<const bf16x8*>(xr + ((it * 7 + r * 3 + q) & 15) * 1024);
            }
#pragma unroll
            for (int a = 0; a < NACC; ++a) acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, x, acc[a], 0, 0, 0);
        }
    }
    float s = 0;
    for (int a = 0; a < NACC; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int LDSR>
void run(const char* name, int wgs_per_cu, float* out) {
    const int iters = getenv("ITERS") ? atoi(getenv("ITERS")) : 2000;
    const int grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, LDSR><<<grid, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC, LDSR><<<grid, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)grid * 4 * iters * 24;
    const double tf = mfma * 16384 / (ms * 1e-3) / 1e12;
    // cycles per MFMA per SIMD at 2.4 GHz
    const double cyc = (ms * 1e-3) * 2.4e9 / ((double)iters * 24 * wgs_per_cu);
    printf("%-34s wg/cu=%d  %8.3f ms  %7.1f TF/s  %5.2f cyc/mfma/SIMD\n", name, wgs_per_cu, ms, tf, cyc);
}

int main() {
    float* out; hipMalloc(&out, 256 * 4 * 256 * 4);
    for (int w = 1; w <= 2; ++w) {
        run<1, 0>("dep chain (1 acc)", w, out);
        run<2, 0>("2 acc", w, out);
        run<3, 0>("3 acc", w, out);
        run<4, 0>("4 acc", w, out);
        run<8, 0>("8 acc", w, out);
        run<4, 4>("4 acc + 4 ds_read/24", w, out);
        run<4, 8>("4 acc + 8 ds_read/24", w, out);
        run<4, 12>("4 acc + 12 ds_read/24", w, out);
        run<4, 24>("4 acc + 24 ds_read/24", w, out);
    }
    return 0;
}
