// x6_row.hip — what does one output row of conv_x6's inner loop cost?  18 dependent v_mfma_f32_16x16x32_bf16 on a fresh
// accumulator (C = 0 for the first), three ds_read_b128 for the next row, the VALU add of the previous row's sum; one wave
// per SIMD (256 threads, 1 workgroup per CU), 512-register budget.  Shader cycles per row (s_memtime), variants:
//   0  MFMA chain only, one running accumulator (never reset)
//   1  chain restarts from C = 0 every row, result added to acc[row] by VALU one row later
//   2  = 1 + three ds_read_b128 per row (operands really come from LDS)
//   3  = 2 with TWO interleaved chains of 9 (two fresh accumulators per row, summed by VALU)
//   4  = 2 with the chain's weights in AGPR-free form but 12 extra VALU per row (epilogue-like work)
//   hipcc --offload-arch=gfx950 -O3 x6_row.hip -o x6_row
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int V>
__global__ __launch_bounds__(256, 1) void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ __attribute__((aligned(16))) char lds[65536];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 16384; i += 256) reinterpret_cast<uint32_t*>(lds)[i] = 0x3c003c00u + (i & 3);
    __syncthreads();
    bf16x8 w[9][3];
    for (int j = 0; j < 27; ++j) w[j / 3][j % 3] = *reinterpret_cast<const bf16x8*>(lds + (j * 64 + lane) * 16 % 65536);
    f32x4 acc[16];
    for (int a = 0; a < 16; ++a) acc[a] = f32x4{0, 0, 0, 0};
    const char* xr = lds + lane * 16;
    bf16x8 xw[4][3];
    for (int j = 0; j < 4; ++j)
        for (int t = 0; t < 3; ++t) xw[j][t] = *reinterpret_cast<const bf16x8*>(xr + (j * 3 + t) * 1024);
    f32x4 tq[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
    f32x4 run = {0, 0, 0, 0};
    float extra[12];
    for (int i = 0; i < 12; ++i) extra[i] = (float)i;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            __builtin_amdgcn_sched_barrier(0);
            if (V >= 2) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    xw[(r + 3) % 4][t] = *reinterpret_cast<const bf16x8*>(xr + (((it + r) & 15) * 3 + t) * 1024);
            }
            if (V == 0) {
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int s = (r + ky) % 4;
                    run = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][2], xw[s][0], run, 0, 0, 0);
                    run = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][2], run, 0, 0, 0);
                    run = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][1], xw[s][1], run, 0, 0, 0);
                    run = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][1], xw[s][0], run, 0, 0, 0);
                    run = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][1], run, 0, 0, 0);
                    run = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][0], run, 0, 0, 0);
                }
            } else if (V == 3) {
                f32x4 s0 = {0, 0, 0, 0}, s1 = {0, 0, 0, 0};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int s = (r + ky) % 4;
                    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][2], xw[s][0], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][2], s1, 0, 0, 0);
                    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][1], xw[s][1], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][1], xw[s][0], s1, 0, 0, 0);
                    s0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][1], s0, 0, 0, 0);
                    s1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][0], s1, 0, 0, 0);
                }
                tq[r & 1] = s0 + s1;
            } else if (V >= 9) {
                // TWO rows at once: independent chains A (row r) and B (row r + 1) alternate, V - 9 VALU behind every MFMA
                if (r & 1) continue;
                f32x4 sa = {0, 0, 0, 0}, sb = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const int ky = i < 15 ? i / 5 : i - 15, pr = i < 15 ? i % 5 : 5;
                    const int wa[6] = {2, 0, 1, 1, 0, 0}, xb[6] = {0, 2, 1, 0, 1, 0};
                    sa = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][wa[pr]], xw[(r + ky) % 4][xb[pr]], sa, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < V - 9; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(extra[(i * 4 + j) % 12]) : "v"(extra[11 - j]));
                    __builtin_amdgcn_sched_barrier(0);
                    sb = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][wa[pr]], xw[(r + 1 + ky) % 4][xb[pr]], sb, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < V - 9; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(extra[(i * 4 + 2 + j) % 12]) : "v"(extra[11 - j]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                acc[r] += sa;
                acc[r + 1] += sb;
            } else if (V == 7 || V == 8) {
                // explicit placement: every MFMA is followed by V - 6 independent VALU instructions, fenced
                f32x4 sm = {0, 0, 0, 0};
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    const int ky = i < 15 ? i / 5 : i - 15, pr = i < 15 ? i % 5 : 5;
                    const int s = (r + ky) % 4;
                    const int wa[6] = {2, 0, 1, 1, 0, 0}, xb[6] = {0, 2, 1, 0, 1, 0};
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][wa[pr]], xw[s][xb[pr]], sm, 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < V - 6; ++j) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(extra[(i * 2 + j) % 10]) : "v"(extra[11 - j]));
                    __builtin_amdgcn_sched_barrier(0);
                }
                tq[r & 1] = sm;
            } else {
                f32x4 sm = {0, 0, 0, 0};
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int s = (r + ky) % 4;
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][2], xw[s][0], sm, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][2], sm, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][1], xw[s][1], sm, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][1], xw[s][0], sm, 0, 0, 0);
                    sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[s][1], sm, 0, 0, 0);
                }
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) sm = __builtin_amdgcn_mfma_f32_16x16x32_bf16(w[ky][0], xw[(r + ky) % 4][0], sm, 0, 0, 0);
                tq[r & 1] = sm;
            }
            if (V >= 1 && V < 9 && r > 0) {
                acc[r - 1] += tq[(r - 1) & 1];
                asm volatile("" : "+v"(acc[r - 1]));
            }
            if (V >= 4 && V <= 6) {
#pragma unroll
                for (int i = 0; i < 12; ++i) extra[i] = extra[i] * 1.0001f + acc[r][i & 3];
            }
            if (V == 5) {       // the same instructions, placed: MFMA, LDS read / VALU, MFMA, ...
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i >= 3) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
                }
            }
            if (V == 6) {
#pragma unroll
                for (int i = 0; i < 18; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (i < 3) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    if (i >= 8) __builtin_amdgcn_sched_group_barrier(0x002, 2, 0);
                }
            }
        }
        if (V >= 1 && V < 9) acc[15] += tq[1];
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = run[0] + run[1] + run[2] + run[3];
    for (int a = 0; a < 16; ++a) s += acc[a][0] + acc[a][1] + acc[a][2] + acc[a][3];
    for (int i = 0; i < 12; ++i) s += extra[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (lane == 0) cyc[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int V>
void run(const char* name, float* out, unsigned long long* cyc) {
    const int iters = 2000, grid = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<V><<<grid, 256>>>(out, cyc, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<V><<<grid, 256>>>(out, cyc, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    static unsigned long long h[1024];
    (void)hipMemcpy(h, cyc, sizeof(unsigned long long) * grid * 4, hipMemcpyDeviceToHost);
    double sum = 0;
    for (int i = 0; i < grid * 4; ++i) sum += (double)h[i];
    const double per_row = sum / (grid * 4) / ((double)iters * 16);
    const double tf = (double)grid * 4 * iters * 16 * 18 * 16384 / (ms * 1e-3) / 1e12;
    printf("%-58s %7.1f cycles/row (%5.2f per MFMA)  %7.1f TF/s bf16  clock %.2f GHz\n", name, per_row, per_row / 18, tf,
           per_row * iters * 16 / (ms * 1e-3) / 1e9);
}

int main() {
    float* out; unsigned long long* cyc;
    (void)hipMalloc(&out, 256 * 256 * 4);
    (void)hipMalloc(&cyc, 1024 * 8);
    run<0>("0 chain only, running accumulator", out, cyc);
    run<1>("1 fresh accumulator per row + VALU add", out, cyc);
    run<2>("2 = 1 + 3 ds_read_b128 per row", out, cyc);
    run<3>("3 = 2 with two interleaved chains of 9", out, cyc);
    run<4>("4 = 2 + 12 VALU per row", out, cyc);
    run<5>("5 = 4, sched_group_barrier: MFMA / 1 other alternating", out, cyc);
    run<6>("6 = 4, sched_group_barrier: 2 VALU behind MFMAs 8..17", out, cyc);
    run<7>("7 = 2 + ONE fma behind every MFMA (18 per row), fenced", out, cyc);
    run<8>("8 = 2 + TWO fma behind every MFMA (36 per row), fenced", out, cyc);
    run<9>("9 two rows at once (independent chains alternate), reads", out, cyc);
    run<10>("10 = 9 + ONE fma behind every MFMA (18 per row)", out, cyc);
    run<11>("11 = 9 + TWO fma behind every MFMA (36 per row)", out, cyc);
    return 0;
}
