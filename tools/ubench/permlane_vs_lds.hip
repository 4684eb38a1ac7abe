// permlane_vs_lds.hip — does v_permlane16_swap_b32 disturb (or get disturbed by) ds_read_b128 data that is still
// in flight?  Wave: N ds_read_b128 issued, then a permlane swap on unrelated registers, then wait and compare both.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NREAD, int GAP>
__global__ __launch_bounds__(256) void k(int* bad_out, int iters) {
    __shared__ __attribute__((aligned(16))) unsigned lds[4 * 64 * 4 * 4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4 * 64 * 4 * 4; i += 256) lds[i] = 0x1000000u + i;
    __syncthreads();
    int bad_r = 0, bad_p = 0;
    const unsigned base = (unsigned)(size_t)(lds) + wave * 4096 + lane * 16;   // LDS byte address (low 32 bits of the generic pointer)
    for (int it = 0; it < iters; ++it) {
        unsigned r[16];
        unsigned pa = lane * 3 + it, pb = lane * 5 + 1000 + it;
        asm volatile(
            "v_mov_b32 v120, %16\n v_mov_b32 v121, %17\n"
            "s_nop 4\n"
            "ds_read_b128 v[100:103], %18\n"
            ".if %19 >= 2\n ds_read_b128 v[104:107], %18 offset:1024\n .endif\n"
            ".if %19 >= 3\n ds_read_b128 v[108:111], %18 offset:2048\n .endif\n"
            ".if %19 >= 4\n ds_read_b128 v[112:115], %18 offset:3072\n .endif\n"
            ".if %20 > 0\n s_nop %20 - 1\n .endif\n"
            "v_permlane16_swap_b32 v120, v121\n"
            "s_waitcnt lgkmcnt(0)\n s_nop 4\n"
            "v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n v_mov_b32 %2, v102\n v_mov_b32 %3, v103\n"
            "v_mov_b32 %4, v104\n v_mov_b32 %5, v105\n v_mov_b32 %6, v106\n v_mov_b32 %7, v107\n"
            "v_mov_b32 %8, v108\n v_mov_b32 %9, v109\n v_mov_b32 %10, v110\n v_mov_b32 %11, v111\n"
            "v_mov_b32 %12, v112\n v_mov_b32 %13, v113\n v_mov_b32 %14, v120\n v_mov_b32 %15, v121\n"
            : "=v"(r[0]), "=v"(r[1]), "=v"(r[2]), "=v"(r[3]), "=v"(r[4]), "=v"(r[5]), "=v"(r[6]), "=v"(r[7]),
              "=v"(r[8]), "=v"(r[9]), "=v"(r[10]), "=v"(r[11]), "=v"(r[12]), "=v"(r[13]), "=v"(r[14]), "=v"(r[15])
            : "v"(pa), "v"(pb), "v"(base), "n"(NREAD), "n"(GAP)
            : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111",
              "v112", "v113", "v114", "v115", "v120", "v121", "memory");
        const int w0 = (wave * 4096 + lane * 16) / 4;
        for (int q = 0; q < NREAD && q < 3; ++q)
            for (int j = 0; j < 4; ++j) bad_r += r[q * 4 + j] != 0x1000000u + w0 + q * 256 + j;
        if (NREAD >= 4) { bad_r += r[12] != 0x1000000u + w0 + 768; bad_r += r[13] != 0x1000000u + w0 + 769; }
        // v_permlane16_swap vdst, src0: odd rows of vdst <-> even rows of src0 (rows of 16 lanes)
        const int row = lane >> 4;
        const unsigned ea = (row & 1) ? (unsigned)((lane - 16) * 5 + 1000 + it) : pa;     // vdst: odd rows take src0's even row below
        const unsigned eb = (row & 1) ? pb : (unsigned)((lane + 16) * 3 + it);            // src0: even rows take vdst's odd row above
        bad_p += (r[14] != ea) + (r[15] != eb);
    }
    atomicAdd(bad_out, bad_r);
    atomicAdd(bad_out + 1, bad_p);
}
int* dbad;
template <int NREAD, int GAP> void run() {
    CK(hipMemset(dbad, 0, 8));
    k<NREAD, GAP><<<64, 256>>>(dbad, 2000);
    int h[2]; CK(hipMemcpy(h, dbad, 8, hipMemcpyDeviceToHost));
    printf("reads in flight %d, gap %2d wait states: wrong LDS data %8d, wrong swap results %8d\n", NREAD, GAP, h[0], h[1]);
}
int main() {
    CK(hipMalloc(&dbad, 8));
    run<1, 0>(); run<2, 0>(); run<4, 0>(); run<4, 2>(); run<4, 4>(); run<4, 8>(); run<4, 16>(); run<1, 8>(); run<0, 0>();
    return 0;
}
