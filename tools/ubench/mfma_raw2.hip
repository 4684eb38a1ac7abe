// mfma_raw.hip — wait states a VALU read of a v_mfma_f32_16x16x32_bf16 result needs on gfx950, with the matrix pipe
// idle or busy (NPRE MFMAs issued right before the producer) and with NMID independent MFMAs between the producer and
// the read.  hipcc 7.2 counts every instruction, MFMAs included, as one wait state.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NPRE, int NMID, int K, int FILL>
__global__ void k(const float* cin, float* dout) {
    const int lane = threadIdx.x;
    float c0 = cin[lane * 4 + 0], c1 = cin[lane * 4 + 1], c2 = cin[lane * 4 + 2], c3 = cin[lane * 4 + 3];
    float d0, d1, d2, d3;
    const unsigned one2 = 0x3f803f80u;   // two bf16 1.0
    asm volatile(
        "v_mov_b32 v100, %4\n v_mov_b32 v101, %5\n v_mov_b32 v102, %6\n v_mov_b32 v103, %7\n"
        "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n"
        "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n"
        "v_mov_b32 v112, %8\n v_mov_b32 v113, %8\n v_mov_b32 v114, %8\n v_mov_b32 v115, %8\n"
        "v_mov_b32 v116, %8\n v_mov_b32 v117, %8\n v_mov_b32 v118, %8\n v_mov_b32 v119, %8\n"
        "s_nop 7\n s_nop 7\n"
        ".rept %9\n v_mfma_f32_16x16x32_bf16 v[108:111], v[112:115], v[116:119], v[108:111]\n .endr\n"
        "v_mfma_f32_16x16x32_bf16 v[100:103], v[112:115], v[116:119], v[100:103]\n"       // producer: c += 32
        ".rept %10\n v_mfma_f32_16x16x32_bf16 v[120:123], v[112:115], v[116:119], v[120:123]\n .endr\n"
        ".if %12 == 0\n .if %11 > 0\n s_nop %11 - 1\n .endif\n .endif\n"
        ".if %12 == 1\n .rept %11\n v_mov_b32 v124, v125\n .endr\n .endif\n"
        ".if %12 == 2\n .rept %11\n s_mov_b32 s20, s21\n .endr\n .endif\n"
        ".if %12 == 3\n .rept %11\n v_cmp_lt_i32 vcc, s20, v125\n .endr\n .endif\n"
        ".if %12 == 4\n .rept %11\n s_and_b64 s[22:23], exec, s[24:25]\n .endr\n .endif\n"
        ".if %12 == 5\n .rept %11\n v_cndmask_b32 v124, v125, v126, vcc\n .endr\n .endif\n"
        "v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n v_mov_b32 %2, v102\n v_mov_b32 %3, v103\n"
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(one2), "n"(NPRE), "n"(NMID), "n"(K), "n"(FILL)
        : "v100", "v101", "v102", "v103", "v108", "v109", "v110", "v111", "v112",
          "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "s20", "s21", "s22", "s23", "s24", "s25", "vcc");
    dout[lane * 4 + 0] = d0; dout[lane * 4 + 1] = d1; dout[lane * 4 + 2] = d2; dout[lane * 4 + 3] = d3;
}

float *dc, *dd;
float hc[256], hd[256];
template <int NPRE, int NMID, int K, int FILL>
int run() {
    k<NPRE, NMID, K, FILL><<<1, 64>>>(dc, dd);
    CK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
    int bad = 0; unsigned long long cols = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i)
            if (hd[l * 4 + i] != 32.f + hc[l * 4 + i]) { ++bad; cols |= 1ull << (l & 15); }
    printf("  K=%2d: %3d wrong (cols 0x%04llx)", K, bad, cols);
    return bad;
}
template <int FILL> void sweep(const char* what) {
    printf("%s between the MFMA and the read of its result:\n", what);
    run<0, 0, 0, FILL>(); run<0, 0, 1, FILL>(); run<0, 0, 2, FILL>(); run<0, 0, 3, FILL>(); printf("\n");
    run<0, 0, 4, FILL>(); run<0, 0, 5, FILL>(); run<0, 0, 6, FILL>(); run<0, 0, 7, FILL>(); printf("\n");
    run<0, 0, 8, FILL>(); run<0, 0, 10, FILL>(); run<0, 0, 12, FILL>(); run<0, 0, 16, FILL>(); printf("\n");
}
int main() {
    for (int i = 0; i < 256; ++i) hc[i] = (float)(i % 97);
    CK(hipMalloc(&dc, sizeof hc)); CK(hipMalloc(&dd, sizeof hd));
    CK(hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice));
    sweep<0>("s_nop wait states"); sweep<1>("independent v_mov_b32"); sweep<2>("s_mov_b32"); sweep<3>("v_cmp_lt_i32 (VOPC)");
    sweep<4>("s_and_b64"); sweep<5>("v_cndmask_b32");
    return 0;
}
