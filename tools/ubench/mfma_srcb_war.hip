// mfma_srcb_war.hip — how long after issuing a v_mfma_f32_16x16x32_bf16 must a VALU write to its SrcA or SrcB wait (WHICH=0: B, 1: A)
// write to the SrcC registers wait on gfx950?  (hipcc 7.2 pads 2-3 wait states.)  For every (preceding MFMAs,
// wait states) the kernel overwrites SrcC with garbage after `K` wait states and checks D = A*B + C_original.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int NPRE, int K, int WHICH>
__global__ void k(const float* cin, float* dout) {
    const int lane = threadIdx.x;
    float c0 = cin[lane * 4 + 0], c1 = cin[lane * 4 + 1], c2 = cin[lane * 4 + 2], c3 = cin[lane * 4 + 3];
    float d0, d1, d2, d3;
    const unsigned one2 = 0x3f803f80u;   // two bf16 1.0
    asm volatile(
        "v_mov_b32 v100, %4\n v_mov_b32 v101, %5\n v_mov_b32 v102, %6\n v_mov_b32 v103, %7\n"
        "v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n"
        "v_mov_b32 v112, %8\n v_mov_b32 v113, %8\n v_mov_b32 v114, %8\n v_mov_b32 v115, %8\n"
        "v_mov_b32 v116, %8\n v_mov_b32 v117, %8\n v_mov_b32 v118, %8\n v_mov_b32 v119, %8\n"
        "s_nop 7\n s_nop 7\n"
        ".if %9 >= 1\n v_mfma_f32_16x16x32_bf16 v[108:111], v[112:115], v[116:119], v[108:111]\n .endif\n"
        ".if %9 >= 2\n v_mfma_f32_16x16x32_bf16 v[108:111], v[112:115], v[116:119], v[108:111]\n .endif\n"
        "v_mfma_f32_16x16x32_bf16 v[104:107], v[112:115], v[116:119], v[100:103]\n"
        ".if %10 > 0\n s_nop %10 - 1\n .endif\n"
        ".if %11 == 0\n v_mov_b32 v116, 0x7f007f00\n v_mov_b32 v117, 0x7f007f00\n v_mov_b32 v118, 0x7f007f00\n v_mov_b32 v119, 0x7f007f00\n .else\n v_mov_b32 v112, 0x7f007f00\n v_mov_b32 v113, 0x7f007f00\n v_mov_b32 v114, 0x7f007f00\n v_mov_b32 v115, 0x7f007f00\n .endif\n"
        "s_nop 15\n s_nop 15\n s_nop 15\n s_nop 15\n"
        "v_mov_b32 %0, v104\n v_mov_b32 %1, v105\n v_mov_b32 %2, v106\n v_mov_b32 %3, v107\n"
        : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
        : "v"(c0), "v"(c1), "v"(c2), "v"(c3), "v"(one2), "n"(NPRE), "n"(K), "n"(WHICH)
        : "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112",
          "v113", "v114", "v115", "v116", "v117", "v118", "v119");
    dout[lane * 4 + 0] = d0; dout[lane * 4 + 1] = d1; dout[lane * 4 + 2] = d2; dout[lane * 4 + 3] = d3;
}

float *dc, *dd;
float hc[256], hd[256];
template <int NPRE, int K, int WHICH>
void run() {
    k<NPRE, K, WHICH><<<1, 64>>>(dc, dd);
    CK(hipMemcpy(hd, dd, sizeof hd, hipMemcpyDeviceToHost));
    int bad = 0; unsigned long long cols = 0;
    for (int l = 0; l < 64; ++l)
        for (int i = 0; i < 4; ++i)
            if (hd[l * 4 + i] != 32.f + hc[l * 4 + i]) { ++bad; cols |= 1ull << (l & 15); }
    printf("%s overwritten, preceding MFMAs %d, wait states %2d: %3d wrong elements, columns mask 0x%04llx\n", WHICH ? "SrcA" : "SrcB", NPRE, K, bad, cols);
}
template <int NPRE, int WHICH> void sweep() {
    run<NPRE, 0, WHICH>(); run<NPRE, 1, WHICH>(); run<NPRE, 2, WHICH>(); run<NPRE, 3, WHICH>(); run<NPRE, 4, WHICH>(); run<NPRE, 5, WHICH>(); run<NPRE, 6, WHICH>();
    run<NPRE, 7, WHICH>(); run<NPRE, 8, WHICH>(); run<NPRE, 10, WHICH>(); run<NPRE, 12, WHICH>(); run<NPRE, 14, WHICH>(); run<NPRE, 16, WHICH>();
}
int main() {
    for (int i = 0; i < 256; ++i) hc[i] = (float)(i % 97);
    CK(hipMalloc(&dc, sizeof hc)); CK(hipMalloc(&dd, sizeof hd));
    CK(hipMemcpy(dc, hc, sizeof hc, hipMemcpyHostToDevice));
    sweep<0, 0>(); sweep<1, 0>(); sweep<2, 0>(); sweep<0, 1>(); sweep<2, 1>();
    return 0;
}
