// mfma_raw_contended.hip — is the software-visible distance "MFMA -> VALU read of its result" (7 wait states, what
// hipcc pads) still enough when OTHER waves keep the SIMD's matrix pipe busy?  Wave 0 repeats the minimal sequence
// MFMA; K wait states; read, the other 7 waves of the workgroup (two waves per SIMD) issue MFMAs back to back.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int K, int FILL>
__global__ __launch_bounds__(512) void k(int* bad_out, int iters, int spam) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (wave != 0) {
        if (!spam) return;
        const unsigned one2 = 0x3f803f80u;
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        u4 ones = {one2, one2, one2, one2};
        bf16x8 a = __builtin_bit_cast(bf16x8, ones);
        f32x4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
        for (int i = 0; i < iters * 8; ++i) {
            acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, a, acc2, 0, 0, 0);
        }
        if (acc[0] + acc2[0] == 12345.f) bad_out[1] = 1;
        return;
    }
    int bad = 0;
    const unsigned one2 = 0x3f803f80u;
    for (int it = 0; it < iters; ++it) {
        float c0 = (float)(lane + it % 7), d0, d1, d2, d3;
        asm volatile(
            "v_mov_b32 v100, %4\n v_mov_b32 v101, %4\n v_mov_b32 v102, %4\n v_mov_b32 v103, %4\n"
            "v_mov_b32 v112, %5\n v_mov_b32 v113, %5\n v_mov_b32 v114, %5\n v_mov_b32 v115, %5\n"
            "v_mov_b32 v116, %5\n v_mov_b32 v117, %5\n v_mov_b32 v118, %5\n v_mov_b32 v119, %5\n"
            "s_nop 7\n s_nop 7\n"
            "v_mfma_f32_16x16x32_bf16 v[100:103], v[112:115], v[116:119], v[100:103]\n"
            ".if %7 == 0\n .if %6 > 0\n s_nop %6 - 1\n .endif\n .endif\n"
            ".if %7 == 1\n .rept %6\n v_mov_b32 v124, v125\n .endr\n .endif\n"
            "v_mov_b32 %0, v100\n v_mov_b32 %1, v101\n v_mov_b32 %2, v102\n v_mov_b32 %3, v103\n"
            "s_nop 15\n s_nop 15\n s_nop 15\n"
            : "=v"(d0), "=v"(d1), "=v"(d2), "=v"(d3)
            : "v"(c0), "v"(one2), "n"(K), "n"(FILL)
            : "v100", "v101", "v102", "v103", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v124", "v125");
        const float e = c0 + 32.f;
        bad += (d0 != e) + (d1 != e) + (d2 != e) + (d3 != e);
    }
    atomicAdd(bad_out, bad);
}

int* dbad;
template <int K, int FILL>
void run(int spam) {
    CK(hipMemset(dbad, 0, 8));
    k<K, FILL><<<1, 512>>>(dbad, 20000, spam);
    int h[2]; CK(hipMemcpy(h, dbad, 8, hipMemcpyDeviceToHost));
    printf("  K=%2d: %8d", K, h[0]);
}
template <int FILL> void sweep(const char* what, int spam) {
    printf("%s, other waves %s (wrong element-reads out of %d):\n", what, spam ? "issuing MFMAs" : "idle", 20000 * 256);
    run<5, FILL>(spam); run<6, FILL>(spam); run<7, FILL>(spam); run<8, FILL>(spam); run<9, FILL>(spam); run<10, FILL>(spam); printf("\n");
    run<11, FILL>(spam); run<12, FILL>(spam); run<14, FILL>(spam); run<16, FILL>(spam); run<20, FILL>(spam); run<24, FILL>(spam); printf("\n");
}
int main() {
    CK(hipMalloc(&dbad, 8));
    sweep<0>("s_nop wait states", 0);
    sweep<0>("s_nop wait states", 1);
    sweep<1>("independent v_mov_b32", 0);
    sweep<1>("independent v_mov_b32", 1);
    return 0;
}
